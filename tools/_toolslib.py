"""The TOOLS build of the library (`make tools`: -DEC_TOOLS -DEC_CONV8_PROFILE): the product library plus the profiling
exports that are not part of include/ec_amd.h.  Import this BEFORE anything loads embodied_clip_amd._lib."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  FIRST: the library must bind to the HIP runtime torch has loaded (loaded before torch it pulls in
#                               /opt/rocm's own copy, a second runtime instance that sees no device: hipErrorNoDevice)
from embodied_clip_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "embodied_clip_amd", "lib", "libec_amd_tools.so")
if not os.path.exists(_lib.LIB_PATH):
    raise SystemExit("tools build missing: run `make tools` first")
lib = _lib.load()
lib.ec_bneck_set_debug.restype = None
lib.ec_bneck_set_debug.argtypes = [C.c_void_p]
lib.ec_debug_stamps.restype = C.c_int
lib.ec_debug_stamps.argtypes = [C.c_void_p, C.c_int]
