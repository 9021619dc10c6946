"""The C-ABI library loads on a box without a GPU and exports exactly the
symbols include/ec_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest


def _declared(repo_root):
    txt = open(os.path.join(repo_root, "include", "ec_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ec_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(repo_root):
    from embodied_clip_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared(repo_root)
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ec_amd.h but not exported"
    # and the Python binding table covers the header, both directions
    assert sorted(_lib.SIGNATURES) == names


def test_error_strings_and_version(repo_root):
    from embodied_clip_amd import _lib
    lib = _lib.load()
    assert lib.ec_version() >= 100
    assert lib.ec_strerror(0) == b"ok"
    assert b"shape" in lib.ec_strerror(-2)
    # argument validation happens before any HIP call, so it is safe without a GPU
    assert lib.ec_conv_bf16(None, None, None, None, None, 1, 1, 1, 8, 32, 1, 0, 0, None) == -1
    assert lib.ec_gemm_bf16(None, None, None, None, None, 1, 32, 8, 0, None) == -1
    # round-5 entry points (the ImageNet tower): null pointers -> EC_ERR_ARG, bad geometry -> EC_ERR_SHAPE, before any HIP call
    assert lib.ec_conv_bf16_s2(None, None, None, None, None, 1, 8, 8, 64, 64, 3, 1, None) == -1
    assert lib.ec_conv_bf16_s2(1, 1, None, None, 1, 1, 7, 8, 64, 64, 3, 1, None) == -2        # odd height
    assert lib.ec_conv_bf16_s2(1, 1, None, None, 1, 1, 8, 8, 64, 48, 3, 1, None) == -2        # Cout % 64
    assert lib.ec_conv_bf16_s2(1, 1, None, None, 1, 1, 8, 8, 64, 64, 5, 1, None) == -2        # 5x5
    assert lib.ec_conv_bf16_s2(1, 1, None, None, 1, 1, 8, 8, 64, 64, 3, 2, None) == -6        # QuickGELU: unsupported
    assert lib.ec_stem7_pool(None, 0, None, None, None, None, None, 1, 224, 224, None) == -1
    assert lib.ec_stem7_pool(1, 0, None, None, 1, 1, 1, 1, 222, 224, None) == -2             # H % 4
    assert lib.ec_stem7_pool(1, 1, None, None, 1, 1, 1, 1, 224, 224, None) == -1             # uint8 frames need mean / std
    # output-row-stride variants and the stream plumbing: argument checks come first as well
    assert lib.ec_conv_bf16_ld(None, None, None, None, None, 1, 1, 1, 8, 32, 1, 0, 0, 32, None) == -1
    assert lib.ec_conv_bf16_ld(1, 1, None, None, 1, 1, 4, 4, 64, 64, 1, 0, 1, 60, None) == -2   # row stride below Cout
    assert lib.ec_avgpool2_bf16_ld(None, None, 1, 2, 2, 8, 8, None) == -1
    assert lib.ec_conv3x3_img_bf16_ld(None, None, None, None, 1, 14, 14, 256, 0, 256, None) == -1
    assert lib.ec_bind_streams(None, 2, 100) == -1
    assert lib.ec_stream_pair_overlap(None, None, 100, None) == -1
    import ctypes as C
    r = C.c_float()
    assert lib.ec_stream_pair_overlap(1, 1, 100, C.byref(r)) == -1                          # one stream is not a pair
    h = C.c_void_p()
    assert lib.ec_rn50tv_create(C.byref(h), None, 224, None, None, 0, None, 0) == -1


def test_missing_library_fails_loudly(monkeypatch, repo_root):
    from embodied_clip_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libec_amd.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
