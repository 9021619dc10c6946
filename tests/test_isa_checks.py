"""Static ISA checks on the device code of the hand-scheduled kernels (CPU: hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="hipcc not found")
def test_partial_lgkmcnt_waits_never_overlap_scalar_loads():
    """The software-pipelined LDS streams wait with `s_waitcnt lgkmcnt(n)`, n > 0; scalar loads share that counter and
    return out of order, so none may be in flight at such a wait (tools/check_lgkmcnt.py scans the disassembly)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lgkmcnt.py")], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(": ok") == 5, r.stdout      # conv_pair, conv_igemm, dw_tn, conv_bneck, policy


def test_checker_flags_a_scalar_load_before_a_partial_wait():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_lgkmcnt as c
    good = "_Zk:\n s_load_dword s0, s[0:1], 0x0\n s_waitcnt lgkmcnt(0)\n ds_read_b128 v[0:3], v4\n s_waitcnt lgkmcnt(1)\n"
    bad = "_Zk:\n s_waitcnt lgkmcnt(0)\n ds_read_b128 v[0:3], v4\n s_load_dword s0, s[0:1], 0x0\n s_waitcnt lgkmcnt(1)\n"
    assert c.check_asm(good) == [] and len(c.check_asm(bad)) == 1
