"""Static check for the hand-pipelined LDS streams (conv_pair.hip lds_stream_mfma, conv_igemm.hip fr_step, dw_tn.hip, conv_bneck.hip's K loops
and epilogues, policy.hip gru_lread).

Those use PARTIAL waits (`s_waitcnt lgkmcnt(n)`, n > 0) on inline-asm LDS reads.  lgkmcnt also counts scalar memory
loads, which return OUT OF ORDER: a partial wait is only meaningful while no `s_load` / `s_buffer_load` is in flight.  The
compiler places scalar loads itself (kernel arguments, possibly re-loaded under SGPR pressure), so this script
disassembles the device code and fails if any scalar load sits between the last full drain (`s_waitcnt lgkmcnt(0)`, a
label or the function entry) and a partial wait.
usage: python tools/check_lgkmcnt.py [file.hip ...]   (default: the five files above); exit code 1 on a violation."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["conv_pair.hip", "conv_igemm.hip", "dw_tn.hip", "conv_bneck.hip", "policy.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def check_asm(text):
    bad = []
    fn = "?"
    window = []          # instructions since the last full drain
    for line in text.split("\n"):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn, window = m.group(1), []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB", s):       # a branch target: control may arrive with a different history -> be strict:
                pass                          # keep the window (a scalar load before the label still counts)
            continue
        op = s.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                if n == 0:
                    window = []
                elif any(w.startswith(("s_load", "s_buffer_load")) for w in window):
                    bad.append((fn, s, [w for w in window if w.startswith(("s_load", "s_buffer_load"))][:2]))
            continue
        if op.startswith(("s_load", "s_buffer_load")):
            window.append(s)
    return bad


def main(files):
    rc = 0
    for f in files:
        src = f if os.path.isabs(f) else os.path.join(ROOT, "embodied_clip_amd", "csrc", f)
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=fast",
                                   "--cuda-device-only", "-S", "-w", "-o", out, src], cwd=d)
            bad = check_asm(open(out).read())
        for fn, wait, loads in bad:
            print(f"{os.path.basename(src)}: {fn[:80]}: `{wait}` with scalar loads possibly in flight: {loads}")
            rc = 1
        print(f"{os.path.basename(src)}: {'VIOLATIONS' if bad else 'ok'}")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or FILES))
