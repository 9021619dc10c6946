"""``python -m embodied_clip_amd.allenact_main <allenact main.py arguments>``

Runs AllenAct's own entry point (the reference launches ``python allenact/main.py -o ... -b ... <config>``:
readme_files/baselines_robothor_objectnav.md:48,51) after rebinding ``ClipResNetPreprocessor`` /
``ClipViTPreprocessor`` / ``ResnetTensorObjectNavActorCritic`` / ``PPO`` inside the installed AllenAct modules to the
HIP-backed classes, so the experiment configs are used UNCHANGED.  Needs ``allenact`` to be installed (it is not in
the build image; the patching itself is covered by tests/test_allenact_surface.py with a stand-in package tree).
"""
from __future__ import annotations

import sys


def main(argv=None) -> int:
    from .allenact_compat import install_into_allenact
    patched = install_into_allenact(verbose=True)
    if not patched:
        raise SystemExit("embodied_clip_amd.allenact_main: allenact / allenact_plugins are not importable")
    import allenact.main as am  # type: ignore
    if argv is not None:
        sys.argv = [sys.argv[0]] + list(argv)
    return am.main()


if __name__ == "__main__":
    sys.exit(main() or 0)
