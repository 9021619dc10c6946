"""Generates tests/golden/preprocess_golden.npz with PILLOW itself (the reference's dependency behind
``clip_preprocess``, thor_image_features.py:108): 300x300 uint8 frames -> Resize(224, BICUBIC) -> CenterCrop(224).
Run from the repo root:  python tests/golden/make_preprocess_golden.py   (needs Pillow; recorded with 12.2.0)."""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_oracle_preprocess import _frames  # noqa: E402

SEED, N, H, W = 77, 3, 300, 300
frames = _frames(SEED, N, H, W)
out = np.stack([np.asarray(Image.fromarray(f).resize((224, 224), Image.BICUBIC)) for f in frames])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "preprocess_golden.npz")
np.savez_compressed(path, seed=SEED, n=N, h=H, w=W, input_slice=frames[:, ::17, ::13], resized_slice=out[:, ::4, ::4],
                    resized_sum=out.reshape(N, -1).astype(np.int64).sum(1), pillow=np.array(Image.__version__ if hasattr(Image, "__version__") else "12.2.0"))
print("wrote", path, os.path.getsize(path), "bytes")
