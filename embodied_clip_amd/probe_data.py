"""Feature cache between the frozen encoder and the probe: writer and reader.

Writer == the CLIP half of ``primitive_probing/generate_data/thor_image_features.py:57-68,105-140`` and
``reachable_image_features.py:60-100``: trunk -> {conv fp32, attnpool, avgpool} per frame, stored as

    thor_{split}.pt                {scene: [ {clip_conv [2048,7,7], clip_attnpool [1024], clip_avgpool [2048],
                                              object_presence int[52], object_localization int[9,52],
                                              free_space int} , ...]}
    reachable_image_features.pt    {image: {clip_avgpool, clip_attnpool}}
    reachable_{split}.pkl          [(image, obj_id, bool), ...]

The encoder runs on the MI355X (``RN50Trunk.forward_u8`` with the CLIP normalisation fused into the stem,
``AttentionPool`` on the native bf16 features, avgpool on the fp32-cast features -- the dtype split of
thor_image_features.py:111-113).  Out of scope here: rendering / semantic-mask labelling (thor_frames.py,
thor_image_features.py:70-127 -- simulator + numpy data prep), the PIL bicubic 300->224 resize of
``clip_preprocess`` (frames are expected at 224x224), and the torchvision-ImageNet ``imagenet_*`` keys (a different
pretrained network; the reader accepts them when present).

Reader == ``primitive_probing/data.py:9-47`` (``THOREmbeddingsDataset``) and the DataLoader collate of
``THOREmbeddingsDataModule`` (data.py:50-88) without pytorch-lightning.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import synthetic as syn
from .probe import MAX_FORWARD_STEPS, NUM_TARGET_OBJECTS

EMBEDDING_TYPES = ("imagenet_avgpool", "clip_avgpool", "clip_attnpool")
PREDICTION_TYPES = ("object_presence", "object_localization", "reachability", "free_space")


# ------------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------------
class ClipFeatureExtractor:
    """frames uint8 [n,H,W,3] -> the three CLIP embeddings the cache stores (all fp32, on the host).  Frames that are
    not 224x224 (the reference renders 300x300: thor_frames.py:33-34) go through CLIP's Resize(224, BICUBIC) +
    CenterCrop(224) on the GPU, bit-exact with the Pillow path of ``clip_preprocess`` (thor_image_features.py:108)."""

    def __init__(self, visual_state_dict, device="cuda:0", batch: int = 64):
        from .encoder import AttentionPool, RN50Trunk
        self.device = torch.device(device)
        self.trunk = RN50Trunk(visual_state_dict, device=self.device)
        self.attnpool = AttentionPool(visual_state_dict, device=self.device)
        self.batch = batch

    @torch.no_grad()
    def __call__(self, frames_u8: torch.Tensor) -> Dict[str, torch.Tensor]:
        assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[3] == 3
        conv, attn, avg = [], [], []
        for i in range(0, frames_u8.shape[0], self.batch):
            x = frames_u8[i:i + self.batch].to(self.device).contiguous()
            feat = self.trunk.forward_u8(x)                       # bf16 NHWC [b,7,7,2048]
            conv.append(self.trunk.to_nchw_f32(feat).cpu())       # clip_features.float()        (:111)
            attn.append(self.attnpool.forward(feat).float().cpu())        # clip_pool(clip_features)     (:112)
            avg.append(self.trunk.spatial_mean(feat).cpu())       # clip_avgpool(features.float()) (:113)
        return {"clip_conv": torch.cat(conv), "clip_attnpool": torch.cat(attn), "clip_avgpool": torch.cat(avg)}


def build_thor_features(extractor, scenes: Dict[str, List[dict]]) -> Dict[str, List[dict]]:
    """``scenes``: {scene_name: [point, ...]}, point = {'frame' uint8 [224,224,3], 'object_presence' int[52],
    'object_localization' int[9,52], 'free_space' int}.  Returns the thor_{split}.pt dictionary."""
    out: Dict[str, List[dict]] = {}
    for scene_name, points in scenes.items():
        if not points:
            out[scene_name] = []
            continue
        frames = torch.stack([torch.as_tensor(p["frame"]) for p in points])
        f = extractor(frames)
        out[scene_name] = [{
            "clip_conv": f["clip_conv"][i].clone(),
            "clip_attnpool": f["clip_attnpool"][i].clone(),
            "clip_avgpool": f["clip_avgpool"][i].clone(),
            "object_presence": torch.as_tensor(p["object_presence"], dtype=torch.int64),
            "object_localization": torch.as_tensor(p["object_localization"], dtype=torch.int64),
            "free_space": int(p["free_space"]),
        } for i, p in enumerate(points)]
    return out


def write_thor_cache(output_dir: str, split: str, features: Dict[str, List[dict]]) -> str:
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, f"thor_{split}.pt")
    torch.save(features, path)
    return path


def build_reachable_features(extractor, images: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """{image_name: uint8 frame} -> {image_name: {clip_avgpool, clip_attnpool}} (reachable_image_features.py:94-98)."""
    names = list(images)
    if not names:
        return {}
    f = extractor(torch.stack([torch.as_tensor(images[n]) for n in names]))
    return {n: {"clip_avgpool": f["clip_avgpool"][i].clone(), "clip_attnpool": f["clip_attnpool"][i].clone()}
            for i, n in enumerate(names)}


def write_reachable_cache(output_dir: str, image_features, split_triples: Dict[str, Sequence[Tuple[str, int, bool]]]):
    os.makedirs(output_dir, exist_ok=True)
    torch.save(image_features, os.path.join(output_dir, "reachable_image_features.pt"))
    for split, triples in split_triples.items():
        with open(os.path.join(output_dir, f"reachable_{split}.pkl"), "wb") as f:
            pickle.dump(list(triples), f)


# ------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------
class THOREmbeddingsDataset:
    """data.py:9-47, same constructor and item format."""

    def __init__(self, data_dir: str, split: str, embedding_type: str, prediction_type: str):
        assert embedding_type in EMBEDDING_TYPES
        assert prediction_type in PREDICTION_TYPES
        self.prediction_type = prediction_type
        self.embeddings: list = []
        self.predictions: list = []
        if prediction_type in ("object_presence", "object_localization", "free_space"):
            if prediction_type == "object_localization":
                assert embedding_type in ("imagenet_avgpool", "clip_avgpool")
                embedding_type = {"imagenet_avgpool": "imagenet_conv", "clip_avgpool": "clip_conv"}[embedding_type]
            data = torch.load(os.path.join(data_dir, f"thor_{split}.pt"))
            for _scene, frames in data.items():
                for frame_features in frames:
                    self.embeddings.append(frame_features[embedding_type])
                    self.predictions.append(frame_features[prediction_type])
        else:
            image_features = torch.load(os.path.join(data_dir, "reachable_image_features.pt"))
            with open(os.path.join(data_dir, f"reachable_{split}.pkl"), "rb") as f:
                data = pickle.load(f)
            for image, obj, reachable in data:
                self.embeddings.append(image_features[image][embedding_type])
                self.predictions.append((obj, torch.tensor(reachable, dtype=torch.int64)))

    def __getitem__(self, index):
        return self.embeddings[index], self.predictions[index]

    def __len__(self):
        return len(self.embeddings)


def collate(items):
    """What torch's default_collate gives the reference for these item types (x stacked; y stacked, a tensor of
    python ints, or the (obj_idx, reachable) pair of tensors)."""
    xs = torch.stack([x for x, _ in items])
    y0 = items[0][1]
    if isinstance(y0, tuple):
        return xs, (torch.tensor([int(y[0]) for _, y in items], dtype=torch.int64),
                    torch.stack([y[1] for _, y in items]))
    if torch.is_tensor(y0):
        return xs, torch.stack([y for _, y in items])
    return xs, torch.tensor([int(y) for _, y in items], dtype=torch.int64)


class _Loader:
    def __init__(self, ds, batch_size: int, shuffle: bool, seed: int):
        self.ds, self.batch_size, self.shuffle, self.seed, self.epoch = ds, batch_size, shuffle, seed, 0

    def __len__(self):
        return (len(self.ds) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator:
        n = len(self.ds)
        order = np.arange(n)
        if self.shuffle:   # portable permutation: argsort of hash keys
            order = np.argsort(syn.hash_u64(self.seed + self.epoch, n, stream=21), kind="stable")
            self.epoch += 1
        for i in range(0, n, self.batch_size):
            yield collate([self.ds[int(j)] for j in order[i:i + self.batch_size]])


class THOREmbeddingsDataModule:
    """data.py:50-88 (batch_size, shuffle on train only); ``num_workers`` accepted and ignored (host loading of a
    few thousand cached vectors is not on the hot path)."""

    def __init__(self, data_dir, embedding_type, prediction_type, batch_size: int = 1, num_workers: int = 0,
                 seed: int = 1):
        self.data_dir, self.embedding_type, self.prediction_type = data_dir, embedding_type, prediction_type
        self.batch_size, self.seed = batch_size, seed

    def setup(self, stage=None):
        mk = lambda s: THOREmbeddingsDataset(self.data_dir, s, self.embedding_type, self.prediction_type)  # noqa: E731
        self.train_dataset, self.val_dataset, self.test_dataset = mk("train"), mk("val"), mk("test")

    def train_dataloader(self):
        return _Loader(self.train_dataset, self.batch_size, True, self.seed)

    def val_dataloader(self):
        return _Loader(self.val_dataset, self.batch_size, False, self.seed)

    def test_dataloader(self):
        return _Loader(self.test_dataset, self.batch_size, False, self.seed)


# ------------------------------------------------------------------------------------------------
# synthetic stand-in for the simulator output (config 1 = 1,000 frames)
# ------------------------------------------------------------------------------------------------
def synthetic_points(seed: int, n: int, res: int = 224) -> List[dict]:
    """n synthetic 'points' with labels that DEPEND on the image (so a probe can learn something):
    object o is 'present' in grid cell g iff the mean of channel (o % 3) over that cell, hashed with o, is high."""
    frames = syn.synthetic_rgb_u8(seed, n, res)
    u = syn.hash_uniform(seed + 17, n * 9 * NUM_TARGET_OBJECTS, stream=3).reshape(n, 9, NUM_TARGET_OBJECTS)
    # brightness of the 3x3 cells per channel -> [n, 9, 3]
    H = res
    bins = [(0, int(H / 3)), (int(H / 3), int(2 * H / 3)), (int(2 * H / 3), H)]
    f = frames.float() / 255.0
    cells = torch.stack([f[:, y0:y1, x0:x1, :].mean(dim=(1, 2)) for (y0, y1) in bins for (x0, x1) in bins], 1)
    bright = cells[:, :, [o % 3 for o in range(NUM_TARGET_OBJECTS)]].numpy()          # [n, 9, 52]
    loc = ((bright - 0.5) * 40.0 + (u - 0.5) > 0.35).astype(np.int64)                  # sparse positives
    pres = (loc.sum(axis=1) > 0).astype(np.int64)
    free = (syn.hash_u64(seed + 23, n, stream=4) % np.uint64(15)).astype(np.int64)     # 0..14 -> clamp exercised
    return [{"frame": frames[i], "object_presence": pres[i], "object_localization": loc[i],
             "free_space": int(free[i])} for i in range(n)]


def synthetic_reachability(seed: int, image_names: Sequence[str], n: int):
    k = syn.hash_u64(seed, 3 * n, stream=5)
    return [(image_names[int(k[3 * i] % np.uint64(len(image_names)))], int(k[3 * i + 1] % np.uint64(110)),
             bool(k[3 * i + 2] & np.uint64(1))) for i in range(n)]
