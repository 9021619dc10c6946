"""Feature cache between the frozen encoders and the probe: writer and reader.

Writer == ``primitive_probing/generate_data/thor_image_features.py:36-68,102-140`` and
``reachable_image_features.py:39-100``: both frozen trunks -> their embeddings per frame, stored as

    thor_{split}.pt                {scene: [ {imagenet_conv [2048,7,7], imagenet_avgpool [2048],
                                              clip_conv [2048,7,7], clip_attnpool [1024], clip_avgpool [2048],
                                              object_presence int[52], object_localization int[9,52],
                                              free_space int} , ...]}
    reachable_image_features.pt    {image: {imagenet_avgpool, clip_avgpool, clip_attnpool}}
    reachable_{split}.pkl          [(image, obj_id, bool), ...]

The encoders run on the MI355X: CLIP -- ``RN50Trunk.forward_u8`` with the CLIP normalisation fused into the stem,
``AttentionPool`` on the native bf16 features, avgpool on the fp32-cast features (the dtype split of
thor_image_features.py:111-113); ImageNet -- ``ImageNetRN50Trunk.forward_u8`` (torchvision ResNet-50 minus avgpool / fc,
ImageNet mean / std fused into its 7x7 stem; the reference runs this tower in fp32, here bf16 storage with fp32
accumulation like every other conv of the path).  Both share ONE Pillow-exact Resize(224, BICUBIC) + CenterCrop(224) pass
over the raw frame (``resnet_preprocess`` and ``clip_preprocess`` apply the same geometry, :36-39,108).  Out of scope
here: rendering / semantic-mask labelling (thor_frames.py, thor_image_features.py:70-127 -- simulator + numpy data prep).

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
CLIP_KEYS = ("clip_conv", "clip_attnpool", "clip_avgpool")
IMAGENET_KEYS = ("imagenet_conv", "imagenet_avgpool")


class ClipFeatureExtractor:
    """frames uint8 [n,H,W,3] -> the embeddings the cache stores (all fp32, on the host): the three ``clip_*`` keys and,
    when a torchvision ResNet-50 state dict is given (``imagenet_state_dict``), the two ``imagenet_*`` keys.  Frames that
    are not 224x224 (the reference renders 300x300: thor_frames.py:33-34) go through Resize(224, BICUBIC) +
    CenterCrop(224) on the GPU ONCE for both towers, bit-exact with the Pillow path of ``clip_preprocess`` /
    ``resnet_preprocess`` (thor_image_features.py:36-39,108)."""

    def __init__(self, visual_state_dict, device="cuda:0", batch: int = 64, imagenet_state_dict=None):
        from .encoder import AttentionPool, ClipResizeCrop, ImageNetRN50Trunk, RN50Trunk
        self.device = torch.device(device)
        self.trunk = RN50Trunk(visual_state_dict, device=self.device)
        self.attnpool = AttentionPool(visual_state_dict, device=self.device)
        self.imagenet = ImageNetRN50Trunk(imagenet_state_dict, device=self.device) if imagenet_state_dict is not None else None
        self.resize = ClipResizeCrop(self.device, self.trunk.input_resolution)
        self.batch = batch

    @property
    def keys(self):
        return (IMAGENET_KEYS if self.imagenet is not None else ()) + CLIP_KEYS

    @torch.no_grad()
    def __call__(self, frames_u8: torch.Tensor) -> Dict[str, torch.Tensor]:
        assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[3] == 3
        cols: Dict[str, list] = {k: [] for k in self.keys}
        R = self.trunk.input_resolution
        for i in range(0, frames_u8.shape[0], self.batch):
            x = frames_u8[i:i + self.batch].to(self.device).contiguous()
            if x.shape[1] != R or x.shape[2] != R:
                x = self.resize(x)
            if self.imagenet is not None:
                f = self.imagenet.forward_u8(x)                                   # resnet_model(resnet_input)      (:103)
                cols["imagenet_conv"].append(self.imagenet.to_nchw_f32(f).cpu())  # resnet_features[0]              (:105)
                cols["imagenet_avgpool"].append(self.imagenet.spatial_mean(f).cpu())   # resnet_pool(...)           (:106)
            feat = self.trunk.forward_u8(x)                                       # bf16 NHWC [b,7,7,2048]
            cols["clip_conv"].append(self.trunk.to_nchw_f32(feat).cpu())          # clip_features.float()           (:111)
            cols["clip_attnpool"].append(self.attnpool.forward(feat).float().cpu())    # clip_pool(clip_features)   (:112)
            cols["clip_avgpool"].append(self.trunk.spatial_mean(feat).cpu())      # clip_avgpool(features.float())  (:113)
        return {k: torch.cat(v) for k, v in cols.items()}


def build_thor_features(extractor, scenes: Dict[str, List[dict]]) -> Dict[str, List[dict]]:
    """``scenes``: {scene_name: [point, ...]}, point = {'frame' uint8 [H,W,3], 'object_presence' int[52],
    'object_localization' int[9,52], 'free_space' int}.  Returns the thor_{split}.pt dictionary
    (thor_image_features.py:129-138; the ``imagenet_*`` entries when the extractor holds the ImageNet tower)."""
    out: Dict[str, List[dict]] = {}
    for scene_name, points in scenes.items():
        if not points:
            out[scene_name] = []
            continue
        f = extractor(torch.stack([torch.as_tensor(p["frame"]) for p in points]))
        rows = []
        for i, p in enumerate(points):
            row = {k: v[i].clone() for k, v in f.items()}
            row["object_presence"] = torch.as_tensor(p["object_presence"], dtype=torch.int64)
            row["object_localization"] = torch.as_tensor(p["object_localization"], dtype=torch.int64)
            row["free_space"] = int(p["free_space"])
            rows.append(row)
        out[scene_name] = rows
    return out


def write_thor_cache(output_dir: str, split: str, features: Dict[str, List[dict]]) -> str:
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, f"thor_{split}.pt")
    torch.save(features, path)
    return path


def build_reachable_features(extractor, images: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """{image_name: uint8 frame} -> {image_name: {imagenet_avgpool, clip_avgpool, clip_attnpool}}
    (reachable_image_features.py:94-98; the pooled embeddings only)."""
    names = list(images)
    if not names:
        return {}
    f = extractor(torch.stack([torch.as_tensor(images[n]) for n in names]))
    keep = [k for k in ("imagenet_avgpool", "clip_avgpool", "clip_attnpool") if k in f]
    return {n: {k: f[k][i].clone() for k in keep} for i, n in enumerate(names)}


def write_reachable_cache(output_dir: str, image_features, split_triples: Dict[str, Sequence[Tuple[str, int, bool]]]):
    os.makedirs(output_dir, exist_ok=True)
    torch.save(image_features, os.path.join(output_dir, "reachable_image_features.pt"))
    for split, triples in split_triples.items():
        with open(os.path.join(output_dir, f"reachable_{split}.pkl"), "wb") as f:
            pickle.dump(list(triples), f)


# ------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------
# which file holds a prediction type's rows, and which stored key feeds the probe (data.py:15-19: the localisation probe
# reads the conv map of the tower whose pooled embedding was asked for)
_FRAME_TASKS = ("object_presence", "object_localization", "free_space")
_CONV_KEY = {"imagenet_avgpool": "imagenet_conv", "clip_avgpool": "clip_conv"}


class THOREmbeddingsDataset:
    """Reader of the cache schema with the constructor and item format of data.py:9-47: item = (embedding, prediction);
    prediction = the stored label tensor / int, or (obj_idx, reachable) for the reachability triples."""

    def __init__(self, data_dir: str, split: str, embedding_type: str, prediction_type: str):
        if embedding_type not in EMBEDDING_TYPES or prediction_type not in PREDICTION_TYPES:
            raise AssertionError((embedding_type, prediction_type))
        self.prediction_type = prediction_type
        if prediction_type in _FRAME_TASKS:
            key = embedding_type
            if prediction_type == "object_localization":
                if embedding_type not in _CONV_KEY:
                    raise AssertionError("object_localization probes the conv map: imagenet_avgpool / clip_avgpool only")
                key = _CONV_KEY[embedding_type]
            scenes = torch.load(os.path.join(data_dir, f"thor_{split}.pt"))
            rows = [row for frames in scenes.values() for row in frames]
            self.embeddings = [row[key] for row in rows]
            self.predictions = [row[prediction_type] for row in rows]
        else:
            table = torch.load(os.path.join(data_dir, "reachable_image_features.pt"))
            with open(os.path.join(data_dir, f"reachable_{split}.pkl"), "rb") as fh:
                triples = pickle.load(fh)
            self.embeddings = [table[image][embedding_type] for image, _obj, _r in triples]
            self.predictions = [(obj, torch.tensor(r, dtype=torch.int64)) for _image, obj, r in triples]

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
