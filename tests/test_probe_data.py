"""Feature-cache schema reader / collate on the host -- no GPU needed: the cache is written here
with stand-in embeddings in the reference's layout (thor_image_features.py:129-140, reachable_image_features.py:94-100,
reachable_metadata.py:54-71) and read back through the data.py:9-47 mirror."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_clip_amd import probe_data as pd  # noqa: E402
from embodied_clip_amd.probe import head_dims  # noqa: E402


def _fake_frame(i):
    g = torch.Generator().manual_seed(i)
    return {"imagenet_conv": torch.randn(2048, 7, 7, generator=g), "imagenet_avgpool": torch.randn(2048, generator=g),
            "clip_conv": torch.randn(2048, 7, 7, generator=g), "clip_attnpool": torch.randn(1024, generator=g),
            "clip_avgpool": torch.randn(2048, generator=g),
            "object_presence": torch.randint(0, 2, (52,), generator=g),
            "object_localization": torch.randint(0, 2, (9, 52), generator=g), "free_space": int(i % 15)}


@pytest.fixture()
def cache(tmp_path):
    d = str(tmp_path)
    for k, split in enumerate(("train", "val", "test")):
        pd.write_thor_cache(d, split, {"FloorPlan1": [_fake_frame(10 * k + i) for i in range(5)],
                                       "FloorPlan2": [_fake_frame(10 * k + 5 + i) for i in range(2)], "Empty": []})
    imgs = {f"img{i}": {"imagenet_avgpool": torch.randn(2048), "clip_avgpool": torch.randn(2048),
                        "clip_attnpool": torch.randn(1024)} for i in range(4)}
    pd.write_reachable_cache(d, imgs, {s: pd.synthetic_reachability(k, list(imgs), 9)
                                       for k, s in enumerate(("train", "val", "test"))})
    return d


@pytest.mark.parametrize("emb,task,xshape", [
    ("clip_avgpool", "object_presence", (2048,)), ("clip_attnpool", "object_presence", (1024,)),
    ("imagenet_avgpool", "free_space", (2048,)), ("clip_avgpool", "object_localization", (2048, 7, 7)),
    ("imagenet_avgpool", "object_localization", (2048, 7, 7)), ("clip_attnpool", "reachability", (1024,))])
def test_dataset_items_and_collate(cache, emb, task, xshape):
    ds = pd.THOREmbeddingsDataset(cache, "train", emb, task)
    assert len(ds) == (9 if task == "reachability" else 7)
    x, y = ds[0]
    assert tuple(x.shape) == xshape
    dm = pd.THOREmbeddingsDataModule(cache, emb, task, batch_size=4)
    dm.setup()
    batches = list(dm.val_dataloader())
    assert sum(b[0].shape[0] for b in batches) == len(dm.val_dataset)
    xb, yb = batches[0]
    assert tuple(xb.shape) == (4,) + xshape
    if task == "reachability":
        assert yb[0].shape == (4,) and yb[1].shape == (4,) and yb[0].dtype == torch.int64
        assert int(yb[0].max()) < 110
    elif task == "free_space":
        assert yb.shape == (4,) and yb.dtype == torch.int64
    elif task == "object_presence":
        assert yb.shape == (4, 52)
    else:
        assert yb.shape == (4, 9, 52)
    assert head_dims(emb, task)[0] == xshape[0]


def test_train_loader_shuffles_deterministically(cache):
    dm = pd.THOREmbeddingsDataModule(cache, "clip_avgpool", "free_space", batch_size=3, seed=4)
    dm.setup()
    a = torch.cat([y for _, y in dm.train_dataloader()])
    dm2 = pd.THOREmbeddingsDataModule(cache, "clip_avgpool", "free_space", batch_size=3, seed=4)
    dm2.setup()
    b = torch.cat([y for _, y in dm2.train_dataloader()])
    assert torch.equal(a, b) and sorted(a.tolist()) == sorted(y for y in dm.train_dataset.predictions)
    # val order is the storage order (shuffle=False, data.py:78)
    assert [int(v) for v in torch.cat([y for _, y in dm.val_dataloader()])] == dm.val_dataset.predictions


def test_localization_requires_conv_embedding(cache):
    with pytest.raises(AssertionError):
        pd.THOREmbeddingsDataset(cache, "train", "clip_attnpool", "object_localization")   # data.py:19
    with pytest.raises(AssertionError):
        head_dims("clip_attnpool", "object_localization")                                    # train.py:43


def test_synthetic_points_schema():
    pts = pd.synthetic_points(1, 5, res=224)
    p = pts[0]
    assert p["frame"].shape == (224, 224, 3) and p["frame"].dtype == torch.uint8
    assert p["object_presence"].shape == (52,) and p["object_localization"].shape == (9, 52)
    assert ((p["object_localization"].sum(0) > 0) == (p["object_presence"] > 0)).all()
    assert 0 <= p["free_space"] < 15
