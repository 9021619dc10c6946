"""Probe CLI (SURVEY.md §8b "Probe CLI"; BASELINE config 1) -- same flags as ``primitive_probing/train.py:119-134``:

    python -m embodied_clip_amd.probe_train --data-dir data --log-dir logs/ \
        --embedding-type clip_avgpool --prediction-type object_presence --gpus 1

Extra flags (not in the reference): ``--synthetic-frames N`` first writes a synthetic feature cache into
``--data-dir`` by running N seeded frames through the HIP CLIP-RN50 and ImageNet-ResNet-50 encoders (there are no simulator frames or
pretrained weights in this environment); ``--epochs`` (default 250 == ``max_epochs=250``, train.py:158);
``--batch-size`` (reference hard-codes 128, train.py:136; BASELINE config 1 quotes 32).

Trainer semantics restated from train.py:153-174 (the Lightning machinery itself is control plane and not rebuilt):
validation runs twice per epoch (``val_check_interval=0.5``, train.py:157), the head with the lowest ``val_loss`` so
far is kept (``ModelCheckpoint(monitor="val_loss", mode="min")``, train.py:160-164) and the test metrics are computed
on THAT head (``trainer.test(ckpt_path='best')``, train.py:170-174), not on the final weights.
Prints one JSON line with the last train loss, the best val loss/acc and the test loss/acc of the best head, and
writes the best head's state_dict to ``<log-dir>/<embedding>_<prediction>.pt``.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch

from . import synthetic as syn
from .probe import LinearEncoder
from .probe_data import (EMBEDDING_TYPES, PREDICTION_TYPES, ClipFeatureExtractor, THOREmbeddingsDataModule,
                         build_reachable_features, build_thor_features, synthetic_points, synthetic_reachability,
                         write_reachable_cache, write_thor_cache)


def write_synthetic_cache(data_dir: str, n_frames: int, device="cuda:0", seed: int = 1) -> float:
    """80/10/10 split of n_frames synthetic points; returns encoder frames/s."""
    # both towers of the feature scripts (thor_image_features.py:46-67): CLIP-RN50 and torchvision ResNet-50
    ex = ClipFeatureExtractor(syn.rn50_visual_state_dict(0), device=device, imagenet_state_dict=syn.tv_resnet_state_dict(0))
    n_val = max(1, n_frames // 10)
    sizes = {"train": n_frames - 2 * n_val, "val": n_val, "test": n_val}
    t0 = time.time()
    for k, (split, n) in enumerate(sizes.items()):
        pts = synthetic_points(seed + 100 * k, n)
        scenes = {f"FloorPlan{1 + s}": pts[s::4] for s in range(4)}
        write_thor_cache(data_dir, split, build_thor_features(ex, scenes))
    images = {f"img{i:05d}": f for i, f in enumerate(syn.synthetic_rgb_u8(seed + 999, min(n_frames, 256)))}
    feats = build_reachable_features(ex, images)
    names = list(images)
    write_reachable_cache(data_dir, feats, {s: synthetic_reachability(seed + 7 * k, names, max(8, sizes[s]))
                                            for k, s in enumerate(sizes)})
    torch.cuda.synchronize()
    return (n_frames + len(images)) / (time.time() - t0)


def evaluate(model: LinearEncoder, loader, which: str):
    tot_l, tot_a, n = 0.0, 0.0, 0
    for i, batch in enumerate(loader):
        (model.validation_step if which == "val" else model.test_step)(batch, i)
        b = batch[0].shape[0]
        tot_l += float(model.logged[f"{which}_loss"]) * b
        tot_a += float(model.logged[f"{which}_acc"]) * b
        n += b
    return tot_l / max(n, 1), tot_a / max(n, 1)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-dir", type=str, dest="data_dir", default="data", help="Path to data directory")
    ap.add_argument("--log-dir", type=str, dest="log_dir", default="logs/", help="Path to log directory")
    ap.add_argument("--embedding-type", dest="embedding_type", type=str, choices=list(EMBEDDING_TYPES),
                    help="Which encoder features to evaluate", default="clip_avgpool")
    ap.add_argument("--prediction-type", dest="prediction_type", type=str, choices=list(PREDICTION_TYPES),
                    help="Which task to evaluate", default="object_presence")
    ap.add_argument("--gpus", type=int, default=1, help="Number of GPUs to use (the probe is a single-GPU job)")
    ap.add_argument("--synthetic-frames", type=int, default=0)
    ap.add_argument("--epochs", type=int, default=250, help="max_epochs (train.py:158)")
    ap.add_argument("--batch-size", type=int, default=128)
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("embodied_clip_amd.probe_train needs an MI355X (no CPU fallback)")
    dev = "cuda:0"
    enc_fps = None
    if a.synthetic_frames > 0:
        enc_fps = write_synthetic_cache(a.data_dir, a.synthetic_frames, dev)
    lr = 0.001                                                        # train.py:137
    dm = THOREmbeddingsDataModule(a.data_dir, a.embedding_type, a.prediction_type, batch_size=a.batch_size)
    dm.setup()
    model = LinearEncoder(a.embedding_type, a.prediction_type, a.batch_size, lr, device=dev)
    train = dm.train_dataloader()
    n_batches = len(train)
    # val_check_interval=0.5 (train.py:157): Lightning validates every int(n_batches * 0.5) training batches, i.e. after
    # batches k, 2k, ... <= n_batches (for an odd n_batches the second check falls one batch BEFORE the epoch's end)
    val_every = max(1, int(n_batches * 0.5))
    best = {"val_loss": float("inf"), "val_acc": 0.0, "epoch": -1, "sd": None}

    def validate(epoch):
        vl, va = evaluate(model, dm.val_dataloader(), "val")
        if vl < best["val_loss"]:                                      # ModelCheckpoint(monitor="val_loss", mode="min")
            best.update(val_loss=vl, val_acc=va, epoch=epoch,
                        sd={k: v.detach().clone() for k, v in model.state_dict().items()})

    t0 = time.time()
    steps = 0
    for ep in range(a.epochs):
        for i, batch in enumerate(train):
            model.training_step(batch, i)
            steps += 1
            if (i + 1) % val_every == 0:
                validate(ep)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if best["sd"] is None:
        validate(-1)
    model.load_state_dict(best["sd"])                                  # trainer.test(ckpt_path='best')
    val_loss, val_acc = best["val_loss"], best["val_acc"]
    test_loss, test_acc = evaluate(model, dm.test_dataloader(), "test")
    os.makedirs(a.log_dir, exist_ok=True)
    torch.save({k: v.cpu() for k, v in model.state_dict().items()},
               os.path.join(a.log_dir, f"{a.embedding_type}_{a.prediction_type}.pt"))
    print(json.dumps({"embedding_type": a.embedding_type, "prediction_type": a.prediction_type,
                      "train_frames": len(dm.train_dataset), "epochs": a.epochs, "best_epoch": best["epoch"], "batch_size": a.batch_size,
                      "train_steps": steps, "train_steps_per_s": round(steps / max(dt, 1e-9), 1),
                      "train_loss": round(float(model.logged["train_loss"]), 6),
                      "val_loss": round(val_loss, 6), "val_acc": round(val_acc, 4),
                      "test_loss": round(test_loss, 6), "test_acc": round(test_acc, 4),
                      "encoder_frames_per_s": None if enc_fps is None else round(enc_fps, 1)}))


if __name__ == "__main__":
    main()
