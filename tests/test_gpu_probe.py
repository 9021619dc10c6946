"""GPU parity of the linear probe (primitive_probing/train.py) against oracle/probe.py."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_clip_amd import synthetic as syn  # noqa: E402
from oracle import probe as oprobe  # noqa: E402

pytestmark = pytest.mark.gpu

CASES = [("clip_avgpool", "object_presence"), ("clip_attnpool", "object_presence"),
         ("clip_avgpool", "object_localization"), ("clip_attnpool", "reachability"), ("clip_avgpool", "free_space")]


def _batch(task, in_dim, B, seed):
    g = torch.Generator().manual_seed(seed)
    if task == "object_localization":
        x = torch.randn(B, in_dim, 7, 7, generator=g)
        y = (torch.rand(B, 9, 52, generator=g) < 0.15).long()
    else:
        x = torch.randn(B, in_dim, generator=g)
        if task == "object_presence":
            y = (torch.rand(B, 52, generator=g) < 0.2).long()
        elif task == "reachability":
            y = (torch.randint(0, 110, (B,), generator=g), torch.randint(0, 2, (B,), generator=g))
        else:
            y = torch.randint(0, 15, (B,), generator=g)     # > 10 exercises the clamp (train.py:65)
    return x, y


@pytest.mark.parametrize("emb,task", CASES)
@pytest.mark.parametrize("B", [1, 32, 128])
def test_probe_forward_loss_metrics_and_training_step(emb, task, B):
    from embodied_clip_amd.probe import LinearEncoder
    m = LinearEncoder(emb, task, batch_size=B, lr=1e-3, device="cuda:0", seed=5)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    w = sd[m._prefix + ".weight"].reshape(m.C, m.K).clone().requires_grad_(True)
    b = sd[m._prefix + ".bias"].clone().requires_grad_(True)
    x, y = _batch(task, m.K, B, seed=B + len(task))
    x = x * 3.0                                            # some saturated sigmoids / peaked softmaxes

    # forward (reference layout) -----------------------------------------------------------------
    ref_pred = oprobe.forward(x, w.detach(), b.detach(), task)
    got_pred = m.forward(x).cpu()
    assert got_pred.shape == ref_pred.shape
    assert torch.allclose(got_pred, ref_pred, atol=2e-6, rtol=1e-5), (got_pred - ref_pred).abs().max()

    # loss + metrics (eval) ------------------------------------------------------------------------
    ycl = (y[0].clone(), y[1].clone()) if isinstance(y, tuple) else y.clone()
    ref_loss, ref_metrics = oprobe.compute_loss(x, ycl, w, b, task, eval=True)
    loss, metrics = m.compute_loss((x, y), eval=True)
    assert abs(float(loss) - float(ref_loss.detach())) <= 2e-6 + 1e-5 * abs(float(ref_loss.detach()))
    assert abs(float(metrics["accuracy"]) - float(ref_metrics["accuracy"])) < 1e-6

    # one training step == autograd + torch.optim.Adam(lr) (train.py:111-113) ----------------------------
    opt = torch.optim.Adam([w, b], lr=1e-3)
    for _ in range(3):
        opt.zero_grad()
        l_ref = oprobe.compute_loss(x, ycl, w, b, task)
        l_ref.backward()
        l_got = m.training_step((x, y))
        if _ == 0:
            gw = m._dW.cpu()
            assert torch.allclose(gw, w.grad, atol=1e-7, rtol=2e-4), (gw - w.grad).abs().max()
            assert torch.allclose(m._db.cpu(), b.grad, atol=1e-7, rtol=2e-4)
        opt.step()
        assert abs(float(l_got) - float(l_ref)) <= 5e-6 + 1e-4 * abs(float(l_ref))
    # After 3 Adam steps every parameter moved by <= 3 lr.  Adam's first steps are g/(|g|+eps): an element whose
    # gradient is ~eps (1e-8) in magnitude amplifies summation-order noise to a fraction of lr, so the max is only
    # bounded by the step size; the bulk must agree far inside one step.
    d = torch.cat([(m.weight.cpu() - w.detach()).abs().flatten(), (m.bias.cpu() - b.detach()).abs().flatten()])
    assert d.max() <= 3.1e-3
    assert d.kthvalue(max(1, int(0.999 * d.numel()))).values < 2e-5, d.kthvalue(int(0.999 * d.numel())).values


def test_probe_golden_losses():
    """The committed oracle losses (tests/golden/oracle_golden.pt, section 'probe') reproduced by the HIP path."""
    from embodied_clip_amd.probe import LinearEncoder
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.pt"))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden
    for task, (x, y, w, bb) in make_golden.probe_cases().items():
        emb = "clip_avgpool"
        m = LinearEncoder(emb, task, batch_size=x.shape[0], lr=1e-3, device="cuda:0")
        m.load_state_dict({m._prefix + ".weight": w, m._prefix + ".bias": bb})
        got = float(m.compute_loss((x, y)))
        assert abs(got - gold["probe"][task]) <= 2e-6 + 1e-5 * abs(gold["probe"][task]), (task, got)


def test_feature_cache_writer_roundtrip_and_cli(tmp_path):
    """encoder -> cache (schema a20) -> reader -> probe; cached embeddings equal the oracle's for the same frames."""
    from embodied_clip_amd import probe_data as pd
    from embodied_clip_amd import probe_train
    from oracle import clip_resnet as ocr
    sd = syn.rn50_visual_state_dict(0)
    ex = pd.ClipFeatureExtractor(sd, device="cuda:0", batch=4)
    pts = pd.synthetic_points(3, 6)
    feats = pd.build_thor_features(ex, {"FloorPlan1": pts[:4], "FloorPlan2": pts[4:], "Empty": []})
    assert set(feats["FloorPlan1"][0]) == {"clip_conv", "clip_attnpool", "clip_avgpool", "object_presence",
                                           "object_localization", "free_space"}
    f0 = feats["FloorPlan1"][0]
    assert f0["clip_conv"].shape == (2048, 7, 7) and f0["clip_conv"].dtype == torch.float32
    assert f0["clip_attnpool"].shape == (1024,) and f0["clip_avgpool"].shape == (2048,)
    assert f0["object_presence"].shape == (52,) and f0["object_localization"].shape == (9, 52)
    assert isinstance(f0["free_space"], int) and feats["Empty"] == []
    # against the oracle on the same two frames (bf16 trunk tolerance; attnpool on bf16 features)
    x = syn.normalize_rgb(torch.stack([p["frame"] for p in pts[:2]]))
    ref_conv = ocr.clip_resnet_preprocessor(x, sd)
    got_conv = torch.stack([feats["FloorPlan1"][i]["clip_conv"] for i in range(2)])
    rel = lambda a, b: float((a - b).norm() / b.norm())  # noqa: E731
    assert rel(got_conv, ref_conv) < 2e-2
    assert rel(torch.stack([feats["FloorPlan1"][i]["clip_avgpool"] for i in range(2)]), ref_conv.mean(dim=(2, 3))) < 2e-2
    ref_attn = ocr.attnpool(ref_conv, sd)
    assert rel(torch.stack([feats["FloorPlan1"][i]["clip_attnpool"] for i in range(2)]), ref_attn) < 3e-2
    # the CLI end to end on a synthetic cache (tiny)
    d = str(tmp_path / "data")
    probe_train.main(["--data-dir", d, "--log-dir", str(tmp_path / "logs"), "--synthetic-frames", "40", "--epochs", "2",
                      "--batch-size", "16", "--embedding-type", "clip_avgpool", "--prediction-type", "object_presence"])
    for task, emb in (("object_localization", "clip_avgpool"), ("reachability", "clip_attnpool"),
                      ("free_space", "clip_avgpool")):
        probe_train.main(["--data-dir", d, "--log-dir", str(tmp_path / "logs"), "--epochs", "1", "--batch-size", "16",
                          "--embedding-type", emb, "--prediction-type", task])
    assert os.path.exists(str(tmp_path / "logs" / "clip_avgpool_object_presence.pt"))


def test_feature_cache_all_five_embedding_keys_and_imagenet_probe(tmp_path):
    """thor_image_features.py:129-138 stores FIVE embeddings per frame: imagenet_conv / imagenet_avgpool (torchvision ResNet-50
    minus avgpool + fc, :46-54,102-106) next to the three clip_* keys; reachable_image_features.py:94-98 the three pooled ones.
    Raw 300x300 frames (thor_frames.py:33-34) take ONE Pillow-exact resize for both towers.  The ImageNet embeddings are
    checked against oracle/tv_resnet.py (pinned vs HuggingFace ResNetModel); then the probe CLI runs on imagenet_avgpool."""
    from embodied_clip_amd import probe_data as pd
    from embodied_clip_amd import probe_train
    from oracle import preprocess as opp
    from oracle import tv_resnet as otv
    sd_clip, sd_tv = syn.rn50_visual_state_dict(0), syn.tv_resnet_state_dict(0)
    ex = pd.ClipFeatureExtractor(sd_clip, device="cuda:0", batch=4, imagenet_state_dict=sd_tv)
    raw = syn.synthetic_rgb_u8(12, 3, 300)
    pts = [{"frame": raw[i], "object_presence": torch.zeros(52, dtype=torch.int64),
            "object_localization": torch.zeros(9, 52, dtype=torch.int64), "free_space": i} for i in range(3)]
    feats = pd.build_thor_features(ex, {"FloorPlan1": pts})
    f0 = feats["FloorPlan1"][0]
    assert set(f0) == {"imagenet_conv", "imagenet_avgpool", "clip_conv", "clip_attnpool", "clip_avgpool", "object_presence",
                       "object_localization", "free_space"}
    assert f0["imagenet_conv"].shape == (2048, 7, 7) and f0["imagenet_conv"].dtype == torch.float32
    assert f0["imagenet_avgpool"].shape == (2048,)
    # oracle: Pillow-exact resize (oracle/preprocess.py, checked against PIL) -> ImageNet normalisation -> torchvision trunk
    resized = torch.stack([torch.from_numpy(opp.clip_resize_crop_u8(raw[i].numpy(), 224)) for i in range(2)])
    ref_conv, ref_avg = otv.imagenet_features(syn.normalize_rgb_imagenet(resized), sd_tv)
    rel = lambda a, b: float((a - b).norm() / b.norm())  # noqa: E731
    got_conv = torch.stack([feats["FloorPlan1"][i]["imagenet_conv"] for i in range(2)])
    got_avg = torch.stack([feats["FloorPlan1"][i]["imagenet_avgpool"] for i in range(2)])
    assert rel(got_conv, ref_conv) < 2e-2, rel(got_conv, ref_conv)
    assert rel(got_avg, ref_avg) < 2e-2
    # cache round trip through the reader, every embedding type
    d = str(tmp_path / "data")
    for split in ("train", "val", "test"):
        pd.write_thor_cache(d, split, feats)
    reach = pd.build_reachable_features(ex, {f"img{i}": raw[i] for i in range(3)})
    assert set(reach["img0"]) == {"imagenet_avgpool", "clip_avgpool", "clip_attnpool"}
    assert torch.equal(reach["img1"]["imagenet_avgpool"], feats["FloorPlan1"][1]["imagenet_avgpool"])
    pd.write_reachable_cache(d, reach, {s: pd.synthetic_reachability(k, list(reach), 5) for k, s in enumerate(("train", "val", "test"))})
    for emb in pd.EMBEDDING_TYPES:
        for task in pd.PREDICTION_TYPES:
            if task == "object_localization" and emb == "clip_attnpool":
                continue
            ds = pd.THOREmbeddingsDataset(d, "train", emb, task)
            x, _y = ds[0]
            want = {"imagenet_avgpool": "imagenet_", "clip_avgpool": "clip_", "clip_attnpool": "clip_"}[emb]
            if task == "object_localization":
                assert torch.equal(x, f0[want + "conv"])
            elif task != "reachability":
                assert torch.equal(x, f0[emb])
    # probe CLI over the ImageNet embeddings, synthetic cache with both towers
    d2 = str(tmp_path / "data2")
    probe_train.main(["--data-dir", d2, "--log-dir", str(tmp_path / "logs"), "--synthetic-frames", "40", "--epochs", "2",
                      "--batch-size", "16", "--embedding-type", "imagenet_avgpool", "--prediction-type", "object_presence"])
    probe_train.main(["--data-dir", d2, "--log-dir", str(tmp_path / "logs"), "--epochs", "1", "--batch-size", "16",
                      "--embedding-type", "imagenet_avgpool", "--prediction-type", "object_localization"])
    assert os.path.exists(str(tmp_path / "logs" / "imagenet_avgpool_object_presence.pt"))
