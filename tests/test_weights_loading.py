"""CPU: real-checkpoint loading paths.  The reference obtains its weights with
``clip.load('RN50', device)`` (primitive_probing/generate_data/thor_image_features.py:57-60): a TorchScript archive
whose state_dict keys carry a ``visual.`` prefix and whose tensors are fp16.  No real checkpoint exists in this image,
so the loaders are exercised on synthetic towers saved in those formats; packing must equal packing the fp32 tower
(after the same fp16 rounding)."""
import os

import pytest
import torch
import torch.nn as nn

from embodied_clip_amd import synthetic as syn
from embodied_clip_amd.clip_preprocessors import _load_visual_state_dict, load_checkpoint_state_dict, visual_state_dict
from embodied_clip_amd.encoder import pack_rn50, pack_vit

KW = dict(width=16, layers=(1, 1, 1, 1), output_dim=32, heads=4, input_resolution=64)


def _whole_clip_fp16(sd):
    full = {"visual." + k: v.half() for k, v in sd.items()}
    full["token_embedding.weight"] = torch.zeros(10, 8).half()          # text-tower / logit entries must be ignored
    full["logit_scale"] = torch.tensor(4.6).half()
    return full


def _assert_same_pack(a, b):
    (cfg_a, stem_a, w_a, bias_a), (cfg_b, stem_b, w_b, bias_b) = a, b
    assert cfg_a == cfg_b
    assert torch.equal(stem_a, stem_b) and torch.equal(w_a, w_b) and torch.equal(bias_a, bias_b)


def test_state_dict_file_with_visual_prefix_and_fp16(tmp_path):
    sd = syn.rn50_visual_state_dict(3, **KW)
    path = tmp_path / "RN50.pt"
    torch.save(_whole_clip_fp16(sd), path)
    got = _load_visual_state_dict("RN50", None, str(path))
    assert set(got) == set(sd) and all(v.dtype == torch.float16 for v in got.values())
    ref = {k: v.half().float() for k, v in sd.items()}                   # what fp16 storage does to the weights
    _assert_same_pack(pack_rn50(got), pack_rn50(ref))
    # the dict can also be handed over directly (state_dict=), prefixed or not
    _assert_same_pack(pack_rn50(_load_visual_state_dict("RN50", _whole_clip_fp16(sd), None)), pack_rn50(ref))
    assert visual_state_dict(sd) is sd


def test_weights_dir_environment_variable(tmp_path, monkeypatch):
    sd = syn.vit_visual_state_dict(2, width=64, layers=2, heads=2, patch_size=32, input_resolution=64, output_dim=32)
    torch.save({"visual." + k: v for k, v in sd.items()}, tmp_path / "ViT-B-32.pt")     # "ViT-B/32" -> "ViT-B-32.pt"
    monkeypatch.setenv("EC_CLIP_WEIGHTS_DIR", str(tmp_path))
    got = _load_visual_state_dict("ViT-B/32", None, None)
    (cfg, w, f), (cfg2, w2, f2) = pack_vit(got), pack_vit(sd)
    assert cfg == cfg2 and torch.equal(w, w2) and torch.equal(f, f2)
    with pytest.raises(RuntimeError):
        _load_visual_state_dict("RN50", None, None)                     # no RN50.pt there and no `clip` package


def test_torchscript_archive_like_the_published_checkpoints(tmp_path):
    """``clip.load`` reads the downloaded file with ``torch.jit.load``: a scripted module whose parameters sit under
    ``visual.`` (fp16), next to other towers."""
    sd = syn.rn50_visual_state_dict(4, **KW)

    class Holder(nn.Module):
        def __init__(self, tensors):
            super().__init__()
            for k, v in tensors.items():
                mod, parts = self, k.split(".")
                for part in parts[:-1]:
                    if not hasattr(mod, part):
                        mod.add_module(part, nn.Module())
                    mod = getattr(mod, part)
                if v.is_floating_point() and "running" not in parts[-1] and "num_batches" not in parts[-1]:
                    mod.register_parameter(parts[-1], nn.Parameter(v.clone(), requires_grad=False))
                else:
                    mod.register_buffer(parts[-1], v.clone())

        def forward(self, x: torch.Tensor) -> torch.Tensor:
            return x

    full = Holder(_whole_clip_fp16(sd))
    path = str(tmp_path / "RN50.pt")
    torch.jit.script(full).save(path)
    loaded = load_checkpoint_state_dict(path)
    assert "visual.conv1.weight" in loaded and "token_embedding.weight" in loaded
    got = _load_visual_state_dict("RN50", None, path)
    assert set(got) == set(sd)
    _assert_same_pack(pack_rn50(got), pack_rn50({k: v.half().float() for k, v in sd.items()}))
