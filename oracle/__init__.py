"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

torch-CPU fp32 restatement of the hot path named by BASELINE.json
(frozen CLIP visual encoder forward -> GRU actor-critic fwd/bwd -> GAE/PPO
update).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; the product
(``embodied_clip_amd``) never does and fails loudly without its HIP library.

PARITY UNPINNED: the reference repository (allenai/embodied-clip @ main)
holds no tests, golden vectors or fixtures for this path, and none of the
third-party modules that hold its arithmetic can be imported here
(SURVEY.md §8c):

  * openai/CLIP @ 40f5484c1c74edd83cb9cf687c6ab92b28d8b656
    (``primitive_probing/environment.yml:22``) -- ``clip/model.py``:
    ModifiedResNet, Bottleneck, AttentionPool2d, VisionTransformer,
    ResidualAttentionBlock.
  * allenai/allenact ~v0.5.0 (``readme_files/baselines_robothor_objectnav.md:6``)
    -- clip_plugin preprocessors, ResnetTensorObjectNavActorCritic,
    RNNStateEncoder, PPO loss, RolloutStorage.compute_returns.

Each function restates the *published* algorithm of those modules and is
anchored on the reference's own call sites (cited per function).  Pins that
do exist: op-level composition from torch-CPU ops, parameter-count
checksums (38,316,896 / 87,849,216 / 3,480,775), an independent
implementation cross-check of the ViT tower against HuggingFace
``transformers.CLIPVisionModel`` (tests/test_oracle_vit_hf.py), and
self-generated golden vectors under tests/golden/.
"""
