"""CPU: the AllenAct plugin surface.

  * ``Memory`` restatement: every method upstream's engine/storage calls on it;
  * with an ``allenact`` package importable (a minimal stand-in tree written to a temp dir: the real package is
    not installable here) the plugin classes SUBCLASS its ABCs, use its Memory / ActorCriticOutput /
    CategoricalDistr, and ``install_into_allenact()`` rebinds the classes inside the allenact modules the
    reference's experiment configs import from -- i.e. configs run unchanged;
  * constructor contracts (uuids, observation spaces, depth-only tower, RGB-D rejected) -- no GPU compute.
"""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_memory_api_matches_upstream_semantics():
    from embodied_clip_amd.allenact_compat import HAVE_ALLENACT, Memory
    assert not HAVE_ALLENACT                      # the build image has no allenact: this exercises the restatement
    rnn = torch.arange(5 * 1 * 3 * 4, dtype=torch.float32).view(5, 1, 3, 4)     # [step, layer, sampler, hidden]
    aux = torch.arange(5 * 3 * 2, dtype=torch.float32).view(5, 3, 2)           # [step, sampler, x]
    m = Memory([("rnn", (rnn, 2)), ("aux", (aux, 1))])
    assert m.tensor("rnn") is rnn and m.sampler_dim("aux") == 1
    with pytest.raises(AssertionError):
        m.check_append("rnn", rnn, 2)                                          # reused key
    with pytest.raises(AssertionError):
        Memory().check_append("x", rnn, 4)                                     # sampler_dim out of range
    s1 = m.step_select(1)
    assert s1.tensor("rnn").shape == (1, 1, 3, 4) and torch.equal(s1.tensor("rnn")[0], rnn[1]) and s1.sampler_dim("rnn") == 2
    last = m.step_select(-1)
    assert torch.equal(last.tensor("aux"), aux[-1:])
    sq = m.step_squeeze(2)
    assert sq.tensor("rnn").shape == (1, 3, 4) and sq.sampler_dim("rnn") == 1 and sq.sampler_dim("aux") == 0
    assert torch.equal(sq.tensor("aux"), aux[2])
    sl = m.slice(dim=0, start=1, stop=4)
    assert sl.tensor("rnn").shape[0] == 3 and torch.equal(sl.tensor("aux"), aux[1:4])
    assert m.slice(dim=0).tensor("rnn") is rnn
    with pytest.raises(AssertionError):
        m.slice(dim=1)                                                         # non-uniform dimension
    keep = m.sampler_select([0, 2])
    assert keep.tensor("rnn").shape == (5, 1, 2, 4) and torch.equal(keep.tensor("aux"), aux[:, [0, 2]])
    assert m.sampler_select([0, 1, 2]) is m                                    # nothing dropped -> same object
    assert m.index_select([1]).tensor("rnn").shape == (5, 1, 1, 4)
    with pytest.raises(AssertionError):
        m.set_tensor("rnn", torch.zeros(1))                                    # shape must match
    m2 = Memory(rnn=(rnn.clone(), 2)).set_tensor("rnn", torch.ones_like(rnn))
    assert float(m2.tensor("rnn").sum()) == rnn.numel()
    assert Memory({"rnn": (rnn, 2)}).to(torch.device("cpu")).tensor("rnn") is rnn


def test_output_and_distribution_standins():
    from embodied_clip_amd.allenact_compat import ActorCriticOutput, CategoricalDistr
    lg = torch.randn(2, 3, 6)
    d = CategoricalDistr(logits=lg)
    out = ActorCriticOutput(distributions=d, values=torch.zeros(2, 3, 1), extras={})
    dist, values, extras = out
    assert dist is d and out.values is values and out[2] is extras
    a = d.sample()
    assert a.shape == (2, 3)
    lp = torch.log_softmax(lg, -1)
    assert torch.allclose(d.log_prob(a), lp.gather(-1, a.unsqueeze(-1)).squeeze(-1), atol=1e-6)
    assert d.log_prob(a.unsqueeze(-1)).shape == (2, 3, 1)
    assert torch.allclose(d.entropy(), -(lp.exp() * lp).sum(-1), atol=1e-6)
    assert torch.equal(d.mode(), lg.argmax(-1))
    assert torch.allclose(d.log_probs_tensor, lp, atol=1e-6) and torch.allclose(d.probs_tensor, lp.exp(), atol=1e-6)


def test_constructor_contracts_without_a_gpu():
    from embodied_clip_amd import spaces
    from embodied_clip_amd.allenact_compat import ActorCriticModel, Preprocessor
    from embodied_clip_amd.clip_preprocessors import ClipResNetPreprocessor, ClipViTPreprocessor
    from embodied_clip_amd.policy import ResnetTensorObjectNavActorCritic
    from embodied_clip_amd.ppo import PPO
    from embodied_clip_amd.allenact_compat import AbstractActorCriticLoss
    p = ClipResNetPreprocessor(rgb_input_uuid="rgb_lowres", clip_model_type="RN50", pool=False,
                               output_uuid="rgb_clip_resnet", device=torch.device("cuda:0"))
    assert isinstance(p, Preprocessor) and p.input_uuids == ["rgb_lowres"] and p.uuid == "rgb_clip_resnet"
    assert p.observation_space.shape == (2048, 7, 7)
    assert ClipResNetPreprocessor("rgb", "RN50", True).observation_space.shape == (2048,)
    assert ClipResNetPreprocessor("rgb", "RN50x16", False).observation_space.shape == (3072, 7, 7)
    v = ClipViTPreprocessor("rgb", "ViT-B/32", class_emb_only=True)
    assert isinstance(v, Preprocessor) and v.observation_space.shape == (768,)
    assert max(abs(a - b) for a, b in zip(p.CLIP_RGB_MEANS, (0.48145466, 0.4578275, 0.40821073))) < 1e-8
    assert issubclass(ResnetTensorObjectNavActorCritic, ActorCriticModel) and issubclass(PPO, AbstractActorCriticLoss)
    obs = spaces.Dict({"rgb_clip_resnet": spaces.Box(-1, 1, (2048, 7, 7)), "goal": spaces.Discrete(12)})
    with pytest.raises(ValueError):              # RGB-D dual tower: its depth tensor must be in the observation space
        ResnetTensorObjectNavActorCritic(spaces.Discrete(6), obs, "goal", "rgb_clip_resnet", "depth_clip_resnet")
    with pytest.raises(ValueError):
        ResnetTensorObjectNavActorCritic(spaces.Discrete(6), obs, "goal")


FAKE_ALLENACT = {
    "allenact/__init__.py": "",
    "allenact/base_abstractions/__init__.py": "",
    "allenact/base_abstractions/misc.py": """
        import torch
        class Memory(dict):
            MARK = 'fake-allenact'
            def check_append(self, key, tensor, sampler_dim):
                self[key] = (tensor, sampler_dim); return self
            def tensor(self, key): return self[key][0]
            def sampler_dim(self, key): return self[key][1]
            def set_tensor(self, key, tensor):
                self[key] = (tensor, self[key][1]); return self
        class ActorCriticOutput(tuple):
            def __new__(cls, distributions, values, extras): return super().__new__(cls, (distributions, values, extras))
        """,
    "allenact/base_abstractions/distributions.py": """
        import torch
        class CategoricalDistr(torch.distributions.Categorical):
            MARK = 'fake-allenact'
        """,
    "allenact/base_abstractions/preprocessor.py": """
        import abc
        class Preprocessor(abc.ABC):
            MARK = 'fake-allenact'
            def __init__(self, input_uuids, output_uuid, observation_space, **kwargs):
                self.uuid, self.input_uuids, self.observation_space = output_uuid, input_uuids, observation_space
        """,
    "allenact/algorithms/__init__.py": "",
    "allenact/algorithms/onpolicy_sync/__init__.py": "",
    "allenact/algorithms/onpolicy_sync/policy.py": """
        import abc, torch.nn as nn
        class ActorCriticModel(nn.Module):
            MARK = 'fake-allenact'
            def __init__(self, action_space, observation_space):
                super().__init__(); self.action_space, self.observation_space = action_space, observation_space
            @property
            def recurrent_memory_specification(self): return self._recurrent_memory_specification()
        """,
    "allenact/algorithms/onpolicy_sync/losses/__init__.py": "from .ppo import PPO\n",
    "allenact/algorithms/onpolicy_sync/losses/abstract_loss.py": """
        class AbstractActorCriticLoss:
            MARK = 'fake-allenact'
            def __init__(self, *a, **k): pass
        """,
    "allenact/algorithms/onpolicy_sync/losses/ppo.py": """
        from .abstract_loss import AbstractActorCriticLoss
        class PPO(AbstractActorCriticLoss):
            ORIGINAL = True
        """,
    "allenact_plugins/__init__.py": "",
    "allenact_plugins/clip_plugin/__init__.py": "",
    "allenact_plugins/clip_plugin/clip_preprocessors.py": """
        class ClipResNetPreprocessor: ORIGINAL = True
        class ClipViTPreprocessor: ORIGINAL = True
        """,
    "projects/__init__.py": "",
    "projects/objectnav_baselines/__init__.py": "",
    "projects/objectnav_baselines/models/__init__.py": "",
    "projects/objectnav_baselines/models/object_nav_models.py": """
        class ResnetTensorObjectNavActorCritic: ORIGINAL = True
        """,
    # what an unchanged experiment config does
    "experiment_config.py": """
        from allenact_plugins.clip_plugin.clip_preprocessors import ClipResNetPreprocessor
        from projects.objectnav_baselines.models.object_nav_models import ResnetTensorObjectNavActorCritic
        from allenact.algorithms.onpolicy_sync.losses import PPO
        """,
}


def test_subclasses_real_abcs_and_patches_configs_when_allenact_is_importable(tmp_path):
    for rel, src in FAKE_ALLENACT.items():
        f = tmp_path / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(textwrap.dedent(src))
    prog = textwrap.dedent("""
        import torch
        from embodied_clip_amd import allenact_compat as ac
        assert ac.HAVE_ALLENACT
        import allenact.base_abstractions.preprocessor as up_pre, allenact.algorithms.onpolicy_sync.policy as up_pol
        import allenact.base_abstractions.misc as up_misc
        from embodied_clip_amd.clip_preprocessors import ClipResNetPreprocessor, ClipViTPreprocessor
        from embodied_clip_amd.policy import ResnetTensorObjectNavActorCritic, Memory, CategoricalDistr
        from embodied_clip_amd.ppo import PPO
        assert issubclass(ClipResNetPreprocessor, up_pre.Preprocessor) and issubclass(ClipViTPreprocessor, up_pre.Preprocessor)
        assert issubclass(ResnetTensorObjectNavActorCritic, up_pol.ActorCriticModel)
        assert Memory is up_misc.Memory and CategoricalDistr.MARK == 'fake-allenact'
        import allenact.algorithms.onpolicy_sync.losses.abstract_loss as up_loss
        assert issubclass(PPO, up_loss.AbstractActorCriticLoss)
        p = ClipResNetPreprocessor('rgb', 'RN50', False, device=torch.device('cuda:0'))
        assert p.MARK == 'fake-allenact' and p.uuid == 'rgb_clip_resnet'
        done = ac.install_into_allenact()
        assert 'allenact_plugins.clip_plugin.clip_preprocessors.ClipResNetPreprocessor' in done, done
        import experiment_config as cfg            # imported AFTER the patch, exactly as allenact_main does
        assert cfg.ClipResNetPreprocessor is ClipResNetPreprocessor
        assert cfg.ResnetTensorObjectNavActorCritic is ResnetTensorObjectNavActorCritic
        assert cfg.PPO is PPO and not hasattr(cfg.PPO, 'ORIGINAL')
        print('OK', len(done))
    """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
