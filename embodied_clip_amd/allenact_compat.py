"""AllenAct base abstractions the plugin surface is written against.

When ``allenact`` is importable (the reference installs it next to the plugin:
readme_files/baselines_robothor_objectnav.md:25) the REAL classes are used, so
``ClipResNetPreprocessor`` *is a* ``allenact.base_abstractions.preprocessor.Preprocessor``,
the policy *is an* ``allenact.algorithms.onpolicy_sync.policy.ActorCriticModel``
and the engine's own ``Memory`` / ``ActorCriticOutput`` / ``CategoricalDistr``
objects flow through unchanged.  When it is not importable (this build image),
behaviour-equivalent stand-ins restated from [U] allenai/allenact ~v0.5.0 are
used instead (same pattern as ``spaces.py`` for gym):

  * ``Memory``              allenact/base_abstractions/misc.py
  * ``ActorCriticOutput``   allenact/base_abstractions/misc.py
  * ``CategoricalDistr``    allenact/base_abstractions/distributions.py
  * ``Preprocessor``        allenact/base_abstractions/preprocessor.py
  * ``ActorCriticModel``    allenact/algorithms/onpolicy_sync/policy.py
  * ``AbstractActorCriticLoss`` allenact/algorithms/onpolicy_sync/losses/abstract_loss.py

``install_into_allenact()`` makes experiment configs pick up the HIP-backed
classes WITHOUT editing their imports.
"""
from __future__ import annotations

import abc
import importlib
from typing import Any, Dict, Generic, List, Optional, Sequence, Tuple, TypeVar, Union

import torch
import torch.nn as nn

HAVE_ALLENACT = False
try:  # pragma: no cover - allenact is absent in the build image; exercised with a fake package in tests
    from allenact.base_abstractions.misc import ActorCriticOutput, Memory  # type: ignore
    from allenact.base_abstractions.distributions import CategoricalDistr  # type: ignore
    from allenact.base_abstractions.preprocessor import Preprocessor  # type: ignore
    from allenact.algorithms.onpolicy_sync.policy import ActorCriticModel  # type: ignore
    from allenact.algorithms.onpolicy_sync.losses.abstract_loss import AbstractActorCriticLoss  # type: ignore
    HAVE_ALLENACT = True
except Exception:  # noqa: BLE001
    DistributionType = TypeVar("DistributionType")

    class Memory(dict):  # type: ignore
        """key -> (tensor, sampler_dim).  Restates every method of upstream's ``Memory``."""

        def __init__(self, *args, **kwargs):
            super().__init__()
            if len(args) > 0:
                assert len(args) == 1, "Only one of Sequence[Tuple[str, Tuple[torch.Tensor, int]]] or Dict accepted"
                if isinstance(args[0], dict):
                    for key in args[0]:
                        tensor, sampler_dim = args[0][key]
                        self.check_append(key, tensor, sampler_dim)
                else:
                    for key, tensor_dim in args[0]:
                        self.check_append(key, tensor_dim[0], tensor_dim[1])
            elif len(kwargs) > 0:
                for key in kwargs:
                    tensor, sampler_dim = kwargs[key]
                    self.check_append(key, tensor, sampler_dim)

        def check_append(self, key: str, tensor: torch.Tensor, sampler_dim: int) -> "Memory":
            assert isinstance(key, str), "key {} must be str".format(key)
            assert isinstance(tensor, torch.Tensor), "tensor {} must be torch.Tensor".format(tensor)
            assert isinstance(sampler_dim, int), "sampler_dim {} must be int".format(sampler_dim)
            assert key not in self, "Reused key {}".format(key)
            assert 0 <= sampler_dim < len(tensor.shape), "invalid sampler_dim {} for tensor {}".format(
                sampler_dim, tuple(tensor.shape))
            self[key] = (tensor, sampler_dim)
            return self

        def tensor(self, key: str) -> torch.Tensor:
            assert key in self, "Missing key {}".format(key)
            return self[key][0]

        def sampler_dim(self, key: str) -> int:
            assert key in self, "Missing key {}".format(key)
            return self[key][1]

        def sampler_select(self, keep: Sequence[int]) -> "Memory":
            res = Memory()
            valid = False
            for name in self:
                sampler_dim = self.sampler_dim(name)
                tensor = self.tensor(name)
                assert len(keep) == 0 or (0 <= min(keep) and max(keep) < tensor.shape[sampler_dim]), \
                    "Got min(keep)={} max(keep)={} for memory type {} with shape {}, dim {}".format(
                        min(keep), max(keep), name, tensor.shape, sampler_dim)
                if tensor.shape[sampler_dim] > len(keep):
                    tensor = tensor.index_select(
                        dim=sampler_dim, index=torch.as_tensor(list(keep), dtype=torch.int64, device=tensor.device))
                    res.check_append(name, tensor, sampler_dim)
                    valid = True
            return res if valid else self

        def index_select(self, keep: Sequence[int]) -> "Memory":   # older upstream name of sampler_select
            return self.sampler_select(keep)

        def set_tensor(self, key: str, tensor: torch.Tensor) -> "Memory":
            assert key in self, "Missing key {}".format(key)
            assert tensor.shape == self[key][0].shape, "setting tensor with shape {} for former {}".format(
                tensor.shape, self[key][0].shape)
            self[key] = (tensor, self[key][1])
            return self

        def step_select(self, step: int) -> "Memory":
            res = Memory()
            for key in self:
                tensor = self.tensor(key)
                assert tensor.shape[0] > step, "attempting to access step {} for memory type {} of shape {}".format(
                    step, key, tensor.shape)
                if step != -1:
                    res.check_append(key, self.tensor(key)[step:step + 1, ...], self.sampler_dim(key))
                else:
                    res.check_append(key, self.tensor(key)[step:, ...], self.sampler_dim(key))
            return res

        def step_squeeze(self, step: int) -> "Memory":
            res = Memory()
            for key in self:
                tensor = self.tensor(key)
                assert tensor.shape[0] > step, "attempting to access step {} for memory type {} of shape {}".format(
                    step, key, tensor.shape)
                res.check_append(key, self.tensor(key)[step, ...], self.sampler_dim(key) - 1)
            return res

        def slice(self, dim: int, start: Optional[int] = None, stop: Optional[int] = None, step: int = 1) -> "Memory":
            checked, total = False, None
            res = Memory()
            for key in self:
                tensor = self.tensor(key)
                assert len(tensor.shape) > dim, "attempting to access dim {} for memory {} of shape {}".format(
                    dim, key, tensor.shape)
                if not checked:
                    total, checked = tensor.shape[dim], True
                assert total == tensor.shape[dim], "attempting to slice along non-uniform dimension {}".format(dim)
                if start is not None or stop is not None or step != 1:
                    slice_tuple = (slice(None),) * dim + (slice(start, stop, step),)
                    res.check_append(key, tensor[slice_tuple], self.sampler_dim(key))
                else:
                    res.check_append(key, tensor, self.sampler_dim(key))
            return res

        def to(self, device: torch.device) -> "Memory":
            for key in self:
                tensor = self.tensor(key)
                if tensor.device != device:
                    self.set_tensor(key, tensor.to(device))
            return self

    class ActorCriticOutput(tuple, Generic[DistributionType]):  # type: ignore
        """``ActorCriticOutput(distributions, values, extras)`` -- a 3-tuple with named fields, like upstream's
        generic tuple (so ``out.values``, ``out[1]`` and ``d, v, e = out`` all work)."""

        def __new__(cls, distributions, values, extras):
            return super().__new__(cls, (distributions, values, extras))

        distributions = property(lambda self: self[0])
        values = property(lambda self: self[1])
        extras = property(lambda self: self[2])

    class CategoricalDistr(torch.distributions.Categorical):  # type: ignore
        """``CategoricalDistr(logits=...)``: torch Categorical + ``mode`` / tensor views / step-shaped ``log_prob``.
        (``.logits`` are normalised log-probabilities, as in torch.)"""

        def mode(self):
            return self._param.argmax(dim=-1, keepdim=False)

        def log_prob(self, value: torch.Tensor):
            if value.shape == self.logits.shape[:-1]:
                return super().log_prob(value)
            if value.shape == self.logits.shape[:-1] + (1,):
                return super().log_prob(value.squeeze(-1)).unsqueeze(-1)
            raise NotImplementedError(
                "log_prob: value shape {} does not match logits shape {}".format(value.shape, self.logits.shape))

        @property
        def log_probs_tensor(self):
            return self.logits

        @property
        def probs_tensor(self):
            return self.probs

    class Preprocessor(abc.ABC):  # type: ignore
        """``Preprocessor(input_uuids, output_uuid, observation_space)`` with abstract ``process`` / ``to``."""

        input_uuids: List[str]
        uuid: str
        observation_space: Any

        def __init__(self, input_uuids: List[str], output_uuid: str, observation_space, **kwargs: Any) -> None:
            self.uuid = output_uuid
            self.input_uuids = input_uuids
            self.observation_space = observation_space

        @abc.abstractmethod
        def process(self, obs: Dict[str, Any], *args: Any, **kwargs: Any) -> Any:
            raise NotImplementedError()

        @abc.abstractmethod
        def to(self, device: torch.device) -> "Preprocessor":
            raise NotImplementedError()

    class ActorCriticModel(Generic[DistributionType], nn.Module):  # type: ignore
        """ABC of every AllenAct policy: spaces, the validated public ``recurrent_memory_specification`` property
        over the abstract ``_recurrent_memory_specification()``, abstract ``forward``."""

        def __init__(self, action_space, observation_space):
            super().__init__()
            self.action_space = action_space
            self.observation_space = observation_space
            self.memory_spec = None

        @property
        def recurrent_memory_specification(self):
            if self.memory_spec is None:
                self.memory_spec = [self._recurrent_memory_specification()]
            spec = self.memory_spec[0]
            if spec is None:
                return None
            for key in spec:
                dims, _ = spec[key]
                dim_names = [d[0] for d in dims]
                assert "step" not in dim_names, "`step` is automatically added and cannot be reused"
                assert "sampler" in dim_names, "`sampler` dim must be defined"
            return self.memory_spec[0]

        @abc.abstractmethod
        def _recurrent_memory_specification(self):
            raise NotImplementedError()

        @abc.abstractmethod
        def forward(self, observations, memory, prev_actions, masks):
            raise NotImplementedError()

    class AbstractActorCriticLoss(abc.ABC):  # type: ignore
        """``loss(step_count, batch, actor_critic_output) -> (scalar tensor, info dict)``."""

        def __init__(self, *args, **kwargs):
            pass

        @abc.abstractmethod
        def loss(self, step_count: int, batch, actor_critic_output, *args, **kwargs):
            raise NotImplementedError()


# The module attributes the reference's experiment configs import the plugin classes from ([U] allenact ~v0.5.0;
# plugin path readme_files/baselines_robothor_objectnav.md:25, config :51)
_PATCH_TARGETS = (
    ("allenact_plugins.clip_plugin.clip_preprocessors", ("ClipResNetPreprocessor", "ClipViTPreprocessor")),
    ("projects.objectnav_baselines.models.object_nav_models", ("ResnetTensorObjectNavActorCritic",)),
    ("allenact_plugins.robothor_plugin.robothor_models", ("ResnetTensorObjectNavActorCritic",)),
    ("allenact.algorithms.onpolicy_sync.losses.ppo", ("PPO",)),
    ("allenact.algorithms.onpolicy_sync.losses", ("PPO",)),
)


def install_into_allenact(verbose: bool = False) -> List[str]:
    """Rebind the plugin classes inside the already-installed AllenAct modules to the HIP-backed ones, so experiment
    configs (``from allenact_plugins.clip_plugin.clip_preprocessors import ClipResNetPreprocessor`` ...) run
    unchanged.  Call before the experiment config module is imported (``python -m embodied_clip_amd.allenact_main``
    does).  Returns the ``module.attr`` names that were patched; modules that are not installed are skipped."""
    from . import clip_preprocessors as cp
    from . import policy as pol
    from . import ppo
    ours = {"ClipResNetPreprocessor": cp.ClipResNetPreprocessor, "ClipViTPreprocessor": cp.ClipViTPreprocessor,
            "ResnetTensorObjectNavActorCritic": pol.ResnetTensorObjectNavActorCritic, "PPO": ppo.PPO}
    done = []
    for modname, attrs in _PATCH_TARGETS:
        try:
            mod = importlib.import_module(modname)
        except Exception:  # noqa: BLE001 - optional module
            continue
        for a in attrs:
            if hasattr(mod, a) or modname.endswith(("clip_preprocessors", "object_nav_models")):
                setattr(mod, a, ours[a])
                done.append(f"{modname}.{a}")
    if verbose:
        print("embodied_clip_amd: patched " + (", ".join(done) if done else "nothing (allenact not installed)"))
    return done
