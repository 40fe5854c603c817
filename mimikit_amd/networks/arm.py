"""The auto-regressive network protocol -- THE drop-in boundary of this package.

Mirrors the reference's ``ARM`` / ``ARMWithHidden`` / ``NetworkConfig``
(networks/arm.py:19-87): the generate loop and the trainer callbacks only ever
talk to a network through ``rf``, ``generate_params``, ``before_generate``,
``generate_step``, ``after_generate`` (and ``train_batch`` / ``test_batch`` for
prompt shapes).  Networks here additionally expose ``generate_block`` so the
loop can hand all steps of a batch to the device in one call.
"""
import abc
import dataclasses as dtc
from typing import Dict, Optional, Set, Tuple

import torch
import torch.nn as nn

from ..config import Config, Configurable
from ..features.item_spec import ItemSpec
from ..io_spec import IOSpec

__all__ = ["NetworkConfig", "ARM", "ARMWithHidden"]


@dtc.dataclass
class NetworkConfig(Config, abc.ABC):
    @property
    @abc.abstractmethod
    def io_spec(self) -> IOSpec:
        ...


class ARM(Configurable, torch.nn.Module):
    """Interface for Auto Regressive Networks"""

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    @abc.abstractmethod
    def config(self) -> NetworkConfig:
        ...

    @property
    @abc.abstractmethod
    def rf(self) -> int:
        ...

    @abc.abstractmethod
    def train_batch(self, item_spec: ItemSpec):
        ...

    @abc.abstractmethod
    def test_batch(self, item_spec: ItemSpec):
        ...

    @abc.abstractmethod
    def before_generate(self, prompts: Tuple[torch.Tensor, ...], batch_index) -> None:
        ...

    @abc.abstractmethod
    def generate_step(self, inputs: Tuple[torch.Tensor, ...], *, t: int = 0,
                      **parameters: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, ...]:
        ...

    @abc.abstractmethod
    def after_generate(self, final_outputs: Tuple[torch.Tensor, ...], batch_index) -> None:
        ...

    @property
    @abc.abstractmethod
    def generate_params(self) -> Set[str]:
        ...

    # -- extension over the reference protocol --------------------------------
    def generate_block(self, tensors: Tuple[torch.Tensor, ...], t0: int, n_steps: int,
                       **parameters) -> Optional[bool]:
        """Run steps t0 .. t0+n_steps-1 of the generate loop on the device, writing every
        target in place into ``tensors`` (the loop's prompt+blank tensors).  Networks
        without a fused path return ``None`` and the loop falls back to calling
        ``generate_step`` once per step (still on the device)."""
        return None


class ARMWithHidden(ARM, abc.ABC):
    @abc.abstractmethod
    def reset_hidden(self) -> None:
        ...


# -- weight_norm (sample_rnn_v2.py:67-81, s2s_lstm_v2.py:86-91 of the reference) ---------------------------------------
def weight_norm_leaves(root: nn.Module) -> None:
    """``nn.utils.weight_norm`` on every parameter of every leaf module (reference :76-81 and SampleRNN.__init__)"""
    for module in root.modules():
        if isinstance(module, nn.ModuleList) or list(module.children()) != []:
            continue
        for name in dict(module.named_parameters()):
            nn.utils.weight_norm(module, name)


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """state_dict as the HIP plan binds it: ``name_g`` / ``name_v`` pairs folded into ``name`` = g v / |v| (norm over
    every dimension but the first, as ``torch._weight_norm(v, g, 0)``), which is what the module's pre-forward hook does"""
    out = {}
    for key, value in sd.items():
        if key.endswith("_v") and key[:-2] + "_g" in sd:
            out[key[:-2]] = torch._weight_norm(value, sd[key[:-2] + "_g"], 0)
        elif key.endswith("_g") and key[:-2] + "_v" in sd:
            continue
        else:
            out[key] = value
    return out
