"""Activation factory (reference modules/activations.py:26-67).

Only plain (unscaled) activations resolve here; the scaled / phase activations of
the reference are legacy options outside the generate path.
"""
import dataclasses as dtc
from enum import auto
from typing import Dict

import torch
from torch import nn

from ..config import Config, private_runtime_field
from ..utils import AutoStrEnum

__all__ = ["ActivationEnum", "ActivationConfig", "Abs", "Sin", "Cos"]


class ActivationEnum(AutoStrEnum):
    Tanh = auto()
    Sigmoid = auto()
    Mish = auto()
    ReLU = auto()
    Softplus = auto()
    Identity = auto()
    Abs = auto()
    Sin = auto()
    Cos = auto()
    GLU = auto()
    Softmax = auto()


class Abs(nn.Module):
    def forward(self, x):
        return x.abs()


class Sin(nn.Module):
    def forward(self, x):
        return torch.sin(x)


class Cos(nn.Module):
    def forward(self, x):
        return torch.cos(x)


_LOCAL = {"Abs": Abs, "Sin": Sin, "Cos": Cos}


@dtc.dataclass
class ActivationConfig(Config, type_field=False):
    act: ActivationEnum = "Identity"
    scaled: bool = False
    static: bool = False
    with_rate: bool = False
    params: Dict = dtc.field(default_factory=dict)
    dim: int = private_runtime_field(None)

    def get(self) -> nn.Module:
        name = str(self.act)
        if self.scaled:
            raise NotImplementedError("scaled activations are outside the generate path covered here")
        cls = _LOCAL.get(name) or getattr(nn, name, None)
        if cls is None:
            raise ValueError(f"unknown activation '{name}'")
        if name == "Softmax":
            return cls(dim=-1, **self.params)
        return cls(**self.params)
