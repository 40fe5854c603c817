"""Input / output module configs (reference modules/io.py).

Each ``IOModule`` is a config dataclass whose runtime fields (``in_dim``,
``out_dim``, ``frame_size`` ...) are wired in by the io spec and the network,
and whose ``module()`` builds an ``nn.Sequential`` with exactly the reference's
child indices, so ``state_dict`` keys match.
Covered: ``LinearIO`` :115-122, ``FramedLinearIO`` :125-133, ``ChunkedLinearIO``
:136-145, ``EmbeddingIO`` :148-154, ``FramedConv1dIO`` :185-198, ``MLPIO``
:201-220, ``Linearizer`` :106-112, ``ZipReduceVariables`` :289-313.
"""
import abc
import dataclasses as dtc
from enum import auto
from typing import Iterable, Optional, Tuple

import torch
from torch import nn

from ..config import Config, private_runtime_field
from .mlp import MLP
from ..utils import AutoStrEnum
from .activations import ActivationConfig
from .misc import Chunk, Flatten, Unfold, Unsqueeze
from .resamplers import Conv1dResampler
from .targets import OutputWrapper

__all__ = [
    "IOModule", "Linearizer", "LinearIO", "ChunkedLinearIO", "FramedLinearIO", "EmbeddingIO", "FramedConv1dIO",
    "MLPIO", "ZipMode", "ZipReduceVariables",
]


@dtc.dataclass
class IOModule(Config, abc.ABC):
    activation: Optional[ActivationConfig] = None
    dropout: float = 0.
    dropout1d: float = 0.

    in_dim: Optional[int] = private_runtime_field(None)
    out_dim: Optional[int] = private_runtime_field(None)
    hop_length: Optional[int] = private_runtime_field(None)
    frame_size: Optional[int] = private_runtime_field(None)
    class_size: Optional[int] = private_runtime_field(None)
    sampler: Optional[nn.Module] = private_runtime_field(None)
    with_linearizer: bool = private_runtime_field(False)
    with_unfold: bool = private_runtime_field(False)
    with_n_chunks: Optional[int] = private_runtime_field(None)

    def set(self, **kwargs) -> "IOModule":
        """wire runtime attributes once; unknown names and double assignment are errors"""
        for name, value in kwargs.items():
            if not hasattr(self, name):
                raise AttributeError(f"attribute '{name}' not found in IOModule")
            current = getattr(self, name)
            if current is not None:
                raise RuntimeError(f"can not set attribute '{name}'. It has already been set to '{current}'")
            setattr(self, name, value)
        return self

    def not_none(self, *names):
        missing = [n for n in names if getattr(self, n) is None]
        if missing:
            raise ValueError("".join(
                f"- '{n}' can not be None with module_type '{type(self).__qualname__}'\n" for n in missing))

    @abc.abstractmethod
    def module(self) -> nn.Module:
        ...

    def wrap(self, core: nn.Module) -> nn.Module:
        head, tail = [], []
        if self.with_linearizer:
            head.append(Linearizer(self.class_size))
        if self.with_unfold:
            self.not_none("frame_size", "hop_length")
            head.append(Unfold(-1, self.frame_size, self.hop_length))
        if self.with_n_chunks is not None:
            tail.append(Chunk(self.with_n_chunks, dim=-1, sum_outputs=True))
        if self.activation is not None and str(self.activation.act) != "Identity":
            if self.activation.scaled:
                self.activation.dim = self.out_dim
            tail.append(self.activation.get())
        if self.dropout > 0:
            tail.append(nn.Dropout(self.dropout))
        if self.dropout1d > 0:
            tail.append(nn.Dropout1d(self.dropout1d))
        seq = nn.Sequential(*head, core, *tail)
        return OutputWrapper(seq, self.sampler) if self.sampler is not None else seq


class Linearizer(nn.Module):
    """class index -> [-1, 1):  ((q / class_size) - .5) * 2"""

    def __init__(self, class_size: int):
        super().__init__()
        self.class_size = class_size

    def forward(self, x):
        return ((x.float() / self.class_size) - .5) * 2


@dtc.dataclass
class LinearIO(IOModule):
    bias: bool = True

    def module(self) -> nn.Module:
        self.not_none("in_dim", "out_dim")
        return self.wrap(nn.Linear(self.in_dim, self.out_dim, bias=self.bias))


@dtc.dataclass
class FramedLinearIO(IOModule):
    def module(self) -> nn.Module:
        self.not_none("frame_size", "hop_length", "out_dim", "class_size")
        self.with_linearizer = True
        self.with_unfold = True
        return self.wrap(nn.Linear(self.frame_size, self.out_dim))


@dtc.dataclass
class ChunkedLinearIO(IOModule):
    bias: bool = True
    n_chunks: int = 1

    def module(self) -> nn.Module:
        self.not_none("in_dim", "out_dim")
        self.with_n_chunks = self.n_chunks
        return self.wrap(nn.Linear(self.in_dim, self.out_dim * self.n_chunks, bias=self.bias))


@dtc.dataclass
class EmbeddingIO(IOModule):
    def module(self) -> nn.Module:
        self.not_none("class_size", "out_dim")
        return self.wrap(nn.Embedding(self.class_size, self.out_dim))


@dtc.dataclass
class FramedConv1dIO(IOModule):
    def module(self) -> nn.Module:
        self.not_none("frame_size", "out_dim")
        core = nn.Sequential(
            Flatten(-2),       # (batch, n_frames * frame_size)
            Unsqueeze(-1),     # (batch, n_frames * frame_size, 1)
            Conv1dResampler(in_dim=1, t_factor=1 / self.frame_size, d_factor=self.out_dim),
        )
        self.with_linearizer = True
        self.with_unfold = True
        return self.wrap(core)


@dtc.dataclass
class MLPIO(IOModule):
    hidden_dim: int = 128
    n_hidden_layers: int = 1
    activation: ActivationConfig = dtc.field(default_factory=lambda: ActivationConfig("Mish"))
    bias: bool = True
    dropout: float = 0.
    dropout1d: float = 0.
    min_temperature: Optional[float] = 1e-4

    def module(self) -> nn.Module:
        self.not_none("in_dim", "out_dim")
        core = MLP(in_dim=self.in_dim, out_dim=self.out_dim, hidden_dim=self.hidden_dim,
                   n_hidden_layers=self.n_hidden_layers, activation=self.activation.get(), bias=self.bias,
                   dropout=self.dropout, dropout1d=self.dropout1d, min_temperature=self.min_temperature)
        self.activation = None  # the activation lives inside the MLP, not after it
        return self.wrap(core)


class ZipMode(AutoStrEnum):
    sum = auto()
    mean = auto()
    static_mix = auto()


class ZipReduceVariables(nn.Module):
    """apply head i to input i and reduce the results with fixed or learned weights"""

    def __init__(self, mode: ZipMode, modules: Iterable[nn.Module]):
        super().__init__()
        self.heads = nn.ModuleList(modules)
        self.M = len(self.heads)
        self.mode = str(mode)
        if self.mode == "sum":
            self.weights = torch.ones(self.M)
        elif self.mode == "mean":
            self.weights = torch.ones(self.M) / self.M
        elif self.mode == "static_mix":
            self.weights = nn.Parameter(-torch.rand(self.M))
        else:
            raise ValueError(f"unknown ZipMode '{mode}'")

    def forward(self, inputs: Tuple[torch.Tensor, ...]):
        w = self.weights.to(inputs[0].device)
        if w.requires_grad:
            w = torch.softmax(w, dim=0)
        y = self.heads[0](inputs[0]) * w[0]
        for i in range(1, self.M):
            y = y + self.heads[i](inputs[i]) * w[i]
        return y
