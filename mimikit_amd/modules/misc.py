"""Shape helpers used inside io modules (reference modules/misc.py:19-112).
They only matter for the differentiable (training) forward and for keeping the
``state_dict`` indices of ``nn.Sequential`` containers identical to the
reference; the HIP generate path fuses them away."""
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import nn

__all__ = ["Chunk", "Flatten", "Transpose", "CausalPad", "Unsqueeze", "Unfold"]


class Transpose(nn.Module):
    def __init__(self, dim1, dim2):
        super().__init__()
        self.dims = (dim1, dim2)

    def forward(self, x):
        return None if x is None else x.transpose(*self.dims).contiguous()


class CausalPad(nn.Module):
    """pad spec given per dimension (first to last); positive = left, negative = right"""

    def __init__(self, pad: Tuple[int, ...]):
        super().__init__()
        flat = []
        for p in reversed(pad):
            flat += [p, 0] if p >= 0 else [0, -p]
        self.pad = tuple(flat)

    def forward(self, x):
        return F.pad(x, self.pad)


class Chunk(nn.Module):
    def __init__(self, chunks: int, dim: int = -1, sum_outputs: bool = False):
        super().__init__()
        self.chunks, self.dim, self.sum_outputs = chunks, dim, sum_outputs

    def forward(self, x):
        parts = torch.chunk(x, self.chunks, dim=self.dim)
        return sum(parts) if self.sum_outputs else parts


class Flatten(nn.Module):
    """merge the last |n| dims (n < 0) or the first n dims (n > 0)"""

    def __init__(self, n_dims: int):
        super().__init__()
        self.n_dims = n_dims

    def forward(self, x):
        if self.n_dims < 0:
            return x.reshape(*x.shape[:self.n_dims], -1)
        return x.reshape(-1, *x.shape[self.n_dims:])


class Unsqueeze(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return x.unsqueeze(self.dim)


class Unfold(nn.Module):
    def __init__(self, dim=-1, size=1, step=1):
        super().__init__()
        self.params = (dim, size, step)

    def forward(self, x):
        return x.unfold(*self.params)
