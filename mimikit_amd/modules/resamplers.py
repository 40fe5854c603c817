"""Time/feature resamplers (reference modules/resamplers.py:13-46): parameter
containers with the reference's names plus a differentiable forward."""
from functools import partial

import numpy as np
import torch.nn as nn

__all__ = ["LinearResampler", "Conv1dResampler"]


class LinearResampler(nn.Module):
    """(B, T, D) -> (B, T*t_factor, D*d_factor) through one Linear (SampleRNN tier up-sampling,
    Seq2Seq decoder)."""

    def __init__(self, in_d, t_factor, d_factor, **kwargs):
        super().__init__()
        self.fc = nn.Linear(in_d, int(in_d * t_factor * d_factor), **kwargs)
        self.tf, self.df = t_factor, d_factor

    def forward(self, x):
        b, t, d = x.size()
        return self.fc(x).reshape(b, int(t * self.tf), int(d * self.df))


class Conv1dResampler(nn.Module):
    """frames of 1/t_factor steps -> one vector each via a strided-by-view Conv1d (bottom SampleRNN tier).
    The view/transposition sequence of the reference (:40-46) is kept verbatim in behaviour,
    including how it lays channels out when more than one frame is passed."""

    def __init__(self, in_dim, t_factor, d_factor, **kwargs):
        super().__init__()
        make = nn.Conv1d if t_factor <= 1 else partial(nn.ConvTranspose1d, stride=t_factor)
        self.kernel_size = int(t_factor) if t_factor >= 1 else int(1 / t_factor)
        self.out_dim = int(in_dim * d_factor)
        self.cv = make(in_dim, self.out_dim, self.kernel_size, **kwargs)
        self.tf, self.df = t_factor, d_factor

    def forward(self, x):
        if x.dim() > 3:
            x = x.view(x.size(0), int(np.prod(x.shape[1:-1])), x.size(-1))
        b, t, d = x.size()
        if self.tf <= 1:
            x = x.view(-1, self.kernel_size, d).transpose(1, 2)
            x = self.cv(x).squeeze(-1).reshape(b, self.out_dim, -1)
        else:
            x = self.cv(x.transpose(1, 2))
        return x.transpose(1, 2)
