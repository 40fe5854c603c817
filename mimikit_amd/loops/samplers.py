"""Prompt position samplers of the generate loop (reference ``mimikit/loops/samplers.py:50-81`` and
``loops/generate.py:76-82``), without the h5mapper / torch DataLoader machinery around them."""
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch

__all__ = ["IndicesSampler", "PromptIndices"]


class IndicesSampler:
    """Yields the start indices of the prompts.  ``indices`` is a tuple of fixed positions, ``None`` entries are drawn
    uniformly from [min_i, max_i) and floored to a multiple of ``sampling_stride`` (the dataset's down-sampling); after each
    pass the random ones are drawn again when ``redraw``.  Anything but a tuple: N uniform draws (reference :70-81)."""

    def __init__(self, N: int = 0, indices: Union[Tuple[Optional[int], ...], Sequence] = (), min_i: int = 0,
                 max_i: Optional[int] = None, redraw: bool = True, sampling_stride: int = 1):
        self.N, self._indices, self.min_i, self.max_i = N, indices, min_i, max_i
        self.redraw, self.sampling_stride = redraw, sampling_stride
        self.indices = self.draw_indices(N, indices)

    def __iter__(self):
        for i in self.indices:
            yield i
        if self.redraw:
            self.indices = self.draw_indices(self.N, self._indices)

    def __len__(self):
        return len(self.indices)

    def draw_indices(self, N, indices):
        if isinstance(indices, tuple):
            return tuple(
                self.sampling_stride * (torch.randint(self.min_i, self.max_i, (1,)).item() // self.sampling_stride)
                if i is None else i
                for i in indices)
        return torch.randint(self.min_i, self.max_i, (N,))


class PromptIndices:
    """first element of every prompt batch: the position the prompt was cut at (reference generate.py:76-82)"""

    def __init__(self, n: int):
        self.n = n

    def __call__(self, item, file=None, **kwargs):
        return np.array([item], dtype=np.int32)
