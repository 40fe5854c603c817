"""Chunked long-form generation (reference ``mimikit/loops/generate_chunks.py:45-56``): generate ``output_duration_sec``
at a time, each chunk prompted with the tail of the one before, so that arbitrarily long clips never hold more than one
chunk of state.  The reference script stores chunks in an HDF5 file; this returns / yields them."""
from typing import Callable, Iterator, Optional, Tuple

import numpy as np
import torch

from .generate import GenerateLoopV2

__all__ = ["generate_chunks"]


def generate_chunks(config: "GenerateLoopV2.Config", network, seed: torch.Tensor, n_chunks: int,
                    update_parameters: Optional[Callable[[dict, int], dict]] = None) -> Iterator[torch.Tensor]:
    """``seed``: (batch, prompt_steps) prompt in the network's input domain (e.g. mu-law classes).  Yields the newly
    generated part of every chunk, (batch, n_steps), in that same domain; ``update_parameters(parameters, chunk)`` may
    return new generate parameters for the next chunk (the reference walks the temperatures randomly, :47-48)."""
    if config.yield_inversed_outputs:
        raise ValueError("generate_chunks feeds every chunk back as the next prompt: set yield_inversed_outputs=False")
    sr = network.config.io_spec.sr
    n_steps = GenerateLoopV2.get_n_steps(config, network)
    n_prompt = seed.shape[1]
    prompt = seed
    for i in range(n_chunks):
        loader = [[np.ones(1), prompt]]
        loop = GenerateLoopV2(config, network, n_steps, loader, logger=None)
        out = None
        for output in loop.run():
            out = output[0]
            break
        new = out[:, n_prompt:]
        yield new
        prompt = out[:, -n_prompt:]
        if update_parameters is not None and config.parameters is not None:
            config.parameters = update_parameters(config.parameters, i)
    del sr
