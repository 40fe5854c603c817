"""Deterministic weight recipe (TEST / BENCH INFRASTRUCTURE).

Fills every tensor of a ``state_dict`` from a counter-based hash of (name, element
index) -- independent of torch's RNG, of tensor creation order and of library
versions -- so that the golden-vector generator (which loads the values into the
reference's networks) and the tests / bench (which load them into this repo's
networks) see bit-identical weights without committing megabytes of parameters.
"""
import zlib
from typing import Dict, Mapping

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform_tensor(name: str, shape, scale: float, seed: int = 0) -> torch.Tensor:
    """uniform(-scale, scale) fp32 values determined by (seed, name, flat index) only"""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode()) + (seed << 32))
    ctr = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(ctr * np.uint64(0x2545F4914F6CDD1D) + key)
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)      # 24 random bits -> [0, 1)
    vals = ((u * 2.0 - 1.0) * scale).astype(np.float32)
    return torch.from_numpy(vals.reshape(tuple(shape)))


def recipe_state_dict(shapes: Mapping[str, tuple], seed: int = 0, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """weights ~ U(-g/sqrt(fan_in), g/sqrt(fan_in)), biases likewise with the same fan_in rule torch uses,
    embeddings ~ U(-1, 1).  0-dim entries (buffers such as `min_temp`) are skipped."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if len(shape) == 0:
            continue
        if len(shape) == 1:
            scale = 0.1 if name.endswith("bias") else gain / np.sqrt(max(shape[0], 1))
        else:
            fan_in = int(np.prod(shape[1:]))
            scale = gain / np.sqrt(max(fan_in, 1))
        out[name] = uniform_tensor(name, shape, float(scale), seed)
    return out


def load_recipe(module: torch.nn.Module, seed: int = 0, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """fill `module` in place from the recipe (keeps non-float / 0-dim entries); returns the values used"""
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if torch.is_floating_point(v) and v.dim() > 0}
    values = recipe_state_dict(shapes, seed, gain)
    with torch.no_grad():
        for k, v in values.items():
            sd[k].copy_(v.to(sd[k].device))
    return values
