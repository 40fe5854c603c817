"""Output wrapper and categorical sampler (reference modules/targets.py:10-52).

In eval mode on the HIP device the sampler is the wavefront-per-row kernel of
``csrc/kernels.hip`` (argmax with first-max tie-break, or inverse-CDF sampling of
softmax(logits / T) driven by uniforms drawn with torch's device generator).
In training mode it passes logits through, as the reference does.
"""
import torch
from torch import nn

__all__ = ["OutputWrapper", "CategoricalSampler", "as_tensor"]


class OutputWrapper(nn.Module):
    def __init__(self, estimator: nn.Module, sampler: nn.Module):
        super().__init__()
        self.estimator = estimator
        self.sampler = sampler

    def forward(self, *inputs, **sampler_kwargs):
        params = self.estimator(*inputs)
        if self.training:
            return params
        return self.sampler(params, **sampler_kwargs)

    @property
    def sampling_params(self):
        return getattr(self.sampler, "sampling_params", {})


def as_tensor(temperature, like: torch.Tensor) -> torch.Tensor:
    """scalar / tuple / per-item temperature -> tensor broadcastable from the left (reference :27-34)"""
    if not isinstance(temperature, torch.Tensor):
        if isinstance(temperature, (int, float)):
            temperature = [temperature]
        temperature = torch.tensor(temperature)
    if temperature.ndim != like.ndim:
        temperature = temperature.view(*temperature.shape, *([1] * (like.ndim - temperature.ndim)))
    return temperature.to(like.device)


def per_row_temperature(temperature, batch: int, device) -> torch.Tensor:
    """one fp32 temperature per clip, as the kernels take it"""
    t = temperature if isinstance(temperature, torch.Tensor) else torch.tensor(
        [temperature] if isinstance(temperature, (int, float)) else list(temperature))
    t = t.to(device=device, dtype=torch.float32).reshape(-1)
    if t.numel() == 1:
        t = t.expand(batch)
    if t.numel() != batch:
        raise ValueError(f"temperature must be a scalar or hold one value per batch item ({batch}), got {t.numel()}")
    return t.contiguous()


class CategoricalSampler(nn.Module):
    sampling_params = {"temperature"}

    def forward(self, logits, *, temperature=None):
        if self.training:
            return logits
        from .. import native
        native.require_device(logits)
        lead = logits.shape[:-1]
        rows = logits.reshape(-1, logits.shape[-1])
        if temperature is None:
            return native.categorical_sample(rows, rows.shape[-1], False, 0., None, None).reshape(lead)
        t = as_tensor(temperature, logits).to(torch.float32).expand(*lead, 1).reshape(-1).contiguous()
        u = torch.rand(rows.shape[0], device=logits.device, dtype=torch.float32)
        out = native.categorical_sample(rows, rows.shape[-1], False, 0., t, u)
        # torch.multinomial(.., 1) of 2-D logits is (B, 1) and the reference only reshapes > 2-D logits (:64-67)
        return out.reshape(lead) if logits.dim() > 2 else (out.reshape(-1, 1) if logits.dim() == 2 else out.reshape(1))
