"""MLP head with a learned-temperature extra logit (reference networks/mlp.py:12-63).
Parameter container + differentiable forward; on the generate path the layers run as
fused linear kernels and the temperature column is applied inside the sampler kernel."""
from typing import Optional

import torch
import torch.nn as nn

__all__ = ["MLP"]


class MLP(nn.Module):
    def __init__(self, in_dim: int, hidden_dim: int, out_dim: int, n_hidden_layers: int = 0,
                 activation: nn.Module = nn.Mish(), bias: bool = True, dropout: float = 0.,
                 dropout1d: float = 0., min_temperature: Optional[float] = 1e-4):
        super().__init__()
        self.learn_temperature = min_temperature is not None
        self.in_dim, self.hidden_dim = in_dim, hidden_dim
        self.out_dim = out_dim + int(self.learn_temperature)
        self.n_hidden_layers = n_hidden_layers
        self.activation = activation
        self.bias, self.dropout, self.dropout1d = bias, dropout, dropout1d

        def block(i, o):
            mods = [nn.Linear(i, o, bias=bias), self.activation]
            if dropout > 0.:
                mods.append(nn.Dropout(dropout))
            if dropout1d > 0.:
                mods.append(nn.Dropout1d(dropout1d))
            return mods

        layers = block(in_dim, hidden_dim)
        # the reference repeats ONE hidden block n times (tuple multiplication, mlp.py:46-49):
        # with n_hidden_layers > 1 the hidden layers share their parameters
        layers += block(hidden_dim, hidden_dim) * n_hidden_layers
        self.fc = nn.Sequential(*layers, nn.Linear(hidden_dim, self.out_dim, bias=bias))
        if self.learn_temperature:
            self.sigmoid = nn.Sigmoid()
            self.register_buffer("min_temp", torch.tensor(min_temperature))

    def forward(self, x: torch.Tensor):
        logits = self.fc(x)
        if self.learn_temperature:
            temp = self.sigmoid(logits[..., -1:])
            logits = logits[..., :-1] / torch.maximum(temp, self.min_temp)
        return logits
