"""Import path kept from the reference (``mimikit.networks.mlp.MLP``); the class lives in
``mimikit_amd.modules.mlp`` to keep the package import order acyclic."""
from ..modules.mlp import MLP

__all__ = ["MLP"]
