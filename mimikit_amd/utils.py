"""Small host-side helpers shared by the package.

Mirrors the helpers the hot path touches in the reference's ``mimikit/utils.py``
(``AutoStrEnum`` :19-25, ``default_device`` :27-35).
"""
import enum

__all__ = ["AutoStrEnum", "default_device"]


class AutoStrEnum(str, enum.Enum):
    """String-valued enum whose ``auto()`` members are named after themselves,
    so that config fields can be given either as members or plain strings."""

    def _generate_next_value_(name, start, count, last_values):  # noqa: N805
        return name

    def __str__(self):
        return str(self.value)


def default_device() -> str:
    """Device the generate loop moves the network to (reference utils.py:27-35).
    On a ROCm build of PyTorch the HIP device is exposed as "cuda"."""
    import torch

    if torch.cuda.is_available():
        return "cuda"
    return "cpu"
