"""Named feature source.

The reference's ``Extractor`` (features/extractor.py:16-59) is an h5mapper
``Feature`` stored in an HDF5 dataset; on the generate path only its ``name`` and
``functional`` (for time unit / element type of an io spec) matter, which is all
that is kept here.
"""
import dataclasses as dtc
from typing import Optional

from ..config import Config
from .functionals import Compose, FileToSignal, Functional, Normalize, RemoveDC

__all__ = ["Extractor"]


@dtc.dataclass
class Extractor(Config, type_field=False):
    name: str
    functional: Functional
    merge_files_labels: bool = False
    consolidate_labels: bool = False
    derived_from: Optional[str] = None

    def load(self, inputs):
        return self.functional(inputs)

    @staticmethod
    def signal(sr: int = 16000) -> "Extractor":
        return Extractor(name="signal", functional=Compose(FileToSignal(sr=sr), Normalize(), RemoveDC()))
