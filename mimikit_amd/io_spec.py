"""Input / target specifications (reference mimikit/io_spec.py).

An ``IOSpec`` ties, for every network input and target, a named feature source
(``Extractor``), a transform (``Functional``), an io module config and -- for
targets -- an objective.  ``bind_to`` wires the element type of the feature into
the module (``class_size`` / ``in_dim`` / ``out_dim`` / sampler), reference
:85-92 and :136-149.  ``IOSpec.mulaw_io`` / ``IOSpec.magspec_io`` are the two
factories the BASELINE configs are built from (:220-285).
"""
import dataclasses as dtc
from enum import auto
from typing import Dict, Tuple

import torch.nn as nn
from typing_extensions import Literal

from .config import Config
from .features.extractor import Extractor
from .features.functionals import (Compose, Continuous, Discrete, FileToSignal, Functional, MagSpec, MuLawCompress,
                                   Normalize, RemoveDC)
from .features.item_spec import Frame, ItemSpec, Sample, Unit
from .modules.activations import ActivationConfig
from .modules.io import ChunkedLinearIO, EmbeddingIO, FramedLinearIO, IOModule, MLPIO
from .modules.targets import CategoricalSampler
from .utils import AutoStrEnum

__all__ = ["InputSpec", "ObjectiveType", "Objective", "TargetSpec", "IOSpec", "BatchItem"]


@dtc.dataclass
class BatchItem:
    """What a feature spec asks a dataset for: a slice of a named feature plus a transform.
    (The reference returns an ``h5mapper.Input`` with an ``AsSlice`` getter, io_spec.py:65-75.)"""
    data: str
    shift: int
    length: int
    downsampling: int
    transform: Functional


@dtc.dataclass
class _FeatureSpec(Config, type_field=False):
    extractor_name: str
    transform: Functional
    module: IOModule
    extractor: Extractor = dtc.field(init=False, repr=False, default=None, metadata=dict(omegaconf_ignore=True))

    def bind_to(self, extractor: Extractor):
        self.extractor = extractor
        return self

    def _chain(self):
        return [self.extractor.functional, self.transform]

    @property
    def units(self):
        return [f.unit for f in self._chain() if f.unit is not None]

    @property
    def unit(self) -> Unit:
        return self.units[-1]

    @property
    def elem_type(self):
        return [f.elem_type for f in self._chain() if f.elem_type is not None][-1]

    @property
    def sr(self):
        found = [f.unit.sr for f in self._chain() if isinstance(f.unit, Sample) and f.unit.sr is not None]
        return found[-1] if found else None

    @property
    def hop_length(self):
        found = [f.unit.hop_length for f in self._chain() if isinstance(f.unit, Frame)]
        return found[-1] if found else None

    def to_batch_item(self, item_spec: ItemSpec) -> BatchItem:
        spec = item_spec.to(self.extractor.functional.unit)
        return BatchItem(self.extractor.name, spec.shift, spec.length, spec.stride, self.transform)

    @property
    def inv(self) -> Functional:
        return self.transform.inv


@dtc.dataclass
class InputSpec(_FeatureSpec, type_field=False):
    def bind_to(self, extractor: Extractor):
        super().bind_to(extractor)
        kind = self.elem_type
        if isinstance(kind, Discrete):
            self.module.set(class_size=kind.size)
        elif isinstance(kind, Continuous):
            self.module.set(in_dim=kind.size)
        return self


class ObjectiveType(AutoStrEnum):
    reconstruction = auto()
    categorical_dist = auto()


@dtc.dataclass
class Objective(Config, type_field=False):
    objective_type: ObjectiveType
    params: Dict = dtc.field(default_factory=dict)
    weight: float = 1.

    def get_sampler(self):
        return CategoricalSampler() if self.objective_type == "categorical_dist" else None

    def get_criterion(self):
        """training losses are outside this package's scope; the two objectives of the BASELINE
        configs get their stock torch criterion so a trainer can still call ``loss_fn``"""
        if self.objective_type == "categorical_dist":
            ce = nn.CrossEntropyLoss(reduction="mean")
            return lambda output, target: ce(output.view(-1, output.size(-1)), target.view(-1))
        if self.objective_type == "reconstruction":
            l1 = nn.L1Loss(reduction="mean")
            return lambda output, target: l1(output, target)
        return None


@dtc.dataclass
class TargetSpec(_FeatureSpec, type_field=False):
    objective: Objective = None
    extra_loss_terms: Tuple[Objective, ...] = ()

    def bind_to(self, extractor: Extractor):
        super().bind_to(extractor)
        kind = self.objective.objective_type
        if kind == "reconstruction":
            assert isinstance(self.elem_type, Continuous)
            self.module.set(out_dim=self.elem_type.size)
        elif kind == "categorical_dist":
            assert isinstance(self.elem_type, Discrete)
            self.module.set(out_dim=self.elem_type.size, sampler=self.objective.get_sampler())
        self.criterion = self.objective.get_criterion()
        return self

    def loss_fn(self, output, target):
        value = self.criterion(output, target) * self.objective.weight
        return {"loss": value, str(self.objective.objective_type): value}


def _single(values, what):
    values = set(values)
    if len(values) > 1:
        raise RuntimeError(f"Expected to find a single {what} but found several: '{values}'")
    return values.pop()


@dtc.dataclass
class IOSpec(Config, type_field=False):
    inputs: Tuple[InputSpec, ...]
    targets: Tuple[TargetSpec, ...]

    def bind_to(self, dataset_config):
        schema = dataset_config.schema
        for f in (*self.inputs, *self.targets):
            f.bind_to(schema[f.extractor_name])
        return self

    @property
    def _all(self):
        return (*self.inputs, *self.targets)

    @property
    def sr(self):
        return _single((f.sr for f in self._all), "sample_rate")

    @property
    def hop_length(self):
        return _single((f.hop_length for f in self._all), "hop_length")

    @property
    def unit(self) -> Unit:
        return _single((f.unit for f in self._all), "time unit")

    @property
    def loss_fn(self):
        def total(output, target):
            out, acc = {}, 0.
            for spec, o, t in zip(self.targets, output, target):
                terms = spec.loss_fn(o, t)
                acc = acc + terms.pop("loss")
                out.update(terms)
            out["loss"] = acc
            return out

        return total

    # -- factories -----------------------------------------------------------
    @dtc.dataclass
    class MuLawIOConfig(Config):
        sr: int = 16000
        q_levels: int = 256
        compression: float = 1.
        input_module_type: Literal["framed_linear", "embedding"] = "framed_linear"
        mlp_dim: int = 128
        n_mlp_layers: int = 0
        min_temperature: float = 1e-4

    @staticmethod
    def mulaw_io(config: "IOSpec.MuLawIOConfig", extractor: Extractor = None) -> "IOSpec":
        c = config
        if extractor is None:
            extractor = Extractor("signal", Compose(FileToSignal(c.sr), Normalize(), RemoveDC()))
        kinds = {"framed_linear": FramedLinearIO, "embedding": EmbeddingIO}
        if c.input_module_type not in kinds:
            raise ValueError(f"Unimplemented input_module_type: '{c.input_module_type}'")
        mu_law = MuLawCompress(c.q_levels, c.compression)
        return IOSpec(
            inputs=(InputSpec(extractor.name, mu_law, kinds[c.input_module_type]()).bind_to(extractor),),
            targets=(TargetSpec(
                extractor.name, mu_law,
                MLPIO(hidden_dim=c.mlp_dim, n_hidden_layers=c.n_mlp_layers, min_temperature=c.min_temperature),
                objective=Objective("categorical_dist")).bind_to(extractor),),
        )

    @dtc.dataclass
    class MagSpecIOConfig(Config):
        sr: int = 22050
        n_fft: int = 2048
        hop_length: int = 512
        activation: str = "Abs"

    @staticmethod
    def magspec_io(config: "IOSpec.MagSpecIOConfig", extractor: Extractor = None) -> "IOSpec":
        c = config
        if extractor is None:
            extractor = Extractor("signal", Compose(FileToSignal(c.sr), Normalize(), RemoveDC()))

        def feature():
            return MagSpec(c.n_fft, c.hop_length, center=False, window="hann")

        return IOSpec(
            inputs=(InputSpec(extractor.name, feature(), ChunkedLinearIO(n_chunks=1)).bind_to(extractor),),
            targets=(TargetSpec(
                extractor.name, feature(),
                ChunkedLinearIO(n_chunks=1, activation=ActivationConfig(act=c.activation)),
                objective=Objective("reconstruction")).bind_to(extractor),),
        )
