"""Checkpoint interchange (reference ``mimikit/checkpoint.py:51-173``).

A reference checkpoint is an HDF5 file (h5mapper ``TypedFile``): group ``network`` holds the ``state_dict`` tensors and
the attribute ``config`` = the network config as YAML; file attributes ``dataset`` / ``training`` hold the other configs.
Neither h5py nor h5mapper / omegaconf exist in this image, so the same CONTENT travels in a flat ``.npz`` here: one array
per ``state_dict`` key plus three string entries with the reference's YAML layout (``Config.serialize``: `type`-tagged
mappings, untagged ones typed by their key).  ``scripts/convert_reference_ckpt.py`` turns a reference ``.ckpt`` into this
file wherever h5py is installed (it cannot be run, hence not tested, in this container).

``Checkpoint`` keeps the reference's surface: ``id`` / ``epoch`` / ``root_dir``, ``os_path``, ``create``, ``network``
(config -> ``io_spec.bind_to(dataset_config)`` -> ``from_config`` -> ``load_state_dict(strict=True)``), ``network_config``,
``dataset_config``, ``from_path``.
"""
import dataclasses as dtc
import os
from functools import cached_property
from typing import Optional, Tuple

import numpy as np
import torch

from .config import Config
from .features.extractor import Extractor

__all__ = ["DatasetConfig", "Checkpoint", "save_network", "load_network"]

_CFG, _DS, _TR = "__network_config__", "__dataset_config__", "__training_config__"


@dtc.dataclass
class DatasetConfig(Config, type_field=False):
    """reference features/dataset.py:15-26 (the h5mapper file handling is dataset preparation, out of scope)"""
    sources: Tuple[str, ...] = tuple()
    filename: str = "dataset.h5"
    extractors: Tuple[Extractor, ...] = tuple()

    @property
    def schema(self):
        return {e.name: e for e in self.extractors}


def _owner_class(config):
    """the network class a ``<Network>.Config`` belongs to (reference config.py:74-79)"""
    import sys
    mod = sys.modules[type(config).__module__]
    obj = mod
    for part in type(config).__qualname__.split(".")[:-1]:
        obj = getattr(obj, part)
    return obj


def save_network(path: str, network, training_config=None) -> str:
    """state_dict + configs -> one .npz (see the module docstring)"""
    sd = {k: v.detach().cpu().numpy() for k, v in network.state_dict().items()}
    extra = {_CFG: np.asarray(network.config.serialize())}
    if training_config is not None and hasattr(training_config, "dataset") and hasattr(training_config, "training"):
        # reference :79-81: a TrainingConfig carries the dataset's config and the loop's config; they are stored side by side
        extra[_DS] = np.asarray(training_config.dataset.serialize())
        extra[_TR] = np.asarray(training_config.training.serialize())
    else:
        # reference :83-87: without one, a schema-only dataset config - enough to load the network later
        features = [*network.config.io_spec.inputs, *network.config.io_spec.targets]
        schema = {f.extractor_name: f.extractor for f in features}
        extra[_DS] = np.asarray(DatasetConfig(filename="unknown", sources=(), extractors=tuple(schema.values())).serialize())
        if training_config is not None:               # a bare loop config (no dataset of its own)
            extra[_TR] = np.asarray(training_config.serialize())
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    with open(path, "wb") as f:                                        # (np.savez would append ".npz" to a ".ckpt" name)
        np.savez(f, **sd, **extra)
    return path


def load_network(path: str, device=None):
    with np.load(path, allow_pickle=False) as z:
        cfg = Config.deserialize(str(z[_CFG]))
        ds = Config.deserialize(str(z[_DS]), as_type=DatasetConfig)
        state = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if not k.startswith("__")}
    cfg.io_spec.bind_to(ds)
    net = _owner_class(cfg).from_config(cfg)
    net.load_state_dict(state, strict=True)
    return net.to(device) if device is not None else net


@dtc.dataclass
class Checkpoint:
    id: str
    epoch: int
    root_dir: str = "./"

    def create(self, network, training_config=None, optimizer=None, trainer_state=None):
        save_network(self.os_path, network, training_config)
        if optimizer is not None:
            torch.save(optimizer.state_dict(), os.path.splitext(self.os_path)[0] + ".opt")
        return self

    @staticmethod
    def get_id_and_epoch(path):
        id_, epoch = path.split("/")[-2:]
        return id_.strip("/"), int(epoch.split(".ckpt")[0].split("=")[-1])

    @staticmethod
    def from_path(path):
        return Checkpoint(*Checkpoint.get_id_and_epoch(path), root_dir=os.path.dirname(os.path.dirname(path)))

    @property
    def os_path(self):
        return os.path.join(self.root_dir, f"{self.id}/epoch={self.epoch}.ckpt")

    def delete(self):
        os.remove(self.os_path)

    def _yaml(self, key) -> Optional[str]:
        with np.load(self.os_path, allow_pickle=False) as z:
            return str(z[key]) if key in z.files else None

    @cached_property
    def dataset_config(self) -> DatasetConfig:
        return Config.deserialize(self._yaml(_DS), as_type=DatasetConfig)

    @cached_property
    def network_config(self):
        return Config.deserialize(self._yaml(_CFG))

    @cached_property
    def training_config(self):
        raw = self._yaml(_TR)
        return Config.deserialize(raw) if raw is not None else None

    @cached_property
    def network(self):
        return load_network(self.os_path)
