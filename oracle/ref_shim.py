"""Import shim for the upstream reference (TEST INFRASTRUCTURE ONLY).

The reference package (``/root/reference/mimikit``) cannot be imported as-is in
the build container: its ``__init__`` star-imports UI/demo packages and it
depends on h5mapper, omegaconf, librosa, torchaudio, numba, pytorch_lightning,
IPython, pydub ... none of which are installed.

This module pre-seeds ``sys.modules`` with minimal stand-ins for those
*third-party* packages (never for reference code) and with path-only package
objects for ``mimikit`` / ``mimikit.loops`` so that the reference's own
hot-path modules (networks, io_spec, features.functionals, loops.generate) are
imported unmodified from where they lie.  It is used only by
``tests/golden/make_golden.py`` to generate golden vectors in this container;
nothing here travels to, or is needed on, the GPU box.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MMK_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "mimikit"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)

    def _fallback(attr, _name=name):
        # any other symbol a reference module imports from a missing
        # third-party package resolves to a permissive placeholder class
        if attr.startswith("__"):
            raise AttributeError(attr)
        return type(attr, (_Anything,), {})

    m.__getattr__ = _fallback
    sys.modules[name] = m
    return m


def _map_nested(batch, test, func):
    """what h5mapper.process_batch is used for at the reference's call sites
    (loops/generate.py:39,197): apply func to every leaf passing test."""
    if test(batch):
        return func(batch)
    if isinstance(batch, (tuple, list)):
        return type(batch)(_map_nested(b, test, func) for b in batch)
    if isinstance(batch, dict):
        return {k: _map_nested(v, test, func) for k, v in batch.items()}
    return batch


class _Anything:
    """permissive placeholder base class / callable"""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Anything()

    def __class_getitem__(cls, item):
        return cls


class _Subscriptable:
    def __getitem__(self, item):
        return self

    def __call__(self, *a, **k):
        return a[0] if a else None


def _install_third_party_stubs():
    # h5mapper ------------------------------------------------------------
    class Input:
        def __init__(self, data=None, getter=None, transform=None, **kw):
            self.data, self.getter, self.transform = data, getter, transform

    class Getter:
        def __init__(self, *a, **k):
            self.n = None

    class AsSlice(Getter):
        def __init__(self, dim=0, shift=0, length=1, downsampling=1, **kw):
            self.dim, self.shift, self.length, self.downsampling = dim, shift, length, downsampling

    class Feature:
        pass

    _mod("h5mapper", process_batch=_map_nested, Input=Input, Getter=Getter, AsSlice=AsSlice,
         Feature=Feature, TypedFile=_Anything, TensorDict=_Anything, FileWalker=_Anything,
         Array=_Anything, Sound=_Anything)

    # omegaconf -----------------------------------------------------------
    _mod("omegaconf", OmegaConf=_Anything, ListConfig=list, DictConfig=dict)

    # audio libs ----------------------------------------------------------
    _mod("librosa", util=_Anything(), sequence=_Anything(), segment=_Anything(), feature=_Anything())
    _mod("librosa.util")
    ta = _mod("torchaudio")
    ta.functional = _mod("torchaudio.functional")
    ta.transforms = _mod("torchaudio.transforms")
    _mod("pydub", AudioSegment=_Anything)
    _mod("soundfile")

    # numba ---------------------------------------------------------------
    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    _mod("numba", njit=njit, prange=range, float32=_Subscriptable(), intp=_Subscriptable(),
         int64=_Subscriptable(), float64=_Subscriptable())

    # lightning -----------------------------------------------------------
    pl = _mod("pytorch_lightning", LightningModule=_Anything, Trainer=_Anything, Callback=_Anything)
    pl.callbacks = _mod("pytorch_lightning.callbacks", Callback=_Anything, ModelCheckpoint=_Anything,
                        TQDMProgressBar=_Anything, ProgressBar=_Anything)
    _mod("pytorch_lightning.callbacks.progress", TQDMProgressBar=_Anything, ProgressBar=_Anything)
    _mod("pytorch_lightning.callbacks.progress.tqdm_progress", TQDMProgressBar=_Anything, Tqdm=_Anything)
    pl.trainer = _mod("pytorch_lightning.trainer")
    _mod("pytorch_lightning.trainer.states", TrainerStatus=_Anything, TrainerFn=_Anything)
    pl.utilities = _mod("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    pl.loggers = _mod("pytorch_lightning.loggers", Logger=_Anything)
    _mod("pytorch_lightning.loggers.logger", Logger=_Anything, rank_zero_experiment=lambda f: f)
    lf = _mod("lightning_fabric")
    lf.loggers = _mod("lightning_fabric.loggers")
    _mod("lightning_fabric.loggers.logger", Logger=_Anything, rank_zero_experiment=lambda f: f)

    # notebook / plotting -------------------------------------------------
    ip = _mod("IPython", get_ipython=lambda: None)
    ip.display = _mod("IPython.display", display=lambda *a, **k: None, Audio=_Anything, HTML=_Anything)
    if "matplotlib" not in sys.modules:
        try:
            importlib.import_module("matplotlib")
        except Exception:
            mpl = _mod("matplotlib")
            mpl.pyplot = _mod("matplotlib.pyplot")
    if "tqdm" not in sys.modules:
        try:
            importlib.import_module("tqdm")
        except Exception:
            _mod("tqdm", tqdm=lambda x, **k: x)


_LOADED = {}


def load_reference():
    """Returns a namespace of the reference's hot-path modules, imported from
    REFERENCE_ROOT without executing the star-importing package __init__s."""
    if _LOADED:
        return _LOADED["ns"]
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    _install_third_party_stubs()
    pkg_dir = os.path.join(REFERENCE_ROOT, "mimikit")
    # path-only package objects: skip `from .x import *` of mimikit/__init__.py
    root = types.ModuleType("mimikit")
    root.__path__ = [pkg_dir]
    sys.modules["mimikit"] = root
    loops = types.ModuleType("mimikit.loops")
    loops.__path__ = [os.path.join(pkg_dir, "loops")]
    sys.modules["mimikit.loops"] = loops
    root.loops = loops

    imp = importlib.import_module
    # order mirrors the real import order (circular imports otherwise)
    arm = imp("mimikit.networks.arm")
    io_spec = imp("mimikit.io_spec")
    functionals = imp("mimikit.features.functionals")
    item_spec = imp("mimikit.features.item_spec")
    extractor = imp("mimikit.features.extractor")
    wavenet = imp("mimikit.networks.wavenet_v2")
    srnn = imp("mimikit.networks.sample_rnn_v2")
    s2s = imp("mimikit.networks.s2s_lstm_v2")
    mio = imp("mimikit.modules.io")
    targets = imp("mimikit.modules.targets")
    generate = imp("mimikit.loops.generate")
    generate.default_device = lambda: "cpu"
    generate.generate_tqdm = lambda rng: rng
    ns = types.SimpleNamespace(
        arm=arm, io_spec=io_spec, functionals=functionals, item_spec=item_spec, extractor=extractor,
        wavenet=wavenet, srnn=srnn, s2s=s2s, io=mio, targets=targets, generate=generate,
        IOSpec=io_spec.IOSpec, WaveNet=wavenet.WaveNet, SampleRNN=srnn.SampleRNN,
        Seq2SeqLSTMNetwork=s2s.Seq2SeqLSTMNetwork, GenerateLoopV2=generate.GenerateLoopV2,
    )
    _LOADED["ns"] = ns
    return ns
