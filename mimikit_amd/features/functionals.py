"""Feature functionals on the generate path.

Protocol and names follow the reference's ``mimikit/features/functionals.py``
(``Functional`` :81-111): a functional is a config dataclass that maps arrays to
arrays, dispatching on the input type, and knows its inverse, its time ``unit``
and its ``elem_type``.  Only the functionals the generate path touches exist
here:

* ``MuLawCompress`` / ``MuLawExpand`` (:313-373) and ``STFT`` / ``MagSpec``
  (:450-528, :576-606): the torch path runs as HIP kernels on the MI355X
  (``mimikit_amd/csrc/features.hip``); a tensor that is not on the HIP device is
  an error, there is no eager fallback.
* ``FileToSignal``, ``Normalize``, ``RemoveDC``, ``Compose``, ``Identity``: unit /
  elem_type carriers needed to build an ``IOSpec`` (dataset extraction itself is
  out of scope).
* ``ISTFT`` / ``GLA`` (:531-573, :609-646), the ``inv`` of STFT / MagSpec at the loop's tail, run on the HIP
  kernels of ``csrc/istft.hip`` / ``spectral2048.hip``; GLA's parity is unpinned (torchaudio is not installed in the
  build container).
* float64 tensors are computed in fp32 on the device (the kernels are fp32; the reference computes in the input's
  dtype, so float64 mu-law codes may differ from it for inputs within one fp32 ulp of a bin edge).
"""
import abc
import dataclasses as dtc
import functools
from typing import Optional, Tuple, Union

import numpy as np
import torch

from ..config import Config
from .item_spec import Frame, Sample, Unit, convert

__all__ = [
    "Continuous", "Discrete", "Functional", "Identity", "Compose", "FileToSignal", "RemoveDC", "Normalize",
    "MuLawCompress", "MuLawExpand", "STFT", "ISTFT", "MagSpec", "GLA", "Resample",
]

N_FFT = 2048
HOP_LENGTH = 512
SR = 22050
Q_LEVELS = 256


@dtc.dataclass
class Continuous:
    min_value: Union[float, int]
    max_value: Union[float, int]
    size: int


@dtc.dataclass
class Discrete:
    size: int


EventType = Union[Continuous, Discrete]


@dtc.dataclass
class Functional(Config, abc.ABC):
    """array -> array map with numpy and torch twins"""

    @property
    def unit(self) -> Optional[Unit]:
        return None

    @property
    def elem_type(self) -> Optional[EventType]:
        return None

    @abc.abstractmethod
    def np_func(self, inputs):
        ...

    @abc.abstractmethod
    def torch_func(self, inputs):
        ...

    def __call__(self, inputs):
        if isinstance(inputs, np.ndarray):
            return self.np_func(inputs)
        if isinstance(inputs, torch.Tensor):
            return self.torch_func(inputs)
        raise KeyError(type(inputs))

    @property
    @abc.abstractmethod
    def inv(self) -> "Functional":
        ...


@dtc.dataclass
class Identity(Functional):
    def np_func(self, inputs):
        return inputs

    def torch_func(self, inputs):
        return inputs

    @property
    def inv(self) -> Functional:
        return Identity()


@dtc.dataclass
class Compose(Functional):
    functionals: Tuple[Functional, ...] = ()

    def __init__(self, *funcs: Functional, functionals=()):
        self.functionals = tuple(funcs) or tuple(functionals)

    def _last(self, attr):
        found = [getattr(f, attr) for f in self.functionals if getattr(f, attr) is not None]
        return found[-1] if found else None

    @property
    def unit(self) -> Optional[Unit]:
        return self._last("unit")

    @property
    def elem_type(self) -> Optional[EventType]:
        return self._last("elem_type")

    def np_func(self, inputs):
        return self(inputs)

    def torch_func(self, inputs):
        return self(inputs)

    def __call__(self, inputs):
        for f in self.functionals:
            inputs = f(inputs)
        return inputs

    @property
    def inv(self) -> Functional:
        return Compose(*(f.inv for f in reversed(self.functionals)))


@dtc.dataclass
class FileToSignal(Functional):
    """Head of an extraction chain: carries the sample rate (reference :150-176).
    Decoding audio files (librosa) belongs to dataset preparation, not to this path."""
    sr: int = SR
    offset: float = 0.
    duration: Optional[float] = None

    @property
    def unit(self) -> Optional[Unit]:
        return Sample(self.sr)

    @property
    def elem_type(self) -> Optional[EventType]:
        return Continuous(-float("inf"), float("inf"), 1)

    def np_func(self, path):
        raise NotImplementedError("decoding audio files is dataset preparation and outside this package's scope")

    def torch_func(self, path):
        return self.np_func(path)

    def __call__(self, path):
        return self.np_func(path)

    @property
    def inv(self) -> Functional:
        return Identity()


@dtc.dataclass
class RemoveDC(Functional):
    """DC-blocking one-pole filter of the extraction chain (reference :211-229); carried for its
    (absent) unit / elem_type only."""

    def np_func(self, inputs):
        from scipy.signal import lfilter
        return lfilter([1.0, -1.0], [1.0, -0.99], inputs, axis=-1).astype(inputs.dtype)

    def torch_func(self, inputs):
        raise NotImplementedError("RemoveDC is a dataset-extraction step (numpy path only)")

    @property
    def inv(self) -> Functional:
        return Identity()


@functools.lru_cache(maxsize=16)
def resample_filter_bank(orig_sr: int, target_sr: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """(orig, new, width, table (new, 2 width + orig) fp32): the Hann-windowed sinc filter bank of
    torchaudio.functional.resample (2.0.1: ``_get_sinc_resample_kernel``, sinc_interp_hann, computed in float64), built with
    torch ops on the host - a filter table like the FFT's twiddles, not a data path."""
    import math
    g = math.gcd(int(orig_sr), int(target_sr))
    orig, new = int(orig_sr) // g, int(target_sr) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = kernels * window * scale
    return orig, new, width, kernels.reshape(new, 2 * width + orig).to(torch.float32).contiguous()


_RESAMPLE_TABLES = {}


@dtc.dataclass
class Resample(Functional):
    """reference :292-310.  The torch path is ``torchaudio.functional.resample`` there; here its polyphase windowed-sinc
    filter runs as a HIP kernel (``csrc/features.hip``).  Parity is UNPINNED (torchaudio is not installed in the build
    container): the oracle restates torchaudio 2.0.1's published algorithm, see DESIGN.md section 4."""
    orig_sr: int = SR
    target_sr: int = 16000

    @property
    def unit(self) -> Optional[Unit]:
        return Sample(self.target_sr)

    def np_func(self, inputs):
        raise NotImplementedError("the numpy path of Resample is librosa's soxr resampler (dataset preparation, out of scope)")

    def torch_func(self, inputs):
        from .. import native
        native.require_device(inputs)
        if int(self.orig_sr) == int(self.target_sr):
            return inputs
        orig, new, width, table = resample_filter_bank(int(self.orig_sr), int(self.target_sr))
        key = (int(self.orig_sr), int(self.target_sr), str(inputs.device))
        if key not in _RESAMPLE_TABLES:
            _RESAMPLE_TABLES[key] = table.to(inputs.device)
        return native.resample(inputs, _RESAMPLE_TABLES[key], orig, new, width)

    @property
    def inv(self) -> Functional:
        return Resample(self.target_sr, self.orig_sr)


@dtc.dataclass
class Normalize(Functional):
    p: float = float("inf")
    dim: int = -1

    @property
    def elem_type(self) -> Optional[EventType]:
        return Continuous(-1., 1., 1)

    def np_func(self, inputs):
        norm = np.linalg.norm(inputs, ord=self.p, axis=self.dim, keepdims=True)
        return (inputs / np.maximum(norm, np.finfo(inputs.dtype).tiny)).astype(inputs.dtype)

    def torch_func(self, inputs):
        raise NotImplementedError("Normalize is a dataset-extraction step (numpy path only)")

    @property
    def inv(self) -> Functional:
        return Identity()


# ---------------------------------------------------------------------------
# mu-law
# ---------------------------------------------------------------------------
def _mulaw_formula_cpu(x: torch.Tensor, q_levels: int, compression: float) -> torch.Tensor:
    """The reference quantisation formula (:330-338), fp32 torch ops on the host.
    Used ONLY to derive the decision-threshold table the kernel quantises with."""
    mu = torch.tensor(q_levels - 1.0, dtype=torch.float32)
    c = torch.tensor(compression, dtype=torch.float32)
    y = torch.sign(x) * torch.log1p(mu * torch.abs(x) * c) / torch.log1p(mu * c)
    return ((y + 1) / 2 * mu + 0.5).to(torch.int64)


def _ordered_key(x: torch.Tensor) -> torch.Tensor:
    """monotone int64 key of fp32 values"""
    bits = x.view(torch.int32).to(torch.int64)
    return torch.where(bits >= 0, bits, -(bits & 0x7FFFFFFF))


def _from_key(k: torch.Tensor) -> torch.Tensor:
    bits = torch.where(k >= 0, k, (-k) | 0x80000000)
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits)
    return bits.to(torch.int32).view(torch.float32)


@functools.lru_cache(maxsize=16)
def mulaw_edges(q_levels: int, compression: float) -> torch.Tensor:
    """edges[c-1] = smallest fp32 x in [-1, 1] whose reference code is >= c (c = 1..q-1).
    Found by bisection over the fp32 number line against the formula above, so the
    kernel's table quantiser reproduces the reference's own rounding at every bin edge."""
    n = q_levels - 1
    targets = torch.arange(1, q_levels, dtype=torch.int64)
    lo = _ordered_key(torch.full((n,), -1.0))   # code(lo) < c
    hi = _ordered_key(torch.full((n,), 1.0))    # code(hi) >= c
    pad = (-n) % 16 + 16                        # keep the evaluation in torch's vectorised loop
    for _ in range(34):
        mid = (lo + hi) // 2
        x = _from_key(torch.cat([mid, mid[-1:].expand(pad)]))
        code = _mulaw_formula_cpu(x, q_levels, compression)[:n]
        ge = code >= targets
        hi = torch.where(ge, mid, hi)
        lo = torch.where(ge, lo, mid)
    return _from_key(hi).contiguous()


@functools.lru_cache(maxsize=16)
def mulaw_table(q_levels: int, compression: float) -> torch.Tensor:
    """expanded value of every in-range code (reference :361-369), fp32 torch ops on the host"""
    codes = torch.arange(q_levels, dtype=torch.float32)
    mu = torch.tensor(q_levels - 1.0, dtype=torch.float32)
    c = torch.tensor(compression, dtype=torch.float32)
    x = (codes / mu) * 2 - 1.0
    return (torch.sign(x) * (torch.exp(torch.abs(x) * torch.log1p(mu * c)) - 1.0) / (mu * c)).contiguous()


_DEVICE_TABLES = {}


def _device_table(kind: str, q_levels: int, compression: float, device) -> torch.Tensor:
    key = (kind, q_levels, float(compression), str(device))
    if key not in _DEVICE_TABLES:
        host = mulaw_edges(q_levels, float(compression)) if kind == "edges" else mulaw_table(q_levels, float(compression))
        _DEVICE_TABLES[key] = host.to(device)
    return _DEVICE_TABLES[key]


@dtc.dataclass
class MuLawCompress(Functional):
    q_levels: int = Q_LEVELS
    compression: float = 1.

    @property
    def elem_type(self) -> Optional[EventType]:
        return Discrete(self.q_levels)

    def np_func(self, inputs):
        mu = self.q_levels - 1.0
        y = np.sign(inputs) * np.log1p(mu * np.abs(inputs) * self.compression) / np.log1p(mu * self.compression)
        return ((y + 1) / 2 * mu + 0.5).astype(np.int64)

    def torch_func(self, inputs):
        from .. import native
        native.require_device(inputs)
        edges = _device_table("edges", self.q_levels, self.compression, inputs.device)
        return native.mulaw_compress(inputs, self.q_levels, float(self.compression), edges)

    @property
    def inv(self) -> Functional:
        return MuLawExpand(self.q_levels, self.compression)


@dtc.dataclass
class MuLawExpand(Functional):
    q_levels: int = Q_LEVELS
    compression: float = 1.

    @property
    def elem_type(self) -> Optional[EventType]:
        return Continuous(-1., 1., 1)

    def np_func(self, inputs):
        mu = self.q_levels - 1.0
        x = (inputs / mu) * 2 - 1.0
        return np.sign(x) * (np.exp(np.abs(x) * np.log1p(mu * self.compression)) - 1.0) / (mu * self.compression)

    def torch_func(self, inputs):
        from .. import native
        native.require_device(inputs)
        if inputs.is_floating_point():
            raise TypeError("MuLawExpand on the HIP path takes integer class indices")
        table = _device_table("table", self.q_levels, self.compression, inputs.device)
        return native.mulaw_expand(inputs, self.q_levels, float(self.compression), table)

    @property
    def inv(self) -> Functional:
        return MuLawCompress(self.q_levels, self.compression)


# ---------------------------------------------------------------------------
# STFT family
# ---------------------------------------------------------------------------
@dtc.dataclass
class STFT(Functional):
    n_fft: int = N_FFT
    hop_length: int = HOP_LENGTH
    coordinate: str = "pol"
    center: bool = True
    window: Optional[str] = "hann"
    pad_mode: str = "constant"
    alignment: Optional[str] = "end"

    @property
    def unit(self) -> Optional[Unit]:
        return Frame(self.n_fft, self.hop_length, padding=self.center)

    @property
    def elem_type(self) -> Optional[EventType]:
        return Continuous(0., float("inf"), 1 + self.n_fft // 2)

    def fixed_length(self, n_samples: int) -> int:
        """number of samples kept by the reference's ``_fix_length`` (:468-486)"""
        n_frames = convert(n_samples, Sample(1), self.unit, as_length=True) + int(self.center)
        return convert(n_frames, self.unit, Sample(1), as_length=True)

    def _fix_length(self, inputs):
        if self.alignment is None:
            return inputs
        keep = self.fixed_length(inputs.shape[-1])
        if self.alignment == "end":
            return inputs[..., -keep:]
        if self.alignment == "start":
            return inputs[..., :keep]
        return inputs

    def np_func(self, inputs):
        raise NotImplementedError("the numpy (librosa) STFT belongs to dataset extraction; use the torch path on the HIP device")

    def torch_func(self, inputs):
        from .. import native
        native.require_device(inputs)
        # the reference ignores self.window on the torch path and always applies a periodic Hann (:513)
        inputs = self._fix_length(inputs)
        if self.coordinate == "mag" and self.pad_mode == "constant":
            return native.stft_mag(inputs, self.n_fft, self.hop_length, bool(self.center))
        if self.coordinate in native.STFT_COORDINATES:
            return native.stft(inputs, self.n_fft, self.hop_length, bool(self.center), self.pad_mode, self.coordinate)
        if self.coordinate == "mag":
            return native.stft(inputs, self.n_fft, self.hop_length, bool(self.center), self.pad_mode, "pol")[..., 0]
        raise ValueError(f"unknown STFT coordinate '{self.coordinate}'")

    @property
    def inv(self) -> Functional:
        return ISTFT(self.n_fft, self.hop_length, self.coordinate, self.center, self.window)


@dtc.dataclass
class ISTFT(Functional):
    n_fft: int = N_FFT
    hop_length: int = HOP_LENGTH
    coordinate: str = "pol"
    center: bool = True
    window: Optional[str] = None
    pad_mode: str = "constant"

    @property
    def unit(self) -> Optional[Unit]:
        return Sample(None)

    @property
    def elem_type(self) -> Optional[EventType]:
        return Continuous(-1., 1., 1)

    def np_func(self, inputs):
        raise NotImplementedError("the numpy (librosa) ISTFT belongs to dataset extraction; use the torch path on the HIP device")

    def torch_func(self, inputs):
        """reference :553-564: torch.istft with torch's defaults -- ``self.center`` and ``self.window`` are ignored
        there (centre trimming always on, periodic Hann always applied), and so they are here."""
        from .. import native
        native.require_device(inputs)
        if self.coordinate == "pol":
            return native.istft(inputs, self.n_fft, self.hop_length, polar=True)
        if self.coordinate == "car":
            # the reference forms ``inputs[..., 0] * (1j * inputs[..., 1])`` (:558): a purely imaginary spectrum
            # whose imaginary part is the PRODUCT of the two planes.  Kept as is.
            prod = inputs[..., 0] * inputs[..., 1]
            return native.istft(torch.stack((torch.zeros_like(prod), prod), dim=-1), self.n_fft, self.hop_length, polar=False)
        raise RuntimeError(f"ISTFT: coordinate '{self.coordinate}' leaves a real tensor, torch.istft (and this path) "
                           "needs a complex spectrum ('pol' or 'car')")

    @property
    def inv(self) -> Functional:
        return STFT(self.n_fft, self.hop_length, self.coordinate, self.center, self.window, self.pad_mode)


@dtc.dataclass
class MagSpec(Functional):
    n_fft: int = N_FFT
    hop_length: int = HOP_LENGTH
    center: bool = True
    window: Optional[str] = "hann"
    pad_mode: str = "constant"
    alignment: Optional[str] = "end"

    @property
    def stft(self) -> STFT:
        return STFT(self.n_fft, self.hop_length, "mag", self.center, self.window, self.pad_mode, alignment=self.alignment)

    @property
    def unit(self) -> Optional[Unit]:
        return Frame(self.n_fft, self.hop_length, padding=self.center)

    @property
    def elem_type(self) -> Optional[EventType]:
        return Continuous(0., float("inf"), 1 + self.n_fft // 2)

    def np_func(self, inputs):
        return self.stft.np_func(inputs)

    def torch_func(self, inputs):
        return self.stft.torch_func(inputs)

    @property
    def inv(self) -> Functional:
        return GLA(self.n_fft, self.hop_length, self.center, self.window, self.pad_mode)


@dtc.dataclass
class GLA(Functional):
    n_fft: int = N_FFT
    hop_length: int = HOP_LENGTH
    center: bool = True
    window: Optional[str] = None
    pad_mode: str = "constant"
    n_iter: int = 32

    @property
    def unit(self) -> Optional[Unit]:
        return Sample(None)

    @property
    def elem_type(self) -> Optional[EventType]:
        return Continuous(-1., 1., 1)

    def np_func(self, inputs):
        raise NotImplementedError("the numpy (librosa) Griffin-Lim belongs to dataset extraction; use the torch path on the HIP device")

    # torchaudio.transforms.GriffinLim defaults; the reference does not forward ``self.n_iter`` on the torch path (:637)
    TORCH_N_ITER = 32
    TORCH_MOMENTUM = 0.99

    def torch_func(self, inputs, init=None):
        """reference :634-642: torchaudio's GriffinLim(n_fft, hop_length, power=1.) over (time x freq) magnitudes with
        its defaults: 32 iterations, momentum 0.99, random initial phases (``torch.rand`` of a complex dtype).  ``init``
        lets a caller (the parity tests) supply those initial estimates instead of drawing them."""
        from .. import native
        native.require_device(inputs)
        if init is None:
            init = torch.rand(inputs.shape, dtype=torch.complex64, device=inputs.device)
        return native.griffin_lim(inputs, self.n_fft, self.hop_length, self.TORCH_N_ITER, self.TORCH_MOMENTUM, init)

    @property
    def inv(self) -> Functional:
        return MagSpec(self.n_fft, self.hop_length, self.center, self.window, self.pad_mode)
