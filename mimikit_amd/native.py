"""ctypes binding of ``libmmk_hip.so`` (the C ABI declared in ``include/mmk.h``).

The library is the only compute path of this package: there is no CPU or
eager-PyTorch fallback.  Every wrapper takes torch tensors that already live on
a HIP device, passes raw ``data_ptr()``s and enqueues on torch's current HIP
stream.  A missing library, or a tensor that is not on a HIP device, raises.
"""
import ctypes as C
import os
from typing import Dict, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMK_DIAG_LIB (read once, at import): "1" selects the diagnostic build (`python -m mimikit_amd.build --diag`): the same library with
# in-kernel phase stamps and the timing experiments compiled in; the path of a `.so` selects that file (the A/B variants of
# scripts/build_variant.sh: loaded from where they lie, the product library is never overwritten).  Unset or empty: the product library,
# which contains neither and must carry the digest of the tree's sources
_DIAG = os.environ.get("MMK_DIAG_LIB", "")
DIAGNOSTIC = _DIAG != ""
LIB_PATH = (os.path.join(_HERE, "libmmk_hip_diag.so") if _DIAG == "1" else os.path.abspath(_DIAG) if _DIAG.endswith(".so")
            else os.path.join(_HERE, "libmmk_hip.so"))
if DIAGNOSTIC and _DIAG != "1" and not _DIAG.endswith(".so"):
    raise ImportError(f"MMK_DIAG_LIB={_DIAG!r}: expected 1 (the diagnostic build) or the path of a library variant (*.so)")

MAX_LAYERS, MAX_COND, MAX_TIERS, MAX_STREAMS = 128, 4, 8, 4
ABI_VERSION = 4          # include/mmk.h: MMK_ABI_VERSION (bumped whenever a config struct or a signature changes)
ACT = {"none": 0, None: 0, "Identity": 0, "Tanh": 1, "Sigmoid": 2, "Mish": 3, "Abs": 4, "ReLU": 5, "Softplus": 6, "Sin": 7, "Cos": 8}      # include/mmk.h: MMK_ACT_*

i32, i64, f32, vp, cp = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_char_p


def only_mlp(estimator) -> bool:
    """an MLPIO's estimator: Sequential(MLP) - or Sequential(MLP, Dropout[, Dropout1d]): IOModule.wrap hangs the spec's dropout modules behind the core as
    well (modules/io.py:94-97); identities in eval mode, which mlp_head_problem insists on"""
    return (isinstance(estimator, torch.nn.Sequential) and len(estimator) >= 1 and type(estimator[0]).__name__ == "MLP"
            and all(isinstance(m, (torch.nn.Dropout, torch.nn.Dropout1d)) for m in list(estimator)[1:]))


def mlp_head_problem(mlp, training: bool):
    """what keeps an MLP head (networks/mlp.py:20-63) off the HIP path, or None: its activation must be one the kernels evaluate (ACT), its Linears need their
    bias, and Dropout / Dropout1d modules are identities only in eval mode (where the generate loop runs a network, loops/generate.py)"""
    if type(mlp.activation).__name__ not in ACT:
        return f"MLP head activation {type(mlp.activation).__name__}"
    if not mlp.bias:
        return "MLP head without bias"
    if (mlp.dropout or mlp.dropout1d) and training:
        return "MLP head with dropout in training mode"
    return None


def mlp_act(mlp) -> int:
    return ACT[type(mlp.activation).__name__]


def mlp_linear_keys(sd: dict, prefix: str, mlp) -> dict:
    """``fc`` of an MLP is Sequential(Linear, act, *dropouts, [Linear, act, *dropouts] * n, Linear) (networks/mlp.py:42-53): with dropout modules in it the
    Linears sit at indices i (2 + n_dropouts), and the plans bind ``fc.{2 i}``.  Returns ``sd`` with the head's keys under the names the plans know."""
    n_dp = int(mlp.dropout > 0) + int(mlp.dropout1d > 0)
    if n_dp == 0:
        return sd
    out = {k: v for k, v in sd.items() if not k.startswith(prefix + "fc.")}
    for i in range(mlp.n_hidden_layers + 2):
        for leaf in ("weight", "bias"):
            src = f"{prefix}fc.{i * (2 + n_dp)}.{leaf}"
            if src in sd:
                out[f"{prefix}fc.{2 * i}.{leaf}"] = sd[src]
    return out


class NativeError(RuntimeError):
    pass


WN_BPIPE_MIN_CLIPS = 105  # csrc/wavenet_plan.hip: kBpipeMinClips (tests/test_host_logic.py holds the two together)
WN_BPIPE_ALWAYS_CLIPS = 129  # csrc/wavenet_plan.hip: kBpipeAlwaysClips = one more than a ring takes


def wn_bpipe_by_default(batch: int) -> bool:
    """csrc/wavenet_plan.hip: bpipe_by_default - which batches of a stage-pipeline network run in groups of 16 clips (wavenet_bpipe.hip) unless the plan
    switch MMK_WN_BPIPE says otherwise: more than one ring's 128, and from 105 on the counts the ring's two-clip visits do not take (odd ones)"""
    return batch >= WN_BPIPE_ALWAYS_CLIPS or (batch >= WN_BPIPE_MIN_CLIPS and batch % 2 != 0)
TUNING_CHARS = 256        # include/mmk.h: MMK_TUNING_CHARS

# Execution switches handed to every plan this process creates, as {"MMK_WN_CHAIN": "0", ...} (merged under a network's own
# ``exec_tuning``).  They travel inside the plan's config (``tuning``): the library reads no environment variable, so nothing outside
# this dictionary and the network decides which kernel a plan gets.  The parity tests use it to put one network on every kernel.
PLAN_TUNING: Dict[str, str] = {}


def tuning_text(*dicts) -> bytes:
    merged = {}
    for d in dicts:
        merged.update(d or {})
    text = ";".join(f"{k}={v}" for k, v in merged.items())
    if len(text) >= TUNING_CHARS:
        raise ValueError(f"execution switches do not fit the config's {TUNING_CHARS} characters: {text}")
    return text.encode()


class WaveNetConfig(C.Structure):
    _fields_ = [
        ("n_layers", i32), ("kernel_size", i32 * MAX_LAYERS), ("dilation", i32 * MAX_LAYERS),
        ("q_levels", i32), ("in_dim", i32), ("dim_dilated", i32), ("residuals_dim", i32), ("skips_dim", i32),
        ("n_cond", i32), ("cond_in_dim", i32 * MAX_COND), ("cond_dim", i32 * MAX_COND), ("cond_q_levels", i32 * MAX_COND),
        ("bias", i32), ("gated", i32), ("act_f", i32), ("act_g", i32), ("head_kind", i32), ("mlp_hidden", i32), ("mlp_n_hidden", i32), ("mlp_act", i32),
        ("out_dim", i32), ("learn_temp", i32), ("min_temp", f32), ("max_batch", i32),
        ("res_explicit", i32), ("layer_has_res", i32 * MAX_LAYERS), ("layerwise_inputs", i32), ("exec_mode", i32), ("with_affine_residuals", i32),
        ("n_targets", i32), ("x_out_dim", i32 * MAX_STREAMS), ("x_mlp_hidden", i32 * MAX_STREAMS), ("x_mlp_n_hidden", i32 * MAX_STREAMS),
        ("x_learn_temp", i32 * MAX_STREAMS), ("x_min_temp", f32 * MAX_STREAMS),
        ("tuning", C.c_char * TUNING_CHARS),
    ]


class SrnnConfig(C.Structure):
    _fields_ = [
        ("n_tiers", i32), ("frame_size", i32 * MAX_TIERS), ("hidden_dim", i32), ("rnn_kind", i32),
        ("rnn_bias", i32), ("h0_ones", i32), ("q_levels", i32), ("mlp_hidden", i32), ("mlp_n_hidden", i32),
        ("learn_temp", i32), ("mlp_act", i32), ("min_temp", f32), ("max_batch", i32), ("n_rnn", i32), ("exec_mode", i32),
        ("n_inputs", i32), ("n_targets", i32), ("inputs_mode", i32), ("in_class", i32 * MAX_STREAMS),
        ("x_q_levels", i32 * MAX_STREAMS), ("x_mlp_hidden", i32 * MAX_STREAMS), ("x_mlp_n_hidden", i32 * MAX_STREAMS),
        ("x_learn_temp", i32 * MAX_STREAMS), ("x_min_temp", f32 * MAX_STREAMS),
        ("tuning", C.c_char * TUNING_CHARS),
    ]


class S2SConfig(C.Structure):
    _fields_ = [
        ("in_dim", i32), ("out_dim", i32), ("model_dim", i32), ("hop", i32), ("enc_n_lstm", i32),
        ("dec_n_lstm", i32), ("out_abs", i32), ("max_batch", i32), ("enc_downsampling", i32), ("dec_upsampling", i32),
        ("enc_apply_residuals", i32), ("dec_apply_residuals", i32),
        ("in_classes", i32), ("head_kind", i32), ("mlp_hidden", i32), ("mlp_n_hidden", i32), ("mlp_act", i32), ("learn_temp", i32), ("min_temp", f32),
        ("exec_mode", i32),
        ("tuning", C.c_char * TUNING_CHARS),
    ]


_SIGNATURES = {
    "mmk_abi_version": (i32, []),
    "mmk_config_bytes": (i64, [i32]),
    "mmk_build_digest": (cp, []),
    "mmk_last_error": (cp, []),
    "mmk_pack_launch_count": (i64, []),
    "mmk_fingerprint_u32": (i32, [vp, i64, vp, vp]),
    "mmk_fingerprint_buffers_u32": (i32, [vp, vp, i32, vp, vp]),
    "mmk_mulaw_compress_f32_i64": (i32, [vp, vp, i64, i32, f32, vp, vp]),
    "mmk_mulaw_expand_i64_f32": (i32, [vp, vp, i64, i32, f32, vp, vp]),
    "mmk_resample_n_out": (i64, [i64, i32, i32]),
    "mmk_resample_f32": (i32, [vp, i64, i32, i64, vp, i32, i32, i32, vp, i64, vp]),
    "mmk_stft_n_frames": (i64, [i64, i32, i32, i32]),
    "mmk_stft_mag_f32": (i32, [vp, i64, i32, i64, i32, i32, i32, vp, vp]),
    "mmk_stft_f32": (i32, [vp, i64, i32, i64, i32, i32, i32, i32, i32, vp, vp]),
    "mmk_istft_n_samples": (i64, [i64, i32, i32]),
    "mmk_istft_workspace_floats": (C.c_size_t, [i32, i64, i32]),
    "mmk_istft_f32": (i32, [vp, i32, i32, i64, i32, i32, vp, vp, vp]),
    "mmk_gla_workspace_floats": (C.c_size_t, [i32, i64, i32, i32]),
    "mmk_gla_f32": (i32, [vp, vp, i32, i64, i32, i32, i32, f32, vp, vp, vp]),
    "mmk_packed_weight_floats": (i64, [i32, i32]),
    "mmk_pack_weight_f32": (i32, [vp, i64, i32, i32, vp, vp]),
    "mmk_linear_f32": (i32, [vp, i64, i32, vp, vp, i32, i32, vp, i64, i32, vp]),
    "mmk_categorical_sample_f32_i64": (i32, [vp, i64, i32, i32, i32, f32, vp, vp, vp, i64, vp]),
    "mmk_wavenet_plan_create": (i32, [C.POINTER(WaveNetConfig), C.POINTER(vp)]),
    "mmk_wavenet_plan_destroy": (None, [vp]),
    "mmk_wavenet_plan_bind": (i32, [vp, cp, vp, i64]),
    "mmk_wavenet_receptive_field": (i64, [vp]),
    "mmk_wavenet_workspace_bytes": (C.c_size_t, [vp]),
    "mmk_wavenet_commit": (i32, [vp, vp, C.c_size_t, vp]),
    "mmk_wavenet_warmup": (i32, [vp, i32, vp, i64, C.POINTER(vp), C.POINTER(i64), i64, i64, vp]),
    "mmk_wavenet_generate": (i32, [vp, i32, vp, i64, C.POINTER(vp), C.POINTER(i64), i64, i64, vp, vp, vp]),
    "mmk_wavenet_last_logits": (i32, [vp, i32, vp, i64, vp]),
    "mmk_wavenet_last_logits_of": (i32, [vp, i32, i32, vp, i64, vp]),
    "mmk_wavenet_profile_steps": (i32, [vp, i32, vp, i64, C.POINTER(vp), C.POINTER(i64), i64, i64,
                                        C.POINTER(C.c_double), C.POINTER(i64), vp]),
    "mmk_wavenet_mode": (i32, [vp]),
    "mmk_wavenet_pair_visits": (i32, [vp]),
    "mmk_wavenet_sync_status": (i32, [vp, vp]),
    "mmk_wavenet_inject_sync_error": (i32, [vp, vp]),
    "mmk_srnn_plan_create": (i32, [C.POINTER(SrnnConfig), C.POINTER(vp)]),
    "mmk_srnn_plan_destroy": (None, [vp]),
    "mmk_srnn_plan_bind": (i32, [vp, cp, vp, i64]),
    "mmk_srnn_workspace_bytes": (C.c_size_t, [vp]),
    "mmk_srnn_commit": (i32, [vp, vp, C.c_size_t, vp]),
    "mmk_srnn_reset": (i32, [vp, vp]),
    "mmk_srnn_warmup": (i32, [vp, i32, vp, i64, i64, vp]),
    "mmk_srnn_generate": (i32, [vp, i32, vp, i64, i64, i64, vp, vp, vp]),
    "mmk_srnn_last_logits": (i32, [vp, i32, vp, i64, vp]),
    "mmk_srnn_warmup_multi": (i32, [vp, i32, C.POINTER(vp), C.POINTER(i64), i64, vp]),
    "mmk_srnn_generate_multi": (i32, [vp, i32, C.POINTER(vp), C.POINTER(i64), i64, i64, vp, vp, vp]),
    "mmk_srnn_last_logits_of": (i32, [vp, i32, i32, vp, i64, vp]),
    "mmk_srnn_resident_blocks": (i64, [vp]),
    "mmk_srnn_resident_warmups": (i64, [vp]),
    "mmk_srnn_sync_status": (i32, [vp, vp]),
    "mmk_srnn_inject_sync_error": (i32, [vp, vp]),
    "mmk_s2s_plan_create": (i32, [C.POINTER(S2SConfig), C.POINTER(vp)]),
    "mmk_s2s_plan_destroy": (None, [vp]),
    "mmk_s2s_plan_bind": (i32, [vp, cp, vp, i64]),
    "mmk_s2s_workspace_bytes": (C.c_size_t, [vp]),
    "mmk_s2s_commit": (i32, [vp, vp, C.c_size_t, vp]),
    "mmk_s2s_step": (i32, [vp, i32, vp, i64, i64, vp, i64, i64, vp]),
    "mmk_s2s_generate": (i32, [vp, i32, vp, i64, i64, i64, i64, i64, vp]),
    "mmk_s2s_step_classes": (i32, [vp, i32, vp, i64, i64, vp, i64, i64, vp]),
    "mmk_s2s_generate_classes": (i32, [vp, i32, vp, i64, i64, i64, i64, i64, vp]),
    "mmk_s2s_last_logits": (i32, [vp, i32, vp, i64, vp]),
    "mmk_s2s_sync_status": (i32, [vp, vp]),
    "mmk_s2s_inject_sync_error": (i32, [vp, vp]),
    "mmk_s2s_resident_launches": (i64, [vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load_library(path: Optional[str] = None):
    """dlopen the HIP library and type its entry points (no GPU needed for this)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise NativeError(
            f"{path} is missing: build it with `python -m mimikit_amd.build` (hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback for the generate path.")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mmk_abi_version() != ABI_VERSION:
        raise NativeError(f"ABI version mismatch: library reports {lib.mmk_abi_version()}, binding expects {ABI_VERSION} "
                          "(a stale libmmk_hip.so: rebuild with `python -m mimikit_amd.build`)")
    if not DIAGNOSTIC or path != LIB_PATH:          # (the product library must be the one these sources make; a diagnostic build or a
        from .build import source_digest             #  variant is asked for by name through MMK_DIAG_LIB and is never found under the product's name)
        have, want = lib.mmk_build_digest().decode(), source_digest()
        if have != want:
            raise NativeError(f"{path} was built from other sources (digest {have}, the tree's is {want}): rebuild with "
                              "`python -m mimikit_amd.build`")
    for which, struct in enumerate((WaveNetConfig, SrnnConfig, S2SConfig)):
        if lib.mmk_config_bytes(which) != C.sizeof(struct):
            raise NativeError(f"{struct.__name__} is {C.sizeof(struct)} bytes here and {lib.mmk_config_bytes(which)} in the library "
                              "(a stale libmmk_hip.so: rebuild with `python -m mimikit_amd.build`)")
    _lib = lib
    return lib


def lib():
    return load_library()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().mmk_last_error().decode("utf-8", "replace")
        kind = {-1: ValueError, -3: NotImplementedError, -6: KeyError}.get(rc, NativeError)
        raise kind(f"{what or 'libmmk_hip'} failed (code {rc}): {msg}")


def require_device(*tensors: torch.Tensor):
    """The hot path only runs on a HIP device; anything else is an error, never a fallback."""
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"expected a torch.Tensor, got {type(t)}")
        if t.device.type != "cuda":
            raise RuntimeError(
                f"mimikit_amd runs its generate path on the MI355X only: got a tensor on '{t.device}'. "
                "Move the network and its inputs to the HIP device ('cuda'); there is no CPU implementation "
                "in this package (the CPU restatement under oracle/ is test infrastructure).")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------
# feature functionals
# ---------------------------------------------------------------------------
def mulaw_compress(x: torch.Tensor, q_levels: int, compression: float, edges: torch.Tensor) -> torch.Tensor:
    require_device(x, edges)
    x = x.contiguous()
    if x.dtype != torch.float32:
        x = x.float()
    out = torch.empty(x.shape, dtype=torch.int64, device=x.device)
    check(lib().mmk_mulaw_compress_f32_i64(ptr(x), ptr(out), x.numel(), q_levels, compression, ptr(edges),
                                           stream_ptr(x.device)), "mmk_mulaw_compress_f32_i64")
    return out


def mulaw_expand(codes: torch.Tensor, q_levels: int, compression: float, table: torch.Tensor) -> torch.Tensor:
    require_device(codes, table)
    codes = codes.contiguous()
    if codes.dtype != torch.int64:
        codes = codes.long()
    out = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
    check(lib().mmk_mulaw_expand_i64_f32(ptr(codes), ptr(out), codes.numel(), q_levels, compression, ptr(table),
                                         stream_ptr(codes.device)), "mmk_mulaw_expand_i64_f32")
    return out


def resample(x: torch.Tensor, table: torch.Tensor, orig: int, new: int, width: int) -> torch.Tensor:
    """x: (..., n) fp32 -> (..., ceil(new * n / orig)); `table`: the (new, 2 * width + orig) polyphase filter bank"""
    require_device(x, table)
    if x.dtype != torch.float32:
        x = x.float()
    lead = x.shape[:-1]
    x2 = _rows(x)
    n_out = lib().mmk_resample_n_out(x2.shape[-1], orig, new)
    out = torch.empty((x2.shape[0], n_out), dtype=torch.float32, device=x.device)
    check(lib().mmk_resample_f32(ptr(x2), x2.stride(0), x2.shape[0], x2.shape[-1], ptr(table), orig, new, width, ptr(out), out.stride(0),
                                 stream_ptr(x.device)), "mmk_resample_f32")
    return out.reshape(*lead, n_out)


def _rows(x: torch.Tensor) -> torch.Tensor:
    """(..., n) -> (rows, n) with unit stride along n and ONE stride between rows, without a copy where the layout already
    is that (the length fix-up of STFT hands over a slice ``x[..., -keep:]`` of a contiguous tensor: rows keep their old
    stride, and the kernels take a row stride)"""
    if x.stride(-1) == 1 or x.shape[-1] == 1:
        if x.dim() == 1:
            return x.unsqueeze(0)
        if x.dim() == 2:
            return x
        lead, st = x.shape[:-1], x.stride()[:-1]
        if all(st[i] == st[i + 1] * lead[i + 1] for i in range(len(lead) - 1)):
            return x.as_strided((int(torch.Size(lead).numel()), x.shape[-1]), (st[-1], 1), x.storage_offset())
    return x.reshape(-1, x.shape[-1]).contiguous()


def stft_mag(x: torch.Tensor, n_fft: int, hop: int, center: bool) -> torch.Tensor:
    """x: (..., n_samples) fp32 -> (..., n_frames, n_fft//2+1)"""
    require_device(x)
    if x.dtype != torch.float32:
        x = x.float()
    lead = x.shape[:-1]
    x2 = _rows(x)
    n = x2.shape[-1]
    n_frames = lib().mmk_stft_n_frames(n, n_fft, hop, int(center))
    if n_frames <= 0:
        raise RuntimeError(f"stft: input of {n} samples is shorter than one frame of {n_fft}")
    out = torch.empty((x2.shape[0], n_frames, n_fft // 2 + 1), dtype=torch.float32, device=x.device)
    check(lib().mmk_stft_mag_f32(ptr(x2), x2.stride(0), x2.shape[0], n, n_fft, hop, int(center), ptr(out),
                                 stream_ptr(x.device)), "mmk_stft_mag_f32")
    return out.reshape(*lead, n_frames, n_fft // 2 + 1)


STFT_COORDINATES = {"car": 0, "pol": 1, "angle": 2}


def stft(x: torch.Tensor, n_fft: int, hop: int, center: bool, pad_mode: str, coordinate: str) -> torch.Tensor:
    """x: (..., n_samples) fp32 -> (..., n_frames, n_fft//2+1[, 2]) in the 'car' / 'pol' / 'angle' coordinate"""
    require_device(x)
    if pad_mode not in ("constant", "reflect"):
        raise NotImplementedError(f"HIP STFT covers pad_mode 'constant' and 'reflect', got '{pad_mode}'")
    if x.dtype != torch.float32:
        x = x.float()
    lead = x.shape[:-1]
    x2 = _rows(x)
    n = x2.shape[-1]
    n_frames = lib().mmk_stft_n_frames(n, n_fft, hop, int(center))
    if n_frames <= 0:
        raise RuntimeError(f"stft: input of {n} samples is shorter than one frame of {n_fft}")
    tail = (n_frames, n_fft // 2 + 1) + (() if coordinate == "angle" else (2,))
    out = torch.empty((x2.shape[0],) + tail, dtype=torch.float32, device=x.device)
    check(lib().mmk_stft_f32(ptr(x2), x2.stride(0), x2.shape[0], n, n_fft, hop, int(center), int(pad_mode == "reflect"),
                             STFT_COORDINATES[coordinate], ptr(out), stream_ptr(x.device)), "mmk_stft_f32")
    return out.reshape(*lead, *tail)


def istft(spec: torch.Tensor, n_fft: int, hop: int, polar: bool) -> torch.Tensor:
    """spec: (..., n_frames, n_fft//2+1, 2) fp32, (re, im) or (abs, angle) -> (..., hop * (n_frames - 1))"""
    require_device(spec)
    if spec.shape[-1] != 2 or spec.shape[-2] != n_fft // 2 + 1:
        raise RuntimeError(f"istft: expected (..., n_frames, {n_fft // 2 + 1}, 2), got {tuple(spec.shape)}")
    lead = spec.shape[:-3]
    s3 = spec.reshape(-1, *spec.shape[-3:]).contiguous().float()
    batch, n_frames = s3.shape[0], s3.shape[1]
    n_out = lib().mmk_istft_n_samples(n_frames, n_fft, hop)
    n_work = lib().mmk_istft_workspace_floats(batch, n_frames, n_fft)
    work = torch.empty(n_work, dtype=torch.float32, device=spec.device) if n_work else None
    out = torch.empty((batch, n_out), dtype=torch.float32, device=spec.device)
    check(lib().mmk_istft_f32(ptr(s3), int(polar), batch, n_frames, n_fft, hop, ptr(work), ptr(out), stream_ptr(spec.device)),
          "mmk_istft_f32")
    return out.reshape(*lead, n_out)


def griffin_lim(mag: torch.Tensor, n_fft: int, hop: int, n_iter: int = 32, momentum: float = 0.99,
                init: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mag: (..., n_frames, n_fft//2+1) fp32; init: complex64 of the same shape (initial phase estimates) or None (ones)"""
    require_device(mag)
    lead = mag.shape[:-2]
    m3 = mag.reshape(-1, *mag.shape[-2:]).contiguous().float()
    batch, n_frames = m3.shape[0], m3.shape[1]
    init_ri = None
    if init is not None:
        if tuple(init.shape) != tuple(mag.shape) or not init.is_complex():
            raise RuntimeError("griffin_lim: init must be a complex tensor shaped like mag")
        init_ri = torch.view_as_real(init.to(torch.complex64).reshape(m3.shape).contiguous())
    n_out = lib().mmk_istft_n_samples(n_frames, n_fft, hop)
    work = torch.empty(lib().mmk_gla_workspace_floats(batch, n_frames, n_fft, hop), dtype=torch.float32, device=mag.device)
    out = torch.empty((batch, n_out), dtype=torch.float32, device=mag.device)
    check(lib().mmk_gla_f32(ptr(m3), ptr(init_ri) if init_ri is not None else None, batch, n_frames, n_fft, hop, n_iter,
                            momentum, ptr(work), ptr(out), stream_ptr(mag.device)), "mmk_gla_f32")
    return out.reshape(*lead, n_out)


# ---------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------
def pack_weight(w: torch.Tensor) -> torch.Tensor:
    require_device(w)
    w = w.contiguous().float()
    n, k = w.shape
    out = torch.empty(lib().mmk_packed_weight_floats(n, k), dtype=torch.float32, device=w.device)
    check(lib().mmk_pack_weight_f32(ptr(w), w.stride(0), n, k, ptr(out), stream_ptr(w.device)), "mmk_pack_weight_f32")
    return out


def linear(x: torch.Tensor, packed_w: torch.Tensor, bias: Optional[torch.Tensor], n: int, k: int,
           act: str = "none") -> torch.Tensor:
    require_device(x, packed_w, bias)
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    y = torch.empty((x2.shape[0], n), dtype=torch.float32, device=x.device)
    check(lib().mmk_linear_f32(ptr(x2), x2.stride(0), x2.shape[0], ptr(packed_w), ptr(bias), n, k, ptr(y), y.stride(0),
                               ACT[act], stream_ptr(x.device)), "mmk_linear_f32")
    return y.reshape(*x.shape[:-1], n)


def categorical_sample(logits: torch.Tensor, n_classes: int, has_temp_col: bool, min_temp: float,
                       temperature: Optional[torch.Tensor], uniforms: Optional[torch.Tensor]) -> torch.Tensor:
    require_device(logits, temperature, uniforms)
    lg = logits.reshape(-1, logits.shape[-1])
    if lg.stride(-1) != 1:
        lg = lg.contiguous()
    rows = lg.shape[0]
    out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    check(lib().mmk_categorical_sample_f32_i64(ptr(lg), lg.stride(0), rows, n_classes, int(has_temp_col), min_temp,
                                               ptr(temperature), ptr(uniforms), ptr(out), 1,
                                               stream_ptr(logits.device)), "mmk_categorical_sample_f32_i64")
    return out.reshape(logits.shape[:-1])


# ---------------------------------------------------------------------------
# plans
# ---------------------------------------------------------------------------
class _Plan:
    """Owns one native plan handle and its torch-allocated workspace."""
    _prefix = ""

    def __init__(self, cfg_struct, device: torch.device):
        self._lib = lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"plans live on the HIP device; got '{device}' (no CPU implementation in this package)")
        self.cfg = cfg_struct
        if not cfg_struct.tuning:         # (a network's own switches are already in it; the process-wide ones otherwise)
            cfg_struct.tuning = tuning_text(PLAN_TUNING)
        handle = vp()
        check(getattr(self._lib, self._prefix + "_plan_create")(C.byref(cfg_struct), C.byref(handle)),
              self._prefix + "_plan_create")
        self.handle = handle
        self.workspace = None
        self._bound = {}

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                if torch.cuda.is_available():
                    torch.cuda.synchronize(self.device)
                getattr(self._lib, self._prefix + "_plan_destroy")(self.handle)
                self.handle = None
        except Exception:
            pass

    def bind_state_dict(self, tensors):
        for key, t in tensors.items():
            if not torch.is_floating_point(t) or t.dim() == 0:
                continue
            require_device(t)
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            self._bound[key] = t  # keep alive: the plan reads these pointers at commit
            check(getattr(self._lib, self._prefix + "_plan_bind")(self.handle, key.encode(), ptr(t), t.numel()),
                  self._prefix + "_plan_bind")

    def commit(self):
        need = getattr(self._lib, self._prefix + "_workspace_bytes")(self.handle)
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        base = self.workspace.data_ptr()
        aligned = (base + 255) // 256 * 256
        check(getattr(self._lib, self._prefix + "_commit")(self.handle, aligned, need, stream_ptr(self.device)),
              self._prefix + "_commit")
        self.workspace_bytes = need


def pack_launch_count() -> int:
    """weight re-packing kernels launched so far (diagnostic; see include/mmk.h)"""
    return int(lib().mmk_pack_launch_count())


def weights_identity(module: torch.nn.Module):
    """host-side identity of a module's weights as a plan packed them: storage address, in-place version counter and shape of
    every state_dict entry.  Optimiser steps and ``load_state_dict`` bump the version, ``.to(device)`` moves the storage."""
    return tuple((k, v.data_ptr(), v._version, tuple(v.shape)) for k, v in module.state_dict(keep_vars=True).items())


def weights_fingerprint(module: torch.nn.Module) -> int:
    """A write through ``.data`` (``p.data.copy_(ema)``, weight surgery) bumps no version counter and moves no storage: a
    position-mixed hash of all fp32 entries is summed on their device (``mmk_fingerprint_buffers_u32``: the entries where they lie,
    one launch per 96 of them, one read-back - which waits for the stream).  Taken where a generation starts (``before_generate``),
    not on every step of one.  Entries on the host are hashed there."""
    entries = [v.detach() for v in module.state_dict(keep_vars=True).values() if v.dtype == torch.float32 and v.numel() > 0]
    if not entries:
        return 0
    dev = entries[0].device
    if dev.type != "cuda":
        import zlib
        return zlib.crc32(b"".join(v.contiguous().cpu().numpy().tobytes() for v in entries))
    entries = [v if v.is_contiguous() else v.contiguous() for v in entries]
    n = len(entries)
    ptrs = (C.c_void_p * n)(*[v.data_ptr() for v in entries])
    counts = (i64 * n)(*[v.numel() for v in entries])
    out = torch.empty(1, dtype=torch.int64, device=dev)
    check(lib().mmk_fingerprint_buffers_u32(ptrs, counts, n, ptr(out), stream_ptr(dev)), "mmk_fingerprint_buffers_u32")
    return int(out.item())


class WeightsTracker:
    """decides whether a plan's packed copy of the weights is still current"""

    def __init__(self):
        self.ident, self.print_ = None, None

    def changed(self, module: torch.nn.Module, content: bool) -> bool:
        """``content``: also compare the content fingerprint (the start of a generation); without it only the host-side identity"""
        ident = weights_identity(module)
        if ident != self.ident:
            return True
        return content and weights_fingerprint(module) != self.print_

    def committed(self, module: torch.nn.Module):
        self.ident, self.print_ = weights_identity(module), weights_fingerprint(module)


def abs_ptr(view: torch.Tensor, t_first: int) -> int:
    """address A such that A + t * stride(1) * itemsize is ``view[:, t - t_first]``: lets a window
    view of a longer (batch, T, ...) tensor be addressed by absolute time on the device"""
    return view.data_ptr() - t_first * view.stride(1) * view.element_size()


def _cond_arrays(cond: Sequence[torch.Tensor], t_first: int):
    n = len(cond)
    ptrs = (vp * max(n, 1))(*[abs_ptr(c, t_first) for c in cond])
    strides = (i64 * max(n, 1))(*[c.stride(0) for c in cond])
    return ptrs, strides


class WaveNetPlan(_Plan):
    """``in0`` / ``cond`` arguments are (batch, T[, dim]) tensors (or views) whose column 0 is
    absolute time ``t_first``; all ``t`` arguments are absolute times."""
    _prefix = "mmk_wavenet"

    @property
    def rf(self) -> int:
        return self._lib.mmk_wavenet_receptive_field(self.handle)

    def _check_inputs(self, in0, cond):
        require_device(in0, *cond)
        if self.cfg.q_levels > 0:
            if in0.dtype != torch.int64 or in0.dim() != 2 or in0.stride(1) != 1:
                raise ValueError("input 0 must be int64 (batch, T) with unit stride along time")
        else:
            if in0.dtype != torch.float32 or in0.dim() != 3 or in0.stride(2) != 1 or in0.stride(1) != in0.shape[2]:
                raise ValueError("input 0 must be fp32 (batch, T, dim), contiguous along time and dim")
        if len(cond) != self.cfg.n_cond:
            raise ValueError(f"expected {self.cfg.n_cond} conditioning inputs, got {len(cond)}")
        for j, c in enumerate(cond):
            if self.cfg.cond_q_levels[j] > 0:
                if c.dtype != torch.int64 or c.dim() != 2 or c.stride(1) != 1:
                    raise ValueError(f"input {j + 1} is a class stream: int64 (batch, T) with unit stride along time")
            elif c.dtype != torch.float32 or c.dim() != 3 or c.stride(2) != 1 or c.stride(1) != c.shape[2]:
                raise ValueError("conditioning inputs must be fp32 (batch, T, dim), contiguous along time and dim")

    def warmup(self, in0: torch.Tensor, cond: Sequence[torch.Tensor], t_begin: int, t_end: int, t_first: int = 0):
        self._check_inputs(in0, cond)
        ptrs, strides = _cond_arrays(cond, t_first)
        check(self._lib.mmk_wavenet_warmup(self.handle, in0.shape[0], abs_ptr(in0, t_first), in0.stride(0), ptrs,
                                           strides, t_begin, t_end, stream_ptr(self.device)), "mmk_wavenet_warmup")

    def generate(self, in0: torch.Tensor, cond: Sequence[torch.Tensor], t0: int, n_steps: int,
                 temperature: Optional[torch.Tensor] = None, uniforms: Optional[torch.Tensor] = None,
                 t_first: int = 0):
        self._check_inputs(in0, cond)
        require_device(temperature, uniforms)
        n_tgt = max(int(self.cfg.n_targets), 1)
        if uniforms is not None and (uniforms.dtype != torch.float32 or not uniforms.is_contiguous()
                                     or uniforms.numel() != n_tgt * in0.shape[0] * n_steps):
            raise ValueError("uniforms must be contiguous fp32 of shape (batch, n_steps) - (n_targets, batch, n_steps) with several targets")
        ptrs, strides = _cond_arrays(cond, t_first)
        check(self._lib.mmk_wavenet_generate(self.handle, in0.shape[0], abs_ptr(in0, t_first), in0.stride(0), ptrs,
                                             strides, t0, n_steps, ptr(temperature), ptr(uniforms),
                                             stream_ptr(self.device)), "mmk_wavenet_generate")

    @property
    def persistent(self) -> bool:
        return bool(self._lib.mmk_wavenet_mode(self.handle))

    @property
    def chain(self) -> bool:
        """persistent mode with one hand-off per layer (csrc/wavenet_chain.hip)"""
        return self._lib.mmk_wavenet_mode(self.handle) == 2

    @property
    def layer_pipelined(self) -> bool:
        """persistent mode with four workgroups per clip that own whole layers (csrc/wavenet_lpipe.hip)"""
        return self._lib.mmk_wavenet_mode(self.handle) == 4

    @property
    def stage_pipelined(self) -> bool:
        """persistent mode with one layer per stage of 8 CUs: the clips streamed through one at a time (csrc/wavenet_spipe.hip) or,
        large batches, in groups of 16 (csrc/wavenet_bpipe.hip)"""
        return self._lib.mmk_wavenet_mode(self.handle) in (5, 6)

    @property
    def pair_visits(self) -> bool:
        """the last launch of the one-clip ring took two clips per visit (csrc/wavenet_spipe_pair.inc)"""
        return bool(self._lib.mmk_wavenet_pair_visits(self.handle))

    @property
    def batch_pipelined(self) -> bool:
        """the stage pipeline with groups of 16 clips per visit on the matrix pipe (csrc/wavenet_bpipe.hip)"""
        return self._lib.mmk_wavenet_mode(self.handle) == 6

    def sync_status(self):
        """wait for the stream and raise if a hand-off inside the persistent kernel timed out"""
        check(self._lib.mmk_wavenet_sync_status(self.handle, stream_ptr(self.device)), "mmk_wavenet_sync_status")

    def inject_sync_error(self):
        """fault injection for tests: the next ``sync_status`` fails as after a timed-out hand-off (include/mmk.h)"""
        check(self._lib.mmk_wavenet_inject_sync_error(self.handle, stream_ptr(self.device)), "mmk_wavenet_inject_sync_error")

    def profile_steps(self, in0: torch.Tensor, cond: Sequence[torch.Tensor], t0: int, n_steps: int, t_first: int = 0):
        """measurement aid: per-kernel-class device time from HIP events (see include/mmk.h);
        returns {"layer_a": (ms_total, launches), "layer_b": ..., "other": ...}"""
        self._check_inputs(in0, cond)
        ptrs, strides = _cond_arrays(cond, t_first)
        ms = (C.c_double * 3)()
        cnt = (i64 * 3)()
        check(self._lib.mmk_wavenet_profile_steps(self.handle, in0.shape[0], abs_ptr(in0, t_first), in0.stride(0), ptrs,
                                                  strides, t0, n_steps, ms, cnt, stream_ptr(self.device)),
              "mmk_wavenet_profile_steps")
        return {name: (ms[i], cnt[i]) for i, name in enumerate(("layer_a", "layer_b", "other"))}

    def last_logits(self, batch: int, target: int = 0) -> torch.Tensor:
        c = self.cfg
        n = (c.out_dim + (1 if c.learn_temp else 0)) if target == 0 else (c.x_out_dim[target] + (1 if c.x_learn_temp[target] else 0))
        out = torch.empty((batch, n), dtype=torch.float32, device=self.device)
        check(self._lib.mmk_wavenet_last_logits_of(self.handle, target, batch, ptr(out), out.stride(0), stream_ptr(self.device)),
              "mmk_wavenet_last_logits_of")
        return out


SPIPE_MAX_CLIPS = 128      # clips one ring of the stage pipeline streams (csrc/wavenet_spipe.h: kSpMaxClips)
BPIPE_MAX_CLIPS = 512      # clips one launch of its large-batch form takes, in groups of 16 (csrc/wavenet_bpipe.h: kBpMaxClips)


class WaveNetPlanSet:
    """Several :class:`WaveNetPlan` s, each owning a contiguous slice of the batch, behind the interface of one.

    The stage pipeline (``csrc/wavenet_spipe.hip``) streams at most ``SPIPE_MAX_CLIPS`` clips through its ring; the reference's
    loop takes any batch (``loops/generate.py:207-219``), so a larger one runs as successive passes of the same kernel, one
    per slice - each slice has its own history rings and message blocks, the weights are packed once per plan."""

    def __init__(self, plans: Sequence["WaveNetPlan"], sizes: Sequence[int]):
        self.plans, self.sizes = list(plans), list(sizes)
        self.cfg, self.device = self.plans[0].cfg, self.plans[0].device

    def _slices(self, batch: int):
        off = 0
        for plan, size in zip(self.plans, self.sizes):
            n = min(size, batch - off)
            if n <= 0:
                break
            yield plan, off, off + n
            off += n
        if off < batch:
            raise ValueError(f"batch of {batch} clips exceeds the {sum(self.sizes)} this plan set was built for")

    def bind_state_dict(self, tensors):
        for plan in self.plans:
            plan.bind_state_dict(tensors)

    def commit(self):
        for plan in self.plans:
            plan.commit()

    @property
    def rf(self) -> int:
        return self.plans[0].rf

    def warmup(self, in0, cond, t_begin, t_end, t_first=0):
        for plan, a, b in self._slices(in0.shape[0]):
            plan.warmup(in0[a:b], [c[a:b] for c in cond], t_begin, t_end, t_first=t_first)

    def generate(self, in0, cond, t0, n_steps, temperature=None, uniforms=None, t_first=0):
        for plan, a, b in self._slices(in0.shape[0]):
            plan.generate(in0[a:b], [c[a:b] for c in cond], t0, n_steps,
                          None if temperature is None else temperature[a:b].contiguous(),
                          None if uniforms is None else uniforms[a:b].contiguous(), t_first=t_first)

    persistent = property(lambda self: self.plans[0].persistent)
    chain = property(lambda self: self.plans[0].chain)
    layer_pipelined = property(lambda self: self.plans[0].layer_pipelined)
    stage_pipelined = property(lambda self: self.plans[0].stage_pipelined)
    batch_pipelined = property(lambda self: self.plans[0].batch_pipelined)
    pair_visits = property(lambda self: all(plan.pair_visits for plan in self.plans))

    def sync_status(self):
        err = None
        for plan in self.plans:          # (every plan's error word is read and cleared; the first error is the one raised)
            try:
                plan.sync_status()
            except NativeError as e:
                err = err or e
        if err is not None:
            raise err

    def inject_sync_error(self):
        self.plans[-1].inject_sync_error()

    def profile_steps(self, *args, **kwargs):
        raise NotImplementedError("profile_steps measures the per-layer launch path of ONE plan")

    def last_logits(self, batch: int, target: int = 0) -> torch.Tensor:
        return torch.cat([plan.last_logits(b - a, target) for plan, a, b in self._slices(batch)], dim=0)


def make_wavenet_plan(describe, batch: int, device) -> "WaveNetPlan":
    """``describe(max_batch)`` -> :class:`WaveNetConfig`.  One plan for the batch, unless the batch is beyond what one launch of the
    stage pipeline takes and the network is one that kernel runs: then evenly sized slices - of at most ``BPIPE_MAX_CLIPS`` clips where
    the plan takes the large-batch form (groups of 16 clips), else of at most ``SPIPE_MAX_CLIPS``."""
    batch = max(int(batch), 1)
    if batch > SPIPE_MAX_CLIPS:
        n = -(-batch // BPIPE_MAX_CLIPS)
        size = -(-batch // n)
        first = WaveNetPlan(describe(size), device)
        if first.batch_pipelined:
            if n == 1:
                return first
            assert max(int(first.cfg.n_targets), 1) == 1, "a plan set slices clips along dimension 0: one target only"
            sizes = [size] * (n - 1) + [batch - size * (n - 1)]
            return WaveNetPlanSet([first] + [WaveNetPlan(describe(sz), device) for sz in sizes[1:]], sizes)
        del first
        n = -(-batch // SPIPE_MAX_CLIPS)
        size = -(-batch // n)
        first = WaveNetPlan(describe(size), device)
        if first.stage_pipelined:
            # (the set slices clips along dimension 0 of every per-clip tensor; with several targets the uniforms are (targets, clips, steps))
            assert max(int(first.cfg.n_targets), 1) == 1, "a plan set slices clips along dimension 0: one target only"
            sizes = [size] * (n - 1) + [batch - size * (n - 1)]
            return WaveNetPlanSet([first] + [WaveNetPlan(describe(sz), device) for sz in sizes[1:]], sizes)
    return WaveNetPlan(describe(batch), device)


class SrnnPlan(_Plan):
    _prefix = "mmk_srnn"

    def reset(self):
        check(self._lib.mmk_srnn_reset(self.handle, stream_ptr(self.device)), "mmk_srnn_reset")

    def _streams(self, idx):
        """one (batch, T) int64 tensor per input of the network (a single tensor: the network's only input)"""
        idx = (idx,) if isinstance(idx, torch.Tensor) else tuple(idx)
        n_in = max(int(self.cfg.n_inputs), 1)
        if len(idx) != n_in:
            raise ValueError(f"expected {n_in} input streams, got {len(idx)}")
        require_device(*idx)
        for x in idx:
            if x.dtype != torch.int64 or x.dim() != 2 or x.stride(1) != 1 or x.shape != idx[0].shape:
                raise ValueError("SampleRNN inputs must be int64 (batch, T) of one shape, contiguous along time")
        return idx

    def warmup(self, idx, prompt_len: int):
        """idx: (batch, >= prompt_len) prompt, column 0 = time 0 (a tuple of them for a network of several inputs)"""
        idx = self._streams(idx)
        if idx[0].shape[1] < prompt_len:
            raise ValueError(f"prompt tensor holds {idx[0].shape[1]} steps, prompt_len={prompt_len}")
        ptrs = (vp * len(idx))(*[ptr(x) for x in idx])
        strides = (i64 * len(idx))(*[x.stride(0) for x in idx])
        check(self._lib.mmk_srnn_warmup_multi(self.handle, idx[0].shape[0], ptrs, strides, prompt_len,
                                              stream_ptr(self.device)), "mmk_srnn_warmup_multi")

    def generate(self, idx, t0: int, n_steps: int, temperature=None, uniforms=None, t_first: int = 0):
        """idx: (batch, T) tensor or view whose column 0 is absolute time t_first (a tuple of them for several inputs: the class of
        target k is written into idx[k]); uniforms: (batch, n_steps), (n_targets, batch, n_steps) with several targets"""
        idx = self._streams(idx)
        require_device(temperature, uniforms)
        n_tgt = max(int(self.cfg.n_targets), 1)
        if uniforms is not None and (uniforms.dtype != torch.float32 or not uniforms.is_contiguous()
                                     or uniforms.numel() != n_tgt * idx[0].shape[0] * n_steps):
            raise ValueError("uniforms must be contiguous fp32 of shape (batch, n_steps) - (n_targets, batch, n_steps) with several targets")
        ptrs = (vp * len(idx))(*[abs_ptr(x, t_first) for x in idx])
        strides = (i64 * len(idx))(*[x.stride(0) for x in idx])
        check(self._lib.mmk_srnn_generate_multi(self.handle, idx[0].shape[0], ptrs, strides, t0, n_steps,
                                                ptr(temperature), ptr(uniforms), stream_ptr(self.device)),
              "mmk_srnn_generate_multi")

    def sync_status(self):
        """wait for the stream and raise if a wait inside the resident-mode kernels timed out"""
        check(self._lib.mmk_srnn_sync_status(self.handle, stream_ptr(self.device)), "mmk_srnn_sync_status")

    def inject_sync_error(self):
        """fault injection for tests: the next ``sync_status`` reports a timed-out wait (include/mmk.h)"""
        check(self._lib.mmk_srnn_inject_sync_error(self.handle, stream_ptr(self.device)), "mmk_srnn_inject_sync_error")

    def resident_blocks(self) -> int:
        """generate blocks run in resident mode so far (diagnostic, see include/mmk.h)"""
        return int(self._lib.mmk_srnn_resident_blocks(self.handle))

    def resident_warmups(self) -> int:
        """warm-ups run as one teacher-forced resident launch so far (diagnostic, see include/mmk.h)"""
        return int(self._lib.mmk_srnn_resident_warmups(self.handle))

    def last_logits(self, batch: int, target: int = 0) -> torch.Tensor:
        c = self.cfg
        n = (c.q_levels + (1 if c.learn_temp else 0)) if target == 0 else (c.x_q_levels[target] + (1 if c.x_learn_temp[target] else 0))
        out = torch.empty((batch, n), dtype=torch.float32, device=self.device)
        check(self._lib.mmk_srnn_last_logits_of(self.handle, target, batch, ptr(out), out.stride(0), stream_ptr(self.device)),
              "mmk_srnn_last_logits_of")
        return out


class S2SPlan(_Plan):
    _prefix = "mmk_s2s"

    def step(self, x: torch.Tensor) -> torch.Tensor:
        require_device(x)
        if x.dtype != torch.float32 or x.stride(2) != 1:
            raise ValueError("Seq2Seq input must be fp32 (batch, hop, n_bins) with unit stride on the last dim")
        y = torch.empty((x.shape[0], self.cfg.hop, self.cfg.out_dim), dtype=torch.float32, device=x.device)
        check(self._lib.mmk_s2s_step(self.handle, x.shape[0], ptr(x), x.stride(0), x.stride(1), ptr(y), y.stride(0),
                                     y.stride(1), stream_ptr(self.device)), "mmk_s2s_step")
        return y

    def generate(self, frames: torch.Tensor, t0: int, n_steps: int):
        require_device(frames)
        if frames.dtype != torch.float32 or frames.stride(2) != 1:
            raise ValueError("Seq2Seq frames must be fp32 (batch, T, n_bins) with unit stride on the last dim")
        check(self._lib.mmk_s2s_generate(self.handle, frames.shape[0], ptr(frames), frames.stride(0), frames.stride(1),
                                         t0, n_steps, frames.shape[1], stream_ptr(self.device)), "mmk_s2s_generate")

    # -- class indices in and out (embedding input + MLP head, IOSpec.mulaw_io) --------------------------------
    def step_classes(self, x: torch.Tensor) -> torch.Tensor:
        require_device(x)
        if x.dtype != torch.int64 or x.dim() != 2:
            raise ValueError("Seq2Seq class input must be int64 (batch, hop)")
        y = torch.empty((x.shape[0], self.cfg.hop), dtype=torch.int64, device=x.device)
        check(self._lib.mmk_s2s_step_classes(self.handle, x.shape[0], ptr(x), x.stride(0), x.stride(1), ptr(y), y.stride(0),
                                             y.stride(1), stream_ptr(self.device)), "mmk_s2s_step_classes")
        return y

    def generate_classes(self, classes: torch.Tensor, t0: int, n_steps: int):
        require_device(classes)
        if classes.dtype != torch.int64 or classes.dim() != 2:
            raise ValueError("Seq2Seq classes must be int64 (batch, T)")
        check(self._lib.mmk_s2s_generate_classes(self.handle, classes.shape[0], ptr(classes), classes.stride(0), classes.stride(1),
                                                 t0, n_steps, classes.shape[1], stream_ptr(self.device)), "mmk_s2s_generate_classes")

    def sync_status(self):
        """wait for the stream and raise if a wait inside the resident bi-LSTM kernel timed out"""
        check(self._lib.mmk_s2s_sync_status(self.handle, stream_ptr(self.device)), "mmk_s2s_sync_status")

    def inject_sync_error(self):
        """fault injection for tests: the next ``sync_status`` reports a timed-out wait (include/mmk.h)"""
        check(self._lib.mmk_s2s_inject_sync_error(self.handle, stream_ptr(self.device)), "mmk_s2s_inject_sync_error")

    def resident_launches(self) -> int:
        """bi-LSTM layers run as one resident launch so far (diagnostic, see include/mmk.h)"""
        return int(self._lib.mmk_s2s_resident_launches(self.handle))

    def last_logits(self, batch: int) -> torch.Tensor:
        """the MLP head's raw outputs of the last step, (batch, hop, out_dim + learn_temp)"""
        n = self.cfg.out_dim + (1 if self.cfg.learn_temp else 0)
        out = torch.empty((batch, self.cfg.hop, n), dtype=torch.float32, device=self.device)
        check(self._lib.mmk_s2s_last_logits(self.handle, batch, ptr(out), n, stream_ptr(self.device)), "mmk_s2s_last_logits")
        return out
