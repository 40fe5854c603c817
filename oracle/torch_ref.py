"""CPU restatement of the reference's generate-path algorithms (TEST INFRASTRUCTURE ONLY).

This file is the ORACLE of the repo: plain fp32 PyTorch/numpy on the host, no
dependence on the product package, each function citing the reference lines it
restates (paths relative to the reference root, ``mimikit/...``).  It executes the
reference's ALGORITHM literally -- in particular the naive WaveNet generation that
re-runs the whole rf-long window for every sample (networks/wavenet_v2.py:276-293,
:447-452) and the Python ``for t`` loop of loops/generate.py:207-219 -- so it is
also what ``bench.py`` times as the reference-algorithm CPU baseline.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against golden
vectors produced by the reference's own code (imported in the build container
through ``oracle/ref_shim.py``; generator: ``tests/golden/make_golden.py``).

Exception: ``griffin_lim`` is PARITY UNPINNED (torchaudio, which the reference calls, is not installed here; see its
docstring) -- it restates torchaudio 2.0.1's published algorithm over torch.stft / torch.istft.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.  All functions take a ``state_dict``-like mapping with the
reference's parameter names.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ---------------------------------------------------------------------------
# features
# ---------------------------------------------------------------------------
def mulaw_compress(x: torch.Tensor, q_levels: int = 256, compression: float = 1.) -> torch.Tensor:
    """MuLawCompress.torch_func, features/functionals.py:330-338"""
    mu = torch.tensor(q_levels - 1.0, dtype=torch.float32)
    c = torch.tensor(compression, dtype=torch.float32)
    x = x.to(torch.float32)
    y = torch.sign(x) * torch.log1p(mu * torch.abs(x) * c) / torch.log1p(mu * c)
    return ((y + 1) / 2 * mu + 0.5).to(torch.int64)


def mulaw_expand(codes: torch.Tensor, q_levels: int = 256, compression: float = 1.) -> torch.Tensor:
    """MuLawExpand.torch_func, features/functionals.py:361-369"""
    mu = torch.tensor(q_levels - 1.0, dtype=torch.float32)
    c = torch.tensor(compression, dtype=torch.float32)
    x = (codes.to(torch.float32) / mu) * 2 - 1.0
    return torch.sign(x) * (torch.exp(torch.abs(x) * torch.log1p(mu * c)) - 1.0) / (mu * c)


def stft_fixed_length(n_samples: int, n_fft: int, hop: int, center: bool) -> int:
    """STFT._fix_length target (features/functionals.py:468-486 with features/item_spec.py:58-112)"""
    extra = 0 if center else n_fft - hop
    n_frames = int((n_samples - extra) // hop) + int(center)
    return int((n_frames - int(center)) * hop) + extra


def magspec(x: torch.Tensor, n_fft: int, hop: int, center: bool = False, alignment: Optional[str] = "end") -> torch.Tensor:
    """MagSpec.torch_func == STFT(coordinate='mag').torch_func, features/functionals.py:507-524:
    length fix-up, torch.stft with a periodic Hann window, (.., frames, bins), abs"""
    if alignment is not None:
        keep = stft_fixed_length(x.shape[-1], n_fft, hop, center)
        x = x[..., -keep:] if alignment == "end" else x[..., :keep]
    s = torch.stft(x, n_fft, hop_length=hop, return_complex=True, center=center,
                   window=torch.hann_window(n_fft, device=x.device), pad_mode="constant")
    return s.transpose(-1, -2).contiguous().abs()


def stft_coord(x: torch.Tensor, n_fft: int, hop: int, coordinate: str = "pol", center: bool = True,
               pad_mode: str = "constant", alignment: Optional[str] = "end") -> torch.Tensor:
    """STFT.torch_func, features/functionals.py:506-523, for every coordinate it knows"""
    if alignment is not None:
        keep = stft_fixed_length(x.shape[-1], n_fft, hop, center)
        x = x[..., -keep:] if alignment == "end" else x[..., :keep]
    s = torch.stft(x, n_fft, hop_length=hop, return_complex=True, center=center,
                   window=torch.hann_window(n_fft, device=x.device), pad_mode=pad_mode)
    s = s.transpose(-1, -2).contiguous()
    if coordinate == "pol":
        return torch.stack((s.abs(), torch.angle(s)), dim=-1)
    if coordinate == "car":
        return torch.stack((s.real, s.imag), dim=-1)
    if coordinate == "mag":
        return s.abs()
    if coordinate == "angle":
        return torch.angle(s)
    return s


def istft(spec: torch.Tensor, n_fft: int, hop: int, coordinate: str = "pol") -> torch.Tensor:
    """ISTFT.torch_func, features/functionals.py:553-564 (spec: (.., frames, bins, 2)); the 'car' branch multiplies
    the two planes exactly as the reference does (:558)"""
    if coordinate == "pol":
        z = spec[..., 0] * torch.exp(1j * spec[..., 1])
    elif coordinate == "car":
        z = spec[..., 0] * (1j * spec[..., 1])
    else:
        z = spec
    return torch.istft(z.transpose(1, 2).contiguous(), n_fft=n_fft, hop_length=hop,
                       window=torch.hann_window(n_fft, device=spec.device))


def griffin_lim(mag: torch.Tensor, n_fft: int, hop: int, n_iter: int = 32, momentum: float = 0.99,
                init: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GLA.torch_func, features/functionals.py:634-642 = torchaudio.transforms.GriffinLim(n_fft, hop_length, power=1.)

    PARITY UNPINNED: torchaudio (pinned to 2.0.1 by the reference's pyproject.toml:63-68) is not installed in the build
    container, so no golden vector of the reference's own GLA exists.  This restates torchaudio 2.0.1's published
    ``functional.griffinlim`` on top of torch.stft / torch.istft (which ARE the reference's kernels): power = 1,
    window = periodic Hann, win_length = n_fft, length = None; ``init`` replaces its ``torch.rand`` draw (complex dtype,
    both parts uniform in [0, 1)), ``None`` is rand_init=False.  mag: (batch, frames, bins), as the functional gets it."""
    assert 0 <= momentum < 1
    m = momentum / (1 + momentum)
    spec = mag.transpose(-1, -2).contiguous()                        # torchaudio layout (batch, freq, time)
    window = torch.hann_window(n_fft, device=mag.device)
    if init is None:
        angles = torch.full(spec.shape, 1, dtype=torch.complex64, device=mag.device)
    else:
        angles = init.transpose(-1, -2).contiguous().to(torch.complex64)
    tprev = torch.tensor(0., dtype=spec.dtype, device=mag.device)
    for _ in range(n_iter):
        inverse = torch.istft(spec * angles, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=window, length=None)
        rebuilt = torch.stft(inverse, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=window, center=True,
                             pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        angles = rebuilt
        if m:
            angles = angles - tprev * m
        angles = angles / (angles.abs() + 1e-16)
        tprev = rebuilt
    return torch.istft(spec * angles, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=window, length=None)


def resample(x: torch.Tensor, orig_sr: int, target_sr: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """Resample.torch_func, features/functionals.py:305-306 = torchaudio.functional.resample(x, orig_sr, target_sr).

    PARITY UNPINNED: torchaudio (pinned to 2.0.1 by the reference) is not installed in the build container.  This restates
    torchaudio 2.0.1's published ``_get_sinc_resample_kernel`` / ``_apply_sinc_resample_kernel`` (sinc_interp_hann):
    the filter bank in float64, then a strided conv1d over the zero-padded signal, cut to ceil(new * T / orig) samples."""
    import math
    if int(orig_sr) == int(target_sr):
        return x
    g = math.gcd(int(orig_sr), int(target_sr))
    orig, new = int(orig_sr) // g, int(target_sr) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * (base_freq / orig)
    kernels = kernels.to(x.dtype)
    shape = x.shape
    w = x.reshape(-1, shape[-1])
    length = w.shape[-1]
    w = F.pad(w, (width, width + orig))
    res = F.conv1d(w[:, None], kernels, stride=orig)
    res = res.transpose(1, 2).reshape(w.shape[0], -1)
    target_length = int(math.ceil(new * length / orig))
    return res[..., :target_length].reshape(*shape[:-1], target_length)


# ---------------------------------------------------------------------------
# head: MLP with learned temperature + categorical sampler
# ---------------------------------------------------------------------------
# the members of mimikit's ActivationEnum (modules/activations.py:25-39) that are plain element-wise functions: nn.<name>() of torch, Abs / Sin / Cos of
# the reference's own three-line modules (:66-78)
ACTIVATIONS = {"Tanh": torch.tanh, "Sigmoid": torch.sigmoid, "Mish": F.mish, "ReLU": torch.relu, "Softplus": F.softplus, "Identity": lambda v: v,
               "Abs": torch.abs, "Sin": torch.sin, "Cos": torch.cos}


def mlp_raw(sd: SD, prefix: str, x: torch.Tensor, n_hidden: int = 0, act: str = "Mish", n_dropouts: int = 0) -> torch.Tensor:
    """MLP.fc (networks/mlp.py:42-53): Linear, act, [Linear, act]*n, Linear -> (.., q+1) raw outputs.  ``act``: MLPIO.activation by its ActivationEnum name
    (modules/io.py:205: Mish); ``n_dropouts``: Dropout / Dropout1d modules behind every activation (:36-40: identities in eval mode, but they shift the
    Linears' indices in the Sequential - and with them the state_dict keys)"""
    f = ACTIVATIONS[act]
    step = 2 + n_dropouts
    h = f(F.linear(x, sd[prefix + "fc.0.weight"], sd[prefix + "fc.0.bias"]))
    for i in range(n_hidden):
        k = prefix + f"fc.{step * (i + 1)}."
        h = f(F.linear(h, sd[k + "weight"], sd[k + "bias"]))
    k = prefix + f"fc.{step * (n_hidden + 1)}."
    return F.linear(h, sd[k + "weight"], sd[k + "bias"])


def mlp_logits(raw: torch.Tensor, min_temp: Optional[float] = 1e-4) -> torch.Tensor:
    """MLP.forward tail (networks/mlp.py:58-63)"""
    if min_temp is None:
        return raw
    temp = torch.sigmoid(raw[..., -1:])
    return raw[..., :-1] / torch.maximum(temp, torch.tensor(min_temp))


def categorical(logits: torch.Tensor, temperature=None, uniforms: Optional[torch.Tensor] = None) -> torch.Tensor:
    """CategoricalSampler.forward (modules/targets.py:37-52).  argmax when temperature is None.
    With a temperature the reference draws from torch.multinomial, whose RNG stream cannot be
    reproduced elsewhere; the oracle instead inverts the CDF of the SAME distribution
    (softmax(logits / T)) at caller-supplied uniforms, which is what the device kernel does."""
    if temperature is None:
        return logits.argmax(dim=-1)
    t = torch.as_tensor(temperature, dtype=torch.float32).reshape(-1, *([1] * (logits.dim() - 1)))
    l = logits / t
    l = l - l.max(dim=-1, keepdim=True).values
    e = torch.exp(l)
    cdf = torch.cumsum(e, dim=-1)
    target = uniforms.reshape(*logits.shape[:-1], 1).to(torch.float32) * cdf[..., -1:]
    hit = (cdf > target) & (e > 0)
    first = torch.where(hit.any(-1), hit.float().argmax(-1), (e > 0).float().cumsum(-1).argmax(-1))
    return first


# ---------------------------------------------------------------------------
# WaveNet
# ---------------------------------------------------------------------------
def wavenet_dilations(kernel_sizes: Sequence[int], blocks: Sequence[int]) -> Tuple[List[int], List[int]]:
    """WaveNet.get_kernels_and_dilation for the `one kernel size, n blocks` form (wavenet_v2.py:319-323)"""
    assert len(kernel_sizes) == 1
    k = kernel_sizes[0]
    return [k] * sum(blocks), [k ** i for b in blocks for i in range(b)]


def wavenet_rf(kernels: Sequence[int], dilations: Sequence[int]) -> int:
    """WaveNet.rf, wavenet_v2.py:337-339"""
    return sum((k - 1) * d for k, d in zip(kernels, dilations)) + 1


def wavenet_window_forward(sd: SD, inputs: Tuple[torch.Tensor, ...], kernels: Sequence[int], dilations: Sequence[int],
                           n_cond: int = 0, has_skips: bool = True, residuals: bool = True,
                           n_mlp_hidden: int = 0, embedding: bool = True, groups: int = 1, head: str = "mlp",
                           gated: bool = True, layerwise_inputs: bool = False,
                           res_layers: Optional[Sequence[bool]] = None, affine: bool = False,
                           cond_classes: Optional[Sequence[int]] = None, heads_n_hidden: Optional[Sequence[int]] = None,
                           act_f: str = "Tanh", act_g: str = "Sigmoid", mlp_act: str = "Mish", mlp_dropouts: int = 0):
    """Full-window eval forward (wavenet_v2.py:276-293 with WNLayer.forward :131-176, pad_side=0):
    returns the RAW head outputs (B, 1, q+1) of the FIRST computable position (eval_slice, :273).
    ``groups`` applies to the dilated convolutions only (:93); ``head`` "linear" / "linear_abs" is the
    (Chunked)LinearIO output module of a magnitude-frame target (io_spec.py:238-243) instead of the MLP.
    ``gated=False`` is act_g=None (:155-163: one tanh, plain Conv1d modules); ``layerwise_inputs`` adds the embedded
    input 0 to every layer's output (:285-286); ``res_layers`` says per layer, in RUN order, whether it has its conv_res
    (reverse_layer_order, :253, moves the layer built without one, :216, to the front) - default: all but the last;
    ``affine`` is with_affine_residuals (:122, :148-149, :157-161; ParametrizedLinear, parametrized.py:34-47): the layer's
    input goes through x_hat * a + b of a 1x1 convolution to three times its width first - the dilated convolution AND the
    residual sum see the transformed input - and, without gated units, every conditioning input c becomes aff(c) + c.
    ``cond_classes[j]`` > 0: conditioning input j is a stream of class indices through an EmbeddingIO (from_config :231-234 builds
    the module of EVERY input from its spec); ``heads_n_hidden`` (one entry per target): the network has that many output modules
    (:240-243) and the function returns the tuple of their raw outputs (:293).  ``act_f`` / ``act_g``: Config.act_f / act_g by their
    ActivationEnum names (modules/activations.py:25-39; WNLayer.forward :151 / :163 applies whatever modules they name)."""
    f_act, g_act = ACTIVATIONS[act_f], ACTIVATIONS[act_g]
    if embedding:
        h = F.embedding(inputs[0], sd["input_modules.0.0.weight"])
    else:
        h = F.linear(inputs[0], sd["input_modules.0.0.weight"], sd["input_modules.0.0.bias"])
    h = h.transpose(1, 2).contiguous()
    x0 = h
    conds = [(F.embedding(inputs[1 + j], sd[f"input_modules.{1 + j}.0.weight"]) if cond_classes and cond_classes[j] > 0 else
              F.linear(inputs[1 + j], sd[f"input_modules.{1 + j}.0.weight"], sd[f"input_modules.{1 + j}.0.bias"]))
             .transpose(1, 2).contiguous() for j in range(n_cond)]
    skips = None
    n_layers = len(kernels)
    if res_layers is None:
        res_layers = [residuals and l != n_layers - 1 for l in range(n_layers)]   # last layer is built without residuals (:216)
    dil_key = "conv_dil.0.0." if gated else "conv_dil.0."
    for l, (k, d) in enumerate(zip(kernels, dilations)):
        p = f"layers.{l}."
        cause = (k - 1) * d

        def aff(v, p=p):
            x_hat, a, b = torch.chunk(F.conv1d(v, sd[p + "aff_res.params.weight"], sd.get(p + "aff_res.params.bias")), 3, dim=1)
            return x_hat * a + b

        if affine:
            h = aff(h)
        z = F.conv1d(h, sd[p + dil_key + "weight"], sd.get(p + dil_key + "bias"), dilation=d, groups=groups)
        cond_sum = 0          # the conditioning features are summed first (:141-147), then added to the dilated product
        for j in range(n_cond):
            ck = p + (f"conv_1x1.{j}.0." if gated else f"conv_1x1.{j}.")
            cj = conds[j][:, :, cause:]
            if affine and not gated:
                cj = aff(cj) + cj
            cond_sum = cond_sum + F.conv1d(cj, sd[ck + "weight"], sd.get(ck + "bias"))
        z = z + cond_sum
        if gated:
            z_f, z_g = torch.chunk(z, 2, dim=1)
            y = f_act(z_f) * g_act(z_g)
        else:
            y = f_act(z)
        if has_skips:
            s = F.conv1d(y, sd[p + "conv_skip.weight"], sd.get(p + "conv_skip.bias"))
            skips = s if skips is None else s + skips[:, :, cause:]
        if res_layers[l]:
            h = h[:, :, cause:] + F.conv1d(y, sd[p + "conv_res.weight"], sd.get(p + "conv_res.bias"))
        else:
            h = y
        if layerwise_inputs:
            h = h + x0[..., -h.size(-1):]
        conds = [c[:, :, cause:] for c in conds]
    y = (skips if has_skips else h).transpose(1, 2).contiguous()[:, 0:1]
    if heads_n_hidden is not None:
        return tuple(mlp_raw(sd, f"output_modules.{k}.estimator.0.", y, n, mlp_act, mlp_dropouts) for k, n in enumerate(heads_n_hidden))
    if head == "mlp":
        return mlp_raw(sd, "output_modules.0.estimator.0.", y, n_mlp_hidden, mlp_act, mlp_dropouts)
    out = F.linear(y, sd["output_modules.0.0.weight"], sd["output_modules.0.0.bias"])
    return out.abs() if head == "linear_abs" else out


def wavenet_generate_frames(sd: SD, prompt: torch.Tensor, n_steps: int, kernels, dilations, **arch) -> torch.Tensor:
    """the same loop (loops/generate.py:195-219) for a network whose input and target are real-valued frames
    (magspec_io): every step's (B, 1, bins) output is written back as the next input frame"""
    rf = wavenet_rf(kernels, dilations)
    prior = prompt.size(1)
    x = torch.cat([prompt, torch.zeros(prompt.size(0), n_steps, prompt.size(2))], dim=1)
    for t in range(prior, prior + n_steps):
        x[:, t:t + 1] = wavenet_window_forward(sd, (x[:, t - rf:t],), kernels, dilations, embedding=False, **arch)
    return x


def wavenet_generate(sd: SD, prompt: torch.Tensor, cond: Sequence[torch.Tensor], n_steps: int, kernels, dilations,
                     min_temp: Optional[float] = 1e-4, temperature=None, uniforms=None, keep_logits: bool = False,
                     forced: Optional[torch.Tensor] = None, **arch):
    """GenerateLoopV2.run's hot loop (loops/generate.py:195-219) around WaveNet.generate_step:
    prompt + blanks, one full-window forward per step, in-place write.
    ``forced`` (batch, prior + n_steps): teacher forcing for step-by-step checks of another implementation's output - every
    step sees THAT history instead of the oracle's own picks (the returned indices are still the oracle's pick per step)."""
    rf = wavenet_rf(kernels, dilations)
    prior = prompt.size(1)
    idx = torch.cat([prompt, torch.zeros(prompt.size(0), n_steps, dtype=prompt.dtype)], dim=1)
    hist = idx if forced is None else forced
    logits_log = []
    for s, t in enumerate(range(prior, prior + n_steps)):
        window = (hist[:, t - rf:t], *[c[:, t - rf:t] for c in cond])
        raw = wavenet_window_forward(sd, window, kernels, dilations, n_cond=len(cond), **arch)
        logits = mlp_logits(raw, min_temp)
        u = None if uniforms is None else uniforms[:, s]
        idx[:, t:t + 1] = categorical(logits, temperature, u)
        if keep_logits:
            logits_log.append(raw[:, 0])
    return (idx, torch.stack(logits_log, 1)) if keep_logits else idx


def wavenet_generate_streams(sd: SD, prompts: Sequence[torch.Tensor], n_steps: int, kernels, dilations, heads_n_hidden: Sequence[int],
                             min_temps: Optional[Sequence[Optional[float]]] = None, temperature=None, uniforms=None,
                             keep_logits: bool = False, forced: Optional[Sequence[torch.Tensor]] = None, **arch):
    """the same loop for a network of several targets: output k of every step is written into input k (loops/generate.py:213-218,
    ``zip(tensors, outputs)``).  ``prompts``: one tensor per input - the first len(heads_n_hidden) are class streams of ``prior`` steps,
    the others (plain conditioning features, if any) cover prior + n_steps.  ``uniforms``: (targets, batch, n_steps).  Returns the tuple
    of the streams (and the tuple of per-target raw outputs with ``keep_logits``)."""
    rf = wavenet_rf(kernels, dilations)
    n_tgt = len(heads_n_hidden)
    prior = prompts[0].size(1)
    min_temps = [1e-4] * n_tgt if min_temps is None else list(min_temps)
    streams = [torch.cat([p, torch.zeros(p.size(0), n_steps, dtype=p.dtype)], dim=1) if k < n_tgt else p for k, p in enumerate(prompts)]
    hist = streams if forced is None else [forced[k] if k < n_tgt else streams[k] for k in range(len(streams))]
    logs = [[] for _ in range(n_tgt)]
    for s, t in enumerate(range(prior, prior + n_steps)):
        raws = wavenet_window_forward(sd, tuple(x[:, t - rf:t] for x in hist), kernels, dilations, n_cond=len(prompts) - 1,
                                      heads_n_hidden=heads_n_hidden, **arch)
        for k, raw in enumerate(raws):
            u = None if uniforms is None else uniforms[k][:, s]
            streams[k][:, t:t + 1] = categorical(mlp_logits(raw, min_temps[k]), temperature, u)
            if keep_logits:
                logs[k].append(raw[:, 0])
    out = tuple(streams[:n_tgt])
    return (out, tuple(torch.stack(l, 1) for l in logs)) if keep_logits else out


# ---------------------------------------------------------------------------
# SampleRNN
# ---------------------------------------------------------------------------
def _linearize(q: torch.Tensor, class_size: int) -> torch.Tensor:
    """Linearizer, modules/io.py:106-112"""
    return ((q.float() / class_size) - .5) * 2


def _rnn_cell(kind: str, sd: SD, p: str, x: torch.Tensor, state, layer: int = 0):
    """one time step of one layer of nn.LSTM / nn.GRU / nn.RNN (gate orders i,f,g,o and r,z,n)"""
    w_ih, w_hh = sd[p + f"weight_ih_l{layer}"], sd[p + f"weight_hh_l{layer}"]
    b_ih, b_hh = sd.get(p + f"bias_ih_l{layer}"), sd.get(p + f"bias_hh_l{layer}")
    if kind == "lstm":
        h, c = state
        g = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        return h, (h, c)
    if kind == "gru":
        h = state
        gi, gh = F.linear(x, w_ih, b_ih), F.linear(h, w_hh, b_hh)
        i_r, i_z, i_n = gi.chunk(3, dim=-1)
        h_r, h_z, h_n = gh.chunk(3, dim=-1)
        r, z = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (h - n) * z + n
        return h, h
    h = torch.tanh(F.linear(x, w_ih, b_ih) + F.linear(state, w_hh, b_hh))
    return h, h


def fold_weight_norm(sd: SD) -> SD:
    """state_dict of a network built with weight_norm=True (sample_rnn_v2.py:67-81): every ``name_g`` / ``name_v`` pair
    becomes ``name`` = g v / |v| with the norm over all dimensions but the first (torch.nn.utils.weight_norm, dim=0)"""
    out = {}
    for key, value in sd.items():
        if key.endswith("_v") and key[:-2] + "_g" in sd:
            g = sd[key[:-2] + "_g"]
            norm = value.reshape(value.shape[0], -1).norm(dim=1).reshape(g.shape) if value.dim() > 1 else value.abs()
            out[key[:-2]] = value * (g / norm)
        elif not (key.endswith("_g") and key[:-2] + "_v" in sd):
            out[key] = value
    return out


class SampleRNNOracle:
    """SampleRNN.before_generate / generate_step / after_generate (sample_rnn_v2.py:226-268) with
    SampleRNNTier.forward (:83-99) for n_rnn = 1, restated over a state_dict."""

    def __init__(self, sd: SD, frame_sizes: Sequence[int], hidden_dim: int, rnn_class: str = "lstm",
                 q_levels: int = 256, n_mlp_hidden: int = 0, min_temp: Optional[float] = 1e-4, h0: str = "zeros",
                 n_rnn: int = 1, in_classes: Optional[Sequence[int]] = None, inputs_mode: str = "sum",
                 heads: Optional[Sequence[dict]] = None, mlp_act: str = "Mish", mlp_dropouts: int = 0):
        """``mlp_act`` / ``mlp_dropouts``: MLPIO.activation and the number of dropout modules behind it (see mlp_raw).
        ``in_classes`` (one class size per input) / ``inputs_mode``: a network of several inputs - every tier's input module is
        a ZipReduceVariables over one framed linear per input (from_config :141-145, :160-173; modules/io.py:289-313);
        ``heads`` (one dict(n_mlp_hidden=, min_temp=) per target): several output modules on the bottom tier's vector (:181-182, :259).
        With either, windows / prompts / results are tuples of streams and output k goes into input k (loops/generate.py:213-218)."""
        self.in_classes = None if in_classes is None else tuple(in_classes)
        self.mlp_act, self.mlp_dropouts = mlp_act, mlp_dropouts
        self.inputs_mode = inputs_mode
        self.heads = None if heads is None else [dict(h) for h in heads]
        self.sd, self.fs, self.H, self.kind = sd, tuple(frame_sizes), hidden_dim, rnn_class
        self.n_rnn = n_rnn            # stacked layers per tier (nn.LSTM / GRU num_layers, sample_rnn_v2.py:65)
        self.q, self.n_mlp_hidden, self.min_temp, self.h0 = q_levels, n_mlp_hidden, min_temp, h0
        self.hidden = [None] * (len(self.fs) - 1)
        self.outputs = [None] * (len(self.fs) - 1)
        self.prompt_length = 0
        self.last_raw = None

    @property
    def rf(self):
        return self.fs[0]

    def reset_hidden(self):
        self.hidden = [None] * (len(self.fs) - 1)

    def _zip(self, i: int, frames, leaf: str) -> torch.Tensor:
        """ZipReduceVariables.forward (modules/io.py:304-313): y = head_0(x_0) w_0, then y += head_m(x_m) w_m"""
        sd, p = self.sd, f"tiers.{i}.input_module."
        frames = (frames,) if isinstance(frames, torch.Tensor) else tuple(frames)
        classes = self.in_classes or (self.q,) * len(frames)
        M = len(frames)
        if self.inputs_mode == "sum":
            w = torch.ones(M)
        elif self.inputs_mode == "mean":
            w = torch.ones(M) / M
        else:
            w = torch.softmax(sd[p + "weights"], dim=0)

        def head(m):
            wt = sd[p + f"heads.{m}.{leaf}weight"]
            return F.linear(_linearize(frames[m], classes[m]), wt.reshape(self.H, -1), sd[p + f"heads.{m}.{leaf}bias"])

        if M == 1 and self.in_classes is None:
            return head(0)            # (one input: every mode weights it by 1)
        y = head(0) * w[0]
        for m in range(1, M):
            y = y + head(m) * w[m]
        return y

    def _tier(self, i: int, frames, upper: Optional[torch.Tensor]) -> torch.Tensor:
        sd, p = self.sd, f"tiers.{i}."
        x = self._zip(i, frames, "2.")
        if upper is not None:
            x = x + upper
        if self.hidden[i] is None:
            init = getattr(torch, self.h0)
            self.hidden[i] = [(init(x.size(0), self.H), init(x.size(0), self.H)) if self.kind == "lstm" else init(x.size(0), self.H)
                              for _ in range(self.n_rnn)]
        for k in range(self.n_rnn):      # layer k's input is layer k-1's output at this time step
            x, self.hidden[i][k] = _rnn_cell(self.kind, sd, p + "rnn.", x, self.hidden[i][k], layer=k)
        up = self.fs[i] // (self.fs[i + 1] if i < len(self.fs) - 2 else 1)
        out = F.linear(x, sd[p + "up_sampler.fc.weight"], sd[p + "up_sampler.fc.bias"])
        return out.reshape(x.size(0), up, self.H)

    def generate_step(self, window, t: int, temperature=None, uniforms=None):
        fs, n = self.fs, len(self.fs)
        multi = not isinstance(window, torch.Tensor)
        streams = tuple(window) if multi else (window,)
        for i in range(n - 1):
            if t % fs[i] == 0:
                upper = None if i == 0 else self.outputs[i - 1][:, (t // fs[i]) % (fs[i - 1] // fs[i])]
                self.outputs[i] = self._tier(i, tuple(x[:, -fs[i]:] for x in streams), upper)
        if t < self.prompt_length:
            return None
        sd = self.sd
        x = self._zip(n - 1, tuple(x[:, -fs[-1]:] for x in streams), "2.2.cv.")
        x = x + self.outputs[-1][:, (t % fs[-2]) - fs[-2]]
        if self.heads is not None:
            raws = tuple(mlp_raw(sd, f"output_modules.{k}.estimator.0.", x, h.get("n_mlp_hidden", 0), self.mlp_act, self.mlp_dropouts) for k, h in enumerate(self.heads))
            self.last_raw = raws
            return tuple(categorical(mlp_logits(raw, h.get("min_temp", 1e-4)), temperature, None if uniforms is None else uniforms[k])
                         for k, (raw, h) in enumerate(zip(raws, self.heads)))
        raw = mlp_raw(sd, "output_modules.0.estimator.0.", x, self.n_mlp_hidden, self.mlp_act, self.mlp_dropouts)
        self.last_raw = raw
        return categorical(mlp_logits(raw, self.min_temp), temperature, uniforms)

    def before_generate(self, prompt):
        self.outputs = [None] * (len(self.fs) - 1)
        self.reset_hidden()
        multi = not isinstance(prompt, torch.Tensor)
        length = (prompt[0] if multi else prompt).size(1)
        offset = length % self.rf
        self.prompt_length = length - offset
        for t in range(self.rf, self.prompt_length):
            lo, hi = t + offset - self.rf, t + offset
            self.generate_step(tuple(p[:, lo:hi] for p in prompt) if multi else prompt[:, lo:hi], t)

    def generate(self, prompt: torch.Tensor, n_steps: int, temperature=None, uniforms=None, keep_logits=False, forced=None):
        """``forced`` (batch, prior + n_steps): teacher forcing - every step sees that history instead of the oracle's own
        picks (used to check another implementation's output step by step); the returned indices are the oracle's picks"""
        if not isinstance(prompt, torch.Tensor):
            return self._generate_streams(tuple(prompt), n_steps, temperature, uniforms, keep_logits, forced)
        self.before_generate(prompt)
        prior = prompt.size(1)
        idx = torch.cat([prompt, torch.zeros(prompt.size(0), n_steps, dtype=prompt.dtype)], dim=1)
        hist = idx if forced is None else forced
        logs = []
        for s, t in enumerate(range(prior, prior + n_steps)):
            u = None if uniforms is None else uniforms[:, s]
            idx[:, t] = self.generate_step(hist[:, t - self.rf:t], t, temperature, u)
            if keep_logits:
                logs.append(self.last_raw)
        self.reset_hidden()
        return (idx, torch.stack(logs, 1)) if keep_logits else idx


    def _generate_streams(self, prompts, n_steps, temperature, uniforms, keep_logits, forced):
        """several inputs / targets: one stream per input, output k written into stream k; ``uniforms`` (targets, batch, n_steps)"""
        self.before_generate(prompts)
        prior = prompts[0].size(1)
        streams = [torch.cat([p, torch.zeros(p.size(0), n_steps, dtype=p.dtype)], dim=1) for p in prompts]
        hist = streams if forced is None else list(forced)
        n_tgt = len(self.heads) if self.heads is not None else 1
        logs = [[] for _ in range(n_tgt)]
        for s, t in enumerate(range(prior, prior + n_steps)):
            u = None if uniforms is None else [uniforms[k][:, s] for k in range(n_tgt)]
            outs = self.generate_step(tuple(x[:, t - self.rf:t] for x in hist), t, temperature, u if self.heads is not None else (None if u is None else u[0]))
            outs = outs if isinstance(outs, tuple) else (outs,)
            raws = self.last_raw if isinstance(self.last_raw, tuple) else (self.last_raw,)
            for k, o in enumerate(outs[:len(streams)]):
                streams[k][:, t] = o
                if keep_logits:
                    logs[k].append(raws[k])
        self.reset_hidden()
        out = tuple(streams)
        return (out, tuple(torch.stack(l, 1) for l in logs)) if keep_logits else out


# ---------------------------------------------------------------------------
# Seq2Seq
# ---------------------------------------------------------------------------
def _bilstm(sd: SD, p: str, x: torch.Tensor, state=None):
    """bidirectional single-layer nn.LSTM, batch_first; returns (B, T, 2D), (h_n, c_n) each (2, B, D)"""
    B, T, _ = x.shape
    D = sd[p + "weight_hh_l0"].shape[1]
    outs, hs, cs = [], [], []
    for d, sfx in enumerate(("", "_reverse")):
        h = torch.zeros(B, D) if state is None else state[0][d]
        c = torch.zeros(B, D) if state is None else state[1][d]
        seq = range(T) if d == 0 else range(T - 1, -1, -1)
        ys = [None] * T
        for t in seq:
            g = F.linear(x[:, t], sd[p + "weight_ih_l0" + sfx], sd[p + "bias_ih_l0" + sfx]) + \
                F.linear(h, sd[p + "weight_hh_l0" + sfx], sd[p + "bias_hh_l0" + sfx])
            i, f, gg, o = g.chunk(4, dim=-1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            ys[t] = h
        outs.append(torch.stack(ys, 1))
        hs.append(h)
        cs.append(c)
    return torch.cat(outs, -1), (torch.stack(hs), torch.stack(cs))


def s2s_step(sd: SD, x: torch.Tensor, hop: int, out_abs: bool = True, downsampling: str = "edge_sum",
             upsampling: str = "linear_resample", enc_residuals: bool = False, dec_residuals: bool = False,
             min_temp: Optional[float] = 1e-4, return_raw: bool = False) -> torch.Tensor:
    """Seq2SeqLSTMNetwork.forward (s2s_lstm_v2.py:246-253): EncoderLSTM.forward with the edge_sum / edge_mean / sum / mean
    poolings and stacked layers (:93-113), DecoderLSTM.forward with linear_resample or repeat (:155-179); the number of
    layers is read off the state_dict; every decoder layer starts from the LAST encoder layer's final state (:171).

    Discrete IO (read off the state_dict too): class indices go through the nn.Embedding of `input_module` (:205-210, the
    ZipReduceVariables weight of a single input is 1), the MLP head's argmax comes back TIMES that weight, i.e. as a float
    tensor (modules/io.py:310, modules/targets.py:43-44; generate_step hands the sampler no temperature, :262-263).  With
    return_raw the head's outputs before the learned-temperature division are returned beside the classes."""
    D = sd["enc.fc_out.weight"].shape[0]
    if "input_module.heads.0.0.weight" in sd:
        x = F.embedding(x, sd["input_module.heads.0.0.weight"]) * torch.ones(())
    n_enc = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("enc.lstm."))
    n_dec = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("dec.lstm."))
    hidden = None
    for n in range(n_enc):
        y, hidden = _bilstm(sd, f"enc.lstm.{n}.", x)
        y = y.view(*y.shape[:-1], D, 2).sum(-1)
        x = x + y if (n > 0 and enc_residuals) else y
    if downsampling == "linear_resample":      # LinearResampler(D, 1 / hop, 1) (:105-106, modules/resamplers.py:13-23)
        y = F.linear(x, sd["enc.fc.fc.weight"], sd["enc.fc.fc.bias"]).reshape(x.size(0), 1, D)
    else:
        y = x.unfold(1, hop, hop)
        if "edge" in downsampling:
            y = y[..., [0, -1]]
        y = y.sum(-1) if "sum" in downsampling else y.mean(-1)
    coded = F.linear(y, sd["enc.fc_out.weight"])
    if upsampling == "linear_resample":
        z = F.linear(coded, sd["dec.fc.fc.weight"], sd["dec.fc.fc.bias"]).reshape(coded.size(0), hop, D)
    elif upsampling == "interp":               # (:162-165) nearest-neighbour spread of the two final states over the hop frames
        z = coded.expand(-1, hop, -1) + F.interpolate(hidden[0].permute(1, 2, 0), (hop,)).permute(0, 2, 1)
    else:
        z = coded.repeat_interleave(hop, 1)
    for n in range(n_dec):
        y, _ = _bilstm(sd, f"dec.lstm.{n}.", z, hidden)
        y = y.view(*y.shape[:-1], D, 2).sum(-1)
        z = z + y if dec_residuals else y
    mlp = "output_module.heads.0.estimator.0."
    if mlp + "fc.0.weight" in sd:
        n_hidden = max(int(k[len(mlp) + 3:].split(".")[0]) for k in sd if k.startswith(mlp + "fc.")) // 2 - 1
        raw = mlp_raw(sd, mlp, z, n_hidden)
        classes = categorical(mlp_logits(raw, min_temp)) * torch.ones(())
        return (classes, raw) if return_raw else classes
    out = F.linear(z, sd["output_module.heads.0.0.weight"], sd["output_module.heads.0.0.bias"])
    return out.abs() if out_abs else out


def s2s_generate(sd: SD, prompt: torch.Tensor, n_steps: int, hop: int, out_abs: bool = True, **arch) -> torch.Tensor:
    """the loop of loops/generate.py:207-219 for a net that returns hop frames (or hop classes) per call"""
    prior = prompt.size(1)
    frames = torch.cat([prompt, torch.zeros(prompt.size(0), n_steps, *prompt.shape[2:], dtype=prompt.dtype)], dim=1)
    until = 0
    for t in range(prior, prior + n_steps):
        if t < until:
            continue
        out = s2s_step(sd, frames[:, t - hop:t], hop, out_abs, **arch)
        n_out = min(out.size(1), frames.size(1) - t)
        frames[:, t:t + n_out] = out[:, :n_out]
        until = t + n_out
    return frames
