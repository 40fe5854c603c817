"""Generates the golden vectors under tests/golden/ FROM THE REFERENCE'S OWN CODE.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
The reference is imported unmodified through oracle/ref_shim.py (third-party
packages it needs but that are not installed are stubbed; see that file).  What is
committed are inputs and expected outputs only (.npz / .json) -- never reference
source.  Network weights come from the deterministic recipe in oracle/weights.py
(loaded into the reference's modules with load_state_dict), so fixtures hold only
prompts and outputs.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.ref_shim import load_reference  # noqa: E402
from oracle.weights import load_recipe  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
ref = load_reference()
torch.set_grad_enabled(False)
META = {"torch": torch.__version__, "numpy": np.__version__}


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def f32_neighbours(x: torch.Tensor, radius: int) -> torch.Tensor:
    bits = x.view(torch.int32).to(torch.int64)
    key = torch.where(bits >= 0, bits, -(bits & 0x7FFFFFFF))
    out = []
    for d in range(-radius, radius + 1):
        k = key + d
        b = torch.where(k >= 0, k, (-k) | 0x80000000)
        b = torch.where(b >= 2 ** 31, b - 2 ** 32, b)
        out.append(b.to(torch.int32).view(torch.float32))
    return torch.cat(out)


# ---------------------------------------------------------------------------
def make_mulaw():
    g = torch.Generator().manual_seed(1234)
    for tag, comp in (("c1", 1.0), ("c05", 0.5)):
        fwd = ref.functionals.MuLawCompress(256, comp)
        inv = ref.functionals.MuLawExpand(256, comp)
        # bin edges of the reference formula by bisection on the fp32 number line
        lo = torch.full((255,), -1.0)
        hi = torch.full((255,), 1.0)
        targets = torch.arange(1, 256)
        for _ in range(60):
            mid = ((lo.double() + hi.double()) / 2).float()
            ge = fwd(mid) >= targets
            hi = torch.where(ge, mid, hi)
            lo = torch.where(ge, lo, mid)
        near = f32_neighbours(hi, 6)
        near = near[(near >= -1) & (near <= 1)]
        x = torch.cat([
            torch.tensor([-1., 1., 0., -0., 1e-8, -1e-8, 1e-4, -1e-4, 0.5, -0.5, 0.999999, -0.999999]),
            torch.linspace(-1, 1, 4097),
            torch.rand(20000, generator=g) * 2 - 1,
            near,
            torch.tensor([1.5, -1.5, 3.0, -2.0]),        # out of range: no clamp in the reference
        ]).float()
        codes = fwd(x)
        all_codes = torch.arange(-2, 259)
        save(f"mulaw_{tag}.npz", x=x, codes=codes, n_near=np.int64(near.numel()),
             all_codes=all_codes, expanded=inv(all_codes), compression=np.float32(comp))


def make_stft():
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(2, 8192, generator=g)
    out = {"x": x}
    for n_fft, hop, center in ((1024, 256, False), (2048, 512, False), (2048, 512, True), (256, 64, False)):
        s = ref.functionals.MagSpec(n_fft, hop, center=center)(x)
        out[f"mag_{n_fft}_{hop}_{int(center)}"] = s
    # a length that is not a multiple of hop exercises _fix_length (alignment='end')
    y = torch.randn(3, 5000, generator=g)
    out["y"] = y
    out["mag_y_512_128_0"] = ref.functionals.MagSpec(512, 128, center=False)(y)
    save("stft.npz", **out)


def make_istft():
    """complex STFT coordinates and the ISTFT functional; GLA needs torchaudio, which is
    not installed here: no reference vector exists for it (oracle/torch_ref.py:griffin_lim is parity-unpinned)"""
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(2, 4096, generator=g) * 0.3
    out = {"x": x}
    for coord in ("pol", "car", "angle"):
        out[f"stft_{coord}_1024_256"] = ref.functionals.STFT(1024, 256, coord, center=True)(x)
    out["stft_car_1024_256_reflect"] = ref.functionals.STFT(1024, 256, "car", center=True, pad_mode="reflect")(x)
    out["stft_pol_1024_200_nc"] = ref.functionals.STFT(1024, 200, "pol", center=False)(x)
    # a random polar spectrum (not the STFT of any signal) through the reference's ISTFT
    spec = torch.stack((torch.rand(3, 11, 513, generator=g), (torch.rand(3, 11, 513, generator=g) * 2 - 1) * np.pi), dim=-1)
    out["spec_pol"] = spec
    out["istft_pol_1024_256"] = ref.functionals.ISTFT(1024, 256, "pol")(spec)
    out["istft_pol_1024_100"] = ref.functionals.ISTFT(1024, 100, "pol")(spec)
    out["istft_car_1024_256"] = ref.functionals.ISTFT(1024, 256, "car")(spec)
    # the round trip the reference's own tests lean on: STFT -> ISTFT
    out["roundtrip_1024_256"] = ref.functionals.ISTFT(1024, 256, "pol")(out["stft_pol_1024_256"])
    save("istft.npz", **out)


def capture_raw(net):
    """records the raw (pre-temperature) outputs of the MLP head at every call"""
    log = []
    mlp = net.output_modules[0].estimator[0]
    handle = mlp.fc.register_forward_hook(lambda m, i, o: log.append(o.detach().clone()))
    return log, handle


def run_loop(net, prompts, n_steps, parameters=None, inversed=False):
    cfg = ref.GenerateLoopV2.Config(parameters=parameters, yield_inversed_outputs=inversed,
                                    display_waveform=False, write_waveform=False)
    loop = ref.GenerateLoopV2(cfg, net, n_steps, dataloader=[[np.arange(prompts[0].size(0)), *prompts]], logger=None)
    outs = [o for o in loop.run()]
    torch.set_grad_enabled(False)   # the reference loop re-enables grad globally in teardown
    return outs[0]


def make_wavenet():
    g = torch.Generator().manual_seed(7)
    arrays = {}
    # (a) tiny unconditioned net, two blocks
    io = ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(input_module_type="embedding", mlp_dim=32))
    cfg = ref.WaveNet.Config(io_spec=io, blocks=(3, 2), dims_dilated=(16,), residuals_dim=16, skips_dim=16)
    net = ref.WaveNet.from_config(cfg).eval()
    load_recipe(net, seed=11, gain=2.0)
    rf = net.rf
    prompt = torch.randint(0, 256, (3, rf + 5), generator=g)
    log, h = capture_raw(net)
    out = run_loop(net, (prompt,), 24)
    h.remove()
    arrays.update(a_prompt=prompt, a_out=out[0], a_raw=torch.cat(log, 1), a_rf=np.int64(rf))
    # loop with yield_inversed_outputs=True returns MuLawExpand(out)
    arrays["a_inversed"] = run_loop(net, (prompt,), 24, inversed=True)[0]
    # single generate_step after before_generate (tests/test_wavenet.py:140-165 path)
    net.before_generate((prompt,), 0)
    arrays["a_step"] = net.generate_step((prompt[:, -rf:],), t=prompt.size(1))[0]

    # (b) conditioned net: one 1x1 input through LinearIO(12 -> 8), no skips, kernel size 3
    mag = ref.functionals.MagSpec(22, 4, center=False)
    ext = ref.extractor.Extractor("signal", ref.functionals.FileToSignal(16000))
    io_b = ref.IOSpec(
        inputs=(io.inputs[0], ref.io_spec.InputSpec("signal", mag, ref.io.LinearIO()).bind_to(ext)),
        targets=io.targets)
    cfg_b = ref.WaveNet.Config(io_spec=io_b, kernel_sizes=(3,), blocks=(3,), dims_dilated=(16,), dims_1x1=(8,),
                               residuals_dim=16, skips_dim=None)
    net_b = ref.WaveNet.from_config(cfg_b).eval()
    load_recipe(net_b, seed=12, gain=2.0)
    rf_b = net_b.rf
    n = 10
    idx = torch.randint(0, 256, (2, rf_b + n), generator=g)
    cond = torch.rand(2, rf_b + n, 12, generator=g)
    log, h = capture_raw(net_b)
    # teacher forced: one step per position, inputs taken from idx (not fed back)
    steps = [net_b.generate_step((idx[:, t - rf_b:t], cond[:, t - rf_b:t]), t=t)[0] for t in range(rf_b, rf_b + n)]
    h.remove()
    arrays.update(b_idx=idx, b_cond=cond, b_raw=torch.cat(log, 1), b_argmax=torch.cat(steps, 1), b_rf=np.int64(rf_b))

    # (c) the shape of BASELINE config 2 (10 layers x 64 ch), 12 free-running steps
    cfg_c = ref.WaveNet.Config(io_spec=ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(input_module_type="embedding")),
                               blocks=(10,), dims_dilated=(64,), residuals_dim=64, skips_dim=64)
    net_c = ref.WaveNet.from_config(cfg_c).eval()
    load_recipe(net_c, seed=13, gain=2.0)
    prompt_c = torch.randint(0, 256, (2, net_c.rf), generator=g)
    log, h = capture_raw(net_c)
    out_c = run_loop(net_c, (prompt_c,), 12)
    h.remove()
    arrays.update(c_prompt=prompt_c, c_out=out_c[0], c_raw=torch.cat(log, 1), c_rf=np.int64(net_c.rf))
    save("wavenet.npz", **arrays)


WAVENET_OPTIONS = {
    "mlp2": dict(io=dict(n_mlp_layers=2)),
    "mlp3_cond": dict(io=dict(n_mlp_layers=3), cond=True),
    "nogate": dict(act_g=None),
    "nogate_cond": dict(act_g=None, cond=True),
    "rev": dict(reverse_layer_order=True),
    "rev_noskip": dict(reverse_layer_order=True, skips_dim=None),
    "lw": dict(layerwise_inputs=True),
    "lw_noskip_rev": dict(layerwise_inputs=True, reverse_layer_order=True, skips_dim=None),
    "tied": dict(tie_io_weights=True),
    "k3": dict(kernel_sizes=(3,)),
    "k3_cond": dict(kernel_sizes=(3,), cond=True),
    "k4_noskip": dict(kernel_sizes=(4,), skips_dim=None),
    "aff": dict(with_affine_residuals=True),
    "aff_nogate_noskip": dict(with_affine_residuals=True, act_g=None, skips_dim=None),
}


def make_wavenet_options():
    """the remaining WaveNet options: deeper MLP heads (ONE hidden block repeated, mlp.py:46-49),
    act_g=None, reverse_layer_order, layerwise_inputs, tie_io_weights - 16 free-running steps through the reference's loop"""
    g = torch.Generator().manual_seed(29)
    arrays = {}
    for tag, kw in WAVENET_OPTIONS.items():
        kw = dict(kw)
        io = ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(input_module_type="embedding", mlp_dim=32, **kw.pop("io", {})))
        cond = kw.pop("cond", False)
        if cond:
            mag = ref.functionals.MagSpec(22, 4, center=False)
            ext = ref.extractor.Extractor("signal", ref.functionals.FileToSignal(16000))
            io = ref.IOSpec(inputs=(io.inputs[0], ref.io_spec.InputSpec("signal", mag, ref.io.LinearIO()).bind_to(ext)), targets=io.targets)
            kw["dims_1x1"] = (8,)
        kw.setdefault("skips_dim", 16)
        cfg = ref.WaveNet.Config(io_spec=io, blocks=(3, 2), dims_dilated=(16,), residuals_dim=16, **kw)
        net = ref.WaveNet.from_config(cfg).eval()
        load_recipe(net, seed=100 + len(tag), gain=2.0)
        rf = net.rf
        n = 16
        prompt = torch.randint(0, 256, (3, rf + 4), generator=g)
        prompts = (prompt,)
        if cond:
            c = torch.rand(3, rf + 4, 12, generator=g)
            prompts = (prompt, c)
            arrays[f"{tag}_cond"] = c
        log, h = capture_raw(net)
        out = run_loop(net, prompts, n)
        h.remove()
        arrays.update({f"{tag}_prompt": prompt, f"{tag}_out": out[0], f"{tag}_raw": torch.cat(log, 1)})
    save("wavenet_options.npz", **arrays)


WAVENET_ACTS = {
    "mish_tanh": dict(act_f="Mish", act_g="Tanh"),
    "relu_nogate": dict(act_f="ReLU", act_g=None),
    "sin_sig_cond": dict(act_f="Sin", act_g="Sigmoid", cond=True),
    "softplus_abs": dict(act_f="Softplus", act_g="Abs"),
    "id_cos_noskip": dict(act_f="Identity", act_g="Cos", skips_dim=None),
    "abs_nogate_cond": dict(act_f="Abs", act_g=None, cond=True),
}


def make_wavenet_acts():
    """Config.act_f / act_g other than Tanh / Sigmoid (wavenet_v2.py:198-199; WNLayer.forward :151, :163): 16 free-running steps through the
    reference's loop per case, as make_wavenet_options"""
    g = torch.Generator().manual_seed(31)
    arrays = {}
    for tag, kw in WAVENET_ACTS.items():
        kw = dict(kw)
        io = ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(input_module_type="embedding", mlp_dim=32))
        cond = kw.pop("cond", False)
        if cond:
            mag = ref.functionals.MagSpec(22, 4, center=False)
            ext = ref.extractor.Extractor("signal", ref.functionals.FileToSignal(16000))
            io = ref.IOSpec(inputs=(io.inputs[0], ref.io_spec.InputSpec("signal", mag, ref.io.LinearIO()).bind_to(ext)), targets=io.targets)
            kw["dims_1x1"] = (8,)
        kw.setdefault("skips_dim", 16)
        cfg = ref.WaveNet.Config(io_spec=io, blocks=(3, 2), dims_dilated=(16,), residuals_dim=16, **kw)
        net = ref.WaveNet.from_config(cfg).eval()
        load_recipe(net, seed=200 + len(tag), gain=1.5)
        rf = net.rf
        n = 16
        prompt = torch.randint(0, 256, (3, rf + 4), generator=g)
        prompts = (prompt,)
        if cond:
            c = torch.rand(3, rf + 4, 12, generator=g)
            prompts = (prompt, c)
            arrays[f"{tag}_cond"] = c
        log, h = capture_raw(net)
        out = run_loop(net, prompts, n)
        h.remove()
        arrays.update({f"{tag}_prompt": prompt, f"{tag}_out": out[0], f"{tag}_raw": torch.cat(log, 1)})
    save("wavenet_acts.npz", **arrays)


MLP_HEADS = {
    "wn_relu_dp": ("wavenet", dict(activation="ReLU", dropout=0.1)),
    "wn_tanh_2": ("wavenet", dict(activation="Tanh", n_hidden_layers=2)),
    "srnn_softplus_dp1d": ("srnn", dict(activation="Softplus", dropout1d=0.2)),
    "srnn_sigmoid": ("srnn", dict(activation="Sigmoid")),
}


def make_mlp_heads():
    """MLPIO.activation other than Mish and heads with Dropout / Dropout1d modules (modules/io.py:200-219, networks/mlp.py:36-53; identities in eval mode,
    where the loop runs a network - but they move the Linears' state_dict keys): a WaveNet and a SampleRNN through the reference's loop"""
    g = torch.Generator().manual_seed(37)
    arrays = {}
    for tag, (kind, head) in MLP_HEADS.items():
        head = dict(head)
        act = head.pop("activation")
        if kind == "wavenet":
            io = ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(input_module_type="embedding", mlp_dim=32))
        else:
            io = ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(mlp_dim=32))
        io.targets[0].module.activation = ref.io.ActivationConfig(act)      # (a user's IOSpec with MLPIO(activation=..., dropout=...) - `set` refuses fields that have a value)
        for k, v in head.items():
            setattr(io.targets[0].module, k, v)
        if kind == "wavenet":
            net = ref.WaveNet.from_config(ref.WaveNet.Config(io_spec=io, blocks=(3, 2), dims_dilated=(16,), residuals_dim=16, skips_dim=16)).eval()
            n, P = 16, None
        else:
            net = ref.SampleRNN.from_config(ref.SampleRNN.Config(io_spec=io, frame_sizes=(8, 2, 2), hidden_dim=32, rnn_class="gru")).eval()
            n, P = 24, 21
        load_recipe(net, seed=300 + len(tag), gain=2.0 if kind == "wavenet" else 8.0)
        prompt = torch.randint(0, 256, (3, (net.rf + 4) if P is None else P), generator=g)
        log, h = capture_raw(net)
        out = run_loop(net, (prompt,), n)
        h.remove()
        arrays.update({f"{tag}_prompt": prompt, f"{tag}_out": out[0], f"{tag}_raw": torch.cat(log, 1)})
    save("mlp_heads.npz", **arrays)


def make_freqnet():
    """WaveNet over magnitude frames (demos/freqnet.py:34-63 at reduced size): linear frame input and output, no residual
    and no skip path, grouped dilated convolutions"""
    g = torch.Generator().manual_seed(17)
    arrays = {}
    for tag, groups, act in (("g1", 1, "Identity"), ("g4", 4, "Identity"), ("g2abs", 2, "Abs")):
        io = ref.IOSpec.magspec_io(ref.IOSpec.MagSpecIOConfig(sr=16000, n_fft=64, hop_length=16, activation=act))
        cfg = ref.WaveNet.Config(io_spec=io, kernel_sizes=(2,), blocks=(3,), dims_dilated=(32,), apply_residuals=False,
                                 residuals_dim=None, skips_dim=None, groups=groups)
        net = ref.WaveNet.from_config(cfg).eval()
        load_recipe(net, seed=50 + groups, gain=1.5)
        rf = net.rf
        prompt = torch.rand(2, rf + 3, 33, generator=g)
        out = run_loop(net, (prompt,), 6)
        arrays.update({f"{tag}_prompt": prompt, f"{tag}_out": out[0], f"{tag}_rf": np.int64(rf)})
    save("freqnet.npz", **arrays)


def make_wavenet_padded():
    """pad_side=1 (every layer pads its cause on the left, eval returns the LAST position, wavenet_v2.py:87-88, :273):
    the generate loop hands rf-long windows over, so the padding is never reached and the samples are those of
    pad_side=0 -- pinned here rather than assumed"""
    g = torch.Generator().manual_seed(23)
    io = ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(input_module_type="embedding", mlp_dim=32))
    cfg = ref.WaveNet.Config(io_spec=io, blocks=(3, 2), dims_dilated=(16,), residuals_dim=16, skips_dim=16, pad_side=1)
    net = ref.WaveNet.from_config(cfg).eval()
    load_recipe(net, seed=11, gain=2.0)            # the weights of case (a) of wavenet.npz
    rf = net.rf
    prompt = torch.randint(0, 256, (3, rf + 5), generator=g)
    log, h = capture_raw(net)
    out = run_loop(net, (prompt,), 24)
    h.remove()
    # eval forward on a longer window: the class of its last position
    fwd = net((prompt,))[0]
    save("wavenet_pad1.npz", prompt=prompt, out=out[0], raw=torch.cat(log[:24], 1), rf=np.int64(rf), forward_last=fwd)


def make_srnn():
    g = torch.Generator().manual_seed(21)
    arrays = {}
    for tag, fs, kind, plen in (("gru", (16, 4, 1), "gru", 40), ("lstm", (16, 8, 8), "lstm", 32), ("rnn", (8, 2, 2), "rnn", 21)):
        io = ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(mlp_dim=32))
        cfg = ref.SampleRNN.Config(io_spec=io, frame_sizes=fs, hidden_dim=32, rnn_class=kind)
        net = ref.SampleRNN.from_config(cfg).eval()
        load_recipe(net, seed=30 + len(tag), gain=2.0)
        prompt = torch.randint(0, 256, (3, plen), generator=g)
        log, h = capture_raw(net)
        out = run_loop(net, (prompt,), 40)
        h.remove()
        arrays.update({f"{tag}_prompt": prompt, f"{tag}_out": out[0], f"{tag}_raw": torch.stack(log, 1)})
    save("srnn.npz", **arrays)


def make_srnn_weight_norm():
    """weight_norm=True (sample_rnn_v2.py:67-81 and SampleRNN.__init__): every parameter of the RNNs, up-samplers, tier input
    linears and of the MLP head is stored as a (g, v) pair; the recipe fills g and v, the network computes g v / |v|"""
    import warnings
    warnings.filterwarnings("ignore")
    g = torch.Generator().manual_seed(27)
    arrays = {}
    for tag, fs, kind, plen in (("gru", (16, 4, 1), "gru", 40), ("lstm", (16, 8, 8), "lstm", 32)):
        io = ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(mlp_dim=32))
        cfg = ref.SampleRNN.Config(io_spec=io, frame_sizes=fs, hidden_dim=32, rnn_class=kind, weight_norm=True)
        net = ref.SampleRNN.from_config(cfg).eval()
        load_recipe(net, seed=60 + len(tag), gain=2.0)
        prompt = torch.randint(0, 256, (3, plen), generator=g)
        log, h = capture_raw(net)
        out = run_loop(net, (prompt,), 40)
        h.remove()
        arrays.update({f"{tag}_prompt": prompt, f"{tag}_out": out[0], f"{tag}_raw": torch.stack(log, 1)})
    save("srnn_wn.npz", **arrays)


SRNN_OPTIONS = {
    "gru_n2": dict(frame_sizes=(16, 4, 1), rnn_class="gru", n_rnn=2),
    "lstm_n3": dict(frame_sizes=(16, 8, 8), rnn_class="lstm", n_rnn=3),
    "rnn_n2_mlp2": dict(frame_sizes=(8, 2, 2), rnn_class="rnn", n_rnn=2, io=dict(n_mlp_layers=2)),
    "gru_mean": dict(frame_sizes=(16, 4, 1), rnn_class="gru", inputs_mode="mean"),
    "lstm_mix_ones": dict(frame_sizes=(16, 8, 8), rnn_class="lstm", inputs_mode="static_mix", h0_init="ones"),
}


def make_srnn_options():
    """stacked recurrent layers per tier (n_rnn > 1), deeper MLP heads, the other inputs_mode values (one input: every mode
    weights it by 1) and h0_init='ones' - 40 free-running steps through the reference's loop"""
    g = torch.Generator().manual_seed(37)
    arrays = {}
    for tag, kw in SRNN_OPTIONS.items():
        kw = dict(kw)
        io = ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(mlp_dim=32, **kw.pop("io", {})))
        net = ref.SampleRNN.from_config(ref.SampleRNN.Config(io_spec=io, hidden_dim=32, **kw)).eval()
        load_recipe(net, seed=130 + len(tag), gain=2.0)
        prompt = torch.randint(0, 256, (3, 2 * kw["frame_sizes"][0] + 5), generator=g)
        log, h = capture_raw(net)
        out = run_loop(net, (prompt,), 40)
        h.remove()
        arrays.update({f"{tag}_prompt": prompt, f"{tag}_out": out[0], f"{tag}_raw": torch.stack(log, 1)})
    save("srnn_options.npz", **arrays)


MULTI_IO = {
    # tag: (network, input class sizes, per-target dict(mlp_dim, n_mlp_layers), network keywords)
    "srnn_sum_2x2": ("srnn", (256, 64), (dict(mlp_dim=32), dict(mlp_dim=48, n_mlp_layers=1)),
                     dict(frame_sizes=(16, 4, 1), rnn_class="gru", inputs_mode="sum")),
    "srnn_mix_2x1": ("srnn", (256, 64), (dict(mlp_dim=32),),
                     dict(frame_sizes=(16, 8, 8), rnn_class="lstm", inputs_mode="static_mix")),
    "srnn_mean_3x3": ("srnn", (256, 64, 32), (dict(mlp_dim=32), dict(mlp_dim=32, n_mlp_layers=2), dict(mlp_dim=16)),
                      dict(frame_sizes=(8, 2, 2), rnn_class="rnn", inputs_mode="mean")),
    "wn_2x2": ("wavenet", (256, 64), (dict(mlp_dim=32), dict(mlp_dim=48, n_mlp_layers=1)),
               dict(blocks=(3, 2), dims_dilated=(32,), dims_1x1=(16,), residuals_dim=32, skips_dim=32)),
    "wn_2x1": ("wavenet", (256, 64), (dict(mlp_dim=32),),
               dict(blocks=(4,), dims_dilated=(32,), dims_1x1=(16,), residuals_dim=32, skips_dim=32)),
    "wn_3x3_noskip": ("wavenet", (128, 64, 16), (dict(mlp_dim=32), dict(mlp_dim=32), dict(mlp_dim=16, n_mlp_layers=1)),
                      dict(blocks=(3,), dims_dilated=(32,), dims_1x1=(16, 16), residuals_dim=32)),
}


def multi_io_spec(net_kind, classes, heads):
    """an IOSpec of several mu-law streams: input m / target k are the input / target of IOSpec.mulaw_io at that class size
    (io_spec.py:222-256) - the composition the reference's IOSpec(inputs=..., targets=...) dataclass is made for"""
    kind = "embedding" if net_kind == "wavenet" else "framed_linear"
    ins = tuple(ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(q_levels=q, input_module_type=kind)).inputs[0] for q in classes)
    tgs = tuple(ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(q_levels=q, input_module_type=kind, **h)).targets[0]
                for q, h in zip(classes, heads))
    return ref.IOSpec(inputs=ins, targets=tgs)


def make_multi_io():
    """networks of several inputs and targets (modules/io.py:289-313 ZipReduceVariables over the inputs of a SampleRNN,
    sample_rnn_v2.py:141-145 / :160-173 / :181-182; one WaveNet input module per input and one output module per target,
    wavenet_v2.py:231-243 / :293) through the reference's loop, which writes output k into input k (loops/generate.py:213-218):
    24 free-running greedy steps; the raw outputs of every head at every step"""
    g = torch.Generator().manual_seed(47)
    arrays = {}
    for tag, (kind, classes, heads, kw) in MULTI_IO.items():
        io = multi_io_spec(kind, classes, heads)
        if kind == "srnn":
            net = ref.SampleRNN.from_config(ref.SampleRNN.Config(io_spec=io, hidden_dim=32, **kw)).eval()
            plen = 2 * kw["frame_sizes"][0] + 3
        else:
            net = ref.WaveNet.from_config(ref.WaveNet.Config(io_spec=io, **kw)).eval()
            plen = net.rf + 5
        load_recipe(net, seed=170 + len(tag), gain=2.0)
        prompts = tuple(torch.randint(0, q, (3, plen), generator=g) for q in classes)
        logs, handles = [], []
        for mod in net.output_modules:
            log = []
            handles.append(mod.estimator[0].fc.register_forward_hook(lambda m, i, o, log=log: log.append(o.detach().clone())))
            logs.append(log)
        out = run_loop(net, prompts, 24)
        for h in handles:
            h.remove()
        for m, (p_m, o_m) in enumerate(zip(prompts, out)):
            arrays[f"{tag}_prompt{m}"], arrays[f"{tag}_out{m}"] = p_m, o_m
        for k, log in enumerate(logs):
            arrays[f"{tag}_raw{k}"] = torch.stack(log, 1)
    save("multi_io.npz", **arrays)


def make_s2s():
    g = torch.Generator().manual_seed(31)
    io = ref.IOSpec.magspec_io(ref.IOSpec.MagSpecIOConfig(n_fft=128, hop_length=32))
    cfg = ref.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=32, hop=4)
    net = ref.Seq2SeqLSTMNetwork.from_config(cfg).eval()
    load_recipe(net, seed=41, gain=1.5)
    x = torch.rand(3, 4, 65, generator=g)
    y = net.generate_step((x,), t=4)
    prompt = torch.rand(2, 6, 65, generator=g)
    out = run_loop(net, (prompt,), 10)
    save("s2s.npz", x=x, y=y, prompt=prompt, out=out[0])


def make_s2s_variants():
    """the other encoder poolings and the 'repeat' decoder up-sampling (s2s_lstm_v2.py:105-113, :158-163)"""
    g = torch.Generator().manual_seed(33)
    io = ref.IOSpec.magspec_io(ref.IOSpec.MagSpecIOConfig(n_fft=128, hop_length=32))
    x = torch.rand(3, 4, 65, generator=g)
    arrays = {"x": x}
    for ds, us in (("edge_mean", "linear_resample"), ("sum", "linear_resample"), ("mean", "repeat"), ("edge_sum", "repeat"),
                   ("linear_resample", "interp"), ("edge_sum", "interp"), ("linear_resample", "linear_resample")):
        cfg = ref.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=32, hop=4, enc_downsampling=ds, dec_upsampling=us)
        net = ref.Seq2SeqLSTMNetwork.from_config(cfg).eval()
        load_recipe(net, seed=41, gain=1.5)
        arrays[f"y_{ds}_{us}"] = net.generate_step((x,), t=4)
    save("s2s_variants.npz", **arrays)


def make_s2s_stacks():
    """stacked bi-LSTMs with and without residuals (s2s_lstm_v2.py:96-104, :168-178)"""
    g = torch.Generator().manual_seed(35)
    io = ref.IOSpec.magspec_io(ref.IOSpec.MagSpecIOConfig(n_fft=128, hop_length=32))
    x = torch.rand(3, 4, 65, generator=g)
    arrays = {"x": x}
    for tag, kw in S2S_STACKS.items():
        cfg = ref.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=32, hop=4, **kw)
        net = ref.Seq2SeqLSTMNetwork.from_config(cfg).eval()
        load_recipe(net, seed=43, gain=1.5)
        arrays[f"y_{tag}"] = net.generate_step((x,), t=4)
    save("s2s_stacks.npz", **arrays)


S2S_MULAW = {"mlp0": dict(hop=4, io=dict(n_mlp_layers=0)),
             "mlp2_stack": dict(hop=2, enc_n_lstm=2, dec_n_lstm=2, dec_apply_residuals=True, enc_downsampling="mean", io=dict(n_mlp_layers=2))}


def make_s2s_mulaw():
    """class indices in (an nn.Embedding under ZipReduceVariables, s2s_lstm_v2.py:205-210), an MLP head with a learned temperature
    and the argmax of CategoricalSampler out (generate_step passes no temperature, :262-263; modules/targets.py:43-44) - the IO the
    reference's tests/test_seq2seq.py:149-154 trains and generates with"""
    g = torch.Generator().manual_seed(39)
    arrays = {}
    for tag, kw in S2S_MULAW.items():
        kw = dict(kw)
        io = ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(input_module_type="embedding", mlp_dim=32, **kw.pop("io")))
        net = ref.Seq2SeqLSTMNetwork.from_config(ref.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=32, **kw)).eval()
        load_recipe(net, seed=47, gain=1.5)
        hop = kw["hop"]
        log = []
        h = net.output_module.heads[0].estimator[0].fc.register_forward_hook(lambda m, i, o: log.append(o.detach().clone()))
        x = torch.randint(0, 256, (3, hop), generator=g)
        y = net.generate_step((x,), t=hop)
        raw = log.pop()
        prompt = torch.randint(0, 256, (2, hop + 2), generator=g)
        out = run_loop(net, (prompt,), 10)
        h.remove()
        arrays.update({f"{tag}_x": x, f"{tag}_y": y, f"{tag}_raw": raw, f"{tag}_prompt": prompt, f"{tag}_out": out[0],
                       f"{tag}_loop_raw": torch.cat(log, 1)})
    save("s2s_mulaw.npz", **arrays)


S2S_STACKS = {"e2d1": dict(enc_n_lstm=2), "e1d3": dict(dec_n_lstm=3), "e2d2res": dict(enc_n_lstm=2, dec_n_lstm=2, enc_apply_residuals=True, dec_apply_residuals=True),
              "e3d1res_sum": dict(enc_n_lstm=3, enc_apply_residuals=True, enc_downsampling="sum")}


def make_sampler():
    g = torch.Generator().manual_seed(51)
    logits = torch.randn(6, 1, 256, generator=g) * 3
    arrays = {"logits": logits, "argmax": ref.targets.CategoricalSampler().eval()(logits)}
    for tag, temp in (("t05", 0.5), ("t1", (1.,)), ("per_item", torch.tensor([0.5, 1., 2., 0.1, 1.5, 1.]))):
        t = ref.targets.as_tensor(temp, logits)
        l = logits / t
        arrays[f"probs_{tag}"] = (l - l.logsumexp(-1, keepdim=True)).exp()
        arrays[f"temp_{tag}"] = t.reshape(-1).expand(6) if t.numel() == 1 else t.reshape(-1)
    # learned-temperature column of the MLP head
    raw = torch.randn(5, 257, generator=g)
    mlp = ref.io.MLP(in_dim=8, hidden_dim=8, out_dim=256)
    temp = torch.sigmoid(raw[..., -1:])
    arrays["raw"] = raw
    arrays["raw_logits"] = raw[..., :-1] / torch.maximum(temp, mlp.min_temp)
    save("sampler.npz", **arrays)


def make_keys():
    """state_dict names and shapes of the reference networks at the BASELINE configs"""
    out = {}
    mu_emb = lambda: ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig(input_module_type="embedding"))
    mu_lin = lambda: ref.IOSpec.mulaw_io(ref.IOSpec.MuLawIOConfig())

    def keys(net):
        return {k: list(v.shape) for k, v in net.state_dict().items()}

    out["wavenet_default"] = keys(ref.WaveNet.from_config(ref.WaveNet.Config(io_spec=mu_emb())))
    out["wavenet_cfg2"] = keys(ref.WaveNet.from_config(ref.WaveNet.Config(
        io_spec=mu_emb(), blocks=(10,), dims_dilated=(64,), residuals_dim=64, skips_dim=64)))
    mag = ref.functionals.MagSpec(1024, 256, center=False)
    ext = ref.extractor.Extractor("signal", ref.functionals.FileToSignal(16000))
    io4 = mu_emb()
    io4 = ref.IOSpec(inputs=(io4.inputs[0], ref.io_spec.InputSpec("signal", mag, ref.io.LinearIO()).bind_to(ext)),
                     targets=io4.targets)
    out["wavenet_cfg4"] = keys(ref.WaveNet.from_config(ref.WaveNet.Config(
        io_spec=io4, blocks=(10, 10, 10), dims_dilated=(256,), dims_1x1=(256,), residuals_dim=256, skips_dim=256)))
    out["srnn_cfg1"] = keys(ref.SampleRNN.from_config(ref.SampleRNN.Config(io_spec=mu_lin())))
    out["srnn_cfg3"] = keys(ref.SampleRNN.from_config(ref.SampleRNN.Config(
        io_spec=mu_lin(), frame_sizes=(16, 4, 1), hidden_dim=512, rnn_class="gru")))
    out["s2s_cfg5"] = keys(ref.Seq2SeqLSTMNetwork.from_config(ref.Seq2SeqLSTMNetwork.Config(
        io_spec=ref.IOSpec.magspec_io(ref.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256)))))
    # host logic: rf / n_steps / unit conversions the loop relies on
    host = {"rf": {}, "n_steps": {}, "convert": []}
    for blocks, ks in (((3,), (2,)), ((10,), (2,)), ((10, 10, 10), (2,)), ((2, 2, 1), (2,)), ((3,), (3,))):
        net = ref.WaveNet.from_config(ref.WaveNet.Config(io_spec=mu_emb(), blocks=blocks, kernel_sizes=ks, dims_dilated=(4,)))
        host["rf"][f"{blocks}|{ks}"] = int(net.rf)
    s2s = ref.Seq2SeqLSTMNetwork.from_config(ref.Seq2SeqLSTMNetwork.Config(
        io_spec=ref.IOSpec.magspec_io(ref.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256)), model_dim=8))
    wn = ref.WaveNet.from_config(ref.WaveNet.Config(io_spec=mu_emb(), dims_dilated=(4,)))
    for dur in (1.0, 0.5, 0.032):
        c = ref.GenerateLoopV2.Config(output_duration_sec=dur)
        host["n_steps"][f"s2s|{dur}"] = int(ref.GenerateLoopV2.get_n_steps(c, s2s))
        host["n_steps"][f"wavenet|{dur}"] = int(ref.GenerateLoopV2.get_n_steps(c, wn))
    I = ref.item_spec
    for n in (22050, 8192, 5000, 1024, 2047):
        for n_fft, hop, pad in ((1024, 256, False), (2048, 512, True), (2048, 512, False), (512, 128, False)):
            fr = I.Frame(n_fft, hop, padding=pad)
            host["convert"].append({
                "n": n, "n_fft": n_fft, "hop": hop, "pad": pad,
                "s2f_len": I.convert(n, I.Sample(1), fr, True), "s2f_pos": I.convert(n, I.Sample(1), fr, False),
                "f2s_len": I.convert(n // hop, fr, I.Sample(1), True), "f2s_pos": I.convert(n // hop, fr, I.Sample(1), False),
            })
    # prompt positions of the generate loop: the reference's IndicesSampler (loops/samplers.py:50-81) under a fixed torch seed
    import importlib
    smp = importlib.import_module("mimikit.loops.samplers")
    # (version skew: the reference calls Sampler.__init__(None), which torch 2.10's Sampler no longer accepts)
    import torch.utils.data as tud
    tud.Sampler.__init__ = lambda self, *a, **k: None
    torch.manual_seed(123)
    sampler = smp.IndicesSampler(N=4, indices=(None, 7, None, None), max_i=10000, redraw=True, sampling_stride=16)
    host["indices_sampler"] = {"seed": 123, "indices": [None, 7, None, None], "max_i": 10000, "stride": 16,
                               "passes": [[int(i) for i in sampler] for _ in range(3)]}
    torch.manual_seed(321)
    sampler = smp.IndicesSampler(N=5, indices=[], min_i=3, max_i=50, redraw=False)
    host["indices_sampler_n"] = {"seed": 321, "N": 5, "min_i": 3, "max_i": 50, "passes": [[int(i) for i in sampler] for _ in range(2)]}
    out["host"] = host
    out["meta"] = META
    with open(os.path.join(OUT, "reference_facts.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("reference_facts.json written")


if __name__ == "__main__":
    if len(sys.argv) > 1:              # python make_golden.py multi_io [...]: only those files
        for name in sys.argv[1:]:
            globals()["make_" + name]()
        sys.exit(0)
    make_mulaw()
    make_stft()
    make_istft()
    make_wavenet()
    make_wavenet_options()
    make_wavenet_acts()
    make_mlp_heads()
    make_freqnet()
    make_wavenet_padded()
    make_srnn()
    make_srnn_weight_norm()
    make_srnn_options()
    make_s2s()
    make_s2s_variants()
    make_s2s_stacks()
    make_s2s_mulaw()
    make_sampler()
    make_multi_io()
    make_keys()
