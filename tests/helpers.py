"""Shared test plumbing: golden fixtures, network builders (product package) and the matching
oracle argument sets."""
import json
import os

import numpy as np
import torch

import mimikit_amd as mmk
from oracle import torch_ref as O
from oracle.weights import load_recipe

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return {k: v for k, v in np.load(os.path.join(GOLDEN, name)).items()}


def facts():
    with open(os.path.join(GOLDEN, "reference_facts.json")) as f:
        return json.load(f)


def T(a):
    return torch.from_numpy(np.asarray(a))


def mu_emb(**kw):
    return mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(input_module_type="embedding", **kw))


def mu_lin(**kw):
    return mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(**kw))


# ---- the networks the golden vectors were produced with (tests/golden/make_golden.py) -------------
def wavenet_a():
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=mu_emb(mlp_dim=32), blocks=(3, 2), dims_dilated=(16,),
                                                     residuals_dim=16, skips_dim=16))
    sd = load_recipe(net, seed=11, gain=2.0)
    arch = dict(kernels=[2] * 5, dilations=[1, 2, 4, 1, 2], has_skips=True, residuals=True)
    return net.eval(), sd, arch


def wavenet_b():
    io = mu_emb(mlp_dim=32)
    ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
    io_b = mmk.IOSpec(inputs=(io.inputs[0], mmk.InputSpec("signal", mmk.MagSpec(22, 4, center=False), mmk.LinearIO()).bind_to(ext)),
                      targets=io.targets)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io_b, kernel_sizes=(3,), blocks=(3,), dims_dilated=(16,),
                                                     dims_1x1=(8,), residuals_dim=16, skips_dim=None))
    sd = load_recipe(net, seed=12, gain=2.0)
    arch = dict(kernels=[3] * 3, dilations=[1, 3, 9], has_skips=False, residuals=True)
    return net.eval(), sd, arch


def wavenet_c():
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=mu_emb(), blocks=(10,), dims_dilated=(64,),
                                                     residuals_dim=64, skips_dim=64))
    sd = load_recipe(net, seed=13, gain=2.0)
    arch = dict(kernels=[2] * 10, dilations=[2 ** i for i in range(10)], has_skips=True, residuals=True)
    return net.eval(), sd, arch


WAVENET_OPTIONS = {       # tests/golden/make_golden.py: make_wavenet_options
    "mlp2": dict(io=dict(n_mlp_layers=2)),
    "mlp3_cond": dict(io=dict(n_mlp_layers=3), cond=True),
    "nogate": dict(act_g=None),
    "nogate_cond": dict(act_g=None, cond=True),
    "rev": dict(reverse_layer_order=True),
    "rev_noskip": dict(reverse_layer_order=True, skips_dim=None),
    "lw": dict(layerwise_inputs=True),
    "lw_noskip_rev": dict(layerwise_inputs=True, reverse_layer_order=True, skips_dim=None),
    "tied": dict(tie_io_weights=True),
    "k3": dict(kernel_sizes=(3,)),                    # kernel sizes above 2: dilations 1, 3, 9 | 1, 3 (get_kernels_and_dilation)
    "k3_cond": dict(kernel_sizes=(3,), cond=True),
    "k4_noskip": dict(kernel_sizes=(4,), skips_dim=None),
    "aff": dict(with_affine_residuals=True),
    "aff_nogate_noskip": dict(with_affine_residuals=True, act_g=None, skips_dim=None),
}


WAVENET_ACTS = {          # tests/golden/make_golden.py: make_wavenet_acts
    "mish_tanh": dict(act_f="Mish", act_g="Tanh"),
    "relu_nogate": dict(act_f="ReLU", act_g=None),
    "sin_sig_cond": dict(act_f="Sin", act_g="Sigmoid", cond=True),
    "softplus_abs": dict(act_f="Softplus", act_g="Abs"),
    "id_cos_noskip": dict(act_f="Identity", act_g="Cos", skips_dim=None),
    "abs_nogate_cond": dict(act_f="Abs", act_g=None, cond=True),
}


def wavenet_act(tag):
    """the network of one wavenet_acts.npz case and the matching oracle arguments"""
    kw = dict(WAVENET_ACTS[tag])
    io = mu_emb(mlp_dim=32)
    cond = kw.pop("cond", False)
    if cond:
        ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
        io = mmk.IOSpec(inputs=(io.inputs[0], mmk.InputSpec("signal", mmk.MagSpec(22, 4, center=False), mmk.LinearIO()).bind_to(ext)),
                        targets=io.targets)
        kw["dims_1x1"] = (8,)
    kw.setdefault("skips_dim", 16)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=(3, 2), dims_dilated=(16,), residuals_dim=16, **kw))
    sd = load_recipe(net, seed=200 + len(tag), gain=1.5)
    arch = dict(kernels=[2] * 5, dilations=[1, 2, 4, 1, 2], has_skips=kw["skips_dim"] is not None, residuals=True,
                gated=kw["act_g"] is not None, n_cond=int(cond), act_f=kw["act_f"], act_g=kw["act_g"] or "Sigmoid")
    return net.eval(), sd, arch


MLP_HEADS = {             # tests/golden/make_golden.py: make_mlp_heads
    "wn_relu_dp": ("wavenet", dict(activation="ReLU", dropout=0.1)),
    "wn_tanh_2": ("wavenet", dict(activation="Tanh", n_hidden_layers=2)),
    "srnn_softplus_dp1d": ("srnn", dict(activation="Softplus", dropout1d=0.2)),
    "srnn_sigmoid": ("srnn", dict(activation="Sigmoid")),
}


def mlp_head_case(tag):
    """the network of one mlp_heads.npz case, its state_dict, the kind ("wavenet" / "srnn") and the oracle's arguments"""
    kind, head = MLP_HEADS[tag]
    head = dict(head)
    act = head.pop("activation")
    io = mu_emb(mlp_dim=32) if kind == "wavenet" else mu_lin(mlp_dim=32)
    io.targets[0].module.activation = mmk.ActivationConfig(act)
    for k, v in head.items():
        setattr(io.targets[0].module, k, v)
    n_dp = int(head.get("dropout", 0) > 0) + int(head.get("dropout1d", 0) > 0)
    if kind == "wavenet":
        net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=(3, 2), dims_dilated=(16,), residuals_dim=16, skips_dim=16))
        load_recipe(net, seed=300 + len(tag), gain=2.0)
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}      # (the hidden blocks of a deeper head share ONE Linear: what it ended up with)
        arch = dict(kernels=[2] * 5, dilations=[1, 2, 4, 1, 2], has_skips=True, residuals=True, n_mlp_hidden=head.get("n_hidden_layers", 0),
                    mlp_act=act, mlp_dropouts=n_dp)
    else:
        net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=io, frame_sizes=(8, 2, 2), hidden_dim=32, rnn_class="gru"))
        sd = load_recipe(net, seed=300 + len(tag), gain=8.0)
        arch = dict(frame_sizes=(8, 2, 2), hidden_dim=32, rnn_class="gru", mlp_act=act, mlp_dropouts=n_dp)
    return net.eval(), sd, kind, arch


def wavenet_option(tag):
    """the network of one wavenet_options.npz case and the matching oracle arguments"""
    kw = dict(WAVENET_OPTIONS[tag])
    io_kw = kw.pop("io", {})
    io = mu_emb(mlp_dim=32, **io_kw)
    cond = kw.pop("cond", False)
    if cond:
        ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
        io = mmk.IOSpec(inputs=(io.inputs[0], mmk.InputSpec("signal", mmk.MagSpec(22, 4, center=False), mmk.LinearIO()).bind_to(ext)),
                        targets=io.targets)
        kw["dims_1x1"] = (8,)
    kw.setdefault("skips_dim", 16)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=(3, 2), dims_dilated=(16,), residuals_dim=16, **kw))
    load_recipe(net, seed=100 + len(tag), gain=2.0)
    # (the hidden blocks of a deeper MLP head share ONE Linear: the state_dict, not the recipe, says what it ended up with)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    rev = bool(kw.get("reverse_layer_order"))
    k = kw.get("kernel_sizes", (2,))[0]
    dil = [1, k, k * k, 1, k]
    arch = dict(kernels=[k] * 5, dilations=dil[::-1] if rev else dil, has_skips=kw["skips_dim"] is not None,
                res_layers=[False, True, True, True, True] if rev else [True, True, True, True, False],
                gated=kw.get("act_g", "Sigmoid") is not None, layerwise_inputs=bool(kw.get("layerwise_inputs")),
                n_mlp_hidden=io_kw.get("n_mlp_layers", 0), n_cond=int(cond), affine=bool(kw.get("with_affine_residuals")))
    return net.eval(), sd, arch


FREQNET_CASES = {"g1": (1, "Identity"), "g4": (4, "Identity"), "g2abs": (2, "Abs")}


def freqnet(tag):
    """demos/freqnet.py at reduced size: magnitude frames in and out, no residual / skip path, grouped dilated convolutions"""
    groups, act = FREQNET_CASES[tag]
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=16000, n_fft=64, hop_length=16, activation=act))
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, kernel_sizes=(2,), blocks=(3,), dims_dilated=(32,),
                                                     apply_residuals=False, residuals_dim=None, skips_dim=None, groups=groups))
    sd = load_recipe(net, seed=50 + groups, gain=1.5)
    arch = dict(kernels=[2] * 3, dilations=[1, 2, 4], has_skips=False, residuals=False, groups=groups,
                head="linear_abs" if act == "Abs" else "linear")
    return net.eval(), sd, arch


SRNN_CASES = {"gru": ((16, 4, 1), "gru", 40), "lstm": ((16, 8, 8), "lstm", 32), "rnn": ((8, 2, 2), "rnn", 21)}


def srnn(tag, hidden=32, mlp_dim=32, seed=None, frame_sizes=None, kind=None, weight_norm=False):
    fs, k, _ = SRNN_CASES.get(tag, (frame_sizes, kind, None))
    fs, k = frame_sizes or fs, kind or k
    net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=mu_lin(mlp_dim=mlp_dim), frame_sizes=fs, hidden_dim=hidden,
                                                         rnn_class=k, weight_norm=weight_norm))
    sd = load_recipe(net, seed=((60 if weight_norm else 30) + len(tag)) if seed is None else seed, gain=2.0)
    return net.eval(), sd, dict(frame_sizes=fs, hidden_dim=hidden, rnn_class=k)


SRNN_OPTIONS = {      # tests/golden/make_golden.py: make_srnn_options
    "gru_n2": dict(frame_sizes=(16, 4, 1), rnn_class="gru", n_rnn=2),
    "lstm_n3": dict(frame_sizes=(16, 8, 8), rnn_class="lstm", n_rnn=3),
    "rnn_n2_mlp2": dict(frame_sizes=(8, 2, 2), rnn_class="rnn", n_rnn=2, io=dict(n_mlp_layers=2)),
    "gru_mean": dict(frame_sizes=(16, 4, 1), rnn_class="gru", inputs_mode="mean"),
    "lstm_mix_ones": dict(frame_sizes=(16, 8, 8), rnn_class="lstm", inputs_mode="static_mix", h0_init="ones"),
}


def srnn_option(tag):
    kw = dict(SRNN_OPTIONS[tag])
    io_kw = kw.pop("io", {})
    net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=mu_lin(mlp_dim=32, **io_kw), hidden_dim=32, **kw))
    load_recipe(net, seed=130 + len(tag), gain=2.0)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    arch = dict(frame_sizes=kw["frame_sizes"], hidden_dim=32, rnn_class=kw["rnn_class"], n_rnn=kw.get("n_rnn", 1),
                n_mlp_hidden=io_kw.get("n_mlp_layers", 0), h0=kw.get("h0_init", "zeros"))
    return net.eval(), sd, arch


MULTI_IO = {          # tests/golden/make_golden.py: make_multi_io (network, input class sizes, per-target head, network keywords)
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


def multi_io(tag):
    """the network of several inputs / targets the fixture ``multi_io.npz`` was made with, its state_dict, and what the oracle needs:
    for a SampleRNN the keywords of O.SampleRNNOracle, for a WaveNet (kernels, dilations, keywords of O.wavenet_generate_streams)"""
    kind, classes, heads, kw = MULTI_IO[tag]
    mtype = "embedding" if kind == "wavenet" else "framed_linear"
    ins = tuple(mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(q_levels=q, input_module_type=mtype)).inputs[0] for q in classes)
    tgs = tuple(mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(q_levels=q, input_module_type=mtype, **h)).targets[0]
                for q, h in zip(classes, heads))
    io = mmk.IOSpec(inputs=ins, targets=tgs)
    n_hidden = [h.get("n_mlp_layers", 0) for h in heads]
    if kind == "srnn":
        net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=io, hidden_dim=32, **kw))
        arch = dict(frame_sizes=kw["frame_sizes"], hidden_dim=32, rnn_class=kw["rnn_class"], in_classes=classes,
                    inputs_mode=kw["inputs_mode"], heads=[dict(n_mlp_hidden=n) for n in n_hidden])
    else:
        net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, **kw))
        ks, ds = O.wavenet_dilations((2,), kw["blocks"])
        arch = (ks, ds, dict(heads_n_hidden=n_hidden, cond_classes=classes[1:], has_skips="skips_dim" in kw))
    load_recipe(net, seed=170 + len(tag), gain=2.0)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return net.eval(), sd, arch, classes


S2S_VARIANTS = (("edge_mean", "linear_resample"), ("sum", "linear_resample"), ("mean", "repeat"), ("edge_sum", "repeat"),
                ("linear_resample", "interp"), ("edge_sum", "interp"), ("linear_resample", "linear_resample"))


S2S_STACKS = {"e2d1": dict(enc_n_lstm=2), "e1d3": dict(dec_n_lstm=3),
              "e2d2res": dict(enc_n_lstm=2, dec_n_lstm=2, enc_apply_residuals=True, dec_apply_residuals=True),
              "e3d1res_sum": dict(enc_n_lstm=3, enc_apply_residuals=True, enc_downsampling="sum")}


def s2s_tiny(downsampling="edge_sum", upsampling="linear_resample", seed=41, **kw):
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(n_fft=128, hop_length=32))
    kw.setdefault("enc_downsampling", downsampling)
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=32, hop=4,
                                                                           dec_upsampling=upsampling, **kw))
    sd = load_recipe(net, seed=seed, gain=1.5)
    return net.eval(), sd


S2S_MULAW = {"mlp0": dict(hop=4, io=dict(n_mlp_layers=0)),
             "mlp2_stack": dict(hop=2, enc_n_lstm=2, dec_n_lstm=2, dec_apply_residuals=True, enc_downsampling="mean", io=dict(n_mlp_layers=2))}


def s2s_mulaw(tag, model_dim=32, mlp_dim=32):
    """Seq2Seq on class indices: embedding in, MLP head + argmax out (the IO of the reference's tests/test_seq2seq.py:149-154)"""
    kw = dict(S2S_MULAW[tag])
    io = mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(input_module_type="embedding", mlp_dim=mlp_dim, **kw.pop("io")))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=model_dim, **kw))
    load_recipe(net, seed=47, gain=1.5)
    # (the hidden blocks of a deeper MLP head share ONE Linear: the state_dict, not the recipe, says what it ended up with)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    arch = dict(downsampling=kw.get("enc_downsampling", "edge_sum"), enc_residuals=kw.get("enc_apply_residuals", False),
                dec_residuals=kw.get("dec_apply_residuals", False))
    return net.eval(), sd, kw["hop"], arch


def margin_ok(raw, min_gap=5e-5):
    """top-1 / top-2 gap of the class logits (temperature column excluded): greedy decode is only
    comparable bit-exactly where no near-tie exists (fp32 re-association noise on these logits is
    ~1e-6; the committed fixtures have a smallest gap of 1.8e-4)"""
    top = torch.topk(T(raw)[..., :-1], 2, dim=-1).values
    return (top[..., 0] - top[..., 1]) > min_gap


def excluded_fraction(ok, what, limit=0.02):
    """share of the checked (clip, step) pairs whose oracle top-2 gap is too small for a bit-exact class comparison
    (margin_ok False): printed, and bounded - a regression that flattens the logits must not hide in the excluded share"""
    frac = 1.0 - float(T(ok).float().mean())
    print(f"[margin] {what}: {frac * 100:.3f} % of {T(ok).numel()} checked steps excluded (top-2 gap <= 5e-5); limit {limit * 100:.1f} %")
    assert frac <= limit, f"{what}: {frac * 100:.2f} % of the steps have near-tied logits"
    return frac


def sampled_picks_ok(raw, temperature, uniforms, picks, min_temp=1e-4, tol=2e-5):
    """Sampled decode, checked step by step: `raw` (B, n, q+1) are the oracle's head outputs for the history the device
    actually produced (teacher forcing), `picks` (B, n) the device's classes, drawn by inverting the CDF of
    softmax(logits / T) at `uniforms` (B, n).  A pick k is right iff  cdf[k-1] <= u * total < cdf[k]; device and oracle
    logits differ by fp32 re-association (~1e-6 relative, 2e-4 is the stated logit tolerance), which moves every CDF
    step by about that fraction of the total, so a draw within `tol` * total of a step may fall on either side of it
    and nowhere else.  Returns (ok (B, n) bool, exact (B, n) bool = inside the interval with no tolerance)."""
    from oracle import torch_ref as O
    raw = T(raw).double()
    logits = O.mlp_logits(raw.float(), min_temp).double()
    t = torch.as_tensor(temperature, dtype=torch.float64).reshape(-1, 1, 1)
    l = logits / t
    e = torch.exp(l - l.max(-1, keepdim=True).values)
    cdf = torch.cumsum(e, -1)
    total = cdf[..., -1]
    target = T(uniforms).double() * total
    k = T(picks).long()
    hi = cdf.gather(-1, k.unsqueeze(-1)).squeeze(-1)
    lo = hi - e.gather(-1, k.unsqueeze(-1)).squeeze(-1)
    exact = (lo <= target) & (target < hi)
    ok = (lo - tol * total <= target) & (target <= hi + tol * total)
    return ok, exact
