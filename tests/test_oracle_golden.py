"""Pins the oracle (oracle/torch_ref.py) to golden vectors produced by the reference's own code
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as O
from tests import helpers as H

torch.set_grad_enabled(False)


@pytest.mark.parametrize("tag", ["c1", "c05"])
def test_mulaw_matches_reference_bit_exact(tag):
    g = H.golden(f"mulaw_{tag}.npz")
    comp = float(g["compression"])
    codes = O.mulaw_compress(H.T(g["x"]), 256, comp)
    assert torch.equal(codes, H.T(g["codes"]))
    exp = O.mulaw_expand(H.T(g["all_codes"]), 256, comp)
    assert torch.equal(exp, H.T(g["expanded"]))


def test_magspec_matches_reference():
    g = H.golden("stft.npz")
    for key in g:
        if not key.startswith("mag_"):
            continue
        parts = key.split("_")
        src = "y" if parts[1] == "y" else "x"
        n_fft, hop, center = (int(p) for p in parts[-3:])
        got = O.magspec(H.T(g[src]), n_fft, hop, bool(center))
        assert got.shape == g[key].shape
        assert torch.allclose(got, H.T(g[key]), rtol=0, atol=1e-6 * float(np.abs(g[key]).max()))


def _angle_diff(a, b):
    d = (a - b).abs()
    return torch.minimum(d, (2 * np.pi - d).abs())


def test_stft_coordinates_and_istft_match_reference():
    g = H.golden("istft.npz")
    x = H.T(g["x"])
    for coord in ("pol", "car", "angle"):
        want = H.T(g[f"stft_{coord}_1024_256"])
        got = O.stft_coord(x, 1024, 256, coord, center=True)
        assert got.shape == want.shape
        if coord == "car":
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-4)
    assert torch.allclose(O.stft_coord(x, 1024, 256, "car", center=True, pad_mode="reflect"), H.T(g["stft_car_1024_256_reflect"]),
                          rtol=1e-5, atol=1e-4)
    pol = O.stft_coord(x, 1024, 200, "pol", center=False)
    want = H.T(g["stft_pol_1024_200_nc"])
    assert torch.allclose(pol[..., 0], want[..., 0], rtol=1e-5, atol=1e-4)
    big = want[..., 0] > 1e-2                                    # the phase of a tiny bin is noise
    assert float(_angle_diff(pol[..., 1], want[..., 1])[big].max()) < 1e-3
    spec = H.T(g["spec_pol"])
    for key, hop, coord in (("istft_pol_1024_256", 256, "pol"), ("istft_pol_1024_100", 100, "pol"), ("istft_car_1024_256", 256, "car")):
        got = O.istft(spec, 1024, hop, coord)
        assert got.shape == g[key].shape, key
        assert torch.allclose(got, H.T(g[key]), rtol=1e-5, atol=1e-6), key
    rt = O.istft(O.stft_coord(x, 1024, 256, "pol", center=True), 1024, 256, "pol")
    assert torch.allclose(rt, H.T(g["roundtrip_1024_256"]), rtol=1e-5, atol=1e-5)
    assert torch.allclose(rt, x[:, :rt.shape[1]], atol=1e-5)


def test_griffin_lim_restatement_properties():
    """parity unpinned (no torchaudio here): what can be checked of the restatement without it -- shapes, the
    rand_init=False branch, and that the iteration reduces the spectral inconsistency it is built to reduce."""
    gen = torch.Generator().manual_seed(9)
    t = torch.arange(8192) / 22050.
    x = (0.5 * torch.sin(2 * np.pi * 440 * t) + 0.25 * torch.sin(2 * np.pi * 1320 * t))[None] + 0.01 * torch.randn(1, 8192, generator=gen)
    mag = O.stft_coord(x, 1024, 256, "mag", center=True)
    init = torch.rand(mag.shape, dtype=torch.complex64, generator=gen)

    def err(y):
        return float((O.stft_coord(y, 1024, 256, "mag", center=True, pad_mode="reflect") - mag).norm() / mag.norm())

    y0 = O.griffin_lim(mag, 1024, 256, 0, 0.99, init)
    y32 = O.griffin_lim(mag, 1024, 256, 32, 0.99, init)
    assert y32.shape == (1, 256 * (mag.shape[1] - 1))
    assert err(y32) < 0.5 * err(y0)
    assert O.griffin_lim(mag, 1024, 256, 2, 0.99, None).shape == y32.shape


def test_sampler_matches_reference():
    g = H.golden("sampler.npz")
    logits = H.T(g["logits"])
    assert torch.equal(O.categorical(logits), H.T(g["argmax"]))
    assert torch.allclose(O.mlp_logits(H.T(g["raw"])), H.T(g["raw_logits"]), rtol=1e-6, atol=0)
    # inverse-CDF sampling draws from the reference's distribution
    gen = torch.Generator().manual_seed(0)
    for tag in ("t05", "t1", "per_item"):
        probs, temp = H.T(g[f"probs_{tag}"])[:, 0], H.T(g[f"temp_{tag}"])
        n = 20000
        u = torch.rand(n, logits.size(0), generator=gen)
        draws = torch.stack([O.categorical(logits, temp, u[i].reshape(-1, 1)) for i in range(0, n, 1)][:4000])[:, :, 0]
        for r in range(logits.size(0)):
            hist = torch.bincount(draws[:, r], minlength=256).float() / draws.size(0)
            assert (hist - probs[r]).abs().max() < 0.03


def test_wavenet_unconditioned_matches_reference():
    g = H.golden("wavenet.npz")
    _, sd, arch = H.wavenet_a()
    prompt = H.T(g["a_prompt"])
    idx, raw = O.wavenet_generate(sd, prompt, (), 24, keep_logits=True, **arch)
    assert torch.allclose(raw, H.T(g["a_raw"]), rtol=1e-5, atol=1e-5)
    assert torch.equal(idx, H.T(g["a_out"]))
    assert torch.equal(O.mulaw_expand(idx), H.T(g["a_inversed"]))
    rf = int(g["a_rf"])
    raw1 = O.wavenet_window_forward(sd, (prompt[:, -rf:],), **arch)
    assert torch.equal(O.categorical(O.mlp_logits(raw1)), H.T(g["a_step"]))


def test_wavenet_conditioned_kernel3_matches_reference():
    g = H.golden("wavenet.npz")
    _, sd, arch = H.wavenet_b()
    idx, cond, rf = H.T(g["b_idx"]), H.T(g["b_cond"]), int(g["b_rf"])
    assert rf == O.wavenet_rf(arch["kernels"], arch["dilations"])
    raws = [O.wavenet_window_forward(sd, (idx[:, t - rf:t], cond[:, t - rf:t]), n_cond=1, **arch)
            for t in range(rf, idx.size(1))]
    raw = torch.cat(raws, 1)
    assert torch.allclose(raw, H.T(g["b_raw"]), rtol=1e-5, atol=1e-5)
    assert torch.equal(O.categorical(O.mlp_logits(raw)), H.T(g["b_argmax"]))


@pytest.mark.parametrize("tag", list(H.FREQNET_CASES))
def test_wavenet_on_magnitude_frames_matches_reference(tag):
    g = H.golden("freqnet.npz")
    _, sd, arch = H.freqnet(tag)
    assert int(g[f"{tag}_rf"]) == O.wavenet_rf(arch["kernels"], arch["dilations"])
    out = O.wavenet_generate_frames(sd, H.T(g[f"{tag}_prompt"]), 6, **arch)
    assert torch.allclose(out, H.T(g[f"{tag}_out"]), rtol=1e-5, atol=1e-6)


def test_wavenet_pad_side_1_generates_like_pad_side_0():
    """the reference's loop on a pad_side=1 network: rf-long windows never reach the padding, so the oracle's
    pad_side=0 algorithm reproduces its samples and the class of an eval forward's last position"""
    g = H.golden("wavenet_pad1.npz")
    _, sd, arch = H.wavenet_a()
    rf = int(g["rf"])
    assert rf == O.wavenet_rf(arch["kernels"], arch["dilations"])
    prompt = H.T(g["prompt"])
    idx, raw = O.wavenet_generate(sd, prompt, (), 24, keep_logits=True, **arch)
    assert torch.allclose(raw, H.T(g["raw"]), rtol=1e-5, atol=1e-5)
    assert torch.equal(idx, H.T(g["out"]))
    last = O.categorical(O.mlp_logits(O.wavenet_window_forward(sd, (prompt[:, -rf:],), **arch)))
    assert torch.equal(last, H.T(g["forward_last"]))


def test_wavenet_cfg2_shape_matches_reference():
    g = H.golden("wavenet.npz")
    _, sd, arch = H.wavenet_c()
    idx, raw = O.wavenet_generate(sd, H.T(g["c_prompt"]), (), 12, keep_logits=True, **arch)
    assert torch.allclose(raw, H.T(g["c_raw"]), rtol=1e-5, atol=1e-5)
    assert torch.equal(idx, H.T(g["c_out"]))


@pytest.mark.parametrize("tag", ["gru", "lstm", "rnn"])
def test_sample_rnn_matches_reference(tag):
    g = H.golden("srnn.npz")
    _, sd, arch = H.srnn(tag)
    o = O.SampleRNNOracle(sd, **arch)
    idx, raw = o.generate(H.T(g[f"{tag}_prompt"]), 40, keep_logits=True)
    ref_raw = H.T(g[f"{tag}_raw"]).reshape(raw.shape)
    assert torch.allclose(raw, ref_raw, rtol=1e-5, atol=1e-5)
    assert torch.equal(idx, H.T(g[f"{tag}_out"]))


@pytest.mark.parametrize("tag", ["gru", "lstm"])
def test_sample_rnn_weight_norm_matches_reference(tag):
    import warnings
    warnings.filterwarnings("ignore")
    g = H.golden("srnn_wn.npz")
    _, sd, arch = H.srnn(tag, weight_norm=True)
    assert any(k.endswith("_g") for k in sd)
    o = O.SampleRNNOracle(O.fold_weight_norm(sd), **arch)
    idx, raw = o.generate(H.T(g[f"{tag}_prompt"]), 40, keep_logits=True)
    assert torch.allclose(raw, H.T(g[f"{tag}_raw"]).reshape(raw.shape), rtol=1e-5, atol=1e-5)
    assert torch.equal(idx, H.T(g[f"{tag}_out"]))


def test_seq2seq_matches_reference():
    g = H.golden("s2s.npz")
    _, sd = H.s2s_tiny()
    y = O.s2s_step(sd, H.T(g["x"]), hop=4)
    assert torch.allclose(y, H.T(g["y"]), rtol=1e-5, atol=1e-6)
    out = O.s2s_generate(sd, H.T(g["prompt"]), 10, hop=4)
    assert torch.allclose(out, H.T(g["out"]), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("ds,us", H.S2S_VARIANTS)
def test_seq2seq_pooling_and_upsampling_variants_match_reference(ds, us):
    g = H.golden("s2s_variants.npz")
    _, sd = H.s2s_tiny(ds, us)
    assert ("dec.fc.fc.weight" in sd) == (us == "linear_resample")
    y = O.s2s_step(sd, H.T(g["x"]), hop=4, downsampling=ds, upsampling=us)
    assert torch.allclose(y, H.T(g[f"y_{ds}_{us}"]), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", list(H.S2S_STACKS))
def test_seq2seq_lstm_stacks_match_reference(tag):
    g = H.golden("s2s_stacks.npz")
    kw = H.S2S_STACKS[tag]
    import warnings
    warnings.filterwarnings("ignore")
    _, sd = H.s2s_tiny(seed=43, **kw)
    # the reference hands dec_apply_residuals to the decoder as its weight_norm flag (s2s_lstm_v2.py:221): (g, v) pairs
    assert any(k.endswith("_g") for k in sd) == bool(kw.get("dec_apply_residuals", False))
    sd = O.fold_weight_norm(sd)
    y = O.s2s_step(sd, H.T(g["x"]), hop=4, downsampling=kw.get("enc_downsampling", "edge_sum"),
                   enc_residuals=kw.get("enc_apply_residuals", False), dec_residuals=kw.get("dec_apply_residuals", False))
    assert torch.allclose(y, H.T(g[f"y_{tag}"]), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", list(H.S2S_MULAW))
def test_seq2seq_on_class_indices_matches_reference(tag):
    """embedding in, MLP head + argmax out: one generate_step (float classes, as the reference returns them) and the loop"""
    import warnings
    warnings.filterwarnings("ignore")
    g = H.golden("s2s_mulaw.npz")
    _, sd, hop, arch = H.s2s_mulaw(tag)
    sd = O.fold_weight_norm(sd)
    y, raw = O.s2s_step(sd, H.T(g[f"{tag}_x"]), hop, return_raw=True, **arch)
    assert y.dtype == torch.float32
    assert torch.allclose(raw, H.T(g[f"{tag}_raw"]), rtol=1e-5, atol=1e-5)
    assert bool(H.margin_ok(raw).all()) and torch.equal(y, H.T(g[f"{tag}_y"]))
    out = O.s2s_generate(sd, H.T(g[f"{tag}_prompt"]), 10, hop, **arch)
    assert out.dtype == torch.int64 and bool(H.margin_ok(g[f"{tag}_loop_raw"]).all())
    assert torch.equal(out, H.T(g[f"{tag}_out"]))


@pytest.mark.parametrize("tag", list(H.WAVENET_OPTIONS))
def test_wavenet_options_oracle_matches_reference(tag):
    """deeper MLP heads, act_g=None, reverse_layer_order, layerwise_inputs, tie_io_weights: the oracle's loop against the
    reference's (classes exact, raw head outputs 1e-5) on the committed fixture"""
    g = H.golden("wavenet_options.npz")
    _, sd, arch = H.wavenet_option(tag)
    n_cond = arch.pop("n_cond")
    prompt = H.T(g[f"{tag}_prompt"])
    n = 16
    cond = ()
    if n_cond:
        c = H.T(g[f"{tag}_cond"])
        cond = (torch.cat([c, torch.zeros(c.size(0), n, c.size(2))], 1),)     # the loop leaves blanks in the generated region
    out, raw = O.wavenet_generate(sd, prompt, cond, n, keep_logits=True, **arch)
    assert torch.equal(out, H.T(g[f"{tag}_out"]))
    assert torch.allclose(raw, H.T(g[f"{tag}_raw"]), rtol=1e-5, atol=1e-5)
    assert bool(H.margin_ok(g[f"{tag}_raw"]).all())


@pytest.mark.parametrize("tag", list(H.WAVENET_ACTS))
def test_wavenet_activations_oracle_matches_reference(tag):
    """Config.act_f / act_g other than Tanh / Sigmoid (Mish, ReLU, Sin, Softplus, Identity, Abs for f; Tanh, Abs, Cos, none for g): the oracle's loop
    against the reference's (classes exact, raw head outputs 1e-5) on the committed fixture"""
    g = H.golden("wavenet_acts.npz")
    _, sd, arch = H.wavenet_act(tag)
    n_cond = arch.pop("n_cond")
    prompt = H.T(g[f"{tag}_prompt"])
    n = 16
    cond = ()
    if n_cond:
        c = H.T(g[f"{tag}_cond"])
        cond = (torch.cat([c, torch.zeros(c.size(0), n, c.size(2))], 1),)     # the loop leaves blanks in the generated region
    out, raw = O.wavenet_generate(sd, prompt, cond, n, keep_logits=True, **arch)
    assert torch.equal(out, H.T(g[f"{tag}_out"]))
    assert torch.allclose(raw, H.T(g[f"{tag}_raw"]), rtol=1e-5, atol=1e-5)
    assert bool(H.margin_ok(g[f"{tag}_raw"]).all())


@pytest.mark.parametrize("tag", list(H.MLP_HEADS))
def test_mlp_head_variants_oracle_matches_reference(tag):
    """MLPIO.activation other than Mish (ReLU, Tanh, Softplus, Sigmoid) and heads with Dropout / Dropout1d modules between their Linears (identities in
    eval mode; the Linears' state_dict keys move): the oracle's loops against the reference's on the committed fixture"""
    g = H.golden("mlp_heads.npz")
    _, sd, kind, arch = H.mlp_head_case(tag)
    prompt = H.T(g[f"{tag}_prompt"])
    if kind == "wavenet":
        out, raw = O.wavenet_generate(sd, prompt, (), 16, keep_logits=True, **arch)
    else:
        out, raw = O.SampleRNNOracle(sd, **arch).generate(prompt, 24, keep_logits=True)
    assert torch.equal(out, H.T(g[f"{tag}_out"]))
    assert torch.allclose(raw.reshape(-1), H.T(g[f"{tag}_raw"]).reshape(-1), rtol=1e-5, atol=1e-5)
    assert bool(H.margin_ok(g[f"{tag}_raw"].reshape(3, -1, 257)).all())


@pytest.mark.parametrize("tag", list(H.SRNN_OPTIONS))
def test_sample_rnn_options_oracle_matches_reference(tag):
    """stacked recurrent layers (n_rnn 2 / 3), deeper MLP head, inputs_mode mean / static_mix, h0_init ones"""
    g = H.golden("srnn_options.npz")
    _, sd, arch = H.srnn_option(tag)
    o = O.SampleRNNOracle(sd, **arch)
    out, raw = o.generate(H.T(g[f"{tag}_prompt"]), 40, keep_logits=True)
    assert torch.equal(out, H.T(g[f"{tag}_out"]))
    assert torch.allclose(raw, H.T(g[f"{tag}_raw"]).reshape(raw.shape), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", list(H.MULTI_IO))
def test_multi_input_multi_target_oracle_matches_reference(tag):
    """networks of several inputs and targets through the reference's loop (output k is written into input k): ZipReduceVariables
    over the inputs of a SampleRNN (sum / mean / static_mix), one output module per target; a WaveNet whose conditioning inputs
    are class streams through EmbeddingIO modules, with and without skips, more inputs than targets included"""
    g = H.golden("multi_io.npz")
    _, sd, arch, classes = H.multi_io(tag)
    prompts = tuple(H.T(g[f"{tag}_prompt{m}"]) for m in range(len(classes)))
    if tag.startswith("srnn"):
        outs, raws = O.SampleRNNOracle(sd, **arch).generate(prompts, 24, keep_logits=True)
    else:
        ks, ds, kw = arch
        n_tgt = len(kw["heads_n_hidden"])       # (the loop fills every tensor with blanks behind the prompt; only the first n_tgt are written)
        blank = tuple(torch.cat([p, torch.zeros(p.size(0), 24, dtype=p.dtype)], 1) for p in prompts[n_tgt:])
        outs, raws = O.wavenet_generate_streams(sd, prompts[:n_tgt] + blank, 24, ks, ds, keep_logits=True, **kw)
        outs = outs + blank
    for m in range(len(classes)):
        assert torch.equal(outs[m], H.T(g[f"{tag}_out{m}"])), (tag, m)
    for k, raw in enumerate(raws):
        assert torch.allclose(raw, H.T(g[f"{tag}_raw{k}"]).reshape(raw.shape), rtol=1e-5, atol=1e-5), (tag, k)
