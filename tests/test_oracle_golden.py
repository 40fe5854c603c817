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


def test_seq2seq_matches_reference():
    g = H.golden("s2s.npz")
    _, sd = H.s2s_tiny()
    y = O.s2s_step(sd, H.T(g["x"]), hop=4)
    assert torch.allclose(y, H.T(g["y"]), rtol=1e-5, atol=1e-6)
    out = O.s2s_generate(sd, H.T(g["prompt"]), 10, hop=4)
    assert torch.allclose(out, H.T(g["out"]), rtol=1e-4, atol=1e-5)
