"""Every BASELINE.json config at its REAL size on the HIP path, against the oracle (the reference's algorithm on the host):
cfg 1 SampleRNN defaults (16, 8, 8) / H = 256 / LSTM, cfg 2 WaveNet 10 x 64 ch / 8 clips, cfg 3 SampleRNN (16, 4, 1) /
H = 512 / GRU / 64 clips, cfg 4 WaveNet 30 x 256 ch + STFT conditioning / 32 clips per GPU, cfg 5 Seq2Seq D = 1024 / hop 8 /
64 clips per GPU.  Greedy classes bit-exact wherever the oracle's top-2 logit gap exceeds fp32 re-association noise
(helpers.margin_ok), raw logits rtol 1e-4 / atol 2e-4, Seq2Seq frames 1e-4 of the largest output."""
import numpy as np
import pytest
import torch

import mimikit_amd as mmk
from oracle import torch_ref as O
from oracle.weights import load_recipe
from tests import helpers as H

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
LOGIT_TOL = dict(rtol=1e-4, atol=2e-4)


class host_threads:
    """the big-window oracle is a handful of large convolutions: let it use more host cores than the suite's default"""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        import os
        self.old = torch.get_num_threads()
        torch.set_num_threads(max(self.old, min(self.n, os.cpu_count() or 1)))

    def __exit__(self, *a):
        torch.set_num_threads(self.old)


def cfg4_network():
    """bench.py's default workload: blocks (10, 10, 10) x 256 channels, one 513-bin conditioning input through
    LinearIO 513 -> 256, mu-law-256 MLP head (BASELINE configs[3] at its per-GPU share of 32 clips)"""
    io = mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(sr=16000, q_levels=256, input_module_type="embedding"))
    ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
    cond = mmk.InputSpec("signal", mmk.MagSpec(1024, 256, center=False), mmk.LinearIO()).bind_to(ext)
    io = mmk.IOSpec(inputs=(io.inputs[0], cond), targets=io.targets)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=(10, 10, 10), dims_dilated=(256,), dims_1x1=(256,),
                                                     residuals_dim=256, skips_dim=256)).eval()
    sd = load_recipe(net, seed=404, gain=2.0)
    arch = dict(kernels=[2] * 30, dilations=[2 ** (i % 10) for i in range(30)], has_skips=True, residuals=True)
    return net, sd, arch


def test_cfg4_wavenet_30x256_conditioned_32_clips(device):
    """2100 free-running steps (three persistent launches: 1024 + 1024 + 52; the d = 512 rings wrap twice), then the
    oracle is teacher-forced on the device's OWN history: the last 4 steps of ALL 32 clips, and the first two steps and
    the three steps around the first launch boundary for one clip of every XCD-local group.  The same generation twice
    must be bit-identical."""
    net, sd, arch = cfg4_network()
    net = net.to(device)
    rf, B, n = net.rf, 32, 2100
    assert rf == 3070
    P = 3072
    gen = torch.Generator().manual_seed(44)
    prompt = torch.randint(0, 256, (B, P), generator=gen)
    cond = torch.rand(B, P + n, 513, generator=gen)
    cond_d = cond.to(device)

    def run():
        idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
        net.before_generate((idx[:, :P], cond_d[:, :P]), None)
        net.generate_block((idx, cond_d), P, n)
        net.after_generate((idx,), None)
        return idx.cpu(), net._plan.last_logits(B).cpu()

    hist, last_raw = run()
    assert net._plan.persistent
    hist2, _ = run()
    assert torch.equal(hist, hist2)
    assert int(hist[:, P:].min()) >= 0 and int(hist[:, P:].max()) < 256
    assert len(torch.unique(hist[:, P:])) > 32          # not stuck on a constant

    def check(clips, steps):
        n_ok = n_all = 0
        for t in steps:
            raw = O.wavenet_window_forward(sd, (hist[clips, t - rf:t], cond[clips, t - rf:t]), n_cond=1, **arch)
            pick = O.categorical(O.mlp_logits(raw))[:, 0]
            gap_ok = H.margin_ok(raw.numpy())[:, 0]
            assert bool(((pick == hist[clips, t]) | ~gap_ok).all()), f"step {t - P}: classes differ from the oracle"
            n_ok += int(gap_ok.sum())
            n_all += len(clips)
            if t == P + n - 1:
                assert torch.allclose(last_raw[clips][gap_ok], raw[:, 0][gap_ok], **LOGIT_TOL)
        return n_ok / n_all

    one_per_group = [4 * g + (g % 4) for g in range(8)]
    with host_threads(32):
        frac = check(list(range(B)), range(P + n - 4, P + n))
        assert frac > 0.9
        check(one_per_group, [P, P + 1, P + 1023, P + 1024, P + 1025])


def test_cfg5_seq2seq_d1024_hop8_64_clips(device):
    """the class defaults (model_dim 1024, hop 8, 1 + 1 bi-LSTM, edge_sum, linear_resample) on magspec_io(22050, 1024, 256):
    one generate_step of 64 clips against the oracle, then two chained steps through generate_block"""
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io)).eval()
    assert net.config.model_dim == 1024 and net.config.hop == 8
    sd = load_recipe(net, seed=505, gain=1.5)
    net.to(device)
    B = 64
    x = torch.rand(B, 8, 513, generator=torch.Generator().manual_seed(55))
    with host_threads(32):
        want = O.s2s_step(sd, x, hop=8)
        want2 = O.s2s_step(sd, want, hop=8)
    got = net.generate_step((x.to(device),), t=8).cpu()
    assert got.shape == (B, 8, 513)
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    frames = torch.cat([x, torch.zeros(B, 16, 513)], 1).to(device)
    net.before_generate((frames[:, :8],), None)
    assert net.generate_block((frames,), 8, 16)
    net.after_generate((frames,), None)
    out = frames.cpu()
    assert float((out[:, 8:16] - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert float((out[:, 16:24] - want2).abs().max()) <= 2e-4 * float(want2.abs().max())


def test_cfg1_sample_rnn_defaults_h256_lstm(device):
    """tests/test_sample_rnn.py:90-113 of the reference: the class defaults (frame sizes (16, 8, 8), hidden 256, LSTM), batch 2,
    prompt 512, 512 new steps - here greedy and compared with the oracle step by step (teacher-forced on the device's
    history, so every step of both clips is checked, not only a common prefix)"""
    io = mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(sr=16000))
    net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=io)).eval()
    c = net.config
    assert tuple(c.frame_sizes) == (16, 8, 8) and c.hidden_dim == 256 and str(c.rnn_class) == "lstm"
    sd = load_recipe(net, seed=101, gain=2.0)
    net.to(device)
    B, P, n = 2, 512, 600
    prompt = torch.randint(0, 256, (B, P), generator=torch.Generator().manual_seed(11))
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx[:, :P],), None)
    net.generate_block((idx,), P, n)
    last_raw = net._plan.last_logits(B).cpu()
    net.after_generate((idx,), None)
    got = idx.cpu()
    o = O.SampleRNNOracle(sd, (16, 8, 8), 256, "lstm")
    want, raw = o.generate(prompt, n, keep_logits=True, forced=got)
    ok = H.margin_ok(raw.numpy())
    assert bool(((got[:, P:] == want[:, P:]) | ~ok).all())
    assert float(ok.float().mean()) > 0.9
    assert torch.allclose(last_raw[ok[:, -1]], raw[:, -1][ok[:, -1]], **LOGIT_TOL)
    # the same network through the loop with a temperature, as the reference's test runs it
    loop = mmk.GenerateLoopV2(mmk.GenerateLoopV2.Config(display_waveform=False, parameters=dict(temperature=(1.,))), net, 512,
                              [[np.arange(B), prompt]], logger=None)
    out = list(loop.run())[0][0]
    torch.set_grad_enabled(False)
    assert out.shape == (B, 1024) and out.dtype == torch.float32 and float(out.abs().max()) <= 1.0


def test_cfg3_sample_rnn_h512_gru_64_clips(device):
    """frame sizes (16, 4, 1), GRU, hidden 512, 64 clips, prompt 512 + 5 (P % rf != 0): 130 steps, every step of every clip
    against the oracle teacher-forced on the device's history"""
    io = mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(sr=16000))
    net = mmk.SampleRNN.from_config(mmk.SampleRNN.Config(io_spec=io, frame_sizes=(16, 4, 1), hidden_dim=512, rnn_class="gru")).eval()
    sd = load_recipe(net, seed=303, gain=2.0)
    net.to(device)
    B, P, n = 64, 517, 130
    prompt = torch.randint(0, 256, (B, P), generator=torch.Generator().manual_seed(33))
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx[:, :P],), None)
    net.generate_block((idx,), P, n)
    last_raw = net._plan.last_logits(B).cpu()
    net.after_generate((idx,), None)
    got = idx.cpu()
    o = O.SampleRNNOracle(sd, (16, 4, 1), 512, "gru")
    want, raw = o.generate(prompt, n, keep_logits=True, forced=got)
    ok = H.margin_ok(raw.numpy())
    assert bool(((got[:, P:] == want[:, P:]) | ~ok).all())
    assert float(ok.float().mean()) > 0.9
    assert torch.allclose(last_raw[ok[:, -1]], raw[:, -1][ok[:, -1]], **LOGIT_TOL)
    # sampled decode at the same size: every pick sits in the CDF interval of its uniform draw
    temp = torch.linspace(0.5, 1.5, B)
    torch.manual_seed(6)
    u = torch.rand((B, n), device=device)
    torch.manual_seed(6)
    idx2 = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx2[:, :P],), None)
    net.generate_block((idx2,), P, n, temperature=temp)
    net.after_generate((idx2,), None)
    got2 = idx2.cpu()
    _, raw2 = o.generate(prompt, n, keep_logits=True, forced=got2)
    okp, exact = H.sampled_picks_ok(raw2, temp, u.cpu(), got2[:, P:])
    assert bool(okp.all()) and float(exact.float().mean()) > 0.98


def test_cfg2_wavenet_10x64_8_clips_every_step(device):
    """10 x {1..512}, 64 channels, 8 clips, prompt 1024: 300 free-running steps, every step of every clip against the oracle
    teacher-forced on the device's history (the free-running comparison of test_gpu_networks stops at the first near-tie)"""
    net, sd, arch = H.wavenet_c()
    net = net.to(device)
    B, P, n = 8, 1024, 300
    prompt = torch.randint(0, 256, (B, P), generator=torch.Generator().manual_seed(22))
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx[:, :P],), None)
    net.generate_block((idx,), P, n)
    net.after_generate((idx,), None)
    got = idx.cpu()
    steps = list(range(0, 40)) + list(range(n - 40, n))      # 80 window forwards of 8 clips on the host
    for s in steps:
        t = P + s
        raw = O.wavenet_window_forward(sd, (got[:, t - net.rf:t],), **arch)
        pick = O.categorical(O.mlp_logits(raw))[:, 0]
        gap_ok = H.margin_ok(raw.numpy())[:, 0]
        assert bool(((pick == got[:, t]) | ~gap_ok).all()), f"step {s}"
