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
    pick_gen = torch.Generator().manual_seed(4)
    mid_steps = sorted(set(int(x) for x in torch.randint(2, n - 4, (40,), generator=pick_gen)))[:36]
    with host_threads(32):
        frac = check(list(range(B)), range(P + n - 4, P + n))
        check(one_per_group, [P, P + 1, P + 1023, P + 1024, P + 1025])
        # 36 random steps inside the blocks, four clips each (another four every time): a transient glitch of a hand-over that
        # leaves the rings intact would show here
        seen = ok_seen = 0
        for k, s_ in enumerate(mid_steps):
            clips = [(7 * k + 8 * j) % B for j in range(4)]
            f = check(clips, [P + s_])
            seen += 4
            ok_seen += round(f * 4)
    print(f"[margin] cfg 4 greedy: last 4 steps of 32 clips {100 * (1 - frac):.2f} % excluded; {len(mid_steps)} mid-block steps x 4 clips "
          f"{100 * (1 - ok_seen / seen):.2f} % excluded")
    assert 1 - frac <= 0.02 and 1 - ok_seen / seen <= 0.05

    # ---- sampled decode at the same size (another instantiation of the head): every checked pick lies in the oracle's CDF
    # interval of its uniform draw (helpers.sampled_picks_ok), for the device's own history
    temp = torch.linspace(0.6, 1.4, B)
    torch.manual_seed(8)
    u = torch.rand((B, n), device=device).cpu()
    torch.manual_seed(8)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx[:, :P], cond_d[:, :P]), None)
    net.generate_block((idx, cond_d), P, n, temperature=temp)
    net.after_generate((idx,), None)
    hist_s = idx.cpu()
    assert len(torch.unique(hist_s[:, P:])) > 64
    n_exact = n_all = 0
    with host_threads(32):
        for k, s_ in enumerate([0, 1, 1023, 1024, n - 2, n - 1] + mid_steps[:10]):
            clips = list(range(B)) if s_ == n - 1 else [(5 * k + 8 * j) % B for j in range(4)]
            t = P + s_
            raw = O.wavenet_window_forward(sd, (hist_s[clips, t - rf:t], cond[clips, t - rf:t]), n_cond=1, **arch)
            okp, exact = H.sampled_picks_ok(raw, temp[clips], u[clips, s_:s_ + 1], hist_s[clips, t:t + 1])
            assert bool(okp.all()), f"sampled step {s_}"
            n_exact += int(exact.sum())
            n_all += len(clips)
    assert n_exact >= 0.97 * n_all


def _cfg4_greedy_against_oracle(device, B, n, n_last, n_mid, seed, expect_set=None, tuning=None, expect_batched=False, expect_pair=None):
    """cfg 4 at ``B`` clips: ``n`` free-running greedy steps, twice (bit-identical), then the oracle teacher-forced on the device's
    own history: the last ``n_last`` steps of all clips, the steps around the start and the first launch boundary for eight clips,
    ``n_mid`` random mid-block steps for four clips each"""
    net, sd, arch = cfg4_network()
    net = net.to(device)
    if tuning:
        net.exec_tuning = dict(tuning)
    rf, P = net.rf, 3072
    gen = torch.Generator().manual_seed(seed)
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
    assert net._plan.stage_pipelined and net._plan.batch_pipelined == expect_batched
    if expect_set is not None:
        assert isinstance(net._plan, mmk.native.WaveNetPlanSet) == expect_set
    if expect_pair is not None:
        assert net._plan.pair_visits == expect_pair
    hist2, _ = run()
    assert torch.equal(hist, hist2)
    assert int(hist[:, P:].min()) >= 0 and int(hist[:, P:].max()) < 256
    assert len(torch.unique(hist[:, P:])) > 32

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
        return n_ok, n_all

    spread = [(B * g) // 8 + (g % max(B // 8, 1)) for g in range(8)]
    pick_gen = torch.Generator().manual_seed(seed + 1)
    mid_steps = sorted(set(int(x) for x in torch.randint(2, n - 4, (n_mid + 4,), generator=pick_gen)))[:n_mid]
    with host_threads(32):
        ok, al = check(list(range(B)), range(P + n - n_last, P + n))
        edge = [P, P + 1] + ([P + 1023, P + 1024, P + 1025] if n > 1030 else [])
        check(spread, edge)
        seen = ok_seen = 0
        for k, s_ in enumerate(mid_steps):
            clips = [(7 * k + (B // 4) * j + k // 3) % B for j in range(4)]
            f_ok, f_all = check(clips, [P + s_])
            seen += f_all
            ok_seen += f_ok
    print(f"[margin] cfg 4 greedy, {B} clips: last {n_last} steps of all clips {100 * (1 - ok / al):.2f} % excluded; {len(mid_steps)} mid-block steps x 4 clips "
          f"{100 * (1 - ok_seen / max(seen, 1)):.2f} % excluded")
    assert 1 - ok / al <= 0.02 and 1 - ok_seen / max(seen, 1) <= 0.05


def test_cfg4_wavenet_64_clips_in_one_ring(device):
    """R = 4 GPUs of the 256-clip job give 64 clips per GPU.  The stage pipeline streams them through ONE ring (two
    clips per stage slot on average): 1100 free-running steps across a launch boundary, against the oracle"""
    _cfg4_greedy_against_oracle(device, B=64, n=1100, n_last=2, n_mid=16, seed=464, expect_set=False, expect_pair=True)


def test_cfg4_wavenet_64_clips_one_clip_per_visit(device):
    """the same 64 clips with the ring's two-clip visits refused (MMK_WN_SPIPE_PAIR=0): the one-clip form of 40 clips and more
    (biases four visits behind, on the matrix pipe) - what 41 to 59 clips and the odd counts run on"""
    _cfg4_greedy_against_oracle(device, B=64, n=300, n_last=1, n_mid=8, seed=4640, expect_set=False, tuning={"MMK_WN_SPIPE_PAIR": "0"}, expect_pair=False)


def test_cfg4_wavenet_two_clips_per_visit_sizes(device):
    """wavenet_spipe_pair.inc at the edges of its range: 24 clips (the fewest, by name: 12 visits per step, fewer than stages), 54 (the first count the plan
    gives it), 100 (a multiple of 4 that is none of 8 or 16: the last batch of four biases of a step is followed by the next step's first);
    62 clips (even, no multiple of 4: the last batch of four biases of a step holds two clips of the next); 63 clips stay on the one-clip form"""
    _cfg4_greedy_against_oracle(device, B=24, n=260, n_last=1, n_mid=6, seed=4024, expect_set=False, tuning={"MMK_WN_SPIPE_PAIR": "1"}, expect_pair=True)
    _cfg4_greedy_against_oracle(device, B=54, n=260, n_last=1, n_mid=6, seed=4054, expect_set=False, expect_pair=True)
    _cfg4_greedy_against_oracle(device, B=100, n=1030, n_last=1, n_mid=8, seed=4100, expect_set=False, expect_pair=True)
    _cfg4_greedy_against_oracle(device, B=62, n=260, n_last=1, n_mid=6, seed=4062, expect_set=False, expect_pair=True)
    _cfg4_greedy_against_oracle(device, B=63, n=200, n_last=1, n_mid=4, seed=4063, expect_set=False, expect_pair=False)


def test_cfg4_wavenet_128_clips_in_one_ring(device):
    """R = 2: 128 clips per GPU, the most one ring takes (the bias image of a stage CU is 64 KB of its LDS then), two clips per visit: the plan's
    default for 128 clips since round 6 (100 us per step against the 16-clip groups' 108)"""
    _cfg4_greedy_against_oracle(device, B=128, n=1030, n_last=1, n_mid=12, seed=4128, expect_set=False, expect_pair=True)


def test_cfg4_wavenet_128_clips_one_clip_per_visit(device):
    """... and one clip per visit, by name"""
    _cfg4_greedy_against_oracle(device, B=128, n=300, n_last=1, n_mid=6, seed=41281, expect_set=False, tuning={"MMK_WN_SPIPE_PAIR": "0"}, expect_pair=False)


def test_cfg4_wavenet_more_clips_than_one_ring(device):
    """the reference's loop takes any batch (loops/generate.py:207-219): 136 clips run as two passes of 68 through the stage
    pipeline (native.WaveNetPlanSet), never on the round-1 fallback kernel (the one-clip ring by name)"""
    _cfg4_greedy_against_oracle(device, B=136, n=1030, n_last=1, n_mid=10, seed=4136, expect_set=True, tuning={"MMK_WN_BPIPE": "0"}, expect_pair=True)


def test_cfg4_wavenet_256_clips_in_groups_of_16(device):
    """SURVEY 8(e) at R = 1: the whole 256-clip job on one GPU.  The plan takes the stage pipeline's large-batch form (wavenet_bpipe.hip: 16 groups
    of 16 clips, a visit is a set of 16x16x4 matrix products): 1030 free-running steps across a launch boundary, against the oracle"""
    _cfg4_greedy_against_oracle(device, B=256, n=1030, n_last=1, n_mid=10, seed=4256, expect_set=False, expect_batched=True)


def test_cfg4_wavenet_128_clips_in_groups_of_16(device):
    """128 clips as eight groups of 16, by name (the plan's default above 128 clips, and from 105 on for the counts that are no multiple of 4)"""
    _cfg4_greedy_against_oracle(device, B=128, n=1030, n_last=1, n_mid=8, seed=41280, expect_set=False, expect_batched=True, tuning={"MMK_WN_BPIPE": "1"})


def test_cfg4_wavenet_ragged_groups_of_16(device):
    """150 clips: nine whole groups and one of six clips (the lanes of the clips that do not exist compute on class 0 and store nothing)"""
    _cfg4_greedy_against_oracle(device, B=150, n=300, n_last=1, n_mid=8, seed=4150, expect_set=False, expect_batched=True)


def test_cfg4_a_small_batch_after_a_large_one_gets_the_one_clip_ring(device):
    """the step kernel follows the CALL's batch: a network that generated 144 clips in groups of 16 (wavenet_bpipe.hip) and is then asked for 6
    clips runs them on the one-clip ring (a new plan), not as one group of 16 on the large-batch kernel - and back; every generation equals the
    oracle at its last step"""
    net, sd, arch = cfg4_network()
    net = net.to(device)
    rf, P, n = net.rf, 3072, 24
    for B, batched in ((144, True), (6, False), (144, True)):
        gen = torch.Generator().manual_seed(4000 + B)
        prompt = torch.randint(0, 256, (B, P), generator=gen)
        cond = torch.rand(B, P + n, 513, generator=gen)
        cond_d = cond.to(device)
        idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
        net.before_generate((idx[:, :P], cond_d[:, :P]), None)
        net.generate_block((idx, cond_d), P, n)
        net.after_generate((idx,), None)
        assert net._plan.stage_pipelined and net._plan.batch_pipelined == batched
        hist = idx.cpu()
        clips = list(range(0, B, max(B // 6, 1)))[:6]
        t = P + n - 1
        with host_threads(32):
            raw = O.wavenet_window_forward(sd, (hist[clips, t - rf:t], cond[clips, t - rf:t]), n_cond=1, **arch)
        gap_ok = H.margin_ok(raw.numpy())[:, 0]
        assert bool(((O.categorical(O.mlp_logits(raw))[:, 0] == hist[clips, t]) | ~gap_ok).all()) and int(gap_ok.sum()) >= 4


def test_cfg4_nan_with_the_poison_payload_is_a_value_not_a_missing_word(device):
    """the stage pipeline's messages mark "not arrived" with 0xFFFFFFFF - which is also a NaN.  A weight that carries exactly that
    NaN (here: one embedding entry of a class the prompt contains) must flow through the ring as the value it is: no hand-off
    waits for it, no timeout, no redo on the launch path (a warning).  What comes out is NaN-driven garbage, as in the reference."""
    import warnings
    net, sd, arch = cfg4_network()
    poison = torch.tensor([-1], dtype=torch.int32).view(torch.float32)
    with torch.no_grad():
        net.input_modules[0][0].weight[5, 7] = poison[0]
    net = net.to(device)
    assert net.input_modules[0][0].weight.view(torch.int32)[5, 7].item() == -1
    B, P, n = 8, 3072, 40
    prompt = torch.full((B, P), 5, dtype=torch.int64)
    cond_d = torch.rand(B, P + n, 513, generator=torch.Generator().manual_seed(3)).to(device)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net.before_generate((idx[:, :P], cond_d[:, :P]), None)
        net.generate_block((idx, cond_d), P, n)
        net.after_generate((idx,), None)
    assert net._plan is not None and net._plan.stage_pipelined
    out = idx[:, P:].cpu()
    assert int(out.min()) >= 0 and int(out.max()) < 256


def test_cfg4_composed_products_against_the_reference_association(device, monkeypatch):
    """The stage pipeline multiplies pre-composed matrices (tap 1 . W_res of the layer below, fc0 . W_skip: fp64 products rounded
    once); the per-layer launch path keeps the reference's association.  Same window, same step, both against the oracle: the
    raw head outputs of the two paths differ by fp32 re-association only - the difference is printed and held to the logit
    tolerance, so the share of the error budget that composition uses is on record."""
    net, sd, arch = cfg4_network()
    net = net.to(device)
    rf, B = net.rf, 8
    gen = torch.Generator().manual_seed(45)
    win = torch.randint(0, 256, (B, rf), generator=gen)
    cond = torch.rand(B, rf + 1, 513, generator=gen)
    raws = {}
    for mode, env in (("composed", {}), ("reference association", {"MMK_WN_PERSISTENT": "0"})):
        for k in ("MMK_WN_PERSISTENT", "MMK_WN_SPIPE", "MMK_WN_CHAIN"):
            monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
        for k, v in env.items():
            monkeypatch.setitem(mmk.native.PLAN_TUNING, k, v)
        net._plan = None
        net.generate_step((win.to(device), cond[:, :rf].to(device)), t=rf)
        assert net._plan.stage_pipelined == (mode == "composed")
        raws[mode] = net._plan.last_logits(B).cpu()
        net.after_generate((), None)
    with host_threads(32):
        want = O.wavenet_window_forward(sd, (win, cond[:, :rf]), n_cond=1, **arch)[:, 0]
    d_paths = float((raws["composed"] - raws["reference association"]).abs().max())
    d_comp = float((raws["composed"] - want).abs().max())
    d_ref = float((raws["reference association"] - want).abs().max())
    scale = float(want.abs().max())
    print(f"[composition] cfg 4 raw head outputs (|max| {scale:.3f}): composed vs launch path {d_paths:.3e}, composed vs oracle {d_comp:.3e}, "
          f"launch path vs oracle {d_ref:.3e}")
    assert torch.allclose(raws["composed"], want, **LOGIT_TOL) and torch.allclose(raws["reference association"], want, **LOGIT_TOL)
    assert d_paths <= 2e-4 + 1e-4 * scale


def test_cfg5_composed_products_against_the_reference_association(device, monkeypatch):
    """Seq2Seq cfg 5 with and without the pre-multiplied dec.fc . enc.fc_out (MMK_S2S_COMPOSED=0: the reference's association):
    one generate_step of 16 clips, both against the oracle; the difference between the two paths is printed and bounded"""
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io)).eval()
    sd = load_recipe(net, seed=505, gain=1.5)
    net.to(device)
    x = torch.rand(16, 8, 513, generator=torch.Generator().manual_seed(56))
    with host_threads(32):
        want = O.s2s_step(sd, x, hop=8)
    outs = {}
    for mode, env in (("composed", None), ("reference association", "0")):
        monkeypatch.delitem(mmk.native.PLAN_TUNING, "MMK_S2S_COMPOSED", raising=False)
        if env is not None:
            monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_S2S_COMPOSED", env)
        net._plan = None
        outs[mode] = net.generate_step((x.to(device),), t=8).cpu()
    scale = float(want.abs().max())
    d_paths = float((outs["composed"] - outs["reference association"]).abs().max())
    print(f"[composition] cfg 5 frames (|max| {scale:.3f}): composed vs uncomposed {d_paths:.3e}, composed vs oracle "
          f"{float((outs['composed'] - want).abs().max()):.3e}, uncomposed vs oracle {float((outs['reference association'] - want).abs().max()):.3e}")
    for o in outs.values():
        assert float((o - want).abs().max()) <= 1e-4 * scale
    assert d_paths <= 1e-4 * scale


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
    assert net._plan.resident_launches() == 2        # encoder and decoder layer as one resident launch each (csrc/lstm_seq.hip)
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
    H.excluded_fraction(ok, "cfg 1 SampleRNN defaults, 2 clips x 600 steps")
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
    H.excluded_fraction(ok, "cfg 3 SampleRNN (16, 4, 1) GRU 512, 64 clips x 130 steps")
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
    oks = []
    for s in steps:
        t = P + s
        raw = O.wavenet_window_forward(sd, (got[:, t - net.rf:t],), **arch)
        pick = O.categorical(O.mlp_logits(raw))[:, 0]
        gap_ok = H.margin_ok(raw.numpy())[:, 0]
        oks.append(gap_ok)
        assert bool(((pick == got[:, t]) | ~gap_ok).all()), f"step {s}"
    H.excluded_fraction(torch.stack(oks), "cfg 2 WaveNet 10 x 64, 8 clips x 80 steps")
