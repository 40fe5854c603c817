"""GPU parity of the three generate paths against the reference's golden vectors and the oracle.
Greedy class indices must be bit-exact (fixtures are margin-checked); head outputs (raw logits)
within fp32 tolerance 2e-4 abs / 1e-4 rel; STFT-frame outputs within 1e-4 relative."""
import numpy as np
import pytest
import torch

import mimikit_amd as mmk
from oracle import torch_ref as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

LOGIT_TOL = dict(rtol=1e-4, atol=2e-4)


def list_loader(*prompts):
    return [[np.arange(prompts[0].size(0)), *prompts]]


def run_loop(net, prompts, n_steps, **cfg):
    cfg.setdefault("yield_inversed_outputs", False)
    loop = mmk.GenerateLoopV2(mmk.GenerateLoopV2.Config(display_waveform=False, **cfg), net, n_steps,
                              list_loader(*prompts), logger=None)
    outs = list(loop.run())
    assert torch.is_grad_enabled()      # teardown restored it
    torch.set_grad_enabled(False)
    return outs[0]


# ---------------------------------------------------------------------------- WaveNet
def test_wavenet_loop_matches_reference_golden(device):
    g = H.golden("wavenet.npz")
    assert bool(H.margin_ok(g["a_raw"]).all())
    net, sd, arch = H.wavenet_a()
    prompt = H.T(g["a_prompt"])
    out = run_loop(net, (prompt,), 24)
    assert isinstance(out, tuple) and out[0].dtype == torch.int64
    assert torch.equal(out[0].cpu(), H.T(g["a_out"]))
    # last step's head outputs are still in the plan
    raw_last = net._plan.last_logits(prompt.size(0)).cpu()
    assert torch.allclose(raw_last, H.T(g["a_raw"])[:, -1], **LOGIT_TOL)
    inv = run_loop(net, (prompt,), 24, yield_inversed_outputs=True)
    # expanded audio is fp32: exact against the oracle on this host, 2 ulp against the fixture
    # (made on another CPU; torch's vectorised exp differs in the last bit between CPU ISAs)
    assert torch.equal(inv[0].cpu(), O.mulaw_expand(H.T(g["a_out"])))
    assert torch.allclose(inv[0].cpu(), H.T(g["a_inversed"]), rtol=3e-7, atol=1e-9)


def test_wavenet_generate_step_protocol(device):
    """tests/test_wavenet.py:140-165 of the reference: before_generate, one generate_step, after_generate"""
    g = H.golden("wavenet.npz")
    net, _, _ = H.wavenet_a()
    net = net.to(device)
    prompt = H.T(g["a_prompt"]).to(device)
    rf = net.rf
    assert rf == int(g["a_rf"])
    net.before_generate((prompt,), 0)
    out = net.generate_step((prompt[:, -rf:],), t=prompt.size(1))
    net.after_generate(out, 0)
    assert type(out) is tuple and out[0].shape == (3, 1) and out[0].ndim == prompt.ndim
    assert torch.equal(out[0].cpu(), H.T(g["a_step"]))
    # without before_generate the queues are rebuilt from the window: same answer
    out2 = net.generate_step((prompt[:, -rf:],), t=prompt.size(1))
    assert torch.equal(out2[0].cpu(), H.T(g["a_step"]))
    for temp in (0.5, (1.,)):
        o = net.generate_step((prompt[:, -rf:],), t=prompt.size(1), temperature=temp)
        assert o[0].shape == (3, 1) and int(o[0].min()) >= 0 and int(o[0].max()) < 256
    # eval forward == one step from the first rf positions of the window
    fwd = net((prompt[:, :rf + 1],))
    want = O.categorical(O.mlp_logits(O.wavenet_window_forward(H.wavenet_a()[1], (prompt[:, :rf].cpu(),), **H.wavenet_a()[2])))
    assert torch.equal(fwd[0].cpu(), want)
    with pytest.raises(RuntimeError):
        net((prompt[:, :rf - 1],))
    with pytest.raises(RuntimeError):
        net.cpu().eval().generate_step((prompt.cpu()[:, -rf:],), t=rf)


def test_wavenet_step_by_step_equals_block(device):
    """the reference's own per-step loop (generate_step per t) and the fused block give the same clip"""
    g = H.golden("wavenet.npz")
    net, _, _ = H.wavenet_a()
    net = net.to(device)
    prompt = H.T(g["a_prompt"]).to(device)
    rf, prior, n = net.rf, prompt.size(1), 24
    tensor = torch.cat([prompt, torch.zeros(3, n, dtype=torch.int64, device=device)], 1)
    net.before_generate((prompt,), None)
    for t in range(prior, prior + n):
        tensor[:, t:t + 1] = net.generate_step((tensor[:, t - rf:t],), t=t)[0]
    net.after_generate((tensor,), None)
    assert torch.equal(tensor.cpu(), H.T(g["a_out"]))


def test_wavenet_conditioned_kernel3_teacher_forced(device):
    g = H.golden("wavenet.npz")
    assert bool(H.margin_ok(g["b_raw"]).all())
    net, sd, arch = H.wavenet_b()
    net = net.to(device)
    idx, cond, rf = H.T(g["b_idx"]).to(device), H.T(g["b_cond"]).to(device), int(g["b_rf"])
    assert net.rf == rf
    raws, picks = [], []
    for t in range(rf, idx.size(1)):
        picks.append(net.generate_step((idx[:, t - rf:t], cond[:, t - rf:t]), t=t)[0])
        raws.append(net._plan.last_logits(idx.size(0)))
    assert torch.equal(torch.cat(picks, 1).cpu(), H.T(g["b_argmax"]))
    assert torch.allclose(torch.stack(raws, 1).cpu(), H.T(g["b_raw"]), **LOGIT_TOL)


def test_wavenet_cfg2_shape_matches_reference(device):
    g = H.golden("wavenet.npz")
    assert bool(H.margin_ok(g["c_raw"]).all())
    net, _, _ = H.wavenet_c()
    out = run_loop(net, (H.T(g["c_prompt"]),), 12)
    assert torch.equal(out[0].cpu(), H.T(g["c_out"]))
    assert torch.allclose(net._plan.last_logits(2).cpu(), H.T(g["c_raw"])[:, -1], **LOGIT_TOL)


def test_wavenet_cfg2_vs_oracle_long_run(device):
    """BASELINE config 2 (10 x {1..512}, 64 ch, batch 8): 100 free-running steps -- through several
    graph replays and ring wrap-arounds of the short-dilation layers -- against the naive oracle"""
    net, sd, arch = H.wavenet_c()
    gen = torch.Generator().manual_seed(5)
    prompt = torch.randint(0, 256, (8, 1024 + 3), generator=gen)
    n = 100
    want, raw = O.wavenet_generate(sd, prompt, (), n, keep_logits=True, **arch)
    got = run_loop(net, (prompt,), n)[0].cpu()
    ok = H.margin_ok(raw.numpy())
    first_bad = (~ok).float().cumsum(1) > 0        # after a near-tie the clips may legitimately diverge
    same = got[:, prompt.size(1):] == want[:, prompt.size(1):]
    assert bool((same | first_bad).all())
    assert float(ok.float().mean()) > 0.95


def test_wavenet_sampling_matches_oracle_given_uniforms(device):
    net, sd, arch = H.wavenet_a()
    net = net.to(device)
    gen = torch.Generator().manual_seed(9)
    prompt = torch.randint(0, 256, (4, net.rf + 2), generator=gen)
    n = 30
    temp = torch.tensor([0.7, 1.0, 1.3, 0.4])
    torch.manual_seed(123)
    u = torch.rand((4, n), device=device)        # what generate_block will draw
    torch.manual_seed(123)
    tensor = torch.cat([prompt, torch.zeros(4, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((tensor,), prompt.size(1), n, temperature=temp)
    # every step of every clip: the oracle is teacher-forced on the device's history, and the device's pick must sit in the
    # CDF interval of its uniform draw (a draw within fp32 rounding of a CDF step may fall on either side of that step)
    got = tensor.cpu()
    _, raw = O.wavenet_generate(sd, prompt, (), n, keep_logits=True, forced=got, **arch)
    ok, exact = H.sampled_picks_ok(raw, temp, u.cpu(), got[:, prompt.size(1):])
    assert bool(ok.all()) and float(exact.float().mean()) > 0.97


def test_wavenet_unsupported_options_fail_loudly(device):
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(), blocks=(2,), dims_dilated=(8,), with_affine_residuals=True,
                                                     pad_side=1)).to(device).eval()      # (padding zeros would have to become aff(0))
    with pytest.raises(NotImplementedError):
        net.before_generate((torch.zeros(1, 8, dtype=torch.int64, device=device),), 0)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(), blocks=(2,), dims_dilated=(8,), stride=2)).to(device).eval()
    with pytest.raises(NotImplementedError):
        net.before_generate((torch.zeros(1, 8, dtype=torch.int64, device=device),), 0)


# ---------------------------------------------------------------------------- SampleRNN
@pytest.mark.parametrize("tag", ["gru", "lstm", "rnn"])
def test_sample_rnn_loop_matches_reference_golden(device, tag):
    g = H.golden("srnn.npz")
    raw = g[f"{tag}_raw"].reshape(3, 40, 257)
    assert bool(H.margin_ok(raw).all())
    net, _, _ = H.srnn(tag)
    out = run_loop(net, (H.T(g[f"{tag}_prompt"]),), 40, parameters=None)
    assert torch.equal(out[0].cpu(), H.T(g[f"{tag}_out"]))
    inv = run_loop(net, (H.T(g[f"{tag}_prompt"]),), 40, yield_inversed_outputs=True)
    assert inv[0].dtype == torch.float32 and inv[0].shape == (3, g[f"{tag}_prompt"].shape[1] + 40)


@pytest.mark.parametrize("tag,fused", [("gru", "1"), ("gru", "0"), ("lstm", "1")])
def test_sample_rnn_weight_norm_matches_reference_golden(device, monkeypatch, tag, fused):
    """weight_norm=True: (g, v) pairs in the state_dict, folded to g v / |v| when the plan binds them (fused tier kernels
    and one launch per op); classes bit-exact against the reference's loop"""
    import warnings
    warnings.filterwarnings("ignore")
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", fused)
    g = H.golden("srnn_wn.npz")
    assert bool(H.margin_ok(g[f"{tag}_raw"].reshape(3, 40, 257)).all())
    net, sd, _ = H.srnn(tag, weight_norm=True)
    assert "tiers.0.rnn.weight_hh_l0_g" in sd and "output_modules.0.estimator.0.fc.0.bias_v" in sd
    out = run_loop(net, (H.T(g[f"{tag}_prompt"]),), 40, parameters=None)
    assert torch.equal(out[0].cpu(), H.T(g[f"{tag}_out"]))


def test_sample_rnn_generate_step_protocol(device):
    """tests/test_sample_rnn.py:62-87 of the reference"""
    net, sd, arch = H.srnn("lstm")
    net = net.to(device)
    gen = torch.Generator().manual_seed(3)
    prompt = torch.randint(0, 256, (2, 32), generator=gen)
    o = O.SampleRNNOracle(sd, **arch)
    o.before_generate(prompt)
    want = o.generate_step(prompt[:, -o.rf:], 32)
    pd = prompt.to(device)
    for temp in (None, 0.5, (1.,)):
        net.before_generate((pd,), 0)
        out = net.generate_step((pd[:, -net.rf:],), t=32, temperature=temp)
        net.after_generate(out, 0)
        assert type(out) is tuple and out[0].shape == (2, 1) and out[0].ndim == pd.ndim
        if temp is None:
            assert torch.equal(out[0].cpu()[:, 0], want)
            assert torch.allclose(net._plan.last_logits(2).cpu(), o.last_raw, **LOGIT_TOL)


def test_sample_rnn_loop_with_temperature(device):
    """tests/test_sample_rnn.py:90-113: prompt 512, 512 new steps, batch 2, temperature (1.,) -> float (2, 1024)"""
    net, _, _ = H.srnn("lstm", frame_sizes=(16, 8, 8))
    prompt = torch.randint(0, 256, (2, 512))
    out = run_loop(net, (prompt,), 512, parameters=dict(temperature=(1.,)), yield_inversed_outputs=True)
    assert out[0].shape == (2, 1024) and out[0].dtype == torch.float32
    assert float(out[0].abs().max()) <= 1.0


@pytest.mark.parametrize("fused,frame_sizes,batch,kind", [("1", (16, 4, 1), 64, "gru"), ("0", (16, 4, 1), 64, "gru"),
                                                          ("1", (8, 4, 2), 6, "gru"), ("1", (16, 4, 1), 21, "lstm"),
                                                          ("1u", (16, 4, 1), 64, "gru"), ("1u", (16, 4, 1), 37, "lstm")])
def test_sample_rnn_cfg3_shape_vs_oracle(device, monkeypatch, fused, frame_sizes, batch, kind):
    """BASELINE config 3 geometry (frame sizes 16/4/1, GRU) at hidden 128, batch 64, prompt with P % rf != 0; with the
    fused bottom-tier kernel (several steps per launch) and with one launch per op; bottom frames of 2 samples, ragged
    last workgroup; LSTM tiers (the reference's default) through the fused tier kernel; "1u": the fused tier kernel with
    the up-sampler as its own launch instead of behind the kernel's grid barrier"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", fused[0])
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED_UP", "0" if fused.endswith("u") else "1")
    net, sd, arch = H.srnn("big", hidden=128, mlp_dim=128, seed=77, frame_sizes=frame_sizes, kind=kind)
    gen = torch.Generator().manual_seed(8)
    prompt = torch.randint(0, 256, (batch, frame_sizes[0] * 5 + 7), generator=gen)
    n = 100
    o = O.SampleRNNOracle(sd, **arch)
    want, raw = o.generate(prompt, n, keep_logits=True)
    got = run_loop(net, (prompt,), n)[0].cpu()
    ok = H.margin_ok(raw.numpy())
    first_bad = (~ok).float().cumsum(1) > 0
    same = got[:, prompt.size(1):] == want[:, prompt.size(1):]
    assert bool((same | first_bad).all())
    assert float(ok.float().mean()) > 0.9


@pytest.mark.parametrize("fused", ["1", "0"])
def test_sample_rnn_demo_geometry_vs_oracle(device, monkeypatch, fused):
    """the network of the reference's SampleRNN demo (demos/srnn.py:45-52): eight tiers with frame sizes
    (256, 128, 64, 32, 16, 8, 4, 8), LSTM, hidden 128, weight_norm=True; 300 free-running steps against the oracle"""
    import warnings
    warnings.filterwarnings("ignore")
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", fused)
    fs = (256, 128, 64, 32, 16, 8, 4, 8)
    net, sd, arch = H.srnn("demo", hidden=128, mlp_dim=128, seed=81, frame_sizes=fs, kind="lstm", weight_norm=True)
    gen = torch.Generator().manual_seed(10)
    prompt = torch.randint(0, 256, (3, 2 * 256 + 37), generator=gen)
    n = 300
    o = O.SampleRNNOracle(O.fold_weight_norm(sd), **arch)
    want, raw = o.generate(prompt, n, keep_logits=True)
    got = run_loop(net, (prompt,), n)[0].cpu()
    ok = H.margin_ok(raw.numpy())
    first_bad = (~ok).float().cumsum(1) > 0
    same = got[:, prompt.size(1):] == want[:, prompt.size(1):]
    assert bool((same | first_bad).all())
    assert float(ok.float().mean()) > 0.9


def test_sample_rnn_grid_barrier_is_deterministic(device, monkeypatch):
    """the up-sampler phase of the tier kernel reads rows that workgroups on other XCDs have just written (write-through
    stores, agent-scope loads, one grid-wide barrier): H = 512, 64 clips, the same generation five times, bit-identical,
    and identical to the path with the up-sampler as a separate launch"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", "1")
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_RESIDENT", "0")      # (the tier kernel of the launch path is what is tested here)
    gen = torch.Generator().manual_seed(12)
    prompt = torch.randint(0, 256, (64, 64), generator=gen).to(device)
    outs = []
    for fused_up in ("1", "1", "1", "1", "1", "0"):
        monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED_UP", fused_up)
        net, _, _ = H.srnn("big", hidden=512, mlp_dim=128, seed=79, frame_sizes=(16, 4, 1), kind="gru")
        net = net.to(device)
        idx = torch.cat([prompt, torch.zeros(64, 320, dtype=torch.int64, device=device)], 1)
        net.before_generate((prompt,), None)
        net.generate_block((idx,), 64, 320)
        net.after_generate((idx,), None)
        outs.append(idx.cpu())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def _assert_resident(net, blocks):
    """the generate blocks meant to run as ONE resident launch (csrc/srnn_resident.hip) did: the mode needs nothing but a CU per workgroup"""
    assert net._plan.resident_blocks() == blocks, (net._plan.resident_blocks(), blocks)


def _srnn_blocks(net, device, prompt, n, parts, **params):
    B, P = prompt.shape
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx[:, :P],), None)
    t = P
    for nb in parts:
        net.generate_block((idx,), t, nb, **params)
        t += nb
    count = net._plan.resident_blocks()
    net.after_generate((idx,), None)
    return idx.cpu(), count


def test_sample_rnn_warmup_as_one_resident_launch(device, monkeypatch):
    """the warm-up over the prompt (before_generate, sample_rnn_v2.py:226-234) as ONE teacher-forced resident launch - the tiers pace each other by their
    progress words instead of by drawn classes - against the same warm-up with one launch per tier update: the generation that follows is bit-identical,
    with a prompt that is not a multiple of rf (the shifted window), GRU and LSTM tiers, a ragged row tile; and equal to the oracle"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", "1")
    for kind, hidden, fs, B in (("gru", 512, (16, 4, 1), 40), ("lstm", 128, (32, 8, 2), 21), ("gru", 128, (4, 1), 5)):
        rf = fs[0]
        P, n = 5 * rf + 7, 2 * rf + 3
        prompt = torch.randint(0, 256, (B, P), generator=torch.Generator().manual_seed(B))
        outs = []
        for flag in ("1", "0"):
            monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_RESIDENT_WARMUP", flag)
            net, sd, arch = H.srnn("big", hidden=hidden, mlp_dim=64, seed=77, frame_sizes=fs, kind=kind)
            net = net.to(device)
            idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
            net.before_generate((idx[:, :P],), None)
            assert net._plan.resident_warmups() == (1 if flag == "1" else 0), (kind, flag)
            net.generate_block((idx,), P, n)
            net.after_generate((idx,), None)
            outs.append(idx.cpu())
        assert torch.equal(outs[0], outs[1]), kind
        ref, raw = O.SampleRNNOracle(sd, **arch).generate(prompt, n, keep_logits=True, forced=outs[0])
        ok = H.margin_ok(raw)
        assert float(ok.float().mean()) > 0.9 and torch.equal(ref[:, P:][ok], outs[0][:, P:][ok])


@pytest.mark.parametrize("kind", ["gru", "lstm"])
def test_sample_rnn_resident_mode_blocks_and_oracle(device, monkeypatch, kind):
    """resident mode (every tier, the bottom tier and the head as ONE launch per block, weights in registers): blocks that start between two
    updates of the top tier (their first steps run with the kernels in turns, the launch takes over at the next multiple of frame_sizes[0] -
    the state passes from one to the other and back), a block too short for the mode in the middle, the same generation in one block and with
    the mode switched off - each equal to the oracle teacher-forced on its own history wherever the oracle's pick is clear"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", "1")
    gen = torch.Generator().manual_seed(31)
    B, P = 21, 48                                   # a ragged last row tile
    prompt = torch.randint(0, 256, (B, P), generator=gen)
    # 48 + 37 = 85: the third block starts 5 steps into a period (11 steps in turns, then 32 resident); 5 < frame_sizes[0]: launch path;
    # the fourth block (16 steps from 133) has 5 steps left after its head: launch path; the fifth 59 steps from 149: 11 in turns + 48 resident
    splits = [((37, 5, 43, 16, 59), 3), ((160,), 1)]
    for env, (parts, want_resident) in (("1", splits[0]), ("1", splits[1]), ("0", (splits[1][0], 0))):
        monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_RESIDENT", env)
        net, sd, arch = H.srnn("big", hidden=256, mlp_dim=128, seed=83, frame_sizes=(16, 4, 1), kind=kind)
        net = net.to(device)
        got, count = _srnn_blocks(net, device, prompt, 160, parts)
        assert count == want_resident, (env, parts, count)
        ref, raw = O.SampleRNNOracle(sd, **arch).generate(prompt, 160, keep_logits=True, forced=got)
        ok = H.margin_ok(raw)
        assert float(ok.float().mean()) > 0.9
        assert torch.equal(ref[:, P:][ok], got[:, P:][ok]), (env, parts)


@pytest.mark.parametrize("frame_sizes,batch,hidden,kind", [
    ((4, 1), 3, 128, "gru"),            # one recurrent tier above the bottom
    ((32, 8, 2), 33, 128, "gru"),       # a bottom frame of 2 samples, a ragged third row tile
    ((64, 16, 4, 4), 5, 128, "lstm"),   # three recurrent tiers, frame sizes above 16 and equal ones
    ((16, 4, 1), 16, 256, "lstm"),
    ((16, 8, 8), 2, 256, "lstm"),       # BASELINE config 1's tiers: a bottom frame of 8 samples
    ((16, 4, 1), 40, 512, "gru"),       # BASELINE config 3's tiers, row tiles of 32 clips (the second one ragged)
    ((8, 2, 1), 70, 128, "gru"),        # more clips than two row tiles of 32
])
def test_sample_rnn_resident_mode_geometries(device, monkeypatch, frame_sizes, batch, hidden, kind):
    """resident mode over the tier geometries the kernel accepts: classes equal to the oracle's, teacher-forced on the device's own
    history; the mode itself must have run"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", "1")
    net, sd, arch = H.srnn("big", hidden=hidden, mlp_dim=64, seed=91, frame_sizes=frame_sizes, kind=kind)
    net = net.to(device)
    rf = frame_sizes[0]
    P, n = 2 * rf + 3, 3 * rf + 5                     # a prompt that is not a multiple of rf (the warm-up window shift)
    gen = torch.Generator().manual_seed(17)
    prompt = torch.randint(0, 256, (batch, P), generator=gen)
    got, count = _srnn_blocks(net, device, prompt, n, (n,))
    assert count == 1
    o = O.SampleRNNOracle(sd, **arch)
    ref, raw = o.generate(prompt, n, keep_logits=True, forced=got)
    ok = H.margin_ok(raw)
    assert float(ok.float().mean()) > 0.9
    assert torch.equal(ref[:, P:][ok], got[:, P:][ok])


@pytest.mark.parametrize("batch,want_resident", [(170, 1), (220, 0)])
def test_sample_rnn_one_tier_many_clips(device, monkeypatch, batch, want_resident):
    """ONE recurrent tier (it is the top AND the last tier) with more clips than row tiles of 16 fit the chip: the tier's workgroups take two
    row tiles (170 clips) - never four, a last tier has no such role: at 220 clips the launch does not fit and the kernels run in turns.
    Every clip, the ones of the last row tile included, equals the oracle teacher-forced on the device's history"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", "1")
    net, sd, arch = H.srnn("big", hidden=128, mlp_dim=64, seed=97, frame_sizes=(4, 1), kind="gru")
    net = net.to(device)
    P, n = 11, 17
    prompt = torch.randint(0, 256, (batch, P), generator=torch.Generator().manual_seed(19))
    got, count = _srnn_blocks(net, device, prompt, n, (n,))
    assert count == want_resident
    ref, raw = O.SampleRNNOracle(sd, **arch).generate(prompt, n, keep_logits=True, forced=got)
    ok = H.margin_ok(raw)
    assert float(ok.float().mean()) > 0.9
    assert torch.equal(ref[:, P:][ok], got[:, P:][ok])
    assert bool((got[-16:, P:] != 0).any())          # (the last row tile was generated at all)


def test_sample_rnn_resident_mode_last_logits_and_state_hand_back(device, monkeypatch):
    """what a resident launch leaves behind is what the kernels in turns would have left: the logits of the block's last step, and the tiers'
    states, counters and up-sampled rows - a block that ends between two updates of every tier, continued step by step through generate_step
    (the launch path), against the same generation with the mode switched off: logits within the tolerance of the two associations"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", "1")
    gen = torch.Generator().manual_seed(23)
    B, P, n = 6, 32, 39                               # 32 + 39 = 71 = 4 * 16 + 7: mid-period, mid-frame
    prompt = torch.randint(0, 256, (B, P), generator=gen)
    logits = []
    for env in ("1", "0"):
        monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_RESIDENT", env)
        net, sd, arch = H.srnn("big", hidden=128, mlp_dim=128, seed=85, frame_sizes=(16, 4, 1), kind="gru")
        net = net.to(device)
        idx = torch.cat([prompt, torch.zeros(B, n + 9, dtype=torch.int64)], 1).to(device)
        net.before_generate((idx[:, :P],), None)
        net.generate_block((idx,), P, n)
        assert net._plan.resident_blocks() == (1 if env == "1" else 0)
        rows = [net._plan.last_logits(B).cpu()]
        for t in range(P + n, P + n + 9):             # the launch path, one step at a time, on the state the block left
            out = net.generate_step((idx[:, t - 16:t],), t=t)
            idx[:, t] = out[0][:, 0]
            rows.append(net._plan.last_logits(B).cpu())
        net.after_generate((idx,), None)
        logits.append((torch.stack(rows), idx.cpu()))
    (la, ia), (lb, ib) = logits
    same = (ia == ib).all(1)                          # clips whose two histories agree to the end (a near-tie may part them)
    assert float(same.float().mean()) > 0.6
    torch.testing.assert_close(la[:, same], lb[:, same], rtol=1e-4, atol=2e-4)


def test_sample_rnn_timeout_is_redone_in_turns(device, monkeypatch):
    """a timed-out wait of the resident mode (injected through mmk_srnn_inject_sync_error) must not return invalid samples: the batch
    is regenerated with the kernels in turns, with a warning, and equals an undisturbed generation"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", "1")
    net, sd, arch = H.srnn("big", hidden=128, mlp_dim=64, seed=93, frame_sizes=(16, 4, 1), kind="gru")
    net = net.to(device)
    prompt = torch.randint(0, 256, (5, 32), generator=torch.Generator().manual_seed(3))

    def generate():
        idx = torch.cat([prompt, torch.zeros(5, 80, dtype=torch.int64)], 1).to(device)
        net.before_generate((idx[:, :32],), None)
        net.generate_block((idx,), 32, 80)
        net.after_generate((idx,), None)
        return idx.cpu()

    want = generate()
    idx = torch.cat([prompt, torch.zeros(5, 80, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx[:, :32],), None)
    net.generate_block((idx,), 32, 80)
    net._plan.inject_sync_error()             # as if a wait of the resident kernels had timed out (include/mmk.h: fault injection)
    with pytest.warns(UserWarning, match="in turns"):
        net.after_generate((idx,), None)
    assert torch.equal(idx.cpu(), want)
    assert torch.equal(generate(), want)      # and the next generation is an ordinary one again


def test_sample_rnn_resident_mode_sampled_decode_and_reuse(device, monkeypatch):
    """resident mode with temperatures (uniforms indexed by the absolute step) and a second generation on the same plan
    (the granules of the first one must not satisfy the second one's waits)"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", "1")
    net, sd, arch = H.srnn("big", hidden=128, mlp_dim=128, seed=84, frame_sizes=(16, 4, 1), kind="gru")
    net = net.to(device)
    o = O.SampleRNNOracle(sd, **arch)
    B, P, n = 7, 32, 96
    temp = torch.tensor([0.6, 1.0, 1.4, 0.8, 1.1, 0.9, 1.2])
    for round_ in range(2):
        gen = torch.Generator().manual_seed(40 + round_)
        prompt = torch.randint(0, 256, (B, P), generator=gen)
        torch.manual_seed(50 + round_)
        u = torch.rand((B, n), device=device)
        torch.manual_seed(50 + round_)
        idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
        net.before_generate((idx[:, :P],), None)
        net.generate_block((idx,), P, n, temperature=temp)
        net.after_generate((idx,), None)
        assert net._plan.resident_blocks() == round_ + 1
        got = idx.cpu()
        _, raw = o.generate(prompt, n, keep_logits=True, forced=got)
        ok, exact = H.sampled_picks_ok(raw, temp, u.cpu(), got[:, P:])
        assert ok.all()
        assert float(exact.float().mean()) > 0.97


def test_sample_rnn_fused_bottom_sampled_decode(device, monkeypatch):
    """temperature sampling through the fused bottom kernel against the oracle's inverse-CDF draw for the same uniforms"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_SRNN_FUSED", "1")
    net, sd, arch = H.srnn("big", hidden=128, mlp_dim=128, seed=78, frame_sizes=(16, 4, 1), kind="gru")
    net = net.to(device)
    gen = torch.Generator().manual_seed(9)
    B, P, n = 5, 48, 60
    prompt = torch.randint(0, 256, (B, P), generator=gen)
    temp = torch.tensor([0.6, 1.0, 1.4, 0.8, 1.1])
    torch.manual_seed(5)
    u = torch.rand((B, n), device=device)
    torch.manual_seed(5)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx[:, :P],), None)
    net.generate_block((idx,), P, n, temperature=temp)
    net.after_generate((idx,), None)
    o = O.SampleRNNOracle(sd, **arch)
    got = idx.cpu()
    _, raw = o.generate(prompt, n, keep_logits=True, forced=got)      # teacher-forced on the device's history
    ok, exact = H.sampled_picks_ok(raw, temp, u.cpu(), got[:, P:])
    assert bool(ok.all()) and float(exact.float().mean()) > 0.97


# ---------------------------------------------------------------------------- Seq2Seq
def test_seq2seq_matches_reference_golden(device):
    g = H.golden("s2s.npz")
    net, sd = H.s2s_tiny()
    net = net.to(device)
    x = H.T(g["x"]).to(device)
    y = net.generate_step((x,), t=4)
    assert isinstance(y, torch.Tensor) and y.shape == x.shape
    scale = float(np.abs(g["y"]).max())
    assert float((y.cpu() - H.T(g["y"])).abs().max()) <= 1e-4 * scale
    assert torch.allclose(net((x,)).cpu(), y.cpu())          # eval forward is the same path
    out = run_loop(net, (H.T(g["prompt"]),), 10)
    assert float((out[0].cpu() - H.T(g["out"])).abs().max()) <= 1e-3 * float(np.abs(g["out"]).max())
    assert bool((out[0][:, -10:] != 0).all())                 # tests/test_seq2seq.py:146


@pytest.mark.parametrize("tag", list(H.S2S_STACKS))
def test_seq2seq_lstm_stacks(device, tag):
    """2 / 3 stacked bi-LSTMs per side, with and without residuals: golden from the reference, a larger batch against
    the oracle, and the loop; fp32 tolerance 1e-4 of the largest output"""
    import warnings
    warnings.filterwarnings("ignore")
    g = H.golden("s2s_stacks.npz")
    kw = H.S2S_STACKS[tag]
    net, sd = H.s2s_tiny(seed=43, **kw)
    net.to(device)
    y = net.generate_step((H.T(g["x"]).to(device),), t=4).cpu()
    want = H.T(g[f"y_{tag}"])
    assert float((y - want).abs().max()) <= 1e-4 * float(want.abs().max())
    okw = dict(downsampling=kw.get("enc_downsampling", "edge_sum"), enc_residuals=kw.get("enc_apply_residuals", False),
               dec_residuals=kw.get("dec_apply_residuals", False))
    x = torch.rand(21, 4, 65, generator=torch.Generator().manual_seed(7))
    want = O.s2s_step(O.fold_weight_norm(sd), x, hop=4, **okw)      # (the decoder is weight-normed when dec_apply_residuals, :221)
    got = net.generate_step((x.to(device),), t=4).cpu()
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    out = run_loop(net, (x[:2].to(device),), 8)[0]
    assert out.shape == (2, 12, 65) and bool(torch.isfinite(out).all())


def test_seq2seq_block_with_a_clipped_last_step_through_the_tiled_gemm(device):
    """20 clips x hop 8 = 160 rows: the output projection takes the tiled GEMM (27 tiles: K split eight ways) and scatters its
    rows into the caller's strided tensor itself; 19 new frames = two whole steps and one clipped to 3 frames, whose other 5
    rows must not be written anywhere (the frames behind the tensor's end belong to the next clip)"""
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=128, hop=8)).eval()
    from oracle.weights import load_recipe
    sd = load_recipe(net, seed=59, gain=1.5)
    net.to(device)
    B, P, n = 20, 8, 19
    prompt = torch.rand(B, P, 513, generator=torch.Generator().manual_seed(15))
    want = O.s2s_generate(sd, prompt, n, hop=8)
    flat = torch.full((B * (P + n) + 8, 513), -7.0)              # clips back to back, a guard of 8 frames behind the last one
    frames = flat[:B * (P + n)].view(B, P + n, 513)
    frames[:, :P] = prompt
    frames[:, P:] = 0
    dev = flat.to(device)
    dframes = dev[:B * (P + n)].view(B, P + n, 513)
    net.before_generate((dframes[:, :P],), 0)
    assert net.generate_block((dframes,), P, n) is True
    net.after_generate((dframes,), 0)
    got = dframes.cpu()
    assert float((got - want).abs().max()) <= 2e-4 * float(want.abs().max())
    assert bool((dev[B * (P + n):] == -7.0).all())                # nothing behind the last clip
    assert torch.equal(got[:, :P], prompt)                        # a clipped step of clip b must not reach into clip b + 1's prompt


def test_seq2seq_adds_up_several_continuous_inputs(device):
    """`input_module = sum` for continuous inputs (s2s_lstm_v2.py:202-204): two frame inputs are added in front of the encoder; the
    loop writes the one output into the first tensor and leaves the second as the dataloader gave it (loops/generate.py:214-219)"""
    io1 = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(n_fft=128, hop_length=32))
    io = mmk.IOSpec(inputs=(io1.inputs[0], io1.inputs[0]), targets=io1.targets)
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=32, hop=4)).eval()
    from oracle.weights import load_recipe
    sd = load_recipe(net, seed=41, gain=1.5)
    net.to(device)
    g = torch.Generator().manual_seed(12)
    a, b = torch.rand(5, 4, 65, generator=g), torch.rand(5, 4, 65, generator=g)
    want = O.s2s_step(sd, a + b, hop=4)
    got = net.generate_step((a.to(device), b.to(device)), t=4).cpu()
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    pa, pb = torch.rand(2, 6, 65, generator=g), torch.rand(2, 6, 65, generator=g)
    out = run_loop(net, (pa, pb), 8)
    assert len(out) == 2 and out[0].shape == (2, 14, 65)
    first = O.s2s_step(sd, (pa + pb)[:, -4:], hop=4)
    assert float((out[0][:, 6:10].cpu() - first).abs().max()) <= 1e-4 * float(first.abs().max())
    assert bool((out[1][:, 6:].cpu() == 0).all()) and torch.equal(out[1][:, :6].cpu(), pb)


@pytest.mark.parametrize("ksplit", ["1", "2", "3", "8"])
def test_seq2seq_gemm_split_k(device, monkeypatch, ksplit):
    """the tiled GEMM with K cut over 1 / 2 / 3 / 8 workgroups per tile (uneven stage ranges, K = 513 with a ragged last stage) and
    the partial sums added in split order: against the oracle at 1e-4 of the largest output, and bit-identical from run to run"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_GEMM_KSPLIT", ksplit)
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=128, hop=8)).eval()
    from oracle.weights import load_recipe
    sd = load_recipe(net, seed=53, gain=1.5)
    net.to(device)
    x = torch.rand(21, 8, 513, generator=torch.Generator().manual_seed(9))
    want = O.s2s_step(sd, x, hop=8)
    got = net.generate_step((x.to(device),), t=8)
    again = net.generate_step((x.to(device),), t=8)
    assert float((got.cpu() - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert torch.equal(got, again)


@pytest.mark.parametrize("tag", list(H.S2S_MULAW))
def test_seq2seq_on_class_indices_matches_reference_golden(device, tag):
    """IOSpec.mulaw_io with an embedding input (tests/test_seq2seq.py:149-154): class indices in, the MLP head's argmax out, as
    a float tensor from generate_step and in place in the loop's int64 tensor; classes exact (the fixture's smallest top-1 /
    top-2 gap is above helpers.margin_ok's bound), raw head outputs rtol 1e-4 / atol 2e-4"""
    import warnings
    warnings.filterwarnings("ignore")
    g = H.golden("s2s_mulaw.npz")
    net, sd, hop, arch = H.s2s_mulaw(tag)
    net.to(device)
    x = H.T(g[f"{tag}_x"]).to(device)
    y = net.generate_step((x,), t=hop)
    assert y.dtype == torch.float32 and y.shape == x.shape
    raw = net._plan.last_logits(x.size(0)).cpu()
    assert torch.allclose(raw, H.T(g[f"{tag}_raw"]), rtol=1e-4, atol=2e-4)
    assert bool(H.margin_ok(g[f"{tag}_raw"]).all()) and torch.equal(y.cpu(), H.T(g[f"{tag}_y"]))
    assert torch.equal(net((x,)).cpu(), y.cpu())              # eval forward is the same path
    assert bool(H.margin_ok(g[f"{tag}_loop_raw"]).all())
    out = run_loop(net, (H.T(g[f"{tag}_prompt"]),), 10)
    assert out[0].dtype == torch.int64 and torch.equal(out[0].cpu(), H.T(g[f"{tag}_out"]))


@pytest.mark.parametrize("hop,batch,model_dim,mlp_dim,n_mlp", [(8, 24, 128, 128, 0), (5, 7, 64, 48, 1), (4, 40, 256, 128, 3)])
def test_seq2seq_on_class_indices_vs_oracle(device, hop, batch, model_dim, mlp_dim, n_mlp):
    """wider nets (batch x hop above and below the tiled GEMM's 128 rows, ragged tiles, a head width that is no multiple of
    16), every step checked against the oracle on the history the device produced: raw head outputs rtol 1e-4 / atol 2e-4,
    classes exact wherever the oracle's top-1 / top-2 gap is above 5e-5; the in-place block call on a strided tensor whose
    length is not a multiple of hop (the last step is clipped)"""
    import warnings
    warnings.filterwarnings("ignore")
    H.S2S_MULAW["_t"] = dict(hop=hop, io=dict(n_mlp_layers=n_mlp))
    try:
        net, sd, hop, arch = H.s2s_mulaw("_t", model_dim=model_dim, mlp_dim=mlp_dim)
    finally:
        del H.S2S_MULAW["_t"]
    net.to(device)
    g = torch.Generator().manual_seed(17)
    n = 3 * hop - 1
    wide = torch.zeros(batch, 2 * (hop + n), dtype=torch.int64)
    seq = wide[:, ::2]                                        # element stride 2
    seq[:, :hop] = torch.randint(0, 256, (batch, hop), generator=g)
    dev = wide.to(device)
    dseq = dev[:, ::2]
    net.before_generate((dseq[:, :hop],), 0)
    assert net.generate_block((dseq,), hop, n) is True
    net.after_generate((dseq,), 0)
    got = dseq.cpu()
    assert bool((dev[:, 1::2] == 0).all())                    # nothing written between the elements
    fsd = O.fold_weight_norm(sd)
    checked = 0
    for t in range(hop, hop + n, hop):
        want, raw = O.s2s_step(fsd, got[:, t - hop:t], hop, return_raw=True, **arch)
        k = min(hop, hop + n - t)
        ok = H.margin_ok(raw)[:, :k]
        assert bool((got[:, t:t + k][ok] == want[:, :k].long()[ok]).all())
        checked += int(ok.sum())
    assert checked > 0.9 * batch * n
    # one step through generate_step on the last window, its raw outputs against the oracle's
    x = got[:, -hop:].contiguous()
    y = net.generate_step((x.to(device),), t=hop)
    want, raw = O.s2s_step(fsd, x, hop, return_raw=True, **arch)
    assert torch.allclose(net._plan.last_logits(batch).cpu(), raw, rtol=1e-4, atol=2e-4)
    ok = H.margin_ok(raw)
    assert torch.equal(y.cpu()[ok], want[ok])


def test_seq2seq_on_class_indices_eval_forward_with_a_temperature(device, monkeypatch):
    """`net(x, temperature=T)` in eval mode samples (s2s_lstm_v2.py:246-253 hands the temperature to the sampler; generate_step does
    not): every draw lies in the CDF interval of softmax(logits / T) that its uniform selects (helpers.sampled_picks_ok), for a
    scalar and for one temperature per clip"""
    import warnings
    warnings.filterwarnings("ignore")
    net, sd, hop, arch = H.s2s_mulaw("mlp0")
    net.to(device)
    x = torch.randint(0, 256, (6, hop), generator=torch.Generator().manual_seed(4))
    _, raw = O.s2s_step(O.fold_weight_norm(sd), x, hop, return_raw=True, **arch)
    drawn = {}
    real_rand = torch.rand

    def rand(*a, **k):
        drawn["u"] = real_rand(*a, **k)
        return drawn["u"]

    monkeypatch.setattr(torch, "rand", rand)
    for temp in (0.7, torch.tensor([0.5, 1.0, 2.0, 0.25, 1.5, 1.0])):
        y = net((x.to(device),), temperature=temp)
        assert y.dtype == torch.float32 and y.shape == (6, hop)
        t = torch.as_tensor(temp, dtype=torch.float32).reshape(-1).expand(6) if not isinstance(temp, torch.Tensor) else temp
        ok, exact = H.sampled_picks_ok(raw, t, drawn["u"].cpu().reshape(6, hop), y.cpu().long())
        assert bool(ok.all())
    assert len({float(v) for v in y.cpu().flatten()}) > 1


def test_seq2seq_timeout_during_a_sampled_forward_keeps_the_logits(device, monkeypatch):
    """a timed-out wait reported right after a `net(x, temperature=T)` call: the call is repeated with one launch per frame AND the
    draw that follows reads the repeated call's logits (the one-launch-per-frame plan must outlive the repeat); the next call runs
    resident again on a fresh plan"""
    H.S2S_MULAW["_t"] = dict(hop=4, io=dict(n_mlp_layers=0))
    try:
        net, sd, hop, arch = H.s2s_mulaw("_t", model_dim=128, mlp_dim=32)
    finally:
        del H.S2S_MULAW["_t"]
    net.to(device)
    x = torch.randint(0, 256, (6, hop), generator=torch.Generator().manual_seed(14))
    _, raw = O.s2s_step(O.fold_weight_norm(sd), x, hop, return_raw=True, **arch)
    net.generate_step((x.to(device),), t=hop)                   # builds the plan
    assert net._plan.resident_launches() > 0
    first_plan = net._plan
    real_step = first_plan.step_classes

    def failing_step(xx):
        y = real_step(xx)
        first_plan.inject_sync_error()
        return y

    monkeypatch.setattr(first_plan, "step_classes", failing_step)
    drawn = {}
    real_rand = torch.rand

    def rand(*a, **k):
        drawn["u"] = real_rand(*a, **k)
        return drawn["u"]

    monkeypatch.setattr(torch, "rand", rand)
    temp = torch.tensor([0.5, 1.0, 2.0, 0.25, 1.5, 1.0])
    with pytest.warns(UserWarning, match="per frame"):
        y = net((x.to(device),), temperature=temp)
    assert net._plan is not first_plan and net._plan_stale
    ok, exact = H.sampled_picks_ok(raw, temp, drawn["u"].cpu().reshape(6, hop), y.cpu().long())
    assert bool(ok.all())
    assert torch.allclose(net._plan.last_logits(6).cpu(), raw, rtol=1e-4, atol=2e-4)
    y2 = net.generate_step((x.to(device),), t=hop)               # a fresh resident plan
    assert not net._plan_stale and net._plan.resident_launches() > 0
    want, _ = O.s2s_step(O.fold_weight_norm(sd), x, hop, return_raw=True, **arch)
    okm = H.margin_ok(raw)
    assert torch.equal(y2.cpu()[okm], want[okm])


def test_seq2seq_class_and_frame_entry_points_do_not_mix(device):
    """a plan for class indices refuses frames and the other way round (the C-ABI's error, not a crash)"""
    net, sd, hop, arch = H.s2s_mulaw("mlp0")
    net.to(device)
    net.generate_step((torch.randint(0, 256, (2, hop)).to(device),), t=hop)
    with pytest.raises(Exception, match="class indices"):
        net._plan.step(torch.rand(2, hop, 32, device=device))
    fnet, _ = H.s2s_tiny()
    fnet.to(device)
    fnet.generate_step((torch.rand(2, 4, 65, device=device),), t=4)
    with pytest.raises(Exception, match="takes frames"):
        fnet._plan.step_classes(torch.randint(0, 256, (2, 4), device=device))


@pytest.mark.parametrize("ds,us", H.S2S_VARIANTS)
def test_seq2seq_pooling_and_upsampling_variants(device, ds, us):
    """enc_downsampling edge_mean / sum / mean and dec_upsampling repeat (no up-sampling weights): golden from the
    reference, and a larger batch against the oracle; fp32 tolerance 1e-4 of the largest output"""
    g = H.golden("s2s_variants.npz")
    net, sd = H.s2s_tiny(ds, us)
    net.to(device)
    y = net.generate_step((H.T(g["x"]).to(device),), t=4).cpu()
    want = H.T(g[f"y_{ds}_{us}"])
    assert float((y - want).abs().max()) <= 1e-4 * float(want.abs().max())
    x = torch.rand(37, 4, 65, generator=torch.Generator().manual_seed(6))
    want = O.s2s_step(sd, x, hop=4, downsampling=ds, upsampling=us)
    got = net.generate_step((x.to(device),), t=4).cpu()
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("fused,hop,batch", [("1", 8, 6), ("0", 8, 6), ("1", 5, 19), ("1", 8, 33), ("0", 3, 45)])
def test_seq2seq_cfg5_geometry_vs_oracle(device, monkeypatch, fused, hop, batch):
    """magspec_io(22050, 1024, 256) -> 513 bins, hop 8, model_dim 128 (cfg 5 at reduced width), batch 6; with the fused
    LSTM time-step kernel (both directions per launch) and with one launch per op; an odd hop (state ends in the
    second buffer) with a ragged row tile; 264 and 135 rows (batch x hop >= 128) take the tiled GEMM for the
    input-to-hidden and output projections, with a ragged last row tile and K = 513"""
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_S2S_FUSED", fused)
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=128, hop=hop)).eval()
    from oracle.weights import load_recipe
    sd = load_recipe(net, seed=99, gain=1.5)
    x = torch.rand(batch, hop, 513, generator=torch.Generator().manual_seed(4))
    want = O.s2s_step(sd, x, hop=hop)
    got = net.to(device).generate_step((x.to(device),), t=hop).cpu()
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    if hop == 8:
        loop_cfg = mmk.GenerateLoopV2.Config(output_duration_sec=1.0)
        assert mmk.GenerateLoopV2.get_n_steps(loop_cfg, net) == 84


@pytest.mark.parametrize("dim,hop,batch,layers", [(128, 2, 3, 1), (128, 8, 16, 2), (256, 5, 17, 1), (256, 8, 40, 2), (512, 3, 33, 1),
                                                  (512, 8, 64, 1), (128, 7, 128, 1)])
def test_seq2seq_resident_bilstm_kernel_vs_oracle(device, monkeypatch, dim, hop, batch, layers):
    """the resident bi-LSTM kernel (csrc/lstm_seq.hip: all frames of a layer in one launch, W_hh in registers, the state exchanged
    through poison-checked images) over its geometry: every register-resident width below 1024 (cfg 5 covers that), one and two
    16-row blocks per workgroup, a ragged last block, one and several row halves, 2 .. 8 frames, stacked layers (the decoder's
    start from the encoder's final state) - three chained generate_steps against the oracle and against the per-frame kernel"""
    for k in ("MMK_S2S_FUSED", "MMK_S2S_SEQ"):
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(n_fft=128, hop_length=32))
    cfg = mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=dim, hop=hop, enc_n_lstm=layers, dec_n_lstm=layers)
    net = mmk.Seq2SeqLSTMNetwork.from_config(cfg).eval()
    from oracle.weights import load_recipe
    sd = load_recipe(net, seed=7 + dim + hop, gain=1.5)
    net.to(device)
    prompt = torch.rand(batch, hop, 65, generator=torch.Generator().manual_seed(batch))
    want = O.s2s_generate(sd, prompt, 3 * hop, hop=hop)

    def generate():
        frames = torch.cat([prompt, torch.zeros(batch, 3 * hop, 65)], 1).to(device)
        net.before_generate((frames[:, :hop],), None)
        assert net.generate_block((frames,), hop, 3 * hop)
        launches = net._plan.resident_launches()
        net.after_generate((frames,), None)
        return frames.cpu(), launches

    got, launches = generate()
    assert launches == 3 * 2 * layers
    assert float((got - want).abs().max()) <= 2e-4 * float(want.abs().max())
    again, _ = generate()
    assert torch.equal(again, got)                    # the exchange images alternate between launches: a second pass sees the same
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_S2S_SEQ", "0")
    net._plan = None
    per_frame, launches = generate()
    assert launches == 0
    assert float((got - per_frame).abs().max()) <= 1e-5 * float(want.abs().max())


def test_seq2seq_timeout_is_redone_frame_by_frame(device, monkeypatch):
    """a timed-out wait inside the resident bi-LSTM kernel (injected through mmk_s2s_inject_sync_error) must not return invalid
    frames: the blocks of the generation are run again with one launch per frame, with a warning"""
    for k in ("MMK_S2S_FUSED", "MMK_S2S_SEQ"):
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(n_fft=128, hop_length=32))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=128, hop=4)).eval()
    from oracle.weights import load_recipe
    sd = load_recipe(net, seed=71, gain=1.5)
    net.to(device)
    prompt = torch.rand(9, 4, 65, generator=torch.Generator().manual_seed(2))
    want = O.s2s_generate(sd, prompt, 8, hop=4)
    frames = torch.cat([prompt, torch.zeros(9, 8, 65)], 1).to(device)
    net.before_generate((frames[:, :4],), None)
    assert net.generate_block((frames,), 4, 8)
    assert net._plan.resident_launches() == 4
    frames[:, 4:] = -3.0                      # what a timed-out launch may leave behind
    net._plan.inject_sync_error()
    with pytest.warns(UserWarning, match="per frame"):
        net.after_generate((frames,), None)
    assert float((frames.cpu() - want).abs().max()) <= 2e-4 * float(want.abs().max())
    assert net._plan_stale                    # the next generation starts on a fresh (resident) plan
    x = torch.rand(9, 4, 65, generator=torch.Generator().manual_seed(3))
    got = net.generate_step((x.to(device),), t=4).cpu()
    assert net._plan.resident_launches() == 2
    assert float((got - O.s2s_step(sd, x, hop=4)).abs().max()) <= 1e-4 * float(got.abs().max())


def test_seq2seq_frames_in_place_ignore_what_lies_behind_a_row(device):
    """the first encoder layer reads the caller's frames where they lie, in 16-float chunks: with 65 bins the last chunk of a row
    covers 15 floats that belong to whatever follows the row in memory.  Non-finite data there (here: NaN in the padding columns
    of a strided view) must not reach any gate"""
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(n_fft=128, hop_length=32))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=128, hop=4)).eval()
    from oracle.weights import load_recipe
    sd = load_recipe(net, seed=72, gain=1.5)
    net.to(device)
    x = torch.rand(9, 4, 65, generator=torch.Generator().manual_seed(5))
    want = O.s2s_step(sd, x, hop=4)
    wide = torch.full((9, 4, 80), float("nan"), device=device)
    wide[:, :, :65] = x.to(device)
    view = wide[:, :, :65]
    assert view.stride(2) == 1 and not view.is_contiguous()
    got = net.generate_step((view,), t=4).cpu()
    assert net._plan.resident_launches() > 0          # (the resident path: the one that reads in place)
    assert bool(torch.isfinite(got).all())
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_wavenet_pad_side_1(device):
    """pad_side=1: golden from the reference's loop and eval forward (classes bit-exact, as for pad_side=0)"""
    g = H.golden("wavenet_pad1.npz")
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(mlp_dim=32), blocks=(3, 2), dims_dilated=(16,),
                                                     residuals_dim=16, skips_dim=16, pad_side=1)).eval()
    from oracle.weights import load_recipe
    load_recipe(net, seed=11, gain=2.0)
    net.to(device)
    prompt = H.T(g["prompt"]).to(device)
    out = run_loop(net, (prompt,), 24)[0].cpu()
    assert torch.equal(out, H.T(g["out"]))
    assert torch.equal(net((prompt,))[0].cpu(), H.T(g["forward_last"]))
    with pytest.raises(NotImplementedError):
        net((prompt[:, :5],))


@pytest.mark.parametrize("tag", list(H.FREQNET_CASES))
def test_wavenet_on_magnitude_frames(device, tag):
    """FreqNet (demos/freqnet.py:34-63) at reduced size: frames in, frames out, no residual / skip path, groups 1 / 4 / 2
    + Abs.  Golden from the reference's loop; fp32 tolerance 1e-4 of the largest output."""
    g = H.golden("freqnet.npz")
    net, sd, arch = H.freqnet(tag)
    net.to(device)
    prompt = H.T(g[f"{tag}_prompt"])
    out = run_loop(net, (prompt.to(device),), 6)[0].cpu()
    want = H.T(g[f"{tag}_out"])
    assert out.shape == want.shape
    assert float((out - want).abs().max()) <= 1e-4 * float(want.abs().max())
    # a longer free run against the oracle, batch 5
    gen = torch.Generator().manual_seed(3)
    p2 = torch.rand(5, net.rf + 9, 33, generator=gen)
    want2 = O.wavenet_generate_frames(sd, p2, 20, **arch)
    out2 = run_loop(net, (p2.to(device),), 20)[0].cpu()
    assert float((out2 - want2).abs().max()) <= 1e-4 * float(want2.abs().max())
    # the loop's waveform for these targets is Griffin-Lim of the frames (n_fft 64, hop 16)
    wave = run_loop(net, (p2.to(device),), 20, yield_inversed_outputs=True)[0]
    assert wave.shape == (5, 16 * (out2.shape[1] - 1)) and bool(torch.isfinite(wave).all())


def test_reference_demo_networks_on_magnitude_frames(device):
    """the two spectral demos of the reference at their real sizes, n_fft 2048 / hop 512 (1025 bins), against the oracle,
    then through the loop into Griffin-Lim:
      demos/freqnet.py:34-63   WaveNet, 3 layers x 2048 channels, groups=8, no residual / skip path, Identity output
      demos/seq2seq.py:35-60   Seq2Seq, model_dim 512, hop 4, 2 + 2 residual bi-LSTMs, edge_sum pooling, repeat up-sampling"""
    import warnings
    warnings.filterwarnings("ignore")
    from oracle.weights import load_recipe
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=2048, hop_length=512, activation="Identity"))
    gen = torch.Generator().manual_seed(13)
    # FreqNet
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, kernel_sizes=(2,), blocks=(3,), dims_dilated=(2048,),
                                                     apply_residuals=False, residuals_dim=None, skips_dim=None, groups=8)).eval()
    sd = load_recipe(net, seed=90, gain=1.5)
    arch = dict(kernels=[2] * 3, dilations=[1, 2, 4], has_skips=False, residuals=False, groups=8, head="linear")
    prompt = torch.rand(2, net.rf + 4, 1025, generator=gen)
    want = O.wavenet_generate_frames(sd, prompt, 6, **arch)
    net.to(device)
    got = run_loop(net, (prompt.to(device),), 6)[0].cpu()
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    wave = run_loop(net, (prompt.to(device),), 6, yield_inversed_outputs=True)[0]
    assert wave.shape == (2, 512 * (got.shape[1] - 1)) and bool(torch.isfinite(wave).all())
    del net
    # Seq2Seq
    io2 = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=2048, hop_length=512, activation="Identity"))
    s2s = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(
        io_spec=io2, model_dim=512, hop=4, enc_downsampling="edge_sum", enc_n_lstm=2, enc_apply_residuals=True,
        dec_upsampling="repeat", dec_n_lstm=2, dec_apply_residuals=True)).eval()
    sd2 = load_recipe(s2s, seed=91, gain=1.5)
    x = torch.rand(3, 4, 1025, generator=gen)
    want = O.s2s_step(O.fold_weight_norm(sd2), x, hop=4, out_abs=False, upsampling="repeat", enc_residuals=True, dec_residuals=True)
    s2s.to(device)
    got = s2s.generate_step((x.to(device),), t=4).cpu()
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    wave = run_loop(s2s, (x.to(device),), 8, yield_inversed_outputs=True)[0]
    assert wave.shape == (3, 512 * 11) and bool(torch.isfinite(wave).all())


def test_seq2seq_loop_ends_in_griffin_lim(device):
    """a MagSpec-target network's loop inverts its frames with GLA inside process_outputs (loops/generate.py:242-245):
    the waveform the loop yields is Griffin-Lim of the frames it generated, phases drawn from torch's device RNG"""
    io = mmk.IOSpec.magspec_io(mmk.IOSpec.MagSpecIOConfig(sr=22050, n_fft=1024, hop_length=256))
    net = mmk.Seq2SeqLSTMNetwork.from_config(mmk.Seq2SeqLSTMNetwork.Config(io_spec=io, model_dim=64, hop=4)).eval()
    from oracle.weights import load_recipe
    load_recipe(net, seed=5, gain=1.5)
    net.to(device)
    prompt = mmk.MagSpec(1024, 256, center=False)((torch.rand(3, 1024 + 256 * 7, generator=torch.Generator().manual_seed(8)) * 2 - 1).to(device))
    frames = run_loop(net, (prompt,), 8)[0]
    assert frames.shape == (3, 16, 513)
    torch.manual_seed(77)
    wave = run_loop(net, (prompt,), 8, yield_inversed_outputs=True)[0]
    assert wave.shape == (3, 256 * 15)
    torch.manual_seed(77)
    init = torch.rand(frames.shape, dtype=torch.complex64, device=device)
    want = O.griffin_lim(frames.cpu(), 1024, 256, 32, 0.99, init.cpu())
    assert float((wave.cpu() - want).norm() / want.norm()) <= 2e-3          # fp32 through 32 iterations (test_gpu_features)


# ---------------------------------------------------------------------------- WaveNet execution modes
def _cond_net(seed=21):
    """persistent-kernel eligible net with one conditioning input: C = S = R = 32, cond 12 -> 16, blocks (3, 2)"""
    from oracle.weights import load_recipe
    io = H.mu_emb(mlp_dim=32)
    ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
    io = mmk.IOSpec(inputs=(io.inputs[0], mmk.InputSpec("signal", mmk.MagSpec(22, 4, center=False), mmk.LinearIO()).bind_to(ext)),
                    targets=io.targets)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=(3, 2), dims_dilated=(32,), dims_1x1=(16,),
                                                     residuals_dim=32, skips_dim=32)).eval()
    sd = load_recipe(net, seed=seed, gain=2.0)
    arch = dict(kernels=[2] * 5, dilations=[1, 2, 4, 1, 2], has_skips=True, residuals=True)
    return net, sd, arch


MODES = {"persistent_xcd": {}, "persistent_agent": {"MMK_WN_XCD_LOCAL": "0"}, "persistent_tiles": {"MMK_WN_SMALL": "0", "MMK_WN_CHAIN": "0"},
         "persistent_step_warmup": {"MMK_WN_PREFILL": "0"}, "launches": {"MMK_WN_PERSISTENT": "0"},
         "two_handoffs_xcd": {"MMK_WN_CHAIN": "0"}, "two_handoffs_agent": {"MMK_WN_CHAIN": "0", "MMK_WN_XCD_LOCAL": "0"},
         "two_handoffs_step_warmup": {"MMK_WN_CHAIN": "0", "MMK_WN_PREFILL": "0"}}


def test_wavenet_timeout_is_redone_on_launch_path(device, monkeypatch):
    """a hand-off timeout reported by the persistent kernel (injected through mmk_wavenet_inject_sync_error) must not return
    blanks: the batch is regenerated on the per-layer launch path, with a warning; both generations are held to the oracle the way
    the modes test does (the two paths associate their sums differently)"""
    for k in ("MMK_WN_XCD_LOCAL", "MMK_WN_PERSISTENT", "MMK_WN_GROUPS", "MMK_WN_SMALL", "MMK_WN_PREFILL", "MMK_WN_CHAIN"):
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    net, sd, arch = _cond_net()
    net = net.to(device)
    gen = torch.Generator().manual_seed(23)
    rf, n, B = net.rf, 40, 3
    prompt = torch.randint(0, 256, (B, rf + 5), generator=gen)
    cond = torch.rand(B, rf + 5 + n, 12, generator=gen)
    want, raw = O.wavenet_generate(sd, prompt, (cond,), n, keep_logits=True, **arch)
    ok = H.margin_ok(raw.numpy())
    first_bad = (~ok).float().cumsum(1) > 0
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx, cond.to(device)), prompt.size(1), n)
    assert net._plan.persistent
    net._plan.inject_sync_error()             # as if a hand-off inside the kernel had timed out (include/mmk.h: fault injection)
    with pytest.warns(UserWarning, match="launch path"):
        net.after_generate((idx,), None)
    same = idx.cpu()[:, prompt.size(1):] == want[:, prompt.size(1):]
    assert bool((same | first_bad).all()) and float(ok.float().mean()) > 0.9
    assert net._plan is None                      # the next generation starts on a fresh (persistent) plan


@pytest.mark.parametrize("mode", list(MODES))
def test_wavenet_modes_agree_with_oracle(device, mode, monkeypatch):
    """the persistent kernels - one hand-off per layer (wavenet_chain.hip) and two (wavenet_persist.hip), XCD-local and
    agent-scope hand-offs, 4x4 MFMA blocks and 16-row tiles, warm-up as a prefill and step by step - and the per-layer launch
    path all reproduce the oracle: conditioned net, batch 5 (ragged clip groups), prompt longer than rf, 70 steps, greedy + sampled"""
    for k in ("MMK_WN_XCD_LOCAL", "MMK_WN_PERSISTENT", "MMK_WN_GROUPS", "MMK_WN_SMALL", "MMK_WN_PREFILL", "MMK_WN_CHAIN"):
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    for k, v in MODES[mode].items():
        monkeypatch.setitem(mmk.native.PLAN_TUNING, k, v)
    net, sd, arch = _cond_net()
    net = net.to(device)
    gen = torch.Generator().manual_seed(17)
    rf, n, B = net.rf, 70, 5
    prompt = torch.randint(0, 256, (B, rf + 9), generator=gen)
    cond = torch.rand(B, rf + 9 + n, 12, generator=gen)
    want, raw = O.wavenet_generate(sd, prompt, (cond,), n, keep_logits=True, **arch)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx, cond.to(device)), prompt.size(1), n)
    net.after_generate((idx,), None)
    assert net._plan.persistent == (mode != "launches")
    assert net._plan.chain == (mode in ("persistent_xcd", "persistent_agent", "persistent_step_warmup"))
    ok = H.margin_ok(raw.numpy())
    first_bad = (~ok).float().cumsum(1) > 0
    same = idx.cpu()[:, prompt.size(1):] == want[:, prompt.size(1):]
    assert bool((same | first_bad).all())
    assert float(ok.float().mean()) > 0.9
    assert torch.allclose(net._plan.last_logits(B).cpu()[ok[:, -1]], raw[:, -1][ok[:, -1]], **LOGIT_TOL)
    # sampled decode with the uniforms generate_block will draw
    temp = torch.tensor([0.6, 1.0, 1.4, 0.8, 1.1])
    torch.manual_seed(5)
    u = torch.rand((B, n), device=device)
    torch.manual_seed(5)
    idx2 = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx2, cond.to(device)), prompt.size(1), n, temperature=temp)
    net.after_generate((idx2,), None)
    got2 = idx2.cpu()
    _, raw2 = O.wavenet_generate(sd, prompt, (cond,), n, keep_logits=True, forced=got2, **arch)
    okp, exact = H.sampled_picks_ok(raw2, temp, u.cpu(), got2[:, prompt.size(1):])
    assert bool(okp.all()) and float(exact.float().mean()) > 0.97


def _small_net(blocks, seed):
    """the geometry family of BASELINE config 2: 64 channels, kernel 2, gated, skips 64, embedding in, MLP head 128 -> 256 (+ temperature)"""
    from oracle.weights import load_recipe
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(mlp_dim=128), blocks=blocks, dims_dilated=(64,), residuals_dim=64,
                                                     skips_dim=64)).eval()
    sd = load_recipe(net, seed=seed, gain=2.0)
    dil = [2 ** i for b in blocks for i in range(b)]
    arch = dict(kernels=[2] * len(dil), dilations=dil, has_skips=True, residuals=True)
    return net, sd, arch


@pytest.mark.parametrize("blocks,B,env", [((10,), 8, {}), ((4, 3), 13, {}), ((3, 3, 3, 2), 3, {}), ((4,), 64, {}),
                                          ((10,), 5, {"MMK_WN_XCD_LOCAL": "0"})])
def test_wavenet_layer_pipeline_agrees_with_oracle(device, monkeypatch, blocks, B, env):
    """wavenet_lpipe.hip - four workgroups per clip that own whole layers: 10 / 7 / 11 / 4 layers (3 + 3 + 2 + 2, 2 + 2 + 2 + 1,
    3 + 3 + 3 + 2, 1 + 1 + 1 + 1 layers per stage), 8 / 13 / 3 / 64 clips (a ragged last group of eight; every CU slot of the grid),
    prompt longer than rf, two generate blocks, greedy against the oracle (classes exact where the oracle's margin allows, last
    logits within tolerance) and sampled against the oracle's CDF intervals; and the same net with the mode switched off"""
    for k in ("MMK_WN_XCD_LOCAL", "MMK_WN_PERSISTENT", "MMK_WN_GROUPS", "MMK_WN_SMALL", "MMK_WN_PREFILL", "MMK_WN_CHAIN", "MMK_WN_LPIPE"):
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    for k, v in env.items():                 # (MMK_WN_XCD_LOCAL=0: every hand-over written through to memory)
        monkeypatch.setitem(mmk.native.PLAN_TUNING, k, v)
    net, sd, arch = _small_net(blocks, seed=60 + len(blocks))
    net = net.to(device)
    gen = torch.Generator().manual_seed(29 + B)
    rf, n = net.rf, 45
    prompt = torch.randint(0, 256, (B, rf + 6), generator=gen)
    want, raw = O.wavenet_generate(sd, prompt, (), n, keep_logits=True, **arch)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx,), prompt.size(1), 20)
    net.generate_block((idx,), prompt.size(1) + 20, n - 20)
    net.after_generate((idx,), None)
    assert net._plan.layer_pipelined
    ok = H.margin_ok(raw.numpy())
    first_bad = (~ok).float().cumsum(1) > 0
    same = idx.cpu()[:, prompt.size(1):] == want[:, prompt.size(1):]
    assert bool((same | first_bad).all())
    assert float(ok.float().mean()) > 0.9
    assert torch.allclose(net._plan.last_logits(B).cpu()[ok[:, -1]], raw[:, -1][ok[:, -1]], **LOGIT_TOL)
    temp = torch.linspace(0.5, 1.5, B)
    torch.manual_seed(5)
    u = torch.rand((B, n), device=device)
    torch.manual_seed(5)
    idx2 = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx2,), prompt.size(1), n, temperature=temp)
    net.after_generate((idx2,), None)
    got2 = idx2.cpu()
    _, raw2 = O.wavenet_generate(sd, prompt, (), n, keep_logits=True, forced=got2, **arch)
    okp, exact = H.sampled_picks_ok(raw2, temp, u.cpu(), got2[:, prompt.size(1):])
    assert bool(okp.all()) and float(exact.float().mean()) > 0.97
    # the per-step protocol (one launch per step) and a forced hand-off timeout (the batch is redone on the launch path)
    net.before_generate((prompt.to(device),), 0)
    step = net.generate_step((prompt[:, -rf:].to(device),), t=prompt.size(1))[0].cpu()
    assert bool(((step[:, 0] == want[:, prompt.size(1)]) | ~ok[:, 0]).all())
    nxt = torch.cat([prompt[:, -rf + 1:], step], 1)
    step2 = net.generate_step((nxt.to(device),), t=prompt.size(1) + 1)[0].cpu()
    assert bool(((step2[:, 0] == want[:, prompt.size(1) + 1]) | first_bad[:, 1]).all())
    net.after_generate((idx,), None)
    idx4 = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx4,), prompt.size(1), n)
    assert net._plan.layer_pipelined
    net._plan.inject_sync_error()
    with pytest.warns(UserWarning, match="launch path"):
        net.after_generate((idx4,), None)
    assert bool(((idx4.cpu()[:, prompt.size(1):] == want[:, prompt.size(1):]) | first_bad).all())
    # switched off: the chain kernel generates the same classes wherever the margin allows
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_WN_LPIPE", "0")
    net2, _, _ = _small_net(blocks, seed=60 + len(blocks))
    net2 = net2.to(device)
    idx3 = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net2.generate_block((idx3,), prompt.size(1), n)
    net2.after_generate((idx3,), None)
    assert not net2._plan.layer_pipelined
    assert bool(((idx3.cpu()[:, prompt.size(1):] == want[:, prompt.size(1):]) | first_bad).all())


def _small_cond_net(blocks, cond_dim, seed):
    """_small_net with one conditioning input - or two (cond_dim a tuple) - (a LinearIO on spectrogram-like frames, a 1x1 convolution of it in every layer)"""
    from oracle.weights import load_recipe
    dims = cond_dim if isinstance(cond_dim, tuple) else (cond_dim,)
    io = H.mu_emb(mlp_dim=128)
    ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
    extra = tuple(mmk.InputSpec("signal", mmk.MagSpec(22, 4, center=False), mmk.LinearIO()).bind_to(ext) for _ in dims)
    io = mmk.IOSpec(inputs=(io.inputs[0], *extra), targets=io.targets)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=blocks, dims_dilated=(64,), dims_1x1=dims, residuals_dim=64,
                                                     skips_dim=64)).eval()
    sd = load_recipe(net, seed=seed, gain=2.0)
    dil = [2 ** i for b in blocks for i in range(b)]
    arch = dict(kernels=[2] * len(dil), dilations=dil, has_skips=True, residuals=True)
    return net, sd, arch


@pytest.mark.parametrize("blocks,B,cond_dim", [((10,), 8, 64), ((4, 3), 13, 16), ((4,), 33, 48), ((5, 2), 9, (32, 16)), ((4,), 20, (16, 48))])
def test_wavenet_layer_pipeline_with_conditioning(device, monkeypatch, blocks, B, cond_dim):
    """the layer pipeline on CONDITIONED 64-channel networks (the FreqNet kind: demos/freqnet.py:47-63): every layer's conv_1x1(c) comes from the
    plan's block GEMM, a thread adds its gate row's term with the delayed taps.  Two generate blocks that straddle nothing and a third call past
    the first conditioning block's positions are not needed to cover the indexing - the block is 1024 positions - so one net generates 1100 steps
    against the device's own history through the oracle (teacher-forced), the others 45 steps free-running; greedy + sampled.  Two conditioning
    inputs (wavenet_v2.py:141-147: their 1x1 products are summed): projections side by side in a row, matrices side by side in the GEMM's K"""
    for k in ("MMK_WN_XCD_LOCAL", "MMK_WN_PERSISTENT", "MMK_WN_GROUPS", "MMK_WN_SMALL", "MMK_WN_PREFILL", "MMK_WN_CHAIN", "MMK_WN_LPIPE"):
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    net, sd, arch = _small_cond_net(blocks, cond_dim, seed=70 + len(blocks))
    net = net.to(device)
    n_in = len(cond_dim) if isinstance(cond_dim, tuple) else 1
    gen = torch.Generator().manual_seed(41 + B)
    rf = net.rf
    n = 1100 if blocks == (4,) else 45                 # (rf 16: 1100 steps cross the 1024-position conditioning block cheaply)
    prompt = torch.randint(0, 256, (B, rf + 6), generator=gen)
    conds = tuple(torch.rand(B, rf + 6 + n, 12, generator=gen) for _ in range(n_in))
    conds_d = tuple(c.to(device) for c in conds)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx, *conds_d), prompt.size(1), 20)
    net.generate_block((idx, *conds_d), prompt.size(1) + 20, n - 20)
    assert net._plan.layer_pipelined
    net.after_generate((idx,), None)
    got = idx.cpu()
    want, raw = O.wavenet_generate(sd, prompt, conds, n, keep_logits=True, forced=got, **arch)
    ok = H.margin_ok(raw.numpy())
    assert float(ok.float().mean()) > 0.9
    assert torch.equal(want[:, prompt.size(1):][ok], got[:, prompt.size(1):][ok])
    temp = torch.linspace(0.6, 1.4, B)
    torch.manual_seed(6)
    u = torch.rand((B, 45), device=device)
    torch.manual_seed(6)
    idx2 = torch.cat([prompt, torch.zeros(B, 45, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx2, *(c[:, :rf + 6 + 45] for c in conds_d)), prompt.size(1), 45, temperature=temp)
    net.after_generate((idx2,), None)
    got2 = idx2.cpu()
    _, raw2 = O.wavenet_generate(sd, prompt, tuple(c[:, :rf + 6 + 45] for c in conds), 45, keep_logits=True, forced=got2, **arch)
    okp, exact = H.sampled_picks_ok(raw2, temp, u.cpu(), got2[:, prompt.size(1):])
    assert bool(okp.all()) and float(exact.float().mean()) > 0.97


def _wide_net(C, cond_dim, seed):
    from oracle.weights import load_recipe
    io = H.mu_emb(mlp_dim=32)
    ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
    io = mmk.IOSpec(inputs=(io.inputs[0], mmk.InputSpec("signal", mmk.MagSpec(22, 4, center=False), mmk.LinearIO()).bind_to(ext)),
                    targets=io.targets)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=(3,), dims_dilated=(C,), dims_1x1=(cond_dim,),
                                                     residuals_dim=C, skips_dim=C)).eval()
    sd = load_recipe(net, seed=seed, gain=2.0)
    arch = dict(kernels=[2] * 3, dilations=[1, 2, 4], has_skips=True, residuals=True)
    return net, sd, arch


@pytest.mark.parametrize("C,B,env", [(96, 3, {}), (128, 72, {}), (128, 37, {"MMK_WN_XCD_LOCAL": "0"}), (256, 40, {}),
                                     (96, 3, {"MMK_WN_CHAIN": "0"}), (128, 37, {"MMK_WN_XCD_LOCAL": "0", "MMK_WN_CHAIN": "0"}), (160, 30, {"MMK_WN_CHAIN": "1"}), (224, 13, {"MMK_WN_XCD_LOCAL": "0", "MMK_WN_CHAIN": "1"}), (64, 32, {}),
                                     (256, 27, {"MMK_WN_CHAIN": "1"})])
def test_wavenet_persistent_kernel_shapes(device, monkeypatch, C, B, env):
    """other instantiations of the persistent kernels against the oracle: 96 channels (2 K-chunks per matrix wave,
    3 matrix waves), 128 channels with 9 clips per group (16-row MFMA tiles, two poll rounds per hand-off),
    agent-scope groups with a ragged last group, 256 channels with 5 clips per group; the one-hand-off kernel at 96 / 160 /
    224 / 64 / 256 channels (3 / 5 / 7 / 2 / 8 K-chunks per matrix wave, odd counts split 2 + 1 ... between the two
    weight pieces), full and ragged groups of up to 4 clips, XCD-local and agent-scope"""
    for k in ("MMK_WN_XCD_LOCAL", "MMK_WN_PERSISTENT", "MMK_WN_GROUPS", "MMK_WN_SMALL", "MMK_WN_PREFILL", "MMK_WN_CHAIN"):
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    for k, v in env.items():
        monkeypatch.setitem(mmk.native.PLAN_TUNING, k, v)
    net, sd, arch = _wide_net(C, 16, seed=40 + C)
    net = net.to(device)
    gen = torch.Generator().manual_seed(C + B)
    rf, n = net.rf, 12
    prompt = torch.randint(0, 256, (B, rf + 3), generator=gen)
    cond = torch.rand(B, rf + 3 + n, 12, generator=gen)
    want, raw = O.wavenet_generate(sd, prompt, (cond,), n, keep_logits=True, **arch)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx, cond.to(device)), prompt.size(1), n)
    net.after_generate((idx,), None)
    assert net._plan.persistent
    ok = H.margin_ok(raw.numpy())
    first_bad = (~ok).float().cumsum(1) > 0
    same = idx.cpu()[:, prompt.size(1):] == want[:, prompt.size(1):]
    assert bool((same | first_bad).all())
    assert float(ok.float().mean()) > 0.9
    assert torch.allclose(net._plan.last_logits(B).cpu()[ok[:, -1]], raw[:, -1][ok[:, -1]], **LOGIT_TOL)


def test_wavenet_persistent_long_block_crosses_cond_blocks(device, monkeypatch):
    """more steps than one conditioning block (1024): two persistent launches chained through the product rings"""
    for k in ("MMK_WN_XCD_LOCAL", "MMK_WN_PERSISTENT", "MMK_WN_GROUPS", "MMK_WN_SMALL", "MMK_WN_PREFILL"):
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    net, sd, arch = _cond_net(seed=22)
    net = net.to(device)
    gen = torch.Generator().manual_seed(3)
    rf, n, B = net.rf, 1030, 2
    prompt = torch.randint(0, 256, (B, rf), generator=gen)
    cond = torch.rand(B, rf + n, 12, generator=gen)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx, cond.to(device)), rf, n)
    net.after_generate((idx,), None)
    assert net._plan.persistent
    # teacher-forced oracle check of the LAST few steps: feed the GPU's own history to the naive window forward
    hist = idx.cpu()
    for t in range(rf + n - 4, rf + n):
        raw = O.wavenet_window_forward(sd, (hist[:, t - rf:t], cond[:, t - rf:t]), n_cond=1, **arch)
        pick = O.categorical(O.mlp_logits(raw))
        gap_ok = H.margin_ok(raw.numpy())[:, 0]
        assert bool(((pick[:, 0] == hist[:, t]) | ~gap_ok).all())


# ---------------------------------------------------------------------------- plans and weights
def test_before_generate_repacks_only_changed_weights(device):
    """the plan holds a re-packed copy of the weights: a second before_generate of an unchanged network launches no packing
    kernel (counter exported by the library), while an in-place update, a load_state_dict or new storage repacks - and the
    new weights are the ones used"""
    from mimikit_amd import native
    g = H.golden("wavenet.npz")
    net, sd, arch = H.wavenet_a()
    net = net.to(device)
    prompt = H.T(g["a_prompt"]).to(device)
    out1 = run_loop(net, (prompt,), 24)[0].cpu()
    assert torch.equal(out1, H.T(g["a_out"]))
    n0 = native.pack_launch_count()
    out2 = run_loop(net, (prompt,), 24)[0].cpu()
    assert native.pack_launch_count() == n0 and torch.equal(out2, out1)
    net.generate_step((prompt[:, -net.rf:],), t=prompt.size(1))       # the window-rebuild path does not repack either
    assert native.pack_launch_count() == n0
    with torch.no_grad():
        net.layers[1].conv_skip.weight.mul_(-1.0)                       # what an optimiser step does
    sd2 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    out3 = run_loop(net, (prompt,), 24)[0].cpu()
    assert native.pack_launch_count() > n0
    assert torch.equal(out3, O.wavenet_generate(sd2, prompt.cpu(), (), 24, **arch))
    net.load_state_dict({k: v.to(device) for k, v in sd.items()}, strict=False)
    assert torch.equal(run_loop(net, (prompt,), 24)[0].cpu(), out1)
    # a write through .data (an EMA copy, checkpoint averaging, weight surgery) bumps no version counter and moves no storage:
    # the content fingerprint of the token has to catch it
    n2 = native.pack_launch_count()
    net.layers[1].conv_skip.weight.data.copy_(sd2["layers.1.conv_skip.weight"].to(device))
    out4 = run_loop(net, (prompt,), 24)[0].cpu()
    assert native.pack_launch_count() > n2 and torch.equal(out4, out3)
    net.layers[1].conv_skip.weight.data.copy_(sd["layers.1.conv_skip.weight"].to(device))
    assert torch.equal(run_loop(net, (prompt,), 24)[0].cpu(), out1)
    # SampleRNN: same contract, and the hidden state is reset without a repack
    snet, _, _ = H.srnn("gru")
    snet.to(device)               # (a network that lives on the host is moved, hence re-packed, by every loop run)
    gs = H.golden("srnn.npz")
    a = run_loop(snet, (H.T(gs["gru_prompt"]),), 40, parameters=None)[0].cpu()
    n1 = native.pack_launch_count()
    b = run_loop(snet, (H.T(gs["gru_prompt"]),), 40, parameters=None)[0].cpu()
    assert native.pack_launch_count() == n1 and torch.equal(a, b) and torch.equal(a, H.T(gs["gru_out"]))


def test_seq2seq_eval_forward_follows_weight_updates(device):
    """eval forward / generate_step after a training step or a load_state_dict must see the new weights (per-epoch
    validation in the reference's trainer), and must not repack when nothing changed"""
    from mimikit_amd import native
    net, sd = H.s2s_tiny()
    net.to(device)
    x = torch.rand(3, 4, 65, generator=torch.Generator().manual_seed(2))
    y0 = net((x.to(device),)).cpu()
    assert float((y0 - O.s2s_step(sd, x, hop=4)).abs().max()) <= 1e-4 * float(y0.abs().max())
    n0 = native.pack_launch_count()
    net.generate_step((x.to(device),), t=4)
    assert native.pack_launch_count() == n0
    with torch.no_grad():
        net.enc.fc_out.weight.mul_(0.5)
    sd1 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    y1 = net((x.to(device),)).cpu()
    want = O.s2s_step(sd1, x, hop=4)
    assert float((y1 - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert float((y1 - y0).abs().max()) > 1e-3 * float(y0.abs().max())


def test_sample_rnn_protocol_details(device):
    """generate_step inside the prompt returns () (the reference only advances the tiers there, sample_rnn_v2.py:254-255);
    the sampler keeps torch.multinomial's (B, 1) shape for 2-D logits (modules/targets.py:45-52)"""
    net, _, _ = H.srnn("gru")
    net = net.to(device)
    prompt = torch.randint(0, 256, (2, 48), generator=torch.Generator().manual_seed(1)).to(device)
    net.before_generate((prompt,), 0)
    assert net.generate_step((prompt[:, :16],), t=20) == ()
    out = net.generate_step((prompt[:, -16:],), t=48)
    assert out[0].shape == (2, 1)
    net.after_generate(out, 0)
    s = mmk.CategoricalSampler().eval()
    logits = torch.randn(5, 256, device=device)
    assert s(logits).shape == (5,) and s(logits, temperature=0.7).shape == (5, 1)
    assert s(logits.reshape(5, 1, 256), temperature=0.7).shape == (5, 1)


def _cfg4_family_net(blocks, seed, cond):
    """the geometry family of BASELINE config 4: 256 channels, kernel 2, gated, skips 256, embedding in, MLP head 128 -> 256 (+ temperature),
    optionally one conditioning input (12 -> 16 channels)"""
    from oracle.weights import load_recipe
    io = H.mu_emb(mlp_dim=128)
    kw = {}
    if cond:
        ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
        io = mmk.IOSpec(inputs=(io.inputs[0], mmk.InputSpec("signal", mmk.MagSpec(22, 4, center=False), mmk.LinearIO()).bind_to(ext)),
                        targets=io.targets)
        kw["dims_1x1"] = (16,)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=blocks, dims_dilated=(256,), residuals_dim=256, skips_dim=256, **kw)).eval()
    sd = load_recipe(net, seed=seed, gain=2.0)
    dil = [2 ** i for b in blocks for i in range(b)]
    arch = dict(kernels=[2] * len(dil), dilations=dil, has_skips=True, residuals=True)
    return net, sd, arch


SPIPE_ENV = ("MMK_WN_XCD_LOCAL", "MMK_WN_PERSISTENT", "MMK_WN_GROUPS", "MMK_WN_SMALL", "MMK_WN_PREFILL", "MMK_WN_CHAIN", "MMK_WN_LPIPE",
             "MMK_WN_SPIPE", "MMK_WN_BPIPE")


@pytest.mark.parametrize("blocks,B,cond,n", [((3,), 1, False, 40), ((4, 2), 5, True, 60), ((1,), 3, True, 24), ((2, 1, 1), 32, True, 30),
                                             ((10, 10, 10, 1), 3, False, 6), ((5, 3), 17, True, 1100), ((3, 1), 8, True, 40), ((2,), 2, False, 30),
                                             ((4, 4, 3), 9, False, 50), ((1, 1), 4, True, 33), ((6,), 70, True, 12)])
def test_wavenet_stage_pipeline_agrees_with_oracle(device, monkeypatch, blocks, B, cond, n):
    """the stage pipeline (wavenet_spipe.hip: one layer per stage of 8 CUs, clips streamed through one at a time) against the oracle,
    teacher-forced on the device's own history: 1 - 31 layers (1 - 8 XCDs in use, layers with d = 1 first, in the middle and last),
    1 - 32 clips, with and without conditioning, more steps than one conditioning block (two launches chained through the rings);
    greedy, then sampled with the uniforms generate_block draws; the same generation twice is bit-identical"""
    _stage_pipeline_against_oracle(device, monkeypatch, blocks, B, cond, n, batched=False)


@pytest.mark.parametrize("blocks,B,cond,n", [((3,), 1, False, 40), ((4, 2), 5, True, 60), ((1,), 3, True, 24), ((2, 1, 1), 40, True, 30),
                                             ((10, 10, 10, 1), 20, False, 6), ((5, 3), 17, True, 1100), ((6,), 70, True, 12), ((2,), 150, False, 20),
                                             ((1, 1), 16, True, 33), ((4, 4, 3), 33, False, 50), ((2,), 530, False, 6)])
def test_wavenet_batch_pipeline_agrees_with_oracle(device, monkeypatch, blocks, B, cond, n):
    """the stage pipeline's large-batch form (wavenet_bpipe.hip: the clips travel in groups of 16, a visit is a set of 16x16x4 matrix products)
    against the oracle, as above: 1 - 31 layers, 1 - 10 groups with ragged last groups (1, 5, 3, 8, 4, 1, 6 clips), exactly one group, with and
    without conditioning, two launches chained through the rings, more clips than one launch takes (530: two plans of 265, `native.WaveNetPlanSet`);
    greedy, sampled, twice bit-identical"""
    _stage_pipeline_against_oracle(device, monkeypatch, blocks, B, cond, n, batched=True)


def _stage_pipeline_against_oracle(device, monkeypatch, blocks, B, cond, n, batched):
    for k in SPIPE_ENV:
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_WN_SPIPE", "1")      # (by name: a ring of few stages is not the default for many clips)
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_WN_BPIPE", "1" if batched else "0")
    net, sd, arch = _cfg4_family_net(blocks, 300 + len(blocks) + B, cond)
    net = net.to(device)
    gen = torch.Generator().manual_seed(B + n)
    rf = net.rf
    P = rf + 3
    prompt = torch.randint(0, 256, (B, P), generator=gen)
    conds = (torch.rand(B, P + n, 12, generator=gen),) if cond else ()
    conds_d = tuple(c.to(device) for c in conds)

    def run(**kw):
        idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
        net.generate_block((idx, *conds_d), P, n, **kw)
        net.after_generate((idx,), None)
        return idx.cpu()

    got = run()
    assert net._plan.stage_pipelined and net._plan.batch_pipelined == batched
    last = net._plan.last_logits(B).cpu()
    assert torch.equal(got, run())
    steps = list(range(n)) if n <= 60 else list(range(0, 4)) + list(range(1020, 1030)) + list(range(n - 4, n))
    n_ok = 0
    for s_ in steps:
        t = P + s_
        raw = O.wavenet_window_forward(sd, (got[:, t - rf:t], *[c[:, t - rf:t] for c in conds]), n_cond=len(conds), **arch)
        pick = O.categorical(O.mlp_logits(raw))[:, 0]
        gap_ok = H.margin_ok(raw.numpy())[:, 0]
        assert bool(((pick == got[:, t]) | ~gap_ok).all()), f"step {s_}"
        n_ok += int(gap_ok.sum())
        if s_ == n - 1:
            assert torch.allclose(last[gap_ok], raw[:, 0][gap_ok], **LOGIT_TOL)
    assert n_ok >= 0.98 * len(steps) * B
    if n > 60 or sum(blocks) > 12:
        return
    temp = torch.linspace(0.6, 1.4, B)
    torch.manual_seed(5)
    u = torch.rand((B, n), device=device)
    torch.manual_seed(5)
    got2 = run(temperature=temp)
    _, raw2 = O.wavenet_generate(sd, prompt, conds, n, keep_logits=True, forced=got2, **arch)
    okp, exact = H.sampled_picks_ok(raw2, temp, u.cpu(), got2[:, P:])
    assert bool(okp.all()) and float(exact.float().mean()) > 0.97
    # a hand-off time-out reported by the kernel (injected): the batch is redone on the launch path from the same rings
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx, *conds_d), P, n)
    assert net._plan.batch_pipelined == batched
    net._plan.inject_sync_error()
    with pytest.warns(UserWarning, match="launch path"):
        net.after_generate((idx,), None)
    redone = idx.cpu()
    _, raw3 = O.wavenet_generate(sd, prompt, conds, n, keep_logits=True, forced=redone, **arch)
    ok3 = H.margin_ok(raw3.numpy())
    assert bool(((O.categorical(O.mlp_logits(raw3)) == redone[:, P:]) | ~ok3).all())


@pytest.mark.parametrize("tag", list(H.WAVENET_OPTIONS))
def test_wavenet_options_match_reference_golden(device, tag):
    """the remaining WaveNet options on the HIP path against the reference's loop: deeper MLP heads (one hidden block repeated),
    act_g=None, reverse_layer_order (the residual-free layer runs first; without skips the head reads the residual sum),
    layerwise_inputs, tie_io_weights; classes bit-exact, raw head outputs within the logit tolerance; then a batch of 9
    against the oracle, teacher-forced on the device's history"""
    g = H.golden("wavenet_options.npz")
    net, sd, arch = H.wavenet_option(tag)
    n_cond = arch.pop("n_cond")
    prompt = H.T(g[f"{tag}_prompt"])
    prompts = (prompt,) if not n_cond else (prompt, H.T(g[f"{tag}_cond"]))
    out = run_loop(net, prompts, 16)
    assert torch.equal(out[0].cpu(), H.T(g[f"{tag}_out"]))
    assert torch.allclose(net._plan.last_logits(3).cpu(), H.T(g[f"{tag}_raw"])[:, -1], **LOGIT_TOL)
    net = net.to(device)
    gen = torch.Generator().manual_seed(len(tag))
    rf, B, n = net.rf, 9, 24
    p2 = torch.randint(0, 256, (B, rf + 2), generator=gen)
    cond = (torch.rand(B, rf + 2 + n, 12, generator=gen),) if n_cond else ()
    idx = torch.cat([p2, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx, *[c.to(device) for c in cond]), p2.size(1), n)
    net.after_generate((idx,), None)
    got = idx.cpu()
    want, raw = O.wavenet_generate(sd, p2, cond, n, keep_logits=True, forced=got, **arch)
    ok = H.margin_ok(raw.numpy())
    assert bool(((got[:, p2.size(1):] == want[:, p2.size(1):]) | ~ok).all()) and float(ok.float().mean()) > 0.9


@pytest.mark.parametrize("tag", list(H.WAVENET_ACTS))
def test_wavenet_activations_match_reference_golden(device, tag):
    """Config.act_f / act_g other than Tanh / Sigmoid on the HIP path (the launch path: the persistent kernels have the default gate built in) against the
    reference's loop: classes bit-exact, raw head outputs within the logit tolerance; then a batch of 9 against the oracle, teacher-forced on the device's
    history"""
    g = H.golden("wavenet_acts.npz")
    net, sd, arch = H.wavenet_act(tag)
    n_cond = arch.pop("n_cond")
    prompt = H.T(g[f"{tag}_prompt"])
    prompts = (prompt,) if not n_cond else (prompt, H.T(g[f"{tag}_cond"]))
    out = run_loop(net, prompts, 16)
    assert torch.equal(out[0].cpu(), H.T(g[f"{tag}_out"]))
    assert not net._plan.persistent
    assert torch.allclose(net._plan.last_logits(3).cpu(), H.T(g[f"{tag}_raw"])[:, -1], **LOGIT_TOL)
    net = net.to(device)
    gen = torch.Generator().manual_seed(len(tag))
    rf, B, n = net.rf, 9, 24
    p2 = torch.randint(0, 256, (B, rf + 2), generator=gen)
    cond = (torch.rand(B, rf + 2 + n, 12, generator=gen),) if n_cond else ()
    idx = torch.cat([p2, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx, *[c.to(device) for c in cond]), p2.size(1), n)
    net.after_generate((idx,), None)
    got = idx.cpu()
    want, raw = O.wavenet_generate(sd, p2, cond, n, keep_logits=True, forced=got, **arch)
    ok = H.margin_ok(raw.numpy())
    assert bool(((got[:, p2.size(1):] == want[:, p2.size(1):]) | ~ok).all()) and float(ok.float().mean()) > 0.9


@pytest.mark.parametrize("tag", list(H.MLP_HEADS))
def test_mlp_head_variants_match_reference_golden(device, tag):
    """MLP heads with another activation than Mish and / or dropout modules (eval mode: identities) on the HIP path - the kernels in turns: the fused and
    resident kernels have Mish built in - against the reference's loop: classes bit-exact, the last step's raw head outputs within the logit tolerance"""
    g = H.golden("mlp_heads.npz")
    net, sd, kind, arch = H.mlp_head_case(tag)
    prompt = H.T(g[f"{tag}_prompt"])
    n = 16 if kind == "wavenet" else 24
    out = run_loop(net, (prompt,), n)
    assert torch.equal(out[0].cpu(), H.T(g[f"{tag}_out"]))
    raw = H.T(g[f"{tag}_raw"]).reshape(3, -1, 257)
    assert torch.allclose(net._plan.last_logits(3).cpu(), raw[:, -1], **LOGIT_TOL)
    if kind == "wavenet":
        assert not net._plan.persistent
    else:
        assert net._plan.resident_blocks() == 0


def test_mlp_head_with_dropout_is_refused_in_training_mode(device):
    """a Dropout module is an identity only in eval mode: a network left in training mode is refused by name, not run without its dropout"""
    net, sd, kind, arch = H.mlp_head_case("wn_relu_dp")
    net = net.to(device).train()
    idx = torch.zeros(2, net.rf + 4, dtype=torch.int64, device=device)
    with pytest.raises(NotImplementedError, match="dropout in training mode"):
        net._describe(2)


def test_wavenet_refuses_activations_it_does_not_evaluate(device):
    """PhaseA / GLU / Softmax and the like (modules/activations.py:25-39) are named in the refusal, not run as something else"""
    io = H.mu_emb(mlp_dim=32)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=(2,), dims_dilated=(16,), residuals_dim=16, skips_dim=16, act_g="Softmax")).eval().to(device)
    idx = torch.zeros(2, net.rf + 4, dtype=torch.int64, device=device)
    with pytest.raises(NotImplementedError, match="act_f / act_g"):
        net.generate_block((idx,), net.rf, 4)


@pytest.mark.parametrize("tag", list(H.SRNN_OPTIONS))
def test_sample_rnn_options_match_reference_golden(device, tag):
    """stacked recurrent layers per tier, deeper MLP heads, inputs_mode mean / static_mix (one input), h0_init='ones' on the HIP
    path against the reference's loop (classes bit-exact where the fixture's logit gap allows), then 11 clips against the
    oracle teacher-forced on the device's history"""
    g = H.golden("srnn_options.npz")
    net, sd, arch = H.srnn_option(tag)
    raw = g[f"{tag}_raw"].reshape(3, 40, 257)
    ok = H.margin_ok(raw)
    out = run_loop(net, (H.T(g[f"{tag}_prompt"]),), 40, parameters=None)[0].cpu()
    P = g[f"{tag}_prompt"].shape[1]
    bad = (~ok).float().cumsum(1) > 0
    assert bool(((out[:, P:] == H.T(g[f"{tag}_out"])[:, P:]) | bad).all()) and float(ok.float().mean()) > 0.9
    net = net.to(device)
    gen = torch.Generator().manual_seed(len(tag))
    fs0 = arch["frame_sizes"][0]
    B, n = 11, 50
    prompt = torch.randint(0, 256, (B, 3 * fs0 + 3), generator=gen)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx[:, :prompt.size(1)],), None)
    net.generate_block((idx,), prompt.size(1), n)
    net.after_generate((idx,), None)
    got = idx.cpu()
    want, raw2 = O.SampleRNNOracle(sd, **arch).generate(prompt, n, keep_logits=True, forced=got)
    ok2 = H.margin_ok(raw2.numpy())
    assert bool(((got[:, prompt.size(1):] == want[:, prompt.size(1):]) | ~ok2).all()) and float(ok2.float().mean()) > 0.9


# ---------------------------------------------------------------------------- several inputs / targets
@pytest.mark.parametrize("tag", list(H.MULTI_IO))
def test_multi_input_multi_target_matches_reference_golden(tag, device):
    """networks of several inputs and targets on the HIP path (the loop writes output k into input k): (a) the fused block against
    the reference's loop - every stream, every head's raw outputs of the last step; (b) the per-step protocol gives the same
    streams; (c) 9 clips, teacher-forced on the device's own history, against the oracle - greedy, then sampled (every pick inside
    the oracle's CDF interval of the same uniform, per target)"""
    g = H.golden("multi_io.npz")
    net, sd, arch, classes = H.multi_io(tag)
    M = len(classes)
    n_tgt = len(H.MULTI_IO[tag][2])
    prompts = tuple(H.T(g[f"{tag}_prompt{m}"]) for m in range(M))
    raws = [H.T(g[f"{tag}_raw{k}"]) for k in range(n_tgt)]
    raws = [r.reshape(3, 24, r.shape[-1]) for r in raws]
    ok = torch.stack([H.margin_ok(r) for r in raws]).all(0)
    assert float(ok.float().mean()) > 0.9
    bad = (~ok).float().cumsum(1) > 0                      # a near-tie in any head frees every later step of the clip
    P, n = prompts[0].size(1), 24
    out = run_loop(net, prompts, n)
    assert len(out) == M
    for m in range(M):
        got, want = out[m].cpu(), H.T(g[f"{tag}_out{m}"])
        assert torch.equal(got[:, :P], want[:, :P]) and bool(((got[:, P:] == want[:, P:]) | bad).all()), (tag, m)
    if bool(ok.all()):
        for k in range(n_tgt):
            assert torch.allclose(net._plan.last_logits(3, k).cpu(), raws[k][:, -1], **LOGIT_TOL), (tag, k)
    # (b) step by step through the ARM protocol
    net = net.to(device)
    rf = net.rf
    tens = [torch.cat([p, torch.zeros(3, n, dtype=torch.int64)], 1).to(device) for p in prompts]
    net.before_generate(tuple(x[:, :P] for x in tens), None)
    for t in range(P, P + n):
        outs = net.generate_step(tuple(x[:, t - rf:t] for x in tens), t=t)
        assert type(outs) is tuple and len(outs) == n_tgt and all(o.shape == (3, 1) for o in outs)
        for x, o in zip(tens, outs):
            x[:, t:t + 1] = o
    net.after_generate(tuple(tens), None)
    for m in range(M):
        assert torch.equal(tens[m].cpu(), out[m].cpu()), (tag, m)
    # (c) more clips than the fixture holds, against the oracle
    gen = torch.Generator().manual_seed(len(tag))
    B, n2 = 9, 40
    P2 = 3 * rf + 2 if tag.startswith("srnn") else rf + 7
    pr = tuple(torch.randint(0, q, (B, P2), generator=gen) for q in classes)

    def oracle(forced, temperature=None, uniforms=None):
        if tag.startswith("srnn"):
            return O.SampleRNNOracle(sd, **arch).generate(pr, n2, temperature=temperature, uniforms=uniforms, keep_logits=True, forced=forced)
        ks, ds, kw = arch
        blank = tuple(torch.cat([p, torch.zeros(B, n2, dtype=p.dtype)], 1) for p in pr[n_tgt:])
        return O.wavenet_generate_streams(sd, pr[:n_tgt] + blank, n2, ks, ds, temperature=temperature, uniforms=uniforms, keep_logits=True,
                                          forced=forced, **kw)

    tens = [torch.cat([p, torch.zeros(B, n2, dtype=torch.int64)], 1).to(device) for p in pr]
    net.before_generate(tuple(x[:, :P2] for x in tens), None)
    assert net.generate_block(tuple(tens), P2, n2)
    net.after_generate(tuple(tens), None)
    got = [x.cpu() for x in tens]
    want, raw2 = oracle(got)
    for k in range(n_tgt):
        ok2 = H.margin_ok(raw2[k].numpy())
        assert bool(((got[k][:, P2:] == want[k][:, P2:]) | ~ok2).all()) and float(ok2.float().mean()) > 0.9, (tag, k)
    for m in range(n_tgt, M):
        assert int(got[m][:, P2:].abs().max()) == 0          # (a stream no target feeds keeps the loop's blanks)
    # sampled decode: the uniforms the network drew are not visible from outside, so the plan is driven directly
    temp = torch.full((B,), 0.8)
    uni = torch.rand(n_tgt, B, n2, generator=gen)
    tens = [torch.cat([p, torch.zeros(B, n2, dtype=torch.int64)], 1).to(device) for p in pr]
    net.before_generate(tuple(x[:, :P2] for x in tens), None)
    u_dev = (uni if n_tgt > 1 else uni[0]).contiguous().to(device)
    if tag.startswith("srnn"):
        net._plan.generate(tuple(tens), P2, n2, temp.to(device), u_dev)
    else:
        net._plan.generate(tens[0], tens[1:], P2, n2, temp.to(device), u_dev)
    torch.cuda.synchronize()
    got = [x.cpu() for x in tens]
    _, raw3 = oracle(got, temperature=temp, uniforms=uni)
    for k in range(n_tgt):
        ok3, exact = H.sampled_picks_ok(raw3[k], temp, uni[k], got[k][:, P2:])
        assert bool(ok3.all()) and float(exact.float().mean()) > 0.99, (tag, k)


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("q,mlp_dim,cond_dims,blocks,B", [(128, 64, (), (3, 2), 5), (64, 40, (16,), (4,), 9), (256, 128, (16, 32), (3, 1), 6),
                                                          (200, 100, (32, 16), (2, 2, 1), 33)])
def test_wavenet_stage_pipeline_takes_narrower_heads_and_two_conditioning_inputs(device, monkeypatch, q, mlp_dim, cond_dims, blocks, B, batched):
    """the stage pipeline beyond BASELINE's exact head: fewer classes (input and target), fewer hidden units (not a multiple of 16 either) - the
    plan pads the head to the kernel's 128 x 256 with zero rows / columns and a bias of -inf for the classes that do not exist - and TWO
    conditioning inputs, whose projections and 1x1 matrices it lays side by side.  Against the oracle, teacher-forced on the device's own
    history: greedy classes, the last step's raw outputs (the network's own columns), and a sampled generation (every pick inside the
    oracle's CDF interval - a class that does not exist must never be drawn)"""
    from oracle.weights import load_recipe
    for k in SPIPE_ENV:
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_WN_SPIPE", "1")
    monkeypatch.setitem(mmk.native.PLAN_TUNING, "MMK_WN_BPIPE", "1" if batched else "0")      # (both forms of the stage pipeline: one clip / 16 clips per visit)
    io = H.mu_emb(mlp_dim=mlp_dim, q_levels=q)
    ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
    extra = tuple(mmk.InputSpec("signal", mmk.MagSpec(22, 4, center=False), mmk.LinearIO()).bind_to(ext) for _ in cond_dims)
    io = mmk.IOSpec(inputs=(io.inputs[0], *extra), targets=io.targets)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=blocks, dims_dilated=(256,), dims_1x1=tuple(cond_dims), residuals_dim=256,
                                                     skips_dim=256)).eval()
    sd = load_recipe(net, seed=500 + q + B, gain=2.0)
    dil = [2 ** i for b in blocks for i in range(b)]
    arch = dict(kernels=[2] * len(dil), dilations=dil, has_skips=True, residuals=True)
    net = net.to(device)
    gen = torch.Generator().manual_seed(q + B)
    rf, n = net.rf, 40
    P = rf + 3
    prompt = torch.randint(0, q, (B, P), generator=gen)
    conds = tuple(torch.rand(B, P + n, 12, generator=gen) for _ in cond_dims)
    conds_d = tuple(c.to(device) for c in conds)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx, *conds_d), P, n)
    net.after_generate((idx,), None)
    assert net._plan.stage_pipelined and net._plan.batch_pipelined == batched
    got = idx.cpu()
    assert int(got.max()) < q
    last = net._plan.last_logits(B).cpu()
    assert last.shape == (B, q + 1)
    want, raw = O.wavenet_generate(sd, prompt, conds, n, keep_logits=True, forced=got, **arch)
    ok = H.margin_ok(raw.numpy())
    assert bool(((got[:, P:] == want[:, P:]) | ~ok).all()) and float(ok.float().mean()) > 0.95
    assert torch.allclose(last[ok[:, -1]], raw[:, -1][ok[:, -1]], **LOGIT_TOL)
    # sampled: the plan driven directly, so that the uniforms are known
    temp = torch.full((B,), 0.9)
    uni = torch.rand(B, n, generator=gen)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx[:, :P], *(c[:, :P] for c in conds_d)), None)
    net._plan.generate(idx, conds_d, P, n, temp.to(device), uni.to(device))
    torch.cuda.synchronize()
    got = idx.cpu()
    assert int(got.max()) < q
    _, raw = O.wavenet_generate(sd, prompt, conds, n, keep_logits=True, forced=got, temperature=temp, uniforms=uni, **arch)
    ok3, exact = H.sampled_picks_ok(raw, temp, uni, got[:, P:])
    assert bool(ok3.all()) and float(exact.float().mean()) > 0.99


@pytest.mark.parametrize("q,mlp_dim,blocks,B", [(128, 64, (10,), 8), (64, 32, (4, 3), 13), (200, 112, (4,), 40)])
def test_wavenet_layer_pipeline_takes_narrower_heads(device, monkeypatch, q, mlp_dim, blocks, B):
    """the layer pipeline (the cfg-2 kernel) beyond BASELINE's exact head: fewer classes and fewer hidden units, padded by the plan to
    the kernel's 128 x 256 (zero rows / columns, -inf bias for the classes that do not exist).  Greedy against the oracle, the last
    step's raw outputs (the network's own columns), sampled picks inside the oracle's CDF intervals - never a class that does not exist"""
    from oracle.weights import load_recipe
    for k in ("MMK_WN_XCD_LOCAL", "MMK_WN_PERSISTENT", "MMK_WN_GROUPS", "MMK_WN_SMALL", "MMK_WN_PREFILL", "MMK_WN_CHAIN", "MMK_WN_LPIPE"):
        monkeypatch.delitem(mmk.native.PLAN_TUNING, k, raising=False)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(mlp_dim=mlp_dim, q_levels=q), blocks=blocks, dims_dilated=(64,), residuals_dim=64,
                                                     skips_dim=64)).eval()
    sd = load_recipe(net, seed=700 + q, gain=2.0)
    dil = [2 ** i for b in blocks for i in range(b)]
    arch = dict(kernels=[2] * len(dil), dilations=dil, has_skips=True, residuals=True)
    net = net.to(device)
    gen = torch.Generator().manual_seed(q + B)
    rf, n = net.rf, 45
    P = rf + 6
    prompt = torch.randint(0, q, (B, P), generator=gen)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.generate_block((idx,), P, n)
    net.after_generate((idx,), None)
    assert net._plan.layer_pipelined
    got = idx.cpu()
    assert int(got.max()) < q
    last = net._plan.last_logits(B).cpu()
    assert last.shape == (B, q + 1)
    want, raw = O.wavenet_generate(sd, prompt, (), n, keep_logits=True, forced=got, **arch)
    ok = H.margin_ok(raw.numpy())
    assert bool(((got[:, P:] == want[:, P:]) | ~ok).all()) and float(ok.float().mean()) > 0.9
    assert torch.allclose(last[ok[:, -1]], raw[:, -1][ok[:, -1]], **LOGIT_TOL)
    temp = torch.linspace(0.6, 1.4, B)
    uni = torch.rand(B, n, generator=gen)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(device)
    net.before_generate((idx[:, :P],), None)
    net._plan.generate(idx, (), P, n, temp.to(device), uni.to(device))
    torch.cuda.synchronize()
    got = idx.cpu()
    assert int(got.max()) < q
    _, raw = O.wavenet_generate(sd, prompt, (), n, keep_logits=True, forced=got, temperature=temp, uniforms=uni, **arch)
    ok3, exact = H.sampled_picks_ok(raw, temp, uni, got[:, P:])
    assert bool(ok3.all()) and float(exact.float().mean()) > 0.97
