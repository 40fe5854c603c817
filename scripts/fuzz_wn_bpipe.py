"""Random geometries of the batched stage pipeline (wavenet_bpipe.hip) against the oracle (GPU box): 1 .. 31 layers in random blocks (dilation 1 in
the middle of the net included), 1 .. 90 clips (ragged last groups), zero / one / two conditioning inputs of random widths, narrower heads, random prompt
lengths and block splits; greedy classes wherever the oracle's margin allows, then a sampled run.  python scripts/fuzz_wn_bpipe.py [cases]
FUZZ_FORM=pair: the same for the ring with two clips per visit (wavenet_spipe_pair.inc): 24 .. 128 clips, even counts."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
import mimikit_amd as mmk  # noqa: E402
from oracle import torch_ref as O  # noqa: E402
from oracle.weights import load_recipe  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
rng = random.Random(int(os.environ.get("FUZZ_SEED", "23")))
PAIR = os.environ.get("FUZZ_FORM", "") == "pair"
mmk.native.PLAN_TUNING["MMK_WN_SPIPE"] = "1"
mmk.native.PLAN_TUNING["MMK_WN_BPIPE"] = "0" if PAIR else "1"
if PAIR:
    mmk.native.PLAN_TUNING["MMK_WN_SPIPE_PAIR"] = "1"
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    L = rng.choice([1, 2, 3, 5, 8, 12, 20, 31])
    blocks, left = [], L
    while left > 0:
        b = rng.randint(1, min(left, 5))
        blocks.append(b)
        left -= b
    B = rng.choice([24, 26, 30, 36, 44, 46, 60, 62, 64, 70, 100, 102, 126, 128]) if PAIR else rng.choice([1, 3, 15, 16, 17, 33, 48, 70, 90, 150, 260, 300])
    cond_dims = rng.choice([(), (), (16,), (48,), (32, 16)])
    q, mlp_dim = rng.choice([(256, 128), (256, 128), (128, 64), (200, 100)])
    io = H.mu_emb(mlp_dim=mlp_dim, q_levels=q)
    ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
    extra = tuple(mmk.InputSpec("signal", mmk.MagSpec(22, 4, center=False), mmk.LinearIO()).bind_to(ext) for _ in cond_dims)
    io = mmk.IOSpec(inputs=(io.inputs[0], *extra), targets=io.targets)
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=tuple(blocks), dims_dilated=(256,), dims_1x1=tuple(cond_dims), residuals_dim=256,
                                                     skips_dim=256)).eval()
    sd = load_recipe(net, seed=900 + case, gain=2.0)
    dil = [2 ** i for b in blocks for i in range(b)]
    arch = dict(kernels=[2] * L, dilations=dil, has_skips=True, residuals=True)
    net = net.to(dev)
    g = torch.Generator().manual_seed(case)
    rf, n = net.rf, rng.randint(2, 24)
    P = rf + rng.randint(0, 9)
    prompt = torch.randint(0, q, (B, P), generator=g)
    conds = tuple(torch.rand(B, P + n, 12, generator=g) for _ in cond_dims)
    conds_d = tuple(c.to(dev) for c in conds)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(dev)
    cut = rng.randint(1, n)
    net.generate_block((idx, *conds_d), P, cut)
    if cut < n:
        net.generate_block((idx, *conds_d), P + cut, n - cut)
    net.after_generate((idx,), None)
    got = idx.cpu()
    want, raw = O.wavenet_generate(sd, prompt, conds, n, keep_logits=True, forced=got, **arch)
    ok = H.margin_ok(raw.numpy())
    good = bool(((got[:, P:] == want[:, P:]) | ~ok).all()) and float(ok.float().mean()) > 0.9 and (net._plan.pair_visits if PAIR else net._plan.batch_pipelined) and int(got.max()) < q
    # sampled: the plan driven directly, so that the uniforms are known
    temp = torch.full((B,), 0.9)
    uni = torch.rand(B, n, generator=g)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(dev)
    net.before_generate((idx[:, :P], *(c[:, :P] for c in conds_d)), None)
    net._plan.generate(idx, conds_d, P, n, temp.to(dev), uni.to(dev))
    torch.cuda.synchronize()
    got = idx.cpu()
    _, raw = O.wavenet_generate(sd, prompt, conds, n, keep_logits=True, forced=got, temperature=temp, uniforms=uni, **arch)
    ok3, exact = H.sampled_picks_ok(raw, temp, uni, got[:, P:])
    good = good and bool(ok3.all()) and int(got.max()) < q
    bad += 0 if good else 1
    print(f"case {case}: blocks={tuple(blocks)} B={B} cond={cond_dims} head={mlp_dim}x{q} P={P} n={n} cut={cut}: {'ok' if good else 'MISMATCH'}", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
