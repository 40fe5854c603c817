"""Random geometries of the layer-pipeline WaveNet kernel against the oracle (GPU box): layers 4 .. 11 in random blocks, 1 .. 64 clips,
random prompt lengths and block splits; greedy classes wherever the oracle's margin allows.  python scripts/fuzz_wn_lpipe.py [cases]"""
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
rng = random.Random(11)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    L = rng.randint(4, 11)
    blocks, left = [], L
    while left > 0:
        b = rng.randint(1, min(left, 6))
        blocks.append(b)
        left -= b
    B = rng.choice([1, 2, 5, 8, 9, 16, 31, 64])
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(mlp_dim=128), blocks=tuple(blocks), dims_dilated=(64,), residuals_dim=64,
                                                     skips_dim=64)).eval()
    sd = load_recipe(net, seed=100 + case, gain=2.0)
    dil = [2 ** i for b in blocks for i in range(b)]
    arch = dict(kernels=[2] * L, dilations=dil, has_skips=True, residuals=True)
    net = net.to(dev)
    g = torch.Generator().manual_seed(case)
    rf, n = net.rf, rng.randint(3, 40)
    P = rf + rng.randint(0, 9)
    prompt = torch.randint(0, 256, (B, P), generator=g)
    want, raw = O.wavenet_generate(sd, prompt, (), n, keep_logits=True, **arch)
    idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(dev)
    cut = rng.randint(1, n)
    net.generate_block((idx,), P, cut)
    if cut < n:
        net.generate_block((idx,), P + cut, n - cut)
    net.after_generate((idx,), None)
    ok = H.margin_ok(raw.numpy())
    first_bad = (~ok).float().cumsum(1) > 0
    good = bool(((idx.cpu()[:, P:] == want[:, P:]) | first_bad).all()) and net._plan.layer_pipelined
    bad += 0 if good else 1
    print(f"case {case}: blocks={tuple(blocks)} B={B} P={P} n={n} cut={cut}: {'ok' if good else 'MISMATCH'}", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
