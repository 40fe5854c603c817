"""cfg 4 in groups of 16 clips (wavenet_bpipe.hip) against the one-clip ring (wavenet_spipe.hip): the same greedy generation on both, classes compared,
us per AR step of each.   python scripts/bpipe_check.py [--clips 256] [--steps 64]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mimikit_amd as mmk  # noqa: E402
from test_gpu_baseline_configs import cfg4_network  # noqa: E402

torch.set_grad_enabled(False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=256)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--prompt", type=int, default=3072)
    ap.add_argument("--only", default="", help="bpipe | spipe: run one of the two (no comparison)")
    ap.add_argument("--pair", default="", help="0 | 1: the one-clip ring with / without two clips per visit (MMK_WN_SPIPE_PAIR); default: by the clip count")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    net, sd, arch = cfg4_network()
    net = net.to(dev)
    B, P, n = args.clips, args.prompt, args.steps
    gen = torch.Generator().manual_seed(11)
    prompt = torch.randint(0, 256, (B, P), generator=gen)
    cond = torch.rand(B, P + n, 513, generator=gen).to(dev)
    out = {}
    for name, tuning in (("bpipe", {"MMK_WN_BPIPE": "1"}), ("spipe", {"MMK_WN_BPIPE": "0"})):
        if args.only and name != args.only:
            continue
        net.exec_tuning = dict(tuning)
        if args.pair and name == "spipe":
            net.exec_tuning["MMK_WN_SPIPE_PAIR"] = args.pair
        net._plan = None
        best = None
        for rep in range(2):
            idx = torch.cat([prompt, torch.zeros(B, n, dtype=torch.int64)], 1).to(dev)
            net.before_generate((idx[:, :P], cond[:, :P]), None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            net.generate_block((idx, cond), P, n)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            net.after_generate((idx,), None)
            best = dt if best is None else min(best, dt)
        mode = "set" if hasattr(net._plan, "plans") else int(net._plan._lib.mmk_wavenet_mode(net._plan.handle))
        out[name] = idx.cpu()
        print(f"{name}: mode {mode}, {best / n * 1e6:.1f} us per step, {B * n / best / 1e3:.0f} k samples/s", flush=True)
    if args.only:
        return
    a, b = out["bpipe"][:, P:], out["spipe"][:, P:]
    same = (a == b)
    print(f"classes equal: {same.float().mean().item() * 100:.2f} % ; first step equal: {same[:, 0].float().mean().item() * 100:.2f} % ; "
          f"clips identical over all steps: {int(same.all(1).sum())} / {B}")
    if not same[:, 0].all():
        bad = (~same[:, 0]).nonzero()[:8, 0].tolist()
        print("first-step mismatches at clips", bad, a[bad, 0].tolist(), b[bad, 0].tolist())


if __name__ == "__main__":
    main()
