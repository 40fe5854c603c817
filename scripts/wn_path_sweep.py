"""Which WaveNet step path wins where: every path forced in turn over a grid of geometries (channels x layers x clips x conditioning),
us per AR step from HIP events around one generate call.  Output: a markdown table and a JSON file.

    python scripts/wn_path_sweep.py [--steps 256] [--out gpurun_out/r04/wn_path_sweep.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimikit_amd as mmk  # noqa: E402

torch.set_grad_enabled(False)
KNOBS = ("MMK_WN_PERSISTENT", "MMK_WN_CHAIN", "MMK_WN_LPIPE", "MMK_WN_SPIPE")
PATHS = {                   # forced through the plan's `tuning` switches (native.PLAN_TUNING)
    "launch": dict(MMK_WN_PERSISTENT="0"),
    "persist": dict(MMK_WN_SPIPE="0", MMK_WN_CHAIN="0", MMK_WN_LPIPE="0"),
    "chain": dict(MMK_WN_SPIPE="0", MMK_WN_CHAIN="1", MMK_WN_LPIPE="0"),
    "lpipe": dict(MMK_WN_SPIPE="0", MMK_WN_CHAIN="0", MMK_WN_LPIPE="1"),
    "spipe": dict(MMK_WN_SPIPE="1"),
    "default": None,        # whatever the plan picks on its own
}
MODE_NAMES = {0: "launch", 1: "persist", 2: "chain", 4: "lpipe", 5: "spipe", 6: "bpipe"}


def network(C, blocks, cond):
    io = mmk.IOSpec.mulaw_io(mmk.IOSpec.MuLawIOConfig(sr=16000, q_levels=256, input_module_type="embedding"))
    kw = {}
    if cond:
        ext = mmk.Extractor("signal", mmk.FileToSignal(16000))
        ci = mmk.InputSpec("signal", mmk.MagSpec(2 * (C - 1), C // 2, center=False), mmk.LinearIO()).bind_to(ext)
        io = mmk.IOSpec(inputs=(io.inputs[0], ci), targets=io.targets)
        kw["dims_1x1"] = (C,)
    torch.manual_seed(7)
    return mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=io, blocks=blocks, dims_dilated=(C,), residuals_dim=C, skips_dim=C, **kw)).eval()


def run(net, B, steps, cond_dim, device):
    P = -(-net.rf // 16) * 16
    gen = torch.Generator().manual_seed(5)
    idx = torch.cat([torch.randint(0, 256, (B, P), generator=gen), torch.zeros(B, steps, dtype=torch.int64)], 1).to(device)
    cond = (torch.rand(B, P + steps, cond_dim, generator=gen).to(device),) if cond_dim else ()
    best = None
    for rep in range(2):
        net.before_generate((idx[:, :P], *[c[:, :P] for c in cond]), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        net.generate_block((idx, *cond), P, steps)
        e1.record()
        torch.cuda.synchronize()
        net.after_generate((idx,), None)
        us = e0.elapsed_time(e1) * 1e3 / steps
        best = us if best is None else min(best, us)
    return best, MODE_NAMES[int(net._plan._lib.mmk_wavenet_mode(net._plan.handle)) if not hasattr(net._plan, "plans") else 5]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--out", default="gpurun_out/r04/wn_path_sweep.json")
    ap.add_argument("--channels", default="64,128,256")
    ap.add_argument("--blocks", default="10;10,10,10")
    ap.add_argument("--clips", default="8,32,64")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    rows = []
    for C in [int(x) for x in args.channels.split(",")]:
        for blocks in [tuple(int(v) for v in b.split(",")) for b in args.blocks.split(";")]:
            for cond in (0, 1):
                net = network(C, blocks, cond).to(device)
                cond_dim = C if cond else 0
                for B in [int(x) for x in args.clips.split(",")]:
                    row = {"C": C, "L": sum(blocks), "cond": cond, "B": B, "us": {}, "ran": {}}
                    for name, env in PATHS.items():
                        mmk.native.PLAN_TUNING.clear()          # (the switches travel in the plan's config: include/mmk.h `tuning`)
                        if env:
                            mmk.native.PLAN_TUNING.update(env)
                        net._plan = None
                        try:
                            us, ran = run(net, B, args.steps, cond_dim, device)
                        except Exception as e:          # a geometry the forced path refuses
                            row["us"][name], row["ran"][name] = None, f"error: {str(e)[:60]}"
                            continue
                        # a forced path that the plan did not take (unsupported geometry) ran on something else: not this path's number
                        row["us"][name] = round(us, 2) if (name == "default" or ran == name) else None
                        row["ran"][name] = ran
                    print(json.dumps(row), flush=True)
                    rows.append(row)
                net._plan = None
                del net
                torch.cuda.empty_cache()
    mmk.native.PLAN_TUNING.clear()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rows, f, indent=1)
    names = [n for n in PATHS if n != "default"]
    print("| C | L | cond | clips | " + " | ".join(names) + " | default picks |")
    print("|---|---|---|---|" + "---|" * (len(names) + 1))
    for r in rows:
        cells = ["-" if r["us"][n] is None else f"{r['us'][n]:.1f}" for n in names]
        valid = {n: r["us"][n] for n in names if r["us"][n] is not None}
        win = min(valid, key=valid.get) if valid else "-"
        cells = [f"**{c}**" if n == win else c for n, c in zip(names, cells)]
        print(f"| {r['C']} | {r['L']} | {r['cond']} | {r['B']} | " + " | ".join(cells) + f" | {r['ran']['default']} ({r['us']['default']}) |")


if __name__ == "__main__":
    main()
