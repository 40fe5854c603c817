"""Small eager (no hipGraph) run of the cfg4 step kernels, as a target for rocprofv3 --pmc."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.set_grad_enabled(False)


class A:
    workload = os.environ.get("WORKLOAD", "wavenet_cfg4")
    clips = int(os.environ.get("CLIPS", "0"))
    seconds = 0.07


job = bench.WaveNetJob(A, torch.device("cuda", 0), 0)
job.to_device()
p = job.prompt_len
net = job.net
for kv in os.environ.get("TUNING", "").split(","):          # plan switches of this run: TUNING="MMK_WN_BPIPE=0,MMK_WN_SPIPE_PAIR=1"
    if "=" in kv:
        net.exec_tuning[kv.split("=")[0]] = kv.split("=")[1]
net._ensure_plan(job.clips, refresh_weights=True)
if net._plan.persistent:
    # one warm-up launch of 12 positions, then ONE persistent launch of 1024 steps with head: the launch that
    # bench.py's roofline times (the counter value of that dispatch = its HBM traffic)
    net._plan.warmup(job.idx, job.cond, p - 13, p - 1)
    net._next_t, net._state_batch = p, job.clips
    net._plan.generate(job.idx, job.cond, p, 1024)
    net._plan.sync_status()
else:
    net._plan.warmup(job.idx, job.cond, p - 13, p - 1)          # 12 eager steps
    net._next_t, net._state_batch = p, job.clips
    net._plan.generate(job.idx, job.cond, p, 12)                # 12 eager steps with head
torch.cuda.synchronize()
print("done")
