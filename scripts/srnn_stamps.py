"""Diagnostic: phase stamps of the fused SampleRNN bottom kernel on the cfg3 geometry (MMK_SRNN_STAMPS=1)."""
import os
import sys

import torch

os.environ["MMK_DIAG_LIB"] = "1"          # the diagnostic build: python -m mimikit_amd.build --diag
os.environ["MMK_SRNN_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mimikit_amd as mmk
if os.environ.get("SRNN_MULTI") == "1":
    mmk.native.PLAN_TUNING["MMK_SRNN_MULTI_UPDATE"] = "1"

torch.set_grad_enabled(False)


class A:
    workload = "srnn_cfg3"
    clips = 0
    seconds = 0.1


job = bench.SrnnJob(A, torch.device("cuda", 0), 0)
job.to_device()
job.one_pass()
torch.cuda.synchronize()
job.net.before_generate((job.idx[:, :job.prompt_len],), None)
job.net.generate_block((job.idx,), job.prompt_len, job.n_steps)
job.net._plan.last_logits(job.clips)
