"""The multi-GPU path's RCCL calls on the one GPU of the test box: `bench.py --gpus 1 --force-dist` initialises a process group of ONE rank with
backend "nccl" (= RCCL on ROCm) and runs what an N-GPU run runs - the weight broadcast on device tensors, the fenced clock (barrier +
all_reduce(MAX) of a device tensor), destroy_process_group - around a short pass of a real workload.  (The 1 -> 8 curve is the driver's to
measure; this pins that the calls themselves execute against RCCL.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_rccl_pass():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "wavenet_cfg2", "--gpus", "1", "--force-dist", "--steps", "1",
                          "--warmup", "1", "--seconds", "0.05", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["collectives"] == {"backend": "nccl", "world_size": 1, "weight_broadcasts": 1, "clock": "barrier + all_reduce(MAX) on the device"}
