"""The multi-GPU path on CPU: two gloo ranks shard clips, broadcast the weight blob once, and
gather -- the only communication the generate path has."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mimikit_amd as mmk
from mimikit_amd.shard import broadcast_weights, clip_slice, gather_clips
from tests import helpers as H


def test_clip_slice_partitions_exactly():
    for n in (0, 1, 7, 8, 32, 256, 257):
        for world in (1, 2, 3, 4, 8):
            spans = [clip_slice(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert clip_slice(256, 3, 8) == (96, 128)
    with pytest.raises(ValueError):
        clip_slice(8, 2, 2)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # ranks start from DIFFERENT weights
    net = mmk.WaveNet.from_config(mmk.WaveNet.Config(io_spec=H.mu_emb(mlp_dim=16), blocks=(3,), dims_dilated=(8,),
                                                     residuals_dim=8, skips_dim=8))
    before = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    nbytes = broadcast_weights(net, src=0)
    after = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    start, stop = clip_slice(5, rank, world)
    local = torch.arange(start, stop).reshape(-1, 1).repeat(1, 3)
    full = gather_clips(local, dst=0)
    out[rank] = dict(changed=bool((before != after).any()), checksum=float(after.double().sum()), nbytes=nbytes,
                     gathered=None if full is None else full.tolist())
    dist.destroy_process_group()


def test_two_rank_weight_broadcast_and_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
        r0, r1 = out[0], out[1]
    assert r0["checksum"] == r1["checksum"]            # identical weights everywhere after ONE broadcast
    assert not r0["changed"] and r1["changed"]
    assert r0["nbytes"] == r1["nbytes"] > 0
    assert r0["gathered"] == [[i] * 3 for i in range(5)] and r1["gathered"] is None


def _bench_worker(rank, world, port, out):
    """the bench contract under gloo with a stub job: per-rank seeds, weak-scaling units, fenced timing with MAX over ranks"""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mimikit_amd.shard import timed_passes
    gen = torch.Generator().manual_seed(1234 + rank)           # bench.py: every rank draws ITS clips' synthetic inputs
    clips = torch.rand(4, 8, generator=gen)
    calls = []

    def one_pass():
        calls.append(time.perf_counter())
        time.sleep(0.05 * (rank + 1))                         # rank 1 is the slow one

    elapsed = timed_passes(one_pass, steps=3, warmup=2, sync=lambda: None)
    start, stop = clip_slice(8, rank, world)
    out[rank] = dict(elapsed=elapsed, n_calls=len(calls), first=float(clips[0, 0]), span=(start, stop))
    dist.destroy_process_group()


def test_bench_timing_contract_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_bench_worker, args=(2, port, out), nprocs=2, join=True)
        r0, r1 = out[0], out[1]
    assert r0["n_calls"] == r1["n_calls"] == 5                 # 2 warm-up + exactly 3 timed passes
    assert r0["elapsed"] == r1["elapsed"]                      # MAX over ranks, agreed by all
    assert 0.29 <= r0["elapsed"] < 0.6                         # the slow rank's 3 x 0.1 s, not the fast rank's 0.15 s
    assert r0["first"] != r1["first"]                          # different synthetic clips per rank
    assert r0["span"] == (0, 4) and r1["span"] == (4, 8)       # weak scaling: per-rank units fixed, global = world x


def _run_bench(*argv):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res.returncode, (json.loads(lines[-1]) if lines else None), res.stderr


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) starts two ranks itself: rank 0's line says n_gpus 2, the
    weights were broadcast once and are equal on both ranks, the time is the SLOWER rank's (rank 1 sleeps 40 ms per pass, rank 0
    20 ms), the value counts both ranks' units"""
    rc, line, err = _run_bench("--workload", "stub", "--gpus", "2", "--steps", "3", "--warmup", "1", "--clips", "4")
    assert rc == 0, err
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    cfg = line["config"]
    assert cfg["broadcasts"] == 1 and cfg["weights_checksum_spread"] == 0.0 and cfg["global_clips"] == 8
    assert 39.0 <= line["ms_per_step"] <= 80.0              # max over ranks: rank 1's 40 ms, not rank 0's 20
    assert abs(line["value"] - 8 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3


def test_launcher_counts_gpus_without_hip(monkeypatch):
    """the launcher's device count comes from the visible-devices list or the KFD topology in sysfs - never from a HIP call in the parent"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2,5")
    assert bench.visible_gpus() == 3
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    assert bench.visible_gpus() >= torch.cuda.device_count()       # (0 on a box without the driver; a container may see fewer than the node has)
    import inspect
    assert "device_count" not in inspect.getsource(bench.launch_ranks)


def test_bench_refuses_more_gpus_than_the_node_has():
    """a box with fewer devices than --gpus must fail loudly, never report a single-GPU number as the N-GPU one"""
    n = torch.cuda.device_count() + 64
    rc, line, err = _run_bench("--gpus", str(n), "--steps", "1", "--warmup", "0")
    assert rc != 0 and line is None
    assert f"--gpus {n}" in err and "visible" in err


def test_bench_rank_failure_is_the_launchers_failure():
    """--gpus disagreeing with the WORLD_SIZE a launcher set is refused by every rank"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "stub", "--gpus", "2"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr
