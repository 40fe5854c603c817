"""The evidence under profiles/ is consistent with itself: no tracked file is a byte copy of another (round 3 shipped a cfg-2 trace
that was the previous round's file under a new name), every bench line that has a rocprofv3 kernel trace beside it names the kernel
that trace shows on top, and every PMC traffic entry says which kernel and commit it was collected on."""
import csv
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def _files():
    return sorted(f for f in os.listdir(PROFILES) if os.path.isfile(os.path.join(PROFILES, f)))


def test_no_two_profile_files_are_byte_identical():
    seen = {}
    for f in _files():
        with open(os.path.join(PROFILES, f), "rb") as fh:
            data = fh.read()
        if len(data) < 1024:        # (a one-line counter summary of a deterministic write count CAN repeat: ISTFT writes exactly its output every round)
            continue
        digest = hashlib.md5(data).hexdigest()
        assert digest not in seen, f"profiles/{f} is a byte copy of profiles/{seen[digest]}"
        seen[digest] = f


def _top_kernel(path):
    with open(path, newline="") as fh:
        rows = list(csv.DictReader(fh))
    rows = [r for r in rows if r.get("Name") and r.get("TotalDurationNs")]
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    m = re.search(r"(\w+_kernel\w*)", top["Name"])
    return (m.group(1) if m else top["Name"]), top["Name"]


def _bench_line(path):
    with open(path) as fh:
        lines = [ln for ln in fh.read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def test_bench_lines_name_the_kernel_their_trace_shows():
    """for every r<NN>_v<M>_bench_<workload>.json with r<NN>_v<M>_<workload>_kernel_stats.csv beside it (round 3 on: the rounds whose
    bench lines carry `roofline.kernel`)"""
    checked = 0
    for f in _files():
        m = re.fullmatch(r"(r(\d+)_v\d+)_bench_(\w+)\.json", f)
        if not m or int(m.group(2)) < 3:
            continue
        trace = os.path.join(PROFILES, f"{m.group(1)}_{m.group(3)}_kernel_stats.csv")
        if not os.path.exists(trace):
            continue
        line = _bench_line(os.path.join(PROFILES, f))
        if not line or "roofline" not in line or "kernel" not in line["roofline"]:
            continue
        short, full = _top_kernel(trace)
        assert short in line["roofline"]["kernel"], (f"profiles/{f} says its dominant kernel is '{line['roofline']['kernel'][:60]}...' but "
                                                     f"{os.path.basename(trace)} is led by '{full[:80]}'")
        checked += 1
    assert checked >= 3


def test_traffic_entries_say_what_they_were_measured_on():
    with open(os.path.join(PROFILES, "traffic.json")) as fh:
        traffic = json.load(fh)
    for workload, entries in traffic.items():
        if workload.startswith("_"):
            continue
        for name, e in entries.items():
            assert isinstance(e, dict) and e.get("kernel") and e.get("commit") and e.get("build"), f"traffic.json: {workload}.{name} lacks kernel / commit / build"
            for src in e.get("source", []):
                assert os.path.exists(os.path.join(PROFILES, src)), f"traffic.json: {workload}.{name} cites a missing file {src}"
