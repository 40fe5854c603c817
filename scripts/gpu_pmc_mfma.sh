#!/bin/bash
# MFMA-busy counters (separate rocprofv3 --pmc pass, kernel-trace only) for the persistent WaveNet kernel and the Seq2Seq kernels
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rm -rf $R/gpurun_out/pmc_mfma_wavenet      # a fresh directory per pass: the summary globs whatever lies in it
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma_wavenet -- python3 $R/scripts/pmc_target.py > $R/gpurun_out/pmc_mfma_wavenet.log 2>&1
echo "wavenet exit: $?"; tail -2 $R/gpurun_out/pmc_mfma_wavenet.log | cut -c1-300
rm -rf $R/gpurun_out/pmc_mfma_s2s      # a fresh directory per pass: the summary globs whatever lies in it
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma_s2s -- python3 $R/bench.py --workload s2s_cfg5 --steps 1 --warmup 0 --seconds 1 --no-cpu-baseline > $R/gpurun_out/pmc_mfma_s2s.log 2>&1
echo "s2s exit: $?"; tail -2 $R/gpurun_out/pmc_mfma_s2s.log | cut -c1-300
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ("wavenet", "s2s"):
    for f in glob.glob(f"gpurun_out/pmc_mfma_{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
        with open(f"gpurun_out/pmc_mfma_{tag}_summary.csv", "w") as out:
            out.write("kernel,dispatches,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,GRBM_GUI_ACTIVE\n")
            for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:8]:
                line = f"\"{k}\",{n[k]},{v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.0f},{v.get('SQ_BUSY_CYCLES', 0):.0f},{v.get('GRBM_GUI_ACTIVE', 0):.0f}"
                out.write(line + "\n"); print(tag, line)
PY
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
