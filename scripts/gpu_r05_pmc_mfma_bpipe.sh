#!/bin/bash
# round 5: matrix-pipe busy counters of the 256-clip and 128-clip launches of wavenet_bpipe_kernel (separate rocprofv3 --pmc passes, kernel trace only)
mkdir -p gpurun_out/r05m
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for n in 256 128; do
  rm -rf $R/gpurun_out/r05m/pmc_$n
  CLIPS=$n timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r05m/pmc_$n -- python3 $R/scripts/pmc_target.py > $R/gpurun_out/r05m/pmc_$n.log 2>&1
  echo "clips $n exit: $?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
with open("gpurun_out/r05m/pmc_mfma_bpipe_summary.csv", "w") as out:
    out.write("clips,kernel,dispatches,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,GRBM_GUI_ACTIVE\n")
    for n in (256, 128):
        for f in glob.glob(f"gpurun_out/r05m/pmc_{n}/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
            for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:3]:
                line = f"{n},\"{k}\",{cnt[k]},{v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.0f},{v.get('SQ_BUSY_CYCLES', 0):.0f},{v.get('GRBM_GUI_ACTIVE', 0):.0f}"
                out.write(line + "\n"); print(line)
PY
find gpurun_out/r05m -name "*counter_collection.csv" -size +4M -delete
find gpurun_out/r05m -name "*kernel_trace.csv" -size +4M -delete
