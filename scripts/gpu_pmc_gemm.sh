#!/bin/bash
# where the tiled GEMM's cycles go: separate rocprofv3 --pmc passes (kernel-trace only), Seq2Seq cfg 5 bench as the target;
# sums per kernel name for the counters of each pass
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm_$i -- python3 $R/bench.py --workload s2s_cfg5 --steps 1 --warmup 0 --seconds 0.1 --no-cpu-baseline > $R/gpurun_out/pmc_gemm_$i.log 2>&1
  echo "pass $i ($set): exit $?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_gemm_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        if "gemm_bias_act" in k or "lstm_step_kernel<8, 2, false" in row["Kernel_Name"]:
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
with open("gpurun_out/pmc_gemm_summary.txt", "w") as out:
    for k, v in acc.items():
        out.write(k + "\n")
        for name, val in sorted(v.items()):
            out.write(f"  {name:34s} {val:16.0f}\n")
print(open("gpurun_out/pmc_gemm_summary.txt").read())
PY
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
