#!/bin/bash
# round 4 evidence, final tree (r04_v4): the whole GPU suite, the default bench line, 64 / 128 / 256 clips per GPU, the other workloads' lines,
# rocprofv3 kernel stats of the default bench command, the clips curve
mkdir -p gpurun_out/r04d
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/r04d/bench_wavenet_cfg4.json 2> gpurun_out/r04d/bench_wavenet_cfg4.err; echo "default bench exit $?"; cut -c1-300 gpurun_out/r04d/bench_wavenet_cfg4.json
for c in 64 128 256; do
  timeout 900 python bench.py --clips $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04d/bench_wavenet_cfg4_clips$c.json 2> /dev/null; echo "clips $c exit $?"
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"us_per_step_in_kernel": [0-9.]*' gpurun_out/r04d/bench_wavenet_cfg4_clips$c.json | tr '\n' ' '; echo
done
for WL in wavenet_cfg2 srnn_cfg3 s2s_cfg5 mulaw stft istft gla; do
  timeout 600 python bench.py --workload $WL > gpurun_out/r04d/bench_$WL.json 2> /dev/null; echo "== $WL exit $?"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r04d/bench_$WL.json | tr '\n' ' '; echo
done
OUT=$R/gpurun_out/prof_cfg4_$$; rm -rf $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline > $OUT.log 2>&1 ); echo "rocprof cfg4 exit $?"
f=$(find $OUT -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04d/wavenet_cfg4_kernel_stats.csv; grep '^{' $OUT.log | tail -1 > gpurun_out/r04d/wavenet_cfg4_profiled_bench_line.json; head -7 gpurun_out/r04d/wavenet_cfg4_kernel_stats.csv | cut -c1-200; rm -rf $OUT
for c in 8 16 24 28 32 36 40 48 64 96 128; do
  r=$(timeout 300 python bench.py --clips $c --no-cpu-baseline --steps 1 --warmup 1 --seconds 0.128 2>/dev/null | grep -o "\"us_per_step_in_kernel\": [0-9.]*" | grep -o "[0-9.]*$")
  echo "clips $c us_per_step $r"
done 2>&1 | tee gpurun_out/r04d/clips_curve.log
echo "== own launcher on a one-GPU box (must fail loudly)"; python bench.py --gpus 2 --steps 1 --warmup 0 --seconds 0.05 --no-cpu-baseline; echo "exit $?"
