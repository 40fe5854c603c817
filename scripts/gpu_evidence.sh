#!/bin/bash
# The ONE GPU-box script: every measurement under profiles/ is made by a mode of this file (run through scripts/run_gpu.sh, which builds first).
#   scripts/run_gpu.sh bash scripts/gpu_evidence.sh MODE [args]        outputs under gpurun_out/$TAG/  (TAG defaults to the mode's name)
# modes
#   check                      build() + smoke() + the GPU suite + the default bench line (what the driver runs at round end)
#   tests [pytest args]        the GPU suite (or a selection: tests -k bpipe)
#   line WORKLOAD [bench args] one bench line                           -> bench_WORKLOAD[_clipsN].json
#   lines                      a bench line of every workload + cfg 4 at 64 / 128 / 256 clips
#   stats WORKLOAD [args]      rocprofv3 --kernel-trace --stats of a short bench run -> WORKLOAD_kernel_stats.csv
#   pmc WORKLOAD [CLIPS]       FETCH_SIZE / WRITE_SIZE passes (separate runs) of scripts/pmc_target.py -> pmc_{fetch,write}_size_*_summary.csv
#   mfma [CLIPS ...]           SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE of the cfg-4 launch at the given clip counts
#   sq WORKLOAD                SQ wave / wait / LDS counters of a feature kernel
#   soak CLIPS [PASSES]        PASSES (20) full passes of cfg 4 at CLIPS per GPU; counts redone batches (a timed-out hand-off)
#   stamps CLIPS [STAGE]       diagnostic build (LIB=<variant built with `build_variant.sh .. diag`> for another): phase stamps of one stage's waves
#   abw "WORKLOADS"            every library under mimikit_amd/variants/ against the product library on bench.py workloads (dominant kernel's launch time)
#   ab CLIPS_A CLIPS_B [STEPS] every library under mimikit_amd/variants/ (scripts/build_variant.sh) against the product library: us per AR step of
#                              scripts/bpipe_check.py.  Variants are loaded BY PATH (MMK_DIAG_LIB=<file>): the product library is never overwritten
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mode=$1; shift
TAG=${TAG:-$mode}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R

trim() { find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*kernel_trace.csv" -size +4M -delete; }
fields() { grep -o '"value": [0-9.]*\|"us_per_ar_step": [0-9.]*\|"us_per_step_in_kernel": [0-9.]*\|"frac": [0-9.]*\|"avg_launch_us": [0-9.]*' $1 | head -${2:-6} | tr '\n' ' '; }

case $mode in
check)
  python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" 2>&1 | tail -3
  timeout 3000 python -m pytest tests -m gpu -q --timeout=900 -x 2>&1 | tail -5
  python bench.py 2> $O/bench.err | tee $O/bench_default.json | cut -c1-400
  ;;
tests)
  timeout 3000 python -m pytest tests -m gpu -q --timeout=900 "$@" 2>&1 | tail -15
  ;;
line)
  wl=$1; shift
  name=bench_$wl$(echo "$*" | grep -o -- "--clips [0-9]*" | sed 's/--clips /_clips/')
  timeout 900 python bench.py --workload $wl "$@" > $O/$name.json 2> $O/$name.err
  echo "$name exit $?: $(fields $O/$name.json)"
  ;;
lines)
  for wl in wavenet_cfg2 srnn_cfg3 s2s_cfg5 mulaw stft istft gla; do
    case $wl in mulaw|stft|istft) k="--steps 20 --warmup 5";; *) k="";; esac      # (sub-millisecond passes: two of them would time the launch overhead)
    timeout 600 python bench.py --workload $wl --no-others $k > $O/bench_$wl.json 2> $O/bench_$wl.err
    echo "$wl exit $?: $(fields $O/bench_$wl.json)"
  done
  for n in 64 128 256; do
    timeout 900 python bench.py --clips $n --no-cpu-baseline --no-strong-leg --no-others > $O/bench_wavenet_cfg4_clips$n.json 2> $O/bench_clips$n.err
    echo "cfg4 clips $n exit $?: $(fields $O/bench_wavenet_cfg4_clips$n.json)"
  done
  ;;
stats)
  wl=$1; shift
  cd /tmp; rm -rf $O/stats_$wl
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-strong-leg --no-others "$@" > $O/stats_$wl.log 2>&1
  echo "stats exit $?"
  f=$(find $O/stats_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${wl}_kernel_stats.csv && head -6 $f | cut -c1-220
  rm -rf $O/stats_$wl
  ;;
pmc)
  wl=$1; clips=${2:-0}
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$C
    WORKLOAD=$wl CLIPS=$clips timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python3 $R/scripts/pmc_target.py > $O/pmc_$C.log 2>&1
    echo "pmc $C exit: $?"
  done
  cd $R
  sfx=$wl$([ "$clips" != 0 ] && echo _clips$clips)
  python scripts/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_fetch_size_${sfx}_summary.csv | head -6
  python scripts/pmc_summary.py $O/pmc_WRITE_SIZE $O/pmc_write_size_${sfx}_summary.csv | head -6
  trim
  ;;
mfma)
  cd /tmp
  for n in ${@:-256 128}; do
    rm -rf $O/pmc_$n
    CLIPS=$n timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/scripts/pmc_target.py > $O/pmc_$n.log 2>&1
    echo "clips $n exit: $?"
  done
  cd $R
  python3 scripts/pmc_summary.py --mfma $O $O/pmc_mfma_summary.csv
  trim
  ;;
sq)
  wl=$1
  cd /tmp; rm -rf $O/sq_$wl
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-others > $O/sq_$wl.log 2>&1
  echo "sq exit $?"
  cd $R
  python3 scripts/pmc_summary.py --raw $O/sq_$wl $O/pmc_sq_${wl}_summary.csv | head -8
  trim
  ;;
sqring)
  # SQ counters of the cfg-4 step kernel at CLIPS clips with plan switches: sqring 128 "MMK_WN_BPIPE=0,MMK_WN_SPIPE_PAIR=1" [name]
  n=$1; tun=$2; name=${3:-ring}
  cd /tmp
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1)); rm -rf $O/sq_${name}_$i
    CLIPS=$n TUNING="$tun" timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq_${name}_$i -- python3 $R/scripts/pmc_target.py > $O/sq_${name}_$i.log 2>&1
    echo "set $i exit $?"
  done
  cd $R
  python3 scripts/pmc_summary.py --raw $O $O/pmc_sq_${name}_clips${n}_summary.csv | grep "pipe" | head -14
  trim
  ;;
soak)
  n=$1; passes=${2:-20}
  timeout 2400 python bench.py --clips $n --steps $passes --warmup 1 --no-cpu-baseline --no-strong-leg --no-others > $O/soak_$n.json 2> $O/soak_$n.err
  echo "clips $n exit $?: $(fields $O/soak_$n.json 2) redone batches: $(grep -c "regenerating\|timed out" $O/soak_$n.err)"
  ;;
stamps)
  n=$1; st=${2:-13}
  MMK_DIAG_LIB=${LIB:-1} MMK_WN_STAMPS=1 MMK_WN_STAMP_STAGE=$st timeout 600 python scripts/bpipe_check.py --only ${ONLY:-bpipe} --clips $n --steps ${STEPS:-256} 2>&1 | grep -v amdgpu.ids | tee $O/stamps_clips${n}_stage$st.log | cut -c1-1800
  ;;
ab)
  a=$1; b=$2; steps=${3:-256}
  for rep in 1 2; do
    for v in mimikit_amd/libmmk_hip.so mimikit_amd/variants/libmmk_*.so; do
      lib=$([ $v = mimikit_amd/libmmk_hip.so ] && echo "" || echo $R/$v)
      r1=$(MMK_DIAG_LIB=$lib timeout 300 python scripts/bpipe_check.py --only ${ONLY:-bpipe} --clips $a --steps $steps 2>/dev/null | grep -o "[0-9.]* us per step")
      r2=$(MMK_DIAG_LIB=$lib timeout 300 python scripts/bpipe_check.py --only ${ONLY:-bpipe} --clips $b --steps $steps 2>/dev/null | grep -o "[0-9.]* us per step")
      echo "$(basename $v) $a: $r1 | $b: $r2" | tee -a $O/ab.log
    done
  done
  ;;
abw)
  # every variant against the product library on bench workloads: scripts/gpu_evidence.sh abw "stft istft gla"
  for rep in 1 2; do
    for v in mimikit_amd/libmmk_hip.so mimikit_amd/variants/libmmk_*.so; do
      lib=$([ $v = mimikit_amd/libmmk_hip.so ] && echo "" || echo $R/$v)
      line="$(basename $v)"
      for wl in $1; do
        r=$(MMK_DIAG_LIB=$lib timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-others 2>/dev/null | grep -o '"avg_launch_us": [0-9.]*\|"us_per_step[a-z_]*": [0-9.]*\|"us_per_generate_step": [0-9.]*' | head -2 | tr '\n' ' ')
        line="$line | $wl: $r"
      done
      echo "$line" | tee -a $O/abw.log
    done
  done
  ;;
*)
  echo "unknown mode $mode"; exit 2;;
esac
