#!/bin/bash
# round 3, everything but cfg 4: bench lines of every workload, Seq2Seq kernel stats / PMC traffic / phase stamps, SampleRNN PMC traffic
bash scripts/gpu_r03_all.sh
bash scripts/gpu_prof_s2s.sh | head -14 | cut -c1-200
bash scripts/gpu_pmc_s2s.sh | cut -c1-170
bash scripts/gpu_pmc_srnn.sh | cut -c1-170
bash scripts/gpu_s2s_stamps.sh | tail -19 | cut -c1-200
WORKLOAD=srnn_cfg3 PROF_SECONDS=1 bash scripts/gpu_prof_wl.sh 2>&1 | tail -8 | cut -c1-200
WORKLOAD=wavenet_cfg2 PROF_SECONDS=1 bash scripts/gpu_prof_wl.sh 2>&1 | tail -6 | cut -c1-200
