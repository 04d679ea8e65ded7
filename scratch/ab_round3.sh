#!/bin/bash
# round-3 A/B runs (one gpurun call): norm records at 2x256, dW3 inside dh2, rollout phase ablation
cd "$(dirname "$0")/.."
b() { python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import json,sys; o=json.load(sys.stdin); print('%.3f M  %.2f ms/iter  k_fused_train %.4f ms  phases %s' % (o['value']/1e6, o['ms_per_step'], o['roofline']['avg_launch_ms'], {k: round(v,2) for k,v in o['phase_ms_per_step'].items()}))"; }
echo "== default build";                 b; b
echo "== MOBROB_NO_NORM_RECORDS=1";      MOBROB_NO_NORM_RECORDS=1 b; MOBROB_NO_NORM_RECORDS=1 b
echo "== dW3 and dh2 as two phases";     MOBROB_PPO_LIB=scratch/lib_dw3_split.so b; MOBROB_PPO_LIB=scratch/lib_dw3_split.so b
echo "== default, all phases bracketed"; b --phases
echo "== records off, all phases";       MOBROB_NO_NORM_RECORDS=1 b --phases
echo "== rollout ablation"; python scratch/time_rollout.py x 0 1 8 16 25 153 0
