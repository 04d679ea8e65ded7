#!/bin/bash
# sample engine clock / socket power while the headline bench runs (diagnostic: is k_chain_train power limited?), for the x3 engine
# and for the all-f32 engine (MOBROB_NO_X3=1).  gpurun -- 'bash scratch/clock_probe.sh > gpurun_out/r5/clock_probe.txt 2>&1'
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "power|cap" | tr -s ' '
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo " <- idle"
for mode in x3 f32; do
  if [ $mode = f32 ]; then export MOBROB_NO_X3=1; else unset MOBROB_NO_X3; fi
  python bench.py --steps 70 --warmup 3 --no-cpu-baseline --no-also --no-host-path > gpurun_out/clock_bench_$mode.json 2>/dev/null &
  BP=$!
  sleep 9
  for i in $(seq 1 10); do
    rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor junction\)|fclk" | tr -s ' ' | tr '\n' ';'
    echo " <- $mode"
    sleep 0.3
  done
  wait $BP
  python3 -c "
import json; d = json.loads(open('gpurun_out/clock_bench_$mode.json').read().strip().splitlines()[-1])
print('$mode', d['ms_per_step'], 'ms per iteration;', d['roofline'].get('us_per_launch', d['roofline']), )"
done
