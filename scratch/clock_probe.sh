#!/bin/bash
# sample engine clock / power while bench.py runs (diagnostic: is the train kernel power limited?)
python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor junction\)|mclk" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.4
done
wait $BP
tail -c 400 gpurun_out/clock_bench.json
