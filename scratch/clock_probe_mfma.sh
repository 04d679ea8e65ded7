#!/bin/bash
# rocm-smi clock / power samples beside scratch/mfma_power_probe (what the matrix pipes sustain under the 1400 W package cap)
#   gpurun -- 'bash scratch/clock_probe_mfma.sh > gpurun_out/r5/mfma_power_probe.txt 2>&1'
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | tr -s ' \t' ' '
scratch/mfma_power_probe ${1:-5} > gpurun_out/mfma_power_probe.out 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  echo "[t=$(date +%s)] $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Package Power' | sed -E 's/.*(sclk clock level: [0-9S]+: \([0-9]+Mhz\)|Power \(W\): [0-9.]+).*/\1/' | tr '\n' ' ')"
  sleep 0.5
done
cat gpurun_out/mfma_power_probe.out
