cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/p_kt -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python3 $R/profiles/tools/pmc_traffic.py /tmp/p_f /tmp/p_w > $R/gpurun_out/hbm_traffic_pmc.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python3 $R/profiles/tools/pmc_sq_summary.py /tmp/p_sq > $R/gpurun_out/sq_counters_summary.txt 2>&1
cd $R && python3 bench.py > gpurun_out/bench_final.json
head -c 600 gpurun_out/bench_final.json; echo; head -12 gpurun_out/kernel_stats.csv | cut -c1-150
python3 bench.py --phases --no-cpu-baseline > gpurun_out/bench_final_phases.json
python3 bench.py --workload point-1024env-2x64 --steps 3 --warmup 1 > gpurun_out/bench_point_2x64.json
python3 bench.py --workload fleet-car-drone-turtlebot3-2x64 --steps 3 --warmup 1 > gpurun_out/bench_fleet.json
python3 bench.py --workload doggo-ref-16env-2x64 --steps 3 --warmup 1 > gpurun_out/bench_doggo_ref16.json
python3 bench.py --workload doggo-4096env-2x256-hostenv --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_hostenv.json
for f in bench_final bench_final_phases bench_point_2x64 bench_fleet bench_doggo_ref16 bench_hostenv; do python3 -c "import sys,json; d=json.load(open('gpurun_out/$f.json')); print('$f', round(d['value']/1e6,3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['roofline']['avg_launch_ms'], d.get('cpu_baseline',{}).get('value'))"; done
