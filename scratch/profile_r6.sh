# Reproduces profiles/r6 on a gpurun box: kernel trace + FETCH / WRITE / SQ counter passes (separate runs) + bench lines.
#   gpurun --timeout 2400 -- 'bash scratch/profile_r6.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w /tmp/p_sq
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-power > $O/fused_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/p_kt -name "*kernel_stats.csv" | head -1) $O/fused_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-power > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-power > /dev/null 2>&1
python3 $R/profiles/tools/pmc_traffic.py /tmp/p_f /tmp/p_w > $O/hbm_traffic_pmc.json
mkdir -p $R/profiles/r6; cp $O/hbm_traffic_pmc.json $R/profiles/r6/hbm_traffic_pmc.json   # the bench lines below report this run's traffic figure (same sources)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-power > /dev/null 2>&1
python3 $R/profiles/tools/pmc_sq_summary.py /tmp/p_sq > $O/sq_counters_summary.txt 2>&1
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench_final.json 2>$O/bench_final.err
python3 bench.py --phases --no-cpu-baseline --no-also --steps 5 --warmup 2 > $O/bench_final_phases.json 2>/dev/null
python3 bench.py --workload point-1024env-2x64 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_point_2x64.json 2>/dev/null
python3 bench.py --workload fleet-car-drone-turtlebot3-2x64 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_fleet.json 2>/dev/null
python3 bench.py --workload doggo-ref-16env-2x64 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_doggo_ref16.json 2>/dev/null
MOBROB_FORCE_DP=1 python3 bench.py --no-cpu-baseline --no-also --steps 5 --warmup 2 > $O/bench_forced_dp_world1.json 2>/dev/null
MOBROB_DP_SAME_DEVICE=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearsal_dp2_callback.json 2>/dev/null
MOBROB_DP_SAME_DEVICE=1 MOBROB_ONESHOT_AR=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearsal_dp2_oneshot.json 2>/dev/null
python3 bench.py --only-host-path > $O/host_path_only.json 2>/dev/null
for f in bench_final bench_final_phases bench_point_2x64 bench_fleet bench_doggo_ref16 bench_forced_dp_world1 rehearsal_dp2_callback rehearsal_dp2_oneshot; do python3 -c "import sys,json; d=json.load(open('$O/$f.json')); print('$f', d['value'] and round(d['value']/1e6,3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['roofline'].get('avg_launch_ms'), d['roofline'].get('traffic'), d.get('cpu_baseline',{}).get('value'), d.get('exchange'), d.get('exchange_selfcheck'))"; done
head -8 $O/fused_kernel_stats.csv | cut -c1-150
cat $O/sq_counters_summary.txt | head -30
