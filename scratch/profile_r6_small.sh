R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_ref /tmp/p_pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ref -- python3 $R/bench.py --workload doggo-ref-16env-2x64 --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-power > $O/ref16_under_rocprof.json 2>/dev/null
cp $(find /tmp/p_ref -name "*kernel_stats.csv" | head -1) $O/ref16_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_pt -- python3 $R/bench.py --workload point-1024env-2x64 --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-power > $O/point_under_rocprof.json 2>/dev/null
cp $(find /tmp/p_pt -name "*kernel_stats.csv" | head -1) $O/point_kernel_stats.csv
cd $R
python3 bench.py --workload doggo-ref-16env-2x64 --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-power > $O/bench_doggo_ref16.json 2>/dev/null
python3 bench.py --workload point-1024env-2x64 --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-power > $O/bench_point_2x64.json 2>/dev/null
for f in ref16_under_rocprof point_under_rocprof bench_doggo_ref16 bench_point_2x64; do python3 -c "import sys,json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']/1e6,3), round(d['ms_per_step'],3))"; done
head -12 $O/ref16_kernel_stats.csv | cut -c1-160
head -12 $O/point_kernel_stats.csv | cut -c1-160
