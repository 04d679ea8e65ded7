# A/B of libraries on the un-instrumented headline bench:  bash scratch/ab_libs.sh lib1.so lib2.so ...
for rep in 1 2; do
for L in "$@"; do
    MOBROB_PPO_LIB=$L python3 bench.py --no-cpu-baseline --no-also --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms']*1e3,1), round(d['roofline']['frac'],4))"
done
done
