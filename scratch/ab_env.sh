# A/B of libraries: rollout phase of the headline bench:  bash scratch/ab_env.sh lib1.so ...
for rep in 1 2; do
for L in "$@"; do
  MOBROB_PPO_LIB=$L python3 bench.py --phases --no-cpu-baseline --no-also --steps 4 --warmup 2 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('$L', round(d['ms_per_step'],2), 'env', round(p['env'],3), 'act', round(p['act'],3))"
done
done
