# A/B of libraries on the headline bench with phase brackets:  bash scratch/ab_reduce.sh lib1.so lib2.so ...
for L in "$@"; do
  for rep in 1 2; do
    MOBROB_PPO_LIB=$L python3 bench.py --phases --no-cpu-baseline --no-also --steps 4 --warmup 2 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('$L', round(d['ms_per_step'],2), {k:round(v,3) for k,v in p.items() if k in ('grad_reduce','apply','train_grad')})"
  done
done
