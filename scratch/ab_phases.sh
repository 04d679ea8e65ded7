# same-box A/B of two engine libraries, per-phase milliseconds: bash scratch/ab_phases.sh <libA> <libB> [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for l in $A $B; do
    MOBROB_PPO_LIB=$R/$l python3 $R/bench.py --phases --steps 4 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('$l', round(d['ms_per_step'],2), 'ms/step', {k: round(v,2) for k,v in p.items()})"
  done
done
