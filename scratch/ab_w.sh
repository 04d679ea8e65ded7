# same-box A/B of two engine libraries on another workload: bash scratch/ab_w.sh <workload> <libA> <libB> [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}
W=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for l in $A $B; do
    MOBROB_PPO_LIB=$R/$l python3 $R/bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['value']/1e6,3), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step  frac', round(d['roofline']['frac'],4), ' launch', round(1e3*d['roofline']['avg_launch_ms'],1), 'us')"
  done
done
