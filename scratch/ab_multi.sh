# same-box comparison of several engine libraries on the headline bench: bash scratch/ab_multi.sh <reps> <lib> <lib> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for l in "$@"; do
    MOBROB_PPO_LIB=$R/$l python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['value']/1e6,3), 'M env-steps/s', round(d['ms_per_step'],2), 'ms/step  frac', round(d['roofline']['frac'],4), ' launch', round(1e3*d['roofline']['avg_launch_ms'],1), 'us')"
  done
done
