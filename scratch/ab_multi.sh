# same-box comparison of several engine libraries on the headline bench: bash scratch/ab_multi.sh <reps> <lib> <lib> ...
# (stderr of a failing bench run is shown, not swallowed)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for l in "$@"; do
    MOBROB_PPO_LIB=$R/$l python3 $R/bench.py --steps ${AB_STEPS:-6} --warmup 2 --no-cpu-baseline --no-also > /tmp/ab_line.json 2> /tmp/ab_err.txt
    python3 - "$l" <<'PY' || tail -5 /tmp/ab_err.txt
import json, sys
d = json.loads(open('/tmp/ab_line.json').read())
print(sys.argv[1], round(d['value'] / 1e6, 3), 'M env-steps/s', round(d['ms_per_step'], 2), 'ms/step  frac', round(d['roofline']['frac'], 4),
      ' launch', round(1e3 * d['roofline']['avg_launch_ms'], 1), 'us')
PY
  done
done
