# kernel-trace stats of the headline bench with another library build: bash scratch/kt_lib.sh <lib> <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_kt_$2
MOBROB_PPO_LIB=$R/$1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt_$2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2>&1
head -7 $(find /tmp/p_kt_$2 -name "*kernel_stats.csv" | head -1) | cut -c1-110
