# FETCH / WRITE passes of the headline bench on the current build + a 10-step bench line:  gpurun -- 'bash scratch/pmc_quick.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_quick
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_f /tmp/p_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also > /dev/null 2>&1
python3 $R/profiles/tools/pmc_traffic.py /tmp/p_f /tmp/p_w > $O/hbm_traffic_pmc.json
python3 -c "
import json; d=json.load(open('$O/hbm_traffic_pmc.json'))
for k,v in d['kernels'].items():
    if 'k_fused_train' in k or 'k_slab_reduce' in k or 'records' in k: print(k[:60], v)"
cd $R
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/bench.json 2>/dev/null
python3 -c "import json; d=json.load(open('$O/bench.json')); print(round(d['value']/1e6,3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['roofline']['avg_launch_ms'])"
