# L2 (TCC) hit / miss / fabric-read counters of the gradient kernel:  gpurun -- 'bash scratch/pmc_tcc.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_tcc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf /tmp/p_$n
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also > /dev/null 2>$O/err_$n.txt
  python3 - "$n" /tmp/p_$n >> $O/tcc.txt <<'PY'
import sys, glob, csv, collections, statistics
f = glob.glob(sys.argv[2] + '/*/*counter_collection.csv')
if not f: print(sys.argv[1], 'no output'); sys.exit()
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    agg[(r['Kernel_Name'].split('(')[0][:50], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    if 'train' in k or 'slab' in k: print(f"{k:52s} {c:20s} median {statistics.median(v):14.0f}  launches {len(v)}")
PY
done
cat $O/tcc.txt
