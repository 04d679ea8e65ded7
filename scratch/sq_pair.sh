# SQ counters of the 64-wide gradient kernels on BASELINE config 2 (pair kernel and block kernel): gpurun -- 'bash scratch/sq_pair.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in pair block; do
  if [ $v = block ]; then export MOBROB_PAIR64_MIN_TILES=0; else unset MOBROB_PAIR64_MIN_TILES; fi
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"; do
    rm -rf /tmp/p_sq
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p_sq -- python3 $R/bench.py --workload point-1024env-2x64 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
    echo "== $v: $set"
    python3 - <<'PY'
import csv, collections, statistics, glob
f = glob.glob('/tmp/p_sq/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].split('(')[0][:40]
    if 'train' in name:
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    c = {n: statistics.median(v) for n, v in d.items()}
    wc = c.get('SQ_WAVE_CYCLES', 1)
    print(k, {n: "%.3g (%.3f of wave cycles)" % (v, v / wc) for n, v in c.items()})
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c: print("   mfma busy / (4 * busy cycles):", c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CYCLES']), " mfma/(4*wave):", c['SQ_VALU_MFMA_BUSY_CYCLES']/(4*wc))
PY
  done
done
