"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

usage: python profiles/tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/rN/hbm_traffic_pmc.json
Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB-ish
units of 1 KB; FETCH_SIZE on gfx950 reports half of the bytes of coalesced streaming reads -> x2.
"""
import collections, csv, glob, hashlib, json, os, statistics, sys


def csrc_sha256():
    """Fingerprint of the kernel sources the profile was taken on (same function as bench.csrc_sha256)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "mobrob_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def load(d):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                 "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline`, MI355X",
       "note": "FETCH_SIZE on gfx950 reports 1/2 of the bytes of coalesced streaming reads "
               "(MI355X_MICROARCH.md HBM section): corrected = 2 x FETCH_SIZE. WRITE_SIZE is exact.",
       "csrc_sha256": csrc_sha256(),
       "kernels": {}}
for k in fetch:
    if not k.startswith(('void mobrob', 'mobrob')):
        continue
    fk, wk = statistics.median(fetch[k]), statistics.median(write.get(k, [0.0]))
    out["kernels"][k] = {"FETCH_SIZE_KB_median_per_launch": fk, "WRITE_SIZE_KB_median_per_launch": wk,
                         "launches": len(fetch[k]),
                         "hbm_bytes_per_launch_corrected": int(2 * fk * 1024 + wk * 1024)}
json.dump(out, sys.stdout, indent=1)
