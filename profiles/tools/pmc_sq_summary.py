import csv, collections, statistics, sys, glob
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0][:40]
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for k in agg:
    if any(t in k for t in ('fused', 'slab', 'chain', 'rollout', 'value_batch', 'adam_pack')):
        c = {n: statistics.median(v) for n, v in agg[k].items()}
        d = statistics.median(dur[k])
        print(k, "median dur us", d / 1e3)
        for n, v in c.items():
            print("    %-32s %.4g" % (n, v))
        if 'SQ_WAVE_CYCLES' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            nw = 1024 if 'train' in k else 512
            print("    mfma util (busy cycles / (4*wave_cycles)):", c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_WAVE_CYCLES']))
            for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_INST_CYCLES_VMEM', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM'):
                if n in c:
                    print("    share %-24s %.3f" % (n, c[n] / c['SQ_WAVE_CYCLES']))
