"""differences the PMC counts of the truncation builds into per-stage dynamic instruction counts per wavefront."""
import csv, glob, os, sys
d = sys.argv[1]
names = ['t1', 't2', 't3', 't4', 't5', 't6', 't7', 't8', 't9', 'full']
stage = ['load', 'chol', 'kepler', 'moments', 'cov', 'update/status', 'observe+aer', 'store', 'stats', '(tail)']
tab = {}
for v in names:
    row = {}
    for ps in 'ab':
        for f in glob.glob(os.path.join(d, v + '_' + ps, '*', '*counter_collection.csv')):
            acc = {}
            for r in csv.DictReader(open(f)):
                if 'step_fast_kernel' in r['Kernel_Name']:
                    acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
            for k, x in acc.items():
                x = x[10:]
                row[k] = sum(x) / len(x)
    w = row.get('SQ_WAVES', 5000.0) or 5000.0
    tab[v] = {k: x / w for k, x in row.items() if k != 'SQ_WAVES'}
keys = ['SQ_INSTS_VALU', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_TRANS_F64', 'SQ_ACTIVE_INST_VALU',
        'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM']
print('%-14s' % 'stage' + ''.join('%10s' % k.replace('SQ_INSTS_', '').replace('SQ_', '')[:9] for k in keys) + '   non-fp VALU')
prev = {k: 0.0 for k in keys}
for v, st in zip(names, stage):
    cur = tab.get(v, {})
    dlt = {k: cur.get(k, float('nan')) - prev[k] for k in keys}
    fp = sum(dlt[k] for k in keys[1:5])
    print('%-14s' % st + ''.join('%10.1f' % dlt[k] for k in keys) + '   %8.1f' % (dlt[keys[0]] - fp))
    prev = {k: cur.get(k, float('nan')) for k in keys}
print('%-14s' % 'total' + ''.join('%10.1f' % prev[k] for k in keys))
