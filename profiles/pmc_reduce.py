"""Reduces the two PMC passes to per-launch HBM traffic of the step kernel -> profiles/traffic.json."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def per_kernel(d, counter):
    f = glob.glob(os.path.join(ROOT, d, '*', '*counter_collection.csv'))[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter: continue
        k = r['Kernel_Name']
        key = 'step' if 'step_fast_kernel' in k else 'propagate' if 'propagate_kernel' in k else 'observe' if 'observe_kernel' in k else None
        if key: acc.setdefault(key, []).append(float(r['Counter_Value']))
    return acc
fe, wr = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
n = 1 << 20
known = {'propagate': (48 * n, 48 * n), 'observe': (384 * n, 128 * n)}
out = {}
for k in ('propagate', 'observe'):
    f = sum(fe[k][-3:]) / 3 * 1024; w = sum(wr[k][-3:]) / 3 * 1024
    out['calib_' + k] = {'fetch_counter_bytes': f, 'fetch_known_bytes': known[k][0], 'fetch_ratio': f / known[k][0],
                         'write_counter_bytes': w, 'write_known_bytes': known[k][1], 'write_ratio': w / known[k][1]}
fs = fe['step'][20:]; ws = wr['step'][20:]
f = sum(fs) / len(fs) * 1024; w = sum(ws) / len(ws) * 1024
cf = out['calib_observe']['fetch_ratio']; cw = out['calib_observe']['write_ratio']
out['step_raw'] = {'fetch_counter_bytes': f, 'write_counter_bytes': w, 'launches': len(fs)}
key = '%s_%s' % (os.environ.get('PROP', 'fg'), os.environ.get('M', '20000'))
out[key] = round(f / cf + w / cw)      # corrected HBM(+Infinity-Cache-side) bytes per launch
out['algorithmic_bytes'] = 896 * 20000
print(json.dumps(out, indent=1))
dst = os.path.join(ROOT, 'profiles', 'traffic.json')
if key != 'fg_20000' and os.path.exists(dst):      # the other configurations are merged into the file (bench.py looks its own key up)
    old = json.load(open(dst))
    old[key] = out[key]
    old.setdefault('other_configurations', {})[key] = {'step_raw': out['step_raw'], 'calib_observe': out['calib_observe']}
    out = old
json.dump(out, open(dst, 'w'), indent=1)
