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
key = '%s_%s' % (os.environ.get('PROP', 'hybrid'), os.environ.get('M', '20000'))     # (pmc_workload.py's defaults: the env default variant)
out[key] = round(f / cf + w / cw)      # corrected HBM(+Infinity-Cache-side) bytes per launch
out['algorithmic_bytes'] = 896 * int(os.environ.get('M', '20000'))
print(json.dumps(out, indent=1))
# every configuration is merged into the committed file under its own key (bench.py looks its own key up); `collected` says which round's
# passes each key comes from (TAG = the round, as collect.sh passes it)
dst = os.path.join(ROOT, 'profiles', 'traffic.json')
old = json.load(open(dst)) if os.path.exists(dst) else {}
old[key] = out[key]
old.setdefault('other_configurations', {})[key] = {'step_raw': out['step_raw'], 'calib_observe': out['calib_observe'], 'calib_propagate': out['calib_propagate']}
if not isinstance(old.get('collected'), dict):
    old['collected'] = {}
old['collected'][key] = os.environ.get('TAG', 'unlabelled')
old['algorithmic_bytes_20000'] = 896 * 20000
json.dump(old, open(dst, 'w'), indent=1)
