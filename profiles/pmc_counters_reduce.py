"""Per-launch and per-wavefront means of the step kernel's counters from rocprofv3 --pmc passes (one directory per pass)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"kernel": None, "workload": "20 000 objects = 5000 wavefronts, profiles/pmc_workload.py", "counters": {}}
for d in sys.argv[1:]:
    fs = glob.glob(os.path.join(ROOT, d, '*', '*counter_collection.csv'))
    if not fs:
        out["counters"][os.path.basename(d)] = "pass produced no counter file (counter not available on this build?)"
        continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if 'step_fast_kernel' not in r['Kernel_Name']:
            continue
        if out["kernel"] is None:
            out["kernel"] = r['Kernel_Name'].split('(')[0]
        acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for k, v in acc.items():
        v = v[20:] if len(v) > 40 else v
        out["counters"][k] = {"mean_per_launch": sum(v) / len(v), "per_wavefront": sum(v) / len(v) / 5000.0, "launches": len(v)}
print(json.dumps(out, indent=1))
