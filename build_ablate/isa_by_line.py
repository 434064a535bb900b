"""static instruction count of one kernel of the library's ISA, attributed to source lines (diagnostic).

    hipcc ... -gline-tables-only -S --cuda-device-only -o k.s ssa_kernels.hip
    python build_ablate/isa_by_line.py k.s '_ZN3ssa16step_fast_kernelILi1ELb0EEEvNS_5StepKEii' [--ops]

Prints, per (file, line), the number of VALU / SALU / LDS / VMEM instructions; lines are the innermost
inlined location (.loc).  With --ops also the most frequent opcodes."""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
want_ops = "--ops" in sys.argv
files = {}
rows = collections.defaultdict(lambda: collections.Counter())
ops = collections.Counter()
ops_by_line = collections.defaultdict(collections.Counter)
inside = False
cur = (0, 0)
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if s.startswith(sym + ":"):
        inside = True
        continue
    if inside and s.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith((".", ";")) or s.endswith(":"):
        continue
    op = s.split()[0]
    kind = ("VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_")
            else "VMEM" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "OTHER")
    rows[cur][kind] += 1
    ops[op] += 1
    ops_by_line[cur][op] += 1
tot = collections.Counter()
for k in sorted(rows):
    c = rows[k]
    tot.update(c)
    extra = ""
    if want_ops:
        extra = "  " + " ".join("%s:%d" % kv for kv in ops_by_line[k].most_common(6))
    print("%-18s %5d  VALU %4d SALU %4d LDS %3d VMEM %3d%s" % (files.get(k[0], "?"), k[1], c["VALU"], c["SALU"], c["LDS"], c["VMEM"], extra))
print("TOTAL", dict(tot))
print("top ops:", ops.most_common(40))
