"""register / scratch / LDS figures of every hot kernel of a built library (LIB=path, default the in-tree one), and where the
scratch accesses of each sit relative to the out-of-line calls (what tests/test_abi_and_host.py asserts, as a table)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("LIB", os.path.join(ROOT, "ssa-gym_amd", "libssa_hip.so"))
b = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp()
fb, co = os.path.join(tmp, "fb.bin"), os.path.join(tmp, "dev.co")
subprocess.check_call([os.path.join(b, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, lib, os.path.join(tmp, "copy.so")])
subprocess.check_call([os.path.join(b, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co])
notes = subprocess.check_output([os.path.join(b, "llvm-readelf"), "--notes", co], text=True)
dis = subprocess.check_output([os.path.join(b, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(dis)
kern = {}
for blk in notes.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    kern[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "group_segment_fixed_size")}
bodies = re.split(r"\n[0-9a-f]+ <([^>]+)>:\n", dis)
for name, body in zip(bodies[1::2], bodies[2::2]):
    if not ("step_fast_kernel" in name or "rollout_kernel" in name or "closed_loop_kernel" in name):
        continue
    ins = [ln.split()[0] for ln in body.splitlines() if ln.strip() and not ln.strip().startswith(("//", ";"))]
    calls = [i for i, op in enumerate(ins) if op == "s_swappc_b64"]
    scr = [i for i, op in enumerate(ins) if op.startswith("scratch_")]
    stray = [i for i in scr if not (calls and min(abs(i - c) for c in calls) <= 96)]
    k = kern.get(name, {})
    mm = re.search(r"(step_fast_kernel|rollout_kernel|closed_loop_kernel)ILi(\d)E(Lb(\d)E)?", name)
    short = "%s<%s%s>" % (mm.group(1), mm.group(2), (", multi" if mm.group(4) == "1" else "") if mm.group(3) else "")
    print("%-62s vgpr %3d sgpr %3d scratch %4d spills %3d lds %5d  instrs %6d calls %2d scratch-ops %3d stray %s" % (
        short, k.get("vgpr_count", -1), k.get("sgpr_count", -1), k.get("private_segment_fixed_size", -1), k.get("vgpr_spill_count", -1),
        k.get("group_segment_fixed_size", -1), len(ins), len(calls), len(scr), stray[:6]))
