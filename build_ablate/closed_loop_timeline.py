"""timeline of two consecutive steps of closed_loop_kernel (diagnostic build -DSSA_CL_TRACE=<step> -> build_ablate/cl/trace.so)"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench  # noqa: E402
from ssa_gym_amd import _lib, _build  # noqa: E402
_build.LIB = os.path.join(ROOT, 'build_ablate', 'cl', 'trace.so')
lib = _lib.load()
m = int(os.environ.get('M', '20000'))
r = bench.closed_loop_rate(m, 240, 0, agent=int(os.environ.get('AGENT', '1')), chunk=240)
print(r["ms_per_step"] * 1e3, "us/step")
nb = (m + 3) // 4
buf = np.zeros((8192, 16), dtype=np.uint64)
lib.ssa_debug_cl_trace_copy.argtypes = [C.c_void_p, C.c_int64]
lib.ssa_debug_cl_trace_copy.restype = C.c_int
assert lib.ssa_debug_cl_trace_copy(buf.ctypes.data, buf.nbytes) == 0
tr = buf[:nb].astype(np.int64)
names = ["step start", "wait begin", "decision seen", "step body done", "part announced", "-", "tile stored"]
for s in range(2):
    t = tr[:, 8 * s:8 * s + 7] * 10.0 / 1e3      # us
    fl = tr[:, 8 * s + 7]
    dec = np.median(t[:, 2])                      # decision of this step seen (most waves: at once)
    print("---- step %d: all times relative to the median 'decision seen' of this step" % s)
    for k, nme in enumerate(names):
        if not np.any(tr[:, 8 * s + k]):      # (a marker this build never stamps: service wavefronts fold, compute wavefronts do not)
            continue
        v = t[:, k] - dec
        print("  %-24s p1 %7.2f  p50 %7.2f  p90 %7.2f  p99 %7.2f  max %7.2f" % (nme, *np.percentile(v, [1, 50, 90, 99, 100])))
    waited = (t[:, 2] - t[:, 1])
    print("  waited for the decision: mean %.2f us, >0.3 us: %d waves" % (waited.mean(), int((waited > 0.3).sum())))
    last = np.argsort(t[:, 4])[-6:]
    print("  last arrivers (wave, arrival, flags[1 = ran the update], wait begin, decision seen, body done):")
    for w in last:
        print("     w%5d  %7.2f  fl %d   %7.2f %7.2f %7.2f" % (w, t[w, 4] - dec, fl[w], t[w, 1] - dec, t[w, 2] - dec, t[w, 3] - dec))
    fin = np.where(fl & 2)[0]
    for w in fin:
        print("  FINAL folder w%d: arrived %.2f, folds done (decision published) %.2f" % (w, t[w, 4] - dec, t[w, 5] - dec))
    if s == 0:
        d1 = np.median(tr[:, 8 + 2] * 10.0 / 1e3)
        print("  next decision seen (median) at +%.2f us" % (d1 - dec))
