"""Per-wave phase timeline of step_fast_kernel (diagnostic build -DSSA_TRACE -> build_ablate/libs/trace.so).

    python build_ablate/wave_timeline.py          (loads build_ablate/libs/trace.so)

Each wavefront stamps the 100 MHz wall clock (s_memrealtime) at the phase boundaries of process_wave; the
script prints when waves start/end relative to the first one, the mean time per phase, and how the 5000
wavefronts of a 20 000-object step were spread over XCDs / CUs / SIMDs."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ['bench.py']
import bench
from ssa_gym_amd import host, engine, parallel, _lib, _build
_build.LIB = os.path.join(ROOT, os.environ.get("LIB", "build_ablate/libs/trace.so"))     # the -DSSA_TRACE build

m = int(os.environ.get("M", 20000))
pb = bench.build_problem(m, seed=100)
consts = host.make_consts(pb["Q"], pb["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, pb["obs_lla"], obs_type='aer',
                          propagator=os.environ.get("PROP", "fg"), covariance=os.environ.get("COV") or None)
z = torch.zeros((1, 480, m, 3), dtype=torch.float64, device='cuda')
eng = engine.HotPathEngine(consts, m, 1, pb["trans"], z, history=2)
if os.environ.get("LAYOUT") == "1":      # the engine's storage layout (objects of one orbit regime share wavefronts)
    from ssa_gym_amd.catalogue import regime_order
    eng.set_layout(regime_order(pb["x_true"]))
eng.load_state(0, pb["x_true"], pb["x"], np.broadcast_to(pb["P0"], (m, 6, 6)))
local = parallel.HipLocalStepper(eng, consts, fast_stats=True)
local.load_schedule(np.arange(4000) % m)
for k in range(int(os.environ.get("STEPS", 100))):
    local.step(-1)
local.flush()
if os.environ.get("ROLL"):      # the stamps then belong to the LAST step of a rollout launch (steady state)
    local.rollout(int(os.environ["ROLL"]))
torch.cuda.synchronize()
lib = _lib.load()
nb = (m + 3) // 4
buf = np.zeros((16384, 16), dtype=np.uint64)
lib.ssa_debug_trace_copy.argtypes = [C.c_void_p, C.c_int64]
assert lib.ssa_debug_trace_copy(buf.ctypes.data, buf.nbytes) == 0
tr = buf[:nb].astype(np.int64)
t = tr[:, :10] * 10.0   # ns
t0 = t[:, 0].min()
names = ["load", "chol", "kepler", "UT-mean", "cov", "update", "observe", "store", "stats"]
print("waves %d; first start -> last end: %.2f us" % (nb, (t[:, 9].max() - t0) / 1e3))
st = (t[:, 0] - t0) / 1e3
en = (t[:, 9] - t0) / 1e3
print("start time  percentiles [us] 0/25/50/75/90/99/100:", np.percentile(st, [0, 25, 50, 75, 90, 99, 100]).round(2))
print("end   time  percentiles [us] 0/25/50/75/90/99/100:", np.percentile(en, [0, 25, 50, 75, 90, 99, 100]).round(2))
life = en - st
print("wave lifetime [us] mean %.2f  p50 %.2f  p99 %.2f" % (life.mean(), np.median(life), np.percentile(life, 99)))
early = st < 1.0
for lab, sel in (("waves started < 1 us", early), ("waves started >= 1 us", ~early)):
    if sel.sum() == 0:
        continue
    d = np.diff(t[sel], axis=1) / 1e3
    print("%s: n=%d  lifetime %.2f us" % (lab, sel.sum(), life[sel].mean()))
    for k, nme in enumerate(names):
        print("    %-8s %6.2f us (p90 %6.2f)" % (nme, d[:, k].mean(), np.percentile(d[:, k], 90)))
slow = (np.diff(t, axis=1)[:, 2] / 1e3) > 3.5      # the propagation stage ran the conic chain
for lab, sel in (("waves whose propagation stage took > 3.5 us", slow), ("the others", ~slow)):
    if sel.sum() == 0:
        continue
    d = np.diff(t[sel], axis=1) / 1e3
    print("%s: n=%d  lifetime %.2f us (p90 %.2f)  start %.2f  end %.2f (p99 %.2f)" % (lab, sel.sum(), life[sel].mean(), np.percentile(life[sel], 90), st[sel].mean(),
                                                                                    en[sel].mean(), np.percentile(en[sel], 99)))
    print("    " + "  ".join("%s %.2f" % (nme, d[:, k].mean()) for k, nme in enumerate(names)))
# which branch of the out-of-line propagation the wavefront's lanes took (bit 1 hyperbolic, 2 near-parabolic band, 4 elliptic beyond the
# series) and their longest hyperbolic Newton run -- words 13 / 14 of the -DSSA_TRACE build
br, it = tr[:, 13] & 7, tr[:, 14]
kep = (t[:, 3] - t[:, 2]) / 1e3
if br.any():
    for code, nme in ((0, "no call"), (1, "hyperbolic only"), (2, "band only"), (3, "hyperbolic + band"), (4, "elliptic beyond the series"), (5, "4+1"), (6, "4+2"), (7, "4+2+1")):
        sel = br == code
        if sel.sum():
            print("  kepler stage by branch  %-28s n=%5d  mean %6.2f us  p90 %6.2f  max %6.2f   Newton iterations p50 %d max %d  lifetime mean %.2f max %.2f" % (
                nme, sel.sum(), kep[sel].mean(), np.percentile(kep[sel], 90), kep[sel].max(), np.median(it[sel]), it[sel].max(), life[sel].mean(), life[sel].max()))
slow = np.argsort(life)[-8:]
print("slowest waves (lifetime, per-stage us):")
for w_ in slow:
    print("   %.2f  " % life[w_], (np.diff(t[w_]) / 1e3).round(2), " branch bits %d  Newton iterations %d" % (br[w_], it[w_]))
hw = tr[:, 15] & 0xffffffff
xcc = (tr[:, 15] >> 32) & 0xf
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
wave = hw & 15
print("XCC histogram:", np.bincount(xcc, minlength=8))
key = xcc * 10000 + se * 1000 + cu * 10 + simd
u, c = np.unique(key, return_counts=True)
print("distinct (xcc,se,cu,simd): %d ; waves per SIMD min/mean/max: %d / %.2f / %d" % (len(u), c.min(), c.mean(), c.max()))
ucu, ccu = np.unique(xcc * 10000 + se * 1000 + cu * 10, return_counts=True)
print("distinct CUs: %d ; waves per CU min/mean/max: %d / %.2f / %d" % (len(ucu), ccu.min(), ccu.mean(), ccu.max()))
if os.environ.get("SIMD_MAP") == "1":
    # how the dispatcher spread the launch: tile -> (xcc, se, cu, simd), and what each SIMD had to run.  (tile = the XCD's contiguous block:
    # position i of XCD x is tile x * per + i, dispatched as workgroup 8 i + x)
    per = (nb + 7) // 8
    pos = np.arange(nb) % per if nb % 8 == 0 else np.arange(nb) - (np.arange(nb) // per) * per
    x0 = xcc == xcc[0]
    print("XCD of tile 0: positions 0..39 -> (se, cu, simd, slot):", [(int(se[i]), int(cu[i]), int(simd[i]), int(wave[i])) for i in np.nonzero(x0)[0][:40]])
    kst = np.diff(t, axis=1)[:, 2] / 1e3
    slow_w = kst > 3.5
    work = np.diff(t, axis=1)[:, 1:9].sum(axis=1) / 1e3
    ids, inv = np.unique(key, return_inverse=True)
    n_slow = np.bincount(inv, weights=slow_w.astype(float))
    n_all = np.bincount(inv)
    end_simd = np.zeros(len(ids)); np.maximum.at(end_simd, inv, en)
    kep_sum = np.bincount(inv, weights=kst)
    print("slow waves per SIMD (histogram 0..5):", np.bincount(n_slow.astype(int), minlength=6))
    for k in range(6):
        sel = n_slow.astype(int) == k
        if sel.sum():
            print("   SIMDs with %d slow waves: n=%4d  waves %.2f  last end mean %.2f  p90 %.2f  max %.2f   sum of kepler stages %.2f us" % (
                k, sel.sum(), n_all[sel].mean(), end_simd[sel].mean(), np.percentile(end_simd[sel], 90), end_simd[sel].max(), kep_sum[sel].mean()))
    print("correlation(last end of a SIMD, sum of its waves' kepler stages) = %.3f ; (last end, number of slow waves) = %.3f" % (
        np.corrcoef(end_simd, kep_sum)[0, 1], np.corrcoef(end_simd, n_slow)[0, 1]))
    print("SIMD last-end percentiles 0/25/50/75/90/99/100:", np.percentile(end_simd, [0, 25, 50, 75, 90, 99, 100]).round(2))
    # positions (within the XCD's block) of the waves of a few SIMDs: is the spread regular?
    for kk in ids[:6]:
        sel = key == kk
        print("   SIMD %d: positions %s  slow %s" % (kk, sorted(pos[sel].tolist()), slow_w[sel].astype(int).tolist()))
    np.save(os.environ.get("SIMD_MAP_OUT", "/tmp/simd_map.npy"), np.stack([np.arange(nb), xcc, se, cu, simd, wave, slow_w.astype(np.int64), (kst * 100).astype(np.int64), (en * 100).astype(np.int64), (st * 100).astype(np.int64)], axis=1))
# concurrency over time
grid = np.linspace(0, en.max(), 60)
conc = [(int(((st <= g) & (en > g)).sum())) for g in grid]
print("resident waves over time:", conc)

upd = np.nonzero(tr[:, 10])[0]
if len(upd):
    print("update wavefront(s): start, end [us] and the update's phases [us]: hx | z mean | residual + S, Pxz sums | inv + K | x, P | records")
    for w_ in upd[:4]:
        seq = [tr[w_, 5], tr[w_, 10], tr[w_, 11], tr[w_, 12], tr[w_, 13], tr[w_, 14], tr[w_, 6]]
        print("   tile %d  start %.2f end %.2f   predict stages %s   update %s" % (w_, st[w_], en[w_], (np.diff(t[w_, :6]) / 1e3).round(2),
              (np.diff(np.array(seq, dtype=np.float64)) * 10.0 / 1e3).round(2)))
