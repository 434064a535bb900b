"""All-cores CPU baseline: throughput of the OpenMP oracle build vs thread count on this host.

The first all-cores figure (256 threads = len(sched_getaffinity)) came out SLOWER than one core; this sweep
separates oversubscription of the container's CPU quota from a problem in the harness itself."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.argv = ['bench.py']
import bench
import oracle as orc
orc.build()
print("os.cpu_count()", os.cpu_count(), " affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "unreadable:", e.__class__.__name__)
m = 20000
pb = bench.build_problem(m, seed=0)
Wm, Wc, scale = orc.merwe_weights(1e-4, 2.0, -3)
P0 = np.tile(pb["P0"], (m, 1, 1))
for name, o, threads in [("serial", orc.Oracle(), [1])] + [("omp", orc.Oracle(omp=True), [1, 2, 4, 8, 16, 32, 64, 128, 256])]:
    oi = o.lla2ecef(pb["obs_lla"])
    for t in threads:
        got = o.lib.orc_omp_threads(t) if name == "omp" else 1
        xt, x, P, st = pb["x_true"], pb["x"], P0, np.zeros(m, dtype=np.int32)
        n, t0 = 0, time.perf_counter()
        while n < 3 or time.perf_counter() - t0 < 3.0:
            r = o.env_step(xt, x, P, st, 20.0, pb["Q"], pb["R"], Wm, Wc, scale, n % m, pb["trans"][(n + 1) % 480],
                           pb["obs_lla"], oi, -np.pi / 2, np.zeros(3))
            xt, x, P = r["x_true"], r["x"], r["P"]
            n += 1
        el = time.perf_counter() - t0
        print("%-6s threads asked %3d got %3d : %6.2f env-steps/s (%d steps, %.1f s)" % (name, t, got, n / el, n, el), flush=True)
