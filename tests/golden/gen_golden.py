"""Golden-vector generator (build container only; run with /opt/conda/bin/python3.9,
the one interpreter here that has pyerfa):

    cd /root/repo && /opt/conda/bin/python3.9 tests/golden/gen_golden.py

Every *output* below is produced by the reference's own source files, loaded
from /root/reference by tests/golden/_ref_loader.py:
    fx   = envs/farnocchia.py::fx_xyz_farnocchia  (+ rv2coe / delta_t_from_nu / nu_from_delta_t)
    hx   = envs/dynamics.py::hx_aer_erfa, hx_xyz
    mean = envs/dynamics.py::mean_z_uvw ; residual = residual_z_aer ; msqrt = robust_cholesky
    envs/transformations.py::lla2ecef, ecef2aer, aer2uvw, uvw2aer, gcrs2irts_matrix_b
The UKF algebra (filterpy: absent) is oracle/ukf_numpy.py driven by those
callbacks -> files whose name starts with "ukf_" / "episode_" are COMPOSITE
goldens (reference callbacks + restated filterpy), all others are pure
reference outputs.  The *.npz / *.npy files written here are committed; the
reference itself never leaves this container.
"""
import os
import sys
from datetime import datetime, timedelta
from itertools import permutations

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import _ref_loader as L  # noqa: E402
import ukf_numpy as U  # noqa: E402

far = L.load_farnocchia()
tr, dy = L.load_transformations_and_dynamics()
eop = L.load_eops()
K = 398600441800000.0
CAT = np.load(L.REF + "/envs/1.5_hour_viz_20000_of_20000_sample_orbits_seed_0.npy")
fx = far.fx_xyz_farnocchia


def out(name, **arrs):
    path = os.path.join(HERE, name)
    if name.endswith(".npy"):
        np.save(path, arrs["a"])
    else:
        np.savez_compressed(path, **arrs)
    print("wrote", name, {k: np.shape(v) for k, v in arrs.items()})


# ------------------------------------------------------------------ catalogue subset
def classify(cat):
    r, v = cat[:, :3], cat[:, 3:]
    h = np.cross(r, v)
    rn = np.linalg.norm(r, axis=1)
    e = ((np.sum(v * v, 1) - K / rn)[:, None] * r - np.sum(r * v, 1)[:, None] * v) / K
    ecc = np.linalg.norm(e, axis=1)
    inc = np.arccos(h[:, 2] / np.linalg.norm(h, axis=1))
    return ecc, inc


def pick_rows():
    ecc, inc = classify(CAT)
    circ, equa = ecc < 1e-8, np.abs(inc) < 1e-8
    rng = np.random.RandomState(20201004)
    groups = [np.where(circ & equa)[0], np.where(circ & ~equa)[0], np.where(~circ & equa)[0],
              np.where(~circ & ~equa & (ecc > 0.7))[0], np.where(~circ & ~equa & (ecc <= 0.7))[0]]
    want = [64, 64, 64, 64, 256]
    rows = []
    for g, w in zip(groups, want):
        w = min(w, len(g))
        rows.append(rng.choice(g, size=w, replace=False))
    rows = np.sort(np.concatenate(rows))
    print("catalogue census: circular", circ.sum(), "equatorial", equa.sum(), "both", (circ & equa).sum(),
          "max ecc", ecc.max(), "-> subset", len(rows))
    return rows


def gen_kepler():
    rows = pick_rows()
    x0 = CAT[rows]
    # sigma-point-like neighbours of special-branch rows: +-17 m / +-1.7e-2 m/s offsets
    rng = np.random.RandomState(7)
    special = x0[:128] + rng.normal(size=(128, 6)) * np.array([17.3, 17.3, 17.3, 1.7e-2, 1.7e-2, 1.7e-2])
    # filter-like states: 1e5 m / 1e2 m/s noise (what x_filter looks like after reset)
    noisy = x0[::4] + rng.normal(size=(len(x0[::4]), 6)) * np.array([1e5] * 3 + [1e2] * 3)
    x = np.vstack([x0, special, noisy])
    dts = np.array([20.0, 30.0, 150.0, 5400.0, 86400.0])
    y = np.empty((len(dts), len(x), 6))
    inter = np.empty((len(dts), len(x), 8))
    for a, dt in enumerate(dts):
        for j, xs in enumerate(x):
            y[a, j] = fx(xs, dt)
            p, ecc, inc, raan, argp, nu0 = far.rv2coe(K, xs[:3], xs[3:])
            q = p / (1 + ecc)
            t0 = far.delta_t_from_nu(nu0, ecc, K, q)
            nu = far.nu_from_delta_t(t0 + dt, ecc, K, q)
            inter[a, j] = [p, ecc, inc, raan, argp, nu0, t0, nu]
    out("kepler_golden.npz", rows=rows, x=x, dts=dts, y=y, inter=inter)
    out("catalogue_subset.npy", a=x0)
    # self-check values quoted in SURVEY.md appendix A
    ref = fx(np.array([34090858.3, 23944774.4, 6503066.82, -1983.78508, 2150.41744, 913.881611]), 30.0)
    assert np.allclose(ref, [3.40312632e7, 2.40092296e7, 6.53046768e6, -1.98922002e3, 2.14659156e3,
                             9.12841766e2], rtol=1e-8)


# ------------------------------------------------------------------------- geometry
OBSERVER = (38.828198, -77.305352, 20.0)


def times(t0, dt, n):
    return [t0 + timedelta(seconds=dt) * i for i in range(n)]


def gen_c2t():
    t0 = datetime(2020, 5, 4, 0, 0, 0)
    m20 = np.array(tr.gcrs2irts_matrix_b(times(t0, 20.0, 480), eop))
    out("c2t_2020-05-04_dt20_n480.npy", a=m20)
    m30 = np.array(tr.gcrs2irts_matrix_b(times(t0, 30.0, 2880), eop))
    out("c2t_2020-05-04_dt30_n2880.npy", a=m30)
    sofa_t = datetime(2007, 4, 5, 12, 0, 0)
    out("c2t_2007-04-05_1200.npy", a=np.array(tr.gcrs2irts_matrix_b(sofa_t, eop)))
    return m20


def gen_geometry(m20):
    rng = np.random.RandomState(11)
    obs_lla = np.array(OBSERVER) * [tr.deg2rad, tr.deg2rad, 1]
    obs_itrs = tr.lla2ecef(obs_lla)
    llas = np.array([[0, 0, 0], [0.5, 1.0, 100.0], [-1.2, -2.9, 8000.0], [np.pi / 2, 0.3, 10.0],
                     list(obs_lla)])
    ecefs = np.array([tr.lla2ecef(l) for l in llas])
    xs = CAT[rng.randint(0, len(CAT), size=256)]
    steps = np.array([0, 137, 479])
    z = np.empty((len(steps), len(xs), 3))
    for a, s in enumerate(steps):
        for j, x in enumerate(xs):
            z[a, j] = dy.hx_aer_erfa(x, m20[s], obs_lla, obs_itrs)
    # ecef2aer with other observers (tests.py Test 3 geometry + Test 9b equator observer)
    e2a_in, e2a_out = [], []
    for l, e in zip(llas, ecefs):
        for _ in range(16):
            sat = e + rng.normal(size=3) * 5e5 + e / np.linalg.norm(e) * 1e6
            e2a_in.append(np.concatenate([l, sat, e]))
            e2a_out.append(tr.ecef2aer(l, sat, e))
    # residual_z_aer: tests.py:197-206 cases
    az = np.radians([0, 0.001, 90.0, 180, 270.0, 359.99, 360])
    el = np.radians([-90.00, -89.99, -0.999, 0, 0.999, 89.99, 90.00])
    sr = [-1000.0001, -1, -0.0001, 0, 0.0001, 1, 1000.0001]
    ra, rb, rc = [], [], []
    for a3, e3, s3 in zip(permutations(az, 2), permutations(el, 2), permutations(sr, 2)):
        a0 = np.asarray([a3[0], e3[0], s3[0]])
        a1 = np.asarray([a3[1], e3[1], s3[1]])
        ra.append(a0), rb.append(a1), rc.append(dy.residual_z_aer(a0, a1))
    # mean_z_uvw: sigma-point style sets (Merwe weights, both alphas) + uniform weights incl. az wrap
    mz_sig, mz_w, mz_out = [], [], []
    for alpha in (1e-3, 1e-4):
        pts = U.MerweScaledSigmaPoints(6, alpha, 2.0, -3, sqrt_method=dy.robust_cholesky)
        for j in range(16):
            x = xs[j]
            P = np.diag([1e10] * 3 + [1e4] * 3) * (1.0 if j % 2 else 1e-4)
            sig = pts.sigma_points(x, P)
            sh = np.array([dy.hx_aer_erfa(s, m20[5], obs_lla, obs_itrs) for s in sig])
            mz_sig.append(sh), mz_w.append(pts.Wm), mz_out.append(dy.mean_z_uvw(sh, pts.Wm))
    for j in range(8):
        sh = np.column_stack([(rng.normal(size=13) * 0.01 + (0.0 if j < 4 else 3.0)) % (2 * np.pi),
                              rng.normal(size=13) * 0.01 + 0.5, 1e7 + rng.normal(size=13) * 1e3])
        w = np.repeat(1 / 13, 13)
        mz_sig.append(sh), mz_w.append(w), mz_out.append(dy.mean_z_uvw(sh, w))
    # aer2uvw / uvw2aer
    aer = np.column_stack([rng.uniform(0, 2 * np.pi, 64), rng.uniform(-1.5, 1.5, 64), rng.uniform(1e5, 5e7, 64)])
    uvw = np.array([tr.aer2uvw(a) for a in aer])
    back = np.array([tr.uvw2aer(u) for u in uvw])
    out("geometry_golden.npz", obs_lla=obs_lla, obs_itrs=obs_itrs, llas=llas, ecefs=ecefs,
        wgs84=np.array([tr.a, tr.f, tr.e]), hx_x=xs, hx_steps=steps, hx_z=z,
        e2a_in=np.array(e2a_in), e2a_out=np.array(e2a_out),
        res_a=np.array(ra), res_b=np.array(rb), res_c=np.array(rc),
        mz_sig=np.array(mz_sig), mz_w=np.array(mz_w), mz_out=np.array(mz_out),
        aer=aer, uvw=uvw, aer_back=back)


def gen_cholesky():
    rng = np.random.RandomState(3)
    A, Uo, ok = [], [], []

    def add(a):
        A.append(a)
        try:
            Uo.append(dy.robust_cholesky(a)), ok.append(1)
        except np.linalg.LinAlgError:
            Uo.append(np.full((6, 6), np.nan)), ok.append(0)

    for _ in range(16):
        B = rng.normal(size=(6, 6)) * np.array([1e5] * 3 + [1e2] * 3)
        add(3e-8 * (B @ B.T + np.diag([1e10] * 3 + [1e4] * 3)))
    add(3e-8 * np.diag([1e10] * 3 + [1e4] * 3))
    add(np.diag([4., 9, -1, 1, 1, 1]))  # SURVEY probe: ladder reaches 10^1
    for k in range(-7, 10):  # -10^k on one diagonal: every rung of the ladder
        d = np.array([4., 9, 1, 1, 1, 1])
        d[2] = -(10.0 ** k)
        add(np.diag(d))
    v = rng.normal(size=6)
    add(np.outer(v, v))  # rank one, semi-definite
    add(np.zeros((6, 6)))
    add(-1e12 * np.eye(6))  # beyond the ladder -> LinAlgError
    bad = np.eye(6)
    bad[1, 2] = np.nan
    add(bad)  # check_finite -> LinAlgError
    out("cholesky_golden.npz", A=np.array(A), U=np.array(Uo), ok=np.array(ok))


# ------------------------------------------------------------------- composite (UKF)
def make_filter(alpha, dt, hx, mean_z, residual_z, resample):
    pts = U.MerweScaledSigmaPoints(n=6, alpha=alpha, beta=2., kappa=3 - 6, sqrt_method=dy.robust_cholesky)
    return U.UnscentedKalmanFilter(dim_x=6, dim_z=3, dt=dt, fx=fx, hx=hx, points=pts, z_mean_fn=mean_z,
                                   residual_z=residual_z, sqrt_fn=dy.robust_cholesky,
                                   resample_after_predict=resample)


def gen_ukf_steps(m20):
    rng = np.random.RandomState(5)
    obs_lla = np.array(OBSERVER) * [tr.deg2rad, tr.deg2rad, 1]
    obs_itrs = tr.lla2ecef(obs_lla)
    dt = 20.0
    Q = U.Q_discrete_white_noise(dim=2, dt=dt, var=0.000025 ** 2, block_size=3, order_by_dim=False)
    R = np.diag([tr.arcsec2rad ** 2] * 2 + [1e3 ** 2])
    P0 = np.diag([1e5 ** 2] * 3 + [1e2 ** 2] * 3)
    z_sigma = np.array([1, 1, 1e3]) * np.array([tr.arcsec2rad, tr.arcsec2rad, 1])
    rows = rng.randint(0, len(CAT), size=64)
    x_true = CAT[rows]
    x0 = x_true + rng.normal(size=(64, 6)) * np.array([1e5] * 3 + [1e2] * 3)
    z_noise = rng.normal(size=(64, 3)) * z_sigma
    res = dict(x_true=x_true, x0=x0, P0=P0, Q=Q, R=R, z_noise=z_noise, M=m20[1], obs_lla=obs_lla,
               obs_itrs=obs_itrs, dt=dt)
    kw = dict(trans_matrix=m20[1], observer_itrs=obs_itrs, observer_lla=obs_lla, time=None)
    for alpha, tag in ((1e-3, "a3"), (1e-4, "a4")):
        for resample, rtag in ((False, ""), (True, "_rs")):
            xp, Pp, sf, xu, Pu, yy, SS, sh, xp2, Pp2 = ([] for _ in range(10))
            for j in range(64):
                f = make_filter(alpha, dt, dy.hx_aer_erfa, dy.mean_z_uvw, dy.residual_z_aer, resample)
                f.x, f.P, f.Q, f.R = x0[j].copy(), P0.copy(), Q.copy(), R.copy()
                f.predict()
                xp.append(f.x.copy()), Pp.append(f.P.copy()), sf.append(f.sigmas_f.copy())
                xt1 = fx(x_true[j], dt)
                z = dy.hx_aer_erfa(xt1, **kw) + z_noise[j]
                f.update(z, **kw)
                xu.append(f.x.copy()), Pu.append(f.P.copy()), yy.append(f.y.copy()), SS.append(f.S.copy())
                sh.append(f.sigmas_h.copy())
                f.predict()  # second predict from the tight posterior (small-P regime)
                xp2.append(f.x.copy()), Pp2.append(f.P.copy())
            for k, v in (("xp", xp), ("Pp", Pp), ("sf", sf), ("xu", xu), ("Pu", Pu), ("y", yy), ("S", SS),
                         ("sh", sh), ("xp2", xp2), ("Pp2", Pp2)):
                res["%s_%s%s" % (k, tag, rtag)] = np.array(v)
    # xyz observation model (tests.py Test 14 configuration)
    Rx = np.diag([5e2 ** 2] * 3)
    zx = rng.normal(size=(64, 3)) * 5e2
    xu, Pu, yy, SS = [], [], [], []
    for j in range(64):
        f = make_filter(1e-4, dt, dy.hx_xyz, dy.mean_xyz, np.subtract, False)
        f.x, f.P, f.Q, f.R = x0[j].copy(), P0.copy(), Q.copy(), Rx.copy()
        f.predict()
        z = dy.hx_xyz(fx(x_true[j], dt)) + zx[j]
        f.update(z)
        xu.append(f.x.copy()), Pu.append(f.P.copy()), yy.append(f.y.copy()), SS.append(f.S.copy())
    res.update(xyz_R=Rx, xyz_znoise=zx, xyz_xu=np.array(xu), xyz_Pu=np.array(Pu), xyz_y=np.array(yy),
               xyz_S=np.array(SS))
    out("ukf_step_golden.npz", **res)


def gen_test67():
    """tests.py:118-188 (Test 6 / Test 7) scenario with the restated UKF + reference fx/hx_xyz."""
    dt = 30.0
    x = np.array([34090858.3, 23944774.4, 6503066.82, -1983.785080, 2150.41744, 913.881611])
    P = np.eye(6) * np.array([1000, 1000, 1000, 1, 1, 1])
    pts = U.MerweScaledSigmaPoints(6, 0.001, 2.0, 3 - 6, sqrt_method=__import__("scipy.linalg").linalg.cholesky)
    Q = U.Q_discrete_white_noise(dim=2, dt=dt, var=0.000001 ** 2, block_size=3, order_by_dim=False)
    res = {}
    for resample, tag in ((False, ""), (True, "_rs")):
        ukf = U.UnscentedKalmanFilter(6, 3, dt, dy.hx_xyz, fx, pts, resample_after_predict=resample)
        ukf.x, ukf.P, ukf.Q = x.copy(), P.copy(), Q.copy()
        xt = x.copy()
        for _ in range(50):
            ukf.predict(dt)
            xt = fx(xt, dt)
        res["x50" + tag], res["P50" + tag], res["xt50"] = ukf.x.copy(), ukf.P.copy(), xt.copy()
        e6 = [np.sqrt(np.sum((ukf.x - xt)[:3] ** 2)), np.sqrt(np.sum((ukf.x - xt)[3:] ** 2))]
        # tests.py:162 passes R as a 1-D array: P += R broadcasts it to 125*ones((3,3))
        ukf.update(z=xt[:3], R=np.array([125., 125., 125.]))
        e7 = [np.sqrt(np.sum((ukf.x - xt)[:3] ** 2)), np.sqrt(np.sum((ukf.x - xt)[3:] ** 2))]
        res["x50u" + tag], res["P50u" + tag] = ukf.x.copy(), ukf.P.copy()
        res["err" + tag] = np.array(e6 + e7)
        print("Test6/7 scenario%s: pos %.4g m vel %.4g m/s | after update pos %.4g vel %.4g" % (tag, *e6, *e7))
        assert e6[0] < 1.0 and e6[1] < 1e-4
    out("test67_golden.npz", x0=x, P0=P, Q=Q, dt=dt, **res)


def run_episode(m, n, dt, alpha, obs_limit_deg, obs_type, seed, c2t, resample=False, every=40):
    """restated ssa_tasker_simple_2.py:193-367 with reference callbacks; round-robin actions."""
    obs_lla = np.array(OBSERVER) * [tr.deg2rad, tr.deg2rad, 1]
    obs_itrs = tr.lla2ecef(obs_lla)
    obs_limit = np.radians(obs_limit_deg)
    rs = np.random.RandomState(seed)
    x_sigma = np.array([1e5] * 3 + [1e2] * 3)
    if obs_type == 'aer':
        z_sigma = np.array([1, 1, 1e3]) * np.array([tr.arcsec2rad, tr.arcsec2rad, 1])
        R = np.diag([tr.arcsec2rad ** 2] * 2 + [1e3 ** 2])
        hx, mz, rz = dy.hx_aer_erfa, dy.mean_z_uvw, dy.residual_z_aer
    else:
        z_sigma = np.array([5e2] * 3)
        R = np.diag([5e2 ** 2] * 3)
        hx, mz, rz = dy.hx_xyz, dy.mean_xyz, np.subtract
    P0 = np.diag(x_sigma ** 2)
    Q = U.Q_discrete_white_noise(dim=2, dt=dt, var=0.000025 ** 2, block_size=3, order_by_dim=False)
    rows = np.empty(m, dtype=int)
    x_noise = np.empty((m, 6))
    for j in range(m):  # draw order of reset() :206-209
        rows[j] = rs.randint(low=0, high=CAT.shape[0])
        x_noise[j] = rs.normal(size=6) * x_sigma
    z_noise = np.empty((n, m, 3))
    for i in range(n):  # :219-221
        for j in range(m):
            z_noise[i, j] = rs.normal(size=3) * z_sigma
    x_true = np.empty((n, m, 6))
    x_f = np.empty((n, m, 6))
    P_f = np.empty((n, m, 6, 6))
    x_true[0] = CAT[rows]
    x_f[0] = x_true[0] + x_noise
    P_f[0] = P0
    filters = []
    for j in range(m):
        f = make_filter(alpha, dt, hx, mz, rz, resample)
        f.x, f.P, f.Q, f.R = x_f[0, j].copy(), P0.copy(), Q.copy(), R.copy()
        filters.append(f)
    rewards = np.zeros(n)
    obs_taken = np.zeros(n, dtype=bool)
    ys = np.full((n, 3), np.nan)
    Ss = np.full((n, 3, 3), np.nan)
    zt = np.full((n, 3), np.nan)
    for i in range(1, n):
        a = (i - 1) % m
        for j in range(m):
            x_true[i, j] = fx(x_true[i - 1, j], dt)
        for j in range(m):
            filters[j].predict()
            x_f[i, j], P_f[i, j] = filters[j].x, filters[j].P
        kw = dict(trans_matrix=c2t[i], observer_itrs=obs_itrs, observer_lla=obs_lla, time=None)
        zt[i] = hx(x_true[i, a], **kw)
        x_itrs = c2t[i] @ x_true[i, a, :3]
        if tr.ecef2aer(obs_lla, x_itrs, obs_itrs)[1] >= obs_limit:
            filters[a].update(zt[i] + z_noise[i, a], **kw)
            ys[i], Ss[i], obs_taken[i] = filters[a].y, filters[a].S, True
            x_f[i, a], P_f[i, a] = filters[a].x, filters[a].P
        dpos = np.sqrt(np.sum((x_f[i, :, :3] - x_true[i, :, :3]) ** 2, axis=1))
        rewards[i] = np.mean(((dpos < 1e4) * 1 + (dpos < 1e7) * 1)) / 2  # results.py:432
    keep = np.arange(0, n, every)
    keep = np.unique(np.concatenate([keep, [1, 2, n - 1]]))
    return dict(rows=rows, x_true0=x_true[0], x0=x_f[0], P0=P0, Q=Q, R=R, z_noise=z_noise, keep=keep,
                x_true=x_true[keep], x_filter=x_f[keep], P_filter=P_f[keep], rewards=rewards,
                obs_taken=obs_taken, y=ys, S=Ss, z_true=zt, obs_lla=obs_lla, obs_itrs=obs_itrs,
                params=np.array([m, n, dt, alpha, obs_limit_deg, 0 if obs_type == 'aer' else 1, seed,
                                 int(resample)], dtype=float))


def gen_episodes(m20):
    ep = run_episode(20, 480, 20.0, 1e-4, -90, 'aer', 0, m20)
    out("episode_aer_m20_n480.npz", **ep)
    ep = run_episode(10, 120, 20.0, 1e-4, 15, 'aer', 1, m20, every=20)
    print("  visibility-limited episode: obs taken", ep["obs_taken"].sum(), "of", 119)
    out("episode_aer_vis15_m10_n120.npz", **ep)
    ep = run_episode(10, 60, 20.0, 1e-4, -90, 'xyz', 2, m20, every=20)
    out("episode_xyz_m10_n60.npz", **ep)


if __name__ == "__main__":
    gen_kepler()
    m20 = gen_c2t()
    gen_geometry(m20)
    gen_cholesky()
    gen_ukf_steps(m20)
    gen_test67()
    gen_episodes(m20)
