"""GPU parity of the fused env step (ssa_env_step_f64) against the CPU oracle, per step
from identical inputs (parity is defined on identical inputs).

Tolerance scheme (DESIGN.md "Numerical conditioning"): the reference algorithm at
alpha = 1e-4 carries Merwe weights of -2e8 / +1.67e7, so ITS OWN fp64 result is only
defined up to an amplified rounding floor.  Three results are compared for every batch:
    gpu   the HIP kernel (fp64, centred summation)
    f64   the oracle in fp64, reference order of operations   (= "the reference's value")
    ld    the oracle in 80-bit long double                      (= the exact value)
Asserted: (1) gpu is within the north_star tolerance (1e-6 means / 1e-5 covariances) of f64
for every object whose reference value is itself defined to that level; (2) for ALL objects
gpu is as close to the exact value as the reference arithmetic is (factor 3 on the batch
statistics), i.e. the kernel adds no error of its own.
"""
import numpy as np
import pytest

import oracle as orc
from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    import ssa_gym_amd
    from ssa_gym_amd import _lib, device, host, engine
    ssa_gym_amd.build()
    _lib.load()
    assert torch.cuda.is_available()

    class H:
        pass
    h = H()
    h.torch, h.lib, h.dev, h.host, h.engine = torch, _lib, device, host, engine
    return h


C2T = None


def c2t():
    global C2T
    if C2T is None:
        C2T = golden("c2t_2020-05-04_dt20_n480.npy")
    return C2T


def make_batch(m, seed, tight_fraction=0.0):
    rs = np.random.RandomState(seed)
    cat = golden("catalogue_subset.npy")
    g = golden("ukf_step_golden.npz")
    xt = cat[rs.randint(0, len(cat), m)]
    x = xt + rs.normal(size=(m, 6)) * np.array([1e5] * 3 + [1e2] * 3)
    P = np.tile(g["P0"], (m, 1, 1))
    if tight_fraction > 0:
        # posterior-like covariances (after an az/el/range update): sample from the golden posteriors
        k = rs.uniform(size=m) < tight_fraction
        idx = rs.randint(0, 64, m)
        P[k] = 0.5 * (g["Pu_a3"][idx[k]] + np.swapaxes(g["Pu_a3"][idx[k]], 1, 2))
        x[k] = xt[k] + rs.normal(size=(k.sum(), 6)) * np.array([30.0] * 3 + [0.05] * 3)
    return xt, x, P, g


def run_gpu(hip, xt, x, P, g, action, tix, alpha, obs_type='aer', propagator='fg', resample=False,
            obs_limit=-np.pi / 2, E=1, status=None, R=None, z_noise=None, zn_strides=None):
    m = x.shape[0] // E
    R = g["R"] if R is None else R
    consts = hip.host.make_consts(g["Q"], R, alpha, 2.0, -3, 20.0, obs_limit, g["obs_lla"], obs_type=obs_type,
                                  propagator=propagator, resample=resample)
    rs = np.random.RandomState(99)
    if z_noise is None:
        z_noise = rs.normal(size=(E, 480, m, 3)) * np.array([4.8e-6, 4.8e-6, 1e3])
    if zn_strides is None:
        eng = hip.engine.HotPathEngine(consts, m, E, c2t(), z_noise, history=2)
    else:   # (env, time, object) strides of a compact noise table
        eng = hip.engine.HotPathEngine(consts, m, E, c2t(), z_noise, history=2, zn_stride_env=zn_strides[0],
                                       zn_stride_time=zn_strides[1], zn_stride_obj=zn_strides[2])
    eng.load_state(0, xt, x, P)
    if status is not None:
        eng.status.copy_(hip.torch.as_tensor(status))
    eng.set_actions(action)
    eng.launch_step(0, 1, tix)
    hip.torch.cuda.synchronize()
    out = dict(x_true=eng.x_true[1].cpu().numpy(), x=eng.x_filter[1].cpu().numpy(),
               P=eng.P_filter[1].cpu().numpy(), obs=eng.obs[1].cpu().numpy(),
               metrics=eng.metrics[1].cpu().numpy(), status=eng.status.cpu().numpy(),
               upd=eng.upd[1].cpu().numpy(), stats=eng.stats[1].cpu().numpy(), z_noise=z_noise)
    return out


def run_oracle(o, xt, x, P, g, action, tix, alpha, obs_type=0, centred=False, resample=False,
               obs_limit=-np.pi / 2, status=None, R=None, z_noise3=None):
    Wm, Wc, scale = orc.merwe_weights(alpha, 2.0, -3)
    st = np.zeros(x.shape[0], dtype=np.int32) if status is None else status.copy()
    R = g["R"] if R is None else R
    r = o.env_step(xt, x, P, st, 20.0, g["Q"], R, Wm, Wc, scale, action, c2t()[tix], g["obs_lla"], g["obs_itrs"],
                   obs_limit, z_noise3, obs_type=obs_type, centred=centred, resample=resample)
    r["status"] = st
    return r


def errs(a, b):
    """per-object relative error of position / velocity blocks and sd-normalised covariance error"""
    ep = np.linalg.norm((a["x"] - b["x"])[:, :3], axis=1) / np.linalg.norm(b["x"][:, :3], axis=1)
    ev = np.linalg.norm((a["x"] - b["x"])[:, 3:], axis=1) / np.linalg.norm(b["x"][:, 3:], axis=1)
    sd = np.sqrt(np.abs(np.einsum('jii->ji', b["P"])))
    eP = np.max(np.abs(a["P"] - b["P"]) / (sd[:, :, None] * sd[:, None, :]), axis=(1, 2))
    return ep, ev, eP


def assert_states_close(a, b, tol, what=""):
    """norm-wise relative comparison of [n,6] states: position block and velocity block."""
    ep = np.linalg.norm((a - b)[:, :3], axis=1) / np.linalg.norm(b[:, :3], axis=1)
    ev = np.linalg.norm((a - b)[:, 3:], axis=1) / np.linalg.norm(b[:, 3:], axis=1)
    assert ep.max() < tol and ev.max() < tol, (what, ep.max(), ev.max())


def inclination(x):
    h = np.cross(x[:, :3], x[:, 3:])
    return np.arctan2(np.hypot(h[:, 0], h[:, 1]), h[:, 2])


def reference_floor(oracle, ld, xt, x, P, g, action, tix, alpha, n_pert=8, **kw):
    """per-object fp64 floor of the REFERENCE arithmetic: the largest distance from the exact value (80-bit witness
    `ld`) over the unperturbed run and n_pert runs whose filter states are moved by at most one ulp per component.
    One run is a single realisation of the rounding noise (an object can sit close to the exact value by luck); the
    maximum over nine tells how well the reference value of that object is DEFINED."""
    rs = np.random.RandomState(5)
    fl = np.zeros((3, x.shape[0]))
    for k in range(n_pert + 1):
        xp = x if k == 0 else x * (1 + 2.220446049250313e-16 * rs.randint(-1, 2, size=x.shape))
        f = run_oracle(oracle, xt, xp, P, g, action, tix, alpha, **kw)
        fl = np.maximum(fl, np.array(errs(f, ld)))
    return fl


def check_parity(gpu, f64, ld, exact_bound, tol_x=1e-6, tol_P=1e-5, well_frac=0.5, min_well=0.25, floor=None, tag="",
                 jump_check=None):
    """the two-sided criterion of the module docstring; returns the error arrays.  `floor` (reference_floor) replaces
    the single-realisation distance f64-vs-exact in the definition of "well defined"; `min_well` is the measured
    fraction of well-defined objects of the batch (oracle-only quantity), asserted so that the strict comparison
    cannot silently shrink."""
    ep, ev, eP = errs(gpu, f64)
    rp, rv, rP = errs(f64, ld)      # the reference arithmetic's own distance from the exact value
    gp, gv, gP = errs(gpu, ld)
    # (1) objects whose reference value is itself defined 2x tighter than the tolerance
    fp, fv, fP = (rp, rv, rP) if floor is None else floor
    well = (fp < well_frac * tol_x) & (fv < well_frac * tol_x) & (fP < well_frac * tol_P)
    print("[parity%s] strictly compared (reference value defined to %.1f x tol): %.4f of %d objects; max |gpu-ref| there: "
          "pos %.2e vel %.2e cov %.2e; all objects gpu-vs-exact max pos %.2e, reference-vs-exact max pos %.2e"
          % (tag, well_frac, well.mean(), len(well), ep[well].max(), ev[well].max(), eP[well].max(), gp.max(), rp.max()))
    assert well.mean() >= min_well, (well.mean(), min_well)
    bad = well & ((ep >= tol_x) | (ev >= tol_x) | (eP >= tol_P))
    if jump_check is None:
        assert not bad.any(), (np.where(bad)[0][:8], ep[well].max(), ev[well].max(), eP[well].max())
    else:
        # SSA_PROP_ELEMENTS evaluates the reference's own formulas, discontinuities included (rv2coe's branch thresholds,
        # the 2 pi wraps, Newton's iteration count): on a few objects a one-ulp change of an input moves the REFERENCE
        # value by more than the tolerance, rarely enough that nine realisations do not reveal it.  Such an object may
        # differ between the kernel and the reference -- but only if the reference demonstrably differs from itself there:
        # jump_check(j) = the largest movement of the reference value of object j over 200 one-ulp perturbations.
        assert bad.sum() <= max(1, int(0.005 * len(well))), bad.sum()
        for j in np.where(bad)[0]:
            jp, jv, jP = jump_check(int(j))
            print("[parity%s] object %d: |gpu-ref| pos %.2e vel %.2e cov %.2e; the reference value itself moves by pos %.2e "
                  "vel %.2e cov %.2e under one-ulp input perturbations" % (tag, j, ep[j], ev[j], eP[j], jp, jv, jP))
            # (factor 4: the device libm's acos / atan2 / tan are 1-2 ulp functions, glibc's are < 1 ulp, and the one-ulp
            # input perturbation probes only part of the rounding noise of the chain)
            assert jp >= 0.25 * ep[j] and jv >= 0.25 * ev[j] and jP >= 0.25 * eP[j], (j, ep[j], ev[j], eP[j], jp, jv, jP)
    # (2) every object: the kernel is at least as accurate as the reference arithmetic
    for g_, r_ in ((gp, rp), (gv, rv), (gP, rP)):
        assert np.median(g_) <= 3 * np.median(r_) + 1e-13
        assert g_.max() <= 3 * r_.max() + 1e-12
    if exact_bound:   # SSA_PROP_FG: within the north_star tolerance of the EXACT value, all objects
        assert gp.max() < tol_x and gv.max() < tol_x and gP.max() < tol_P, (gp.max(), gv.max(), gP.max())
    return ep, rp, gp


@pytest.mark.parametrize("propagator", ["fg", "elements", "hybrid"])
@pytest.mark.parametrize("alpha", [1e-3, 1e-4])
def test_predict_parity_2000_objects(hip, oracle, oracle_ld, alpha, propagator):
    """BASELINE config 2 size (2 000 objects), predict only (action -1)."""
    m = 2000
    xt, x, P, g = make_batch(m, seed=1)
    gpu = run_gpu(hip, xt, x, P, g, [-1], 1, alpha, propagator=propagator)
    f64 = run_oracle(oracle, xt, x, P, g, -1, 1, alpha, z_noise3=np.zeros(3))
    ld = run_oracle(oracle_ld, xt, x, P, g, -1, 1, alpha, centred=True, z_noise3=np.zeros(3))
    assert np.all(gpu["status"] == 0) and np.all(f64["status"] == 0)
    # truth propagation: plain Kepler parity
    assert_states_close(gpu["x_true"], f64["x_true"], 1e-9, "truth")
    # Fraction of objects whose reference value is defined to 0.5 x tolerance (oracle-only numbers, measured for this
    # batch): alpha = 1e-3: 1.0; alpha = 1e-4: 0.8865 by the single run, 0.8365 by the nine-realisation floor.
    # SSA_PROP_FG is compared on the former set (and against the exact value on ALL objects); SSA_PROP_ELEMENTS
    # carries the same, independent, rounding sensitivity as the reference (its distance to the reference value is
    # the sum of both), so it is compared on the objects whose reference value is defined to 0.5 x tolerance under
    # one-ulp input perturbations -- the excluded 16 % are the objects where the reference's own fp64 result moves
    # by more than half the tolerance when an input changes by one ulp.
    floor = reference_floor(oracle, ld, xt, x, P, g, -1, 1, alpha, z_noise3=np.zeros(3)) if propagator == "elements" else None
    # ('hybrid' runs the series solver on these strong-elliptic states -- judged as SSA_PROP_FG -- with the covariance in the
    # reference's own arithmetic: its cancellation noise is the reference value's too, hence no exact bound on the covariance)
    min_well = {("fg", 1e-3): 0.99, ("fg", 1e-4): 0.88, ("elements", 1e-3): 0.99, ("elements", 1e-4): 0.83,
                ("hybrid", 1e-3): 0.99, ("hybrid", 1e-4): 0.88}[(propagator, alpha)]
    def jump(j):
        rs = np.random.RandomState(1000 + j)
        sl = slice(j, j + 1)
        base = {k: f64[k][sl] for k in ("x", "P")}
        out = np.zeros(3)
        for _ in range(200):
            xp = x[sl] * (1 + 2.220446049250313e-16 * rs.randint(-1, 2, size=(1, 6)))
            f = run_oracle(oracle, xt[sl], xp, P[sl], g, -1, 1, alpha, z_noise3=np.zeros(3))
            out = np.maximum(out, np.array(errs(f, base))[:, 0])
        return out
    ep, rp, gp = check_parity(gpu, f64, ld, exact_bound=(propagator == "fg"), well_frac=0.5, min_well=min_well, floor=floor,
                              tag=" %s alpha=%g" % (propagator, alpha), jump_check=jump if propagator == "elements" else None)
    # obs / metrics are consistent with the state the kernel wrote
    assert np.array_equal(gpu["obs"][:, :6], gpu["x"])
    assert np.array_equal(gpu["obs"][:, 6:], np.einsum('jii->ji', gpu["P"]))
    o_obs, o_met = oracle.observe(gpu["x_true"], gpu["x"], gpu["P"])
    np.testing.assert_allclose(gpu["metrics"][0], o_met, rtol=1e-14)
    # covariance symmetric by construction, P prior = P + growth
    assert np.array_equal(gpu["P"], np.swapaxes(gpu["P"], 1, 2))


@pytest.mark.parametrize("obs_type", ["aer", "xyz"])
@pytest.mark.parametrize("resample", [False, True])
def test_update_parity_every_object(hip, oracle, oracle_ld, obs_type, resample):
    """one update per env: run E = 48 single-object-action envs in ONE launch (vector-env path)
    so that 48 different objects get their update; compare each with the oracle."""
    m, E, alpha = 16, 48, 1e-4
    xt, x, P, g = make_batch(m * E, seed=3)
    R = g["R"] if obs_type == 'aer' else g["xyz_R"]
    actions = [(7 * e) % m for e in range(E)]
    gpu = run_gpu(hip, xt, x, P, g, actions, 5, alpha, obs_type=obs_type, resample=resample, E=E, R=R)
    ot = 0 if obs_type == 'aer' else 1
    e_x, r_x, e_P, r_P, e_y, e_S = [], [], [], [], [], []
    for e in range(E):
        sl = slice(e * m, (e + 1) * m)
        a = actions[e]
        zn = gpu["z_noise"][e, 5, a]
        f64 = run_oracle(oracle, xt[sl], x[sl], P[sl], g, a, 5, alpha, obs_type=ot, resample=resample, R=R, z_noise3=zn)
        ld = run_oracle(oracle_ld, xt[sl], x[sl], P[sl], g, a, 5, alpha, obs_type=ot, centred=True, resample=resample,
                        R=R, z_noise3=zn)
        rec = gpu["upd"][e]
        assert rec[hip.lib.UPD_OBS_TAKEN] == 1.0 and f64["obs_taken"]
        assert rec[hip.lib.UPD_ACTION] == a and rec[hip.lib.UPD_VISIBLE] == 1.0
        assert np.linalg.norm(rec[1:4] - f64["z_true"]) <= 1e-12 * np.linalg.norm(f64["z_true"]) + 1e-12
        sub = {k: gpu[k][sl] for k in ("x", "P")}
        ep, ev, eP = errs({k: v[a:a + 1] for k, v in sub.items()}, {k: f64[k][a:a + 1] for k in ("x", "P")})
        gp, gv, gP = errs({k: v[a:a + 1] for k, v in sub.items()}, {k: ld[k][a:a + 1] for k in ("x", "P")})
        rp, rv, rP = errs({k: f64[k][a:a + 1] for k in ("x", "P")}, {k: ld[k][a:a + 1] for k in ("x", "P")})
        e_x.append(gp[0]), r_x.append(rp[0]), e_P.append(gP[0]), r_P.append(rP[0])
        assert gp[0] < 1e-5 and gv[0] < 1e-5   # sanity bound; the statistical criterion is below
        # innovation: compare with the exact value in units of its standard deviation
        Sd = np.sqrt(np.diag(ld["S"]))
        e_y.append(np.max(np.abs(rec[4:7] - ld["y"]) / Sd))
        e_S.append(np.max(np.abs(rec[7:16].reshape(3, 3) - ld["S"]) / np.outer(Sd, Sd)))
        if not resample:   # propagated sigma points: identical inputs -> identical measurements
            np.testing.assert_allclose(rec[16:55].reshape(13, 3), f64["sigmas_h"], rtol=1e-12, atol=1e-7)
        else:              # redrawn from the prior mean, which carries the fp64 floor of the UT
            np.testing.assert_allclose(rec[16:55].reshape(13, 3), ld["sigmas_h"], rtol=1e-5, atol=1e-5)
        # every non-selected object of this env is predict-only: within tolerance of the exact value
        others = np.arange(m) != a
        assert_states_close(sub["x"][others], ld["x"][others], 1e-6, "predict-only objects")
    # the first update collapses P from ~1e10 to ~1e3 (P - K S K^T): the posterior covariance is
    # conditioned to ~1e-3 relative for ANY fp64 evaluation; the kernel must not be worse than
    # the reference arithmetic (batch statistics, factor 3)
    assert np.median(e_P) <= 3 * np.median(r_P) + 1e-9
    assert max(e_P) <= 3 * max(r_P) + 1e-9
    assert np.median(e_x) <= 3 * np.median(r_x) + 1e-12
    assert max(e_y) < 1e-3 and max(e_S) < 1e-2


def test_second_step_from_tight_covariances(hip, oracle, oracle_ld):
    """predict from posterior-like (tight, ~50 m) covariances: the sigma spread is ~1 cm around
    |r| ~ 4e7 m, the worst case for the fp64 floor."""
    m, alpha = 512, 1e-4
    xt, x, P, g = make_batch(m, seed=4, tight_fraction=1.0)
    gpu = run_gpu(hip, xt, x, P, g, [-1], 2, alpha)
    f64 = run_oracle(oracle, xt, x, P, g, -1, 2, alpha, z_noise3=np.zeros(3))
    ld = run_oracle(oracle_ld, xt, x, P, g, -1, 2, alpha, centred=True, z_noise3=np.zeros(3))
    ok = (f64["status"] == 0) & (gpu["status"] == 0)
    assert ok.mean() > 0.9
    # objects whose (n+lambda) P is within rounding of the positive-definiteness boundary get a
    # different robust_cholesky jitter rung from different arithmetic (worth +33 (m/s)^2 per 1e-6):
    # compare the objects on which the three implementations pick the same rung
    _, _, scale = orc.merwe_weights(alpha, 2.0, -3)
    rung_gpu = hip.dev.robust_cholesky(hip.dev.as_dev(scale * P))[1].cpu().numpy()
    rung_f64 = np.array([oracle.robust_cholesky(scale * Pj)[1] for Pj in P])
    rung_ld = np.array([oracle_ld.robust_cholesky(scale * Pj)[1] for Pj in P])
    same = ok & (rung_gpu == rung_f64) & (rung_gpu == rung_ld)
    assert same.mean() > 0.9 and (rung_gpu == rung_f64).mean() > 0.97
    # filter states that have converged onto a (near-)equatorial orbit (GEO rows: inc < 1e-3 rad) hit
    # the reference's inc = acos(h_z/|h|) (farnocchia.py:274): ~1e-3 m of error per sigma point, times
    # Wi = 1.67e7.  There the reference-order fp64 value is off by up to 1e-2 RELATIVE (hundreds of km)
    # and even the 80-bit witness by ~1e-5; the kernel (no acos) must simply beat both.
    equatorial = inclination(x) < 1e-3
    assert 0.1 < equatorial.mean() < 0.5
    rp_all, _, _ = errs({k: f64[k] for k in ("x", "P")}, {k: ld[k] for k in ("x", "P")})
    gp_all, _, _ = errs({k: gpu[k] for k in ("x", "P")}, {k: ld[k] for k in ("x", "P")})
    eq = same & equatorial
    assert rp_all[eq].max() > 1e-4                      # the reference arithmetic really is broken there
    assert gp_all[eq].max() < 1e-4 and np.median(gp_all[eq]) < 1e-6
    sel = same & ~equatorial
    sub = lambda d: {k: d[k][sel] for k in ("x", "P")}   # noqa: E731
    rp, rv, rP = errs(sub(f64), sub(ld))
    gp, gv, gP = errs(sub(gpu), sub(ld))
    assert gp.max() < 1e-6 and gv.max() < 1e-6, (gp.max(), gv.max())
    assert np.median(gp) <= 3 * np.median(rp) + 1e-13
    assert np.median(gP) <= 3 * np.median(rP) + 1e-9
    # jitter-ladder decisions (robust_cholesky) agree with the oracle wherever the input is not
    # within rounding of the positive-definiteness boundary
    assert np.mean(gpu["status"] == f64["status"]) > 0.99


def test_fused_ladder_picks_the_reference_rung_for_every_rung(hip, oracle_ld):
    """The fused kernel tries all sixteen rungs of an object's robust_cholesky ladder side by side (one lane per rung) and takes the
    lowest that factorises: the answer must be the sequential ladder's (dynamics.py:402-417) for EVERY rung -1, 0..15 and beyond
    (LinAlgError) -- on well-separated cases here; test_ladder_first_success_on_the_non_monotone_case pins the case where success is
    not monotone in the jitter, which a search that skips rungs gets wrong.  Indefinite
    priors built so that (n + lambda) P needs a given rung (smallest eigenvalue -0.3 x that rung's jitter), mixed in wavefronts of four
    with healthy objects and with each other; checked through the sequential ladder of the single-operator entry point, the failure
    status, and the propagated covariance against the oracle run with that rung (a neighbouring rung changes P by 10 x the jitter)."""
    alpha = 1e-4
    _, _, scale = orc.merwe_weights(alpha, 2.0, -3)
    rs = np.random.RandomState(12)
    want = np.array([-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16] * 4)
    rs.shuffle(want)
    m = len(want)
    xt, x, P, g = make_batch(m, seed=12)
    for j, r in enumerate(want):
        if r < 0:
            continue
        jit = 10.0 ** (min(r, 16) - 6)
        lam_min = -0.3 * jit / scale if r < 16 else -30.0 * 1e9 / scale      # beyond the last rung (1e9 I): LinAlgError
        Qm, _ = np.linalg.qr(rs.normal(size=(6, 6)))
        lam = abs(lam_min) * np.array([-1.0, 3.0, 10.0, 100.0, 300.0, 1e3])    # (every rung's jitter is 3e-2 of the largest variance)
        P[j] = (Qm * lam) @ Qm.T
        P[j] = 0.5 * (P[j] + P[j].T)
    rung_seq = hip.dev.robust_cholesky(hip.dev.as_dev(scale * P))[1].cpu().numpy()      # the sequential ladder on the device
    rung_orc = np.array([oracle_ld.robust_cholesky(scale * Pj)[1] if r < 16 else 16 for Pj, r in zip(P, want)])
    assert np.array_equal(rung_seq, want) and np.array_equal(rung_orc, want), (rung_seq, want)
    gpu = run_gpu(hip, xt, x, P, g, [-1], 2, alpha)
    ld = run_oracle(oracle_ld, xt, x, P, g, -1, 2, alpha, centred=True, z_noise3=np.zeros(3))
    assert np.array_equal(gpu["status"] != 0, want == 16) and np.array_equal(ld["status"] != 0, want == 16)
    ok = want < 16
    sub = lambda d: {k: d[k][ok] for k in ("x", "P")}   # noqa: E731
    gp, gv, gP = errs(sub(gpu), sub(ld))
    assert gp.max() < 1e-6 and gv.max() < 1e-6 and gP.max() < 1e-4, (gp.max(), gv.max(), gP.max())   # (tight covariances: fp64 floor 1e-5)
    # sensitivity of this check: with 9 x the rung's jitter added to (n + lambda) P -- the plain factorisation then sees nearly the NEXT
    # rung's matrix -- the oracle's covariance moves by far more than the tolerance above, for every rung
    P10 = P.copy()
    for j, r in enumerate(want):
        if 0 <= r < 15:
            P10[j] = P[j] + (9.0 * 10.0 ** (r - 6) / scale) * np.eye(6)
    ld10 = run_oracle(oracle_ld, xt, x, P10, g, -1, 2, alpha, centred=True, z_noise3=np.zeros(3))
    sel = (want >= 0) & (want < 15)
    _, _, dP = errs({k: ld10[k][sel] for k in ("x", "P")}, {k: ld[k][sel] for k in ("x", "P")})
    assert dP.min() > 2e-3, dP.min()       # (20 x the tolerance: a rung off by one cannot hide)


def test_ladder_on_the_ill_conditioned_tile(hip, oracle):
    """robust_cholesky's answer is the FIRST rung that factorises (dynamics.py:406-414), of the matrix the reference factorises:
    round((n + lambda) P) + jitter, two roundings (:410).  tests/golden/ladder_illconditioned_tile.npz is the wavefront (four prior
    covariances) at which round 3's two-pass ladder build parted from the sequential one over an episode: (n + lambda) P of its row 1 has
    condition 1e20.  ssa_ladder_probe_f64 runs the fused kernels' ladder on it and reports, per rung, whether it factorises in that
    arithmetic:
      * the fused rung IS the lowest rung that factorises -- for this tile (rung 1: the plain attempt and rung 0 fail), for 4 096 one-ulp
        perturbations of it (each a different realisation of the rounding noise) and whatever the neighbours in the wavefront are;
        success is monotone in the jitter in every one of them (round 3's explanation of the two-pass failure does not hold: the two-pass
        rule applied to these masks picks the same rungs);
      * THE PIN: the factor the ladder leaves is bit for bit the plain factor of the matrix round(scale * P) + jitter * I formed on the
        host in two roundings -- not of the fused fma(scale, P, jitter), which differs from it in a diagonal entry of this matrix and whose
        factor differs in its last rows by ~1e-8 relative (computed here: that is how two builds of the same source parted);
      * the sequential register ladder (ssa_robust_cholesky6_f64: IEEE square roots and divisions, another rounding realisation) and the
        oracle pick the same rung on this tile, and rungs within the noise band on the perturbed ensemble."""
    from fractions import Fraction
    g = golden("ladder_illconditioned_tile.npz")
    _, _, scale = orc.merwe_weights(1e-4, 2.0, -3)
    P4 = g["P_tile"]
    j = int(g["obj"]) % 4
    rung, mask, U = hip.dev.ladder_probe(hip.dev.as_dev(P4), scale)
    rung, mask, U = rung.cpu().numpy(), mask.cpu().numpy(), U.cpu().numpy()
    assert rung.tolist() == [-1, 1, -1, -1] and j == 1
    assert (mask[j] >> 16) & 1 == 0 and (mask[j] & 0xFFFF) == 0xFFFE           # plain and rung 0 fail, rungs 1 .. 15 factorise: monotone
    # ---- the pin: two roundings
    jit = 10.0 ** (int(rung[j]) - 6)
    A2 = scale * P4[j] + jit * np.eye(6)                                       # numpy: round(scale * p), then + jitter -- the reference's arithmetic
    fused = np.array([[float(Fraction(scale) * Fraction(float(P4[j][r, c])) + (Fraction(jit) if r == c else 0)) for c in range(6)] for r in range(6)])
    assert (np.diag(A2) != np.diag(fused)).any() and np.array_equal(A2 - np.diag(np.diag(A2)), fused - np.diag(np.diag(fused)))
    tile = np.stack([np.eye(6), A2, fused, np.eye(6)])
    r1, m1, U1 = hip.dev.ladder_probe(hip.dev.as_dev(tile), 1.0)              # scale 1: the plain attempt factorises the matrix as given
    r1, U1 = r1.cpu().numpy(), U1.cpu().numpy()
    assert r1.tolist() == [-1, -1, -1, -1]
    assert np.array_equal(U1[1].view(np.int64), U[j].view(np.int64))           # the ladder's factor == the factor of the two-rounding matrix
    rel = np.abs(U1[2] - U1[1])[3:, 3:].max() / np.abs(U1[1])[3:, 3:].max()
    print("[ladder] tile of object %d at step %d: rung %s; diagonal entries where fma(scale, p, jit) != round(scale p) + jit: %d of 6; the fused matrix's "
          "factor differs from the reference-arithmetic one by %.1e relative in its last rows" % (int(g["obj"]), int(g["step"]), rung.tolist(),
                                                                                               int((np.diag(A2) != np.diag(fused)).sum()), rel))
    assert 1e-12 < rel < 1e-4                                                  # (one rounding, amplified by the conditioning)
    M0 = scale * P4[j] + jit * np.eye(6)
    assert np.abs(U[j].T @ U[j] - M0).max() <= 4e-16 * np.abs(M0).max() and np.allclose(np.tril(U[j], -1), 0.0)
    # ---- first success over an ensemble of rounding realisations; monotone masks; every position in the wavefront
    rs = np.random.RandomState(271)
    n_pert = 4096
    P = np.tile(P4[j], (n_pert, 1, 1))
    ulp = 2.220446049250313e-16
    for k in range(1, n_pert):          # symmetric one-ulp perturbations; entry 0 is the captured matrix itself
        e = rs.randint(-1, 2, size=(6, 6))
        e = np.triu(e) + np.triu(e, 1).T
        P[k] = P[k] * (1.0 + ulp * e)
    rg, mk, _ = hip.dev.ladder_probe(hip.dev.as_dev(P), scale)
    rg, mk = rg.cpu().numpy(), mk.cpu().numpy()
    m16 = mk & 0xFFFF
    first = np.array([16 if m == 0 else int(m & -m).bit_length() - 1 for m in m16])
    want = np.where((mk >> 16) & 1 == 1, -1, first)
    assert np.array_equal(rg, want), (np.where(rg != want)[0][:8], rg[:8], want[:8])
    nonmono = np.array([first[i] < 16 and int(m16[i]) != (0xFFFF >> first[i]) << first[i] for i in range(n_pert)])

    def two_pass(m):
        for grp in range(4):
            if (m >> (4 * grp + 3)) & 1:
                return next(i for i in range(4 * grp, 4 * grp + 4) if (m >> i) & 1)
        return 16
    tp = np.array([two_pass(int(m)) for m in m16])
    print("[ladder] %d one-ulp realisations: rungs %s; non-monotone masks %d; the two-pass rule differs on %d"
          % (n_pert, dict(zip(*np.unique(rg, return_counts=True))), nonmono.sum(), (tp != first).sum()))
    assert not nonmono.any() and np.array_equal(tp, first)
    rung_seq = hip.dev.robust_cholesky(hip.dev.as_dev(scale * P))[1].cpu().numpy()
    rung_orc = np.array([oracle.robust_cholesky(scale * Pk)[1] for Pk in P[:256]])
    assert rung_seq[0] == rg[0] == rung_orc[0] == 1
    for name, other in (("sequential register ladder", rung_seq), ("oracle", rung_orc)):
        r = rg[:len(other)]
        print("[ladder] %s: same rung as the fused kernel on %.3f of the realisations" % (name, np.mean(other == r)))
        assert np.abs(other - r).max() <= 4          # all decisions inside the band where the jitter is below the pivots' rounding noise
    Q = np.tile(P4[j], (8, 1, 1))
    Q[1::2] = np.diag([1e10] * 3 + [1e4] * 3)          # healthy neighbours in between: every row position
    r2 = hip.dev.ladder_probe(hip.dev.as_dev(Q), scale)[0].cpu().numpy()
    assert np.all(r2[0::2] == 1) and np.all(r2[1::2] == -1)


def test_argmax_sigma_pos_on_every_one_launch_path(hip):
    """np.argmax(sigma_pos) (the 'shaped' reward, ssa_tasker_simple_2.py:339-352) without the post kernel: the step kernel's per-tile
    slots reduced by whoever folds the statistics -- fold kernel, deferred fold inside the next launch, the last wavefront
    (SSA_LAUNCH_FOLD_INSIDE), the rollout's fold -- against numpy on the metrics and against the three-launch exact path; ties (first
    index wins) and NaN (first NaN wins) included; ragged last tile; several envs of whole tiles; the multi-tile instance."""
    torch, lib = hip.torch, hip.lib
    for m, E in ((4, 1), (5, 1), (1003, 1), (20000, 1), (20484, 1), (24, 3), (20000, 2)):
        xt, x, P, g = make_batch(m * E, seed=31 + m)
        consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"], obs_type='aer', propagator='fg')
        rs = np.random.RandomState(1)
        zn = rs.normal(size=(E, 480, 1, 3)) * np.array([4.8e-6, 4.8e-6, 1e3])
        def engine():
            eng = hip.engine.HotPathEngine(consts, m, E, c2t(), zn, history=4, zn_stride_env=480 * 3, zn_stride_time=3, zn_stride_obj=0)
            eng.load_state(0, xt, x, P)
            eng.set_actions([1 % m] * E)
            return eng
        ref = engine()
        for i in (1, 2, 3):
            ref.launch_step(i - 1, i, i)                                   # exact path: step + post + final
        torch.cuda.synchronize()
        want = ref.stats[1:4].cpu().numpy()
        sp = ref.metrics[1:4, :, 2].cpu().numpy()
        assert np.array_equal(want[:, :, lib.STAT_ARGMAX_SPOS], np.argmax(sp, axis=2))
        for mode in ("fold_kernel", "deferred", "inside", "rollout"):
            eng = engine()
            if mode == "rollout":
                acts = torch.full((3, E), 1 % m, dtype=torch.int32, device="cuda")
                eng.launch_rollout(0, 1, acts, argmax_spos=True)
            else:
                for i in (1, 2, 3):
                    eng.launch_step(i - 1, i, i, fast_stats=True, defer_fold=(mode == "deferred"), fold_inside=(mode == "inside"), argmax_spos=True)
                eng.flush_stats()
            torch.cuda.synchronize()
            got = eng.stats[1:4].cpu().numpy()
            for col in (lib.STAT_ARGMAX_SPOS, lib.STAT_MAX_SPOS, lib.STAT_MAX_DPOS, lib.STAT_CNT_LT_1E4, lib.STAT_CNT_LT_1E7, lib.STAT_N_FAILED):
                assert np.array_equal(got[:, :, col], want[:, :, col], equal_nan=True), (m, E, mode, col, got[:, :, col], want[:, :, col])
    # ties and NaN: equal covariances give equal sigma_pos -> the first index; a NaN covariance ranks above everything
    m = 64
    xt, x, P, g = make_batch(m, seed=5)
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"], obs_type='aer', propagator='fg')
    zn = np.zeros((1, 480, m, 3))
    for poison in (None, 37, 11):
        eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), zn, history=2)
        xx, PP = x.copy(), P.copy()
        xx[:] = xx[0]; xt2 = np.tile(xt[0], (m, 1))                         # identical objects: identical sigma_pos
        eng.load_state(0, xt2, xx, PP)
        st = np.zeros(m, dtype=np.int32)
        if poison is not None:
            st[poison] = lib.ST_PREDICT_NAN                                   # a failed filter passes through: give it a NaN covariance
            eng.P_filter[0, poison] = float("nan")
        eng.status.copy_(torch.as_tensor(st))
        eng.set_actions([-1])
        eng.launch_step(0, 1, 1, fast_stats=True, argmax_spos=True)
        torch.cuda.synchronize()
        s = eng.stats[1, 0].cpu().numpy()
        spv = eng.metrics[1, 0, 2].cpu().numpy()
        assert int(s[lib.STAT_ARGMAX_SPOS]) == int(np.argmax(spv)) == (0 if poison is None else poison)
        assert np.array_equal(s[lib.STAT_MAX_SPOS], np.max(spv), equal_nan=True)
    # envs that are not whole tiles: the library says so (the caller takes the three-launch path)
    xt, x, P, g = make_batch(14, seed=3)
    eng = hip.engine.HotPathEngine(consts, 7, 2, c2t(), np.zeros((2, 480, 7, 3)), history=2)
    assert not eng.supports_argmax
    with pytest.raises(lib.SsaHipError):
        eng.launch_step(0, 1, 1, fast_stats=True, argmax_spos=True)


def test_visibility_gate_and_no_update_paths(hip, oracle):
    m, alpha = 64, 1e-4
    xt, x, P, g = make_batch(m, seed=6)
    lim = np.radians(15.0)
    z = oracle.hx_aer(oracle.propagate(xt, 20.0), c2t()[7], g["obs_lla"], g["obs_itrs"])
    vis = z[:, 1] >= lim
    assert vis.any() and (~vis).any()
    for a in (int(np.where(vis)[0][0]), int(np.where(~vis)[0][0])):
        gpu = run_gpu(hip, xt, x, P, g, [a], 7, alpha, obs_limit=lim)
        rec = gpu["upd"][0]
        assert rec[hip.lib.UPD_VISIBLE] == float(vis[a]) and rec[hip.lib.UPD_OBS_TAKEN] == float(vis[a])
        np.testing.assert_allclose(rec[1:4], z[a], rtol=1e-12)   # z_true recorded either way (:298)
        ld = run_oracle(orc.Oracle(True), xt, x, P, g, a, 7, alpha, obs_limit=lim, centred=True,
                        z_noise3=gpu["z_noise"][0, 7, a])
        assert ld["obs_taken"] == bool(vis[a])
        assert_states_close(gpu["x"], ld["x"], 1e-6, "visibility gate")
    gpu = run_gpu(hip, xt, x, P, g, [-1], 7, alpha)
    assert gpu["upd"][0, hip.lib.UPD_ACTION] == -1.0 and gpu["upd"][0, hip.lib.UPD_OBS_TAKEN] == 0.0


def test_failure_sentinels_and_skip(hip, oracle):
    """F1: NaN state -> 'predict returned nan'; indefinite P beyond the jitter ladder ->
    LinAlgError; an already failed filter is passed through untouched (:272, 369-382)."""
    m, alpha = 12, 1e-4
    xt, x, P, g = make_batch(m, seed=8)
    x[2, 1] = np.nan
    P[5] = -1e18 * np.eye(6)
    P[7, 0, 0] = np.inf
    status = np.zeros(m, dtype=np.int32)
    status[9] = 3
    x[9] = hip.host.X_FAILED
    P[9] = hip.host.P_FAILED
    gpu = run_gpu(hip, xt, x, P, g, [2], 1, alpha, status=status)
    f64 = run_oracle(oracle, xt, x, P, g, 2, 1, alpha, status=status, z_noise3=np.zeros(3))
    assert np.array_equal(gpu["status"], f64["status"])
    assert gpu["status"][2] == 1 and gpu["status"][5] == 2 and gpu["status"][7] == 2 and gpu["status"][9] == 3
    for j in (2, 5, 7, 9):
        assert np.array_equal(gpu["x"][j], hip.host.X_FAILED)
        assert np.array_equal(gpu["P"][j], hip.host.P_FAILED)
    ok = gpu["status"] == 0
    ld = run_oracle(orc.Oracle(True), xt, x, P, g, 2, 1, alpha, status=status, centred=True, z_noise3=np.zeros(3))
    assert_states_close(gpu["x"][ok], ld["x"][ok], 1e-6, "healthy objects")
    assert gpu["upd"][0, hip.lib.UPD_OBS_TAKEN] == 0.0         # action pointed at a failed filter
    # ... which the reference skips entirely (:293): no z_true, the record says "no update attempted"
    assert gpu["upd"][0, hip.lib.UPD_ACTION] == -1.0
    gpu9 = run_gpu(hip, xt, x, P, g, [9], 1, alpha, status=status)           # a filter that had failed in an EARLIER step
    assert gpu9["upd"][0, hip.lib.UPD_ACTION] == -1.0 and gpu9["upd"][0, hip.lib.UPD_OBS_TAKEN] == 0.0
    assert gpu["stats"][0, hip.lib.STAT_N_FAILED] == 4
    assert gpu["stats"][0, hip.lib.STAT_MAX_DPOS] > 1e19       # sentinel dominates -> 'jones' done


@pytest.mark.parametrize("m", [1, 3, 4, 5, 63, 257])
def test_ragged_sizes(hip, oracle, m):
    xt, x, P, g = make_batch(m, seed=10 + m)
    gpu = run_gpu(hip, xt, x, P, g, [m - 1], 3, 1e-3)
    f64 = run_oracle(oracle, xt, x, P, g, m - 1, 3, 1e-3, z_noise3=gpu["z_noise"][0, 3, m - 1])
    ld = run_oracle(orc.Oracle(True), xt, x, P, g, m - 1, 3, 1e-3, centred=True, z_noise3=gpu["z_noise"][0, 3, m - 1])
    assert_states_close(gpu["x"], ld["x"], 1e-7, "ragged")
    assert_states_close(gpu["x_true"], f64["x_true"], 1e-9, "ragged truth")
    assert gpu["upd"][0, 0] == 1.0


@pytest.mark.parametrize("propagator", ["hybrid", "fg"])
def test_full_size_20000_properties(hip, propagator):
    """BASELINE config 3 size: properties that need no oracle (it would take minutes):
    truth energy conserved, P symmetric positive definite, mean inside the sigma cloud,
    growth of trace(P) positive for predict-only objects, and run-to-run determinism.  The env default (hybrid) and fg."""
    m = 20000
    xt, x, P, g = make_batch(m, seed=12)
    a = run_gpu(hip, xt, x, P, g, [123], 1, 1e-4, propagator=propagator)
    b = run_gpu(hip, xt, x, P, g, [123], 1, 1e-4, propagator=propagator)
    for k in ("x", "P", "x_true", "obs", "metrics"):
        assert np.array_equal(a[k], b[k]), k                    # bitwise reproducible
    assert np.all(a["status"] == 0)
    w = np.linalg.eigvalsh(a["P"])
    assert w.min() > 0
    tr0, tr1 = np.trace(P, axis1=1, axis2=2), np.trace(a["P"], axis1=1, axis2=2)
    others = np.arange(m) != 123
    assert np.all(tr1[others] > tr0[others]) and tr1[123] < 1e-3 * tr0[123]
    mu = 398600441800000.0

    def energy(s):
        return 0.5 * np.sum(s[:, 3:] ** 2, 1) - mu / np.linalg.norm(s[:, :3], axis=1)
    np.testing.assert_allclose(energy(a["x_true"]), energy(xt), rtol=1e-13)
    # predicted mean stays within a fraction of sigma of the propagated previous mean
    assert a["stats"][0, hip.lib.STAT_N_FAILED] == 0


@pytest.mark.parametrize("propagator", ["fg", "elements", "hybrid"])
def test_diverged_filter_states_all_conic_branches(hip, oracle, oracle_ld, propagator):
    """filters that have left the strong-elliptic regime (what a predict-only UKF at alpha=1e-4 does
    after ~250 steps): hyperbolic, near-parabolic and high-eccentricity states inside the fused step.
    SSA_PROP_FG handles them in the step kernel itself (one universal-variable solver for every conic),
    SSA_PROP_ELEMENTS through the out-of-line complete restatement of farnocchia(); both must agree with the oracle's
    farnocchia() branches wherever the oracle itself produces a finite prior."""
    from ssa_gym_amd.catalogue import coe2rv_host
    rs = np.random.RandomState(23)
    m = 600
    ecc = np.concatenate([rs.uniform(0.95, 0.9899, m // 4), rs.uniform(0.9901, 1.0099, m // 4),
                          rs.uniform(1.0101, 1.05, m // 4), rs.uniform(1.05, 2.5, m // 4)])
    rp = rs.uniform(6.8e6, 3e7, m)
    nu_max = np.where(ecc > 1, 0.7 * np.arccos(-1 / np.maximum(ecc, 1.000001)), 2.5)
    x = coe2rv_host(rp * (1 + ecc), ecc, rs.uniform(0.2, 2.9, m), rs.uniform(0, 6.28, m), rs.uniform(0, 6.28, m),
                    rs.uniform(-1, 1, m) * nu_max)
    xt, _, P, g = make_batch(m, seed=24)
    gpu = run_gpu(hip, xt, x, P, g, [-1], 4, 1e-4, propagator=propagator)
    f64 = run_oracle(oracle, xt, x, P, g, -1, 4, 1e-4, z_noise3=np.zeros(3))
    ld = run_oracle(oracle_ld, xt, x, P, g, -1, 4, 1e-4, centred=True, z_noise3=np.zeros(3))
    ok = (ld["status"] == 0) & (f64["status"] == 0)
    assert ok.mean() > 0.9
    if propagator in ("elements", "hybrid"):      # same branch structure as the reference: same failures
        assert np.mean(gpu["status"] == f64["status"]) > 0.98
    both = ok & (gpu["status"] == 0)
    assert both.mean() > 0.9
    sub = lambda d: {k: d[k][both] for k in ("x", "P")}   # noqa: E731
    gp, gv, gP = errs(sub(gpu), sub(ld))
    rp_, rv_, rP_ = errs(sub(f64), sub(ld))
    # against the exact value: as good as the reference arithmetic (batch statistics), and in absolute terms
    assert np.median(gp) <= 3 * np.median(rp_) + 1e-12
    assert np.quantile(gp, 0.99) <= 3 * np.quantile(rp_, 0.99) + 1e-9
    assert np.median(gp) < 1e-6
    assert_states_close(gpu["x_true"], ld["x_true"], 1e-9, "truth")


def test_fused_aer_payload_matches_operator(hip, oracle):
    """the post / final kernels can write the (az, el, range, trace P) block and the statistics straight
    into a caller buffer (the sharded all-gather payload): identical to the stand-alone O4 / O3 operators."""
    m = 1000
    xt, x, P, g = make_batch(m, seed=31)
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
    eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), np.zeros((1, 480, m, 3)), history=2)
    eng.load_state(0, xt, x, P)
    eng.set_actions([17])
    send = hip.torch.full((4 * m + 8,), -7.0, dtype=hip.torch.float64, device="cuda")
    eng.launch_step(0, 1, 9, aer_out=send[:4 * m].data_ptr(), stats_out=send[4 * m:].data_ptr())
    hip.torch.cuda.synchronize()
    M = eng.trans[9].reshape(3, 3).contiguous()
    ref = hip.dev.aer_obs(eng.x_filter[1], eng.P_filter[1], M, consts).reshape(-1)
    assert hip.torch.equal(send[:4 * m], ref)
    st = hip.dev.reward_stats(eng.metrics[1], eng.status, m, 1)[0]
    assert hip.torch.equal(send[4 * m:4 * m + 6], st[:6])
    o = oracle.aer_obs(eng.x_filter[1].cpu().numpy(), eng.P_filter[1].cpu().numpy(), c2t()[9], g["obs_lla"], g["obs_itrs"])
    np.testing.assert_allclose(send[:4 * m].cpu().numpy(), o, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("payload", [False, True])
@pytest.mark.parametrize("E,m", [(1, 2003), (3, 50), (2, 4)])
def test_two_launch_statistics_path_matches_exact_path(hip, E, m, payload):
    """ssa_step_params.stat_shards: max delta_pos / trinary counts / failures accumulated by the common-path
    kernel with sharded atomics must equal the post+final kernels' values (arg-max is documented as absent),
    and the filter state must be identical -- including envs that straddle a wavefront (m % 4 != 0), a NaN
    metric and failed filters."""
    xt, x, P, g = make_batch(E * m, seed=41)
    x[1, 0] = np.nan                               # -> predict NaN failure in env 0
    P[E * m - 1] = -1e18 * np.eye(6)               # -> LinAlgError in the last env
    xt[2, 0] = np.nan                              # NaN truth -> NaN delta_pos propagates into max (np.max semantics)
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
    outs = []
    for fast in (False, True):
        eng = hip.engine.HotPathEngine(consts, m, E, c2t(), np.zeros((E, 480, m, 3)), history=2)
        eng.load_state(0, xt, x, P)
        aer = hip.torch.zeros((E * m, 4), dtype=hip.torch.float64, device="cuda")   # O4 payload: the fold then rides in the post kernel
        ap = aer.data_ptr() if payload else 0
        eng.set_actions([(3 * e + 1) % m for e in range(E)])
        eng.launch_step(0, 1, 2, fast_stats=fast, aer_out=ap)
        eng.set_actions([(3 * e + 2) % m for e in range(E)])
        eng.launch_step(1, 0, 3, fast_stats=fast, aer_out=ap)   # second step: the shards were cleared by the fold
        hip.torch.cuda.synchronize()
        outs.append((eng.stats[0].cpu().numpy(), eng.x_filter[0].cpu().numpy(), eng.P_filter[0].cpu().numpy(),
                     eng.status.cpu().numpy(), eng.metrics[0].cpu().numpy(), aer.cpu().numpy()))
    (s0, x0, P0, st0, m0, a0), (s1, x1, P1, st1, m1, a1) = outs
    assert np.array_equal(st0, st1) and np.array_equal(x0, x1, equal_nan=True) and np.array_equal(P0, P1, equal_nan=True)
    assert np.array_equal(a0, a1) and (not payload or np.abs(a0).max() > 0)
    L = hip.lib
    for e in range(E):
        for k in (L.STAT_MAX_DPOS, L.STAT_CNT_LT_1E4, L.STAT_CNT_LT_1E7, L.STAT_N_FAILED):
            assert np.array_equal(s0[e, k], s1[e, k], equal_nan=True), (e, k, s0[e], s1[e])
        assert s1[e, L.STAT_ARGMAX_SPOS] == -1.0 and np.isnan(s1[e, L.STAT_MAX_SPOS])
        assert s0[e, L.STAT_ARGMAX_SPOS] == np.nanargmax(np.where(np.isnan(m0[e, 2]), np.inf, m0[e, 2]))
    assert np.isnan(s0[0, L.STAT_MAX_DPOS]) and s0[0, L.STAT_N_FAILED] >= 1 and s0[E - 1, L.STAT_N_FAILED] >= 1


@pytest.mark.parametrize("m", [3, 203, 511, 513, 20000, 41003])
def test_statistics_folded_inside_the_step_kernel(hip, m):
    """SSA_LAUNCH_FOLD_INSIDE: the step kernel's last wavefront folds the statistics (per-shard tile counters, one more counter
    over the shards) -- ONE launch per step with the statistics complete when it ends.  Same values as the fold-kernel path over
    consecutive steps (the counters and shards reset themselves), for fewer tiles than shards, ragged tiles, exactly / just over
    128 tiles, the whole 20 000-object launch and the multi-tile instance; NaN delta_pos and failed filters included."""
    xt, x, P, g = make_batch(m, seed=43)
    if m > 3:
        x[1, 0] = np.nan
        xt[2, 0] = np.nan
        P[m - 1] = -1e18 * np.eye(6)
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
    outs = []
    for inside in (False, True):
        eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), np.zeros((1, 480, m, 3)), history=2)
        eng.load_state(0, xt, x, P)
        stats = []
        for k in range(4):
            eng.launch_step(k % 2, (k + 1) % 2, k + 1, action=(7 * k + 1) % m, fast_stats=True, fold_inside=inside)
            hip.torch.cuda.synchronize()
            stats.append(eng.stats[(k + 1) % 2, 0].cpu().numpy().copy())
        outs.append((np.array(stats), eng.x_filter[0].cpu().numpy(), eng.status.cpu().numpy(), eng._shard_sets.cpu().numpy()))
    (s0, x0, st0, sh0), (s1, x1, st1, sh1) = outs
    assert np.array_equal(s0, s1, equal_nan=True), (s0, s1)
    assert np.array_equal(x0, x1, equal_nan=True) and np.array_equal(st0, st1)
    assert not sh1.any()                     # sums and counters are back at zero after every launch
    if m > 3:
        assert np.isnan(s1[0, hip.lib.STAT_MAX_DPOS]) and s1[-1, hip.lib.STAT_N_FAILED] >= 2


@pytest.mark.parametrize("fast", [False, True])
def test_trace_only_payload_is_column_3_of_the_aer_block(hip, fast):
    """ssa_step_params.aer_cols = 1 (the per-object covariance-trace observation of the sharded 160 000-object configuration):
    the payload is [E*m] trace P -- bit for bit the fourth column of the four-column block, from the epilogue of the step
    kernel (fast) and from the post kernel alike; any other column count is refused."""
    E, m = 2, 1003
    xt, x, P, g = make_batch(E * m, seed=43)
    P[7] = np.nan * np.eye(6)                      # a failed filter: its covariance becomes the sentinel diagonal (trace 3e20)
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
    res = {}
    for cols in (4, 1):
        eng = hip.engine.HotPathEngine(consts, m, E, c2t(), np.zeros((E, 480, m, 3)), history=2)
        eng.load_state(0, xt, x, P)
        eng.set_actions([5, 11])
        out = hip.torch.full((E * m * cols,), -7.0, dtype=hip.torch.float64, device="cuda")
        eng.launch_step(0, 1, 4, fast_stats=fast, aer_out=out.data_ptr(), aer_cols=cols)
        hip.torch.cuda.synchronize()
        res[cols] = out.cpu().numpy()
    assert np.array_equal(res[1], res[4].reshape(-1, 4)[:, 3]) and res[1][7] > 1e20 and (res[1] > 0).all()
    with pytest.raises(hip.lib.SsaHipError):
        eng.launch_step(1, 0, 5, fast_stats=fast, aer_out=out.data_ptr(), aer_cols=3)


@pytest.mark.parametrize("E,m,fast", [(1, 61443, False), (1, 61443, True), (8, 7001, True), (3, 20001, False),
                                      (9, 1, True), (7, 2, False), (5, 3, True)])   # (tiny envs: several selected objects -- several updates -- in ONE wavefront)
def test_multi_tile_wavefronts_equal_single_tile_results(hip, E, m, fast):
    """Above 20 480 objects a wavefront advances several tiles (grid-stride, next tile's loads in flight).
    An object's result must not depend on the batch it travels in: every env of a large batch is compared
    BITWISE with the same env run alone where one wavefront owns exactly one tile (<= 20 480 objects),
    statistics included; ragged sizes put the last tile and the env boundaries inside a wavefront's tile."""
    L = hip.lib
    xt, x, P, g = make_batch(E * m, seed=77)
    x[5, 1] = np.nan                                    # a failure, so statistics are not all-zero
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
    acts = [(977 * e + 13) % m for e in range(E)]
    n_time = c2t().shape[0]

    def run(lo, n_env, n_obj, actions):
        zn = hip.torch.zeros((n_time, n_obj, 3), dtype=hip.torch.float64, device="cuda")   # shared by the envs (stride 0)
        eng = hip.engine.HotPathEngine(consts, n_obj, n_env, c2t(), zn, history=2, zn_stride_env=0)
        sl = slice(lo, lo + n_env * n_obj)
        eng.load_state(0, xt[sl], x[sl], P[sl])
        eng.set_actions(actions)
        eng.launch_step(0, 1, 2, fast_stats=fast)
        hip.torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (eng.x_true[1], eng.x_filter[1], eng.P_filter[1], eng.obs[1], eng.metrics[1],
                                          eng.status, eng.stats[1])]
    big = run(0, E, m, acts)
    chunk = 20000 if m > 20000 else m                   # one tile per wavefront below 20 480 objects
    for e in range(E):
        if m <= 20480:
            ref = run(e * m, 1, m, [acts[e]])
            for k in (0, 1, 2, 3, 5):
                assert np.array_equal(big[k].reshape((E, m) + big[k].shape[1:])[e] if k != 5 else big[5].reshape(E, m)[e],
                                      ref[k].reshape((m,) + ref[k].shape[1:]) if k != 5 else ref[5].reshape(m), equal_nan=True), (e, k)
            assert np.array_equal(big[4][e], ref[4][0], equal_nan=True)
            for s in (L.STAT_MAX_DPOS, L.STAT_CNT_LT_1E4, L.STAT_CNT_LT_1E7, L.STAT_N_FAILED):
                assert np.array_equal(big[6][e, s], ref[6][0, s], equal_nan=True), (e, s)
        else:
            # pieces of the env (the selected object's piece gets the action, the others none)
            for lo in range(0, m, chunk):
                n = min(chunk, m - lo)
                a = acts[e] - lo if lo <= acts[e] < lo + n else -1
                ref = run(e * m + lo, 1, n, [a])
                for k in (0, 1, 2, 3):
                    assert np.array_equal(big[k].reshape((E, m) + big[k].shape[1:])[e, lo:lo + n], ref[k].reshape((n,) + ref[k].shape[1:]),
                                          equal_nan=True), (e, lo, k)
                assert np.array_equal(big[5].reshape(E, m)[e, lo:lo + n], ref[5].reshape(n))
                assert np.array_equal(big[4][e][:, lo:lo + n], ref[4][0], equal_nan=True)
    mt = big[4]                                          # statistics of the large batch against numpy on its metrics
    for e in range(E):
        assert np.array_equal(big[6][e, L.STAT_MAX_DPOS], np.max(mt[e, 0]), equal_nan=True)
        assert big[6][e, L.STAT_CNT_LT_1E4] == np.sum(mt[e, 0] < 1e4) and big[6][e, L.STAT_CNT_LT_1E7] == np.sum(mt[e, 0] < 1e7)
        assert big[6][e, L.STAT_N_FAILED] == np.sum(big[5].reshape(E, m)[e] != 0)


def test_engine_rejects_short_noise_table(hip):
    """operand extents are checked on the host before anything is launched (a short z_noise would be read out of bounds)."""
    xt, x, P, g = make_batch(8, seed=3)
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
    with pytest.raises(hip.lib.SsaHipError):
        hip.engine.HotPathEngine(consts, 4, 2, c2t(), np.zeros((1, 4, 4, 3)), history=2)


@pytest.mark.parametrize("E,m", [(1, 2003), (3, 50), (2, 30001)])
def test_deferred_statistics_fold_equals_immediate_fold(hip, E, m):
    """One launch per step: with defer_fold the statistics of step k are folded by extra wavefronts riding in
    step k+1's launch (flush_stats() folds the last; an immediate step after a deferred one folds it first).
    Every step's statistics and the whole filter state must equal the fold-kernel path's."""
    xt, x, P, g = make_batch(E * m, seed=91)
    x[7, 2] = np.nan
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
    n_time = c2t().shape[0]
    K = 6
    outs = []
    for mode in ("immediate", "deferred", "mixed"):
        zn = hip.torch.zeros((n_time, m, 3), dtype=hip.torch.float64, device="cuda")
        eng = hip.engine.HotPathEngine(consts, m, E, c2t(), zn, history=K + 1, zn_stride_env=0)
        eng.load_state(0, xt, x, P)
        for k in range(1, K + 1):
            eng.set_actions([(5 * e + k) % m for e in range(E)])
            defer = mode == "deferred" or (mode == "mixed" and k in (1, 2, 4))   # 3 and 5 fold their predecessor first
            eng.launch_step(k - 1, k, k, fast_stats=True, defer_fold=defer)
        eng.flush_stats()
        eng.flush_stats()                                  # idempotent
        hip.torch.cuda.synchronize()
        outs.append((eng.stats.cpu().numpy(), eng.x_filter[K].cpu().numpy(), eng.P_filter[K].cpu().numpy(),
                     eng.status.cpu().numpy(), eng._shard_sets.cpu().numpy()))
    for o in outs[1:]:
        assert np.array_equal(outs[0][0][1:], o[0][1:], equal_nan=True)      # statistics of steps 1..K
        for a, b in zip(outs[0][1:4], o[1:4]):
            assert np.array_equal(a, b, equal_nan=True)
        assert not o[4].any()                                                # every shard set folded and cleared
    assert outs[0][0][1, 0, hip.lib.STAT_MAX_DPOS] > 1e19 and outs[0][0][K, 0, hip.lib.STAT_N_FAILED] >= 1   # the NaN state failed: sentinels


@pytest.mark.parametrize("propagator,obs_type,resample,interval", [("fg", "aer", False, 1), ("j2", "aer", False, 1),
                                                                    ("elements", "aer", False, 1),
                                                                    ("fg", "xyz", True, 1), ("fg", "aer", True, 3)])
@pytest.mark.parametrize("E,m,K,H", [(1, 2003, 7, 8), (3, 50, 9, 4), (2, 30001, 3, 2)])
def test_rollout_equals_single_steps(hip, E, m, K, H, propagator, obs_type, resample, interval):
    """ssa_env_rollout_f64: K steps in one launch (state resident in LDS across the steps) must reproduce K
    ssa_env_step_f64 launches BITWISE: every surviving history slot (states, covariances, observations,
    metrics, update records), status, and the statistics of the last min(K, H) steps.  Sizes cover several
    envs per wavefront, ragged tiles, several tiles per wavefront (> 20 480 objects), ring wrap-around
    (K > H), a failing filter and an update in every step."""
    xt, x, P, g = make_batch(E * m, seed=123)
    x[9, 1] = np.nan
    R = g["R"] if obs_type == "aer" else np.diag([5e2 ** 2] * 3)
    consts = hip.host.make_consts(g["Q"], R, 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"], propagator=propagator,
                                  obs_type=obs_type, resample=resample, update_interval=interval)
    n_time = c2t().shape[0]
    acts = np.array([[(7 * e + 3 * k) % m for e in range(E)] for k in range(K)], dtype=np.int32)
    zn_host = np.random.RandomState(8).normal(size=(n_time, m, 3)) * np.array([4.8e-6, 4.8e-6, 1e3]) if m < 5000 else None
    outs = []
    for mode in ("steps", "rollout"):
        zn = hip.torch.as_tensor(zn_host).cuda() if zn_host is not None else \
            hip.torch.zeros((n_time, m, 3), dtype=hip.torch.float64, device="cuda")
        eng = hip.engine.HotPathEngine(consts, m, E, c2t(), zn, history=H, zn_stride_env=0)
        eng.load_state(0, xt, x, P)
        if mode == "steps":
            for k in range(K):
                eng.set_actions(acts[k])
                eng.launch_step(k % H, (k + 1) % H, 1 + k, fast_stats=True)
        else:
            eng.launch_rollout(0, 1, hip.torch.as_tensor(acts).cuda())
        hip.torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (eng.x_true, eng.x_filter, eng.P_filter, eng.obs, eng.metrics, eng.upd, eng.status,
                                               eng.stats)] + [eng._shard_sets.cpu().numpy()])
    a, b = outs
    L = hip.lib

    def defined_fields(u):
        """an update record defines its flags always, z_true when the object was visible, y / S / sigmas_h when the
        observation was taken; the other words of a ring slot are leftovers of the step that used it before"""
        u = u.copy()
        vis, taken = u[..., L.UPD_VISIBLE] == 1.0, u[..., L.UPD_OBS_TAKEN] == 1.0
        keep = np.zeros(u.shape, dtype=bool)
        keep[..., [L.UPD_OBS_TAKEN, L.UPD_VISIBLE, L.UPD_ACTION]] = True
        keep[..., L.UPD_Z_TRUE:L.UPD_Z_TRUE + 3] = vis[..., None]
        keep[..., L.UPD_Y:L.UPD_SIGMAS_H + 39] = taken[..., None]
        u[~keep] = 0.0
        return u
    a[5], b[5] = defined_fields(a[5]), defined_fields(b[5])
    names = ("x_true", "x_filter", "P_filter", "obs", "metrics", "upd", "status")
    for nme, u, v in zip(names, a, b):
        if nme == "metrics" or u.ndim == 1 or K >= H:
            assert np.array_equal(u, v, equal_nan=True), nme
        else:   # slots never written keep their initial fill
            sl = [s_ % H for s_ in range(0, K + 1)]
            assert np.array_equal(u[sl], v[sl], equal_nan=True), nme
    for k in range(max(0, K - H), K):                     # statistics of the steps whose slot survives
        so = (k + 1) % H
        for sidx in (L.STAT_MAX_DPOS, L.STAT_CNT_LT_1E4, L.STAT_CNT_LT_1E7, L.STAT_N_FAILED):
            assert np.array_equal(a[7][so, :, sidx], b[7][so, :, sidx], equal_nan=True), (k, sidx)
    assert a[7][K % H, 0, L.STAT_N_FAILED] >= 1 and np.any(a[5][..., L.UPD_OBS_TAKEN] == 1.0)


def test_direct_rccl_all_gather_equals_torch_distributed(hip):
    """parallel.ShardedStepper enqueues RCCL's ncclAllGather itself (rccl.py) instead of going through the
    ProcessGroup's internal stream; on a 1-rank group both routes must deliver the same payload (observation
    block + statistics) for the same steps.  (More ranks: tests/test_parallel_gloo.py covers the host logic.)"""
    import torch.distributed as dist
    torch = hip.torch
    from ssa_gym_amd import parallel
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29537", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        m = 1003
        xt, x, P, g = make_batch(m, seed=5)
        consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
        n_time = c2t().shape[0]
        got = []
        for direct in (True, False):
            zn = torch.zeros((n_time, m, 3), dtype=torch.float64, device="cuda")
            eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), zn, history=2, zn_stride_env=0)
            eng.load_state(0, xt, x, P)
            local = parallel.HipLocalStepper(eng, consts, fast_stats=True)
            sh = parallel.ShardedStepper(parallel.ShardPlan(m, 1, 0), local, direct_rccl=direct)
            assert (sh._rccl is not None) == direct
            for k in range(9):   # in-stream and overlapped steps mixed: three payload buffers rotate, every step carries the RAW
                sh.step(7 * k + 1, overlap=k in (2, 3, 4, 6, 8))   # shard words (no fold launch), step k zeroes buffer (k + 1) % 3's words
                sh.wait()
                torch.cuda.synchronize()
                assert sh._raw[k % 3]
                st = sh.global_stats()     # folded on arrival they must equal the numpy reductions of this step's metrics
                dp = eng.metrics[local.tick % 2, 0, 0].cpu().numpy()
                assert st[hip.lib.STAT_MAX_DPOS] == dp.max() and st[hip.lib.STAT_CNT_LT_1E4] == (dp < 1e4).sum(), k
                assert st[hip.lib.STAT_CNT_LT_1E7] == (dp < 1e7).sum() and st[hip.lib.STAT_N_FAILED] == 0, k
            for k in range(9, 14):   # and without a host sync in between (the events alone order the buffers)
                sh.step(7 * k + 1, overlap=True)
            sh.wait()
            torch.cuda.synchronize()
            st = sh.global_stats()
            dp = eng.metrics[local.tick % 2, 0, 0].cpu().numpy()
            assert st[hip.lib.STAT_MAX_DPOS] == dp.max() and st[hip.lib.STAT_CNT_LT_1E7] == (dp < 1e7).sum()
            got.append((sh.global_obs().cpu().numpy(), sh.global_stats(), sh.recv[0].cpu().numpy(), sh.recv[1].cpu().numpy(),
                        sh.recv[2].cpu().numpy()))
            if sh._rccl is not None:
                sh._rccl.close()
        for a, b in zip(got[0], got[1]):
            assert np.array_equal(a, b, equal_nan=True)
        assert np.abs(got[0][0]).max() > 0
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("propagator", ["hybrid", "fg"])
def test_full_size_20000_parity_with_update(hip, oracle, oracle_ld, propagator):
    """BASELINE config 3 size against the oracle itself (not only properties): 20 000 objects, alpha = 1e-4,
    the selected object updated, fp64 oracle and 80-bit witness on the same inputs (a few seconds of CPU).
    Tolerances: north_star (1e-6 means, 1e-5 covariances), criterion of check_parity.  The env default (hybrid: the covariance in the
    reference's own arithmetic, so no bound against the exact value -- as in test_predict_parity_2000_objects) and fg."""
    m = 20000
    xt, x, P, g = make_batch(m, seed=2024)
    a = 12345
    gpu = run_gpu(hip, xt, x, P, g, [a], 2, 1e-4, propagator=propagator)
    zn = gpu["z_noise"][0, 2, a]
    f64 = run_oracle(oracle, xt, x, P, g, a, 2, 1e-4, z_noise3=zn)
    ld = run_oracle(oracle_ld, xt, x, P, g, a, 2, 1e-4, centred=True, z_noise3=zn)
    assert np.all(gpu["status"] == 0) and np.all(f64["status"] == 0)
    assert_states_close(gpu["x_true"], f64["x_true"], 1e-9, "truth")
    others = np.arange(m) != a
    sub = lambda d: {k: d[k][others] for k in ("x", "P")}       # the updated object is judged separately (finding 3 of DESIGN section 4)
    check_parity(sub(gpu), sub(f64), sub(ld), exact_bound=(propagator == "fg"), min_well=0.875, tag=" %s 20000 objects" % propagator)   # measured 0.8842
    assert gpu["upd"][0, hip.lib.UPD_OBS_TAKEN] == 1.0
    # the update: posterior mean within the reference arithmetic's own distance from the witness (x3), trace P collapsed
    ep = np.linalg.norm(gpu["x"][a, :3] - ld["x"][a, :3]) / np.linalg.norm(ld["x"][a, :3])
    rp = np.linalg.norm(f64["x"][a, :3] - ld["x"][a, :3]) / np.linalg.norm(ld["x"][a, :3])
    assert ep <= 3 * rp + 1e-9, (ep, rp)
    assert np.trace(gpu["P"][a]) < 1e-3 * np.trace(P[a])
    assert gpu["stats"][0, hip.lib.STAT_N_FAILED] == 0


def test_predict_only_drift_over_150_steps(hip, oracle, oracle_ld):
    """Trajectory drift (reported separately from the per-step gates): 150 predict-only steps of 500
    objects in ONE rollout launch against 150 oracle steps.  The truth trajectories stay at Kepler parity; the
    filter means and covariances are compared with the 80-bit witness: the kernel must be no further from it than
    the reference arithmetic is (whose own summation noise accumulates just the same)."""
    m, K = 500, 150
    xt, x, P, g = make_batch(m, seed=314)
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
    n_time = c2t().shape[0]
    zn = hip.torch.zeros((n_time, m, 3), dtype=hip.torch.float64, device="cuda")
    eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), zn, history=2, zn_stride_env=0)
    eng.load_state(0, xt, x, P)
    eng.launch_rollout(0, 1, hip.torch.full((K, 1), -1, dtype=hip.torch.int32, device="cuda"))
    hip.torch.cuda.synchronize()
    slot = K % 2
    gx, gP, gt = eng.x_filter[slot].cpu().numpy(), eng.P_filter[slot].cpu().numpy(), eng.x_true[slot].cpu().numpy()
    assert int((eng.status != 0).sum().item()) == 0
    Wm, Wc, scale = orc.merwe_weights(1e-4, 2.0, -3)
    res = {}
    for name, o, centred in (("f64", oracle, False), ("ld", oracle_ld, True)):
        a, b, c, st = xt.copy(), x.copy(), P.copy(), np.zeros(m, dtype=np.int32)
        for k in range(K):
            r = o.env_step(a, b, c, st, 20.0, g["Q"], g["R"], Wm, Wc, scale, -1, c2t()[(1 + k) % n_time], g["obs_lla"], g["obs_itrs"],
                           -np.pi / 2, np.zeros(3), centred=centred, do_update=False)
            a, b, c = r["x_true"], r["x"], r["P"]
        assert np.all(st == 0)
        res[name] = (a, b, c)
    assert_states_close(gt, res["ld"][0], 1e-9, "truth after %d steps" % K)
    rel = lambda u, v: np.linalg.norm((u - v)[:, :3], axis=1) / np.linalg.norm(v[:, :3], axis=1)
    e_gpu, e_ref = rel(gx, res["ld"][1]), rel(res["f64"][1], res["ld"][1])
    print("drift after %d predict-only steps vs 80-bit witness: GPU median %.2e max %.2e | reference arithmetic median %.2e max %.2e"
          % (K, np.median(e_gpu), e_gpu.max(), np.median(e_ref), e_ref.max()))
    # 150 steps of this (unstable, alpha = 1e-4) recursion amplify the per-step rounding of ANY fp64 evaluation to
    # 1e-6..1e-4 -- the 80-bit witness included: measured GPU median 3.6e-6 / max 3.2e-5, reference arithmetic
    # 4.2e-6 / 3.0e-4.  The gate is therefore relative to the reference arithmetic, plus a loose absolute bound.
    assert np.median(e_gpu) <= 3 * np.median(e_ref) + 1e-12 and e_gpu.max() <= 3 * e_ref.max() + 1e-12
    assert e_gpu.max() < 1e-3
    sd = np.sqrt(np.einsum('jii->ji', res["ld"][2]))
    nP = lambda c: np.max(np.abs(c - res["ld"][2]) / (sd[:, :, None] * sd[:, None, :]), axis=(1, 2))
    eP_gpu, eP_ref = nP(gP), nP(res["f64"][2])
    assert np.median(eP_gpu) <= 3 * np.median(eP_ref) + 1e-12 and eP_gpu.max() <= 3 * eP_ref.max() + 1e-9


@pytest.mark.parametrize("propagator", ["fg", "elements", "hybrid"])
@pytest.mark.parametrize("resample", [False, True])
def test_reference_test6_test7_on_the_hip_path(hip, resample, propagator):
    """The reference's only numeric UKF pins (tests.py:118-188), run through ssa_env_step_f64 and held to the
    reference's OWN thresholds: Test 6 -- 50 predicts at dt = 30 s, alpha = 1e-3, P0 = diag(1000 x3, 1 x3), qvar 1e-6^2:
    |pos error| < 1 m, |vel error| < 1e-4 m/s; Test 7 -- one update with the exact position (hx_xyz, R given 1-D =
    125 * ones(3,3) after filterpy's broadcast): both < 1e-2; 50 more predicts and a second
    exact update: both < 1e-2 again (tests.py:174-186).  The env step propagates the truth and predicts in the same
    launch; the update of step 50 uses that step's propagated sigma points exactly as predict() x 50 then update()."""
    g = golden("test67_golden.npz")
    gs = golden("ukf_step_golden.npz")
    m = 5                                     # the same object in every row of a ragged tile (4 + 1)
    dt, alpha = float(g["dt"]), 1e-3
    R = 125.0 * np.ones((3, 3))
    consts = hip.host.make_consts(g["Q"], R, alpha, 2.0, -3, dt, -np.pi / 2, gs["obs_lla"], obs_type='xyz',
                                  propagator=propagator, resample=resample)
    zn = hip.torch.zeros((1, 480, m, 3), dtype=hip.torch.float64, device="cuda")      # z = x_true[:3] exactly
    eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), zn, history=2)
    eng.load_state(0, np.tile(g["x0"], (m, 1)), np.tile(g["x0"], (m, 1)), np.tile(g["P0"], (m, 1, 1)))

    def err(slot, j):
        d = eng.x_filter[slot, j].cpu().numpy() - eng.x_true[slot, j].cpu().numpy()
        return np.linalg.norm(d[:3]), np.linalg.norm(d[3:])

    k = 0
    for _ in range(49):
        eng.set_actions([-1])
        eng.launch_step(k % 2, (k + 1) % 2, k + 1)
        k += 1
    # Test 6: state after the 50th predict = prior of step 50; take it from a predict-only twin of that step
    eng.set_actions([-1])
    eng.launch_step(k % 2, (k + 1) % 2, k + 1)
    hip.torch.cuda.synchronize()
    np.testing.assert_allclose(eng.x_true[(k + 1) % 2, 0].cpu().numpy(), g["xt50"], rtol=1e-12)
    e6 = err((k + 1) % 2, 0)
    P50 = eng.P_filter[(k + 1) % 2, 0].cpu().numpy()
    np.testing.assert_allclose(P50, g["P50"], rtol=1e-5, atol=1e-5 * np.abs(g["P50"]).max())
    # the same step again, now with the update of object 2 (slot k still holds the state after 49 steps)
    eng.set_actions([2])
    eng.launch_step(k % 2, (k + 1) % 2, k + 1)
    hip.torch.cuda.synchronize()
    k += 1
    assert eng.upd[k % 2, 0, hip.lib.UPD_OBS_TAKEN].item() == 1.0 and int((eng.status != 0).sum().item()) == 0
    e7 = err(k % 2, 2)
    e6_twin = err(k % 2, 0)                   # rows that were not selected stay predict-only
    x50u = eng.x_filter[k % 2, 2].cpu().numpy()
    for _ in range(49):
        eng.set_actions([-1])
        eng.launch_step(k % 2, (k + 1) % 2, k + 1)
        k += 1
    eng.set_actions([2])
    eng.launch_step(k % 2, (k + 1) % 2, k + 1)
    hip.torch.cuda.synchronize()
    k += 1
    e7b = err(k % 2, 2)
    print("[Test 6/7 on HIP, %s%s] after 50 predicts: pos %.4g m vel %.4g m/s | after update: pos %.4g vel %.4g | "
          "after 50 more predicts + update: pos %.4g vel %.4g   (reference restatement: %s)"
          % (propagator, " resample" if resample else "", *e6, *e7, *e7b, g["err_rs" if resample else "err"]))
    assert e6 == e6_twin
    assert e6[0] < 1.0 and e6[1] < 1.0e-4                 # tests.py:156-157
    assert e7[0] < 1.0e-2 and e7[1] < 1.0e-2              # tests.py:170-171
    assert e7b[0] < 1.0e-2 and e7b[1] < 1.0e-2            # tests.py:185-186
    # and the posterior agrees with the restated-filterpy value of the same scenario (composite golden)
    ref = g["x50u_rs" if resample else "x50u"]
    assert np.linalg.norm((x50u - ref)[:3]) < 2e-2 and np.linalg.norm((x50u - ref)[3:]) < 2e-3


def _energy_j2(s, j2, r_eq):
    """specific energy including the J2 potential (conserved by the J2 dynamics; z-angular momentum too)"""
    mu = 398600441800000.0
    r = np.linalg.norm(s[:, :3], axis=1)
    sinphi = s[:, 2] / r
    return 0.5 * np.sum(s[:, 3:] ** 2, 1) - mu / r + mu * j2 * r_eq ** 2 / (2 * r ** 3) * (3 * sinphi ** 2 - 1)


def test_full_size_20000_j2_leg(hip):
    """BASELINE config 3 with "J2 on" at its full size (20 000 objects): the extension propagator in the fused step.
    No reference counterpart -> size-independent properties on all objects (J2 energy and h_z
    conserved along the truth, covariances SPD, bitwise determinism, the J2 = 0 limit reproduces the two-body
    kernel's filter output) plus spot parity of the truth against scipy DOP853 (the reference's unused
    fx_xyz_cowell configuration, dynamics.py:184-193) on 24 objects."""
    import j2_reference as J
    m = 20000
    xt, x, P, g = make_batch(m, seed=77)
    a = run_gpu(hip, xt, x, P, g, [4321], 1, 1e-4, propagator='j2')
    b = run_gpu(hip, xt, x, P, g, [4321], 1, 1e-4, propagator='j2')
    for k in ("x", "P", "x_true", "obs", "metrics"):
        assert np.array_equal(a[k], b[k]), k
    assert np.all(a["status"] == 0) and a["upd"][0, hip.lib.UPD_OBS_TAKEN] == 1.0
    assert np.linalg.eigvalsh(a["P"]).min() > 0
    j2, req = hip.host.J2_EARTH, hip.host.R_EQ_EARTH
    np.testing.assert_allclose(_energy_j2(a["x_true"], j2, req), _energy_j2(xt, j2, req), rtol=2e-12)
    hz = lambda s: s[:, 0] * s[:, 4] - s[:, 1] * s[:, 3]   # noqa: E731
    np.testing.assert_allclose(hz(a["x_true"]), hz(xt), rtol=1e-11, atol=1e-3)
    # J2 really acts: the truth differs from the two-body truth by metres in LEO, and matches DOP853 + J2
    kep = run_gpu(hip, xt, x, P, g, [4321], 1, 1e-4, propagator='fg')
    d = np.linalg.norm((a["x_true"] - kep["x_true"])[:, :3], axis=1)
    assert d.max() > 1.0 and np.median(d) > 1e-3
    rs = np.random.RandomState(0)
    for j in rs.choice(m, 24, replace=False):
        ref = J.fx_xyz_cowell_j2(xt[j], 20.0)
        assert np.linalg.norm((a["x_true"][j] - ref)[:3]) / np.linalg.norm(ref[:3]) < 1e-10
        assert np.linalg.norm((a["x_true"][j] - ref)[3:]) / np.linalg.norm(ref[3:]) < 1e-9
    # the filter side: with j2 = 0 the RK4 path must reproduce the parity-checked two-body kernel
    c0 = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"], propagator='j2', j2=0.0)
    eng = hip.engine.HotPathEngine(c0, m, 1, c2t(), kep["z_noise"], history=2)
    eng.load_state(0, xt, x, P)
    eng.set_actions([4321])
    eng.launch_step(0, 1, 1)
    hip.torch.cuda.synchronize()
    x0 = eng.x_filter[1].cpu().numpy()
    others = np.arange(m) != 4321
    assert_states_close(x0[others], kep["x"][others], 1e-6, "J2 = 0 limit vs two-body kernel")
    sd = np.sqrt(np.einsum('jii->ji', kep["P"]))
    eP = np.max(np.abs(eng.P_filter[1].cpu().numpy() - kep["P"]) / (sd[:, :, None] * sd[:, None, :]), axis=(1, 2))
    assert eP[others].max() < 1e-5


def test_160000_objects_single_launch_and_8x20000_vector(hip, oracle, oracle_ld):
    """The per-GPU loads of BASELINE configs 4 and 5 at full size on one GPU: 160 000 objects of ONE env in a single
    launch (each wavefront advances 8 tiles, prefetching the next), and the same objects as 8 envs x 20 000 with one
    action per env.  Objects are independent, so (i) every object that is not selected must come out bit-identical
    in both groupings and identical to a 20 000-object single-tile launch of its block; (ii) a sample is compared
    with the CPU oracle; (iii) statistics equal numpy reductions over the metrics."""
    m, E = 20000, 8
    N = m * E
    xt, x, P, g = make_batch(N, seed=4242)
    rs = np.random.RandomState(7)
    zn = rs.normal(size=(480, 3)) * np.array([4.8e-6, 4.8e-6, 1e3])     # one noise triple per time step (strides 0, 3, 0)
    one = run_gpu(hip, xt, x, P, g, [150001], 3, 1e-4, z_noise=zn, zn_strides=(0, 3, 0))   # config 4's object count, one env
    acts = [11, 19999, 0, 7777, 12345, 1, 19998, 4242]
    vec = run_gpu(hip, xt, x, P, g, acts, 3, 1e-4, E=E, z_noise=zn, zn_strides=(0, 3, 0))
    assert np.all(one["status"] == 0) and np.all(vec["status"] == 0)
    sel_one = np.zeros(N, dtype=bool)
    sel_one[150001] = True
    sel_vec = np.zeros(N, dtype=bool)
    sel_vec[[e * m + a for e, a in enumerate(acts)]] = True
    same = ~(sel_one | sel_vec)
    for k in ("x", "P", "x_true", "obs"):
        assert np.array_equal(one[k][same], vec[k][same]), k
    assert np.array_equal(one["metrics"].reshape(4, N)[:, same], np.transpose(vec["metrics"], (1, 0, 2)).reshape(4, N)[:, same])
    for e in range(E):
        assert vec["upd"][e, hip.lib.UPD_OBS_TAKEN] == 1.0 and vec["upd"][e, hip.lib.UPD_ACTION] == acts[e]
    assert one["upd"][0, hip.lib.UPD_ACTION] == 150001
    # a 20 000-object launch (one tile per wavefront) of block 5 gives the same bits as the 8-tile wavefronts
    sl = slice(5 * m, 6 * m)
    blk = run_gpu(hip, xt[sl], x[sl], P[sl], g, [-1], 3, 1e-4)
    keep = same[sl]
    for k in ("x", "P", "x_true"):
        assert np.array_equal(blk[k][keep], one[k][sl][keep]), k
    # statistics of the 160 000-object env and of each 20 000-object env against numpy
    dp = one["metrics"][0, 0]
    assert one["stats"][0, hip.lib.STAT_MAX_DPOS] == dp.max()
    assert one["stats"][0, hip.lib.STAT_CNT_LT_1E4] == (dp < 1e4).sum() and one["stats"][0, hip.lib.STAT_CNT_LT_1E7] == (dp < 1e7).sum()
    for e in range(E):
        dpe = vec["metrics"][e, 0]
        assert vec["stats"][e, hip.lib.STAT_MAX_DPOS] == dpe.max() and vec["stats"][e, hip.lib.STAT_CNT_LT_1E7] == (dpe < 1e7).sum()
    # oracle parity on a sample of 2 000 objects spread over the whole range (incl. the last, ragged-free, tile)
    idx = np.sort(rs.choice(N, 2000, replace=False))
    f64 = run_oracle(oracle, xt[idx], x[idx], P[idx], g, -1, 3, 1e-4, z_noise3=np.zeros(3))
    ok = same[idx]
    assert_states_close(one["x_true"][idx], f64["x_true"], 1e-9, "truth 160k")
    # ... through the same two-sided criterion as the 2 000- and 20 000-object tests: within the north_star tolerance of the
    # reference value wherever that value is defined to half the tolerance (fraction measured and asserted), never further from
    # the 80-bit witness than the reference arithmetic, and within the tolerance of the witness on ALL objects (SSA_PROP_FG)
    sub = idx[ok]
    f64s = {k: f64[k][ok] for k in ("x", "P")}
    ld = run_oracle(oracle_ld, xt[sub], x[sub], P[sub], g, -1, 3, 1e-4, centred=True, z_noise3=np.zeros(3))
    gpu_s = {k: one[k][sub] for k in ("x", "P")}
    check_parity(gpu_s, f64s, ld, exact_bound=True, well_frac=0.5, min_well=0.86, tag=" 160000-object sample")


def test_graphed_sharded_steps_equal_eager_steps(hip):
    """parallel.GraphedShardedSteps: units of sharded steps (step kernel + all-gather each) captured into a hipGraph and
    replayed -- with the all-gather in the compute stream and forked onto the communication stream inside the graph -- leave
    the same states, payloads and statistics as the same steps enqueued one by one from the host.  One rank (RCCL's
    ncclAllGather is captured too); the device-side statistics fold equals the host one."""
    import torch.distributed as dist
    torch = hip.torch
    from ssa_gym_amd import parallel
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        m, U, NU = 2003, 4, 6
        xt, x, P, g = make_batch(m, seed=15)
        consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"])
        n_time = c2t().shape[0]
        zn_h = np.random.RandomState(3).normal(size=(n_time, m, 3)) * np.array([4.8e-6, 4.8e-6, 1e3])
        plan = parallel.ShardPlan(m, 1, 0)
        sched = (np.arange(4 * U * NU) * 37 + 5) % m            # global actions of consecutive steps (cyclic)

        def fresh():
            eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), torch.as_tensor(zn_h).cuda()[None].contiguous(), history=2)
            eng.load_state(0, xt, x, P)
            local = parallel.HipLocalStepper(eng, consts, fast_stats=True)
            return eng, local, parallel.ShardedStepper(plan, local, obs_cols=1)
        eng, local, sh = fresh()
        for k in range(U * NU):
            sh.step(int(sched[k]))
        sh.wait()
        torch.cuda.synchronize()
        want = (eng.x_filter[local.tick % 2].cpu().numpy(), eng.P_filter[local.tick % 2].cpu().numpy(), eng.x_true[local.tick % 2].cpu().numpy(),
                eng.status.cpu().numpy(), sh.global_obs().cpu().numpy(), sh.global_stats())
        assert np.isfinite(want[4]).all() and want[5][hip.lib.STAT_CNT_LT_1E7] == m
        sh.close()
        for overlap in (False, True):
            eng, local, sh = fresh()
            gs = parallel.GraphedShardedSteps(sh, U, sched, overlap=overlap)
            gs.rewind()
            for _ in range(NU):
                gs.run_unit()
            sh.wait()
            torch.cuda.synchronize()
            assert len(gs._graphs) == 3 and local.tick == U * NU          # (one capture per phase, the other three units were replays)
            got = (eng.x_filter[local.tick % 2].cpu().numpy(), eng.P_filter[local.tick % 2].cpu().numpy(), eng.x_true[local.tick % 2].cpu().numpy(),
                   eng.status.cpu().numpy(), sh.global_obs().cpu().numpy(), sh.global_stats())
            for a, b in zip(got, want):
                assert np.array_equal(a, b, equal_nan=True), overlap
            dev_stats = sh.global_stats_device().cpu().numpy()
            assert np.array_equal(dev_stats[:3], want[5][:3]) and dev_stats[hip.lib.STAT_N_FAILED] == want[5][hip.lib.STAT_N_FAILED]
            assert int(eng.env_time0.item()) == U * NU and int(gs.cursor.item()) == U * NU
            sh.close()
        # the all-gather by direct peer stores (peer.py; at one rank the "peer" is the rank's own arena): per step and as replayed units,
        # the same bits; the wait on a step nobody pushed ends at its bound and says which source is missing
        def fresh_peer():
            eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), torch.as_tensor(zn_h).cuda()[None].contiguous(), history=2)
            eng.load_state(0, xt, x, P)
            local = parallel.HipLocalStepper(eng, consts, fast_stats=True)
            return eng, local, parallel.ShardedStepper(plan, local, obs_cols=1, exchange="peer")
        for graphed in (False, True):
            eng, local, sh = fresh_peer()
            assert sh._peer is not None and sh._rccl is None
            if graphed:
                gs = parallel.GraphedShardedSteps(sh, U, sched)
                gs.rewind()
                for _ in range(NU):
                    gs.run_unit()
                assert gs.capture_failed is None and len(gs._graphs) == 3
            else:
                for k in range(U * NU):
                    sh.step(int(sched[k]))
            sh.wait()
            torch.cuda.synchronize()
            sh._peer.check()
            assert int(sh._peer.flags.max().item()) == U * NU and sh._peer.seq0_host == (U * NU if graphed else 0) == int(sh._peer.seq0.item())
            got = (eng.x_filter[local.tick % 2].cpu().numpy(), eng.P_filter[local.tick % 2].cpu().numpy(), eng.x_true[local.tick % 2].cpu().numpy(),
                   eng.status.cpu().numpy(), sh.global_obs().cpu().numpy(), sh.global_stats())
            for a, b in zip(got, want):
                assert np.array_equal(a, b, equal_nan=True), graphed
            if not graphed:
                sh._peer.timeout_ticks = 100000          # 1 ms
                sh._peer.wait(0, U * NU + 5, torch.cuda.current_stream().cuda_stream)      # a step that was never pushed
                torch.cuda.synchronize()
                with pytest.raises(hip.lib.SsaHipError, match="rank 0's payload did not arrive"):
                    sh._peer.check()
                sh.step(int(sched[0]))                   # (the exchange still works afterwards)
                sh.wait()
                torch.cuda.synchronize()
                sh._peer.check()
            sh.close()
        # a phase whose capture fails keeps the host and device bookkeeping in step (the eager unit ran; tick and k advanced) and is
        # enqueued eagerly from then on: same results, no exception, `capture_failed` says why
        eng, local, sh = fresh()
        gs = parallel.GraphedShardedSteps(sh, U, sched, overlap=False)
        gs.rewind()
        gs._force_capture_failure = True
        for _ in range(NU):
            gs.run_unit()
        sh.wait()
        torch.cuda.synchronize()
        assert gs.capture_failed is not None and all(v is False for v in gs._graphs.values()) and local.tick == U * NU and sh.k == U * NU
        got = (eng.x_filter[local.tick % 2].cpu().numpy(), eng.P_filter[local.tick % 2].cpu().numpy(), eng.x_true[local.tick % 2].cpu().numpy(),
               eng.status.cpu().numpy(), sh.global_obs().cpu().numpy(), sh.global_stats())
        for a, b in zip(got, want):
            assert np.array_equal(a, b, equal_nan=True)
        assert int(eng.env_time0.item()) == U * NU and int(gs.cursor.item()) == U * NU
        sh.close()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("resample", [False, True])
def test_resample_variant_fails_in_the_predict_that_draws_the_points(hip, oracle, resample):
    """The two published predict() variants differ in WHEN an exhausted robust_cholesky ladder is seen: the variant that
    keeps the propagated points draws sigma points only at the start of the NEXT predict (failure flagged one step later);
    the variant that redraws at the end of predict() fails the filter in the SAME step -- for every filter, not only the
    one being updated.  A hugely negative process noise makes the prior indefinite beyond the ladder (1e9 I)."""
    m, alpha = 6, 1e-3
    xt, x, P, g = make_batch(m, seed=21)
    Q = g["Q"].copy()
    Q[2, 2] = -1e30
    consts = hip.host.make_consts(Q, g["R"], alpha, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"], resample=resample)
    eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), np.zeros((1, 480, m, 3)), history=2)
    eng.load_state(0, xt, x, P)
    Wm, Wc, scale = orc.merwe_weights(alpha, 2.0, -3)
    st = np.zeros(m, dtype=np.int32)
    a, b, c = xt, x, P
    for k in (1, 2):
        eng.set_actions([1])
        eng.launch_step((k - 1) % 2, k % 2, k)
        hip.torch.cuda.synchronize()
        r = oracle.env_step(a, b, c, st, 20.0, Q, g["R"], Wm, Wc, scale, 1, c2t()[k], g["obs_lla"], g["obs_itrs"], -np.pi / 2,
                            np.zeros(3), resample=resample)
        a, b, c = r["x_true"], r["x"], r["P"]
        gst = eng.status.cpu().numpy()
        assert np.array_equal(gst, st), (k, gst, st)
        if k == 1:
            assert np.all(gst == (2 if resample else 0))       # redraw: every filter fails in step 1's predict
        else:
            assert np.all(gst == 2)                            # keep: the same failure, one step later
    assert np.array_equal(eng.x_filter[0].cpu().numpy(), np.tile(hip.host.X_FAILED, (m, 1)))


@pytest.mark.parametrize("prop", ["hybrid", "fg"])
def test_an_objects_arithmetic_does_not_depend_on_its_position(hip, prop):
    """The storage layout of round 4 (catalogue.regime_order + HotPathEngine.set_layout: objects of one regime share wavefronts) rests on this:
    the same objects, filter states and noise stored in another order give, object by object, the SAME BITS -- states, covariances, truth,
    status, observation rows, statistics -- over 40 steps with an update in every step, with a third of the filters inflated so far that their
    sigma points leave the strong-elliptic regime (conic tier, jitter ladder, failures included).  2 016 objects (whole tiles per XCD run)."""
    torch = hip.torch
    from ssa_gym_amd import parallel
    from ssa_gym_amd.catalogue import regime_order
    m, n_steps = 2016, 40
    xt, x, P, g = make_batch(m, seed=31)
    rs = np.random.RandomState(8)
    wild = rs.uniform(size=m) < 0.33
    P[wild] *= 3e4                                   # (sd 1.7e7 m / 1.7e4 m/s: hyperbolic sigma points, rank-deficient (n + lambda) P after a few steps)
    zn = rs.normal(size=(1, 480, m, 3)) * np.array([4.8e-6, 4.8e-6, 1e3])
    order = regime_order(xt)
    assert not np.array_equal(order, np.arange(m))
    inv = np.empty(m, dtype=np.int64)
    inv[order] = np.arange(m)
    acts = (np.arange(n_steps) * 53 + 7) % m         # original object ids
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"], propagator=prop,
                                  **({"covariance": "reference"} if prop == "hybrid" else {}))

    def run(perm, actions):
        eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), np.ascontiguousarray(zn[:, :, perm]), history=2)
        eng.load_state(0, xt[perm], x[perm], P[perm])
        local = parallel.HipLocalStepper(eng, consts, fast_stats=True)
        stats = []
        for a in actions:
            st = torch.zeros(8, dtype=torch.float64, device="cuda")
            local.step(int(a), None, st)
            stats.append(st)
        torch.cuda.synchronize()
        s = local.tick % 2
        return (eng.x_true[s].cpu().numpy(), eng.x_filter[s].cpu().numpy(), eng.P_filter[s].cpu().numpy(), eng.status.cpu().numpy(),
                eng.obs[s].cpu().numpy()), torch.stack(stats).cpu().numpy()
    ident, st_a = run(np.arange(m), acts)
    sorted_, st_b = run(order, inv[acts])
    for name, a, b in zip(("x_true", "x_filter", "P_filter", "status", "obs"), ident, sorted_):
        assert np.array_equal(a[order], b, equal_nan=True), name
    k = [hip.lib.STAT_MAX_DPOS, hip.lib.STAT_CNT_LT_1E4, hip.lib.STAT_CNT_LT_1E7, hip.lib.STAT_N_FAILED]
    assert np.array_equal(st_a[:, k], st_b[:, k], equal_nan=True)
    if prop == "hybrid":
        assert (ident[3] != 0).sum() > 0             # the workload does reach the failure path


@pytest.mark.parametrize("prop,m", [("hybrid", 2016), ("hybrid", 2014), ("fg", 2016)])
def test_storage_layout_speaks_the_callers_indices(hip, prop, m):
    """ssa_step_params.obj_ids / HotPathEngine.set_layout: the objects stored in another order than the caller numbers them (a random
    permutation here; the env uses catalogue.regime_order).  Fed the SAME actions, the engine with a layout must hand the caller -- step by
    step, bit for bit -- the same host-facing observation rows (obs_mirror: at the caller's row), the same 'aer' block, statistics
    including np.argmax(sigma_pos) (ties between failed filters: the lowest CALLER index), update records and failure records, and after
    to_caller_order() the same state tensors.  A third of the filters is inflated so that the conic tier, the ladder and the failure path
    run; 2 014 objects: a ragged last tile."""
    torch = hip.torch
    xt, x, P, g = make_batch(m, seed=33)
    rs = np.random.RandomState(9)
    P[rs.uniform(size=m) < 0.33] *= 3e4
    zn = rs.normal(size=(1, 480, m, 3)) * np.array([4.8e-6, 4.8e-6, 1e3])
    order = rs.permutation(m)
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"], propagator=prop)
    n_steps = 70
    acts = (np.arange(n_steps) * 53 + 7) % m

    def run(layout):
        eng = hip.engine.HotPathEngine(consts, m, 1, c2t(), zn, history=2)
        eng.set_layout(layout)
        eng.load_state(0, xt, x, P)
        mirror = torch.zeros(m * 12, dtype=torch.float64).pin_memory()
        aer = torch.zeros(m * 4, dtype=torch.float64).pin_memory()
        st = torch.zeros(8, dtype=torch.float64).pin_memory()
        upd = torch.zeros(hip.lib.UPD_STRIDE, dtype=torch.float64).pin_memory()
        out = []
        init_stats = eng.stats[0].cpu().numpy().copy()
        for k, a in enumerate(acts):
            t = k + 1
            eng.launch_step((t - 1) % 2, t % 2, t, action=int(a), obs_mirror=mirror.data_ptr() if k % 2 == 0 else 0,
                            aer_out=aer.data_ptr() if k % 2 == 1 else 0, stats_out=st.data_ptr(), upd_out=upd.data_ptr(),
                            fast_stats=True, fold_inside=True, argmax_spos=True)
            torch.cuda.synchronize()
            out.append((mirror.numpy().copy(), aer.numpy().copy(), st.numpy().copy(), upd.numpy().copy()))
        # ... and five more steps as ONE rollout launch (ssa_env_rollout_f64 keeps the layout: actions, failure records and the per-step
        # arg-max slots speak the caller's indices there too)
        ra = torch.as_tensor(((np.arange(5) * 31 + 11) % m).astype(np.int32)).cuda().view(5, 1)
        eng.launch_rollout(n_steps % 2, n_steps + 1, ra, argmax_spos=True)
        torch.cuda.synchronize()
        out.append((eng.stats.cpu().numpy().copy(), eng.upd.cpu().numpy().copy(), np.zeros(1), np.zeros(1)))
        nf = int(eng.fail_count.item())
        fails = sorted((int(r[hip.lib.FAIL_OBJ]), int(r[hip.lib.FAIL_STATUS]), int(r[hip.lib.FAIL_TIME]), tuple(r[hip.lib.FAIL_ERR:hip.lib.FAIL_ERR + 4]))
                       for r in eng.fail_log[:nf])
        assert (eng._order is None) == (layout is None)
        eng.to_caller_order()
        assert eng._order is None
        s = (n_steps + 5) % 2
        state = (eng.x_true[s].cpu().numpy(), eng.x_filter[s].cpu().numpy(), eng.P_filter[s].cpu().numpy(), eng.obs[s].cpu().numpy(),
                 eng.metrics[s].cpu().numpy(), eng.status.cpu().numpy())
        return init_stats, out, fails, state
    a = run(None)
    b = run(order)
    assert np.array_equal(a[0], b[0], equal_nan=True)                     # the reset's statistics (first maximum in the caller's order)
    for k, (ua, ub) in enumerate(zip(a[1], b[1])):
        for name, va, vb in zip(("obs rows", "aer block", "statistics", "update record"), ua, ub):
            assert np.array_equal(va, vb, equal_nan=True), (k, name)
    assert a[2] == b[2] and (prop != "hybrid" or len(a[2]) > 0)
    for name, va, vb in zip(("x_true", "x_filter", "P_filter", "obs", "metrics", "status"), a[3], b[3]):
        assert np.array_equal(va, vb, equal_nan=True), name
    if prop == "hybrid":     # (the failure path did run)
        assert int(a[1][-2][2][hip.lib.STAT_N_FAILED]) >= 2


@pytest.mark.parametrize("E,m", [(3, 52), (2, 10244)])
def test_storage_layout_with_several_envs(hip, E, m):
    """ssa_step_params.obj_ids with several envs (HotPathEngine.set_layout([n_env][n_obj])): one ARBITRARY permutation per env, indices within
    the env.  Per-env actions, statistics (np.argmax(sigma_pos) included), failure records, the 'aer' block and the second copy of the
    observation rows (row e m + the caller's index) must equal the engine without a layout step by step, bit for bit; one env's row is
    replaced in between (set_env_layout + load_env_state: a vector env's auto-reset); to_caller_order() restores every state tensor.
    3 x 52 objects: the one-tile launch; 2 x 10 244: the grid-stride one (with its fold kernel)."""
    torch = hip.torch
    xt, x, P, g = make_batch(E * m, seed=35)
    rs = np.random.RandomState(12)
    P[rs.uniform(size=E * m) < 0.33] *= 3e4
    x[5, 2] = np.nan                  # (a filter of env 0 and one of env 1 fail at their first predict: failure records, status words)
    x[m + 3, 1] = np.nan
    consts = hip.host.make_consts(g["Q"], g["R"], 1e-4, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"], propagator="hybrid")
    zn = torch.as_tensor(rs.normal(size=(E, 480, 1, 3)) * np.array([4.8e-6, 4.8e-6, 1e3])).cuda()
    orders = np.stack([rs.permutation(m) for _ in range(E)])
    new_order = rs.permutation(m)
    xt2, x2, P2, _ = make_batch(m, seed=36)
    n_steps = 24
    acts = rs.randint(m, size=(n_steps, E))

    def run(layout):
        eng = hip.engine.HotPathEngine(consts, m, E, c2t(), zn, history=2, zn_stride_env=480 * 3, zn_stride_time=3, zn_stride_obj=0)
        if layout:
            eng.set_layout(orders)
        eng.load_state(0, xt, x, P)
        mirror = torch.zeros(E * m * 12, dtype=torch.float64, device="cuda")
        aer = torch.zeros(E * m * 4, dtype=torch.float64, device="cuda")
        st = torch.zeros((E, 8), dtype=torch.float64).pin_memory()
        out = [eng.stats[0].cpu().numpy().copy()]
        tix = np.zeros(E, dtype=np.int64)
        for k in range(n_steps):
            t = k + 1
            if k == 11:      # env 1 starts over from another state, under another permutation
                if layout:
                    eng.set_env_layout(1, new_order)
                eng.load_env_state(k % 2, 1, xt2, x2, P2[0])
                tix[1] = 0
            tix += 1
            eng.launch_step((t - 1) % 2, t % 2, 0, aer_out=aer.data_ptr(), obs_mirror=mirror.data_ptr(), stats_out=st.data_ptr(),
                            fast_stats=True, fold_inside=True, env_words=(tix.tolist(), acts[k].tolist()), argmax_spos=True)
            torch.cuda.synchronize()
            out.append((mirror.cpu().numpy().copy(), aer.cpu().numpy().copy(), st.numpy().copy()))
        nf = int(eng.fail_count.item())
        fails = sorted((int(r[hip.lib.FAIL_ENV]), int(r[hip.lib.FAIL_OBJ]), int(r[hip.lib.FAIL_STATUS]), int(r[hip.lib.FAIL_TIME])) for r in eng.fail_log[:nf])
        eng.to_caller_order()
        s_ = n_steps % 2
        state = (eng.x_true[s_].cpu().numpy(), eng.x_filter[s_].cpu().numpy(), eng.P_filter[s_].cpu().numpy(), eng.obs[s_].cpu().numpy(),
                 eng.metrics[s_].cpu().numpy(), eng.status.cpu().numpy())
        return out, fails, state
    a, b = run(False), run(True)
    assert np.array_equal(a[0][0], b[0][0], equal_nan=True)
    for k in range(1, n_steps + 1):
        for name, va, vb in zip(("obs rows", "aer block", "statistics"), a[0][k], b[0][k]):
            assert np.array_equal(va, vb, equal_nan=True), (k, name)
    assert a[1] == b[1] and len(a[1]) > 0
    for name, va, vb in zip(("x_true", "x_filter", "P_filter", "obs", "metrics", "status"), a[2], b[2]):
        assert np.array_equal(va, vb, equal_nan=True), name
