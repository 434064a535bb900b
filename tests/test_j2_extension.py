"""J2 + RK4 EXTENSION propagator (SSA_PROP_J2_RK4): no reference counterpart, parity UNPINNED.
Validated against scipy DOP853 configured like the reference's unused fx_xyz_cowell (dynamics.py:184-193) with
poliastro's J2 acceleration, and against the parity-checked two-body path in the J2 = 0 limit."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    import ssa_gym_amd
    from ssa_gym_amd import _lib, device, host, engine
    ssa_gym_amd.build()
    assert torch.cuda.is_available()

    class H:
        pass
    h = H()
    h.torch, h.lib, h.dev, h.host, h.engine = torch, _lib, device, host, engine
    return h


def relnorm(a, b, sl):
    return np.linalg.norm((a - b)[..., sl], axis=-1) / np.linalg.norm(b[..., sl], axis=-1)


def test_rk4_j2_vs_dop853(hip):
    import j2_reference as J
    cat = golden("catalogue_subset.npy")
    x = cat[::7][:64]
    for dt, nsub in ((20.0, 4), (150.0, 30)):
        y = hip.dev.propagate_j2(hip.dev.as_dev(x), dt, J.J2_EARTH, J.R_EQ_EARTH, nsub).cpu().numpy()
        ref = np.array([J.fx_xyz_cowell_j2(xi, dt) for xi in x])
        assert relnorm(y, ref, slice(0, 3)).max() < 1e-10, dt      # RK4 at h = 5 s vs DOP853 rtol 1e-11
        assert relnorm(y, ref, slice(3, 6)).max() < 1e-9
        # the perturbation is actually there: LEO rows move by metres relative to two-body over 150 s
        kep = hip.dev.propagate(hip.dev.as_dev(x), dt).cpu().numpy()
        d = np.linalg.norm((y - kep)[:, :3], axis=1)
        leo = np.linalg.norm(x[:, :3], axis=1) < 9e6
        if leo.any() and dt > 100:
            assert d[leo].max() > 1.0


def test_two_body_limit_matches_farnocchia(hip):
    cat = golden("catalogue_subset.npy")
    x = hip.dev.as_dev(cat)
    y0 = hip.dev.propagate_j2(x, 20.0, 0.0, 6378136.6, 8).cpu().numpy()
    kep = hip.dev.propagate(x, 20.0).cpu().numpy()
    assert relnorm(y0, kep, slice(0, 3)).max() < 1e-11
    assert relnorm(y0, kep, slice(3, 6)).max() < 1e-10


def test_ukf_step_with_j2_propagator(hip):
    """the fused step with SSA_PROP_J2_RK4 against the restated UKF (oracle/ukf_numpy.py) driven by the DOP853 J2 fx."""
    import j2_reference as J
    import ukf_numpy as U
    import oracle as orc
    g = golden("ukf_step_golden.npz")
    c2t = golden("c2t_2020-05-04_dt20_n480.npy")
    m, alpha = 12, 1e-3
    xt, x0 = g["x_true"][:m], g["x0"][:m]
    P = np.tile(g["P0"], (m, 1, 1))
    consts = hip.host.make_consts(g["Q"], g["R"], alpha, 2.0, -3, 20.0, -np.pi / 2, g["obs_lla"], propagator='j2')
    assert consts.rk4_substeps == 4
    eng = hip.engine.HotPathEngine(consts, m, 1, c2t, np.zeros((1, 480, m, 3)), history=2)
    eng.load_state(0, xt, x0, P)
    eng.set_actions([-1])
    eng.launch_step(0, 1, 1)
    hip.torch.cuda.synchronize()
    xg, Pg, xtg = eng.x_filter[1].cpu().numpy(), eng.P_filter[1].cpu().numpy(), eng.x_true[1].cpu().numpy()
    o = orc.Oracle()
    pts = U.MerweScaledSigmaPoints(6, alpha, 2.0, -3, sqrt_method=lambda A: o.robust_cholesky(A)[0])
    for j in range(m):
        f = U.UnscentedKalmanFilter(6, 3, 20.0, None, J.fx_xyz_cowell_j2, pts)
        f.x, f.P, f.Q = x0[j].copy(), g["P0"].copy(), g["Q"].copy()
        f.predict()
        assert np.linalg.norm((xg[j] - f.x)[:3]) / np.linalg.norm(f.x[:3]) < 1e-6      # alpha = 1e-3: fp64 floor ~1e-9
        sd = np.sqrt(np.diag(f.P))
        assert np.max(np.abs(Pg[j] - f.P) / np.outer(sd, sd)) < 1e-5
        ref_t = J.fx_xyz_cowell_j2(xt[j], 20.0)
        assert np.linalg.norm(xtg[j] - ref_t) / np.linalg.norm(ref_t) < 1e-10
    assert np.all(eng.status.cpu().numpy() == 0)
