"""CPU pins of the product's per-lane device math: csrc/ssa_math.hpp compiled for the host (tests/hostmath: HIP qualifiers
shimmed away, hardware reciprocal estimates emulated with their 2^-26-class error) and held to the goldens the reference's
own farnocchia.py produced.  What this covers without a GPU: the universal-variable solvers of SSA_PROP_FG (series / Halley
and closed-form / Laguerre, incl. which lanes each one accepts), the strong-elliptic SSA_PROP_ELEMENTS chain with its
refined-estimate divisions, bounded-argument sincos and the exactness requirements of the equatorial test, the third-order
reciprocal refinements.  The GPU tests (-m gpu) then only have to show that the device executes the same arithmetic."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import golden

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hostmath") / "libhostmath.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D_GNU_SOURCE", "-fPIC", "-shared", "-ffp-contract=off",
                           "-I" + os.path.join(HERE, "hostmath"), "-o", so, os.path.join(HERE, "hostmath", "hostmath.cpp")])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _run(fn, x, dt, *pre):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    flag = np.zeros(len(x), dtype=np.int32)
    fn(_p(x), C.c_long(len(x)), C.c_double(dt), *pre, _p(out), _p(flag))
    return out, flag.astype(bool)


def relnorm(a, b, sl):
    return np.linalg.norm((a - b)[:, sl], axis=1) / np.linalg.norm(b[:, sl], axis=1)


@pytest.mark.parametrize("idt", range(5))
def test_fg_universal_solvers_vs_reference_golden(hm, idt):
    g = golden("kepler_golden.npz")
    dt = float(g["dts"][idt])
    y, ok = _run(hm.hm_propagate, g["x"], dt, C.c_int(1))
    assert ok.all()
    ref = g["y"][idt]
    inc = g["inter"][idt][:, 2]
    good = (inc > 1e-3) | (inc < 1e-8)       # (near-equatorial rows: the REFERENCE is ill-conditioned there, see test_hip_ops)
    assert relnorm(y, ref, slice(0, 3))[good].max() < 2e-12 and relnorm(y, ref, slice(3, 6))[good].max() < 2e-12
    assert relnorm(y, ref, slice(0, 3)).max() < 5e-10
    # which solver took which lane: every catalogue state at the env's step sizes is a series / Halley lane; at 5 400 s the
    # long-period orbits still are, at one day none is (closed-form / Laguerre for all)
    _, handled = _run(hm.hm_uv_fast, g["x"], dt)
    assert handled.all() if dt <= 150 else (0 < handled.sum() < len(handled) if dt < 8e4 else not handled.any())
    yg, okg = _run(hm.hm_uv_general, g["x"], dt)
    assert okg.all() and relnorm(yg, ref, slice(0, 3))[good].max() < 2e-12      # the general solver alone covers everything


def test_fg_hyperbolic_and_near_parabolic_states(hm, oracle_ld):
    """diverged filter states (scaled velocities): every conic through the same equation, against the 80-bit oracle"""
    cat = golden("catalogue_subset.npy")
    rs = np.random.RandomState(0)
    for lo, hi in ((1.45, 2.5), (1.40, 1.43), (1.0, 1.4), (0.3, 0.9), (3.0, 40.0)):
        x = cat.copy()
        x[:, 3:] *= rs.uniform(lo, hi, size=len(x))[:, None]
        for dt in (20.0, 150.0):
            y, ok = _run(hm.hm_propagate, x, dt, C.c_int(1))
            ref = oracle_ld.propagate(x, dt)
            fin = np.isfinite(ref).all(1)
            assert ok[fin].all()
            assert relnorm(y, ref, slice(0, 3))[fin].max() < 1e-12 and relnorm(y, ref, slice(3, 6))[fin].max() < 1e-12


@pytest.mark.parametrize("idt", range(3))
def test_elements_strong_elliptic_chain_vs_reference_golden(hm, idt):
    g = golden("kepler_golden.npz")
    y, ok = _run(hm.hm_propagate, g["x"], float(g["dts"][idt]), C.c_int(0))
    assert ok.all() and np.isfinite(y).all()        # incl. the exactly equatorial / circular rows: acos(h_z / |h|) must see exactly 1
    ref = g["y"][idt]
    inc = g["inter"][idt][:, 2]
    good = (inc > 1e-3) | (inc < 1e-8)
    assert relnorm(y, ref, slice(0, 3))[good].max() < 2e-12 and relnorm(y, ref, slice(3, 6))[good].max() < 2e-12
    assert relnorm(y, ref, slice(0, 3)).max() < 5e-10 and relnorm(y, ref, slice(3, 6)).max() < 2e-9


def test_fast_sincos_and_reciprocals(hm):
    rs = np.random.RandomState(1)
    x = np.concatenate([rs.uniform(-63.9, 63.9, 200000), [0.0, np.pi / 2, -np.pi, np.pi, 2 * np.pi, 1e-300, 100.0, -1e6]])
    s, c = np.empty_like(x), np.empty_like(x)
    hm.hm_sincos_fast(_p(x), C.c_long(len(x)), _p(s), _p(c))
    ls, lc = np.sin(x.astype(np.longdouble)), np.cos(x.astype(np.longdouble))
    # 1.5 ulp for results of ordinary size; next to a zero of sin / cos the two-part reduction leaves an ABSOLUTE error of
    # ~1e-17 (libm reduces further): the element chain only multiplies these by O(1) quantities
    es = np.abs((s - ls).astype(np.float64)) / np.spacing(np.maximum(np.abs(ls.astype(np.float64)), 1e-2))
    ec = np.abs((c - lc).astype(np.float64)) / np.spacing(np.maximum(np.abs(lc.astype(np.float64)), 1e-2))
    assert es[:-2].max() < 2.0 and ec[:-2].max() < 2.0, (es.max(), ec.max())
    assert np.abs((s - ls).astype(np.float64))[-2:].max() < 1e-15          # |x| >= 64: the libm branch
    v = 10.0 ** rs.uniform(-8, 20, 100000)
    r, q = np.empty_like(v), np.empty_like(v)
    hm.hm_recip(_p(v), C.c_long(len(v)), _p(r), _p(q))
    assert np.abs(r * v - 1).max() < 4.5e-16 and np.abs(q * q * v - 1).max() < 9e-16     # from a 1e-8 estimate: third order


def test_fast_atan2(hm):
    """atan2_fast (azimuth, and elevation as atan2(u, hypot(e, n))): error < 1.5 ulp (6e-16 at pi) over every octant, the fold points
    and the axes; atan2(0, 0) = 0 and the sign conventions of libm (azimuth wraps to [0, 2 pi) from those)."""
    rs = np.random.RandomState(2)
    ang = rs.uniform(-np.pi, np.pi, 300000)
    rad = 10.0 ** rs.uniform(-3, 8, len(ang))
    y = np.concatenate([rad * np.sin(ang), [0.0, 0.0, 0.0, 1.0, -1.0, 1.0, 1.0, -1.0, np.tan(np.pi / 8), 1e-300, 3.0]])
    x = np.concatenate([rad * np.cos(ang), [0.0, 1.0, -1.0, 0.0, 0.0, 1.0, -1.0, -1.0, 1.0, 1.0, 1e300]])
    r = np.empty_like(x)
    hm.hm_atan2_fast(_p(y), _p(x), C.c_long(len(x)), _p(r))
    ref = np.arctan2(y.astype(np.longdouble), x.astype(np.longdouble))
    err = np.abs((r - ref).astype(np.float64))
    assert err.max() < 6e-16 and (err / np.spacing(np.maximum(np.abs(r), 0.5))).max() < 1.5, (err.max(), np.argmax(err))
    small = np.abs(ref) < 0.3
    assert (err[small] / np.maximum(np.abs(ref[small]).astype(np.float64), 1e-300)).max() < 5e-16   # relative where the angle is small
    assert r[-11] == 0.0 and r[-10] == 0.0 and r[-9] == np.pi and r[-8] == np.pi / 2 and r[-7] == -np.pi / 2
    # elevation: asin(u / r) == atan2(u, hypot(e, n))
    e, n, u = rs.normal(size=(3, 100000)) * 1e6
    el = np.empty_like(u)
    hm.hm_atan2_fast(_p(u), _p(np.hypot(e, n)), C.c_long(len(u)), _p(el))
    el_, nl, ul = (v.astype(np.longdouble) for v in (e, n, u))
    ref = np.arcsin(ul / np.sqrt(el_ * el_ + nl * nl + ul * ul))      # (in fp64 asin(u / r) itself loses digits towards the zenith)
    assert np.abs((el - ref).astype(np.float64)).max() < 4e-16


def test_fast_exp(hm):
    """exp_fast on [0, 700) (the hyperbolic Stumpff functions of the general solver): < 2 ulp."""
    rs = np.random.RandomState(3)
    x = np.concatenate([rs.uniform(0.0, 700.0, 200000), rs.uniform(0.0, 2.0, 50000), [0.0, 0.5, np.log(2.0) / 2, 699.999]])
    r = np.empty_like(x)
    hm.hm_exp_fast(_p(x), C.c_long(len(x)), _p(r))
    ref = np.exp(x.astype(np.longdouble))
    assert (np.abs((r - ref) / ref).astype(np.float64)).max() < 4.5e-16
