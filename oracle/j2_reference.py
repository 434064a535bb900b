"""CPU reference for the J2 EXTENSION propagator (test infrastructure only).

The reference repo has no perturbed propagator on its hot path; its only numerical integrator is
fx_xyz_cowell (envs/dynamics.py:168-201): scipy DOP853, rtol 1e-11, atol 1e-12, with the perturbing
acceleration passed as `ad`.  This module configures exactly that integrator with the J2 acceleration of
poliastro.core.perturbations.J2_perturbation (the function the reference's ecosystem would pass as `ad`),
so the device RK4 stepper is validated against an independent high-order integration.  Parity with the
reference is "unpinned" by construction.
"""
import numpy as np
from scipy.integrate import DOP853, solve_ivp

MU = 398600441800000.0
J2_EARTH = 0.00108263
R_EQ_EARTH = 6378136.6


def j2_accel(r, j2=J2_EARTH, r_eq=R_EQ_EARTH, k=MU):
    rn = np.linalg.norm(r)
    factor = 1.5 * k * j2 * r_eq ** 2 / rn ** 5
    zz = 5.0 * r[2] ** 2 / rn ** 2
    return np.array([zz - 1, zz - 1, zz - 3]) * r * factor


def fx_xyz_cowell_j2(x, dt, j2=J2_EARTH, r_eq=R_EQ_EARTH, rtol=1e-11):
    def f(t, u):
        r = u[:3]
        a = -MU * r / np.linalg.norm(r) ** 3 + j2_accel(r, j2, r_eq)
        return np.concatenate([u[3:], a])
    res = solve_ivp(f, (0, dt), np.asarray(x, dtype=float), rtol=rtol, atol=1e-12, method=DOP853, dense_output=True)
    if not res.success:
        raise RuntimeError("Integration failed")
    return res.sol(dt)
