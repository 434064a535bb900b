// ssa_conics.hpp -- the conic branches of farnocchia() beyond the strong-elliptic one, per lane (gfx950; also compiled for the
// host by tests/hostmath, where the fast forms are pinned against the libm-level restatement and the oracle).
//   gen::   complete restatement of farnocchia() with libm (every branch, NaN by NaN)        farnocchia.py:165-313, 847-1006
//   genf::  the same branches with this library's fast primitives (SSA_PROP_HYBRID, and the fallback of SSA_PROP_ELEMENTS)
#pragma once
#include "ssa_math.hpp"

namespace ssa {

// ------------------------------------------------------------------------------------------
// Complete farnocchia() restatement (all conic branches).  Scalar, out of line.
// farnocchia.py:165-313 (rv2coe), :847-921 (delta_t_from_nu), :925-1006 (nu_from_delta_t),
// :337-353 (newton), :692-843 (near-parabolic series), :101-161 (coe2rv).
namespace gen {
__device__ static double pymod(double a, double b)
{
    double m = fmod(a, b);
    if (m != 0.0 && ((m < 0.0) != (b < 0.0))) m += b;
    return m;
}
__device__ static double newton(bool hyper, double x0, double M, double ecc, int maxiter)
{
    double p0 = x0;
    for (int i = 0; i < maxiter; ++i) {
        double fval, fder;
        if (hyper) {
            fval = (ecc * sinh(p0) - p0) - M;
            fder = ecc * cosh(p0) - 1.0;
        } else {
            fval = (p0 - ecc * sin(p0)) - M;
            fder = 1.0 - ecc * cos(p0);
        }
        double p = p0 - fval / fder;
        if (fabs(p - p0) < NEWTON_TOL) return p;
        p0 = p;
    }
    return __builtin_nan("");
}
// S_x / dS_x_alt (farnocchia.py:692-760) sum (ecc - 1/(2k+3)) [(2k+3)] x^k until the term drops below
// 1e-12 -- tens of thousands of terms as |x| -> 1.  Both series have closed forms, used here
// (they agree with the truncated sums to the truncation tolerance):
//   sum x^k/(2k+3) = (A(x) - 1)/x,  A = atanh(sqrt x)/sqrt x (x>0) | atan(sqrt -x)/sqrt -x (x<0)
//   sum (2k+3) x^k = 2x/(1-x)^2 + 3/(1-x)
__device__ static double S_x(double ecc, double x, bool alt)
{
    if (!(fabs(x) < 1.0)) return __builtin_nan("");   // the reference asserts abs(x) < 1
    const double omx = 1.0 - x;
    if (alt) return ecc * (2.0 * x / (omx * omx) + 3.0 / omx) - 1.0 / omx;
    if (fabs(x) < 0.05) {   // short series; the closed form cancels as x -> 0
        double S = 0.0, xk = 1.0;
        for (int k = 0; k < 12; ++k) { S += (ecc - 1.0 / (2 * k + 3)) * xk; xk *= x; }
        return S;
    }
    const double sx = sqrt(fabs(x));
    const double A = (x > 0.0) ? atanh(sx) / sx : atan(sx) / sx;
    return ecc / omx - (A - 1.0) / x;
}
__device__ static double D_to_M_np(double D, double ecc)
{
    double x = (ecc - 1.0) / (ecc + 1.0) * (D * D);
    double S = S_x(ecc, x, false);
    double ope = 1.0 + ecc;
    return sqrt(2.0 / ope) * D + sqrt(2.0 / (ope * ope * ope)) * (D * D * D) * S;
}
__device__ static double M_to_D(double M)
{
    double B = 3.0 * M / 2.0;
    double A = pow(B + sqrt(1.0 + B * B), 2.0 / 3.0);
    return 2.0 * A * B / (1.0 + A + A * A);
}
__device__ static double M_to_D_np(double M, double ecc)
{
    double D0 = M_to_D(M);
    double ope = 1.0 + ecc;
    for (int i = 0; i < 50; ++i) {
        double fval = D_to_M_np(D0, ecc) - M;
        double x = (ecc - 1.0) / ope * (D0 * D0);
        double S = S_x(ecc, x, true);
        double fder = sqrt(2.0 / ope) + sqrt(2.0 / (ope * ope * ope)) * (D0 * D0) * S;
        double D = D0 - fval / fder;
        if (fabs(D - D0) < NEWTON_TOL) return D;
        D0 = D;
    }
    return __builtin_nan("");
}
__device__ static double E_to_nu(double E, double ecc) { return 2.0 * atan(sqrt((1.0 + ecc) / (1.0 - ecc)) * tan(E / 2.0)); }
__device__ static double nu_to_E(double nu, double ecc) { return 2.0 * atan(sqrt((1.0 - ecc) / (1.0 + ecc)) * tan(nu / 2.0)); }
__device__ static double F_to_nu(double F, double ecc) { return 2.0 * atan(sqrt((ecc + 1.0) / (ecc - 1.0)) * tanh(F / 2.0)); }
__device__ static double nu_to_F(double nu, double ecc) { return 2.0 * atanh(sqrt((ecc - 1.0) / (ecc + 1.0)) * tan(nu / 2.0)); }

__device__ __forceinline__ static double delta_t_from_nu(double nu, double ecc, double k, double q)
{
    const double delta = 1e-2;
    double M, n;
    double q3 = q * q * q;
    if (ecc < 1.0 - delta) {
        double E = nu_to_E(nu, ecc);
        M = E - ecc * sin(E);
        n = sqrt(k * (1.0 - ecc) * (1.0 - ecc) * (1.0 - ecc) / q3);
    } else if (1.0 - delta <= ecc && ecc < 1.0) {
        double E = nu_to_E(nu, ecc);
        if (delta <= 1.0 - ecc * cos(E)) {
            M = E - ecc * sin(E);
            n = sqrt(k * (1.0 - ecc) * (1.0 - ecc) * (1.0 - ecc) / q3);
        } else {
            M = D_to_M_np(tan(nu / 2.0), ecc);
            n = sqrt(k / (2.0 * q3));
        }
    } else if (ecc == 1.0) {
        double D = tan(nu / 2.0);
        M = D + D * D * D / 3.0;
        n = sqrt(k / (2.0 * q3));
    } else if (1.0 + ecc * cos(nu) < 0.0) {
        return __builtin_nan("");
    } else if (1.0 < ecc && ecc <= 1.0 + delta) {
        double F = nu_to_F(nu, ecc);
        if (delta <= ecc * cosh(F) - 1.0) {
            M = ecc * sinh(F) - F;
            n = sqrt(k * (ecc - 1.0) * (ecc - 1.0) * (ecc - 1.0) / q3);
        } else {
            M = D_to_M_np(tan(nu / 2.0), ecc);
            n = sqrt(k / (2.0 * q3));
        }
    } else if (1.0 + delta < ecc) {
        double F = nu_to_F(nu, ecc);
        M = ecc * sinh(F) - F;
        n = sqrt(k * (ecc - 1.0) * (ecc - 1.0) * (ecc - 1.0) / q3);
    } else {
        return __builtin_nan("");
    }
    return M / n;
}
__device__ static double M_to_E(double M, double ecc)
{
    double E0 = (ecc < 0.8) ? M : PI * ((M > 0.0) - (M < 0.0));
    return newton(false, E0, M, ecc, 50);
}
__device__ __forceinline__ static double nu_from_delta_t(double delta_t, double ecc, double k, double q)
{
    const double delta = 1e-2;
    double q3 = q * q * q;
    if (ecc < 1.0 - delta) {
        double n = sqrt(k * (1.0 - ecc) * (1.0 - ecc) * (1.0 - ecc) / q3);
        double M = n * delta_t;
        return E_to_nu(M_to_E(pymod(M + PI, TWO_PI) - PI, ecc), ecc);
    } else if (1.0 - delta <= ecc && ecc < 1.0) {
        double E_delta = acos((1.0 - delta) / ecc);
        double n = sqrt(k * (1.0 - ecc) * (1.0 - ecc) * (1.0 - ecc) / q3);
        double M = n * delta_t;
        if (E_delta - ecc * sin(E_delta) <= fabs(M))
            return E_to_nu(M_to_E(pymod(M + PI, TWO_PI) - PI, ecc), ecc);
        n = sqrt(k / (2.0 * q3));
        return 2.0 * atan(M_to_D_np(n * delta_t, ecc));
    } else if (ecc == 1.0) {
        double n = sqrt(k / (2.0 * q3));
        return 2.0 * atan(M_to_D(n * delta_t));
    } else if (1.0 < ecc && ecc <= 1.0 + delta) {
        double F_delta = acosh((1.0 + delta) / ecc);
        double n = sqrt(k * (ecc - 1.0) * (ecc - 1.0) * (ecc - 1.0) / q3);
        double M = n * delta_t;
        if (ecc * sinh(F_delta) - F_delta <= fabs(M))
            return F_to_nu(newton(true, asinh(M / ecc), M, ecc, 100), ecc);
        n = sqrt(k / (2.0 * q3));
        return 2.0 * atan(M_to_D_np(n * delta_t, ecc));
    } else {
        double n = sqrt(k * (ecc - 1.0) * (ecc - 1.0) * (ecc - 1.0) / q3);
        double M = n * delta_t;
        return F_to_nu(newton(true, asinh(M / ecc), M, ecc, 100), ecc);
    }
}
}  // namespace gen

__device__ static void kepler_general_impl(const double* x, double tof, double* out, double* diag)
{
    const double tol = 1e-8;
    const double* r = x;
    const double* v = x + 3;
    double h[3] = {r[1] * v[2] - r[2] * v[1], r[2] * v[0] - r[0] * v[2], r[0] * v[1] - r[1] * v[0]};
    double n[3] = {-h[1], h[0], 0.0};
    double rn = sqrt(dot3(r, r)), vv = dot3(v, v), rv = dot3(r, v);
    double c1 = vv - MU / rn;
    double e[3] = {(c1 * r[0] - rv * v[0]) / MU, (c1 * r[1] - rv * v[1]) / MU, (c1 * r[2] - rv * v[2]) / MU};
    double ecc = sqrt(dot3(e, e));
    double p = dot3(h, h) / MU;
    double hn = sqrt(dot3(h, h));
    double inc = acos(h[2] / hn);
    bool circular = ecc < tol, equatorial = fabs(inc) < tol;
    double raan, argp, nu;
    if (equatorial && !circular) {
        raan = 0.0;
        argp = gen::pymod(atan2(e[1], e[0]), TWO_PI);
        double t[3] = {e[1] * r[2] - e[2] * r[1], e[2] * r[0] - e[0] * r[2], e[0] * r[1] - e[1] * r[0]};
        nu = atan2(dot3(h, t) / hn, dot3(r, e));
    } else if (!equatorial && circular) {
        raan = gen::pymod(atan2(n[1], n[0]), TWO_PI);
        argp = 0.0;
        double t[3] = {h[1] * n[2] - h[2] * n[1], h[2] * n[0] - h[0] * n[2], h[0] * n[1] - h[1] * n[0]};
        nu = atan2(dot3(r, t) / hn, dot3(r, n));
    } else if (equatorial && circular) {
        raan = 0.0;
        argp = 0.0;
        nu = gen::pymod(atan2(r[1], r[0]), TWO_PI);
    } else {
        double a = p / (1.0 - ecc * ecc);
        double ka = MU * a;
        if (a > 0.0) {
            double e_se = rv / sqrt(ka);
            double e_ce = rn * vv / MU - 1.0;
            nu = gen::E_to_nu(atan2(e_se, e_ce), ecc);
        } else {
            double e_sh = rv / sqrt(-ka);
            double e_ch = rn * vv / MU - 1.0;
            nu = gen::F_to_nu(log((e_ch + e_sh) / (e_ch - e_sh)) / 2.0, ecc);
        }
        raan = gen::pymod(atan2(n[1], n[0]), TWO_PI);
        double t[3] = {h[1] * n[2] - h[2] * n[1], h[2] * n[0] - h[0] * n[2], h[0] * n[1] - h[1] * n[0]};
        double px = dot3(r, n), py = dot3(r, t) / hn;
        argp = gen::pymod(atan2(py, px) - nu, TWO_PI);
    }
    nu = gen::pymod(nu + PI, TWO_PI) - PI;
    double q = p / (1.0 + ecc);
    double dt0 = gen::delta_t_from_nu(nu, ecc, MU, q);
    double nu1 = gen::nu_from_delta_t(dt0 + tof, ecc, MU, q);
    coe2rv(p, ecc, inc, raan, argp, nu1, out);
    if (diag) {
        diag[0] = p; diag[1] = ecc; diag[2] = inc; diag[3] = raan; diag[4] = argp; diag[5] = nu;
        diag[6] = dt0; diag[7] = nu1;
    }
}

// ------------------------------------------------------------------------------------------
// SSA_PROP_HYBRID: the reference's BRANCHES where they matter, at a fraction of their cost.
// What makes the reference lose filters late in an episode is (a) the cancellation in its covariance sum
// (SSA_FLAG_REFERENCE_COV) acting on (b) priors that have left the strong-elliptic regime, which farnocchia() then propagates
// through its hyperbolic / near-parabolic formulas -- tens of metres off for such states.  On strong-elliptic
// states its chain and the universal-variable solver agree to 1e-14, so there the hybrid runs the series solver of
// SSA_PROP_FG; every other sigma point goes through the reference's formulas, branch by branch and NaN by NaN as
// kepler_general_impl above, but with this file's fast primitives (atan2_fast, sincos_fast, one exponential for sinh AND
// cosh in the hyperbolic Newton loop) instead of libm -- the complete restatement costs ~3 000 vector instructions per
// call, four times the whole SSA_PROP_FG step, and late in an episode most wavefronts hold a diverged sigma point.
// Episode-level failure statistics: as SSA_PROP_ELEMENTS / the oracle (tests/test_episode_failures.py).
#ifdef SSA_TRACE   // (diagnostic build: which branch of the out-of-line propagation a workgroup's lanes took, and their longest Newton run)
__device__ unsigned g_kep_dbg[16384 * 2];
#define SSA_KEP_DBG_BRANCH(b) atomicOr(&g_kep_dbg[(blockIdx.x & 16383) * 2], (unsigned)(b))
#define SSA_KEP_DBG_ITERS(n) atomicMax(&g_kep_dbg[(blockIdx.x & 16383) * 2 + 1], (unsigned)(n))
#else
#define SSA_KEP_DBG_BRANCH(b) do { } while (0)
#define SSA_KEP_DBG_ITERS(n) do { } while (0)
#endif
namespace genf {
// log x for finite x > 0 (the arguments of this path: ratios and sums of positive magnitudes; anything else takes libm):
// x = m 2^k with m in [sqrt(1/2), sqrt 2), log m = 2 atanh(s), s = (m - 1)/(m + 1), by the fdlibm kernel polynomial; < 1 ulp,
// ~35 instructions (libm's log: ~80)
__device__ static double log_pos(double x)
{
    // (total without libm: log of a negative number or NaN is NaN, of zero -inf, of +inf +inf; denormals are scaled into range)
    const bool tiny = x < 2.2250738585072014e-308;
    const double xs = tiny ? x * 18446744073709551616.0 : x;            // 2^64
    int k;
    double m = frexp(xs, &k);                // m in [0.5, 1)
    if (m < 0.70710678118654752440) { m += m; k -= 1; }
    if (tiny) k -= 64;
    const double f = m - 1.0;
    const double sq = div_fast(f, 2.0 + f);
    const double z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    double res = dk * 6.93147180369123816490e-01 - ((hfsq - (sq * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
    if (!(x > 0.0)) res = (x == 0.0) ? -__builtin_inf() : __builtin_nan("");
    if (x > 1.79769313486231570e308) res = x;
    return res;
}
__device__ static double F_to_nu(double F, double ecc)
{
    double sh, chm1;
    sinh_coshm1(F, sh, chm1);
    return 2.0 * atan2_fast(sqrt_fast(div_fast(ecc + 1.0, ecc - 1.0)) * div_fast(sh, chm1 + 2.0), 1.0);   // tanh(F/2) = sinh F / (cosh F + 1)
}
__device__ static double nu_to_F(double nu, double ecc)
{
    double s, c;
    sincos_fast(nu, s, c);
    const double x = sqrt_fast(div_fast(ecc - 1.0, ecc + 1.0)) * div_fast(s, 1.0 + c);     // tan(nu/2) = sin nu / (1 + cos nu)
    return log_pos(div_fast(1.0 + x, 1.0 - x));                                              // 2 atanh(x)
}
// newton() on e sinh F - F - M (farnocchia.py:337-353: step tolerance 1.48e-8, 100 iterations, NaN when it gives up)
__device__ static double newton_hyp(double x0, double M, double ecc)
{
    double p0 = x0, res = __builtin_nan("");
    bool done = false;
    double sh, chm1;
    sinh_coshm1(p0, sh, chm1);
    for (int i = 0; i < 100; ++i) {
        const double fval = (ecc * sh - p0) - M;
        const double fder = ecc * (chm1 + 1.0) - 1.0;
        const double p = p0 - div_fast(fval, fder);
        const double d = p - p0;
        if (!done && fabs(d) < NEWTON_TOL) { res = p; done = true; }
        p0 = p;
        if (!(fabs(p0) <= 1.79769313486231570e308)) done = true;      // (inf / NaN iterate: it will never converge)
        if (__ballot(!done) == 0ull) { SSA_KEP_DBG_ITERS(i + 1); break; }
        if (i == 99) SSA_KEP_DBG_ITERS(100);
        // sinh / cosh of the new iterate: by the addition formulas where the lane's step is small (all steps after the first or
        // second: the exponential of a fresh evaluation is four times as long), afresh otherwise.  The choice is the LANE's own
        // (round 3 took a wavefront-wide vote: a lane's rounding then depended on who shared its wavefront); the fresh
        // evaluation is skipped when no live lane wants it.
        const bool big = !(fabs(d) <= 0.02);
        const double d2 = d * d;
        const double sd = d * (1.0 + d2 * (1.0 / 6.0) * (1.0 + d2 * (1.0 / 20.0) * (1.0 + d2 * (1.0 / 42.0))));
        const double cdm1 = d2 * 0.5 * (1.0 + d2 * (1.0 / 12.0) * (1.0 + d2 * (1.0 / 30.0) * (1.0 + d2 * (1.0 / 56.0))));
        const double sh2 = fma(sh, cdm1, sh) + (chm1 + 1.0) * sd;
        const double ch2 = fma(chm1, cdm1, chm1) + cdm1 + sh * sd;
        sh = sh2;
        chm1 = ch2;
        if (__ballot(!done && big) != 0ull) {
            double shf, chf;
            sinh_coshm1(p0, shf, chf);
            if (big) { sh = shf; chm1 = chf; }
        }
    }
    return res;
}
}  // namespace genf
// ---- the rare bands of farnocchia() -- near-parabolic (|ecc - 1| <= 1e-2), parabolic, elliptic beyond the series solver's
// domain -- with the fast primitives.  gen::delta_t_from_nu / gen::nu_from_delta_t above are the libm-level restatement of
// the same branches; one wavefront in a hundred holds such a sigma point late in an episode and, with libm, ran 13-19 us in
// them and ended the launch (profiles/r04_wave_timeline_hybrid_step400_before.txt).  Branch by branch and NaN by NaN as the
// restatement; every loop leaves when the live lanes are done and every decision is the lane's own.  Pinned on the host
// against gen:: and the 80-bit oracle (tests/test_device_math_host.py).
namespace genf {
// tan(x / 2) for |x| <= pi
SSA_DEV double tan_half(double x)
{
    double s, c;
    sincos_small(0.5 * x, s, c);
    return div_fast(s, c);
}
SSA_DEV double atanh_pos(double s) { return 0.5 * log_pos(div_fast(1.0 + s, 1.0 - s)); }      // 0 <= s < 1
SSA_DEV double asinh_fast(double x)
{
    const double ax = fabs(x);
    return copysign(log_pos(ax + sqrt_fast(ax * ax + 1.0)), x);
}
// S_x / dS_x_alt (farnocchia.py:692-760) in closed form, as gen::S_x
SSA_DEV double S_x(double ecc, double x, bool alt)
{
    if (!(fabs(x) < 1.0)) return __builtin_nan("");   // the reference asserts abs(x) < 1
    const double iomx = rcp_nr(1.0 - x);
    if (alt) return ecc * (2.0 * x * iomx * iomx + 3.0 * iomx) - iomx;
    double S;
    if (fabs(x) < 0.05) {   // short series; the closed form cancels as x -> 0
        S = 0.0;
        double xk = 1.0;
#pragma unroll
        for (int k = 0; k < 12; ++k) { S += (ecc - 1.0 / (2 * k + 3)) * xk; xk *= x; }
    } else {
        const double sx = sqrt_fast(fabs(x));
        const double A = (x > 0.0) ? div_fast(atanh_pos(sx), sx) : div_fast(atan2_fast(sx, 1.0), sx);
        S = ecc * iomx - div_fast(A - 1.0, x);
    }
    return S;
}
SSA_DEV double D_to_M_np(double D, double ecc)
{
    const double ope = 1.0 + ecc;
    const double x = div_fast(ecc - 1.0, ope) * (D * D);
    const double S = S_x(ecc, x, false);
    const double t = div_fast(2.0, ope);
    return sqrt_fast(t) * D + sqrt_fast(div_fast(t, ope * ope)) * (D * D * D) * S;
}
SSA_DEV double M_to_D(double M)      // Barker's equation (farnocchia.py:605-627): A = (B + sqrt(1 + B^2))^(2/3) = exp(2/3 asinh B)
{
    const double B = 1.5 * M;
    const double t = (2.0 / 3.0) * asinh_fast(B);
    const double at = fabs(t);
    const double ea = (at < 709.0) ? exp_fast(at) : ((at == at) ? __builtin_inf() : at);
    const double A = (t >= 0.0) ? ea : rcp_nr(ea);
    return div_fast(2.0 * A * B, 1.0 + A + A * A);
}
SSA_DEV double M_to_D_np(double M, double ecc)      // newton() on D_to_M_np (farnocchia.py:337-353: 50 iterations, NaN when it gives up)
{
    double D0 = M_to_D(M), res = __builtin_nan("");
    bool done = false;
    const double ope = 1.0 + ecc;
    const double t = div_fast(2.0, ope), k1 = sqrt_fast(t), k3 = sqrt_fast(div_fast(t, ope * ope)), kx = div_fast(ecc - 1.0, ope);
    for (int i = 0; i < 50; ++i) {
        const double fval = D_to_M_np(D0, ecc) - M;
        const double S = S_x(ecc, kx * (D0 * D0), true);
        const double fder = k1 + k3 * (D0 * D0) * S;
        const double D = D0 - div_fast(fval, fder);
        if (!done && fabs(D - D0) < NEWTON_TOL) { res = D; done = true; }
        D0 = D;
        if (!(fabs(D0) <= 1.79769313486231570e308)) done = true;
        if (__ballot(!done) == 0ull) break;
    }
    return res;
}
SSA_DEV double E_to_nu(double E, double ecc) { return 2.0 * atan2_fast(sqrt_fast(div_fast(1.0 + ecc, 1.0 - ecc)) * tan_half(E), 1.0); }
SSA_DEV double nu_to_E(double nu, double ecc) { return 2.0 * atan2_fast(sqrt_fast(div_fast(1.0 - ecc, 1.0 + ecc)) * tan_half(nu), 1.0); }
// delta_t_from_nu (farnocchia.py:847-921) for ecc <= 1 + delta (the strong-hyperbolic branch is the caller's)
SSA_DEV double delta_t_from_nu_band(double nu, double ecc, double q)
{
    const double delta = 1e-2;
    const double q3 = q * q * q;
    double M = __builtin_nan(""), n = 1.0;
    if (ecc < 1.0) {
        const double E = nu_to_E(nu, ecc);
        double sE, cE;
        sincos_small(E, sE, cE);
        if (ecc < 1.0 - delta || delta <= 1.0 - ecc * cE) {
            M = E - ecc * sE;
            const double ome = 1.0 - ecc;
            n = sqrt_fast(div_fast(MU * ome * ome * ome, q3));
        } else {
            M = D_to_M_np(tan_half(nu), ecc);
            n = sqrt_fast(div_fast(MU, 2.0 * q3));
        }
    } else if (ecc == 1.0) {
        const double D = tan_half(nu);
        M = D + D * D * D * (1.0 / 3.0);
        n = sqrt_fast(div_fast(MU, 2.0 * q3));
    } else if (ecc <= 1.0 + delta) {
        double sn, cn;
        sincos_small(nu, sn, cn);
        if (1.0 + ecc * cn < 0.0) return __builtin_nan("");            // (:885-888: beyond the asymptote)
        const double F = nu_to_F(nu, ecc);
        double sh, chm1;
        sinh_coshm1(F, sh, chm1);
        if (delta <= ecc * (chm1 + 1.0) - 1.0) {
            M = ecc * sh - F;
            const double em1 = ecc - 1.0;
            n = sqrt_fast(div_fast(MU * em1 * em1 * em1, q3));
        } else {
            M = D_to_M_np(tan_half(nu), ecc);
            n = sqrt_fast(div_fast(MU, 2.0 * q3));
        }
    }       // (NaN ecc: every comparison false -> NaN, as the restatement's last branch)
    return div_fast(M, n);
}
// nu_from_delta_t (farnocchia.py:925-1006) for ecc <= 1 + delta
SSA_DEV double nu_from_delta_t_band(double delta_t, double ecc, double q)
{
    const double delta = 1e-2;
    const double q3 = q * q * q;
    const double n_par = sqrt_fast(div_fast(MU, 2.0 * q3));
    double nu = __builtin_nan("");
    if (ecc < 1.0) {
        const double ome = 1.0 - ecc;
        const double n = sqrt_fast(div_fast(MU * ome * ome * ome, q3));
        const double M = n * delta_t;
        bool elliptic = ecc < 1.0 - delta;
        if (!elliptic) {
            // E_delta = acos((1 - delta) / ecc); its sine from (1 - c)(1 + c)
            const double c = div_fast(1.0 - delta, ecc);
            const double sE = sqrt_fast((1.0 - c) * (1.0 + c));
            const double E_delta = atan2_fast(sE, c);
            elliptic = E_delta - ecc * sE <= fabs(M);
        }
        if (elliptic) nu = E_to_nu(solve_kepler_E(wrap_pi(M), ecc), ecc);
        else nu = 2.0 * atan2_fast(M_to_D_np(n_par * delta_t, ecc), 1.0);
    } else if (ecc == 1.0) {
        nu = 2.0 * atan2_fast(M_to_D(n_par * delta_t), 1.0);
    } else if (ecc <= 1.0 + delta) {
        const double em1 = ecc - 1.0;
        const double n = sqrt_fast(div_fast(MU * em1 * em1 * em1, q3));
        const double M = n * delta_t;
        // F_delta = acosh((1 + delta) / ecc) = log(y + sqrt((y - 1)(y + 1)))
        const double y = div_fast(1.0 + delta, ecc);
        const double sF = sqrt_fast((y - 1.0) * (y + 1.0));          // = sinh(F_delta)
        const double F_delta = log_pos(y + sF);
        const bool hyper = ecc * sF - F_delta <= fabs(M);
        if (hyper) nu = F_to_nu(newton_hyp(asinh_fast(div_fast(M, ecc)), M, ecc), ecc);
        else nu = 2.0 * atan2_fast(M_to_D_np(n_par * delta_t, ecc), 1.0);
    }
    return nu;
}
}  // namespace genf
SSA_DEV double kepler_band_nu_inl(double nu, double ecc, double q, double tof)
{
#ifdef SSA_BAND_LIBM   // diagnostic: the libm-level restatement (what round 3 shipped)
    const double dt0 = gen::delta_t_from_nu(nu, ecc, MU, q);
    return gen::nu_from_delta_t(dt0 + tof, ecc, MU, q);
#else
    const double dt0 = genf::delta_t_from_nu_band(nu, ecc, q);
    return genf::nu_from_delta_t_band(dt0 + tof, ecc, q);
#endif
}
// (as a call of their own for the complete restatement's special-orientation branches: inlined there the bands pushed the whole
// function past the step kernels' 96 registers)
template <int TAG>
__device__ __noinline__ double kepler_band_nu(double nu, double ecc, double q, double tof)
{
#ifdef SSA_BAND_LIBM   // diagnostic: the libm-level restatement (what round 3 shipped)
    const double dt0 = gen::delta_t_from_nu(nu, ecc, MU, q);
    return gen::nu_from_delta_t(dt0 + tof, ecc, MU, q);
#else
    const double dt0 = genf::delta_t_from_nu_band(nu, ecc, q);
    return genf::nu_from_delta_t_band(dt0 + tof, ecc, q);
#endif
}
// ---- the conic branches INLINE, for the general orientation (SSA_PROP_HYBRID's second tier; the fallback of SSA_PROP_ELEMENTS too).
// A filter that has diverged late in a predict-mostly episode lives in the strong-hyperbolic branch (ecc > 1 + 1e-2:
// farnocchia.py:909-912, :1001-1004), and by step 400 a quarter of the objects -- two thirds of the wavefronts -- hold such sigma
// points.  As an out-of-line call of the complete restatement (kepler_general_fast_impl below) that cost every such wavefront ~8 us:
// 24 registers saved to and restored from scratch memory around ~1 300 dependent instructions, a libm acos among them
// (profiles/r04_wave_timeline_hybrid_step400_before.txt).  This is the same ANOMALY chain, operation by operation -- rv2coe's F
// from (e sinh F, e cosh F), F -> nu, nu -> F again (delta_t_from_nu), M = e sinh F - F, the hyperbolic Newton solve, F -> nu,
// r = p / (1 + e cos nu): the round trips through the true anomaly whose conditioning for far-out states IS the reference's
// propagation error there; the near-parabolic bands and elliptic states beyond the series solver go through the
// band call (kepler_band_nu) from here -- but the ORIENTATION of the orbit is carried by the unit vectors r / |r| and
// (h x r) / (|h| |r|) of the state itself instead of the Euler angles acos(h_z / |h|), atan2(n_y, n_x), atan2(.) - nu and the four
// sincos that turn them back into a basis (coe2rv): the same frame to ~1e-16 (1e-6 m at r = 1e10 m, against the anomaly chain's
// metres to kilometres), a third of the instructions, no call on the hyperbolic branch, one call level less on the bands.
// Declines -- false: the caller takes the complete restatement -- rv2coe's special branches (circular, (near-)equatorial:
// farnocchia.py:278-309) and non-finite input.
// HYPER_ONLY: the instance inlined into the step kernels -- the strong-hyperbolic branch alone (no call, no spill); the other conics of
// the general orientation take this same function as an out-of-line call of its own (kepler_conic_lean_tagged in ssa_kernels.hip: the
// bands inline there), and only what THAT declines reaches the complete restatement (kepler_general_fast_impl).
template <int TAG, bool HYPER_ONLY>
SSA_DEV bool kepler_conic_lean(const double* x, double tof, double* out)
{
    const double* r = x;
    const double* v = x + 3;
    const double h[3] = {r[1] * v[2] - r[2] * v[1], r[2] * v[0] - r[0] * v[2], r[0] * v[1] - r[1] * v[0]};
    const double inv_mu = 1.0 / MU;
    const double rn = sqrt_fast(dot3(r, r)), vv = dot3(v, v), rv = dot3(r, v);
    const double c1 = vv - div_fast(MU, rn);
    const double e[3] = {(c1 * r[0] - rv * v[0]) * inv_mu, (c1 * r[1] - rv * v[1]) * inv_mu, (c1 * r[2] - rv * v[2]) * inv_mu};
    const double ee = dot3(e, e), hh = dot3(h, h);
    const double ecc = sqrt_fast(ee);
    const double p = hh * inv_mu;
    // rv2coe's general branch only: not circular (ecc >= 1e-8), not equatorial -- |inc| < 1e-8 means h_z / |h| rounds to exactly 1; a
    // margin of 4e-15 on the square keeps every such state (and the retrograde mirror) on the complete path
    const bool lean = (HYPER_ONLY ? (ecc > 1.0 + 1e-2) : (ecc >= 1e-8)) && (h[2] * h[2] < hh * (1.0 - 4e-15)) && (hh <= 1.79769313486231570e308) &&
                      (ecc <= 1.79769313486231570e308) && (rn > 0.0);
    if (!lean) return false;                 // (lane-divergent from here on: only the lanes that take this tier run its loops)
#ifndef SSA_HYBRID_REFERENCE_BANDS
    // TAG 1 = SSA_PROP_HYBRID: the BANDS -- near-parabolic |ecc - 1| <= 1e-2, parabolic, elliptic beyond the series solver's domain -- go
    // through the universal-variable solver (kepler_uv_general: SSA_PROP_FG's, 1e-15 on every conic) instead of the reference's
    // near-parabolic machinery (farnocchia.py:876-908, 975-1000).  What makes the hybrid behaviour-faithful is the strong-hyperbolic
    // chain below -- its F <-> nu round trips are the reference's propagation error on far-out states, and that error shapes the failure
    // statistics; in the bands the reference is itself accurate to its Newton tolerance, the two solvers agree to
    // ~1e-12, and the gate (tests/test_episode_failures.py) does not tell them apart.  But 2.7 % of the late-episode wavefronts hold such a
    // sigma point next to hyperbolic ones, ran this tier's hyperbolic branch AND ~1 500 dependent instructions of band arithmetic one after
    // the other, and ended every late launch 3 us behind the 99th percentile (profiles/r04_wave_timeline_hybrid_step330_layout.txt).
    // SSA_PROP_ELEMENTS (TAG 2) and the operator kernels (TAG 0) keep the reference's bands; -DSSA_HYBRID_REFERENCE_BANDS: the hybrid too.
    if (TAG == 1 && !HYPER_ONLY && !(ecc > 1.0 + 1e-2)) {
        SSA_KEP_DBG_BRANCH(ecc >= 1.0 - 1e-2 ? 2 : 4);
        if (!kepler_uv_general(x, tof, out)) {       // (no convergence in 16 iterations: NaN, as the reference's newton() gives up -- farnocchia.py:353)
#pragma unroll
            for (int c = 0; c < 6; ++c) out[c] = __builtin_nan("");
        }
        return true;
    }
#endif
    const double a = div_fast(p, 1.0 - ecc * ecc);
    const double ka = MU * a;
    const double e_c = rn * vv * inv_mu - 1.0;                                                   // e cos E | e cosh F
    double nu;
    if (!HYPER_ONLY && a > 0.0) {                                                                // (farnocchia.py:295-300)
        const double e_se = rv * rsqrt_nr(ka);
        double sh, ch;
        sincos_small(0.5 * atan2_fast(e_se, e_c), sh, ch);
        nu = 2.0 * atan2_fast(sqrt_fast(div_fast(1.0 + ecc, 1.0 - ecc)) * div_fast(sh, ch), 1.0);
    } else {                                                                                     // (:301-304)
        const double e_sh = rv * rsqrt_nr(-ka);
        nu = genf::F_to_nu(0.5 * genf::log_pos(div_fast(e_c + e_sh, e_c - e_sh)), ecc);
    }
    nu = wrap_pi(nu);
    const double q = div_fast(p, 1.0 + ecc);
    double sn, cn;
    sincos_small(nu, sn, cn);                     // (nu is wrapped)
    double nu1;
    const bool hyper = HYPER_ONLY || ecc > 1.0 + 1e-2;
    SSA_KEP_DBG_BRANCH(hyper ? 1 : (ecc >= 1.0 - 1e-2 ? 2 : 4));   // hyperbolic | near-parabolic band | elliptic the series declined
    if (hyper) {              // the strong-hyperbolic branch (farnocchia.py:909-912, :1001-1004)
        const bool beyond = 1.0 + ecc * cn < 0.0;                                                // (:885-888: beyond the asymptote -> NaN)
        // delta_t_from_nu (:909-912): F from nu again, M = e sinh F - F
        const double xh = sqrt_fast(div_fast(ecc - 1.0, ecc + 1.0)) * div_fast(sn, 1.0 + cn);
        const double F0 = genf::log_pos(div_fast(1.0 + xh, 1.0 - xh));
        double sh, chm1;
        sinh_coshm1(F0, sh, chm1);
        const double M0 = ecc * sh - F0;
        const double em1 = ecc - 1.0;
        const double nmm = sqrt_fast(div_fast(MU * em1 * em1 * em1, q * q * q));
        const double M = nmm * (div_fast(M0, nmm) + tof);
        // nu_from_delta_t (:1001-1004): newton from asinh(M / e)
        const double me = div_fast(M, ecc), am = fabs(me);
        const double F = genf::newton_hyp(copysign(genf::log_pos(am + sqrt_fast(am * am + 1.0)), me), M, ecc);
        nu1 = genf::F_to_nu(F, ecc);
        if (beyond) nu1 = __builtin_nan("");
    } else {                  // elliptic / parabolic / near-parabolic bands (inline: this instance lives inside the out-of-line restatement)
        nu1 = kepler_band_nu_inl(nu, ecc, q, tof);
    }
    // coe2rv (:101-161).  The reference places the perifocal frame by argp = (argument of latitude of r) - nu (rv2coe, :305-307) and
    // turns by nu1 from there: the new direction is the OLD POSITION'S direction advanced by nu1 - nu in the orbital plane -- the
    // error nu carries from its F round trip cancels in that difference (measuring nu1 from e / |e| instead leaves it in: 10 m
    // of tangential error at r = 7e9 m where the reference has 0.6 m).  So: rhat = r / |r|, that = (h x r) / (|h| |r|),
    //   r1 = fr (cos(nu1 - nu) rhat + sin(nu1 - nu) that)
    //   v1 = fv ((e sin nu - sin(nu1 - nu)) rhat + (e cos nu + cos(nu1 - nu)) that)
    // -- the perifocal expressions of (:101-161) with P = cos nu rhat - sin nu that, Q = sin nu rhat + cos nu that.
    double s1, c1n, sd, cd;
    sincos_small(nu1, s1, c1n);                   // (|nu1| <= pi: 2 atan(.))
    sincos_small(nu1 - nu, sd, cd);
    const double fr = div_fast(p, 1.0 + ecc * c1n), fv = sqrt_fast(div_fast(MU, p));
    const double ir = rcp_nr(rn), it = ir * rsqrt_nr(hh);
    const double rh[3] = {r[0] * ir, r[1] * ir, r[2] * ir};
    const double th[3] = {(h[1] * r[2] - h[2] * r[1]) * it, (h[2] * r[0] - h[0] * r[2]) * it, (h[0] * r[1] - h[1] * r[0]) * it};
    const double px = cd * fr, py = sd * fr, vx = (ecc * sn - sd) * fv, vy = (ecc * cn + cd) * fv;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        out[c] = px * rh[c] + py * th[c];
        out[3 + c] = vx * rh[c] + vy * th[c];
    }
    return true;
}
template <int TAG>
SSA_DEV Vec6 kepler_general_fast_impl(Vec6 xin, double tof)
{
    const double tol = 1e-8;
    const double* r = xin.v;
    const double* v = xin.v + 3;
    Vec6 outv;

    double h[3] = {r[1] * v[2] - r[2] * v[1], r[2] * v[0] - r[0] * v[2], r[0] * v[1] - r[1] * v[0]};
    double n[3] = {-h[1], h[0], 0.0};
    const double inv_mu = 1.0 / MU;
    double rn = sqrt_fast(dot3(r, r)), vv = dot3(v, v), rv = dot3(r, v);
    double c1 = vv - div_fast(MU, rn);
    double e[3] = {(c1 * r[0] - rv * v[0]) * inv_mu, (c1 * r[1] - rv * v[1]) * inv_mu, (c1 * r[2] - rv * v[2]) * inv_mu};
    double ecc = sqrt_fast(dot3(e, e));
    double p = dot3(h, h) * inv_mu;
    double hn = sqrt(dot3(h, h));
    const double inv_hn = rcp_nr(hn);
    double inc = acos(h[2] / hn);        // (as rv2coe_elliptic: the equatorial test needs the correctly rounded quotient)
    bool circular = ecc < tol, equatorial = fabs(inc) < tol;
    double raan, argp, nu;
    if (equatorial && !circular) {
        raan = 0.0;
        argp = mod_2pi(atan2_fast(e[1], e[0]));
        double t[3] = {e[1] * r[2] - e[2] * r[1], e[2] * r[0] - e[0] * r[2], e[0] * r[1] - e[1] * r[0]};
        nu = atan2_fast(dot3(h, t) * inv_hn, dot3(r, e));
    } else if (!equatorial && circular) {
        raan = mod_2pi(atan2_fast(n[1], n[0]));
        argp = 0.0;
        double t[3] = {h[1] * n[2] - h[2] * n[1], h[2] * n[0] - h[0] * n[2], h[0] * n[1] - h[1] * n[0]};
        nu = atan2_fast(dot3(r, t) * inv_hn, dot3(r, n));
    } else if (equatorial && circular) {
        raan = 0.0;
        argp = 0.0;
        nu = mod_2pi(atan2_fast(r[1], r[0]));
    } else {
        double a = div_fast(p, 1.0 - ecc * ecc);
        double ka = MU * a;
        if (a > 0.0) {
            double e_se = rv * rsqrt_nr(ka);
            double e_ce = rn * vv * inv_mu - 1.0;
            double sh, ch;
            sincos_fast(0.5 * atan2_fast(e_se, e_ce), sh, ch);
            nu = 2.0 * atan2_fast(sqrt_fast(div_fast(1.0 + ecc, 1.0 - ecc)) * div_fast(sh, ch), 1.0);
        } else {
            double e_sh = rv * rsqrt_nr(-ka);
            double e_ch = rn * vv * inv_mu - 1.0;
            nu = genf::F_to_nu(0.5 * genf::log_pos(div_fast(e_ch + e_sh, e_ch - e_sh)), ecc);
        }
        raan = mod_2pi(atan2_fast(n[1], n[0]));
        double t[3] = {h[1] * n[2] - h[2] * n[1], h[2] * n[0] - h[0] * n[2], h[0] * n[1] - h[1] * n[0]};
        double px = dot3(r, n), py = dot3(r, t) * inv_hn;
        argp = mod_2pi(atan2_fast(py, px) - nu);
    }
    nu = wrap_pi(nu);
    double q = div_fast(p, 1.0 + ecc);
    double nu1;
    SSA_KEP_DBG_BRANCH(ecc > 1.0 + 1e-2 ? 1 : (ecc >= 1.0 - 1e-2 ? 2 : 4));   // hyperbolic | near-parabolic band | elliptic the series declined
    if (ecc > 1.0 + 1e-2) {   // the strong-hyperbolic branch (farnocchia.py:909-912, :1001-1004): where a diverged filter lives
        double sn, cn;
        sincos_fast(nu, sn, cn);
        if (1.0 + ecc * cn < 0.0) nu1 = __builtin_nan("");          // (:885-888: beyond the asymptote)
        else {
            const double F0 = genf::nu_to_F(nu, ecc);
            double sh, chm1;
            sinh_coshm1(F0, sh, chm1);
            const double M0 = ecc * sh - F0;
            const double em1 = ecc - 1.0;
            const double nmm = sqrt_fast(div_fast(MU * em1 * em1 * em1, q * q * q));
            const double M = nmm * (div_fast(M0, nmm) + tof);
            const double me = div_fast(M, ecc);
            // asinh(M / e) = sign log(|.| + sqrt(.^2 + 1))
            const double am = fabs(me);
            const double F = genf::newton_hyp(copysign(genf::log_pos(am + sqrt_fast(am * am + 1.0)), me), M, ecc);
            nu1 = genf::F_to_nu(F, ecc);
        }
    } else {                  // elliptic / parabolic / near-parabolic bands: the complete restatement
        nu1 = kepler_band_nu<TAG>(nu, ecc, q, tof);
    }
    coe2rv(p, ecc, inc, raan, argp, nu1, outv.v);
    return outv;
}

}  // namespace ssa
