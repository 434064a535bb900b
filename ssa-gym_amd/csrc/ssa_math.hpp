// ssa_math.hpp -- per-lane fp64 device math for the ssa-gym hot path (gfx950).
//
// Everything here runs in one lane on one state vector; the kernels in
// ssa_kernels.hip decide which lane works on which sigma point / object.
// Citations are file:line under the reference root.
#pragma once
#include <hip/hip_runtime.h>

#ifndef SSA_STEP_WAVES
#define SSA_STEP_WAVES 5   // waves per SIMD the step kernel (and its out-of-line callees) must leave room for:
                           // 96 VGPRs; 256 CUs x 4 SIMDs x 5 = 5120 resident wavefronts = 20 480 objects in one round
#endif
namespace ssa {

constexpr double MU = 398600441800000.0;  // farnocchia.py:1060 (k hard-coded)
constexpr double PI = 3.141592653589793;
constexpr double TWO_PI = 6.283185307179586;
constexpr double NEWTON_TOL = 1.48e-08;   // farnocchia.py:337

#define SSA_DEV __device__ __forceinline__
// An optimisation barrier on a double: what follows cannot be contracted with what produced it (under -ffp-contract=fast a pragma inside
// an inlined function did not keep `scale * p + jit` from becoming ONE fma -- measured: tests/test_hip_step.py::
// test_ladder_on_the_ill_conditioned_tile).  No instruction: the value merely passes through a register the compiler cannot see into.
#if defined(__HIP_DEVICE_COMPILE__)
#define SSA_OPAQUE(x) asm volatile("" : "+v"(x))
#elif defined(__x86_64__)
#define SSA_OPAQUE(x) asm volatile("" : "+x"(x))
#else
#define SSA_OPAQUE(x) asm volatile("" : "+r"(x))
#endif

SSA_DEV double dot3(const double* a, const double* b) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }

// 1/v and 1/sqrt(v) to double precision (<= ~1 ulp) from the hardware estimates (v_rcp_f64 / v_rsq_f64, relative
// error e ~ 2^-26) plus ONE third-order correction -- 1/v = y (1 + e + e^2), 1/sqrt(v) = y (1 + e/2 + 3 e^2/8), residual
// e^3 ~ 2^-78 -- 3 / 5 VALU instructions instead of the 10-15 of the IEEE division / square-root sequences (and 1 / 3
// fewer than two Newton steps).  For finite, non-denormal positive-magnitude arguments.
SSA_DEV double rcp_nr(double v)
{
    const double y = __builtin_amdgcn_rcp(v);
    const double e = fma(-v, y, 1.0);
    return fma(y, fma(e, e, e), y);
}
SSA_DEV double rsqrt_nr(double v)
{
    const double y = __builtin_amdgcn_rsq(v);
    const double e = fma(-v * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}

// a / b and sqrt(v) through the refined hardware estimates (<= 1-2 ulp: the reference's numpy / libm chain is correctly
// rounded here, the same class; the IEEE sequences cost 14-25 instructions each and SSA_PROP_ELEMENTS has 13 + 8 of them
// per sigma point)
SSA_DEV double div_fast(double a, double b) { return a * rcp_nr(b); }
SSA_DEV double sqrt_fast(double v) { return (v == 0.0) ? 0.0 : v * rsqrt_nr(v); }

// sin and cos of a moderately sized angle (|x| < 64: every angle of the element chain): Cody-Waite reduction by pi/2 in two
// parts and the fdlibm kernel polynomials on |r| <= pi/4; 1.5 ulp.  Larger arguments (whole-wave branch) go to libm.
SSA_DEV void sincos_fast(double x, double& so, double& co)
{
    const bool small = fabs(x) < 64.0;
    {
        const double k = rint(x * 0.63661977236758134308);
        double r = fma(-k, 1.57079632673412561417e+00, x);
        r = fma(-k, 6.07710050650619224932e-11, r);
        const double z = r * r;
        double ps = 1.58969099521155010221e-10;
        ps = fma(ps, z, -2.50507602534068634195e-08);
        ps = fma(ps, z, 2.75573137070700676789e-06);
        ps = fma(ps, z, -1.98412698298579493134e-04);
        ps = fma(ps, z, 8.33333333332248946124e-03);
        ps = fma(ps, z, -1.66666666666666324348e-01);
        const double s = fma(z * r, ps, r);
        double pc = -1.13596475577881948265e-11;
        pc = fma(pc, z, 2.08757232129817482790e-09);
        pc = fma(pc, z, -2.75573143513906633035e-07);
        pc = fma(pc, z, 2.48015872894767294178e-05);
        pc = fma(pc, z, -1.38888888888741095749e-03);
        pc = fma(pc, z, 4.16666666666666019037e-02);
        const double c = 1.0 - fma(-z * z, pc, 0.5 * z);
        const int n = (int)k & 3;
        const double a = (n & 1) ? c : s, b = (n & 1) ? s : c;   // quadrant
        so = (n & 2) ? -a : a;
        co = ((n + 1) & 2) ? -b : b;
    }
    if (!__all(small)) {     // (whole-wave branch; the lanes inside the range keep the result above: lane-local arithmetic)
        double sl, cl;
        sincos(x, &sl, &cl);
        if (!small) { so = sl; co = cl; }
    }
}
// the same for an argument KNOWN to lie inside the range (a wrapped angle, 2 atan(.)): no libm branch at all; NaN in, NaN out
SSA_DEV void sincos_small(double x, double& so, double& co)
{
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double s = fma(z * r, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double c = 1.0 - fma(-z * z, pc, 0.5 * z);
    const int n = (int)k & 3;
    const double a = (n & 1) ? c : s, b = (n & 1) ? s : c;   // quadrant
    so = (n & 2) ? -a : a;
    co = ((n + 1) & 2) ? -b : b;
}
// (s, c) <- (sin, cos)(angle + d), |d| <= 0.02: truncation d^9/9! < 2e-21
SSA_DEV void rot_small(double d, double& s, double& c)
{
    const double d2 = d * d;
    const double sd = d * (1.0 - d2 * (1.0 / 6.0) * (1.0 - d2 * (1.0 / 20.0) * (1.0 - d2 * (1.0 / 42.0))));
    const double cd = 1.0 - d2 * 0.5 * (1.0 - d2 * (1.0 / 12.0) * (1.0 - d2 * (1.0 / 30.0) * (1.0 - d2 * (1.0 / 56.0))));
    const double s2 = s * cd + c * sd;
    c = c * cd - s * sd;
    s = s2;
}

// atan2(y, x) in (-pi, pi] (atan2(0, 0) = 0 as libm): ONE division.  The ratio is folded into |t| <= tan(pi/8) before
// it is formed -- min/max swap (pi/2 - .) and, above tan(pi/8), the shift by pi/4, t = (mn - mx) / (mx + mn) -- and the
// fdlibm atan kernel polynomial (|t| < 7/16) does the rest: ~2 ulp at pi/4, absolute error < 3e-16 everywhere, ~45
// instructions against ~100 for the libm atan2 (and the asin of the elevation becomes a second call of THIS routine:
// asin(u / r) = atan2(u, sqrt(e^2 + n^2)), so that two lanes can share one instruction stream for azimuth and elevation).
SSA_DEV double atan2_fast(double y, double x)
{
    const double ax = fabs(x), ay = fabs(y);
    const bool swap = ay > ax;
    const double mn = swap ? ax : ay, mx = swap ? ay : ax;
    const bool mid = mn > 0.41421356237309503 * mx;
    const double num = mid ? mn - mx : mn, den = mid ? mx + mn : mx;
    const double t = (den == 0.0) ? 0.0 : div_fast(num, den);
    const double z = t * t, w = z * z;
    double s1 = 1.62858201153657823623e-02;
    s1 = fma(s1, w, 4.97687799461593236017e-02);
    s1 = fma(s1, w, 6.66107313738753120669e-02);
    s1 = fma(s1, w, 9.09088713343650656196e-02);
    s1 = fma(s1, w, 1.42857142725034663711e-01);
    s1 = fma(s1, w, 3.33333333333329318027e-01);
    double s2 = -3.65315727442169155270e-02;
    s2 = fma(s2, w, -5.83357013379057348645e-02);
    s2 = fma(s2, w, -7.69187620504482999495e-02);
    s2 = fma(s2, w, -1.11111104054623557880e-01);
    s2 = fma(s2, w, -1.99999999998764832476e-01);
    double r = fma(-t, fma(z, s1, w * s2), t);
    // pi/4 and pi/2 in two parts so that the folded-back angle keeps its last bit
    if (mid) r = 7.85398163397448278999e-01 + (r + 3.06161699786838301793e-17);
    if (swap) r = 1.57079632679489655800e+00 - (r - 6.12323399573676603587e-17);
    if (__builtin_signbit(x)) r = 3.14159265358979311600e+00 - (r - 1.22464679914735317720e-16);   // (signed zeros as libm)
    return __builtin_copysign(r, y);
}

// Python / numpy `a % (2 pi)` (result in [0, 2 pi)); exact remainder via one FMA.
SSA_DEV double mod_2pi(double a)
{
    double k = floor(a * (1.0 / TWO_PI));
    double r = fma(-k, TWO_PI, a);
    if (r < 0.0) r += TWO_PI;
    if (r >= TWO_PI) r -= TWO_PI;
    return r;
}
// (a + pi) % (2 pi) - pi   (farnocchia.py:311, :953)
SSA_DEV double wrap_pi(double a) { return mod_2pi(a + PI) - PI; }

// ---------------------------------------------------------------------------
// Kepler's equation, elliptic: Newton on E - e sin E - M with the reference's
// starter, step tolerance and iteration cap; NaN when it does not converge
// (farnocchia.py:337-353, :573-601).
SSA_DEV double solve_kepler_E(double M, double ecc)
{
    double p0 = (ecc < 0.8) ? M : (M > 0.0 ? PI : (M < 0.0 ? -PI : 0.0));
    double res = __builtin_nan("");
    bool done = false;
    double s, c;
    sincos_fast(p0, s, c);
    for (int it = 0; it < 50; ++it) {
        const double fval = (p0 - ecc * s) - M;
        const double fder = 1.0 - ecc * c;
        const double p = p0 - div_fast(fval, fder);
        const double d = p - p0;
        if (!done && fabs(d) < NEWTON_TOL) { res = p; done = true; }
        p0 = p;
        if (__all(done)) break;
        // sin / cos of the new iterate: carried along by the angle-addition formulas while the Newton step is small
        // (every step after the first on the catalogue), evaluated afresh otherwise
        if (fabs(d) <= 0.02) rot_small(d, s, c);
        else sincos_fast(p0, s, c);
    }
    return res;
}

// coe2rv (farnocchia.py:101-161): perifocal state rotated by R3(raan) R1(inc) R3(argp).
SSA_DEV void coe2rv(double p, double ecc, double inc, double raan, double argp, double nu, double* out)
{
    double sn, cn, sO, cO, si, ci, sw, cw;
    sincos_fast(nu, sn, cn);
    sincos_fast(raan, sO, cO);
    sincos_fast(inc, si, ci);
    sincos_fast(argp, sw, cw);
    double fr = div_fast(p, 1.0 + ecc * cn), fv = sqrt_fast(div_fast(MU, p));
    double px = cn * fr, py = sn * fr, vx = -sn * fv, vy = (ecc + cn) * fv;
    double r00 = cO * cw - sO * ci * sw, r01 = -cO * sw - sO * ci * cw;
    double r10 = sO * cw + cO * ci * sw, r11 = -sO * sw + cO * ci * cw;
    double r20 = si * sw, r21 = si * cw;
    out[0] = px * r00 + py * r01;
    out[1] = px * r10 + py * r11;
    out[2] = px * r20 + py * r21;
    out[3] = vx * r00 + vy * r01;
    out[4] = vx * r10 + vy * r11;
    out[5] = vx * r20 + vy * r21;
}

// (The inverse trigonometric calls of this chain stay libm's: SSA_PROP_ELEMENTS mirrors the reference's arithmetic, whose
// rounding noise is amplified ~1e8 x by the alpha = 1e-3 .. 1e-4 unscented transform; with atan2_fast -- same error class, 3 %
// faster -- the reference's Test 7 threshold after 50 + 50 predicts (tests.py:185-186, met by a 15 % margin by the
// reference itself) fell on the wrong side of that noise: 0.035 m against < 0.01 m.)
// rv2coe (farnocchia.py:165-313), elliptic side of the general branch inline; returns false
// when the orbit is not a strong-elliptic one (a <= 0 or ecc >= 1 - 1e-2) so that the caller
// can take the complete restatement below.  coe = p, ecc, inc, raan, argp, nu.
#ifdef SSA_EL_INVTRIG_FAST   // (diagnostic build: what the libm inverse trigonometry of this chain costs -- NOT for use: Test 7 moves)
#define SSA_EL_ATAN2(y, x) atan2_fast((y), (x))
#define SSA_EL_ATAN(v) atan2_fast((v), 1.0)
#define SSA_EL_ACOS(c) atan2_fast(sqrt_fast(fma(-(c), (c), 1.0)), (c))
#else
#define SSA_EL_ATAN2(y, x) atan2((y), (x))
#define SSA_EL_ATAN(v) atan(v)
#define SSA_EL_ACOS(c) acos(c)
#endif
SSA_DEV bool rv2coe_elliptic(const double* r, const double* v, double* coe)
{
    const double tol = 1e-8;
    double h[3] = {r[1] * v[2] - r[2] * v[1], r[2] * v[0] - r[0] * v[2], r[0] * v[1] - r[1] * v[0]};
    double n[3] = {-h[1], h[0], 0.0};  // cross([0,0,1], h)
    double rn = sqrt_fast(dot3(r, r)), vv = dot3(v, v), rv = dot3(r, v);
    const double inv_mu = 1.0 / MU;
    double c1 = vv - div_fast(MU, rn);
    double e[3] = {(c1 * r[0] - rv * v[0]) * inv_mu, (c1 * r[1] - rv * v[1]) * inv_mu, (c1 * r[2] - rv * v[2]) * inv_mu};
    double ecc = sqrt_fast(dot3(e, e));
    double p = dot3(h, h) * inv_mu;
    double hn = sqrt(dot3(h, h));   // correctly rounded, as is h_z / |h| below: the equatorial test |inc| < 1e-8 holds only
    const double inv_hn = rcp_nr(hn);   // when that quotient is EXACTLY 1 (acos(1 - 1 ulp) = 1.5e-8), farnocchia.py:278
    double inc = SSA_EL_ACOS(h[2] / hn);   // (IEEE division: for an equatorial orbit h_z / |h| must be exactly 1, not 1 + 1 ulp -> acos NaN)
    bool circular = ecc < tol, equatorial = fabs(inc) < tol;
    double raan, argp, nu;
    if (equatorial && !circular) {
        raan = 0.0;
        argp = mod_2pi(SSA_EL_ATAN2(e[1], e[0]));
        double t[3] = {e[1] * r[2] - e[2] * r[1], e[2] * r[0] - e[0] * r[2], e[0] * r[1] - e[1] * r[0]};
        nu = SSA_EL_ATAN2(dot3(h, t) * inv_hn, dot3(r, e));
    } else if (!equatorial && circular) {
        raan = mod_2pi(SSA_EL_ATAN2(n[1], n[0]));
        argp = 0.0;
        double t[3] = {h[1] * n[2] - h[2] * n[1], h[2] * n[0] - h[0] * n[2], h[0] * n[1] - h[1] * n[0]};
        nu = SSA_EL_ATAN2(dot3(r, t) * inv_hn, dot3(r, n));
    } else if (equatorial && circular) {
        raan = 0.0;
        argp = 0.0;
        nu = mod_2pi(SSA_EL_ATAN2(r[1], r[0]));
    } else {
        double a = div_fast(p, 1.0 - ecc * ecc);
        if (!(a > 0.0)) return false;
        double e_se = rv * rsqrt_nr(MU * a);
        double e_ce = rn * vv * inv_mu - 1.0;
        double E = SSA_EL_ATAN2(e_se, e_ce);
        {
            double sh, ch;
            sincos_fast(0.5 * E, sh, ch);
            nu = 2.0 * SSA_EL_ATAN(sqrt_fast(div_fast(1.0 + ecc, 1.0 - ecc)) * div_fast(sh, ch));   // tan(E/2) = sin / cos
        }
        raan = mod_2pi(SSA_EL_ATAN2(n[1], n[0]));
        double t[3] = {h[1] * n[2] - h[2] * n[1], h[2] * n[0] - h[0] * n[2], h[0] * n[1] - h[1] * n[0]};
        double px = dot3(r, n), py = dot3(r, t) * inv_hn;
        argp = mod_2pi(SSA_EL_ATAN2(py, px) - nu);
    }
    nu = wrap_pi(nu);
    coe[0] = p; coe[1] = ecc; coe[2] = inc; coe[3] = raan; coe[4] = argp; coe[5] = nu;
    return ecc < 1.0 - 1e-2;
}

// Complete restatement of farnocchia() with every conic branch (parabolic, near-parabolic,
// hyperbolic): farnocchia.py:847-1050.  Out of line: only objects that have left the
// strong-elliptic regime (a diverged filter) ever reach it.
// Arguments travel BY VALUE (registers): passing pointers to the caller's arrays would force those
// arrays -- and every inline use of them -- into scratch memory.
struct Vec6 { double v[6]; };
struct Vec8 { double v[8]; };
__device__ __noinline__ Vec6 kepler_general_v(Vec6 x, double tof);
__device__ __noinline__ Vec8 kepler_general_diag_v(Vec6 x, double tof, Vec6* out);
SSA_DEV void kepler_general(const double* x, double tof, double* out, double* diag)
{
    Vec6 xi;
#pragma unroll
    for (int i = 0; i < 6; ++i) xi.v[i] = x[i];
    if (diag) {
        Vec6 o;
        Vec8 d = kepler_general_diag_v(xi, tof, &o);
#pragma unroll
        for (int i = 0; i < 6; ++i) out[i] = o.v[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) diag[i] = d.v[i];
    } else {
        Vec6 o = kepler_general_v(xi, tof);
#pragma unroll
        for (int i = 0; i < 6; ++i) out[i] = o.v[i];
    }
}

// SSA_PROP_ELEMENTS: farnocchia() for the strong-elliptic regime, operation by operation
// (rv2coe -> delta_t_from_nu :871-875 -> nu_from_delta_t :946-954 -> coe2rv).
// diag (optional, 8 doubles): p, ecc, inc, raan, argp, nu0, delta_t0, nu.
SSA_DEV void kepler_elements(const double* x, double tof, double* out, double* diag)
{
    double coe[6];
    if (!rv2coe_elliptic(x, x + 3, coe)) {
        kepler_general(x, tof, out, diag);
        return;
    }
    double p = coe[0], ecc = coe[1], nu0 = coe[5];
#if defined(SSA_EL_CUT) && SSA_EL_CUT == 1   // (diagnostic builds: the chain cut behind a stage, for per-stage instruction counts)
    for (int i = 0; i < 6; ++i) out[i] = coe[i];
    return;
#endif
    double q = div_fast(p, 1.0 + ecc);
    double ome = 1.0 - ecc;
    double sh, ch;
    sincos_fast(0.5 * nu0, sh, ch);
    double E0 = 2.0 * SSA_EL_ATAN(sqrt_fast(div_fast(ome, 1.0 + ecc)) * div_fast(sh, ch));
    double sE0, cE0;
    sincos_fast(E0, sE0, cE0);
    double M0 = E0 - ecc * sE0;
    double nmm = sqrt_fast(div_fast(MU * ome * ome * ome, q * q * q));
    double dt0 = div_fast(M0, nmm);
    double M = nmm * (dt0 + tof);
#if defined(SSA_EL_CUT) && SSA_EL_CUT == 2
    for (int i = 0; i < 6; ++i) out[i] = coe[i] + M;
    return;
#endif
    double E = solve_kepler_E(wrap_pi(M), ecc);
#if defined(SSA_EL_CUT) && SSA_EL_CUT == 3
    for (int i = 0; i < 6; ++i) out[i] = coe[i] + E;
    return;
#endif
    sincos_fast(0.5 * E, sh, ch);
    double nu = 2.0 * SSA_EL_ATAN(sqrt_fast(div_fast(1.0 + ecc, ome)) * div_fast(sh, ch));
#if defined(SSA_EL_CUT) && SSA_EL_CUT == 4
    for (int i = 0; i < 6; ++i) out[i] = coe[i] + nu;
    return;
#endif
    coe2rv(p, ecc, coe[2], coe[3], coe[4], nu, out);
    if (diag) {
        for (int i = 0; i < 6; ++i) diag[i] = coe[i];
        diag[6] = dt0;
        diag[7] = nu;
    }
}

// e^x for 0 <= x < 709: two-part Cody-Waite reduction by ln 2, Taylor polynomial of degree 13 on |r| <= ln 2 / 2 (truncation
// 4e-18), v_ldexp_f64; < 2 ulp, ~20 instructions (libm's exp: ~50)
SSA_DEV double exp_fast(double x)
{
    const double k = rint(x * 1.44269504088896338700e+00);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// ---------------------------------------------------------------------------
// sinh x and cosh x - 1 without cancellation: Taylor series for |x| < 0.5 (truncation < 1e-19 relative),
// libm beyond.
SSA_DEV void sinh_coshm1(double x, double& sh, double& chm1)
{
    if (fabs(x) < 0.5) {
        const double x2 = x * x;
        sh = x * (1.0 + x2 * (1.0 / 6.0) * (1.0 + x2 * (1.0 / 20.0) * (1.0 + x2 * (1.0 / 42.0) * (1.0 + x2 * (1.0 / 72.0) *
             (1.0 + x2 * (1.0 / 110.0) * (1.0 + x2 * (1.0 / 156.0) * (1.0 + x2 * (1.0 / 210.0))))))));
        chm1 = x2 * 0.5 * (1.0 + x2 * (1.0 / 12.0) * (1.0 + x2 * (1.0 / 30.0) * (1.0 + x2 * (1.0 / 56.0) * (1.0 + x2 * (1.0 / 90.0) *
               (1.0 + x2 * (1.0 / 132.0) * (1.0 + x2 * (1.0 / 182.0) * (1.0 + x2 * (1.0 / 240.0))))))));
    } else {   // one exponential instead of two libm calls: no cancellation for |x| >= 0.5 (cosh x - 1 >= 0.127)
        const double ax = fabs(x);
        // (beyond 709 the exponential leaves the double range: infinity -- libm's sinh stays finite up to 710.47, a distinction no
        // state of this path can feel: F = 709 is a hyperbolic anomaly of e^709 radii)
        const double ex = (ax < 709.0) ? exp_fast(ax) : ((ax == ax) ? __builtin_inf() : ax), ie = rcp_nr(ex);
        const double shp = 0.5 * (ex - ie);
        sh = (x < 0.0) ? -shp : shp;
        chm1 = 0.5 * (ex + ie) - 1.0;
    }
}

// Stumpff functions c2(z) = (1 - cos sqrt z)/z, c3(z) = (sqrt z - sin sqrt z)/sqrt(z)^3 and their
// hyperbolic continuations; series around z = 0.
SSA_DEV void stumpff(double z, double& c2, double& c3)
{
    if (fabs(z) < 0.5) {
        c2 = 0.5 - z * (1.0 / 24.0 - z * (1.0 / 720.0 - z * (1.0 / 40320.0 - z * (1.0 / 3628800.0 - z * (1.0 / 479001600.0 -
             z * (1.0 / 87178291200.0 - z * (1.0 / 20922789888000.0)))))));
        c3 = 1.0 / 6.0 - z * (1.0 / 120.0 - z * (1.0 / 5040.0 - z * (1.0 / 362880.0 - z * (1.0 / 39916800.0 - z * (1.0 / 6227020800.0 -
             z * (1.0 / 1307674368000.0 - z * (1.0 / 355687428096000.0)))))));
    } else if (z > 0.0) {
        const double isz = rsqrt_nr(z), sz = z * isz, iz = isz * isz;
        double sn, cs;
        sincos_fast(sz, sn, cs);
        c2 = (1.0 - cs) * iz;
        c3 = (sz - sn) * (isz * iz);
    } else {
        const double isz = rsqrt_nr(-z), sz = -z * isz, iz = isz * isz;
        double sh, chm1;
        sinh_coshm1(sz, sh, chm1);
        c2 = chm1 * iz;
        c3 = (sh - sz) * (isz * iz);
    }
}

// ---------------------------------------------------------------------------
// One solver for every conic of a SHORT step: universal variables.  With chi the universal anomaly, z = alpha chi^2
// (alpha = 1/a: > 0 ellipse, < 0 hyperbola, ~0 parabola) and the Stumpff functions c2(z), c3(z), Kepler's equation
//     F(chi) = r0 chi + sigma0 chi^2 c2 + (1 - r0 alpha) chi^3 c3 - sqrt(mu) tof = 0,     sigma0 = r.v / sqrt(mu)
// is the same expression for the strong-elliptic, the strong-hyperbolic and the near-parabolic branch of farnocchia()
// (farnocchia.py:871-919, 946-1004 switch between E, F and D there), F' = r > 0, F'' = (1 - r0 alpha) chi (1 - z c3) +
// sigma0 (1 - z c2).  For the steps of the env (tens of seconds: |z| is the squared eccentric-anomaly increment, << 1)
// c2, c3 are 8-term series (a lane whose |z| leaves the series' range, 0.5, is handed to the branch-wise solvers), the starter inverts the cubic truncation of F (relative error ~ z^(3/2)) and ONE Halley
// step lands within 1e-15 of the root; the loop runs a second one only where the first correction exceeded 1e-6.  Against
// the 80-bit oracle: 1.3e-15 relative on the catalogue at dt = 20 .. 150 s, 2e-15 on hyperbolic and near-parabolic
// states (the reference's own fp64 chain: 2.7e-14).  Because ellipses and hyperbolas share ONE instruction stream, a
// wavefront that holds diverged (hyperbolic) filters next to healthy ones -- two thirds of the wavefronts late in an
// episode -- no longer runs two solvers back to back.  `handled` is false for lanes outside the series domain (long
// steps, degenerate / non-finite states): those take the branch-wise solvers below.
SSA_DEV void stumpff_small(double z, double& c2, double& c3)
{
    c2 = 1.0 / 20922789888000.0;
    c2 = fma(-c2, z, 1.0 / 87178291200.0);
    c2 = fma(-c2, z, 1.0 / 479001600.0);
    c2 = fma(-c2, z, 1.0 / 3628800.0);
    c2 = fma(-c2, z, 1.0 / 40320.0);
    c2 = fma(-c2, z, 1.0 / 720.0);
    c2 = fma(-c2, z, 1.0 / 24.0);
    c2 = fma(-c2, z, 0.5);
    c3 = 1.0 / 355687428096000.0;
    c3 = fma(-c3, z, 1.0 / 1307674368000.0);
    c3 = fma(-c3, z, 1.0 / 6227020800.0);
    c3 = fma(-c3, z, 1.0 / 39916800.0);
    c3 = fma(-c3, z, 1.0 / 362880.0);
    c3 = fma(-c3, z, 1.0 / 5040.0);
    c3 = fma(-c3, z, 1.0 / 120.0);
    c3 = fma(-c3, z, 1.0 / 6.0);
}
// the same series cut after z^3 for |z| < 2e-3 (truncation z^4 / 10! : 1e-17 relative) -- every sigma point of a 20 s step of the
// catalogue (z ~ (n dt)^2 < 1e-3): six FMAs instead of fourteen per evaluation
SSA_DEV void stumpff_tiny(double z, double& c2, double& c3)
{
    c2 = fma(-fma(-fma(-1.0 / 40320.0, z, 1.0 / 720.0), z, 1.0 / 24.0), z, 0.5);
    c3 = fma(-fma(-fma(-1.0 / 362880.0, z, 1.0 / 5040.0), z, 1.0 / 120.0), z, 1.0 / 6.0);
}
// TINY: every lane of the wavefront is in the short series' range by the first-order estimate below (the instance without the long
// series).  Otherwise each lane still gets the series ITS OWN z asks for, so that a lane's rounding does not depend on who shares its
// wavefront: until round 4 the whole wavefront took the long series as soon as one lane needed it, the two forms differ by an ulp now and
// then, and an object's result depended -- in the last bit -- on its neighbours (found when a storage layout, which only changes who shares
// a wavefront, ended a 20 000-object episode with 776 failed filters instead of 775).  The rule is the series' own range, |z| < 2e-3,
// taken from the z at hand (nothing to carry along: a flag cost the rollout kernel a spill); a lane whose estimate is below 1.5e-3 stays
// below 2e-3 throughout -- what the TINY instance relies on for every lane -- so it sees the short series in either instance.
template <bool TINY>
SSA_DEV void stumpff_sel(double z, double& c2, double& c3)
{
    if (TINY) stumpff_tiny(z, c2, c3);
    else {
        stumpff_small(z, c2, c3);
        const bool tiny_lane = fabs(z) < 2e-3;
        if (__any(tiny_lane)) {
            double a, b;
            stumpff_tiny(z, a, b);
            if (tiny_lane) { c2 = a; c3 = b; }
        }
    }
}
template <bool TINY>
SSA_DEV bool kepler_uv_fast_t(const double* x, double tof, double* out, bool& handled);
SSA_DEV bool kepler_uv_fast(const double* x, double tof, double* out, bool& handled)
{
    // the first-order estimate z ~ alpha (sqrt(mu) tof / r0)^2 (bound 1.5e-3: the converged z stays below the 2e-3 the short series is good
    // for; a NaN lane takes either); the wavefront skips the long series when nobody needs it
    const double rr = dot3(x, x), vv = dot3(x + 3, x + 3);
    const double ir = rsqrt_nr(rr);
    const double z0 = (2.0 * ir - vv * (1.0 / MU)) * (MU * tof * tof) * (ir * ir);
    if (__all(!(fabs(z0) >= 1.5e-3))) return kepler_uv_fast_t<true>(x, tof, out, handled);
    return kepler_uv_fast_t<false>(x, tof, out, handled);
}
template <bool TINY>
SSA_DEV bool kepler_uv_fast_t(const double* x, double tof, double* out, bool& handled)
{
    const double* r = x;
    const double* v = x + 3;
    const double sqrt_mu = sqrt(MU), inv_sqrt_mu = 1.0 / sqrt(MU), inv_mu = 1.0 / MU;
    const double rr = dot3(r, r), vv = dot3(v, v), rv = dot3(r, v);
    const double inv_r0 = rsqrt_nr(rr);
    const double r0 = rr * inv_r0;
    const double alpha = 2.0 * inv_r0 - vv * inv_mu;
    const double sig = rv * inv_sqrt_mu, T = sqrt_mu * tof;
    const double k3 = 1.0 - r0 * alpha;
    // starter: fourth-order series inversion of F / r0 = chi + a2 chi^2 + a3 chi^3 + a4 chi^4 = t1 (a4 = -alpha a2 / 12 from the
    // z-terms of c2): relative error ~ z^2 / 12 + (a2 t1)^4, 1e-8 on the catalogue at 20 s -- the first Halley correction then
    // lies below the loop's 1e-6 and a second iteration (a quarter of this function) runs only for long steps
    const double t1 = T * inv_r0, a2 = 0.5 * sig * inv_r0, a3 = k3 * inv_r0 * (1.0 / 6.0);
    const double a22 = a2 * a2;
    const double c4 = a2 * fma(-5.0, a22, fma(5.0, a3, alpha * (1.0 / 12.0)));
    double chi = t1 * (1.0 - t1 * (a2 - t1 * ((2.0 * a22 - a3) + t1 * c4)));
    // series domain (NaN / inf / r0 = 0 compare false)
    // domain: the estimated |z| below 4 and a modest second-order term (NaN / inf / r0 = 0 compare false).  Beyond it are
    // long steps and filter states that have collapsed towards the Earth's centre (0.1 % of the sigma points of a late
    // episode): the branch-wise solvers take those.
    const bool small = (fabs(alpha) * t1 * t1 < 4.0) && (fabs(a2 * t1) < 1.0) && (r0 > 0.0) && (fabs(chi) <= 1.79769313486231570e308);
    bool live = small, conv = false;     // live: inside the series' range so far; conv: last correction below 1e-6
    double c2, c3, chi2, z;
#pragma unroll 1
    for (int it = 0; it < 6; ++it) {
        chi2 = chi * chi;
        z = alpha * chi2;
        stumpff_sel<TINY>(z, c2, c3);
        const double w3 = 1.0 - z * c3, w2 = 1.0 - z * c2;
        const double F = fma(r0, chi, chi2 * fma(k3 * chi, c3, sig * c2)) - T;
        const double rad = fma(chi2, c2, fma(sig * chi, w3, r0 * w2));
        const double rp = fma(k3 * chi, w3, sig * w2);
        const double den = fma(rad, rad, -0.5 * F * rp);
        double y = __builtin_amdgcn_rcp(den);
        y = y * fma(-den, y, 2.0);
        const double d = -F * rad * y;              // Halley
        live = live && (fabs(z) < 0.5);       // a lane whose |z| leaves the series' range stops here and is not handled
        if (live && !conv) {
            chi += d;
            conv = fabs(d) <= 1e-6 * fabs(chi);
        }
        if (__all(conv || !live)) break;
    }
    chi2 = chi * chi;
    z = alpha * chi2;
    stumpff_sel<TINY>(z, c2, c3);
    const double rad = fma(chi2, c2, fma(sig * chi, 1.0 - z * c3, r0 * (1.0 - z * c2)));
    const double inv_rad = rcp_nr(rad);
    const double f = 1.0 - chi2 * c2 * inv_r0;
    const double g = tof - chi2 * chi * c3 * inv_sqrt_mu;
    const double fd = sqrt_mu * chi * (z * c3 - 1.0) * inv_rad * inv_r0;
    const double gd = 1.0 - chi2 * c2 * inv_rad;
    out[0] = f * r[0] + g * v[0];
    out[1] = f * r[1] + g * v[1];
    out[2] = f * r[2] + g * v[2];
    out[3] = fd * r[0] + gd * v[0];
    out[4] = fd * r[1] + gd * v[1];
    out[5] = fd * r[2] + gd * v[2];
    handled = live && conv && (fabs(z) < 0.5) && (rad > 0.0) && (fabs(out[0]) <= 1.79769313486231570e308);
    return handled;
}

// The same equation for everything the series solver declines -- long steps (many revolutions), filter states that
// have collapsed towards the Earth's centre, |z| beyond the series: Stumpff functions in closed form (libm beyond
// |z| = 0.5) and Laguerre-Conway iterations (n = 5),
//     dchi = -5 F / (F' + sign(F') sqrt(|16 F'^2 - 20 F F''|)),
// which converge from any starting point on this monotone F without bracketing (2-4 iterations on the reference's
// catalogue up to one-day steps, <= 7 on 99.98 % of the diverged sigma points of a late episode; the 80-bit oracle and a
// DOP853 integration agree with it to 3e-13 at dt = 86 400 s).  Sixteen iterations without convergence is the analogue of
// newton() giving up (farnocchia.py:353): the caller sees NaN and the filter is marked failed.
SSA_DEV bool kepler_uv_general(const double* x, double tof, double* out)
{
    const double* r = x;
    const double* v = x + 3;
    const double sqrt_mu = sqrt(MU), inv_sqrt_mu = 1.0 / sqrt(MU), inv_mu = 1.0 / MU;
    const double rr = dot3(r, r), vv = dot3(v, v), rv = dot3(r, v);
    const double inv_r0 = rsqrt_nr(rr), r0 = rr * inv_r0;
    const double alpha = 2.0 * inv_r0 - vv * inv_mu;
    const double sig = rv * inv_sqrt_mu, T = sqrt_mu * tof;
    const double k3 = 1.0 - r0 * alpha;
    const double t1 = T * inv_r0;
    // starter: first order for short steps; the mean-motion guess for long elliptic steps; for strongly hyperbolic steps
    // (|z| > 1: F grows exponentially in chi) the logarithmic guess of the universal-variable literature (Vallado,
    // algorithm 8) -- from chi = T / r0 the iteration would walk down the exponential for dozens of steps
    const double z1 = alpha * t1 * t1;
    double chi = t1;
    if (z1 > 1.0) chi = T * alpha;
    else if (z1 < -1.0) {
        const double sa = sqrt(-alpha);
        const double h = log((-2.0 * alpha * T) / (sig + k3 / sa)) / sa;
        if (h > 0.0 && h <= 1.79769313486231570e308) chi = h;
    }
    const bool sane = (r0 > 0.0) && (fabs(alpha) <= 1.79769313486231570e308) && (fabs(chi) <= 1.79769313486231570e308);
    bool conv = false;
    double c2, c3, chi2, z;
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
        chi2 = chi * chi;
        z = alpha * chi2;
        stumpff(z, c2, c3);
        const double w3 = 1.0 - z * c3, w2 = 1.0 - z * c2;
        const double F = fma(r0, chi, chi2 * fma(k3 * chi, c3, sig * c2)) - T;
        const double rad = fma(chi2, c2, fma(sig * chi, w3, r0 * w2));
        const double rp = fma(k3 * chi, w3, sig * w2);
        const double disc = fabs(16.0 * rad * rad - 20.0 * F * rp);
        const double d = div_fast(-5.0 * F, rad + copysign(sqrt_fast(disc), rad));
        if (sane && !conv) {
            chi += d;
            conv = fabs(d) <= 1e-6 * fabs(chi);
        }
        if (__all(conv || !sane)) break;
    }
    chi2 = chi * chi;
    z = alpha * chi2;
    stumpff(z, c2, c3);
    const double rad = fma(chi2, c2, fma(sig * chi, 1.0 - z * c3, r0 * (1.0 - z * c2)));
    const double inv_rad = rcp_nr(rad);
    const double f = 1.0 - chi2 * c2 * inv_r0;
    const double g = tof - chi2 * chi * c3 * inv_sqrt_mu;
    const double fd = sqrt_mu * chi * (z * c3 - 1.0) * inv_rad * inv_r0;
    const double gd = 1.0 - chi2 * c2 * inv_rad;
    out[0] = f * r[0] + g * v[0];
    out[1] = f * r[1] + g * v[1];
    out[2] = f * r[2] + g * v[2];
    out[3] = fd * r[0] + gd * v[0];
    out[4] = fd * r[1] + gd * v[1];
    out[5] = fd * r[2] + gd * v[2];
    return sane && conv && (rad > 0.0) && (fabs(out[0]) <= 1.79769313486231570e308) && (fabs(out[3]) <= 1.79769313486231570e308);
}

// SSA_PROP_FG: every conic branch of farnocchia() through ONE equation -- the series / Halley solver for the steps it
// covers, the closed-form / Laguerre solver for the rest (a whole-wave branch: skipped when every lane was handled).
// false = no convergence / degenerate input = the NaN of farnocchia.py:353.
template <int TAG = 0>
SSA_DEV bool kepler_fg_fast(const double* x, double tof, double* out)
{
    bool handled;
    bool ok = kepler_uv_fast(x, tof, out, handled);
    if (__any(!handled)) {   // whole-wave branch
        __builtin_amdgcn_s_setprio(3);   // a straggler in the making (see robust_chol_row_lds): issue priority from here on
        if (!handled) ok = kepler_uv_general(x, tof, out);
    }
    return ok;
}

SSA_DEV bool kepler_elements_fast(const double* x, double tof, double* out)
{
    double coe[6] = {1e7, 0.0, 0.0, 0.0, 0.0, 0.0};
    bool ok = rv2coe_elliptic(x, x + 3, coe);
    double p = coe[0], ecc = ok ? coe[1] : 0.0, nu0 = coe[5];
    double q = div_fast(p, 1.0 + ecc);
    double ome = 1.0 - ecc;
    double sh, ch;
    sincos_fast(0.5 * nu0, sh, ch);
    double E0 = 2.0 * atan(sqrt_fast(div_fast(ome, 1.0 + ecc)) * div_fast(sh, ch));
    double sE0, cE0;
    sincos_fast(E0, sE0, cE0);
    double M0 = E0 - ecc * sE0;
    double nmm = sqrt_fast(div_fast(MU * ome * ome * ome, q * q * q));
    double dt0 = div_fast(M0, nmm);
    double M = nmm * (dt0 + tof);
    double E = solve_kepler_E(wrap_pi(M), ecc);
    sincos_fast(0.5 * E, sh, ch);
    double nu = 2.0 * atan(sqrt_fast(div_fast(1.0 + ecc, ome)) * div_fast(sh, ch));
    coe2rv(p, ecc, coe[2], coe[3], coe[4], nu, out);
    return ok && (E == E);
}

// ---------------------------------------------------------------------------
// SSA_PROP_J2_RK4 -- EXTENSION (the reference has no perturbed propagator on its hot path; its only
// numerical integrator, fx_xyz_cowell at dynamics.py:168-201, takes the perturbation as `ad` and is
// never called).  Cowell's formulation: r'' = -mu r/|r|^3 + a_J2 with the J2 acceleration of
// poliastro.core.perturbations.J2_perturbation (factor 3/2 mu J2 R^2 / r^5 and the 5 z^2/r^2 - {1,1,3} terms),
// integrated with the classical 4th-order Runge-Kutta scheme in `nsub` equal sub-steps.
struct J2Params { double j2, r_eq; int nsub; };

SSA_DEV void accel_j2(const double* r, const J2Params& q, double* a)
{
    const double r2 = dot3(r, r);
    const double inv_r = rsqrt_nr(r2);
    const double inv_r2 = inv_r * inv_r;
    const double inv_r3 = inv_r2 * inv_r;
    const double zz = 5.0 * r[2] * r[2] * inv_r2;
    const double fj = 1.5 * q.j2 * q.r_eq * q.r_eq * inv_r2;      // (3/2) J2 (R/r)^2
    const double k0 = -MU * inv_r3;
    a[0] = k0 * r[0] * (1.0 - fj * (zz - 1.0));
    a[1] = k0 * r[1] * (1.0 - fj * (zz - 1.0));
    a[2] = k0 * r[2] * (1.0 - fj * (zz - 3.0));
}

SSA_DEV bool propagate_j2_rk4(const double* x, double tof, const J2Params& q, double* out)
{
    double r[3] = {x[0], x[1], x[2]}, v[3] = {x[3], x[4], x[5]};
    const double h = tof / (double)q.nsub;
    for (int it = 0; it < q.nsub; ++it) {
        double k1[3], k2[3], k3[3], k4[3], rt[3], v2[3], v3[3], v4[3];
        accel_j2(r, q, k1);
#pragma unroll
        for (int c = 0; c < 3; ++c) { rt[c] = r[c] + 0.5 * h * v[c]; v2[c] = v[c] + 0.5 * h * k1[c]; }
        accel_j2(rt, q, k2);
#pragma unroll
        for (int c = 0; c < 3; ++c) { rt[c] = r[c] + 0.5 * h * v2[c]; v3[c] = v[c] + 0.5 * h * k2[c]; }
        accel_j2(rt, q, k3);
#pragma unroll
        for (int c = 0; c < 3; ++c) { rt[c] = r[c] + h * v3[c]; v4[c] = v[c] + h * k3[c]; }
        accel_j2(rt, q, k4);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r[c] += (h / 6.0) * (v[c] + 2.0 * v2[c] + 2.0 * v3[c] + v4[c]);
            v[c] += (h / 6.0) * (k1[c] + 2.0 * k2[c] + 2.0 * k3[c] + k4[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { out[c] = r[c]; out[3 + c] = v[c]; }
    return (out[0] == out[0]) && (out[3] == out[3]);
}

template <int PROP, int TAG = 0>
SSA_DEV bool kepler_step_fast(const double* x, double tof, double* out)
{
    if (PROP == 1) return kepler_fg_fast<TAG>(x, tof, out);
    return kepler_elements_fast(x, tof, out);
}

// SSA_PROP_FG with complete semantics: the f,g / universal-variable path, anything it declines (NaN
// input, no convergence) through the out-of-line restatement of farnocchia().
SSA_DEV void kepler_fg(const double* x, double tof, double* out)
{
    if (!kepler_fg_fast(x, tof, out)) kepler_general(x, tof, out, nullptr);
}

template <int PROP>
SSA_DEV void kepler_step(const double* x, double tof, double* out)
{
    if (PROP == 1) kepler_fg(x, tof, out);
    else kepler_elements(x, tof, out, nullptr);
}

// ---------------------------------------------------------------------------
// U2: upper Cholesky of a symmetric 6x6 given by its upper triangle (packed row-major,
// 21 values: (0,0)(0,1)..(0,5)(1,1)..(5,5)), LAPACK dpotf2('U') order; `jit` is added to the
// diagonal.  Returns false if a pivot is <= 0 or NaN (scipy -> LinAlgError).
SSA_DEV constexpr int tri(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

SSA_DEV bool chol6_upper(const double* A, double jit, double* U)
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        // (the jitter is ADDED to the entry as given -- dynamics.py:410 cholesky(a + e) -- never fused with a product that formed the entry in
        // the caller: see scaled_entry in ssa_kernels.hip)
        double a0 = A[tri(j, j)];
        SSA_OPAQUE(a0);
        double ajj = a0 + jit;
#pragma unroll
        for (int i = 0; i < j; ++i) ajj = fma(-U[tri(i, j)], U[tri(i, j)], ajj);
        ok = ok && (ajj > 0.0);
        double d = sqrt(ajj);
        U[tri(j, j)] = d;
        double inv = 1.0 / d;
#pragma unroll
        for (int cidx = j + 1; cidx < 6; ++cidx) {
            double sacc = A[tri(j, cidx)];
#pragma unroll
            for (int i = 0; i < j; ++i) sacc = fma(-U[tri(i, j)], U[tri(i, cidx)], sacc);
            U[tri(j, cidx)] = sacc * inv;
        }
    }
    return ok;
}

// robust_cholesky (dynamics.py:402-417): plain, then a + 10^i I for i = -6..9; rung = -1,
// 0..15, or 16 when the ladder is exhausted (LinAlgError).  Non-finite input fails every rung
// (scipy check_finite).
SSA_DEV int robust_chol6(const double* A, double* U)
{
    const double JIT[16] = {1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0, 10.0, 100.0, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9};
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 21; ++i) finite = finite && (fabs(A[i]) <= 1.79769313486231570e308);
    if (!finite) return 16;
    if (chol6_upper(A, 0.0, U)) return -1;
    for (int t = 0; t < 16; ++t)
        if (chol6_upper(A, JIT[t], U)) return t;
    return 16;
}

// ---------------------------------------------------------------------------
// H1: hx_aer_erfa (dynamics.py:219) = ecef2aer(M x[:3]) (transformations.py:330-352).
// enu = trans_uvw_ecef (row-major) for the observer.
// Also hands back R, the observer-local cartesian vector az / el / range are formed from: aer2uvw(z) (transformations.py
// :284-297: r cos el cos az, r cos el sin az, r sin el) IS that vector, so the update's mean_z_uvw takes it from here
// instead of going through two more sincos per sigma point.
SSA_DEV void hx_aer_enu(const double* x, const double* M, const double* enu, const double* obs, double* z, double* R)
{
    double xi[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) xi[i] = M[i * 3] * x[0] + M[i * 3 + 1] * x[1] + M[i * 3 + 2] * x[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = xi[i] - obs[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) R[j] = enu[j] * d[0] + enu[3 + j] * d[1] + enu[6 + j] * d[2];
    // elevation as atan2(u, hypot(e, n)) == asin(u / range) (better conditioned towards the zenith than the reference's asin)
    double az = atan2_fast(R[1], R[0]);
    if (az < 0.0) az += TWO_PI;
    z[0] = az;
    z[1] = atan2_fast(R[2], sqrt_fast(R[0] * R[0] + R[1] * R[1]));
    z[2] = sqrt_fast(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}
SSA_DEV void hx_aer(const double* x, const double* M, const double* enu, const double* obs, double* z)
{
    double R[3];
    hx_aer_enu(x, M, enu, obs, z, R);
}
SSA_DEV void aer2uvw(const double* aer, double* uvw)  // transformations.py:284
{
    double sa, ca, se, ce;
    sincos(aer[0], &sa, &ca);
    sincos(aer[1], &se, &ce);
    uvw[0] = aer[2] * ce * ca;
    uvw[1] = aer[2] * ce * sa;
    uvw[2] = aer[2] * se;
}
SSA_DEV void uvw2aer(const double* uvw, double* aer)  // transformations.py:301
{
    double az = atan2_fast(uvw[1], uvw[0]);
    if (az < 0.0) az += TWO_PI;
    aer[0] = az;
    aer[1] = atan2_fast(uvw[2], sqrt_fast(uvw[0] * uvw[0] + uvw[1] * uvw[1]));
    aer[2] = sqrt_fast(uvw[0] * uvw[0] + uvw[1] * uvw[1] + uvw[2] * uvw[2]);
}
SSA_DEV void residual_z_aer(const double* a, const double* b, double* c)  // dynamics.py:260
{
    double d = a[0] - b[0], s, co;
    sincos(d, &s, &co);
    c[0] = atan2(s, co);
    c[1] = a[1] - b[1];
    c[2] = a[2] - b[2];
}

// the same residual for the fused update: atan2(sin d, cos d) is d wrapped into (-pi, pi]; formed directly (exact for
// |d| < pi, where the sin / cos / atan2 route carries 1e-16 of absolute error) it saves a sincos and an atan2 on the
// update's critical path
SSA_DEV void residual_z_aer_wrapped(const double* a, const double* b, double* c)
{
    const double d = a[0] - b[0];
    c[0] = fma(-rint(d * (1.0 / TWO_PI)), TWO_PI, d);
    c[1] = a[1] - b[1];
    c[2] = a[2] - b[2];
}

// 3x3 inverse by cofactors (numpy.linalg.inv in UKF.update); false if singular / non-finite.
SSA_DEV bool inv3(const double* S, double* SI)
{
    double c00 = S[4] * S[8] - S[5] * S[7], c01 = S[5] * S[6] - S[3] * S[8], c02 = S[3] * S[7] - S[4] * S[6];
    double det = S[0] * c00 + S[1] * c01 + S[2] * c02;
    if (det == 0.0 || !(fabs(det) <= 1.79769313486231570e308)) return false;
    double id = 1.0 / det;
    SI[0] = c00 * id;
    SI[1] = (S[2] * S[7] - S[1] * S[8]) * id;
    SI[2] = (S[1] * S[5] - S[2] * S[4]) * id;
    SI[3] = c01 * id;
    SI[4] = (S[0] * S[8] - S[2] * S[6]) * id;
    SI[5] = (S[2] * S[3] - S[0] * S[5]) * id;
    SI[6] = c02 * id;
    SI[7] = (S[1] * S[6] - S[0] * S[7]) * id;
    SI[8] = (S[0] * S[4] - S[1] * S[3]) * id;
    return true;
}

// ---------------------------------------------------------------------------
// wave-level primitives: the 16-lane DPP rows of a wavefront are the "one sigma-point set".
// 64-bit values move as one builtin: two v_mov_b32_dpp for the rotations, ONE v_mov_b64_dpp for a row broadcast
// (row_newbcast is the DPP control gfx90a+ accepts on 64-bit operands); `old` = the source and bound_ctrl set, so no
// preparatory move of the destination is needed.
template <int CTRL>
SSA_DEV double dpp_row(double v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, true);
}
// sum over the 16 lanes of a DPP row, result in every lane (row_ror 8,4,2,1 butterfly).
SSA_DEV double row_allsum(double v)
{
    v += dpp_row<0x128>(v);  // row_ror:8
    v += dpp_row<0x124>(v);  // row_ror:4
    v += dpp_row<0x122>(v);  // row_ror:2
    v += dpp_row<0x121>(v);  // row_ror:1
    return v;
}
// value of lane SRC (0..15) of the own row, in every lane of the row: v_mov_b64_dpp row_newbcast:SRC (no LDS crossbar
// traffic, no lgkmcnt wait -- the former __shfl went through ds_bpermute_b32 twice per double).
template <int SRC>
SSA_DEV double row_bcast(double v)
{
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + SRC, 0xF, 0xF, true);
}

}  // namespace ssa
