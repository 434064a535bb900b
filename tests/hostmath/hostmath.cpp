// The per-lane device math of the product (ssa-gym_amd/csrc/ssa_math.hpp) compiled for the HOST: a shim hip/hip_runtime.h
// turns the HIP qualifiers into nothing, wave votes into the lane's own value and the hardware reciprocal estimates into
// 1/x with a 1e-8 relative error (what v_rcp_f64 / v_rsq_f64 deliver before refinement).  CPU tests can then pin the
// propagators' arithmetic against the reference goldens without a GPU (tests/test_device_math_host.py).
#include "hip/hip_runtime.h"
#include "../../ssa-gym_amd/csrc/ssa_math.hpp"
#include "../../ssa-gym_amd/csrc/ssa_conics.hpp"
namespace ssa {   // (the out-of-line wrappers of ssa_kernels.hip)
Vec6 kepler_general_v(Vec6 x, double tof) { Vec6 o; kepler_general_impl(x.v, tof, o.v, nullptr); return o; }
Vec8 kepler_general_diag_v(Vec6 x, double tof, Vec6* out) { Vec8 d; Vec6 o; kepler_general_impl(x.v, tof, o.v, d.v); *out = o; return d; }
}
extern "C" {
// prop 1: SSA_PROP_FG (kepler_fg_fast), prop 0: SSA_PROP_ELEMENTS strong-elliptic path (kepler_elements_fast); ok[i] = handled
void hm_propagate(const double* x, long n, double dt, int prop, double* out, int* ok)
{
    for (long i = 0; i < n; ++i)
        ok[i] = prop == 1 ? ssa::kepler_fg_fast<0>(x + 6 * i, dt, out + 6 * i) : ssa::kepler_elements_fast(x + 6 * i, dt, out + 6 * i);
}
void hm_uv_fast(const double* x, long n, double dt, double* out, int* handled)
{
    for (long i = 0; i < n; ++i) { bool h; ssa::kepler_uv_fast(x + 6 * i, dt, out + 6 * i, h); handled[i] = h; }
}
void hm_uv_general(const double* x, long n, double dt, double* out, int* ok)
{
    for (long i = 0; i < n; ++i) ok[i] = ssa::kepler_uv_general(x + 6 * i, dt, out + 6 * i);
}
// the conic branches beyond the strong-elliptic one (ssa_conics.hpp): libm-level restatement, fast restatement (its bands through
// genf::), the inline strong-hyperbolic tier; ok[i] = the tier accepted the state
void hm_general_libm(const double* x, long n, double dt, double* out, int* ok)
{
    for (long i = 0; i < n; ++i) { ssa::kepler_general_impl(x + 6 * i, dt, out + 6 * i, nullptr); ok[i] = 1; }
}
void hm_general_fast(const double* x, long n, double dt, double* out, int* ok)
{
    for (long i = 0; i < n; ++i) {
        ssa::Vec6 xi; for (int c = 0; c < 6; ++c) xi.v[c] = x[6 * i + c];
        const ssa::Vec6 o = ssa::kepler_general_fast_impl<0>(xi, dt);
        for (int c = 0; c < 6; ++c) out[6 * i + c] = o.v[c];
        ok[i] = 1;
    }
}
void hm_conic_lean(const double* x, long n, double dt, double* out, int* ok)
{
    for (long i = 0; i < n; ++i) {
        for (int c = 0; c < 6; ++c) out[6 * i + c] = __builtin_nan("");
        ok[i] = ssa::kepler_conic_lean<0, false>(x + 6 * i, dt, out + 6 * i);
    }
}
// the band functions alone: nu(t0 + tof) from (nu, ecc, q), fast and libm
void hm_band(const double* nu, const double* ecc, const double* q, long n, double tof, double* fast, double* libm)
{
    for (long i = 0; i < n; ++i) {
        fast[i] = ssa::genf::nu_from_delta_t_band(ssa::genf::delta_t_from_nu_band(nu[i], ecc[i], q[i]) + tof, ecc[i], q[i]);
        libm[i] = ssa::gen::nu_from_delta_t(ssa::gen::delta_t_from_nu(nu[i], ecc[i], ssa::MU, q[i]) + tof, ecc[i], ssa::MU, q[i]);
    }
}
void hm_log_pos(const double* x, long n, double* r) { for (long i = 0; i < n; ++i) r[i] = ssa::genf::log_pos(x[i]); }
int hm_robust_chol6(const double* A21, double* U21) { return ssa::robust_chol6(A21, U21); }
void hm_sincos_fast(const double* x, long n, double* s, double* c) { for (long i = 0; i < n; ++i) ssa::sincos_fast(x[i], s[i], c[i]); }
void hm_atan2_fast(const double* y, const double* x, long n, double* r) { for (long i = 0; i < n; ++i) r[i] = ssa::atan2_fast(y[i], x[i]); }
void hm_exp_fast(const double* x, long n, double* r) { for (long i = 0; i < n; ++i) r[i] = ssa::exp_fast(x[i]); }
void hm_recip(const double* x, long n, double* r, double* rs) { for (long i = 0; i < n; ++i) { r[i] = ssa::rcp_nr(x[i]); rs[i] = ssa::rsqrt_nr(x[i]); } }
}
