// The per-lane device math of the product (ssa-gym_amd/csrc/ssa_math.hpp) compiled for the HOST: a shim hip/hip_runtime.h
// turns the HIP qualifiers into nothing, wave votes into the lane's own value and the hardware reciprocal estimates into
// 1/x with a 1e-8 relative error (what v_rcp_f64 / v_rsq_f64 deliver before refinement).  CPU tests can then pin the
// propagators' arithmetic against the reference goldens without a GPU (tests/test_device_math_host.py).
#include "hip/hip_runtime.h"
#include "../../ssa-gym_amd/csrc/ssa_math.hpp"
namespace ssa {   // the out-of-line complete restatement lives in ssa_kernels.hip (device only): not reachable from these entry points' domains
Vec6 kepler_general_v(Vec6, double) { Vec6 o; for (int i = 0; i < 6; ++i) o.v[i] = __builtin_nan(""); return o; }
Vec8 kepler_general_diag_v(Vec6, double, Vec6*) { Vec8 d; for (int i = 0; i < 8; ++i) d.v[i] = __builtin_nan(""); return d; }
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
int hm_robust_chol6(const double* A21, double* U21) { return ssa::robust_chol6(A21, U21); }
void hm_sincos_fast(const double* x, long n, double* s, double* c) { for (long i = 0; i < n; ++i) ssa::sincos_fast(x[i], s[i], c[i]); }
void hm_atan2_fast(const double* y, const double* x, long n, double* r) { for (long i = 0; i < n; ++i) r[i] = ssa::atan2_fast(y[i], x[i]); }
void hm_exp_fast(const double* x, long n, double* r) { for (long i = 0; i < n; ++i) r[i] = ssa::exp_fast(x[i]); }
void hm_recip(const double* x, long n, double* r, double* rs) { for (long i = 0; i < n; ++i) { r[i] = ssa::rcp_nr(x[i]); rs[i] = ssa::rsqrt_nr(x[i]); } }
}
