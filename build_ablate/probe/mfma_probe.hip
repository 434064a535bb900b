// layout + issue-cost probe of the fp64 MFMA forms on gfx950 (diagnostic; build_ablate/probe/run_mfma_probe.py)
#include <hip/hip_runtime.h>
typedef double d4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void probe16(const double* A, const double* B, double* D)
{
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l], B[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}
extern "C" __global__ void probe4(const double* A, const double* B, double* D)
{
    const int l = threadIdx.x;
    double acc = 0;
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], acc, 0, 0, 0);
    D[l] = acc;
}
// cycles per instruction: N back-to-back MFMAs, (a) one dependent accumulator chain, (b) 4 independent accumulators
extern "C" __global__ void time16(const double* A, const double* B, double* D, long long* cyc, int n)
{
    const int l = threadIdx.x;
    const double a = A[l], b = B[l];
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    long long t1 = clock64();
    for (int i = 0; i < n; i += 4) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t2 = clock64();
    D[l] = c0[0] + c1[1] + c2[2] + c3[3];
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
extern "C" __global__ void time4(const double* A, const double* B, double* D, long long* cyc, int n)
{
    const int l = threadIdx.x;
    const double a = A[l], b = B[l];
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    long long t1 = clock64();
    for (int i = 0; i < n; i += 4) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    long long t2 = clock64();
    D[l] = c0 + c1 + c2 + c3;
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
// reference points: the same loop with a dependent v_fma_f64 chain
extern "C" __global__ void timefma(const double* A, const double* B, double* D, long long* cyc, int n)
{
    const int l = threadIdx.x;
    double a = A[l], b = B[l], c0 = 0, c1 = 1, c2 = 2, c3 = 3;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) c0 = __builtin_fma(a, b, c0);
    long long t1 = clock64();
    for (int i = 0; i < n; i += 4) {
        c0 = __builtin_fma(a, c0, b); c1 = __builtin_fma(a, c1, b); c2 = __builtin_fma(a, c2, b); c3 = __builtin_fma(a, c3, b);
    }
    long long t2 = clock64();
    D[l] = c0 + c1 + c2 + c3;
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
