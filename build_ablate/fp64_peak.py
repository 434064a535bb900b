"""fp64 vector FMA micro-benchmark (measure the box's fp64 VALU peak before quoting the datasheet's
78.6 TFLOP/s).  A HIP kernel of 8 independent FMA chains per lane, compiled at run time with hipcc."""
import ctypes as C, os, subprocess, sys, tempfile
import torch
SRC = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void __launch_bounds__(256) fma_chain(double* out, double a, double b, int iters)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
            x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
extern "C" int run(double* out, int blocks, int iters, void* stream)
{
    hipLaunchKernelGGL(fma_chain, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, 0.999999, 1e-9, iters);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "k.hip"), "w").write(SRC)
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", os.path.join(d, "k.so"), os.path.join(d, "k.hip")])
lib = C.CDLL(os.path.join(d, "k.so"))
lib.run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
blocks, iters = 256 * 8, 2000
out = torch.zeros(blocks * 256, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    lib.run(out.data_ptr(), blocks, iters, s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
lib.run(out.data_ptr(), blocks, iters, s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
flop = blocks * 256 * iters * 16 * 8 * 2.0
print("fp64 FMA chains: %.1f TFLOP/s (%.2f ms, %d blocks x 256 threads, 8 chains per lane)" % (flop / ms / 1e9, ms, blocks))
