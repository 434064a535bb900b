// round-trip latencies of the cross-wavefront primitives the closed-loop kernel chains (agent-scope relaxed atomics), measured
// per wavefront with the 100 MHz wall clock over REP repetitions; G workgroups of one wavefront run concurrently (each on its own
// 128-byte line), so the figure includes the contention of a whole grid doing the same thing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
constexpr int REP = 64;
__global__ void __launch_bounds__(64) rtt_kernel(unsigned long long* buf, unsigned long long* out, int mode, int shared_line)
{
    const int w = blockIdx.x, lane = threadIdx.x;
    unsigned long long* p = buf + (shared_line ? (long long)(w & 63) * 16 : (long long)w * 16);
    unsigned long long acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < REP; ++i) {
        if (mode == 0) {          // returning atomic add, agent scope
            unsigned long long v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += __builtin_amdgcn_readfirstlane((unsigned)v);
        } else if (mode == 1) {   // atomic load, agent scope (dependent chain through acc)
            unsigned long long v = __hip_atomic_load(p + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += __builtin_amdgcn_readfirstlane((unsigned)v) & 1;
        } else if (mode == 2) {   // atomic store + acknowledgement
            if (lane == 0) __hip_atomic_store(p, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (mode == 3) {   // plain load (L2 hit after the first)
            unsigned long long v = *(volatile unsigned long long*)(p + (acc & 1));
            acc += __builtin_amdgcn_readfirstlane((unsigned)v) & 1;
        } else if (mode == 4) {   // returning atomic add, workgroup scope (executes in this XCD's L2)
            unsigned long long v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            acc += __builtin_amdgcn_readfirstlane((unsigned)v);
        } else if (mode == 5) {   // 4 atomic loads in one burst (one part)
            unsigned long long v = __hip_atomic_load(p + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v += __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v += __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v += __hip_atomic_load(p + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += __builtin_amdgcn_readfirstlane((unsigned)v) & 1;
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) { out[2 * w] = t1 - t0; out[2 * w + 1] = acc; }
}
int main(int argc, char** argv)
{
    const char* names[] = {"atomic add (agent, returning)", "atomic load (agent)", "atomic store + ack (agent)", "plain volatile load",
                           "atomic add (workgroup scope)", "4 atomic loads, one burst"};
    unsigned long long *buf, *out;
    const int GMAX = 5120;
    hipMalloc(&buf, GMAX * 128);
    hipMalloc(&out, GMAX * 16);
    std::vector<unsigned long long> h(2 * GMAX);
    for (int shared = 0; shared < 2; ++shared)
        for (int G : {1, 64, 1024, 5000})
            for (int mode = 0; mode < 6; ++mode) {
                hipMemset(buf, 0, GMAX * 128);
                rtt_kernel<<<G, 64>>>(buf, out, mode, shared);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), out, G * 16, hipMemcpyDeviceToHost);
                std::vector<double> t(G);
                for (int i = 0; i < G; ++i) t[i] = h[2 * i] * 10.0 / REP;   // ns per operation
                std::sort(t.begin(), t.end());
                printf("%-34s lines %-9s G=%5d : median %7.0f ns  max %7.0f ns\n", names[mode], shared ? "64 shared" : "own", G, t[G / 2], t[G - 1]);
            }
    return 0;
}
