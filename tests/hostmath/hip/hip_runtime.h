#pragma once
#include <cmath>
#include <cstdint>
#define __device__
#define __forceinline__ inline
#define __noinline__
#define __global__
static inline double __builtin_amdgcn_rcp(double x) { return (1.0 / x) * (1.0 + 1e-8); }
static inline double __builtin_amdgcn_rsq(double x) { return (1.0 / std::sqrt(x)) * (1.0 - 1e-8); }
static inline bool __all(bool x) { return x; }
static inline bool __any(bool x) { return x; }
static inline unsigned long long __ballot(bool x) { return x ? 1ull : 0ull; }
static inline void __builtin_amdgcn_s_setprio(int) {}

struct dim3_ { unsigned x; };
static dim3_ threadIdx;
template <class T> static inline T __builtin_amdgcn_update_dpp(T a, T b, int, int, int, bool) { return b; }
using std::fabs; using std::fma; using std::sqrt; using std::rint; using std::floor; using std::atan; using std::atan2; using std::acos; using std::asin; using std::tan; using std::sin; using std::cos; using std::exp; using std::log; using std::copysign; using std::sinh; using std::cosh; using std::tanh; using std::atanh; using std::asinh; using std::acosh; using std::pow; using std::fmod; using std::frexp; using std::ldexp;
