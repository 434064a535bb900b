// ssa_kernels.hip -- gfx950 kernels + C ABI (include/ssa_hip.h) for the ssa-gym hot path.
//
// Mapping (DESIGN.md "Kernel"): one 16-lane DPP row of a wavefront owns one object:
//   lane 0      sigma_0 = x
//   lanes 1-6   x + U[k]        lanes 7-12  x - U[k]       (U = robust_cholesky((n+lambda) P), rows)
//   lane 13     the TRUE state of the same object (x_true[i-1] -> x_true[i])
//   lanes 14-15 idle
// so one wavefront advances 4 objects and every Kepler solve of the step (13 m sigma points
// + m true states) runs in its own lane.  State / covariance tiles are staged through LDS
// with block-contiguous (fully coalesced) global loads and stores; the 13-point mean is a
// DPP row reduction; the covariance outer products run from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/ssa_hip.h"
#include "ssa_math.hpp"
#include "ssa_conics.hpp"

namespace ssa {

__device__ __noinline__ Vec6 kepler_general_v(Vec6 x, double tof)
{
    Vec6 o;
    kepler_general_impl(x.v, tof, o.v, nullptr);
    return o;
}
// the step kernels' own instance (inherits their register budget; spills inside it are confined to the rare call)
template <int TAG>
__device__ __noinline__ Vec6 kepler_general_tagged(Vec6 x, double tof)
{
    Vec6 o;
    kepler_general_impl(x.v, tof, o.v, nullptr);
    return o;
}
// the step kernels' own instance (inherits their register budget, see kepler_general_tagged)
template <int TAG>
__device__ __noinline__ Vec6 kepler_general_fast_tagged(Vec6 x, double tof) { return kepler_general_fast_impl<TAG>(x, tof); }
// What the series solver declined, out of line: every conic of the general orientation in the lean form, bands inline (ssa_conics.hpp);
// what THAT declines -- rv2coe's special branches, non-finite input: practically never inside a step -- goes on to the complete
// restatement from here (a nested call: the kernels see ONE call site, as they always did).
template <int TAG>
__device__ __noinline__ Vec6 kepler_beyond_series_tagged(Vec6 x, double tof)
{
    Vec6 o;
    const bool ok = kepler_conic_lean<TAG, false>(x.v, tof, o.v);
    if (__any(!ok)) {
        if (!ok) o = kepler_general_fast_tagged<TAG>(x, tof);
    }
    return o;
}
// the hybrid's lane-level choice: true = the series solver's result stands (strong-elliptic state inside its domain)
SSA_DEV bool kepler_hybrid_fast(const double* s, double tof, double* o)
{
    const double rr = dot3(s, s), vv = dot3(s + 3, s + 3), rv = dot3(s, s + 3);
    const double c1 = vv - MU * rsqrt_nr(rr);
    const double ex = c1 * s[0] - rv * s[3], ey = c1 * s[1] - rv * s[4], ez = c1 * s[2] - rv * s[5];
    const double e2 = (ex * ex + ey * ey + ez * ez) * (1.0 / (MU * MU));
    bool handled;
    const bool ok = kepler_uv_fast(s, tof, o, handled);
    return ok && (e2 < (1.0 - 1e-2) * (1.0 - 1e-2));          // (farnocchia.py:871: ecc < 1 - delta)
}

__device__ __noinline__ Vec8 kepler_general_diag_v(Vec6 x, double tof, Vec6* out)
{
    Vec8 d;
    Vec6 o;
    kepler_general_impl(x.v, tof, o.v, d.v);
    *out = o;
    return d;
}

// ------------------------------------------------------------------------------------------
// fused env step: ONE launch per step over every object.  The common path (plain Cholesky,
// strong-elliptic Kepler) and robust_cholesky's jitter ladder are inline; the other conic branches
// are out-of-line calls taken only by the lanes that need them.  The per-env UKF update runs in the
// row that owns the selected object, hidden among the other waves of the launch.
struct StepK {
    ssa_consts c;
    ssa_step_params p;
};

constexpr int OBJ_PER_WAVE = 4;
constexpr double X_FAILED_POS = 1e20, X_FAILED_VEL = 1e12;  // ssa_tasker_simple_2.py:157-158

// LDS working set of one wavefront (4 objects): 7 600 bytes -- it must stay <= 7 680: with 7 984 only 18 wavefronts fit a CU
// (the allocation granule is larger than 512 bytes) and the 20 000-object step needs a second round (+2 us, measured).
// D holds the centred propagated sigma points d_i = sigma_i' - sigma_0' (i = 1..12) of the four objects as rows of 8
// doubles [d_i[0..5], 1.0, 0.0]: the layout the matrix unit reads its operands from (below); the object blocks start at
// {0, 100, 208, 308} doubles so that the 32 lanes of one ds_read_b64 half hit 32 different 8-byte bank slots.
struct alignas(16) Tiles {
    // The first 1 984 bytes are laid out for the tile's LDS-DMA loads (tile_dma_issue): a global_load_lds_dwordx4 writes
    // LDS at (wave-uniform base) + lane x 16, so one of them with base &P[0] fills P[0..128) from all 64 lanes, and three masked
    // ones with base &P[128] put lanes 0-7 on the tail of P, lanes 32-43 on X (byte 1024 + 32 x 16 = 1536) and lanes 48-59 on T
    // (byte 1024 + 48 x 16 = 1792).  The small members live in the gaps.
    double P[OBJ_PER_WAVE * 36];       // @0     P_in, later P_out
    double Q[36];                      // @1152  process noise (read per covariance entry with a lane-dependent index)
    double Z[8];                       // @1440  six zeros: the "factor row" of the lanes that add nothing (sigma_0, truth, idle)
    int St[OBJ_PER_WAVE];              // @1504
    int Oid[OBJ_PER_WAVE];             // @1520  the caller's index of each row's object (ssa_step_params.obj_ids; else unused)
    double X[OBJ_PER_WAVE * 6];        // @1536  x_in, then sigma_0', then x_out
    double pad1[8];                    // @1728
    double T[OBJ_PER_WAVE * 6];        // @1792  x_true_in, later x_true_out
    double UA[OBJ_PER_WAVE * 36];      // Cholesky factor rows [4][36] (the transform's moment sums stay in registers)
    double D[408];                     // centred propagated sigma points (see above); scratch of the update
    double M[OBJ_PER_WAVE * 12];       // s = Wi sum d_i | m' = mean - sigma_0'
    double Obs[OBJ_PER_WAVE * 12];     // until the update has run: its prefetched inputs (GCRS->ITRS matrix [9] | measurement noise [3]) per object
    double Met[OBJ_PER_WAVE * 4];
};
static_assert(offsetof(Tiles, P) == 0 && offsetof(Tiles, X) == 1536 && offsetof(Tiles, T) == 1792, "LDS-DMA image of the tile (tile_dma_issue)");
SSA_DEV int dbase(int g) { return g * 96 + (g & 1) * 4 + (g >> 1) * 16; }   // 0, 100, 208, 308
static_assert(sizeof(Tiles) <= 7680, "Tiles must fit 20 wavefronts per CU (see above)");

// Wave-contiguous tile I/O: the 4 objects of a wavefront are consecutive, so P / x / x_true / obs are
// single contiguous spans (1152 / 192 / 192 / 384 B) moved as 16-byte lanes -- whole cache lines per
// instruction instead of four 288-byte pieces.  `cnt` = valid objects in the tile (1..4).
// The tile's inputs travel global -> registers -> LDS in two halves, so that a wavefront that advances
// several tiles can have the NEXT tile's loads in flight while it works on the current one:
//   tile_issue : 16-byte wave-contiguous loads into 8 VGPRs.  `main` = P entries [2 lane, 2 lane + 2);
//                `aux` by lane: 0-7 the tail of P | 32-43 x_filter | 48-59 x_true | 60-63 status (bits)
//   tile_commit: registers -> LDS tiles (rows beyond `cnt` are zero / marked failed)
struct TileRegs { double2 main, aux; int st; bool dma; };   // (the status word travels in its own register: packing it into `aux` put a wait for
                                                 // EVERY outstanding load right behind the tile's loads)
// one 16-byte lane of a tile load (plain: marking the inputs streaming was measured slower, 45.4 k vs 46.5 k)
SSA_DEV double2 load16(const double* src) { return *reinterpret_cast<const double2*>(src); }
SSA_DEV void tile_issue_from(TileRegs& r, const double* P_in, const double* x_in, const double* x_true_in, const int32_t* status,
                            int lane, int64_t base, int cnt)
{
    const double2 zero = make_double2(0.0, 0.0);
    r.main = zero;
    r.aux = zero;
    r.st = SSA_ST_PREDICT_NAN;
    r.dma = false;
    if (cnt <= 0) return;
    const double* Pin = P_in + base * 36;
    if (lane < cnt * 18) r.main = load16(Pin + 2 * lane);
    // the rest of the tile -- lanes [0, 8): the tail of P, [32, 44): x, [48, 60): x_true -- is ONE 16-byte load per lane whose
    // source is selected by lane range (a tree of nested lane-range branches cost 80 scalar instructions and their branch
    // latencies in front of the loads); a ragged tile (cnt < 4) shortens every range through `lim`
    const bool sX = lane >= 32, sT = lane >= 48;
    const int i = lane - (sT ? 48 : sX ? 32 : 0);
    const int lim = sX ? cnt * 3 : cnt * 18 - 64;
    const double* src = sT ? x_true_in + base * 6 : sX ? x_in + base * 6 : Pin + 128;
    if (i < lim) r.aux = load16(src + 2 * i);
    if (lane >= 60 && lane - 60 < cnt) r.st = status[base + (lane - 60)];
}
SSA_DEV void tile_issue(TileRegs& r, const ssa_step_params& p, int lane, int64_t base, int cnt)
{
    tile_issue_from(r, p.P_in, p.x_in, p.x_true_in, p.status, lane, base, cnt);
}
SSA_DEV void tile_commit(Tiles& t, const TileRegs& r, int lane)
{
    reinterpret_cast<double2*>(t.P)[lane] = r.main;
    const bool sX = lane >= 32, sT = lane >= 48;
    const int i = lane - (sT ? 48 : sX ? 32 : 0);
    double* dst = sT ? t.T : sX ? t.X : t.P + 128;
    if (i < (sX ? 12 : 8)) reinterpret_cast<double2*>(dst)[i] = r.aux;
    if (lane >= 60) t.St[lane - 60] = r.st;
}
// The one-tile kernels' load: the tile travels global -> LDS directly (LDS-DMA, no VGPR staging, no ds_write pass, the source
// addresses in scalar-base + lane-offset form: four instructions and a handful of scalar ones where the register path spends
// ~80 vector instructions on lane-range selects, 64-bit address arithmetic and the commit).  A ragged last tile (cnt < 4) takes
// the register path, which zero-fills what it does not load.  Completion: s_waitcnt vmcnt(0) (tile_dma_wait) -- the compiler
// does not know these loads write LDS.
typedef const __attribute__((address_space(1))) void* GlobalVoidPtr;
typedef __attribute__((address_space(3))) void* LdsVoidPtr;
SSA_DEV void glds16(const double* src, double* lds_base)
{
    __builtin_amdgcn_global_load_lds((GlobalVoidPtr)src, (LdsVoidPtr)lds_base, 16, 0, 0);
}
SSA_DEV void tile_dma_issue(Tiles& t, TileRegs& r, const double* P_in, const double* x_in, const double* x_true_in, const int32_t* status,
                            int lane, int64_t base, int cnt)
{
    r.dma = cnt == OBJ_PER_WAVE;
    if (!r.dma) {
        tile_issue_from(r, P_in, x_in, x_true_in, status, lane, base, cnt);
        return;
    }
    r.st = SSA_ST_PREDICT_NAN;
    const double* Pin = P_in + base * 36;
    glds16(Pin + 2 * lane, t.P);                                                           // P[0 .. 128)
    if (lane < 8) glds16(Pin + 128 + 2 * lane, t.P + 128);                                 // P[128 .. 144)
    if (lane >= 32 && lane < 44) glds16(x_in + base * 6 - 64 + 2 * lane, t.P + 128);       // -> X: entries 2 (lane - 32) ..
    if (lane >= 48 && lane < 60) glds16(x_true_in + base * 6 - 96 + 2 * lane, t.P + 128);  // -> T: entries 2 (lane - 48) ..
    if (lane >= 60) r.st = status[base + (lane - 60)];
}
SSA_DEV void tile_dma_wait(Tiles& t, const TileRegs& r, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane >= 60) t.St[lane - 60] = r.st;
}
typedef double v2d __attribute__((ext_vector_type(2)));
// one 16-byte lane of a tile store.  NT (non-temporal): for launches of up to 20 480 objects and for rollouts the
// outputs are marked streaming, which leaves less dirty data for the end-of-kernel L2 write-back (-0.55 us per launch
// at 20 000 objects).  Not for the multi-tile step kernel: at 160 000 objects the next step re-reads 61 MB of these
// outputs from the Infinity Cache, and streaming them past it cost 19 %.
template <bool NT>
SSA_DEV void store16(double* dst, const double* src)
{
#ifdef SSA_PLAIN_STORES   // diagnostic: never stream
    *reinterpret_cast<v2d*>(dst) = *reinterpret_cast<const v2d*>(src);
#else
#ifdef SSA_NT_ALWAYS   // diagnostic: always stream
    __builtin_nontemporal_store(*reinterpret_cast<const v2d*>(src), reinterpret_cast<v2d*>(dst));
#else
    if (NT) __builtin_nontemporal_store(*reinterpret_cast<const v2d*>(src), reinterpret_cast<v2d*>(dst));
    else *reinterpret_cast<v2d*>(dst) = *reinterpret_cast<const v2d*>(src);
#endif
#endif
}
// first row of the env a tile belongs to (a storage layout with several envs: whole tiles per env, so the tile has ONE env)
SSA_DEV int64_t env_row0(const ssa_step_params& p, int64_t base)
{
    return p.n_env > 1 ? (int64_t)((uint32_t)base / (uint32_t)p.n_obj) * p.n_obj : 0;
}
template <bool NT>
SSA_DEV void store_tile(const Tiles& t, const ssa_step_params& p, int lane, int64_t base, int cnt)
{
#ifdef SSA_SKIP_STORES   // diagnostic builds (write-traffic attribution): bit 1 = P, 2 = obs, 4 = x / x_true, 8 = metrics, 16 = status
#define SSA_SKIP(b) ((SSA_SKIP_STORES) & (b))
#else
#define SSA_SKIP(b) 0
#endif
    if (cnt == OBJ_PER_WAVE) {
        // A whole tile (wave-uniform; every tile of a launch but a ragged last one).  Every group of lanes stores from
        // (uniform pointer) + lane x 16 bytes on both sides -- the global address in scalar-base + lane-offset form, the LDS address
        // one shared register plus an immediate -- instead of selecting 64-bit destination and source pointers per lane range
        // (that was 58 vector + 60 scalar instructions per wavefront).  The observation rows [x, diag P] (results.py:61) are read
        // from the state / covariance tiles by the lanes that store them: no staging copy.
        if (!SSA_SKIP(1)) {
            store16<NT>(p.P_out + base * 36 + 2 * lane, t.P + 2 * lane);
            if (lane < 8) store16<NT>(p.P_out + base * 36 + 128 + 2 * lane, t.P + 128 + 2 * lane);
        }
        if (!SSA_SKIP(4)) {
            if ((unsigned)(lane - 8) < 12u) store16<NT>(p.x_out + base * 6 - 16 + 2 * lane, t.X - 16 + 2 * lane);
            if ((unsigned)(lane - 20) < 12u) store16<NT>(p.x_true_out + base * 6 - 40 + 2 * lane, t.T - 40 + 2 * lane);
        }
        if (!SSA_SKIP(2) && lane >= 32 && lane < 56) {   // obs: 24 lanes x 16 bytes; lane i = entries [2 i, 2 i + 2) of the tile's 48
            const int i = lane - 32;
            const int jj = (i * 43) >> 8;                // i / 6: the object
            const int r = 2 * i - 12 * jj;               // 0, 2, 4: x | 6, 8, 10: diag P
            double* dst = p.obs + base * 12 - 64 + 2 * lane;
            v2d pair;
            if (r < 6) pair = *reinterpret_cast<const v2d*>(t.X + jj * 6 + r);
            else {
                const double* d = t.P + jj * 36 + 7 * (r - 6);
                pair = v2d{d[0], d[7]};
            }
            if (NT) __builtin_nontemporal_store(pair, reinterpret_cast<v2d*>(dst));
            else *reinterpret_cast<v2d*>(dst) = pair;
            if (p.obs_mirror) {   // (e.g. host-mapped memory; with obj_ids: row by row at the caller's index of the object)
                // (in observation entries; several envs: the table holds indices within the env, whose rows start at env_row0)
                const int64_t at = p.obj_ids ? (env_row0(p, base) + t.Oid[jj]) * 12 + r : base * 12 - 64 + 2 * lane;
                if (p.launch_mask & SSA_LAUNCH_MIRROR_F32)      // the host-facing copy in single precision: half the bytes over PCIe
                    *reinterpret_cast<float2*>(reinterpret_cast<float*>(p.obs_mirror) + at) = make_float2((float)pair.x, (float)pair.y);
                else *reinterpret_cast<v2d*>(p.obs_mirror + at) = pair;
            }
        }
        if (!SSA_SKIP(16) && lane >= 56 && lane < 60) p.status[base - 56 + lane] = t.St[lane - 56];
        if (!SSA_SKIP(8) && lane < 16) {   // metrics [E][4][m]: four 32-byte runs per tile
            const int kk = lane >> 2, jj = lane & 3;
            const int64_t obj = base + jj;
            if (p.n_env == 1) {   // (scalar row stride, one 64-bit multiply-add per lane)
                p.metrics[(int64_t)kk * p.n_obj + obj] = t.Met[jj * 4 + kk];
            } else {
                const int64_t e = (int64_t)((uint32_t)obj / (uint32_t)p.n_obj), j = obj - e * p.n_obj;
                p.metrics[(e * 4 + kk) * p.n_obj + j] = t.Met[jj * 4 + kk];
            }
        }
        return;
    }
    // ragged last tile: the general form (the observation rows were staged in t.Obs by observe_rows)
    if (!SSA_SKIP(1) && lane < cnt * 18) store16<NT>(p.P_out + base * 36 + 2 * lane, t.P + 2 * lane);
    {   // lanes [0, 12): x | [16, 28): x_true | [32, 40): the tail of P | [40, 64): obs -- one 16-byte store per lane, destination and
        // LDS source selected by lane range (see tile_issue_from)
        const bool sT = lane >= 16, sP = lane >= 32, sO = lane >= 40;
        const int i = lane - (sO ? 40 : sP ? 32 : sT ? 16 : 0);
        const int lim = sO ? cnt * 6 : sP ? cnt * 18 - 64 : cnt * 3;
        double* dst = sO ? p.obs + base * 12 : sP ? p.P_out + base * 36 + 128 : sT ? p.x_true_out + base * 6 : p.x_out + base * 6;
        const double* src = sO ? t.Obs : sP ? t.P + 128 : sT ? t.T : t.X;
        const bool skip = sO ? SSA_SKIP(2) : sP ? SSA_SKIP(1) : SSA_SKIP(4);
        if (!skip && i < lim) store16<NT>(dst + 2 * i, src + 2 * i);
        if (sO && p.obs_mirror && i < lim) {
            const int jj = (i * 43) >> 8;     // i / 6: the object (rows beyond cnt are masked by `lim`)
            const int64_t at = p.obj_ids ? (env_row0(p, base) + t.Oid[jj]) * 12 + (2 * i - 12 * jj) : base * 12 + 2 * i;
            if (p.launch_mask & SSA_LAUNCH_MIRROR_F32)
                *reinterpret_cast<float2*>(reinterpret_cast<float*>(p.obs_mirror) + at) = make_float2((float)src[2 * i], (float)src[2 * i + 1]);
            else store16<false>(p.obs_mirror + at, src + 2 * i);
        }
    }
    if (!SSA_SKIP(16) && lane >= 12 && lane < 16) {
        const int i = lane - 12;
        if (i < cnt) p.status[base + i] = t.St[i];
    }
    if (!SSA_SKIP(8) && lane < 16) {   // metrics [E][4][m]: four 32-byte runs per tile
        const int kk = lane >> 2, jj = lane & 3;
        if (jj < cnt) {
            const int64_t obj = base + jj;
            if (p.n_env == 1) {   // (scalar row stride, one 64-bit multiply-add per lane)
                p.metrics[(int64_t)kk * p.n_obj + obj] = t.Met[jj * 4 + kk];
            } else {
                const int64_t e = (int64_t)((uint32_t)obj / (uint32_t)p.n_obj), j = obj - e * p.n_obj;
                p.metrics[(e * 4 + kk) * p.n_obj + j] = t.Met[jj * 4 + kk];
            }
        }
    }
}

// O1/O2 for the row's object from the tiles (results.py:61, :37)
SSA_DEV void observe_rows(Tiles& t, int g, int l, bool stage_obs)
{
    // (a whole tile's observation rows leave straight from the state / covariance tiles: store_tile)
    if (stage_obs && l < 12) t.Obs[g * 12 + l] = (l < 6) ? t.X[g * 6 + l] : t.P[g * 36 + 7 * (l - 6)];
    if (l < 4) {
        const int off = (l & 1) * 3;  // 0: position block, 1: velocity block
        double v;
        if (l < 2) {
            double a0 = t.X[g * 6 + off] - t.T[g * 6 + off];
            double a1 = t.X[g * 6 + off + 1] - t.T[g * 6 + off + 1];
            double a2 = t.X[g * 6 + off + 2] - t.T[g * 6 + off + 2];
            v = sqrt_fast(a0 * a0 + a1 * a1 + a2 * a2);
        } else {   // (a negative sum -- indefinite covariance -- gives NaN, as numpy's sqrt)
            v = sqrt_fast(t.P[g * 36 + 7 * off] + t.P[g * 36 + 7 * (off + 1)] + t.P[g * 36 + 7 * (off + 2)]);
        }
        t.Met[g * 4 + l] = v;
    }
}

// U3 on the matrix unit.  With D^(g) the 12 x 8 matrix of object g's rows [d_i, 1, 0], the transform needs
//   A^(g) = D^T D :  A[a][b] = sum_i d_i[a] d_i[b]  (a, b < 6)   and   A[a][6] = sum_i d_i[a]
// -- 21 row reductions over the 16 lanes plus 6 more for the mean if done with DPP butterflies.  v_mfma_f64_4x4x4_4b_f64
// multiplies four independent 4x4x4 blocks per instruction; its block index is lane bits 3:2, its k (operands) / i
// (result) index lane bits 5:4, its i / j index lane bits 1:0 (measured, build_ablate/probe): block b = object b, three
// k-chunks of four sigma points, 4x4 tiles (0,0), (0,1), (1,1) of the symmetric 8x8 result = 9 instructions of 16 cycles
// on the otherwise idle matrix pipe, operands by six conflict-free ds_read_b64.  Lane (hi, mid, lo) ends up with
// A^(mid)[hi][lo], A^(mid)[hi][4 + lo], A^(mid)[4 + hi][4 + lo] and KEEPS them in registers: covariance_finish() below turns
// them into the entries of P in the same lanes.  Only the column of sums A[a][6] goes to LDS (t.M[mid][a]: the mean needs it).
struct Moments { double c00, c01, c11; };
template <bool SUMS_ONLY = false>   // SUMS_ONLY: the column of sums alone (SSA_FLAG_REFERENCE_COV forms its own second moments): block (0,0) is skipped
SSA_DEV Moments moment_sums_mfma(Tiles& t, int lane)
{
    const int hi = lane >> 4, mid = (lane >> 2) & 3, lo = lane & 3;
    const double* src = &t.D[dbase(mid) + hi * 8 + lo];
    double a0[3], a1[3];
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) {
        a0[kc] = src[kc * 32];
        a1[kc] = src[kc * 32 + 4];
    }
    Moments mo = {0.0, 0.0, 0.0};
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) {
        if (!SUMS_ONLY) mo.c00 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[kc], a0[kc], mo.c00, 0, 0, 0);
        mo.c01 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[kc], a1[kc], mo.c01, 0, 0, 0);
        mo.c11 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[kc], a1[kc], mo.c11, 0, 0, 0);
    }
    if (lo == 2) {                                   // column 6 of the blocks (0,1) and (1,1): the sums over the sigma points
        t.M[mid * 12 + hi] = mo.c01;
        if (hi < 2) t.M[mid * 12 + 4 + hi] = mo.c11;
    }
    return mo;
}

// U3 (second half): P = sum Wc_i y_i y_i^T + Q with y_i = sigma_i' - x, expanded around sigma_0':
//   P = Wi A - m' s^T - s m'^T + sum(Wc) m' m'^T + Q       (A in the registers moment_sums_mfma left; s, m' in t.M)
// in the matrix unit's result layout: lane (hi, mid, lo) finishes object mid's entries (hi, lo), (hi, 4 + lo), (4 + hi, 4 + lo)
// -- all 64 lanes, no index arithmetic beyond the lane's bit fields, no second LDS copy of A (the earlier version ran the
// 21 upper-triangle entries over the object's 16 lanes in two passes, half of its instructions index arithmetic).  A is
// bit-symmetric (same products, same order) but the expression below is not under a <-> b, so only lanes with a <= b
// write, to both [a][b] and [b][a]: P leaves exactly symmetric.
SSA_DEV double covariance_entry(const ssa_consts& C, double acc, double sa_, double ma, double sb_, double mb, double q)
{
    return C.Wi * acc - ma * sb_ - sa_ * mb + C.sum_wc * ma * mb + q;
}
SSA_DEV void covariance_finish(Tiles& t, const ssa_consts& C, int lane, const Moments& mo)
{
    const int hi = lane >> 4, mid = (lane >> 2) & 3, lo = lane & 3;
    const int hi1 = hi & 1, lo1 = lo & 1;            // (rows / columns 4, 5: the lanes beyond them compute on valid addresses and do not write)
    const double* M = &t.M[mid * 12];
    double* P = &t.P[mid * 36];
    const double s_a = M[hi], m_a = M[6 + hi], s_b = M[lo], m_b = M[6 + lo];
    const double s_a4 = M[4 + hi1], m_a4 = M[10 + hi1], s_b4 = M[4 + lo1], m_b4 = M[10 + lo1];
    const double p00 = covariance_entry(C, mo.c00, s_a, m_a, s_b, m_b, t.Q[hi * 6 + lo]);
    const double p01 = covariance_entry(C, mo.c01, s_a, m_a, s_b4, m_b4, t.Q[hi * 6 + 4 + lo1]);
    const double p11 = covariance_entry(C, mo.c11, s_a4, m_a4, s_b4, m_b4, t.Q[(4 + hi1) * 6 + 4 + lo1]);
    if (hi <= lo) {
        P[hi * 6 + lo] = p00;
        P[lo * 6 + hi] = p00;
    }
    if (lo < 2) {
        P[hi * 6 + 4 + lo] = p01;
        P[(4 + lo) * 6 + hi] = p01;
        if (hi <= lo) {                              // (hi, lo) in {(0,0), (0,1), (1,1)}
            P[(4 + hi) * 6 + 4 + lo] = p11;
            P[(4 + lo) * 6 + 4 + hi] = p11;
        }
    }
}

// O3 accumulator.  NaN ranks above everything (np.max / np.argmax semantics), ties keep the lowest index.
struct StatAcc {
    double mx, sm;       // max delta_pos (NaN excluded), max sigma_pos (valid when !sm_nan)
    long long arg;       // index of sm (or of the first NaN sigma_pos)
    unsigned c4, c7, nf;
    int mx_nan, sm_nan;
};
SSA_DEV void stat_merge(StatAcc& a, const StatAcc& b)
{
    a.mx = fmax(a.mx, b.mx);
    a.mx_nan |= b.mx_nan;
    a.c4 += b.c4; a.c7 += b.c7; a.nf += b.nf;
    bool take_b;
    if (a.sm_nan || b.sm_nan) take_b = b.sm_nan && (!a.sm_nan || b.arg < a.arg);
    else take_b = (b.sm > a.sm) || (b.sm == a.sm && b.arg < a.arg);
    if (take_b) { a.sm = b.sm; a.arg = b.arg; a.sm_nan = b.sm_nan; }
}
SSA_DEV StatAcc stat_shfl_down(const StatAcc& a, int off)
{
    StatAcc b;
    b.mx = __shfl_down(a.mx, off, 64);
    b.sm = __shfl_down(a.sm, off, 64);
    b.arg = __shfl_down(a.arg, off, 64);
    b.c4 = __shfl_down(a.c4, off, 64);
    b.c7 = __shfl_down(a.c7, off, 64);
    b.nf = __shfl_down(a.nf, off, 64);
    b.mx_nan = __shfl_down(a.mx_nan, off, 64);
    b.sm_nan = __shfl_down(a.sm_nan, off, 64);
    return b;
}
SSA_DEV StatAcc stat_identity()
{
    StatAcc a;
    a.mx = -1.0; a.sm = -1.0; a.arg = 0x7fffffffffffffffLL; a.c4 = a.c7 = a.nf = 0; a.mx_nan = a.sm_nan = 0;
    return a;
}
SSA_DEV StatAcc stat_wave_reduce(StatAcc a)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        StatAcc b = stat_shfl_down(a, off);
        stat_merge(a, b);
    }
    return a;
}

// ------------------------------------------------------------------------------------------
// Intra-wave LDS hand-off.  The step kernel's workgroup IS one wavefront, all lanes run in
// lockstep and the LDS queue is in order, so "every lane's earlier LDS writes are visible to every
// lane's later LDS reads" only needs the compiler not to move accesses across this point and the
// outstanding DS operations to have completed.  (Usable inside row-divergent branches, unlike a
// workgroup barrier.)
SSA_DEV void wave_lds_sync()
{
#ifdef SSA_LDS_WAIT   // the former form: additionally drains the DS queue (not needed, see above)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    // wavefront-scope fence: no instruction on gfx9 (the DS unit executes one wavefront's operations in
    // issue order, so a lane's read issued after another lane's write observes it); it only pins the
    // compiler's ordering of the LDS accesses around this point
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#endif
}

// U3, covariance in the REFERENCE's own arithmetic (SSA_FLAG_REFERENCE_COV): filterpy's unscented_transform evaluates
//   P = sum_i (sigma_i' - x) (Wc_i (sigma_i' - x))^T + Q,      i = 0 .. 12,  Wc_0 ~ -2e8 at alpha = 1e-4
// -- a sum of thirteen outer products whose first term cancels the other twelve.  For a filter whose prior has diverged
// (|sigma_0' - x| ~ 1e9 m late in a predict-only episode) that cancellation leaves rounding noise of 1e10 m^2 in P: the
// covariance turns indefinite, robust_cholesky (dynamics.py:402-417) exhausts its ladder at the next predict and the
// reference marks the filter FAILED (ssa_tasker_simple_2.py:271-285, 369-382): 2-3 % of the filters of a 480-step
// episode.  covariance_finish() above evaluates the same matrix expanded around sigma_0' without that cancellation and
// such filters survive; this form reproduces the reference's failure behaviour (tests/test_episode_failures.py).
// The thirteen rows y_i = sigma_i' - x go to LDS (factor tile + D, contiguous and free by now), the matrix unit forms the
// three 4x4 tiles of sum_i y_i (Wc_i y_i)^T, i = 0 .. 11, in three k-chunks and three vector fmas add point 12; the weight enters
// with the right operand, as in the reference.  (What the form costs -- every wavefront of a SIMD runs this stage at the same time, so
// its instructions add up: a vector instruction ~8 ns of a 20 000-object step, a 4x4x4 matrix instruction ~33 ns; 13.0 us per healthy
// step against 11.9 with covariance_finish: build_ablate/healthy_phase_ab.py.)
SSA_DEV int ybase(int g) { return g * 104 + (g & 1) * 28 + (g >> 1) * 32; }   // 0, 132, 240, 372: bank slots as dbase()
// (xb_l: lanes 0 .. 5 of a row hold component l of the row's prior mean -- the value they have just written to t.X.  The other lanes take
// it from there by DPP row broadcasts: the detour through t.X was three dependent LDS round trips in front of the y rows' own)
SSA_DEV void covariance_reference(Tiles& t, const ssa_consts& C, int lane, const double (&o)[6], double xb_l)
{
    static_assert(372 + 16 * 8 <= OBJ_PER_WAVE * 36 + 408, "y rows fit the factor tile + D");
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    double* const Y = t.UA;
    {
        const int g = lane >> 4, l = lane & 15;
#ifdef SSA_COV_MEAN_FROM_LDS   // (diagnostic: the former form)
        const double* xb = &t.X[g * 6];
        const double x0 = xb[0], x1 = xb[1], x2 = xb[2], x3 = xb[3], x4 = xb[4], x5 = xb[5];
#else
        const double x0 = row_bcast<0>(xb_l), x1 = row_bcast<1>(xb_l), x2 = row_bcast<2>(xb_l), x3 = row_bcast<3>(xb_l),
                     x4 = row_bcast<4>(xb_l), x5 = row_bcast<5>(xb_l);
#endif
        if (l <= 12) {
            v2d_t* dst = reinterpret_cast<v2d_t*>(&Y[ybase(g) + l * 8]);
            dst[0] = v2d_t{o[0] - x0, o[1] - x1};
            dst[1] = v2d_t{o[2] - x2, o[3] - x3};
            dst[2] = v2d_t{o[4] - x4, o[5] - x5};
            dst[3] = v2d_t{0.0, 0.0};
        }
    }
    wave_lds_sync();
    const int hi = lane >> 4, mid = (lane >> 2) & 3, lo = lane & 3;
    const double* src = &Y[ybase(mid) + hi * 8 + lo];
    Moments mo = {0.0, 0.0, 0.0};
#ifndef SSA_COV_FOUR_CHUNKS
    constexpr int NKC = 3;
#else
    constexpr int NKC = 4;
#endif
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        double a0 = src[kc * 32], a1 = src[kc * 32 + 4];
        if (kc == 3 && hi != 0) { a0 = 0.0; a1 = 0.0; }           // rows 13 .. 15 do not exist
        const double w = (kc == 0 && hi == 0) ? C.Wc0 : C.Wi;
        const double b0 = w * a0, b1 = w * a1;
        mo.c00 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, mo.c00, 0, 0, 0);
        mo.c01 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b1, mo.c01, 0, 0, 0);
        mo.c11 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, mo.c11, 0, 0, 0);
    }
#ifndef SSA_COV_FOUR_CHUNKS
    {   // point 12, the last term of the sum, by three vector fmas: the fourth k-chunk held it alone (three matrix instructions that
        // multiplied three rows of zeros) -- entry (i, j) += y12[i] (Wi y12[j]), the product rounded as the matrix unit's operand was;
        // whole 20 000-object episodes bit-identical to the four-chunk form (build_ablate/r04_run52.sh), 0.2 us per step less with
        // the broadcasts above
        const double* y12 = &Y[ybase(mid) + 12 * 8];
        const double ai = y12[hi], ai4 = y12[4 + hi], bj = C.Wi * y12[lo], bj4 = C.Wi * y12[4 + lo];
        mo.c00 = fma(ai, bj, mo.c00);
        mo.c01 = fma(ai, bj4, mo.c01);
        mo.c11 = fma(ai4, bj4, mo.c11);
    }
#endif
    const int hi1 = hi & 1, lo1 = lo & 1;
    double* P = &t.P[mid * 36];
    const double p00 = mo.c00 + t.Q[hi * 6 + lo];
    const double p01 = mo.c01 + t.Q[hi * 6 + 4 + lo1];
    const double p11 = mo.c11 + t.Q[(4 + hi1) * 6 + 4 + lo1];
    // scipy's cholesky reads the upper triangle only; the entries a <= b are mirrored (as covariance_finish)
    if (hi <= lo) {
        P[hi * 6 + lo] = p00;
        P[lo * 6 + hi] = p00;
    }
    if (lo < 2) {
        P[hi * 6 + 4 + lo] = p01;
        P[(4 + lo) * 6 + hi] = p01;
        if (hi <= lo) {
            P[(4 + hi) * 6 + 4 + lo] = p11;
            P[(4 + lo) * 6 + 4 + hi] = p11;
        }
    }
}

// U2 (common case): upper Cholesky of scale*P for the row's object, lane-distributed: lane c owns column c of the factor in
// registers; step j (LAPACK dpotf2('U') order) needs column j's finished entries U[i][j], i < j, and the pivot -- all held
// by lane j -- in every lane, which is one v_mov_b64_dpp row_newbcast:j each (21 per factorisation, no LDS round trip, no
// wait; the earlier version went through LDS and paid a write -> read hand-off per pivot).  Lane j forms the pivot from its
// own column with the same operations as before (same bits).  The factor rows land in t.U at the end (the sigma points
// need ROWS of U: lane l reads row (l-1)%6, a transposition of the register layout).
// Returns false (row-uniform) when a pivot is <= 0 / NaN or P holds a non-finite entry -> the ladder.
template <int J>
SSA_DEV void chol_step(const double (&a)[6], double (&uc)[6], int lc, bool& ok)
{
    double v = a[J];                          // A[J][lc] (meaningful for lc >= J; lane J: the diagonal)
#pragma unroll
    for (int i = 0; i < J; ++i) {
        const double uij = row_bcast<J>(uc[i]);   // U[i][J]
        v = fma(-uij, uc[i], v);                  // lane J: ajj -= U[i][J]^2 ; lane c > J: A[J][c] -= U[i][J] U[i][c]
    }
    const double y = row_bcast<J>(rsqrt_nr(v));   // 1 / sqrt(pivot) of lane J; NaN / inf when the pivot is <= 0 or NaN
    // a bad pivot poisons everything downstream (row J becomes inf / NaN, every later pivot in its column picks up -inf or NaN),
    // so the positivity test of the LAST pivot covers all six
    if (J == 5) ok = ok && (y > 0.0) && (y <= 1.79769313486231570e308);
    uc[J] = (lc >= J) ? v * y : 0.0;              // lane J: ajj / sqrt(ajj) = the diagonal entry
}
// One entry of the matrix a rung factorises, in the REFERENCE's two roundings: sigma_points() hands robust_cholesky the product
// (n + lambda) P already rounded, and the ladder adds its jitter to that (dynamics.py:410: cholesky(a + e)).  Written out with an
// optimisation barrier between the two: under -ffp-contract=fast the compiler fused scale * p + jit into one fma at some call sites and -- where it could
// share scale * p with a neighbouring factorisation of the same matrix -- not at others.  One rounding more or less in a diagonal entry
// is nothing, except for the matrices the ladder exists for: (n + lambda) P of a diverged filter has condition 1e20, the factor's last
// rows move by 1e-8 relative with that bit, and two builds of the same source parted over an episode (round 3's "two-pass" ladder
// picked the SAME rungs as the sequential one -- ssa_ladder_probe_f64, tests/golden/ladder_illconditioned_tile.npz -- and still left
// different filters: its second factorisation was the unfused instance).  Now every instance is the reference's arithmetic.
SSA_DEV double scaled_entry(double scale, double pv, double jit)
{
    double sp = scale * pv;
    SSA_OPAQUE(sp);              // (two roundings: the product is a value of its own before the jitter is added)
    return sp + jit;
}
// factorises the matrix at Pg (row-major 6x6 in LDS) in the calling row `grow`; the factor's column lc stays in uc[]
template <bool PLAIN>   // PLAIN: the jitter-free first attempt (no diagonal select / add on the common path)
SSA_DEV bool chol_row_regs(const double* Pg, double scale, double jit, int grow, int l, double (&uc)[6])
{
    const int lc = l < 6 ? l : 5;   // lanes 6..15 shadow column 5 (their stores are masked)
    // column lc of the UPPER triangle of scale*P + jit*I (scipy.linalg.cholesky reads the upper triangle only); the six
    // owning lanes see all 36 entries between them: scipy's check_finite
    double a[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double pv = Pg[j * 6 + lc];
        ok = ok && (fabs(pv) <= 1.79769313486231570e308);
        a[j] = (PLAIN || lc != j) ? scale * pv : scaled_entry(scale, pv, jit);
    }
    ok = ((__ballot(!ok) >> (grow * 16)) & 0xFFFFull) == 0;
    chol_step<0>(a, uc, lc, ok);
    chol_step<1>(a, uc, lc, ok);
    chol_step<2>(a, uc, lc, ok);
    chol_step<3>(a, uc, lc, ok);
    chol_step<4>(a, uc, lc, ok);
    chol_step<5>(a, uc, lc, ok);
    return ok;
}
SSA_DEV void chol_store_rows(double* Ug, const double (&uc)[6], int l)
{
    if (l < 6) {
#pragma unroll
        for (int j = 0; j < 6; ++j) Ug[j * 6 + l] = uc[j];
    }
}

// robust_cholesky (dynamics.py:402-417) for the four objects of the wavefront: plain attempt, then a + 10^i I for
// i = -6..9 (first success wins); returns the calling row's -1, 0..15, or 16 (LinAlgError).
// The plain attempt runs row-parallel (each row its own object).  Rows that fail are then served ONE AT A TIME BY THE
// WHOLE WAVEFRONT: the four rows try four consecutive rungs of that object's ladder at once and the lowest successful
// one wins -- the same answer as the reference's sequential ladder in a quarter of the factorisations (a diverged
// filter late in an episode needs rung 10-15: without this its wavefront ran 11-16 factorisations back to back and held
// the end of the launch, build_ablate/wave_timeline.py).
__constant__ double JITTER[16] = {1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0, 10.0, 100.0, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9};
// Does scale * P + jit * I factorise?  ONE lane decides, alone, in its own registers: the ladder's unit of work when every lane
// of a row tries its own rung (robust_chol_row_lds below).  Operation by operation the arithmetic of chol_row_regs<false> --
// the same matrix entries (scaled_entry), the same fused multiply-adds in the same order, the same refined reciprocal square
// root -- so a rung succeeds here exactly when it succeeds there (with round 3's fused diagonal the one-pass ladder reproduced the
// round-3 library bit for bit over 20 000-object episodes of all three propagators: profiles/r04_ab_ladder_against_round3.txt).
// Only the verdict leaves: an entry U[i][c] is dead once step c has used it, so at most nine entries are live at a time
// (a lane that kept its whole factor, 42 registers, made the closed-loop kernels spill); the winning rung's factor is then
// formed once more by the row (chol_row_regs<false>), which is where it is needed in the row-distributed layout anyway.
// The matrix is read from LDS where it is used (the lanes of a row read the same address: a broadcast).
SSA_DEV bool chol_lane_ok(const double* Pg, double scale, double jit)
{
    double U[21];
    double y = 0.0;
#pragma unroll
    for (int J = 0; J < 6; ++J) {
        double vp = scaled_entry(scale, Pg[J * 6 + J], jit);
#pragma unroll
        for (int i = 0; i < J; ++i) vp = fma(-U[tri(i, J)], U[tri(i, J)], vp);
        y = rsqrt_nr(vp);               // NaN / inf when the pivot is <= 0 or NaN: poisons everything behind it (see chol_step)
#pragma unroll
        for (int c = J + 1; c < 6; ++c) {
            double v = scale * Pg[J * 6 + c];
#pragma unroll
            for (int i = 0; i < J; ++i) v = fma(-U[tri(i, J)], U[tri(i, c)], v);
            U[tri(J, c)] = v * y;
        }
    }
    return (y > 0.0) && (y <= 1.79769313486231570e308);
}
SSA_DEV int robust_chol_row_lds(Tiles& t, double scale, int g, int l)
{
    double uc[6];
    int rung = 16;
    {
        const bool ok = chol_row_regs<true>(&t.P[g * 36], scale, 0.0, g, l, uc);
        if (ok) {
            chol_store_rows(&t.UA[g * 36], uc, l);
            rung = -1;
        }
        const unsigned long long bad = __ballot(!ok);
        wave_lds_sync();
        if (bad == 0ull) return rung;                  // the common case, wave-uniform
        __builtin_amdgcn_s_setprio(3);                 // a straggler in the making: issue priority for the rest of its life
#ifndef SSA_LADDER_BY_PASSES
        // The ladder in ONE pass, every failed row at once: lane l of a row decides, whole and alone (chol_lane_ok), whether its
        // row's matrix factorises with rung l's jitter, so the sixteen rungs of up to four objects are tried side by side; the
        // FIRST rung that succeeds -- the reference's sequential answer, dynamics.py:406-414, also where success is not monotone
        // in the jitter: every rung is actually tried -- is the lowest set bit of the row's ballot, and the row forms that rung's
        // factor.  Two factorisations' latency whatever the rung and however many of the four objects need the ladder (before:
        // the rows took turns, four rungs per pass: a wavefront with four diverged objects at rungs 12-15 ran sixteen passes back
        // to back and held the end of a late-episode launch, build_ablate/wave_timeline.py).
        {
            const bool mine = ((bad >> (g * 16)) & 1ull) != 0ull;   // row-uniform
            if (mine) {
                const double* Pg = &t.P[g * 36];
                const double jit = JITTER[l];
                bool finite = true;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int idx = l + 16 * r;
                    if (idx < 36) finite = finite && (fabs(Pg[idx]) <= 1.79769313486231570e308);
                }
                finite = ((__ballot(!finite) >> (g * 16)) & 0xFFFFull) == 0ull;      // scipy's check_finite: the whole matrix
                const bool okr = chol_lane_ok(Pg, scale, jit) && finite;
                const unsigned won = (unsigned)((__ballot(okr) >> (g * 16)) & 0xFFFFull);
                if (won != 0u) {
                    rung = __ffs((int)won) - 1;
                    const double jw = __shfl(jit, g * 16 + rung, 64);            // the winning rung's jitter, from the lane that tried it
                    chol_row_regs<false>(Pg, scale, jw, g, l, uc);               // (succeeds: the same arithmetic said so)
                    chol_store_rows(&t.UA[g * 36], uc, l);
                }
            }
            wave_lds_sync();
            return rung;
        }
#endif
        for (int gf = 0; gf < OBJ_PER_WAVE; ++gf) {    // wave-uniform loop over the rows that failed (the former form: -DSSA_LADDER_BY_PASSES)
            if (((bad >> (gf * 16)) & 1ull) == 0ull) continue;
            const double* Pg = &t.P[gf * 36];
            bool finite = true;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int idx = l + 16 * r;
                if (idx < 36) finite = finite && (fabs(Pg[idx]) <= 1.79769313486231570e308);
            }
            int found = 16;
            if (__ballot(!finite) == 0ull) {
                // (Group by group from the bottom: the reference's answer is the FIRST rung that factorises.  Round 3 believed success was not
                // monotone in the jitter for a diverged filter's (n + lambda) P ~ 1e14 and blamed a two-pass search for three lost filters per
                // episode; measured in round 4 (ssa_ladder_probe_f64, tests/test_hip_step.py::test_ladder_on_the_ill_conditioned_tile) it IS
                // monotone on that tile and 4 096 perturbations of it -- the two builds differed in ONE rounding of the diagonal entry,
                // scaled_entry above.)
#ifdef SSA_LADDER_TWO_PASS   // diagnostic ONLY (build_ablate/ladder_probe_tile.py, profiles/r04_ladder_case.txt): the search round 3 tried and
                // reverted -- last rung of each group of four, then the group.  It assumes success is monotone in the jitter (it is, on every case seen).
                const bool ok1 = chol_row_regs<false>(Pg, scale, JITTER[4 * g + 3], g, l, uc);
                const unsigned long long won1 = __ballot(ok1);
                if (won1 != 0ull) {
                    const int grp = (won1 & 0xFFFFull) ? 0 : ((won1 >> 16) & 0xFFFFull) ? 1 : ((won1 >> 32) & 0xFFFFull) ? 2 : 3;
                    const bool okr = chol_row_regs<false>(Pg, scale, JITTER[grp * 4 + g], g, l, uc);
                    const unsigned long long won = __ballot(okr);
                    const int win = (won & 0xFFFFull) ? 0 : ((won >> 16) & 0xFFFFull) ? 1 : ((won >> 32) & 0xFFFFull) ? 2 : 3;
                    found = grp * 4 + win;
                    if (g == win) chol_store_rows(&t.UA[gf * 36], uc, l);
                }
                for (int pass = 4; pass < 4; ++pass) {
#else
                for (int pass = 0; pass < 4; ++pass) {
#endif
                    const bool okr = chol_row_regs<false>(Pg, scale, JITTER[pass * 4 + g], g, l, uc);   // row g tries rung 4 pass + g
                    const unsigned long long won = __ballot(okr);
                    if (won != 0ull) {
                        const int win = (won & 0xFFFFull) ? 0 : ((won >> 16) & 0xFFFFull) ? 1 : ((won >> 32) & 0xFFFFull) ? 2 : 3;
                        found = pass * 4 + win;
                        if (g == win) chol_store_rows(&t.UA[gf * 36], uc, l);
                        break;
                    }
                }
            }
            if (g == gf) rung = found;
            wave_lds_sync();
        }
    }
    return rung;
}

// the action of env e: the word in memory, or the value in the parameter block (SSA_LAUNCH_INLINE_ACTION, one env)
// (word e of an array in the parameter block, by a chain of selects over constant indices: a register-indexed read would make
// the kernels that keep a modified copy of the block -- rollout, closed loop -- spill the whole array to scratch)
SSA_DEV int inline_word(const int32_t (&w)[SSA_INLINE_ENVS], int e)
{
    int v = w[0];
#pragma unroll
    for (int i = 1; i < SSA_INLINE_ENVS; ++i) v = (e == i) ? w[i] : v;
    return v;
}
// INL = false: the instance never sees SSA_LAUNCH_INLINE_ENVS (rollout and closed-loop kernels, whose per-step copy of the block
// stays in scalar registers only while nothing indexes into it)
template <bool INL = true>
SSA_DEV int env_action(const ssa_step_params& p, int e)
{
    if (INL && (p.launch_mask & SSA_LAUNCH_INLINE_ENVS)) return inline_word(p.inline_action, e);
    return (p.launch_mask & SSA_LAUNCH_INLINE_ACTION) ? p.action0 : p.actions[e];
}
// ... and its time index (before time_offset): the word in memory, or the parameter block's (SSA_LAUNCH_INLINE_ENVS)
template <bool INL = true>
SSA_DEV int env_time_of(const ssa_step_params& p, int e)
{
    return (INL && (p.launch_mask & SSA_LAUNCH_INLINE_ENVS)) ? inline_word(p.inline_time, e) : p.env_time[e];
}
// tix % n_time, the division (a ~25-instruction sequence on the vector unit) only when the index has actually wrapped
SSA_DEV int time_row(int tix, int n_time)
{
    if (n_time <= 0) return 0;
    return ((unsigned)tix < (unsigned)n_time) ? tix : tix % n_time;
}

// O4 for one object (ssa_tasker_simple_2.py:834-840): [hx(x_filter[:3]), trace(P)], NaN/inf -> 0.001
SSA_DEV void aer_obs_row(const double* x, const double* P, const ssa_step_params& p, const ssa_consts& C, int e, int64_t obj)
{
    if (p.aer_cols == 1) {   // trace P only
        const double tr1 = P[0] + P[7] + P[14] + P[21] + P[28] + P[35];
        p.aer_out[obj] = (fabs(tr1) <= 1.79769313486231570e308) ? tr1 : 0.001;
        return;
    }
    const int tix = env_time_of(p, e) + p.time_offset;
    const double* M = p.trans + (int64_t)((p.n_time > 0) ? tix % p.n_time : 0) * 9;
    double Mm[9], xx[3] = {x[0], x[1], x[2]}, z[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) Mm[i] = M[i];
    hx_aer(xx, Mm, C.enu, C.obs_itrs, z);
    const double tr = P[0] + P[7] + P[14] + P[21] + P[28] + P[35];
    const double v[4] = {z[0], z[1], z[2], tr};
#pragma unroll
    for (int c = 0; c < 4; ++c) p.aer_out[obj * 4 + c] = (fabs(v[c]) <= 1.79769313486231570e308) ? v[c] : 0.001;
}

// The same block in the step kernel's epilogue, from the tiles, on lanes 0..3 of the object's row: lane c produces component
// c.  Azimuth (lane 0) and elevation (lane 1) are both ONE atan2 -- az = atan2(n, e), el = atan2(u, hypot(e, n)) -- so the
// two lanes walk the same instruction stream once (hx_aer evaluates them one after the other), lane 2 keeps the range,
// lane 3 trace(P); one 8-byte store per lane, 32 contiguous bytes per object.
SSA_DEV void aer_obs_tile_at(const Tiles& t, const ssa_step_params& p, const ssa_consts& C, int g, int l, int64_t obj, int tix)
{
    const double* M = p.trans + (int64_t)time_row(tix, p.n_time) * 9;
    const double* x = &t.X[g * 6];
    const double* P = &t.P[g * 36];
    double d[3], R[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = M[i * 3] * x[0] + M[i * 3 + 1] * x[1] + M[i * 3 + 2] * x[2] - C.obs_itrs[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) R[j] = C.enu[j] * d[0] + C.enu[3 + j] * d[1] + C.enu[6 + j] * d[2];
    const double h2 = R[0] * R[0] + R[1] * R[1];
    const double rt = sqrt_fast((l == 1) ? h2 : d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);   // lane 1: hypot(e, n); others: the range
    double a = atan2_fast((l == 1) ? R[2] : R[1], (l == 1) ? rt : R[0]);
    if (l == 0 && a < 0.0) a += TWO_PI;
    const double tr = P[0] + P[7] + P[14] + P[21] + P[28] + P[35];
    const double v = (l < 2) ? a : (l == 2) ? rt : tr;
    const double w = (fabs(v) <= 1.79769313486231570e308) ? v : 0.001;
    if (p.launch_mask & SSA_LAUNCH_MIRROR_F32) reinterpret_cast<float*>(p.aer_out)[obj * 4 + l] = (float)w;
    else p.aer_out[obj * 4 + l] = w;
}
template <bool INL>
SSA_DEV void aer_obs_tile(const Tiles& t, const ssa_step_params& p, const ssa_consts& C, int g, int l, int e, int64_t obj)
{
    if (p.aer_cols == 1) {   // trace P only: lane 0 of the row
        if (l == 0) {
            const double* P = &t.P[g * 36];
            const double tr = P[0] + P[7] + P[14] + P[21] + P[28] + P[35];
            const double w = (fabs(tr) <= 1.79769313486231570e308) ? tr : 0.001;
            if (p.launch_mask & SSA_LAUNCH_MIRROR_F32) reinterpret_cast<float*>(p.aer_out)[obj] = (float)w;
            else p.aer_out[obj] = w;
        }
        return;
    }
    // one env: the time index is wave-uniform, so the GCRS->ITRS matrix arrives by scalar loads (nine per-lane loads otherwise)
    if (p.n_env > 1) aer_obs_tile_at(t, p, C, g, l, obj, env_time_of<INL>(p, e) + p.time_offset);
    else aer_obs_tile_at(t, p, C, g, l, obj, env_time_of<INL>(p, 0) + p.time_offset);
}

#ifdef SSA_CL_TRACE   // diagnostic build only (build_ablate/closed_loop_timeline.py): closed_loop_kernel, steps SSA_CL_TRACE and + 1
__device__ unsigned long long g_cl_trace[8192 * 16];
#endif
#ifdef SSA_TRACE   // diagnostic build only (build_ablate/wave_timeline.py): per-wave phase timestamps, 100 MHz wall clock
__device__ unsigned long long g_trace[16384 * 16];
// (the FIRST ACTIVE lane stamps: markers 10-14 sit inside the update's row-divergent code, where lane 0 is active only when the
// selected object is the tile's first)
#define SSA_TR(k) do { if (lane == __builtin_amdgcn_readfirstlane(lane) && tile < 16384) g_trace[tile * 16 + (k)] = wall_clock64(); } while (0)
#elif defined(SSA_TRUNC)   // diagnostic build only (build_ablate/trunc_counters.sh): the wave ends at marker SSA_TRUNC, so that the
#define SSA_TR(k) do { if ((k) == SSA_TRUNC) return; } while (0)   // PMC instruction counts of successive builds difference into stages
#else
#define SSA_TR(k) do { } while (0)
#endif

// SSA_LAUNCH_FOLD_INSIDE: the statistics of the step are folded by the LAST wavefront of the launch to finish its atomics instead
// of by a fold kernel behind it (a caller that needs them on the host right after the step -- the gym env -- saves a dependent
// launch, ~4 us on the GPU and 2 us of enqueueing).  Word 3 of every shard counts the tiles that have added to it, word 4 of shard 0
// the shards that are complete; a wavefront bumps them only after its own atomics are acknowledged, so whoever completes the last
// shard reads finished sums (agent-scope loads), writes `stats` and clears shards and counters for the next step.
// (one lane, after its atomics into shard `tile & 127` of env e are acknowledged) counts the tile; true = it completed env e.
// Tiles that add to env e: t_lo = first object / 4 ... t_hi = last object / 4 (a tile that straddles two envs counts in both).
SSA_DEV bool stat_tile_counted(const ssa_step_params& p, int64_t e, int tile)
{
    const int64_t first = e * p.n_obj;
    const int t_lo = (int)(first / OBJ_PER_WAVE), t_hi = (int)((first + p.n_obj - 1) / OBJ_PER_WAVE);
    const int shard = tile & (SSA_STAT_SHARDS - 1);
    // tiles of [t_lo, t_hi] congruent to `shard` modulo 128 (arithmetic shifts: floor division of the negative operands too)
    const unsigned expect = (unsigned)(((t_hi - shard) >> 7) - ((t_lo - 1 - shard) >> 7));
    unsigned long long* env0 = (unsigned long long*)p.stat_shards + e * SSA_STAT_SHARDS * SSA_STAT_SHARD_WORDS;
    unsigned long long* sh = env0 + (int64_t)shard * SSA_STAT_SHARD_WORDS;
    const unsigned old = (unsigned)__hip_atomic_fetch_add(sh + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old != expect - 1u) return false;
    __hip_atomic_store(sh + 3, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int nt = t_hi - t_lo + 1;
    const unsigned used = (unsigned)(nt < SSA_STAT_SHARDS ? nt : SSA_STAT_SHARDS);
    const unsigned old2 = (unsigned)__hip_atomic_fetch_add(env0 + 4, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old2 != used - 1u) return false;
    __hip_atomic_store(env0 + 4, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
// ---- arg-max of sigma_pos on the one-launch paths (the 'shaped' reward needs np.argmax(sigma_pos[i - 1]),
// ssa_tasker_simple_2.py:346).  Every wavefront leaves (key, index) of its tile's first maximum in the tile's slot of `spos_tiles`
// -- no atomics -- and whoever folds the statistics shards reduces the slots with np.argmax's semantics: the first maximum, and a
// NaN ranks above everything (the first NaN wins).  key = the double's bits (sigma_pos >= 0: non-negative doubles order like
// unsigned integers, NaN -- canonicalised -- above infinity); index = the object's index in its env.
constexpr unsigned long long SPOS_NAN_KEY = 0x7ff8000000000000ull;
SSA_DEV unsigned long long spos_key(double v)
{
    return (v != v) ? SPOS_NAN_KEY : ((unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull);
}
SSA_DEV double spos_of_key(unsigned long long k) { return __longlong_as_double((long long)k); }
// first maximum of the tile's `cnt` objects (lane-local: any lane may call it; reads t.Met)
// (ids: the rows' indices as the CALLER numbers them -- ssa_step_params.obj_ids, a layout that stores the objects in another order: the
// first maximum is then the one with the lowest such index, whatever row it sits in)
SSA_DEV void spos_tile_best(const Tiles& t, int cnt, int64_t j0, unsigned long long& key, unsigned long long& idx, bool ids = false)
{
    key = spos_key(t.Met[2]);
    idx = ids ? (unsigned long long)t.Oid[0] : (unsigned long long)j0;
#pragma unroll
    for (int g = 1; g < OBJ_PER_WAVE; ++g) {
        const unsigned long long k = spos_key(t.Met[g * 4 + 2]);
        const unsigned long long ig = ids ? (unsigned long long)t.Oid[g] : (unsigned long long)(j0 + g);
        if (g < cnt && (k > key || (ids && k == key && ig < idx))) { key = k; idx = ig; }
    }
}
// 64-bit wave folds by DPP rotations + readlanes (see the closed loop, which introduced them)
template <int CTRL>
SSA_DEV unsigned long long dpp_u64(unsigned long long v)
{
    return (unsigned long long)__builtin_amdgcn_update_dpp((long long)v, (long long)v, CTRL, 0xF, 0xF, true);
}
SSA_DEV unsigned long long lane_u64(unsigned long long v, int l)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, l);
}
struct OpMax { SSA_DEV unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a > b ? a : b; } };
struct OpMin { SSA_DEV unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a < b ? a : b; } };
struct OpAdd { SSA_DEV unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a + b; } };
template <class OP>
SSA_DEV unsigned long long wave_fold_u64(unsigned long long v, OP op)
{
    v = op(v, dpp_u64<0x128>(v));   // row_ror:8
    v = op(v, dpp_u64<0x124>(v));   // row_ror:4
    v = op(v, dpp_u64<0x122>(v));   // row_ror:2
    v = op(v, dpp_u64<0x121>(v));   // row_ror:1
    return op(op(lane_u64(v, 0), lane_u64(v, 16)), op(lane_u64(v, 32), lane_u64(v, 48)));
}
// reduces the slots of env e's tiles (tiles [t_lo, t_hi]: whole tiles of ONE env -- the launcher refuses spos_tiles for envs whose
// object count is not a multiple of four unless there is only one env) into stats.  COHERENT: agent-scope loads (the slots were
// written by other wavefronts of the SAME launch: SSA_LAUNCH_FOLD_INSIDE); otherwise a kernel boundary lies in between.
template <bool COHERENT>
SSA_DEV void fold_spos_tiles(const unsigned long long* __restrict__ slots, int t_lo, int t_hi, double* __restrict__ stats, int lane)
{
    unsigned long long key = 0ull, idx = ~0ull;
    bool any = false;
    for (int tb = t_lo + lane; tb <= t_hi; tb += 64) {
        unsigned long long k, i;
        if (COHERENT) {
            k = __hip_atomic_load(slots + 2 * (int64_t)tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            i = __hip_atomic_load(slots + 2 * (int64_t)tb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(slots + 2 * (int64_t)tb);
            k = v.x; i = v.y;
        }
        if (!any || k > key || (k == key && i < idx)) { key = k; idx = i; any = true; }      // (ties: the lower index -- the earlier tile, unless a layout renumbers)
    }
    const unsigned long long top = wave_fold_u64(any ? key : 0ull, OpMax());
    const unsigned long long who = wave_fold_u64((any && key == top) ? idx : ~0ull, OpMin());
    if (lane == 0) {
        stats[SSA_STAT_ARGMAX_SPOS] = (double)(long long)who;
        stats[SSA_STAT_MAX_SPOS] = spos_of_key(top);
    }
}
// tiles of env e (whole tiles; one env: all of them)
SSA_DEV void env_tile_range(int64_t n_obj, int e, int& t_lo, int& t_hi)
{
    const int64_t first = (int64_t)e * n_obj;
    t_lo = (int)(first / OBJ_PER_WAVE);
    t_hi = (int)((first + n_obj - 1) / OBJ_PER_WAVE);
}
SSA_DEV void fold_stat_shards_inside(unsigned long long* __restrict__ shards, double* __restrict__ stats, int lane)
{
    unsigned long long* sh = shards + (int64_t)lane * SSA_STAT_SHARD_WORDS;
    unsigned long long* sh2 = sh + 64 * SSA_STAT_SHARD_WORDS;
    unsigned long long mx = __hip_atomic_load(sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long cn = __hip_atomic_load(sh + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long nf = __hip_atomic_load(sh + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long mb = __hip_atomic_load(sh2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long cb = __hip_atomic_load(sh2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long nb = __hip_atomic_load(sh2 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        __hip_atomic_store(sh + q, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sh2 + q, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    mx = mb > mx ? mb : mx;
    cn += cb;
    nf += nb;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long m2 = __shfl_down(mx, off, 64), c2 = __shfl_down(cn, off, 64), n2 = __shfl_down(nf, off, 64);
        mx = m2 > mx ? m2 : mx;
        cn += c2;
        nf += n2;
    }
    if (lane == 0) {
        stats[SSA_STAT_MAX_DPOS] = __longlong_as_double((long long)mx);
        stats[SSA_STAT_CNT_LT_1E4] = (double)(cn & 0xffffffffull);
        stats[SSA_STAT_CNT_LT_1E7] = (double)(cn >> 32);
        stats[SSA_STAT_ARGMAX_SPOS] = -1.0;
        stats[SSA_STAT_N_FAILED] = (double)nf;
        stats[SSA_STAT_MAX_SPOS] = __builtin_nan("");
        stats[6] = 0.0; stats[7] = 0.0;
    }
}

// One wavefront advances up to 4 consecutive objects (one per 16-lane row) by one env step, complete semantics:
// the robust_cholesky ladder is inline, conic branches beyond the strong-elliptic one are out-of-line calls taken
// only by the lanes that need them.
// TILE: 0 = the tile is loaded here; 1 = it was prefetched (commit now, request the next one after the
// Kepler stage); 2 = the tiles already hold the state (a rollout's later steps)
// Where a wavefront learns its env's action.  ActEarly: the action word is in memory when the step starts (per-step launches,
// rollouts).  ActLate (closed_loop_kernel, one env): the action of step k is decided ON THE DEVICE from the state step k - 1
// left, while the predicts of step k are already running -- a wavefront asks for it only where the update needs it, behind
// its predict, and every row prefetches the update's inputs for its OWN object in case it turns out to be the selected one.
struct ActEarly {
    static constexpr bool late = false;
    SSA_DEV int get() { return -1; }
    SSA_DEV void before_wait(Tiles&, int, int) {}
    SSA_DEV void mid_step(Tiles&, int) {}
};
struct ActLate;
struct LoopK;
SSA_DEV void closed_loop_prescore(ActLate& a, Tiles& t, int lane, int cnt);   // (defined with the closed-loop kernel)
SSA_DEV void closed_loop_store(const LoopK* lk, Tiles& t, int lane, int kk, int64_t base, int cnt);
constexpr unsigned CL_ABORT_GEN = 0xFFFFFFFFu;            // decision number that means "give up" (a wavefront timed out)
constexpr unsigned long long CL_TIMEOUT_TICKS = 200000000ull;   // default bound of a wait: 2 s of the 100 MHz wall clock (ssa_closed_loop_params.wait_ticks)
struct ActLate {
    static constexpr bool late = true;
    unsigned long long* flag;        // this wavefront's group flag: (decision number << 32) | action
    unsigned long long* all_flags;   // [nflags] flags, 16 words apart (abort broadcast)
    int* err;                        // device word set to 1 on a timeout (may be host-mapped)
    int nflags;
    unsigned long long timeout;      // bound of every wait, 100 MHz ticks
    unsigned want;                   // decision needed: the step's index (>= 1); 0 = `first`
    int first;                       // action of the launch's first step (decided by the caller)
    int last;                        // the action get() returned most recently
    bool aborted;
#ifdef SSA_CL_TRACE
    unsigned long long t_wait, t_seen;
#endif
    int agent;                       // SSA_AGENT_*
    const ssa_consts* C;             // (kernarg copy)
    const double* M;                 // GCRS -> ITRS matrix of the current step
    int* vis;                        // [OBJ_PER_WAVE] in LDS: visibility of the tile's objects at the current step
    // work that does not depend on the decision, done where a wavefront would otherwise only wait for it: the visibility of its
    // objects' NEW true states (the agents' mask, agents.py:36-42) -- off the path decision -> update -> score -> decision
    SSA_DEV void before_wait(Tiles& t, int lane, int cnt) { closed_loop_prescore(*this, t, lane, cnt); }
    // The outputs of step k leave for HBM inside step k + 1, behind its Cholesky stage (the tile is intact until then): right
    // after the decision every wavefront is released at once, and 18 MB of tile stores issued at that moment delayed the
    // acknowledgements of the parts the NEXT decision waits for (late announcers: +3 us, build_ablate/closed_loop_timeline.py);
    // here they overlap the Kepler stage instead.
    const LoopK* lk;                 // the kernel's argument block
    int pend;                        // step whose tile is still to be stored (-1: none)
    int64_t base;
    int cnt;
    SSA_DEV void mid_step(Tiles& t, int lane)
    {
        if (pend < 0) return;
        closed_loop_store(lk, t, lane, pend, base, cnt);
        pend = -1;
    }
    SSA_DEV void abort_all()
    {
        const int lane = threadIdx.x;
        for (int i = lane; i < nflags; i += 64)
            __hip_atomic_store(all_flags + (int64_t)i * 16, (unsigned long long)CL_ABORT_GEN << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0 && err) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // the decision `want`: polls the group flag (agent-scope loads: coherent across the XCDs' L2s) with a bounded wait -- every
    // wavefront reaches an exit whatever the others do
    SSA_DEV int get()
    {
        if (want == 0u) { last = first; return first; }
#ifdef SSA_CL_NOWAIT   // diagnostic (build_ablate/closed_loop_variants.py): no wait for the decision -- what the exchange's latency costs
        return first;
#endif
        const unsigned long long t0 = wall_clock64();
#ifdef SSA_CL_TRACE
        t_wait = t0;
#endif
        unsigned lo, hi;
        for (;;) {
            const unsigned long long v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo = __builtin_amdgcn_readfirstlane((unsigned)v);
            hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            if (hi >= want) break;
            if (wall_clock64() - t0 > timeout) {
                abort_all();
                hi = CL_ABORT_GEN;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (hi == CL_ABORT_GEN) {
            aborted = true;
            return -2;
        }
#ifdef SSA_CL_TRACE
        t_seen = wall_clock64();
#endif
        last = (int)lo;
        return (int)lo;
    }
};

template <int PROP, int TILE, class ACT>
SSA_DEV void process_wave(Tiles& t, const ssa_consts& C, const ssa_step_params& p, int lane, int64_t obj_in, bool valid,
                          int64_t base, int cnt, TileRegs& pf, int64_t next_base, int next_cnt, int tile, ACT& asrc)
{
    constexpr bool INL = (TILE != 2) && !ACT::late;   // (the per-step launches: SSA_LAUNCH_INLINE_ENVS may be set)
    // SSA_LAUNCH_FOLD_INSIDE exists in the one-tile instance only: counting a tile means waiting for its atomics' acknowledgement
    // and for a returning atomic -- once, at the end of a one-tile wavefront's life, but ~1.5 us per tile in the grid-stride
    // instance (8 x 20 000 objects: 100.7 us per step instead of 89.0); the launcher sends those launches a fold kernel instead.
    // (Round 4 tried counting once per WAVEFRONT, at the end of its life -- lane i counts the walk's i-th tile, one round trip:
    // build_ablate/grid_stride_fold_inside_experiment.patch.  Correct, and slower than the 4.3 us fold kernel it replaces: the wavefronts of
    // a grid-stride launch end together, and the 128 shard-complete increments per env land on ONE word one after the other -- the vector
    // env's step 123.9 -> 130.4 us although its host side got 4 us shorter.)
    constexpr bool FOLD_OK = (TILE == 0);
    int g = lane >> 4, l = lane & 15;
    int64_t obj = obj_in;
    // (TILE 0: the kernel issued the tile's loads from its preloaded pointer arguments before anything else.)  First thing here,
    // so that these scalar loads complete with the action / time words' below -- placed behind the wait for the tile they
    // put one more scalar-memory round trip between the tile's arrival and its commit to LDS
#ifndef SSA_NO_EARLY_ARGS
    if (TILE == 0) {
        // the epilogue's output pointers are fetched NOW: their scalar loads (kernarg segment) overlap the tile's HBM round
        // trip instead of each adding a scalar-memory round trip to the store path of a latency-bound wavefront
        asm volatile("" ::"s"(p.P_out), "s"(p.x_out), "s"(p.x_true_out), "s"(p.obs), "s"(p.metrics), "s"(p.stat_shards), "s"(p.upd),
                     "s"(p.aer_out), "s"(p.n_obj));
    }
#endif

    // env of the object: no division for the single-env case, a 32-bit one otherwise (n_env * n_obj < 2^31)
    int e = (valid && p.n_env > 1) ? (int)((uint32_t)obj / (uint32_t)p.n_obj) : 0;
    const int64_t j = valid ? obj - (int64_t)e * p.n_obj : 0;
    // the action / time index of this object's env, fetched early (used after the transform)
    // (one env: wave-uniform scalar loads; per-lane loads with their 64-bit address arithmetic only for vectorised envs)
    int act, tix;
    if (ACT::late) {   // (one env; the action arrives behind the predict)
        act = -1;
        tix = valid ? p.env_time[0] + p.time_offset : 0;
    } else if (p.n_env > 1) {
        act = valid ? env_action<INL>(p, e) : -1;
        tix = valid ? env_time_of<INL>(p, e) + p.time_offset : 0;
    } else {
        const int a0 = env_action<INL>(p, 0), t0 = env_time_of<INL>(p, 0);
        act = valid ? a0 : -1;
        tix = valid ? t0 + p.time_offset : 0;
    }
    // ---- the one update of this env (ssa_tasker_simple_2.py:292-315) runs in the row that owns the selected object.  Its
    // wavefront is the longest-living one of the launch, so its inputs (this step's GCRS->ITRS matrix, the measurement noise)
    // leave HBM now and wait in LDS, instead of costing two memory round trips when the update starts
    const bool interval_ok = (C.update_interval <= 1) || (tix % C.update_interval == 0);
    // ssa_step_params.obj_ids (the per-step kernels; per env, indices within the env): the objects are stored in another order than the caller numbers them; the
    // action, the failure records, the arg-max of sigma_pos and the host-facing observation rows speak the CALLER's indices
    int64_t jid = j;
    if (p.obj_ids) {
        // (the tile's four indices by ONE wave-uniform 16-byte load -- scalar memory: it does not queue behind the tile's vector loads, which a
        // per-lane load would, and the update's input prefetch below hangs on `my_update`; the table is padded to whole tiles)
        const int4 ids = *reinterpret_cast<const int4*>(p.obj_ids + base);
        const int mine = (g == 0) ? ids.x : (g == 1) ? ids.y : (g == 2) ? ids.z : ids.w;
        jid = valid ? (int64_t)mine : 0;
        if (l == 0) t.Oid[g] = mine;
    }
    bool my_update = !ACT::late && valid && act >= 0 && (int64_t)act == jid && interval_ok;
    // (ActLate: every row prefetches for its own object)
    const bool may_update = ACT::late ? (valid && interval_ok) : my_update;
    // ... and it issues ahead of its SIMD's other wavefronts from here on: at equal priority its predict runs at a fifth of the
    // SIMD and the update then starts when everybody else is finishing (the kernel's tail)
    if (!ACT::late && __any(my_update)) __builtin_amdgcn_s_setprio(3);
    int tmod = 0;                                           // row of `trans` / `z_noise` (episodes wrap)
    double upd_in = 0.0;
    if (may_update && l < 12) {
        tmod = time_row(tix, p.n_time);
        const int64_t aobj = ACT::late ? jid : (int64_t)act;   // (the measurement noise is indexed as the caller numbers the objects)
        const double* src = (l < 9) ? p.trans + (int64_t)tmod * 9 + l
                                    : p.z_noise + (int64_t)e * p.zn_stride_env + (int64_t)tmod * p.zn_stride_time + aobj * p.zn_stride_obj + (l - 9);
        upd_in = *src;
    }
    SSA_TR(0);
#ifdef SSA_TRACE
    if (lane == 0 && tile < 16384) {
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_trace[tile * 16 + 15] = ((unsigned long long)xcc << 32) | hw;
        g_kep_dbg[(blockIdx.x & 16383) * 2] = 0u;
        g_kep_dbg[(blockIdx.x & 16383) * 2 + 1] = 0u;
    }
#endif

    if (TILE == 0 && pf.dma) tile_dma_wait(t, pf, lane);   // (the one-tile kernels: the tile came by LDS-DMA)
    else if (TILE != 2) tile_commit(t, pf, lane);        // TILE 1: requested one tile ago (or by the kernel prologue)
    if (lane < 8) t.Z[lane] = 0.0;
    if (TILE != 2 && lane < 36) t.Q[lane] = C.Q[lane];   // (a rollout's later steps find it in place)
    wave_lds_sync();
    SSA_TR(1);

    const int st_in = t.St[g];
    const bool active = valid && st_in == SSA_ST_OK;

    // ---- U1/U2: sigma points
#if defined(SSA_ABLATE) && (SSA_ABLATE & 2)
    const int rung = -1;
    if (l < 6) for (int c = 0; c < 6; ++c) t.UA[g * 36 + l * 6 + c] = 1e-3 * t.P[g * 36 + l * 6 + c];
#else
    const int rung = robust_chol_row_lds(t, C.scale, g, l);
#endif
    asrc.mid_step(t, lane);   // (closed loop: the PREVIOUS step's tile leaves for HBM now, its last reader of t.Obs)
    if (may_update && l < 12) t.Obs[g * 12 + l] = upd_in;
    wave_lds_sync();
    SSA_TR(2);
    const bool chol_fail = (rung == 16);
    const bool is_sigma = (l <= 12);
    const bool is_pm = (l >= 1 && l <= 12);
    double s[6];
    {
        const int krow = (l >= 7) ? l - 7 : l - 1;          // factor row of lanes 1..12
        const double sgn = (l >= 7) ? -1.0 : 1.0;
        const bool use_filter = active && !chol_fail && l != 13;
        // inactive rows (failed / out-of-range objects) and lane 13 propagate the true state; sigma_0 and the idle lanes
        // add the zero row: the operands are chosen by ADDRESS (three 16-byte LDS reads each), not by value
        const double* base = use_filter ? &t.X[g * 6] : &t.T[g * 6];
        const double* urow = (use_filter && is_pm) ? &t.UA[g * 36 + krow * 6] : &t.Z[0];
#pragma unroll
        for (int c = 0; c < 6; ++c) s[c] = fma(sgn, urow[c], base[c]);
    }
    // ---- P1-P5: one Kepler solve per lane; strong-elliptic fast path inline, every other conic
    // branch through the out-of-line complete restatement
    double o[6];
    {
#if defined(SSA_ABLATE) && (SSA_ABLATE & 1)
        for (int c = 0; c < 6; ++c) o[c] = s[c] + 1e-3 * C.dt * s[(c + 3) % 6];
        const bool kep_ok = true;
#else
        bool kep_ok;
        if (PROP == 2) {
            const J2Params jq = {C.j2, C.r_eq, C.rk4_substeps};
            kep_ok = propagate_j2_rk4(s, C.dt, jq, o);
        } else {
            if (PROP == 3) kep_ok = kepler_hybrid_fast(s, C.dt, o);
            else
            kep_ok = kepler_step_fast<PROP == 2 ? 1 : PROP, 1>(s, C.dt, o);
        }
#endif
        if (PROP == 3) {
            if (__any(!kep_ok)) {   // sigma points outside the strong-elliptic regime: the reference's branches (whole-wave branch)
                // (no issue priority for the inline tier: late in an episode three wavefronts in four take it, and boosting the many
                // only starves the few -- they became the launch's tail, 8-11 us in their Cholesky stage; the rare out-of-line call keeps it)
                // second tier, inline: the conic branches for the general orientation (the strong-hyperbolic one -- where a diverged filter
                // lives -- without a call); third tier, out of line: the complete restatement for rv2coe's special branches and NaN input
                bool served = kep_ok;
                if (!ACT::late) {     // (the closed-loop instance keeps the call alone: with the tier inline it spilled until round 4's band change and is
                                      // 6 % slower since -- 56.8 k against 60.2 k env-steps/s; the callee takes the lean form too)
                    if (!kep_ok) served = kepler_conic_lean<1, true>(s, C.dt, o);
                }
                if (__any(!served)) {
                    __builtin_amdgcn_s_setprio(3);
                    if (!served) {
                        Vec6 si;
#pragma unroll
                        for (int c = 0; c < 6; ++c) si.v[c] = s[c];
                        Vec6 oo = kepler_beyond_series_tagged<1>(si, C.dt);
#pragma unroll
                        for (int c = 0; c < 6; ++c) o[c] = oo.v[c];
                    }
                }
            }
        } else if (PROP != 0) {
            // SSA_PROP_FG / J2: the solvers cover every conic; they decline only NaN / degenerate input or a
            // non-converging iteration, which IS a NaN result (farnocchia.py:353) -> 'predict returned nan'
            if (__any(!kep_ok)) {   // (whole-wave branch: twelve selects on the common path otherwise)
                asm volatile("");   // (keeps the optimiser from flattening the branch back into those selects)
                if (!kep_ok) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) o[c] = __builtin_nan("");
                }
            }
        } else if (__any(!kep_ok)) {
            // SSA_PROP_ELEMENTS outside the strong-elliptic regime (or NaN input): the complete restatement of
            // farnocchia(), out of line, for the lanes that need it (whole-wave branch: skipped otherwise)
            // (the reference's formulas branch by branch with this file's fast primitives, as SSA_PROP_HYBRID: the strong-hyperbolic
            // branch inline, the rest through the out-of-line restatement)
            bool served = kep_ok;
            if (!ACT::late) {     // (the closed-loop instance of this variant keeps the call alone: with the tier inline it spilled)
                if (!kep_ok) served = kepler_conic_lean<2, true>(s, C.dt, o);
            }
            if (__any(!served)) {
                if (!served) {
                    Vec6 si;
#pragma unroll
                    for (int c = 0; c < 6; ++c) si.v[c] = s[c];
                    Vec6 oo = kepler_beyond_series_tagged<2>(si, C.dt);
#pragma unroll
                    for (int c = 0; c < 6; ++c) o[c] = oo.v[c];
                }
            }
        }
    }
    wave_lds_sync();   // every lane has consumed t.X / t.T / t.U
    // The propagator is the register-pressure peak of the kernel.  Re-derive the lane coordinates behind it, so that the
    // lane-derived LDS addresses of the stages that follow are formed there instead of being carried across it (the
    // alternative the allocator picks for the RK4 instance is a spill).
    asm volatile("" : "+v"(lane));
    g = lane >> 4;
    l = lane & 15;
    // ... and so are the object index and its env (the multi-tile instance spilled these and the action word across the
    // propagator: 24 bytes per lane and tile = 61 MB of scratch writes per 160 000-object step); the action word is read
    // again where the rare paths below need it (a scalar load here would sit in front of every LDS wait that follows)
    obj = base + g;
    e = (valid && p.n_env > 1) ? (int)((uint32_t)obj / (uint32_t)p.n_obj) : 0;
    SSA_TR(3);
    // the next tile's inputs: in flight during the transform / covariance / observation / store of this one
    if (TILE == 1) tile_issue(pf, p, lane, next_base, next_cnt);

    // ---- U3: unscented transform, centred form of x = dot(Wm, sigmas_f):
    //   x = sigma_0' + m',   m' = (sum(Wm) - 1) sigma_0' + Wi sum_{i>=1} (sigma_i' - sigma_0')
    // (the reference's sum evaluated without the 1e8-fold cancellation of Wm0 ~ -2e8)
    {
        typedef double v2d_t __attribute__((ext_vector_type(2)));
        double d[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) d[c] = o[c] - row_bcast<0>(o[c]);
        if (is_pm) {          // row [d_i, 1, 0] of the operand matrix
            v2d_t* dst = reinterpret_cast<v2d_t*>(&t.D[dbase(g) + (l - 1) * 8]);
            dst[0] = v2d_t{d[0], d[1]};
            dst[1] = v2d_t{d[2], d[3]};
            dst[2] = v2d_t{d[4], d[5]};
            dst[3] = v2d_t{1.0, 0.0};
        } else if (l == 0 || l == 13) {   // sigma_0' (until the mean replaces it) | x_true[i]
            v2d_t* dst = reinterpret_cast<v2d_t*>(l == 0 ? &t.X[g * 6] : &t.T[g * 6]);
            dst[0] = v2d_t{o[0], o[1]};
            dst[1] = v2d_t{o[2], o[3]};
            dst[2] = v2d_t{o[4], o[5]};
        }
    }
    wave_lds_sync();
    Moments mo = {0.0, 0.0, 0.0};
#if !(defined(SSA_ABLATE) && (SSA_ABLATE & 8))
#ifndef SSA_COV_FOUR_CHUNKS   // (diagnostic: round 4's first form of the reference covariance -- nine + twelve matrix instructions)
    if (C.flags & SSA_FLAG_REFERENCE_COV) mo = moment_sums_mfma<true>(t, lane);   // (wave-uniform branch: the sums alone, six matrix instructions)
    else
#endif
    mo = moment_sums_mfma(t, lane);
#endif
    wave_lds_sync();
    SSA_TR(4);
    double xb_l = 0.0;   // lanes 0..5: component l of the prior mean
    if (l < 6) {
        const double s0 = t.X[g * 6 + l];
        const double ssum = C.Wi * t.M[g * 12 + l];
        const double mp = C.sum_wm_m1 * s0 + ssum;
        xb_l = s0 + mp;
        t.M[g * 12 + l] = ssum;
        t.M[g * 12 + 6 + l] = mp;
        t.X[g * 6 + l] = xb_l;
    }
    const bool nan_x = ((__ballot(l < 6 && xb_l != xb_l) >> (g * 16)) & 0xFFFFull) != 0;
    wave_lds_sync();
#if !(defined(SSA_ABLATE) && (SSA_ABLATE & 4))
    if (C.flags & SSA_FLAG_REFERENCE_COV) covariance_reference(t, C, lane, o, xb_l);   // (wave-uniform branch)
    else covariance_finish(t, C, lane, mo);
#endif
    wave_lds_sync();
    SSA_TR(5);

    int st_new = st_in;
    if (active) {
        if (chol_fail) st_new = SSA_ST_PREDICT_LINALG;
        else if (nan_x) st_new = SSA_ST_PREDICT_NAN;
    }
    // SSA_FLAG_RESAMPLE (the predict() of filterpy's development branch): sigma_points(x_prior, P_prior) is drawn at the
    // end of predict() for EVERY filter, so an exhausted robust_cholesky ladder fails the filter in THIS step's predict
    // (not one step later); the update then uses those points (t.U holds the factor rows)
    if (C.flags & SSA_FLAG_RESAMPLE) {
        const int rg = robust_chol_row_lds(t, C.scale, g, l);
        wave_lds_sync();
        if (active && st_new == SSA_ST_OK && rg == 16) st_new = SSA_ST_PREDICT_LINALG;
    }

    // ---- U5: the one update of this env (ssa_tasker_simple_2.py:292-315) for the selected object.
    // Phase 1 (row-local, lanes of the selected object's row): measurement of the sigma points, predicted measurement,
    // residuals; the rows [sigma - x | rz] go to the row's staging matrix.  Phase 2 (whole wavefront, below): the two
    // weighted moment matrices on the matrix unit (the weights enter with the right operand), inverse, gain, state and
    // covariance over all 64 lanes.
    if (ACT::late) {   // the closed loop's decision for this step: needed from here on, and normally made long ago
        asrc.before_wait(t, lane, cnt);
        act = asrc.get();
        my_update = valid && act >= 0 && (int64_t)act == (p.obj_ids ? (int64_t)t.Oid[g] : obj) && interval_ok;   // (one env: the object index IS the index in the env)
        if (__any(my_update)) __builtin_amdgcn_s_setprio(3);
    }
    if (__any(my_update)) {   // whole-wave branch: a wavefront without a selected object skips the block, its variables included
    bool upd_go = false, taken = false, visible = false, attempted = false;
    double z[3] = {0.0, 0.0, 0.0}, y_row[3] = {0.0, 0.0, 0.0};   // (y_row: lane 13 of the row keeps the innovation)
    double* rec = nullptr;
    // staging matrix of the row's update: [13][9], row i = [sigma_i - x | rz_i].  The four rows' matrices (468 doubles) lie in the
    // transform's scratch, free by now: the factor tile and the head of t.D (contiguous members of Tiles)
    double* const STG = &t.UA[g * 117];
    double* const W = &t.D[330];        // small matrices: W[0..9) S | W[9..27) Pxz | W[36..54) K | W[54..57) y
    static_assert(offsetof(Tiles, D) == offsetof(Tiles, UA) + sizeof(double) * OBJ_PER_WAVE * 36, "UA and D contiguous");
    static_assert(4 * 117 <= OBJ_PER_WAVE * 36 + 330 && 330 + 57 <= 408, "update staging fits");
    if (my_update) {
        rec = p.upd ? p.upd + (int64_t)e * SSA_UPD_STRIDE : nullptr;
        // a filter that has failed (earlier, or in this step's predict) is skipped entirely (:293): no z_true, no record
        attempted = (st_new == SSA_ST_OK);
        if (attempted) {
            const double* M = &t.Obs[g * 12];
            // sigma points handed to update(): the propagated ones or, with
            // SSA_FLAG_RESAMPLE, the set drawn from the prior at the end of predict (factor rows in t.U)
            double sf[6], xb[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                sf[c] = o[c];
                xb[c] = t.X[g * 6 + c];   // the prior mean
            }
            if (C.flags & SSA_FLAG_RESAMPLE) {
                const int krow = is_pm ? (l - 1) % 6 : 0;
                const double sgn = (l >= 1 && l <= 6) ? 1.0 : ((l >= 7 && l <= 12) ? -1.0 : 0.0);
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    if (l != 13) sf[c] = xb[c] + sgn * t.UA[g * 36 + krow * 6 + c];
            }
            // (the left half of the staging row leaves now -- harmless if the object turns out not to be visible -- so that the
            // prior mean does not have to live across the measurement function)
            if (l <= 12) {
#pragma unroll
                for (int c = 0; c < 6; ++c) STG[l * 9 + c] = sf[c] - xb[c];
            }
            // H1/H2: measurement of every sigma point (lanes 0-12) and of the true state (lane 13)
            double enu_vec[3];
            double el_mine;
            {
                double Mm[9], aer[3];
#pragma unroll
                for (int i = 0; i < 9; ++i) Mm[i] = M[i];
                hx_aer_enu(sf, Mm, C.enu, C.obs_itrs, aer, enu_vec);
                el_mine = aer[1];
                if (C.obs_type == SSA_OBS_AER) { z[0] = aer[0]; z[1] = aer[1]; z[2] = aer[2]; }
                else { z[0] = sf[0]; z[1] = sf[1]; z[2] = sf[2]; }
            }
            SSA_TR(10);
            visible = row_bcast<13>(el_mine) >= C.obs_limit;  // object_visible(): elevation of the TRUE state (:418-425)
            if (rec && l == 13) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rec[SSA_UPD_Z_TRUE + c] = z[c];
            }
            if (visible) {
                // H3/H5: predicted measurement
                double zp[3];
                const double wl = (l == 0) ? C.Wc0 : (is_pm ? C.Wi : 0.0);
                if (C.obs_type == SSA_OBS_AER) {
                    double um[3];   // mean_z_uvw (dynamics.py:343): aer2uvw of a sigma point is its local vector enu_vec
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        double u0 = row_bcast<0>(enu_vec[c]);
                        double du = is_pm ? (enu_vec[c] - u0) : 0.0;
                        um[c] = u0 + (C.sum_wm_m1 * u0 + C.Wi * row_allsum(du));
                    }
                    uvw2aer(um, zp);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        double u0 = row_bcast<0>(z[c]);
                        double du = is_pm ? (z[c] - u0) : 0.0;
                        zp[c] = u0 + (C.sum_wm_m1 * u0 + C.Wi * row_allsum(du));
                    }
                }
                SSA_TR(11);
                // H4: residuals; lane 13 forms the innovation of the noisy measurement
                double rz[3];
                {
                    double zin[3];
                    if (l == 13) {
                        const double* zn = &t.Obs[g * 12 + 9];
#pragma unroll
                        for (int c = 0; c < 3; ++c) zin[c] = z[c] + zn[c];
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) zin[c] = z[c];
                    }
                    if (C.obs_type == SSA_OBS_AER) residual_z_aer_wrapped(zin, zp, rz);
                    else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) rz[c] = zin[c] - zp[c];
                    }
                }
                upd_go = true;
                if (l <= 12) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) STG[l * 9 + 6 + c] = rz[c];
                }
                if (l == 13) {   // the innovation
#pragma unroll
                    for (int c = 0; c < 3; ++c) y_row[c] = rz[c];
                }
            }
        }
    }
    // ---- Phase 2: whole wavefront, one selected object at a time
    {
        unsigned long long pend = __ballot(upd_go);
        while (pend) {   // (several selected objects in one wavefront -- vectorised envs of fewer than four objects -- take turns)
            const int gu = (__ffsll((long long)pend) - 1) >> 4;
            pend &= ~(0xFFFFull << (gu * 16));
            if (g == gu && l == 13) {
#pragma unroll
                for (int c = 0; c < 3; ++c) W[54 + c] = y_row[c];
            }
            wave_lds_sync();
            // G = [sigma - x | rz]^T (Wc rz)  (9 x 3): rows 0..5 = Pxz, rows 6..8 = S - R.  One v_mfma_f64_4x4x4 per chunk of four
            // sigma points; its blocks (lane bits 3:2) take the row tiles 0-3, 4-7 and 8 of the left operand;
            // k = lane bits 5:4, i / j = lane bits 1:0 (see moment_sums_mfma)
            {
                const int kk = lane >> 4, blk = (lane >> 2) & 3, ij = lane & 3;
                const double* SG = &t.UA[gu * 117];
                const int acol = (blk < 2) ? 4 * blk + ij : 8;     // (block 2: row 8 in every lane, rows 9..11 of G are not used; block 3 idles on it)
                double acc = 0.0;
#pragma unroll
                for (int cch = 0; cch < 4; ++cch) {
                    const int k = (cch < 3) ? 4 * cch + kk : 12;   // sigma point; the last chunk holds point 12 and three empty slots
                    const double* row = &SG[k * 9];
                    double av = row[acol];
                    double bv = row[6 + (ij < 3 ? ij : 2)] * ((cch == 0 && kk == 0) ? C.Wc0 : C.Wi);
                    if (cch == 3 && kk != 0) { av = 0.0; bv = 0.0; }
                    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc, 0, 0, 0);
                }
                // result lane (i = lane bits 5:4, blk, j = lane bits 1:0) holds G[4 blk + i][j]
                const int ri = 4 * blk + kk;
                if (ij < 3 && ri < 6) W[9 + ri * 3 + ij] = acc;
                if (ij < 3 && ri >= 6 && ri < 9 && ri - 6 <= ij) {   // S symmetric by construction: the upper triangle, mirrored
                    const int a = ri - 6;
                    W[a * 3 + ij] = acc + C.R[a * 3 + ij];
                    W[ij * 3 + a] = acc + C.R[ij * 3 + a];
                }
            }
            wave_lds_sync();
            SSA_TR(12);
            bool inv_ok;
            double SI[9];
            {
                double S[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) S[c] = W[c];
                inv_ok = inv3(S, SI);          // (every lane: the inverse stays in registers)
            }
            bool nan_u = false;
            if (inv_ok) {
                // K = Pxz inv(S): 18 entries, one per lane
                if (lane < 18) {
                    const int a = lane / 3, b = lane - 3 * a;
                    W[36 + lane] = W[9 + a * 3] * SI[b] + W[9 + a * 3 + 1] * SI[3 + b] + W[9 + a * 3 + 2] * SI[6 + b];
                }
                wave_lds_sync();
                SSA_TR(13);
                // x += K y  (lanes 0..5); P -= K S K^T (36 entries, one per lane)
                double xn = 0.0;
                if (lane < 6) xn = t.X[gu * 6 + lane] + (W[36 + lane * 3] * W[54] + W[36 + lane * 3 + 1] * W[55] + W[36 + lane * 3 + 2] * W[56]);
                nan_u = __ballot(lane < 6 && xn != xn) != 0;
                if (lane < 36) {
                    const int a = lane / 6, b = lane - 6 * a;
                    double corr = 0.0;
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const double sk = W[u * 3] * W[36 + b * 3] + W[u * 3 + 1] * W[36 + b * 3 + 1] + W[u * 3 + 2] * W[36 + b * 3 + 2];  // (S K^T)[u][b]
                        corr = fma(W[36 + a * 3 + u], sk, corr);
                    }
                    t.P[gu * 36 + lane] = t.P[gu * 36 + lane] - corr;
                }
                if (lane < 6) t.X[gu * 6 + lane] = xn;
                SSA_TR(14);
            }
            if (g == gu) {
                if (!inv_ok) st_new = SSA_ST_UPDATE_LINALG;
                else {
                    taken = true;
                    if (nan_u) st_new = SSA_ST_UPDATE_NAN;
                    if (rec) {
                        if (l < 3) rec[SSA_UPD_Y + l] = W[54 + l];
                        if (l < 9) rec[SSA_UPD_S + l] = W[l];
                        if (is_sigma) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) rec[SSA_UPD_SIGMAS_H + l * 3 + c] = z[c];
                        }
                    }
                }
            }
            wave_lds_sync();   // (the next selected object reuses W)
        }
    }
    if (my_update && rec && l == 0) {
        rec[SSA_UPD_OBS_TAKEN] = taken ? 1.0 : 0.0;
        rec[SSA_UPD_VISIBLE] = visible ? 1.0 : 0.0;
        rec[SSA_UPD_ACTION] = attempted ? (double)(ACT::late ? act : env_action<INL>(p, e)) : -1.0;   // (my_update: the env's action IS this object)
    }
    // The update is the register-pressure peak behind the propagator and only ONE wavefront of a launch runs it: whatever is
    // live across it would be spilled by EVERY wavefront.  So the values that are cheap to get again are got again behind
    // it: the lane coordinates, the object / env, and the next tile's loads (issued a second time).
    asm volatile("" : "+v"(lane));
    g = lane >> 4;
    l = lane & 15;
    obj = base + g;
    e = (valid && p.n_env > 1) ? (int)((uint32_t)obj / (uint32_t)p.n_obj) : 0;
    if (TILE == 1) tile_issue(pf, p, lane, next_base, next_cnt);
    }   // wavefronts holding a selected object
    // envs whose action selects nobody still get a cleared record (written by object 0's row)
    if (valid && p.upd && obj == (int64_t)e * p.n_obj && l == 0) {   // (object 0 of an env: one lane per env)
      const int a_env = ACT::late ? act : env_action<INL>(p, e);
      if (!(a_env >= 0 && interval_ok && (int64_t)a_env < p.n_obj)) {
        double* rec = p.upd + (int64_t)e * SSA_UPD_STRIDE;
        rec[SSA_UPD_OBS_TAKEN] = 0.0;
        rec[SSA_UPD_VISIBLE] = 0.0;
        rec[SSA_UPD_ACTION] = -1.0;
      }
    }

    // ---- F1: failed filters carry the sentinels (ssa_tasker_simple_2.py:157-158, 369-382)
    wave_lds_sync();
    SSA_TR(6);
    if (valid && st_new != SSA_ST_OK && st_in == SSA_ST_OK) {
        for (int idx = l; idx < 36; idx += 16) {
            int a = idx / 6, b = idx - a * 6;
            t.P[g * 36 + idx] = (a == b) ? (a < 3 ? X_FAILED_POS : X_FAILED_VEL) : 0.0;
        }
        if (l < 6) t.X[g * 6 + l] = (l < 3) ? X_FAILED_POS : X_FAILED_VEL;
        if (p.fail_log && l == 0) {
            // filter_error()'s record (:369-382): who, why, when, and error_failed() of the state the filter failed FROM (:376-378) -- the
            // step's inputs, still in HBM (a rare, row-divergent branch: three loads per failing filter)
            if (ACT::late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the closed loop stores the previous step's tile inside this one)
            const double* xt = p.x_true_in + obj * 6;
            const double* xf = p.x_in + obj * 6;
            const double* Pd = p.P_in + obj * 36;
            const double a0 = xf[0] - xt[0], a1 = xf[1] - xt[1], a2 = xf[2] - xt[2];
            const double b0 = xf[3] - xt[3], b1 = xf[4] - xt[4], b2 = xf[5] - xt[5];
            const unsigned at = atomicAdd(p.fail_count, 1u);
            if ((int)at < p.fail_cap) {
                double* rec = p.fail_log + (int64_t)at * SSA_FAIL_STRIDE;
                rec[SSA_FAIL_ENV] = (double)e;
                rec[SSA_FAIL_OBJ] = p.obj_ids ? (double)t.Oid[g] : (double)(obj - (int64_t)e * p.n_obj);
                rec[SSA_FAIL_STATUS] = (double)st_new;
                rec[SSA_FAIL_TIME] = (double)((ACT::late ? p.env_time[0] : env_time_of<INL>(p, e)) + p.time_offset);
                rec[SSA_FAIL_ERR + 0] = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
                rec[SSA_FAIL_ERR + 1] = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
                rec[SSA_FAIL_ERR + 2] = sqrt(Pd[0] + Pd[7] + Pd[14]);
                rec[SSA_FAIL_ERR + 3] = sqrt(Pd[21] + Pd[28] + Pd[35]);
            }
        }
    }
    if (valid && st_in != SSA_ST_OK) {  // already failed: the filter state passes through unchanged (:272)
        if (ACT::late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (its previous state left for HBM earlier in THIS step)
        for (int idx = l; idx < 36; idx += 16) t.P[g * 36 + idx] = p.P_in[obj * 36 + idx];
        if (l < 6) t.X[g * 6 + l] = p.x_in[obj * 6 + l];
    }
    if (l == 0) t.St[g] = st_new;
    wave_lds_sync();
#if !(defined(SSA_ABLATE) && (SSA_ABLATE & 16))
    observe_rows(t, g, l, cnt != OBJ_PER_WAVE);
#endif
    // O4 in the epilogue (atomics-statistics path): the (az, el, range, trace P) block of the NEW state -- the 'aer'
    // observation mode and the multi-GPU all-gather payload -- from the tiles, so that no second pass over x / P (the
    // former post kernel: 6.7 MB re-read per 20 000 objects plus a launch) is needed
    if (p.aer_out && p.stat_shards) {
        if (l < 4 && valid) aer_obs_tile<INL>(t, p, C, g, l, e, p.obj_ids ? (int64_t)e * p.n_obj + t.Oid[g] : obj);
    }
    wave_lds_sync();
    SSA_TR(7);
    if (!ACT::late) {   // (closed_loop_kernel stores the tile itself, AFTER it has announced its part of the decision)
#if !(defined(SSA_ABLATE) && (SSA_ABLATE & 32))
        if (p.spos_tiles && p.stat_shards && lane == 0) {   // the tile's first maximum of sigma_pos (np.argmax for the 'shaped' reward): one slot, no atomics
            unsigned long long key, idx;
            spos_tile_best(t, cnt, obj - (int64_t)e * p.n_obj, key, idx, p.obj_ids != nullptr);
            unsigned long long* slot = (unsigned long long*)p.spos_tiles + 2 * (int64_t)tile;
            __hip_atomic_store(slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (agent scope: SSA_LAUNCH_FOLD_INSIDE reads them in this launch)
            __hip_atomic_store(slot + 1, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        store_tile<TILE != 1>(t, p, lane, base, cnt);
        SSA_TR(8);
        // O3 by sharded atomics: max delta_pos (as ordered bits: non-negative doubles and NaN order like unsigned
        // integers, so NaN wins exactly as in np.max), trinary counts (packed in one word), failures
        const bool one_env = p.n_env == 1 || ((uint32_t)base / (uint32_t)p.n_obj == (uint32_t)(base + cnt - 1) / (uint32_t)p.n_obj);
        if (p.stat_shards && one_env) {
            // common case, the tile lies in one env.  Every lane of a row reads its row's delta_pos, so the COUNTS are ballots:
            // one bit per row (lanes 0, 16, 32, 48) of the comparison's mask, counted by the scalar unit -- no packing into
            // words, no cross-row adds.  The maximum crosses rows by two DPP steps (row_bcast:15 into rows 1 and 3, row_bcast:31
            // into rows 2 and 3) and ends in row 3.
            const double dp = t.Met[g * 4 + 0];
            const bool rv = g < cnt;
            const unsigned long long rowbit = 0x0001000100010001ull;
            const unsigned c4 = (unsigned)__popcll(__ballot(rv && dp < 1e4) & rowbit);
            const unsigned c7 = (unsigned)__popcll(__ballot(rv && dp < 1e7) & rowbit);
            const unsigned nfl = (unsigned)__popcll(__ballot(rv && t.St[g] != 0) & rowbit);
            unsigned long long mx = rv ? ((unsigned long long)__double_as_longlong(dp) & 0x7fffffffffffffffull) : 0ull;
#define SSA_XROW(CTRL, ROWMASK)                                                                                          \
            {                                                                                                            \
                const unsigned long long m2 = ((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, (int)(mx >> 32), CTRL, ROWMASK, 0xF, false) << 32) | \
                                              (unsigned)__builtin_amdgcn_update_dpp(0, (int)(mx & 0xffffffffull), CTRL, ROWMASK, 0xF, false);                 \
                mx = m2 > mx ? m2 : mx;                                                                                  \
            }
            SSA_XROW(0x142, 0xA)   // row_bcast:15
            SSA_XROW(0x143, 0xC)   // row_bcast:31
#undef SSA_XROW
            bool i_fold = false;
            if (lane == 63) {
                const int64_t e_tile = (p.n_env > 1) ? (int64_t)((uint32_t)base / (uint32_t)p.n_obj) : 0;
                unsigned long long* sh = (unsigned long long*)p.stat_shards + ((e_tile * SSA_STAT_SHARDS) + (tile & (SSA_STAT_SHARDS - 1))) * SSA_STAT_SHARD_WORDS;
#ifndef SSA_NO_ATOMICS   // (diagnostic builds only)
                atomicMax(sh, mx);
                atomicAdd(sh + 1, (unsigned long long)c4 | ((unsigned long long)c7 << 32));
                if (nfl) atomicAdd(sh + 2, (unsigned long long)nfl);
#endif
                if (FOLD_OK && (p.launch_mask & SSA_LAUNCH_FOLD_INSIDE)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this tile's sums are in before it is counted
                    i_fold = stat_tile_counted(p, e_tile, tile);
                }
            }
            if (FOLD_OK && (p.launch_mask & SSA_LAUNCH_FOLD_INSIDE) && __any(i_fold)) {    // the env's last tile: every shard of it is complete
                const int64_t e_tile = (p.n_env > 1) ? (int64_t)((uint32_t)base / (uint32_t)p.n_obj) : 0;
                fold_stat_shards_inside((unsigned long long*)p.stat_shards + e_tile * SSA_STAT_SHARDS * SSA_STAT_SHARD_WORDS,
                                        p.stats + e_tile * SSA_STAT_STRIDE, lane);
                if (p.spos_tiles) {
                    int t_lo, t_hi;
                    env_tile_range(p.n_obj, (int)e_tile, t_lo, t_hi);
                    fold_spos_tiles<true>((const unsigned long long*)p.spos_tiles, t_lo, t_hi, p.stats + e_tile * SSA_STAT_STRIDE, lane);
                }
            }
        } else if (p.stat_shards) {   // a tile that straddles envs: one group of atomics per env (lane 0)
          unsigned fold_envs = 0u;    // bit i: env e_first + i was completed by this tile (a tile spans at most four envs)
          const int64_t e_first = (p.n_env > 1) ? (int64_t)((uint32_t)base / (uint32_t)p.n_obj) : 0;
          if (lane == 0) {
            int64_t e_cur = -1;
            unsigned long long mx = 0ull, cnts = 0ull, nf = 0ull;
            int64_t j_run = base - e_first * p.n_obj, e_run = e_first;   // (env, index) of row gg, advanced without dividing
            for (int gg = 0; gg <= cnt; ++gg) {
                while (j_run >= p.n_obj) { j_run -= p.n_obj; ++e_run; }
                const int64_t eg = (gg < cnt) ? e_run : -2;
                ++j_run;
                if (eg != e_cur) {
                    if (e_cur >= 0) {
                        unsigned long long* sh = (unsigned long long*)p.stat_shards + ((e_cur * SSA_STAT_SHARDS) + (tile & (SSA_STAT_SHARDS - 1))) * SSA_STAT_SHARD_WORDS;
                        atomicMax(sh, mx);
                        atomicAdd(sh + 1, cnts);
                        if (nf) atomicAdd(sh + 2, nf);
                        if (FOLD_OK && (p.launch_mask & SSA_LAUNCH_FOLD_INSIDE)) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            if (stat_tile_counted(p, e_cur, tile)) fold_envs |= 1u << (unsigned)(e_cur - e_first);
                        }
                    }
                    e_cur = eg; mx = 0ull; cnts = 0ull; nf = 0ull;
                }
                if (gg < cnt) {
                    const double dp = t.Met[gg * 4 + 0];
                    unsigned long long bits = (unsigned long long)__double_as_longlong(dp) & 0x7fffffffffffffffull;
                    mx = bits > mx ? bits : mx;
                    cnts += (unsigned long long)(dp < 1e4) + ((unsigned long long)(dp < 1e7) << 32);
                    nf += t.St[gg] != 0;
                }
            }
          }
          if (FOLD_OK && (p.launch_mask & SSA_LAUNCH_FOLD_INSIDE)) {
              fold_envs = (unsigned)__builtin_amdgcn_readfirstlane((int)fold_envs);   // (lane 0's word)
              for (int i = 0; i < OBJ_PER_WAVE; ++i)
                  if (fold_envs & (1u << i))
                      fold_stat_shards_inside((unsigned long long*)p.stat_shards + (e_first + i) * SSA_STAT_SHARDS * SSA_STAT_SHARD_WORDS,
                                              p.stats + (e_first + i) * SSA_STAT_STRIDE, lane);      // (straddling tiles: no spos_tiles, see step_launch)
          }
        }
        SSA_TR(9);
#ifdef SSA_TRACE
        if (lane == 0 && tile < 16384 && !__any(my_update)) {   // (the update's wavefront uses words 10-14 for its own phases)
            g_trace[tile * 16 + 13] = __hip_atomic_load(&g_kep_dbg[(blockIdx.x & 16383) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g_trace[tile * 16 + 14] = __hip_atomic_load(&g_kep_dbg[(blockIdx.x & 16383) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
#endif
        // raw-shard consumers (stat_shards_clear): the first tile's wavefront zeroes the shard set the NEXT step accumulates
        // into -- at the very end, so that no wavefront waits for this pointer's kernarg line before its tile loads
        if (tile == 0) {
            if (p.stat_shards_clear) {
                unsigned long long* z = (unsigned long long*)p.stat_shards_clear;
                for (int i = lane; i < p.n_env * SSA_STAT_SHARDS * 4; i += 64) z[(i >> 2) * SSA_STAT_SHARD_WORDS + (i & 3)] = 0ull;   // (the used words)
            }
        }
    }
}

// folds the SSA_STAT_SHARDS accumulators of the atomics path of env e into stats and clears them (one wavefront)
SSA_DEV void fold_stat_shards(unsigned long long* __restrict__ shards, double* __restrict__ stats, int e, int lane,
                              const unsigned long long* __restrict__ spos = nullptr, int64_t n_obj = 0)
{
    static_assert(SSA_STAT_SHARDS == 128, "two shards per lane");
    unsigned long long* sh = shards + ((int64_t)e * SSA_STAT_SHARDS + lane) * SSA_STAT_SHARD_WORDS;
    unsigned long long* sh2 = sh + 64 * SSA_STAT_SHARD_WORDS;
    unsigned long long mx = sh[0], cn = sh[1], nf = sh[2];
    const unsigned long long mb = sh2[0], cb = sh2[1], nb = sh2[2];
    sh[0] = 0ull; sh[1] = 0ull; sh[2] = 0ull;
    sh2[0] = 0ull; sh2[1] = 0ull; sh2[2] = 0ull;
    mx = mb > mx ? mb : mx;
    cn += cb;
    nf += nb;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long m2 = __shfl_down(mx, off, 64), c2 = __shfl_down(cn, off, 64), n2 = __shfl_down(nf, off, 64);
        mx = m2 > mx ? m2 : mx;
        cn += c2;
        nf += n2;
    }
    if (lane == 0 && stats) {
        double* o = stats + (int64_t)e * SSA_STAT_STRIDE;
        o[SSA_STAT_MAX_DPOS] = __longlong_as_double((long long)mx);
        o[SSA_STAT_CNT_LT_1E4] = (double)(cn & 0xffffffffull);
        o[SSA_STAT_CNT_LT_1E7] = (double)(cn >> 32);
        o[SSA_STAT_ARGMAX_SPOS] = -1.0;
        o[SSA_STAT_N_FAILED] = (double)nf;
        o[SSA_STAT_MAX_SPOS] = __builtin_nan("");
        o[6] = 0.0; o[7] = 0.0;
    }
    if (spos && stats) {     // np.argmax / np.max of sigma_pos from the tiles' slots (a kernel boundary behind their writers)
        int t_lo, t_hi;
        env_tile_range(n_obj, e, t_lo, t_hi);
        fold_spos_tiles<false>(spos, t_lo, t_hi, stats + (int64_t)e * SSA_STAT_STRIDE, lane);
    }
}
__global__ void __launch_bounds__(64) reward_fold_kernel(unsigned long long* __restrict__ shards, double* __restrict__ stats,
                                                         const unsigned long long* __restrict__ spos, int64_t n_obj)
{
    fold_stat_shards(shards, stats, blockIdx.x, threadIdx.x, spos, n_obj);
}

// Workgroup -> tile, XCD-aware.  Workgroups are handed to the eight XCDs round-robin (block b runs on XCD b % 8) and every XCD
// has its own L2.  With tile = b, neighbouring tiles -- which share the 128-byte lines of x / x_true (192 B per tile), the
// metrics (32-byte runs) and the status words -- always sat on different XCDs: both fetched the shared input lines and both
// wrote their part of the shared output lines back (PMC at 160 000 objects: reads 1.26x, writes 1.66x the algorithmic
// bytes).  Here the blocks of one XCD take a CONTIGUOUS run of tiles: XCD x owns tiles [x q + min(x, r), ...) with
// q = n / 8, r = n % 8.
SSA_DEV int xcd_tile(int b, int n)
{
    const int x = b & 7, i = b >> 3, q = n >> 3, r = n & 7;
    return x * q + (x < r ? x : r) + i;
}

// Grid-stride over tiles: wavefront w advances tiles w, w + G, w + 2G, ... (G = gridDim.x, chosen by the
// launcher so that every wavefront is resident at once and all get the same number of tiles) with the
// next tile's loads issued while the current one is being worked on.
//
// MULTI = false is the one-tile-per-wavefront instance (every launch up to 20 480 objects): no loop, no staging
// registers.
typedef const __attribute__((address_space(4))) StepK* KernargPtr;
// The kernarg segment of step_fast_kernel as a struct: kernel arguments are laid out in declaration order at their natural
// alignment, exactly as the members of this mirror are (tests/test_abi_and_host.py parses the shipped code object's metadata
// and checks the offset of the by-value block against it).
struct StepFastArgs {
    int ntiles, nwork;
    const double *pre_P_in, *pre_x_in, *pre_x_true_in;
    const int32_t* pre_status;
    StepK k;
};
// The leading arguments -- the two tile counts and the pointers k_arg.p.{P_in, x_in, x_true_in, status} repeated -- are plain
// scalars: they are PRELOADED into SGPRs at wavefront launch (-amdgpu-kernarg-preload-count, _build.py), so the tile's loads -- the first link
// of every wavefront's dependency chain -- leave without waiting for a scalar-memory round trip to the kernarg segment.
template <int PROP, bool MULTI>
__global__ void __launch_bounds__(64, SSA_STEP_WAVES) step_fast_kernel(int ntiles, int nwork, const double* pre_P_in,
                                                                       const double* pre_x_in, const double* pre_x_true_in,
                                                                       const int32_t* pre_status, const StepK k_arg)
{
    __shared__ Tiles t;
    int lane = threadIdx.x;
    const int unit = (int)blockIdx.x;
    if (unit >= nwork) {   // deferred fold of the previous step's statistics: one extra wavefront per env
        fold_stat_shards((unsigned long long*)k_arg.p.stat_shards_prev, k_arg.p.stats_prev, unit - nwork, lane,
                         (const unsigned long long*)k_arg.p.spos_tiles_prev, k_arg.p.n_obj);
        return;
    }
    const int64_t total = (int64_t)k_arg.p.n_env * k_arg.p.n_obj;
    TileRegs pf;
    int tile = xcd_tile(unit, nwork);
    if (!MULTI) {
        const int64_t base = (int64_t)tile * OBJ_PER_WAVE;
        const int cnt = (int)((total - base) < OBJ_PER_WAVE ? (total - base) : OBJ_PER_WAVE);
        tile_dma_issue(t, pf, pre_P_in, pre_x_in, pre_x_true_in, pre_status, lane, base, cnt);
        ActEarly early;
        process_wave<PROP, 0>(t, k_arg.c, k_arg.p, lane, base + (lane >> 4), (lane >> 4) < cnt, base, cnt, pf, 0, 0, tile, early);
        return;
    }
    {
        const int64_t b0 = (int64_t)tile * OBJ_PER_WAVE;
        tile_issue(pf, k_arg.p, lane, b0, tile < ntiles ? (int)((total - b0) < OBJ_PER_WAVE ? (total - b0) : OBJ_PER_WAVE) : 0);
    }
    KernargPtr kp = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned wave_slot, turn = 0;   // issue priority rotated per tile: see rollout_kernel
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(wave_slot));
    for (; tile < ntiles; tile += nwork) {
        switch ((wave_slot + turn++) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
        // the body must compile like a one-tile kernel: re-derive the argument block and the lane id per
        // tile, so that the ~100 argument scalars and the lane-derived LDS addresses are produced on demand
        // instead of being carried around the loop in registers
        asm volatile("" : "+s"(kp));
        asm volatile("" : "+v"(lane));
        const StepK& k = *(const StepK*)((const char*)kp + offsetof(StepFastArgs, k));   // behind the preloaded scalar arguments
        const int64_t base = (int64_t)tile * OBJ_PER_WAVE;
        const int cnt = (int)((total - base) < OBJ_PER_WAVE ? (total - base) : OBJ_PER_WAVE);
        const int nt = tile + nwork;
        const int64_t nbase = (int64_t)nt * OBJ_PER_WAVE;
        const int ncnt = nt < ntiles ? (int)((total - nbase) < OBJ_PER_WAVE ? (total - nbase) : OBJ_PER_WAVE) : 0;
        ActEarly early;
        process_wave<PROP, 1>(t, k.c, k.p, lane, base + (lane >> 4), (lane >> 4) < cnt, base, cnt, pf, nbase, ncnt, tile, early);
        wave_lds_sync();   // the tile's LDS reads (store) precede the next tile's commit
    }
}

// Rollout: K consecutive env steps of the same objects in ONE launch.  An object's trajectory depends on no other
// object (the single update per step touches only the selected one; the statistics are reductions), so a wavefront
// loads its tile once and advances it K steps with state, covariance, truth and status resident in LDS, writing every
// step's outputs to that step's ring slot exactly as K single-step launches would; per-step statistics go to per-step
// shard sets, folded by rollout_fold_kernel.  Needs the actions of all K steps up front (open-loop schedules).
struct RollK {   // ONE kernel argument, so that the per-step re-derivation below can address both halves
    StepK k;
    ssa_rollout_params r;
};
template <int PROP>
__global__ void __launch_bounds__(64, SSA_STEP_WAVES) rollout_kernel(const RollK a, int ntiles, int nwork)
{
    const StepK& k_arg = a.k;
    __shared__ Tiles t;
    int lane = threadIdx.x;
    const int64_t total = (int64_t)k_arg.p.n_env * k_arg.p.n_obj;
    const int E = k_arg.p.n_env, H = a.r.history, K = a.r.n_steps;
    const int64_t sx = total * 6, sP = total * 36, so_ = total * 12, sm = (int64_t)E * 4 * k_arg.p.n_obj, su = (int64_t)E * SSA_UPD_STRIDE;
    TileRegs pf;
    typedef const __attribute__((address_space(4))) RollK* RollArgPtr;
    RollArgPtr kp = (RollArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    for (int tile = xcd_tile((int)blockIdx.x, nwork); tile < ntiles; tile += nwork) {
        const int64_t base = (int64_t)tile * OBJ_PER_WAVE;
        const int cnt = (int)((total - base) < OBJ_PER_WAVE ? (total - base) : OBJ_PER_WAVE);
        {   // the tile's state from the input slot
            const int si0 = (a.r.slot_out + H - 1) % H;
            ssa_step_params p0 = k_arg.p;
            p0.x_true_in = a.r.x_true_ring + si0 * sx;
            p0.x_in = a.r.x_ring + si0 * sx;
            p0.P_in = a.r.P_ring + si0 * sP;
            wave_lds_sync();   // the previous tile's last stores have read the tiles
            tile_issue(pf, p0, lane, base, cnt);
            tile_commit(t, pf, lane);
            if (lane < 36) t.Q[lane] = k_arg.c.Q[lane];   // process_wave<.., 2> expects the process noise in place
        }
        unsigned wave_slot;   // the wavefront's slot on its SIMD (HW_ID bits 3:0)
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(wave_slot));
        for (int kk = 0; kk < K; ++kk) {
#ifndef SSA_ROLL_NOPRIO
            // The SIMD arbiter serves the oldest wavefront first: left alone, the 5 co-resident wavefronts finish
            // their K steps one after the other and the SIMD runs the tail of the launch with 4, 3, 2, 1 of them
            // (latency-bound).  Rotating the issue priority per step keeps them level, so all stay resident and
            // the stages of different wavefronts interleave until the end.
            switch ((wave_slot + (unsigned)kk) & 3u) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
#endif
            asm volatile("" : "+s"(kp));      // per step, as per tile in step_fast_kernel: nothing carried around the loop
            asm volatile("" : "+v"(lane));
            const StepK& k = ((const RollK*)kp)->k;
            const ssa_rollout_params& r = ((const RollK*)kp)->r;
            const int so = (r.slot_out + kk) % H, si = (so + H - 1) % H;
            ssa_step_params pk = k.p;
            pk.time_offset = k.p.time_offset + kk;
            pk.x_true_in = r.x_true_ring + si * sx;  pk.x_true_out = r.x_true_ring + so * sx;
            pk.x_in = r.x_ring + si * sx;            pk.x_out = r.x_ring + so * sx;
            pk.P_in = r.P_ring + si * sP;            pk.P_out = r.P_ring + so * sP;
            pk.obs = r.obs_ring + so * so_;
            pk.metrics = r.metrics_ring + so * sm;
            // per-ENV outputs are written by whichever wavefront owns the selected object, and wavefronts advance at
            // their own pace: only the step that finally owns a ring slot may write it (per-object outputs have one
            // writer, in order)
            pk.upd = (r.upd_ring && kk >= K - H) ? r.upd_ring + so * su : nullptr;
            pk.actions = r.actions + (int64_t)kk * E;
            pk.stat_shards = r.stat_shards + (int64_t)kk * E * SSA_STAT_SHARDS * SSA_STAT_SHARD_WORDS;
            pk.spos_tiles = r.spos_tiles ? r.spos_tiles + (int64_t)kk * ntiles * 2 : nullptr;
            pk.aer_out = nullptr;
            ActEarly early;
            process_wave<PROP, 2>(t, k.c, pk, lane, base + (lane >> 4), (lane >> 4) < cnt, base, cnt, pf, 0, 0, tile, early);
            wave_lds_sync();
        }
    }
}
// grid (n_steps, n_env): folds step k's shard set into the statistics slot of step k -- when that slot still
// belongs to step k at the end of the rollout (the last `history` steps) -- and clears it
__global__ void __launch_bounds__(64) rollout_fold_kernel(unsigned long long* __restrict__ shards, double* __restrict__ stats_ring,
                                                          int n_env, int n_steps, int slot_out, int history,
                                                          const unsigned long long* __restrict__ spos, int64_t n_obj, int64_t ntiles)
{
    const int kk = blockIdx.x, e = blockIdx.y;
    const int so = (slot_out + kk) % history;
    double* dst = (kk >= n_steps - history) ? stats_ring + (int64_t)so * n_env * SSA_STAT_STRIDE : nullptr;
    fold_stat_shards(shards + (int64_t)kk * n_env * SSA_STAT_SHARDS * SSA_STAT_SHARD_WORDS, dst, e, threadIdx.x,
                     spos ? spos + (int64_t)kk * ntiles * 2 : nullptr, n_obj);
}

// Post kernel, grid (nparts, n_env) x 256 threads, launched when a payload or the exact statistics are wanted:
// (1) the (az, el, range, trace P) observation block O4 of this block's slice (aer_out); (2) the statistics: the
// fold of the step kernel's shards in the first wavefront, or -- without stat_shards -- this block's slice of the
// exact reduction (arg-max sigma_pos included), one StatAcc per block for reward_final_kernel.
constexpr int POST_T = 256, POST_ILP = 4;
template <int PROP>
__global__ void __launch_bounds__(POST_T) step_post_kernel(const StepK k, StatAcc* __restrict__ parts, int nparts)
{
    __shared__ StatAcc part[POST_T / 64];
    const ssa_consts& C = k.c;
    const ssa_step_params& p = k.p;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int e = blockIdx.y;
    const int64_t m = p.n_obj;
    StatAcc acc = stat_identity();
    if (p.aer_out) {
        for (int64_t i = (int64_t)blockIdx.x * POST_T + tid; i < m; i += (int64_t)nparts * POST_T) {
            const int64_t obj = (int64_t)e * m + i;
            aer_obs_row(p.x_out + obj * 6, p.P_out + obj * 36, p, C, e, obj);
        }
    }
    if (p.stats) {
        const double* dpos = p.metrics + ((int64_t)e * 4 + 0) * m;
        const double* spos = p.metrics + ((int64_t)e * 4 + 2) * m;
        const int32_t* st = p.status + (int64_t)e * m;
        for (int64_t b0 = (int64_t)blockIdx.x * POST_T * POST_ILP; b0 < m; b0 += (int64_t)nparts * POST_T * POST_ILP) {
            double dp[POST_ILP], sp[POST_ILP];
            int sv[POST_ILP];
#pragma unroll
            for (int q = 0; q < POST_ILP; ++q) {
                int64_t i = b0 + (int64_t)q * POST_T + tid;
                bool in = i < m;
                dp[q] = in ? dpos[i] : 0.0;
                sp[q] = in ? spos[i] : -2.0;
                sv[q] = in ? st[i] : 0;
            }
#pragma unroll
            for (int q = 0; q < POST_ILP; ++q) {
                int64_t i = b0 + (int64_t)q * POST_T + tid;
                if (i < m) {
                    if (dp[q] != dp[q]) acc.mx_nan = 1; else acc.mx = fmax(acc.mx, dp[q]);
                    acc.c4 += dp[q] < 1e4;
                    acc.c7 += dp[q] < 1e7;
                    acc.nf += sv[q] != 0;
                    bool better = acc.sm_nan ? false : ((sp[q] != sp[q]) ? true : (sp[q] > acc.sm || (sp[q] == acc.sm && i < acc.arg)));
                    if (better) { acc.sm = sp[q]; acc.arg = i; acc.sm_nan = (sp[q] != sp[q]); }
                }
            }
        }
        acc = stat_wave_reduce(acc);
        if (lane == 0) part[w] = acc;
        __syncthreads();
        if (tid == 0) {
            StatAcc r = part[0];
            for (int q = 1; q < POST_T / 64; ++q) stat_merge(r, part[q]);
            parts[(int64_t)e * nparts + blockIdx.x] = r;
        }
    }
}


// ------------------------------------------------------------------------------------------
// O3: per-env reward statistics
// Two small launches: `reward_partial_kernel` spreads the 20-byte-per-object read over many CUs
// (a single CU only pulls ~24 GB/s, MI355X_MICROARCH.md) and leaves one StatAcc per block;
// `reward_final_kernel` folds those.  The kernel boundary orders the two, no device-scope fence.
constexpr int STAT_T = 256, STAT_ILP = 4, STAT_MAX_PARTS = 1024;

__global__ void __launch_bounds__(STAT_T) reward_partial_kernel(const double* __restrict__ metrics,
                                                                const int32_t* __restrict__ status,
                                                                StatAcc* __restrict__ parts, int64_t m, int nparts)
{
    const int e = blockIdx.y, t = threadIdx.x;
    const double* dpos = metrics + ((int64_t)e * 4 + 0) * m;
    const double* spos = metrics + ((int64_t)e * 4 + 2) * m;
    const int32_t* st = status + (int64_t)e * m;
    StatAcc a = stat_identity();
    for (int64_t base = (int64_t)blockIdx.x * STAT_T * STAT_ILP; base < m; base += (int64_t)nparts * STAT_T * STAT_ILP) {
        double dp[STAT_ILP], sp[STAT_ILP];
        int sv[STAT_ILP];
#pragma unroll
        for (int k = 0; k < STAT_ILP; ++k) {
            int64_t i = base + (int64_t)k * STAT_T + t;
            bool in = i < m;
            dp[k] = in ? dpos[i] : 0.0;
            sp[k] = in ? spos[i] : -2.0;
            sv[k] = in ? st[i] : 0;
        }
#pragma unroll
        for (int k = 0; k < STAT_ILP; ++k) {
            int64_t i = base + (int64_t)k * STAT_T + t;
            if (i < m) {
                if (dp[k] != dp[k]) a.mx_nan = 1; else a.mx = fmax(a.mx, dp[k]);
                a.c4 += dp[k] < 1e4;
                a.c7 += dp[k] < 1e7;
                a.nf += sv[k] != 0;
                bool better = a.sm_nan ? false : ((sp[k] != sp[k]) ? true : (sp[k] > a.sm));
                if (better) { a.sm = sp[k]; a.arg = i; a.sm_nan = (sp[k] != sp[k]); }
            }
        }
    }
    a = stat_wave_reduce(a);
    __shared__ StatAcc part[STAT_T / 64];
    if ((t & 63) == 0) part[t >> 6] = a;
    __syncthreads();
    if (t == 0) {
        StatAcc r = part[0];
        for (int w = 1; w < STAT_T / 64; ++w) stat_merge(r, part[w]);
        parts[(int64_t)e * nparts + blockIdx.x] = r;
    }
}
__global__ void __launch_bounds__(64) reward_final_kernel(const StatAcc* __restrict__ parts, double* __restrict__ stats, int nparts)
{
    const int e = blockIdx.x, t = threadIdx.x;
    StatAcc a = stat_identity();
    for (int i = t; i < nparts; i += 64) stat_merge(a, parts[(int64_t)e * nparts + i]);
    a = stat_wave_reduce(a);
    if (t == 0 && stats) {
        double* o = stats + (int64_t)e * SSA_STAT_STRIDE;
        o[SSA_STAT_MAX_DPOS] = a.mx_nan ? __builtin_nan("") : a.mx;
        o[SSA_STAT_CNT_LT_1E4] = (double)a.c4;
        o[SSA_STAT_CNT_LT_1E7] = (double)a.c7;
        o[SSA_STAT_ARGMAX_SPOS] = (double)a.arg;
        o[SSA_STAT_N_FAILED] = (double)a.nf;
        o[SSA_STAT_MAX_SPOS] = a.sm_nan ? __builtin_nan("") : a.sm;
        o[6] = 0.0; o[7] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// single-operator kernels (one lane per item)
template <int PROP>
__global__ void propagate_kernel(const double* __restrict__ xin, double* __restrict__ xout, int64_t n, double dt)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = i < n;
    int64_t ii = ok ? i : 0;  // keep the wave convergent for the __all() in the Newton loops
    double x[6], o[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) x[c] = xin[ii * 6 + c];
    kepler_step<PROP>(x, dt, o);
    if (ok) {
#pragma unroll
        for (int c = 0; c < 6; ++c) xout[i * 6 + c] = o[c];
    }
}
__global__ void propagate_hybrid_kernel(const double* __restrict__ xin, double* __restrict__ xout, int64_t n, double dt)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = i < n;
    int64_t ii = ok ? i : 0;  // keep the wave convergent for the __all() in the solver loops
    double x[6], o[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) x[c] = xin[ii * 6 + c];
    bool fast = kepler_hybrid_fast(x, dt, o);
    if (__any(!fast)) {       // the step kernels' tiers: strong-hyperbolic inline, the rest out of line
        if (!fast) fast = kepler_conic_lean<0, true>(x, dt, o);
    }
    if (__any(!fast)) {
        if (!fast) {
            Vec6 si;
#pragma unroll
            for (int c = 0; c < 6; ++c) si.v[c] = x[c];
            Vec6 oo = kepler_beyond_series_tagged<0>(si, dt);
#pragma unroll
            for (int c = 0; c < 6; ++c) o[c] = oo.v[c];
        }
    }
    if (ok) {
#pragma unroll
        for (int c = 0; c < 6; ++c) xout[i * 6 + c] = o[c];
    }
}
__global__ void propagate_j2_kernel(const double* __restrict__ xin, double* __restrict__ xout, int64_t n, double dt, J2Params q)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x[6], o[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) x[c] = xin[i * 6 + c];
    propagate_j2_rk4(x, dt, q, o);
#pragma unroll
    for (int c = 0; c < 6; ++c) xout[i * 6 + c] = o[c];
}
__global__ void elements_kernel(const double* __restrict__ xin, double* __restrict__ coe, int64_t n, double dt)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = i < n;
    int64_t ii = ok ? i : 0;
    double x[6], o[6], dg[8];
#pragma unroll
    for (int c = 0; c < 6; ++c) x[c] = xin[ii * 6 + c];
    kepler_elements(x, dt, o, dg);
    if (ok) {
#pragma unroll
        for (int c = 0; c < 8; ++c) coe[i * 8 + c] = dg[c];
    }
}
__global__ void cholesky_kernel(const double* __restrict__ A, double* __restrict__ U, int32_t* __restrict__ rung, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a[21], u[21];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) a[tri(r, c)] = A[i * 36 + r * 6 + c];
    // scipy.linalg.cholesky checks the WHOLE matrix for non-finite entries
    bool finite = true;
    for (int t = 0; t < 36; ++t) finite = finite && (fabs(A[i * 36 + t]) <= 1.79769313486231570e308);
    int rg = finite ? robust_chol6(a, u) : 16;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) U[i * 36 + r * 6 + c] = (c >= r && rg != 16) ? u[tri(r, c)] : 0.0;
    rung[i] = rg;
}
// U2 as the fused step kernels run it: one wavefront per four matrices, the row-distributed ladder itself (robust_chol_row_lds) plus,
// for the record, WHICH rungs factorise in that arithmetic (chol_lane_ok per lane = per rung; bit 16 = the plain attempt).
__global__ void __launch_bounds__(64) ladder_probe_kernel(const double* __restrict__ A, double scale, int32_t* __restrict__ rung_out,
                                                          int32_t* __restrict__ mask_out, double* __restrict__ U, int64_t n)
{
    __shared__ Tiles t;
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
    const int64_t base = (int64_t)blockIdx.x * OBJ_PER_WAVE;
    const int cnt = (int)((n - base) < OBJ_PER_WAVE ? (n - base) : OBJ_PER_WAVE);
    for (int i = lane; i < OBJ_PER_WAVE * 36; i += 64) {
        const int gg = i / 36;
        t.P[i] = (gg < cnt) ? A[base * 36 + i] : ((i % 36) % 7 == 0 ? 1.0 : 0.0);     // (rows beyond n: the identity)
        t.UA[i] = 0.0;
    }
    wave_lds_sync();
    const int rung = robust_chol_row_lds(t, scale, g, l);
    wave_lds_sync();
    const double* Pg = &t.P[g * 36];
    bool finite = true;
    for (int i = 0; i < 36; ++i) finite = finite && (fabs(Pg[i]) <= 1.79769313486231570e308);
    const bool okl = finite && chol_lane_ok(Pg, scale, JITTER[l]);
    const bool ok0 = finite && chol_lane_ok(Pg, scale, 0.0);
    const unsigned m16 = (unsigned)((__ballot(okl) >> (g * 16)) & 0xFFFFull);
    if (g < cnt) {
        if (l == 0) {
            rung_out[base + g] = rung;
            mask_out[base + g] = (int32_t)(m16 | (ok0 ? 0x10000u : 0u));
        }
        if (U) for (int i = l; i < 36; i += 16) U[(base + g) * 36 + i] = (rung != 16) ? t.UA[g * 36 + i] : 0.0;
    }
}
__global__ void sigma_points_kernel(const double* __restrict__ x, const double* __restrict__ P, double scale,
                                    double* __restrict__ sig, int32_t* __restrict__ fail, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a[21], u[21];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) a[tri(r, c)] = scale * P[i * 36 + r * 6 + c];
    int rg = robust_chol6(a, u);
    fail[i] = rg == 16 ? SSA_ST_PREDICT_LINALG : SSA_ST_OK;
    double xx[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        xx[c] = x[i * 6 + c];
        sig[i * 78 + c] = xx[c];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double uv = (c >= r && rg != 16) ? u[tri(r, c)] : 0.0;
            sig[i * 78 + (r + 1) * 6 + c] = xx[c] + uv;
            sig[i * 78 + (r + 7) * 6 + c] = xx[c] - uv;
        }
}
struct GeoK {
    double enu[9];
    double obs[3];
    double obs_limit;
    double Wi, sum_wm_m1;
};
__global__ void hx_kernel(const double* __restrict__ x, int64_t stride, const double* __restrict__ M, GeoK g,
                          double* __restrict__ z, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double Mm[9], xx[3], zz[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) Mm[c] = M[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) xx[c] = x[i * stride + c];
    hx_aer(xx, Mm, g.enu, g.obs, zz);
#pragma unroll
    for (int c = 0; c < 3; ++c) z[i * 3 + c] = zz[c];
}
__global__ void mean_z_kernel(const double* __restrict__ sig, GeoK g, double* __restrict__ zp, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double u0[3], acc[3] = {0.0, 0.0, 0.0}, a[3], um[3], out[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = sig[i * 39 + c];
    aer2uvw(a, u0);
    for (int s = 1; s < 13; ++s) {
        double u[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = sig[i * 39 + s * 3 + c];
        aer2uvw(a, u);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += u[c] - u0[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) um[c] = u0[c] + (g.sum_wm_m1 * u0[c] + g.Wi * acc[c]);
    uvw2aer(um, out);
#pragma unroll
    for (int c = 0; c < 3; ++c) zp[i * 3 + c] = out[c];
}
__global__ void residual_kernel(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ c, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double aa[3] = {a[i * 3], a[i * 3 + 1], a[i * 3 + 2]}, bb[3] = {b[i * 3], b[i * 3 + 1], b[i * 3 + 2]}, cc[3];
    residual_z_aer(aa, bb, cc);
    c[i * 3] = cc[0]; c[i * 3 + 1] = cc[1]; c[i * 3 + 2] = cc[2];
}
// (tix: optional device word; with it M is the TABLE and the matrix is row (tix[0] + tix_off) % n_time -- for callers inside a captured
// graph, whose time index lives on the device)
SSA_DEV const double* matrix_at(const double* M, const int32_t* tix, int32_t tix_off, int32_t n_time)
{
    if (!tix) return M;
    const int t = tix[0] + tix_off;
    return M + (int64_t)((n_time > 0) ? ((t % n_time) + n_time) % n_time : 0) * 9;
}
__global__ void visible_kernel(const double* __restrict__ x, const double* __restrict__ M0, GeoK g,
                               uint8_t* __restrict__ mask, double* __restrict__ el, int64_t n, const int32_t* __restrict__ tix, int32_t tix_off, int32_t n_time)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* M = matrix_at(M0, tix, tix_off, n_time);
    double Mm[9], xx[3], zz[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) Mm[c] = M[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) xx[c] = x[i * 6 + c];
    hx_aer(xx, Mm, g.enu, g.obs, zz);
    mask[i] = zz[1] >= g.obs_limit ? 1 : 0;
    if (el) el[i] = zz[1];
}
__global__ void observe_kernel(const double* __restrict__ xt, const double* __restrict__ x, const double* __restrict__ P,
                               double* __restrict__ obs, double* __restrict__ met, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double xx[6], dg[6], tt[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        xx[c] = x[i * 6 + c];
        tt[c] = xt[i * 6 + c];
        dg[c] = P[i * 36 + 7 * c];
        obs[i * 12 + c] = xx[c];
        obs[i * 12 + 6 + c] = dg[c];
    }
    double a0 = xx[0] - tt[0], a1 = xx[1] - tt[1], a2 = xx[2] - tt[2];
    double b0 = xx[3] - tt[3], b1 = xx[4] - tt[4], b2 = xx[5] - tt[5];
    met[i] = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
    met[n + i] = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
    met[2 * n + i] = sqrt(dg[0] + dg[1] + dg[2]);
    met[3 * n + i] = sqrt(dg[3] + dg[4] + dg[5]);
}
__global__ void aer_obs_kernel(const double* __restrict__ x, const double* __restrict__ P, const double* __restrict__ M,
                               GeoK g, double* __restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double Mm[9], xx[3], zz[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) Mm[c] = M[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) xx[c] = x[i * 6 + c];
    hx_aer(xx, Mm, g.enu, g.obs, zz);
    double tr = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) tr += P[i * 36 + 7 * c];
    double v[4] = {zz[0], zz[1], zz[2], tr};
#pragma unroll
    for (int c = 0; c < 4; ++c) out[i * 4 + c] = (fabs(v[c]) <= 1.79769313486231570e308) ? v[c] : 0.001;
}

// agents.py score arrays + visibility (one lane per object).  sc[0..3] = trace P | log(det P_cur / det P_prev) | |dpos| | |dvel|;
// WANT selects what is evaluated (bit k = row k, bit 4 = visibility) -- the two log-determinants are the expensive part.
// log det through the (plain) Cholesky factor: det > 0 for the covariances the filter keeps; anything else gives NaN, which
// the arg-max skips
SSA_DEV double logdet_chol(const double (&A)[21])
{
    double U[21];
    double ld = __builtin_nan("");
    if (chol6_upper(A, 0.0, U)) {
        // det P = (prod U_ii)^2: ONE logarithm of the product (as the reference: np.log(np.linalg.det(P)), agents.py:24) instead of
        // six of the factors -- the diagonal of the factor is sqrt-sized (1e-3 .. 1e6), its product far from over- / underflow
        double pr = U[tri(0, 0)];
#pragma unroll
        for (int c = 1; c < 6; ++c) pr *= U[tri(c, c)];
        ld = 2.0 * log(pr);
    }
    return ld;
}
// the scores from values in registers: xt / x = true and estimated state, A = upper triangle of P_cur, ld_prev = logdet_chol of
// P_prev (NaN when there is none); *ld_cur (optional) receives logdet_chol(P_cur) -- the closed-loop kernel hands it on as the
// next step's ld_prev instead of factorising the same matrix twice.  M = GCRS -> ITRS matrix of the step.
template <int WANT>
SSA_DEV bool agent_score_core(const double (&xt)[6], const double (&x)[6], const double (&A)[21], double ld_prev,
                              const double* __restrict__ M, const GeoK& g, double* sc, double* ld_cur)
{
    if (WANT & 1) {
        double tr = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) tr += A[tri(c, c)];
        sc[0] = tr;
    }
    if (WANT & 2) {
        const double ld_c = logdet_chol(A);
        if (ld_cur) *ld_cur = ld_c;
        sc[1] = ld_c - ld_prev;
    }
    if (WANT & 12) {
        double d[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) d[c] = x[c] - xt[c];
        sc[2] = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        sc[3] = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    }
    bool vis = true;
    if (WANT & 16) {
        double Mm[9], xx[3] = {xt[0], xt[1], xt[2]}, zz[3];
#pragma unroll
        for (int c = 0; c < 9; ++c) Mm[c] = M[c];
        hx_aer(xx, Mm, g.enu, g.obs, zz);
        vis = zz[1] >= g.obs_limit;
    }
    return vis;
}
template <int WANT>
SSA_DEV bool agent_score_rows(const double* __restrict__ xt, const double* __restrict__ x, const double* __restrict__ Pc,
                              const double* __restrict__ Pp, const double* __restrict__ M, const GeoK& g, int64_t i, double* sc)
{
    double A[21] = {0.0}, xtv[6] = {0.0}, xv[6] = {0.0};
    if (WANT & 3) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) A[tri(r, c)] = Pc[i * 36 + r * 6 + c];
    }
    double ld_p = __builtin_nan("");
    if ((WANT & 2) && Pp) {
        double B[21];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) B[tri(r, c)] = Pp[i * 36 + r * 6 + c];
        ld_p = logdet_chol(B);
    }
    if (WANT & 12) {
#pragma unroll
        for (int c = 0; c < 6; ++c) xv[c] = x[i * 6 + c];
    }
    if (WANT & 28) {
#pragma unroll
        for (int c = 0; c < 6; ++c) xtv[c] = xt[i * 6 + c];
    }
    return agent_score_core<WANT>(xtv, xv, A, ld_p, M, g, sc, nullptr);
}
__global__ void agent_scores_kernel(const double* __restrict__ xt, const double* __restrict__ x, const double* __restrict__ Pc,
                                    const double* __restrict__ Pp, const double* __restrict__ M0, GeoK g,
                                    double* __restrict__ scores, uint8_t* __restrict__ mask, int64_t n, const int32_t* __restrict__ tix,
                                    int32_t tix_off, int32_t n_time)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* M = mask ? matrix_at(M0, tix, tix_off, n_time) : M0;
    double sc[4];
    bool vis;
    if (mask) vis = agent_score_rows<31>(xt, x, Pc, Pp, M, g, i, sc);
    else vis = agent_score_rows<15>(xt, x, Pc, Pp, M, g, i, sc);
    scores[i] = sc[0];
    scores[n + i] = sc[1];
    scores[2 * n + i] = sc[2];
    scores[3 * n + i] = sc[3];
    if (mask) mask[i] = vis ? 1 : 0;
}

// ---- closed loop on the device: the reference's heuristic agents (agents.py:7-81) choose the next action
// from the state the step just wrote; here that choice never leaves the GPU.  agent_partial_kernel scores one object per
// lane and reduces each block to its first maximum; agent_final_kernel folds the blocks of an env and writes the action
// word the NEXT ssa_env_step_f64 reads (a fallback action when nothing is visible: the reference samples at random).
struct AgentPart { double best; long long arg; };
SSA_DEV void agent_merge(double& b, long long& a, double b2, long long a2)
{
    const bool take = a2 >= 0 && (a < 0 || b2 > b || (b2 == b && a2 < a));
    if (take) { b = b2; a = a2; }
}
constexpr int AGENT_T = 256;
template <int KIND>
__global__ void __launch_bounds__(AGENT_T) agent_partial_kernel(const double* __restrict__ xt, const double* __restrict__ x,
                                                                const double* __restrict__ Pc, const double* __restrict__ Pp,
                                                                const double* __restrict__ trans, const int32_t* __restrict__ env_time,
                                                                int32_t time_offset, int32_t n_time, GeoK g,
                                                                AgentPart* __restrict__ parts, int64_t n, const int32_t* __restrict__ ids)
{
    // (ids: a storage layout's table -- ssa_step_params.obj_ids -- or NULL: the candidate is named as the CALLER numbers it, and the first
    // maximum is the one with the lowest such index)
    const int e = blockIdx.y, t = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * AGENT_T + t;
    double best = 0.0;
    long long arg = -1;
    if (i < n) {
        const int64_t obj = (int64_t)e * n + i;
        const int tix = env_time[e] + time_offset;
        const double* M = trans + (int64_t)((n_time > 0) ? tix % n_time : 0) * 9;
        double sc[4];
        constexpr int ROW = (KIND == SSA_AGENT_SHANNON) ? 1 : (KIND == SSA_AGENT_POS_ERROR) ? 2 : (KIND == SSA_AGENT_VEL_ERROR) ? 3 : 0;
        constexpr int WANT = (ROW == 0 ? 1 : ROW == 1 ? 2 : 12) | (KIND == SSA_AGENT_NAIVE_GREEDY ? 0 : 16);
        const bool vis = agent_score_rows<WANT>(xt, x, Pc, Pp, M, g, obj, sc);
        const double v = sc[ROW];
        if (vis && v == v) { best = v; arg = ids ? (long long)ids[i] : (long long)i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double b2 = __shfl_down(best, off, 64);
        const long long a2 = __shfl_down(arg, off, 64);
        agent_merge(best, arg, b2, a2);
    }
    __shared__ AgentPart part[AGENT_T / 64];
    if ((t & 63) == 0) { part[t >> 6].best = best; part[t >> 6].arg = arg; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < AGENT_T / 64; ++w) agent_merge(best, arg, part[w].best, part[w].arg);
        AgentPart r;
        r.best = best; r.arg = arg;
        parts[(int64_t)e * gridDim.x + blockIdx.x] = r;
    }
}
__global__ void __launch_bounds__(64) agent_final_kernel(const AgentPart* __restrict__ parts, int nparts,
                                                         const int32_t* __restrict__ fallback, int32_t* __restrict__ action_out,
                                                         int64_t* __restrict__ pick_out)
{
    const int e = blockIdx.x, t = threadIdx.x;
    double best = 0.0;
    long long arg = -1;
    for (int i = t; i < nparts; i += 64) agent_merge(best, arg, parts[(int64_t)e * nparts + i].best, parts[(int64_t)e * nparts + i].arg);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double b2 = __shfl_down(best, off, 64);
        const long long a2 = __shfl_down(arg, off, 64);
        agent_merge(best, arg, b2, a2);
    }
    if (t == 0) {
        action_out[e] = (arg >= 0) ? (int32_t)arg : (fallback ? fallback[e] : -1);
        if (pick_out) { pick_out[2 * e] = arg; pick_out[2 * e + 1] = __double_as_longlong(best); }
    }
}

// ------------------------------------------------------------------------------------------
// Closed loop in ONE launch (the reference's `a = agent(obs, env); obs, r, done, _ = env.step(a)` of
// run_environment.py:26-29 for its greedy agents, agents.py:7-81).  As in rollout_kernel a wavefront keeps its four objects'
// state in LDS across the K steps and writes every step's outputs to that step's ring slot; in addition the NEXT step's
// action is decided inside the launch:
//   * after step k every compute wavefront scores its objects (agent_score_core: the arithmetic of ssa_agent_select_f64, so
//     the choices are identical to the multi-launch loop's), merges them with its share of the step's reward statistics into
//     one 32-byte part, stores it and bumps its GROUP's arrival counter (64 wavefronts per group) -- and goes on with step
//     k + 1's predicts, which do not depend on the action;
//   * one SERVICE wavefront per group (extra workgroups of the same launch, no objects of their own) polls that counter, folds
//     the group's 64 parts and bumps the launch's counter; one more service wavefront polls that one, folds the groups, writes
//     the action (or the caller's fallback when nothing qualifies) and the step's statistics, and publishes (decision number,
//     action) in every group's flag word;
//   * a compute wavefront reads its group's flag only where the update needs it (ActLate), so the decision's latency hides
//     behind the predict.
// (First version: the last wavefront to arrive did the folding.  The wavefront that runs the update arrives last, lost three
// more microseconds folding while the others were already in their next predict, so it was the last to arrive at the NEXT
// step too -- every step waited for one wavefront's predict + update + folds in series: 14.4 us per step against 10.2 without
// the wait, build_ablate/closed_loop_timeline.py.  Folders without objects cannot fall behind.)
// No compute wavefront can be two steps ahead (the update of step k + 1 needs every part of step k), hence two sets of parts,
// indexed by the step's parity, suffice; the counters only ever grow.  All words that cross wavefronts are agent-scope atomics
// (coherent across the XCDs' L2s); a store is ordered before the atomic that announces it by waiting for its
// acknowledgement (s_waitcnt vmcnt(0)).  Every wait is bounded: a wavefront that times out publishes the abort generation in
// every flag and all wavefronts leave -- the grid drains whatever happens.  Needs every wavefront resident at once (one tile
// per compute wavefront + the service wavefronts): the launcher checks that against the occupancy the runtime reports and
// refuses otherwise.
// (skey, sarg: the part's first maximum of sigma_pos -- ordered key and object index -- for np.argmax(sigma_pos), the 'shaped' reward;
// exchanged only with SSA_LOOP_ARGMAX_SPOS (`wide`): two more words per part)
struct ClPart { double best; long long arg; unsigned long long mx, cnt, skey, sarg; };   // cnt: [< 1e4] | [< 1e7] << 21 | [failed] << 42
constexpr int CL_PART_WORDS = 8;      // words between parts (64 bytes: six used)
SSA_DEV void cl_merge(ClPart& a, const ClPart& b)
{
    agent_merge(a.best, a.arg, b.best, b.arg);
    a.mx = b.mx > a.mx ? b.mx : a.mx;
    a.cnt += b.cnt;
    if (b.skey > a.skey || (b.skey == a.skey && b.sarg < a.sarg)) { a.skey = b.skey; a.sarg = b.sarg; }
}
SSA_DEV ClPart cl_identity()
{
    ClPart r;
    r.best = 0.0; r.arg = -1; r.mx = 0ull; r.cnt = 0ull; r.skey = 0ull; r.sarg = ~0ull;
    return r;
}
SSA_DEV ClPart cl_load(const unsigned long long* w, bool wide)
{
    ClPart r;
    r.best = __longlong_as_double((long long)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    r.arg = (long long)__hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.mx = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.cnt = __hip_atomic_load(w + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.skey = 0ull; r.sarg = ~0ull;
    if (wide) {
        r.skey = __hip_atomic_load(w + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.sarg = __hip_atomic_load(w + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return r;
}
SSA_DEV void cl_store(unsigned long long* w, const ClPart& v, bool wide)
{
    __hip_atomic_store(w, (unsigned long long)__double_as_longlong(v.best), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 1, (unsigned long long)v.arg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 2, v.mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 3, v.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wide) {
        __hip_atomic_store(w + 4, v.skey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(w + 5, v.sarg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Fold over the 64 lanes, result wave-uniform.  Four DPP rotations leave every row's fold in all of its lanes, four readlanes
// combine the rows (the ds_bpermute shuffle tree this replaces took ~1.5 us per fold: 48 dependent LDS-crossbar round trips).
// The agent's "first maximum" (highest score, lowest index among equals, agent_merge) becomes two folds of integers: the
// score as an order-preserving 64-bit key (0 = no candidate), then the lowest index among the lanes that hold the maximum.
SSA_DEV unsigned long long score_key(double v)   // order-preserving; > 0 for every non-NaN double (-0 ranks as +0)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v + 0.0);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
SSA_DEV double score_of_key(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
SSA_DEV ClPart cl_wave_reduce(const ClPart& a, bool wide)
{
    const unsigned long long key = a.arg >= 0 ? score_key(a.best) : 0ull;
    const unsigned long long top = wave_fold_u64(key, OpMax());
    const unsigned long long cand = (a.arg >= 0 && key == top) ? (unsigned long long)a.arg : ~0ull;
    const unsigned long long who = wave_fold_u64(cand, OpMin());
    ClPart r;
    r.arg = top ? (long long)who : -1;
    r.best = top ? score_of_key(top) : 0.0;
    r.mx = wave_fold_u64(a.mx, OpMax());
    r.cnt = wave_fold_u64(a.cnt, OpAdd());
    r.skey = 0ull; r.sarg = ~0ull;
    if (wide) {
        r.skey = wave_fold_u64(a.skey, OpMax());
        r.sarg = wave_fold_u64((a.skey == r.skey) ? a.sarg : ~0ull, OpMin());
    }
    return r;
}
// own stores acknowledged (visible at agent scope) before what follows
SSA_DEV void cl_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// workspace layout in 8-byte words (ng = groups of 64 compute wavefronts); everything before `parts` must be zero at launch
struct ClLayout {
    int64_t flags, gcount, fcount, parts, gparts, cparts, total;
    int ng;
};
static __host__ __device__ inline ClLayout cl_layout(int nwork)
{
    ClLayout L;
    L.ng = (nwork + 63) / 64;
    L.flags = 0;                                   // [ng] x 16 words: one 128-byte line per group
    L.gcount = (int64_t)L.ng * 16;                 // [ng] x 16: arrivals so far (all steps)
    L.fcount = L.gcount + (int64_t)L.ng * 16;      // x 16: folded groups so far (all steps)
    L.parts = L.fcount + 16;                       // [2][nwork] x CL_PART_WORDS
    L.gparts = L.parts + (int64_t)2 * CL_PART_WORDS * nwork;       // [2][ng] x CL_PART_WORDS
    L.cparts = L.gparts + (int64_t)2 * CL_PART_WORDS * L.ng;       // [2] x CL_PART_WORDS: the part of the wavefront that ran the update (straight to the decision)
    L.total = L.cparts + 2 * CL_PART_WORDS;
    return L;
}

SSA_DEV void closed_loop_prescore(ActLate& a, Tiles& t, int lane, int cnt)
{
    if (a.agent == SSA_AGENT_NAIVE_GREEDY) return;       // (no visibility mask: agents.py:7)
    const int g = lane >> 4, l = lane & 15;
    if (l == 0 && g < cnt) {
        GeoK geo;
        for (int i = 0; i < 9; ++i) geo.enu[i] = a.C->enu[i];
        for (int i = 0; i < 3; ++i) geo.obs[i] = a.C->obs_itrs[i];
        geo.obs_limit = a.C->obs_limit; geo.Wi = a.C->Wi; geo.sum_wm_m1 = a.C->sum_wm_m1;
        double xt[6], x[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, A[21] = {0.0}, sc[4];
#pragma unroll
        for (int c = 0; c < 6; ++c) xt[c] = t.T[g * 6 + c];
        a.vis[g] = agent_score_core<16>(xt, x, A, 0.0, a.M, geo, sc, nullptr) ? 1 : 0;
    }
}

// The wavefront that runs the update of a step announces its part last (the update is three microseconds the others do not
// have): it hands it to the deciding wavefront DIRECTLY instead of through its group's fold -- one exchange level less on the
// path decision -> update -> decision.  Who that is follows from the action alone, so everybody agrees on it: the tile of the
// selected object, or tile 0 when the action selects nobody.
// (slot_of: with a storage layout -- ssa_step_params.obj_ids -- the position of the object the caller calls `act`; NULL: the identity)
SSA_DEV int closer_tile(int act, int64_t total, const int32_t* __restrict__ slot_of = nullptr)
{
    if (!(act >= 0 && (int64_t)act < total)) return 0;
    return (slot_of ? slot_of[act] : act) >> 2;
}
SSA_DEV int block_of_tile(int tile, int n)   // inverse of xcd_tile()
{
    const int q = n >> 3, r = n & 7;
    int x = 0;
#pragma unroll
    for (int c = 1; c < 8; ++c)
        if (tile >= c * q + (c < r ? c : r)) x = c;
    return (tile - (x * q + (x < r ? x : r))) * 8 + x;
}
// bounded wait of a service wavefront: until *counter >= target (returns true), or the launch was abandoned / the wait timed
// out (false; on a timeout the abort generation has been published)
SSA_DEV bool cl_service_wait(const unsigned long long* counter, unsigned long long target, ActLate& ab)
{
    // polling at the LOWEST issue priority: the service wavefronts share their SIMDs with compute wavefronts, and a top-priority
    // polling loop took a third of those SIMDs' issue slots (the compute wavefronts next to them finished microseconds late and
    // everybody waited for them); top priority only for the fold that follows
    __builtin_amdgcn_s_setprio(0);
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned long long v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane_u64(v, 0) >= target) {
            __builtin_amdgcn_s_setprio(3);
            return true;
        }
        const unsigned long long f = __hip_atomic_load(ab.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(lane_u64(f, 0) >> 32) == CL_ABORT_GEN) return false;
        if (wall_clock64() - t0 > ab.timeout) {
            ab.abort_all();
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

struct LoopK {   // ONE kernel argument (see RollK)
    StepK k;
    ssa_closed_loop_params c;
};
SSA_DEV void closed_loop_store(const LoopK* lk, Tiles& t, int lane, int kk, int64_t base, int cnt)
{
    const ssa_closed_loop_params& r = lk->c;
    const int64_t total = lk->k.p.n_obj;
    const int so = (r.slot_out + kk) % r.history;
    ssa_step_params q = lk->k.p;
    q.x_true_out = r.x_true_ring + so * total * 6;
    q.x_out = r.x_ring + so * total * 6;
    q.P_out = r.P_ring + so * total * 36;
    q.obs = r.obs_ring + so * total * 12;
    q.metrics = r.metrics_ring + so * 4 * total;
    store_tile<true>(t, q, lane, base, cnt);
}
template <int PROP>
__global__ void __launch_bounds__(64, SSA_STEP_WAVES) closed_loop_kernel(const LoopK a, int ntiles, int nwork)
{
    const StepK& k_arg = a.k;
    __shared__ Tiles t;
    __shared__ double ld_prev[OBJ_PER_WAVE];   // agent_shannon: log det of each object's covariance one step ago
    __shared__ ClPart rowpart[OBJ_PER_WAVE];
    __shared__ int vis_row[OBJ_PER_WAVE];
    int lane = threadIdx.x;
    const int64_t total = k_arg.p.n_obj;       // (one env)
    const int H = a.c.history, K = a.c.n_steps;
    const int w = (int)blockIdx.x;
    const ClLayout L = cl_layout(nwork);
    unsigned long long* const ws = (unsigned long long*)a.c.workspace;
    ActLate asrc;
    asrc.all_flags = ws + L.flags;
    asrc.err = a.c.error;
    asrc.nflags = L.ng;
    asrc.timeout = a.c.wait_ticks > 0 ? (unsigned long long)a.c.wait_ticks : CL_TIMEOUT_TICKS;
    asrc.aborted = false;
    asrc.last = -1;
    asrc.agent = a.c.agent;
    asrc.vis = vis_row;
    const bool wide = (a.c.flags & SSA_LOOP_ARGMAX_SPOS) != 0u;      // np.argmax(sigma_pos) travels with the parts

    // ---------------- service wavefronts: w in [nwork, nwork + ng) fold one group each, w == nwork + ng decides
    if (w >= nwork) {
        __builtin_amdgcn_s_setprio(3);
        const int G = w - nwork;
        if (G < L.ng) {
            const int gsize = (nwork - G * 64) < 64 ? (nwork - G * 64) : 64;
            asrc.flag = ws + L.flags + (int64_t)G * 16;
            asrc.first = a.c.actions[0];
            unsigned long long target = 0ull;   // arrivals of this group so far, all steps
            for (int kk = 0; kk < K; ++kk) {
                asrc.want = (unsigned)kk;       // this step's action says whose part bypasses the groups
                __builtin_amdgcn_s_setprio(0);
                const int act = asrc.get();
                if (asrc.aborted) return;
                const int cw = block_of_tile(closer_tile(act, total, a.c.slot_of), nwork);
                target += (unsigned long long)(gsize - ((cw >> 6) == G ? 1 : 0));
                if (!cl_service_wait(ws + L.gcount + (int64_t)G * 16, target, asrc)) return;
                const int q = kk & 1;
                ClPart me = cl_identity();
                if (lane < gsize && G * 64 + lane != cw) me = cl_load(ws + L.parts + ((int64_t)q * nwork + (int64_t)G * 64 + lane) * CL_PART_WORDS, wide);
                const ClPart red = cl_wave_reduce(me, wide);
                if (lane == 0) {
                    cl_store(ws + L.gparts + ((int64_t)q * L.ng + G) * CL_PART_WORDS, red, wide);
                    cl_stores_done();
                    __hip_atomic_fetch_add(ws + L.fcount, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
        asrc.flag = ws + L.flags;
        for (int kk = 0; kk < K; ++kk) {
            if (!cl_service_wait(ws + L.fcount, (unsigned long long)(kk + 1) * (unsigned long long)(L.ng + 1), asrc)) return;
            const int q = kk & 1;
            ClPart f = cl_identity();
            if (lane == 63) f = cl_load(ws + L.cparts + (int64_t)q * CL_PART_WORDS, wide);    // the update's wavefront
            for (int i0 = 0; i0 < L.ng; i0 += 128) {     // (two loads in flight per lane: 128 groups = 32 768 objects per round)
                ClPart b0 = cl_identity(), b1 = cl_identity();
                if (i0 + lane < L.ng) b0 = cl_load(ws + L.gparts + ((int64_t)q * L.ng + i0 + lane) * CL_PART_WORDS, wide);
                if (i0 + lane + 64 < L.ng) b1 = cl_load(ws + L.gparts + ((int64_t)q * L.ng + i0 + lane + 64) * CL_PART_WORDS, wide);
                cl_merge(f, b0);
                cl_merge(f, b1);
            }
            f = cl_wave_reduce(f, wide);
            const ssa_closed_loop_params& r = a.c;
            const int action = (f.arg >= 0) ? (int)f.arg : (r.fallback ? r.fallback[kk + 1] : -1);   // (wave-uniform)
            const unsigned long long word = ((unsigned long long)(unsigned)(kk + 1) << 32) | (unsigned long long)(unsigned)action;
            if (!(r.flags & SSA_LOOP_DEBUG_WITHHOLD))      // (diagnostic: the decision is never published -> every wait runs into its bound)
                for (int i = lane; i < L.ng; i += 64)
                    __hip_atomic_store(ws + L.flags + (int64_t)i * 16, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) {   // (for the host: nobody in the launch reads these)
                r.actions[kk + 1] = action;
                if (r.picks) { r.picks[2 * (int64_t)(kk + 1)] = f.arg; r.picks[2 * (int64_t)(kk + 1) + 1] = __double_as_longlong(f.best); }
                double* o = r.stats_out + (int64_t)kk * SSA_STAT_STRIDE;
                o[SSA_STAT_MAX_DPOS] = __longlong_as_double((long long)f.mx);
                o[SSA_STAT_CNT_LT_1E4] = (double)(f.cnt & 0x1fffffull);
                o[SSA_STAT_CNT_LT_1E7] = (double)((f.cnt >> 21) & 0x1fffffull);
                o[SSA_STAT_ARGMAX_SPOS] = wide ? (double)(long long)f.sarg : -1.0;
                o[SSA_STAT_N_FAILED] = (double)(f.cnt >> 42);
                o[SSA_STAT_MAX_SPOS] = wide ? spos_of_key(f.skey) : __builtin_nan("");
                o[6] = 0.0; o[7] = 0.0;
            }
        }
        return;
    }

    // ---------------- compute wavefronts
    const int64_t sx = total * 6, sP = total * 36, so_ = total * 12, sm = 4 * k_arg.p.n_obj;
    const int tile = xcd_tile(w, nwork);
    const int64_t base = (int64_t)tile * OBJ_PER_WAVE;
    const int cnt = (int)((total - base) < OBJ_PER_WAVE ? (total - base) : OBJ_PER_WAVE);
    const int G = w >> 6;
    asrc.flag = ws + L.flags + (int64_t)G * 16;
    asrc.first = a.c.actions[0];
    asrc.pend = -1;
    asrc.base = base;
    asrc.cnt = cnt;
    TileRegs pf;
    typedef const __attribute__((address_space(4))) LoopK* LoopArgPtr;
    LoopArgPtr kp = (LoopArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    GeoK geo;
    {   // the tile's state from the input slot
        const int si0 = (a.c.slot_out + H - 1) % H;
        ssa_step_params p0 = k_arg.p;
        p0.x_true_in = a.c.x_true_ring + si0 * sx;
        p0.x_in = a.c.x_ring + si0 * sx;
        p0.P_in = a.c.P_ring + si0 * sP;
        tile_issue(pf, p0, lane, base, cnt);
        tile_commit(t, pf, lane);
        if (lane < 36) t.Q[lane] = k_arg.c.Q[lane];
        wave_lds_sync();
        if (a.c.agent == SSA_AGENT_SHANNON && (lane & 15) == 0) {   // log det of the covariances the launch starts from
            const int g = lane >> 4;
            double A[21];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c) A[tri(r, c)] = t.P[g * 36 + r * 6 + c];
            ld_prev[g] = logdet_chol(A);
        }
        wave_lds_sync();
    }
    unsigned wave_slot;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(wave_slot));
    bool boost = false;   // this wavefront ran the previous step's update: it is behind the others
    for (int kk = 0; kk < K; ++kk) {
        // issue priority rotated per step (see rollout_kernel) -- except for the wavefront that ran the update: it lost
        // microseconds the others spent on the next predict, and if it stays behind it is the last to announce its part of
        // the NEXT step too, with every wavefront waiting for it.  It catches up at top priority.
        if (boost) __builtin_amdgcn_s_setprio(3);
#ifdef SSA_CL_NOROT   // diagnostic: no rotation of the issue priority
        else __builtin_amdgcn_s_setprio(1);
#else
        else switch ((wave_slot + (unsigned)kk) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
        asm volatile("" : "+s"(kp));
        asm volatile("" : "+v"(lane));
        const StepK& k = ((const LoopK*)kp)->k;
        const ssa_closed_loop_params& r = ((const LoopK*)kp)->c;
        const int so = (r.slot_out + kk) % H, si = (so + H - 1) % H;
        ssa_step_params pk = k.p;
        pk.time_offset = k.p.time_offset + kk;
        pk.x_true_in = r.x_true_ring + si * sx;  pk.x_true_out = r.x_true_ring + so * sx;
        pk.x_in = r.x_ring + si * sx;            pk.x_out = r.x_ring + so * sx;
        pk.P_in = r.P_ring + si * sP;            pk.P_out = r.P_ring + so * sP;
        pk.obs = r.obs_ring + so * so_;
        pk.metrics = r.metrics_ring + so * sm;
        pk.upd = r.upd_out ? r.upd_out + (int64_t)kk * SSA_UPD_STRIDE : nullptr;
        pk.actions = r.actions + kk;
        pk.stat_shards = nullptr;      // the statistics travel with the decision (ClPart)
        pk.stats = nullptr;
        pk.aer_out = nullptr;
        pk.stat_shards_clear = nullptr;
#ifdef SSA_CL_TRACE
        const bool trc = (kk == SSA_CL_TRACE || kk == SSA_CL_TRACE + 1) && w < 8192 && lane == 0;
        unsigned long long* const trp = g_cl_trace + (int64_t)w * 16 + (kk - SSA_CL_TRACE) * 8;
        if (trc) { trp[0] = wall_clock64(); trp[7] = 0ull; }
#define SSA_CLT(i) do { if (trc) trp[i] = wall_clock64(); } while (0)
#define SSA_CLF(b) do { if (trc) trp[7] |= (b); } while (0)
#else
#define SSA_CLT(i) do { } while (0)
#define SSA_CLF(b) do { } while (0)
#endif
        asrc.want = (unsigned)kk;
        asrc.C = &k.c;
        asrc.lk = (const LoopK*)kp;
        asrc.M = k.p.trans + (int64_t)time_row(k.p.env_time[0] + pk.time_offset, k.p.n_time) * 9;
        process_wave<PROP, 2>(t, k.c, pk, lane, base + (lane >> 4), (lane >> 4) < cnt, base, cnt, pf, 0, 0, tile, asrc);
        wave_lds_sync();
        if (asrc.aborted) return;
        boost = tile == closer_tile(asrc.last, total, r.slot_of);           // the update ran here (or would have)
#ifdef SSA_CL_TRACE
        if (trc) { trp[1] = asrc.t_wait; trp[2] = asrc.t_seen; }
        if (boost) SSA_CLF(1ull);
#endif
        SSA_CLT(3);
        // ---- this wavefront's part: the agent's score of its objects (lane 0 of each row) and the step's statistics
#ifdef SSA_CL_NOTREE   // diagnostic: no exchange at all (1: scores still computed, 2: not even those)
        if (SSA_CL_NOTREE == 2) { asrc.pend = kk; continue; }
#endif
        {
            const int g = lane >> 4, l = lane & 15;
            if (l == 0) {
                ClPart me = cl_identity();
                if (g < cnt) {
                    double xt[6], x[6], A[21], sc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int c = 0; c < 6; ++c) { xt[c] = t.T[g * 6 + c]; x[c] = t.X[g * 6 + c]; }
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr)
#pragma unroll
                        for (int c = rr; c < 6; ++c) A[tri(rr, c)] = t.P[g * 36 + rr * 6 + c];
                    // (the visibility of the new true state was evaluated where the wavefront waited for this step's action)
                    const bool vis = (r.agent == SSA_AGENT_NAIVE_GREEDY) || vis_row[g] != 0;
                    double v = 0.0;
                    switch (r.agent) {
                        case SSA_AGENT_NAIVE_GREEDY:
                        case SSA_AGENT_VISIBLE_GREEDY: agent_score_core<1>(xt, x, A, 0.0, nullptr, geo, sc, nullptr); v = sc[0]; break;
                        case SSA_AGENT_SHANNON: {
                            double ld_c;
                            agent_score_core<2>(xt, x, A, ld_prev[g], nullptr, geo, sc, &ld_c);
                            ld_prev[g] = ld_c;
                            v = sc[1];
                            break;
                        }
                        case SSA_AGENT_POS_ERROR: agent_score_core<12>(xt, x, A, 0.0, nullptr, geo, sc, nullptr); v = sc[2]; break;
                        default: agent_score_core<12>(xt, x, A, 0.0, nullptr, geo, sc, nullptr); v = sc[3]; break;
                    }
                    const long long gid = k.p.obj_ids ? (long long)t.Oid[g] : (long long)(base + g);
                    if (vis && v == v) { me.best = v; me.arg = gid; }
                    const double dp = t.Met[g * 4 + 0];
                    me.mx = (unsigned long long)__double_as_longlong(dp) & 0x7fffffffffffffffull;   // (ordered bits; NaN on top: np.max)
                    me.cnt = (unsigned long long)(dp < 1e4) | ((unsigned long long)(dp < 1e7) << 21) | ((unsigned long long)(t.St[g] != 0) << 42);
                    if (wide) { me.skey = spos_key(t.Met[g * 4 + 2]); me.sarg = (unsigned long long)gid; }
                }
                rowpart[g] = me;
            }
        }
        wave_lds_sync();
#ifdef SSA_CL_NOTREE
        if (lane == 0 && rowpart[0].arg == -12345) a.c.stats_out[0] = rowpart[1].best;   // (keeps the scores alive)
        asrc.pend = kk;
        continue;
#endif
        if (lane == 0) {
            ClPart me = rowpart[0];
            cl_merge(me, rowpart[1]);
            cl_merge(me, rowpart[2]);
            cl_merge(me, rowpart[3]);
            const bool closer = tile == closer_tile(asrc.last, total, r.slot_of);
            cl_store(closer ? ws + L.cparts + (int64_t)(kk & 1) * CL_PART_WORDS : ws + L.parts + ((int64_t)(kk & 1) * nwork + w) * CL_PART_WORDS, me, wide);
            cl_stores_done();
            __hip_atomic_fetch_add(closer ? ws + L.fcount : ws + L.gcount + (int64_t)G * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (nobody waits for it here)
        }
        SSA_CLT(4);
        asrc.pend = kk;      // the step's outputs leave inside the next step (ActLate::mid_step)
    }
    if (asrc.pend >= 0) closed_loop_store((const LoopK*)kp, t, lane, asrc.pend, base, cnt);
}

// first maximum of score over mask: ONE workgroup of 16 wavefronts (a policy's arg-max head sits between two step launches: a second launch
// for a cross-workgroup fold would cost more than it saves at 20 000 entries).  16 loads in flight per thread (20 000 entries = two rounds),
// the fold by wavefront shuffles + one LDS exchange of the 16 wavefront results: 3-4 us where the first version (8 loads per round, a
// 10-level __syncthreads tree over 1 024 LDS slots) took 11.8 (profiles/r04_run_policy_timeline.txt).
__global__ void __launch_bounds__(1024) masked_argmax_kernel(const double* __restrict__ score, const uint8_t* __restrict__ mask,
                                                             int64_t n, int64_t* __restrict__ out)
{
    const int t = threadIdx.x;
    double best = 0.0;
    long long arg = -1;
    constexpr int Q = 16;
    for (int64_t b0 = 0; b0 < n; b0 += 1024 * Q) {
        double v[Q];
        bool ok[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t i = b0 + (int64_t)q * 1024 + t;
            const bool in = i < n;
            v[q] = in ? score[i] : 0.0;
            ok[q] = in && (mask ? mask[i] != 0 : true);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {    // (ascending indices per thread: `>` keeps the first maximum)
            const int64_t i = b0 + (int64_t)q * 1024 + t;
            if (ok[q] && v[q] == v[q] && (arg < 0 || v[q] > best)) { best = v[q]; arg = i; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double b2 = __shfl_down(best, off, 64);
        const long long a2 = __shfl_down(arg, off, 64);
        agent_merge(best, arg, b2, a2);
    }
    __shared__ AgentPart part[16];
    if ((t & 63) == 0) { part[t >> 6].best = best; part[t >> 6].arg = arg; }
    __syncthreads();
    if (t < 64) {
        best = (t < 16) ? part[t].best : 0.0;
        arg = (t < 16) ? part[t].arg : -1;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const double b2 = __shfl_down(best, off, 64);
            const long long a2 = __shfl_down(arg, off, 64);
            agent_merge(best, arg, b2, a2);
        }
        if (t == 0) {
            out[0] = arg;
            out[1] = __double_as_longlong(best);
        }
    }
}

// The same over many workgroups, for callers that own a small workspace (a policy's arg-max head inside a replayed graph): ONE workgroup
// reads 20 000 entries at the ~40 GB/s a single CU gets (8.6 us, profiles/r04_run_policy_timeline.txt); here every workgroup of 256 lanes
// folds 2 048 entries, stores its part (agent-scope stores), takes a ticket, and the LAST one to arrive folds the parts.  ws: [ticket | pad to
// 64 bytes | parts]; the ticket wraps back to 0 with the last arrival (atomicInc), so the workspace needs zeroing ONCE.  One call at a time
// per workspace (calls in one stream are).
constexpr int AMAX_T = 256, AMAX_Q = 8;
__global__ void __launch_bounds__(AMAX_T) masked_argmax_blocks_kernel(const double* __restrict__ score, const uint8_t* __restrict__ mask,
                                                                      int64_t n, int64_t* __restrict__ out, unsigned long long* __restrict__ ws)
{
    const int t = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * (AMAX_T * AMAX_Q);
    double best = 0.0;
    long long arg = -1;
    {
        double v[AMAX_Q];
        bool ok[AMAX_Q];
#pragma unroll
        for (int q = 0; q < AMAX_Q; ++q) {
            const int64_t i = b0 + q * AMAX_T + t;
            const bool in = i < n;
            v[q] = in ? score[i] : 0.0;
            ok[q] = in && (mask ? mask[i] != 0 : true);
        }
#pragma unroll
        for (int q = 0; q < AMAX_Q; ++q) {
            const int64_t i = b0 + q * AMAX_T + t;
            if (ok[q] && v[q] == v[q] && (arg < 0 || v[q] > best)) { best = v[q]; arg = i; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double b2 = __shfl_down(best, off, 64);
        const long long a2 = __shfl_down(arg, off, 64);
        agent_merge(best, arg, b2, a2);
    }
    __shared__ AgentPart part[AMAX_T / 64];
    __shared__ int last;
    if ((t & 63) == 0) { part[t >> 6].best = best; part[t >> 6].arg = arg; }
    __syncthreads();
    unsigned long long* parts = ws + 8;
    if (t == 0) {
        for (int w = 1; w < AMAX_T / 64; ++w) agent_merge(best, arg, part[w].best, part[w].arg);
        __hip_atomic_store(parts + 2 * blockIdx.x, (unsigned long long)__double_as_longlong(best), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(parts + 2 * blockIdx.x + 1, (unsigned long long)arg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicInc(reinterpret_cast<unsigned int*>(ws), gridDim.x - 1) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last || t >= 64) return;
    __threadfence();
    best = 0.0; arg = -1;
    for (int i = t; i < (int)gridDim.x; i += 64) {     // (ascending block index per lane; ties go to the lower object index in agent_merge)
        const double b2 = __longlong_as_double((long long)__hip_atomic_load(parts + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const long long a2 = (long long)__hip_atomic_load(parts + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        agent_merge(best, arg, b2, a2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double b2 = __shfl_down(best, off, 64);
        const long long a2 = __shfl_down(arg, off, 64);
        agent_merge(best, arg, b2, a2);
    }
    if (t == 0) {
        out[0] = arg;
        out[1] = __double_as_longlong(best);
    }
}

// ---- diagnostic reductions of SURVEY 8f-4 (ssa_tasker_simple_2.py:436-446, 750-775): NEES = d^T inv(P) d with
// d = x_true - x_filter, NIS = y^T inv(S) y.  One lane per (step, object): Gaussian elimination with partial pivoting on
// the augmented system [P | d] (the arithmetic class of numpy.linalg.inv's LU), then the dot product.
__global__ void nees_kernel(const double* __restrict__ xt, const double* __restrict__ x, const double* __restrict__ P,
                            double* __restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double A[6][7], d[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        d[r] = xt[i * 6 + r] - x[i * 6 + r];
#pragma unroll
        for (int c = 0; c < 6; ++c) A[r][c] = P[i * 36 + r * 6 + c];
        A[r][6] = d[r];
    }
    bool singular = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        // pivot: the largest |A[r][k]|, r >= k, swapped into row k with compile-time indices (registers, no scratch)
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const bool sw = fabs(A[r][k]) > fabs(A[k][k]);
#pragma unroll
            for (int c = k; c < 7; ++c) {
                const double a = A[k][c], b = A[r][c];
                A[k][c] = sw ? b : a;
                A[r][c] = sw ? a : b;
            }
        }
        singular = singular || (A[k][k] == 0.0);
        const double inv = 1.0 / A[k][k];
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const double f = A[r][k] * inv;
#pragma unroll
            for (int c = k + 1; c < 7; ++c) A[r][c] = fma(-f, A[k][c], A[r][c]);
        }
    }
    double z[6];
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double acc = A[r][6];
#pragma unroll
        for (int c = r + 1; c < 6; ++c) acc = fma(-A[r][c], z[c], acc);
        z[r] = acc / A[r][r];
    }
    double q = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) q = fma(d[r], z[r], q);
    out[i] = singular ? __builtin_nan("") : q;   // numpy raises LinAlgError('Singular matrix') there
}
__global__ void nis_kernel(const double* __restrict__ y, const double* __restrict__ S, double* __restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s[9], si[9], yy[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) s[c] = S[i * 9 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) yy[c] = y[i * 3 + c];
    if (!inv3(s, si)) { out[i] = __builtin_nan(""); return; }
    double q = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) q += yy[a] * (si[a * 3] * yy[0] + si[a * 3 + 1] * yy[1] + si[a * 3 + 2] * yy[2]);
    out[i] = q;
}

// fitness_test()'s chi-square containment (ssa_tasker_simple_2.py:750-775): how many of v[0..n) lie strictly inside (lo, hi) -- the
// two-sided 95 % critical points of chi2(df) the caller passes in -- next to the number of non-NaN entries (the reference drops
// NaN NIS values before taking the mean, :757, and keeps them in the NEES mean, :771).  counts[0] = inside, counts[1] = non-NaN.
__global__ void __launch_bounds__(64) chi2_zero_kernel(unsigned long long* __restrict__ counts)
{
    if (threadIdx.x < 2) counts[threadIdx.x] = 0ull;
}
__global__ void __launch_bounds__(256) chi2_contained_kernel(const double* __restrict__ v, int64_t n, double lo, double hi,
                                                             unsigned long long* __restrict__ counts)
{
    unsigned inside = 0, valid = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = v[i];
        inside += (x > lo) && (x < hi);
        valid += (x == x);
    }
    // per wavefront: two ballots per 64 values would need a loop of its own; a shuffle tree of two 32-bit counters is 12 steps
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        inside += __shfl_down(inside, off, 64);
        valid += __shfl_down(valid, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (inside) atomicAdd(counts, (unsigned long long)inside);
        if (valid) atomicAdd(counts + 1, (unsigned long long)valid);
    }
}

static GeoK make_geo(const ssa_consts* c)
{
    GeoK g;
    for (int i = 0; i < 9; ++i) g.enu[i] = c->enu[i];
    for (int i = 0; i < 3; ++i) g.obs[i] = c->obs_itrs[i];
    g.obs_limit = c->obs_limit;
    g.Wi = c->Wi;
    g.sum_wm_m1 = c->sum_wm_m1;
    return g;
}
static inline int launch_status()
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return SSA_OK;
    fprintf(stderr, "libssa_hip: kernel launch failed: %s (%s)\n", hipGetErrorName(e), hipGetErrorString(e));   // fail loudly
    return SSA_E_LAUNCH;
}
static inline unsigned nblk(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }


// ---- all-gather by direct peer stores (SURVEY 8e "Collective"): every rank writes its payload of one step straight into the slot it owns
// in every peer's receive buffer (pointers mapped once over hipIpc, host side: ssa-gym_amd/peer.py) and then raises a per-source flag word
// next to it; a consumer waits for the flags of the step it wants.  Nothing here is a collective: no communicator, no rendezvous inside the
// launch, plain kernels a hipGraph captures at any world size.  The step number a launch stands for is seq_base[0] + seq_off (seq_base in
// device memory: a replayed graph advances it on the device, as the time index of the step launches).
//   ordering: every lane's stores, a system-scope fence, the workgroup barrier, then ONE release store of the flag (system scope): a peer
//   that reads the flag >= seq with an acquire load sees the whole payload.  Flags only grow (a 64-bit step counter).
//   a push is PEER_SPLIT workgroups per peer (one workgroup moves 160 KB at a single CU's ~40 GB/s: 4 us); each fences at system scope and
//   takes a ticket, the last one raises the flag.  prev_flags (optional): the rank's OWN flag words of the previous step's buffer -- the
//   workgroups that write to peer r first wait (bounded) until r's payload of step seq - 1 has arrived here, by which time r's readers of the
//   slot about to be overwritten (step seq - 3) have passed in r's stream order: the reuse rule of the rotating buffers inside the push itself,
//   one dispatch per step instead of two.
constexpr int PEER_SPLIT = 8;
__global__ void __launch_bounds__(256) peer_push_kernel(const double* __restrict__ src, int64_t n_words, double* const* __restrict__ dst,
                                                        unsigned long long* const* __restrict__ flag,
                                                        const unsigned long long* __restrict__ seq_base, unsigned long long seq_off,
                                                        const unsigned long long* __restrict__ prev_flags, long long timeout_ticks,
                                                        int32_t* __restrict__ error, unsigned int* __restrict__ tickets)
{
    const int r = blockIdx.x, sl = blockIdx.y, t = threadIdx.x;
    const unsigned long long seq = seq_base[0] + seq_off;
    if (prev_flags && seq >= 2ull) {
        if (t == 0) {
            const long long t0 = (long long)wall_clock64();
            while (__hip_atomic_load(&prev_flags[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq - 1ull) {
                if ((long long)wall_clock64() - t0 > timeout_ticks) {
                    if (error) atomicMax(error, 1 + r);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
    }
    double* __restrict__ d = dst[r];
    int64_t chunk = (n_words + gridDim.y - 1) / gridDim.y;
    chunk += chunk & 1;                                          // (even: the slices keep the 16-byte alignment of the whole)
    const int64_t lo = (int64_t)sl * chunk, hi = (lo + chunk < n_words) ? lo + chunk : n_words;
    if (lo < hi) {
        const bool wide = ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
        if (wide) {
            const double2* __restrict__ s2 = reinterpret_cast<const double2*>(src + lo);
            double2* __restrict__ d2 = reinterpret_cast<double2*>(d + lo);
            const int64_t n2 = (hi - lo) >> 1;
            for (int64_t i = t; i < n2; i += blockDim.x) d2[i] = s2[i];
            if (((hi - lo) & 1) && t == 0) d[hi - 1] = src[hi - 1];
        } else {
            for (int64_t i = lo + t; i < hi; i += blockDim.x) d[i] = src[i];
        }
    }
    __threadfence_system();
    __syncthreads();
    if (t == 0 && atomicInc(&tickets[r], gridDim.y - 1) == gridDim.y - 1) {     // (the ticket wraps back to 0 with the last arrival)
        __threadfence_system();
        __hip_atomic_store(flag[r], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// one wavefront: lane r waits for source r's flag to reach the step; `error` (optional) is set to 1 + r of a source that did not arrive
// within timeout_ticks of the 100 MHz wall clock -- the wait ALWAYS ends (a peer that died must not hang the queue).
__global__ void __launch_bounds__(64) peer_wait_kernel(const unsigned long long* __restrict__ flags, int32_t n_src,
                                                       const unsigned long long* __restrict__ seq_base, unsigned long long seq_off,
                                                       long long timeout_ticks, int32_t* __restrict__ error)
{
    const unsigned long long want = seq_base[0] + seq_off;
    const long long t0 = (long long)wall_clock64();
    for (int r = threadIdx.x; r < n_src; r += 64) {
        while (__hip_atomic_load(&flags[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
            if ((long long)wall_clock64() - t0 > timeout_ticks) {
                if (error) atomicMax(error, 1 + r);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

}  // namespace ssa

// ============================================================================= C ABI
using namespace ssa;

extern "C" {

int ssa_abi_version(void) { return SSA_ABI_VERSION; }
#ifdef SSA_CL_TRACE
int ssa_debug_cl_trace_copy(void* host, int64_t nbytes) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cl_trace), (size_t)nbytes) == hipSuccess ? 0 : -1; }
#endif
#ifdef SSA_TRACE
int ssa_debug_trace_copy(void* host, int64_t nbytes) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), (size_t)nbytes) == hipSuccess ? 0 : -1; }
#endif
const char* ssa_build_info(void) { return "libssa_hip gfx950 fp64 (" __DATE__ " " __TIME__ ")"; }

static int device_cu_count()
{
    static int cached = 0;   // per process: one GPU per process (the launcher model of this library)
    if (cached <= 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached = n;
    }
    return cached;
}
static int post_parts(int64_t n_obj, int32_t n_env)
{
    (void)n_env;
    int64_t want = (n_obj + POST_T - 1) / POST_T;   // one payload row per thread (the statistics slice strides)
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    return (int)want;
}

static int step_launch(const ssa_consts* c, const ssa_step_params* p, void* stream, hipEvent_t ev0, hipEvent_t ev1)
{
    if (!c || !p || p->n_obj <= 0 || p->n_env <= 0) return SSA_E_INVALID;
    if (!p->x_true_in || !p->x_true_out || !p->x_in || !p->x_out || !p->P_in || !p->P_out || !p->status ||
        !p->obs || !p->metrics || !p->trans || !p->env_time || !p->z_noise || !p->stat_ws)
        return SSA_E_INVALID;
    if (p->launch_mask & SSA_LAUNCH_INLINE_ENVS) {
        if (p->n_env > SSA_INLINE_ENVS || (p->launch_mask & SSA_LAUNCH_INLINE_ACTION)) return SSA_E_INVALID;
    } else if (p->launch_mask & SSA_LAUNCH_INLINE_ACTION) {
        if (p->n_env != 1) return SSA_E_INVALID;
    } else if (!p->actions) return SSA_E_INVALID;
    if ((p->launch_mask & SSA_LAUNCH_FOLD_INSIDE) &&
        (!p->stat_shards || !p->stats || (p->launch_mask & SSA_LAUNCH_DEFER_FOLD))) return SSA_E_INVALID;
    if (p->spos_tiles && !p->stat_shards) return SSA_E_INVALID;
    if (p->fail_log && (!p->fail_count || p->fail_cap <= 0)) return SSA_E_INVALID;
    if ((p->launch_mask & SSA_LAUNCH_MIRROR_F32) && !p->stat_shards) return SSA_E_UNSUPPORTED;   // (the post kernel writes aer_out in double)
    // (the statistics of the one-launch paths: the post kernel's arg-max would speak storage positions; several envs: one table row per env,
    // indices within the env, whole tiles per env)
    if (p->obj_ids && (!p->stat_shards || (p->n_env != 1 && (p->n_obj % OBJ_PER_WAVE) != 0))) return SSA_E_UNSUPPORTED;
    if ((p->spos_tiles || p->spos_tiles_prev) && p->n_env > 1 && (p->n_obj % OBJ_PER_WAVE) != 0) return SSA_E_UNSUPPORTED;   // whole tiles per env
    if (c->obs_type != SSA_OBS_AER && c->obs_type != SSA_OBS_XYZ) return SSA_E_INVALID;
    if (p->aer_cols != 0 && p->aer_cols != 1 && p->aer_cols != 4) return SSA_E_INVALID;
    StepK k;
    k.c = *c;
    k.p = *p;
    const int64_t total = (int64_t)p->n_env * p->n_obj;
    if (total >= ((int64_t)1 << 31)) return SSA_E_INVALID;
    // tiles per wavefront T = ceil(tiles / resident wavefront slots); G = ceil(tiles / T) wavefronts
    const int64_t ntiles = (total + OBJ_PER_WAVE - 1) / OBJ_PER_WAVE;
#ifdef SSA_SLOTS_DIV   // (diagnostic: fewer, longer wavefronts)
    const int64_t slots = (int64_t)device_cu_count() * 4 * SSA_STEP_WAVES / SSA_SLOTS_DIV;
#else
    const int64_t slots = (int64_t)device_cu_count() * 4 * SSA_STEP_WAVES;
#endif
    const int64_t per_wave = (ntiles + slots - 1) / slots;
    const int nwork = (int)((ntiles + per_wave - 1) / per_wave);
    const bool fast_stats = p->stat_shards != nullptr;   // statistics by the common-path kernel's atomics
    const bool defer = fast_stats && (p->launch_mask & SSA_LAUNCH_DEFER_FOLD);
    if (defer && p->stat_shards_prev && (!p->stats_prev || p->stat_shards_prev == p->stat_shards)) return SSA_E_INVALID;
    const int nfold = (defer && p->stat_shards_prev) ? p->n_env : 0;
    dim3 grid((unsigned)(nwork + nfold)), block(64);
    const int nparts = post_parts(p->n_obj, p->n_env);
    StatAcc* parts = (StatAcc*)p->stat_ws;
    hipStream_t s = (hipStream_t)stream;
    const unsigned mask = (p->launch_mask & 7u) ? (p->launch_mask & 7u) : 7u;   // diagnostic: time one launch alone
    if (c->propagator != SSA_PROP_FG && c->propagator != SSA_PROP_ELEMENTS && c->propagator != SSA_PROP_J2_RK4 && c->propagator != SSA_PROP_HYBRID) return SSA_E_INVALID;
    if (c->propagator == SSA_PROP_J2_RK4 && (c->rk4_substeps < 1 || c->rk4_substeps > 4096)) return SSA_E_INVALID;
    const int prop = c->propagator;
    if (mask & 1u) {   // (ev0, ev1: dispatch timestamps of this kernel for ssa_env_step_profiled_f64, else null)
        const int nt = (int)ntiles;
        if (per_wave == 1) {
            if (prop == SSA_PROP_FG) hipExtLaunchKernelGGL((step_fast_kernel<1, false>), grid, block, 0, s, ev0, ev1, 0, nt, nwork, p->P_in, p->x_in, p->x_true_in, p->status, k);
            else if (prop == SSA_PROP_ELEMENTS) hipExtLaunchKernelGGL((step_fast_kernel<0, false>), grid, block, 0, s, ev0, ev1, 0, nt, nwork, p->P_in, p->x_in, p->x_true_in, p->status, k);
            else if (prop == SSA_PROP_HYBRID) hipExtLaunchKernelGGL((step_fast_kernel<3, false>), grid, block, 0, s, ev0, ev1, 0, nt, nwork, p->P_in, p->x_in, p->x_true_in, p->status, k);
            else hipExtLaunchKernelGGL((step_fast_kernel<2, false>), grid, block, 0, s, ev0, ev1, 0, nt, nwork, p->P_in, p->x_in, p->x_true_in, p->status, k);
        } else {
            if (prop == SSA_PROP_FG) hipExtLaunchKernelGGL((step_fast_kernel<1, true>), grid, block, 0, s, ev0, ev1, 0, nt, nwork, p->P_in, p->x_in, p->x_true_in, p->status, k);
            else if (prop == SSA_PROP_ELEMENTS) hipExtLaunchKernelGGL((step_fast_kernel<0, true>), grid, block, 0, s, ev0, ev1, 0, nt, nwork, p->P_in, p->x_in, p->x_true_in, p->status, k);
            else if (prop == SSA_PROP_HYBRID) hipExtLaunchKernelGGL((step_fast_kernel<3, true>), grid, block, 0, s, ev0, ev1, 0, nt, nwork, p->P_in, p->x_in, p->x_true_in, p->status, k);
            else hipExtLaunchKernelGGL((step_fast_kernel<2, true>), grid, block, 0, s, ev0, ev1, 0, nt, nwork, p->P_in, p->x_in, p->x_true_in, p->status, k);
        }
    }
    if (fast_stats) {   // (the aer_out payload, if any, was the step kernel's epilogue) a one-wave fold finishes the step
                        // (2 launches), unless deferred (1 launch)
        if ((p->launch_mask & SSA_LAUNCH_FOLD_INSIDE) && per_wave == 1) return launch_status();   // (folded by the step kernel's last wavefronts)
        if ((mask & 6u) && !defer && p->stats)   // (stats NULL: the caller consumes the raw shard words, see stat_shards_clear)
            hipLaunchKernelGGL(reward_fold_kernel, dim3(p->n_env), dim3(64), 0, s, (unsigned long long*)p->stat_shards, p->stats,
                               (const unsigned long long*)p->spos_tiles, p->n_obj);
        return launch_status();
    }
    if (mask & 2u) {
        if (prop == SSA_PROP_FG) hipLaunchKernelGGL(step_post_kernel<1>, dim3(nparts, p->n_env), dim3(POST_T), 0, s, k, parts, nparts);
        else if (prop == SSA_PROP_ELEMENTS) hipLaunchKernelGGL(step_post_kernel<0>, dim3(nparts, p->n_env), dim3(POST_T), 0, s, k, parts, nparts);
        else if (prop == SSA_PROP_HYBRID) hipLaunchKernelGGL(step_post_kernel<3>, dim3(nparts, p->n_env), dim3(POST_T), 0, s, k, parts, nparts);
        else hipLaunchKernelGGL(step_post_kernel<2>, dim3(nparts, p->n_env), dim3(POST_T), 0, s, k, parts, nparts);
    }
    // folds the per-block statistics
    if ((mask & 4u) && p->stats)
        hipLaunchKernelGGL(reward_final_kernel, dim3(p->n_env), dim3(64), 0, s, (const StatAcc*)parts, p->stats, nparts);
    return launch_status();
}
int ssa_env_step_f64(const ssa_consts* c, const ssa_step_params* p, void* stream)
{
    return step_launch(c, p, stream, nullptr, nullptr);
}
// dispatch-timestamp event pairs, created on first use (a ring, so that back-to-back launches can be timed
// without draining the queue after each of them)
static hipEvent_t g_prof_ev[SSA_PROFILE_SLOTS][2];
static bool g_prof_made[SSA_PROFILE_SLOTS];
int ssa_env_step_profiled_f64(const ssa_consts* c, const ssa_step_params* p, void* stream, int32_t slot)
{
    if (slot < 0 || slot >= SSA_PROFILE_SLOTS) return SSA_E_INVALID;
    if (!g_prof_made[slot]) {
        if (hipEventCreate(&g_prof_ev[slot][0]) != hipSuccess || hipEventCreate(&g_prof_ev[slot][1]) != hipSuccess) return SSA_E_LAUNCH;
        g_prof_made[slot] = true;
    }
    return step_launch(c, p, stream, g_prof_ev[slot][0], g_prof_ev[slot][1]);
}
int ssa_env_step_profile_ms(int32_t slot, float* kernel_ms)
{
    if (slot < 0 || slot >= SSA_PROFILE_SLOTS || !kernel_ms || !g_prof_made[slot]) return SSA_E_INVALID;
    if (hipEventSynchronize(g_prof_ev[slot][1]) != hipSuccess ||
        hipEventElapsedTime(kernel_ms, g_prof_ev[slot][0], g_prof_ev[slot][1]) != hipSuccess) return SSA_E_LAUNCH;
    return SSA_OK;
}
int ssa_env_rollout_f64(const ssa_consts* c, const ssa_step_params* p, const ssa_rollout_params* r, void* stream)
{
    if (!c || !p || !r || p->n_obj <= 0 || p->n_env <= 0 || r->n_steps < 1 || r->history < 2) return SSA_E_INVALID;
    if (r->slot_out < 0 || r->slot_out >= r->history) return SSA_E_INVALID;
    if (!r->x_true_ring || !r->x_ring || !r->P_ring || !r->obs_ring || !r->metrics_ring || !r->stats_ring || !r->actions || !r->stat_shards)
        return SSA_E_INVALID;
    if (!p->status || !p->trans || !p->env_time || !p->z_noise) return SSA_E_INVALID;
    if (c->obs_type != SSA_OBS_AER && c->obs_type != SSA_OBS_XYZ) return SSA_E_INVALID;
    if (c->propagator != SSA_PROP_FG && c->propagator != SSA_PROP_ELEMENTS && c->propagator != SSA_PROP_J2_RK4 && c->propagator != SSA_PROP_HYBRID) return SSA_E_INVALID;
    if (c->propagator == SSA_PROP_J2_RK4 && (c->rk4_substeps < 1 || c->rk4_substeps > 4096)) return SSA_E_INVALID;
    const int64_t total = (int64_t)p->n_env * p->n_obj;
    if (total >= ((int64_t)1 << 31)) return SSA_E_INVALID;
    if (r->spos_tiles && p->n_env > 1 && (p->n_obj % OBJ_PER_WAVE) != 0) return SSA_E_UNSUPPORTED;
    RollK rk;
    rk.k.c = *c;
    rk.k.p = *p;
    rk.k.p.aer_out = nullptr;
    rk.k.p.spos_tiles = nullptr;
    rk.k.p.spos_tiles_prev = nullptr;
    if (p->obj_ids && p->n_env != 1) return SSA_E_UNSUPPORTED;
    rk.r = *r;
    const int64_t ntiles = (total + OBJ_PER_WAVE - 1) / OBJ_PER_WAVE;
    const int64_t slots = (int64_t)device_cu_count() * 4 * SSA_STEP_WAVES;
    const int64_t per_wave = (ntiles + slots - 1) / slots;
    const int nwork = (int)((ntiles + per_wave - 1) / per_wave);
    hipStream_t s = (hipStream_t)stream;
    if (c->propagator == SSA_PROP_FG) hipLaunchKernelGGL(rollout_kernel<1>, dim3(nwork), dim3(64), 0, s, rk, (int)ntiles, nwork);
    else if (c->propagator == SSA_PROP_ELEMENTS) hipLaunchKernelGGL(rollout_kernel<0>, dim3(nwork), dim3(64), 0, s, rk, (int)ntiles, nwork);
    else if (c->propagator == SSA_PROP_HYBRID) hipLaunchKernelGGL(rollout_kernel<3>, dim3(nwork), dim3(64), 0, s, rk, (int)ntiles, nwork);
    else hipLaunchKernelGGL(rollout_kernel<2>, dim3(nwork), dim3(64), 0, s, rk, (int)ntiles, nwork);
    hipLaunchKernelGGL(rollout_fold_kernel, dim3(r->n_steps, p->n_env), dim3(64), 0, s, (unsigned long long*)r->stat_shards, r->stats_ring,
                       p->n_env, r->n_steps, r->slot_out, r->history, (const unsigned long long*)r->spos_tiles, p->n_obj, ntiles);
    return launch_status();
}
int64_t ssa_closed_loop_workspace_bytes(int64_t n_obj, int32_t n_env)
{
    if (n_obj <= 0 || n_env != 1) return 0;
    const int64_t ntiles = (n_obj + OBJ_PER_WAVE - 1) / OBJ_PER_WAVE;
    if (ntiles >= ((int64_t)1 << 30)) return 0;
    return cl_layout((int)ntiles).total * 8;
}
// wavefronts of closed_loop_kernel<prop> the device holds at once (the runtime's occupancy figure x compute units)
static int64_t closed_loop_capacity(int prop)
{
    static int64_t cached[4] = {0, 0, 0, 0};
    if (cached[prop] <= 0) {
        int per_cu = 0;
        hipError_t e;
        if (prop == SSA_PROP_FG) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, closed_loop_kernel<1>, 64, 0);
        else if (prop == SSA_PROP_ELEMENTS) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, closed_loop_kernel<0>, 64, 0);
        else if (prop == SSA_PROP_HYBRID) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, closed_loop_kernel<3>, 64, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, closed_loop_kernel<2>, 64, 0);
        if (e != hipSuccess || per_cu <= 0) return 0;
        cached[prop] = (int64_t)per_cu * device_cu_count();
    }
    return cached[prop];
}
int ssa_env_closed_loop_f64(const ssa_consts* c, const ssa_step_params* p, const ssa_closed_loop_params* r, void* stream)
{
    if (!c || !p || !r || p->n_obj <= 0 || r->n_steps < 1 || r->history < 2) return SSA_E_INVALID;
    if (r->slot_out < 0 || r->slot_out >= r->history) return SSA_E_INVALID;
    if (!r->x_true_ring || !r->x_ring || !r->P_ring || !r->obs_ring || !r->metrics_ring || !r->stats_out || !r->actions || !r->workspace)
        return SSA_E_INVALID;
    if (!p->status || !p->trans || !p->env_time || !p->z_noise) return SSA_E_INVALID;
    if (r->agent < SSA_AGENT_NAIVE_GREEDY || r->agent > SSA_AGENT_VEL_ERROR) return SSA_E_INVALID;
    if (c->obs_type != SSA_OBS_AER && c->obs_type != SSA_OBS_XYZ) return SSA_E_INVALID;
    if (c->propagator != SSA_PROP_FG && c->propagator != SSA_PROP_ELEMENTS && c->propagator != SSA_PROP_J2_RK4 && c->propagator != SSA_PROP_HYBRID) return SSA_E_INVALID;
    if (c->propagator == SSA_PROP_J2_RK4 && (c->rk4_substeps < 1 || c->rk4_substeps > 4096)) return SSA_E_INVALID;
    if (p->n_env != 1) return SSA_E_UNSUPPORTED;
    if ((p->obj_ids != nullptr) != (r->slot_of != nullptr)) return SSA_E_INVALID;     // (a storage layout comes with its inverse table)
    const int64_t ntiles = (p->n_obj + OBJ_PER_WAVE - 1) / OBJ_PER_WAVE;
    const int64_t cap = closed_loop_capacity(c->propagator);
    const ClLayout L = cl_layout((int)ntiles);
    if (ntiles + L.ng + 1 > cap) return SSA_E_UNSUPPORTED;   // every wavefront (compute + service) must be resident: the decision is a grid-wide exchange
    if (r->workspace_bytes < L.total * 8) return SSA_E_INVALID;
    if (r->wait_ticks < 0 || (r->flags & ~(SSA_LOOP_ARGMAX_SPOS | SSA_LOOP_DEBUG_WITHHOLD))) return SSA_E_INVALID;
    LoopK lk;
    lk.k.c = *c;
    lk.k.p = *p;
    lk.k.p.aer_out = nullptr;
    lk.k.p.spos_tiles = nullptr;
    lk.k.p.spos_tiles_prev = nullptr;
    lk.c = *r;
    hipStream_t s = (hipStream_t)stream;
    // counters, flags: zero (a memset node: capturable)
    if (hipMemsetAsync(r->workspace, 0, (size_t)L.parts * 8, s) != hipSuccess) return SSA_E_LAUNCH;
    int nwork = (int)ntiles;
    const dim3 grid((unsigned)(nwork + L.ng + 1));
    // A COOPERATIVE launch: the decision is a grid-wide exchange, so every wavefront must be resident at once -- with a cooperative
    // launch that is the runtime's guarantee (it refuses a grid the device cannot hold next to what else is running), not only this
    // function's occupancy estimate above.  SSA_LOOP_PLAIN_LAUNCH=1 in the environment: the plain launch of round 3 (diagnostic).
    static const bool plain = []() { const char* e = getenv("SSA_LOOP_PLAIN_LAUNCH"); return e && e[0] == '1'; }();
    if (plain) {
        if (c->propagator == SSA_PROP_FG) hipLaunchKernelGGL(closed_loop_kernel<1>, grid, dim3(64), 0, s, lk, nwork, nwork);
        else if (c->propagator == SSA_PROP_ELEMENTS) hipLaunchKernelGGL(closed_loop_kernel<0>, grid, dim3(64), 0, s, lk, nwork, nwork);
        else if (c->propagator == SSA_PROP_HYBRID) hipLaunchKernelGGL(closed_loop_kernel<3>, grid, dim3(64), 0, s, lk, nwork, nwork);
        else hipLaunchKernelGGL(closed_loop_kernel<2>, grid, dim3(64), 0, s, lk, nwork, nwork);
        return launch_status();
    }
    void* args[3] = {(void*)&lk, (void*)&nwork, (void*)&nwork};
    const void* fn = (c->propagator == SSA_PROP_FG) ? (const void*)closed_loop_kernel<1>
                     : (c->propagator == SSA_PROP_ELEMENTS) ? (const void*)closed_loop_kernel<0>
                     : (c->propagator == SSA_PROP_HYBRID) ? (const void*)closed_loop_kernel<3> : (const void*)closed_loop_kernel<2>;
    const hipError_t ce = hipLaunchCooperativeKernel(fn, grid, dim3(64), args, 0, s);
    if (ce == hipErrorCooperativeLaunchTooLarge) {
        (void)hipGetLastError();
        return SSA_E_UNSUPPORTED;          // (the device cannot hold the grid right now: the caller takes the per-step launches)
    }
    if (ce != hipSuccess) {
        fprintf(stderr, "libssa_hip: cooperative launch of the closed loop failed: %s (%s)\n", hipGetErrorName(ce), hipGetErrorString(ce));
        (void)hipGetLastError();
        return SSA_E_LAUNCH;
    }
    return launch_status();
}
int ssa_stats_fold_f64(uint64_t* stat_shards, double* stats, int32_t n_env, void* stream)
{
    if (!stat_shards || !stats || n_env <= 0) return SSA_E_INVALID;
    hipLaunchKernelGGL(reward_fold_kernel, dim3(n_env), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)stat_shards, stats,
                       (const unsigned long long*)nullptr, (int64_t)0);
    return launch_status();
}
int ssa_stats_fold_spos_f64(uint64_t* stat_shards, const uint64_t* spos_tiles, double* stats, int64_t n_obj, int32_t n_env, void* stream)
{
    if (!stat_shards || !stats || n_env <= 0 || n_obj <= 0) return SSA_E_INVALID;
    if (spos_tiles && n_env > 1 && (n_obj % OBJ_PER_WAVE) != 0) return SSA_E_UNSUPPORTED;
    hipLaunchKernelGGL(reward_fold_kernel, dim3(n_env), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)stat_shards, stats,
                       (const unsigned long long*)spos_tiles, n_obj);
    return launch_status();
}
int64_t ssa_env_step_work_bytes(int64_t n_obj, int32_t n_env)
{
    (void)n_obj; (void)n_env;
    return 16;   // `work` is not used any more (the exception queue is gone); a token size keeps old callers valid
}

int ssa_reward_stats_f64(const double* metrics, const int32_t* status, double* stats, void* workspace, int64_t n_obj,
                         int32_t n_env, void* stream)
{
    if (!metrics || !status || !stats || !workspace || n_obj <= 0 || n_env <= 0) return SSA_E_INVALID;
    int64_t want = (n_obj + (int64_t)STAT_T * STAT_ILP - 1) / ((int64_t)STAT_T * STAT_ILP);
    int nparts = (int)(want < 1 ? 1 : (want > STAT_MAX_PARTS / 4 ? STAT_MAX_PARTS / 4 : want));
    StatAcc* parts = (StatAcc*)workspace;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(reward_partial_kernel, dim3(nparts, n_env), dim3(STAT_T), 0, s, metrics, status, parts, n_obj, nparts);
    hipLaunchKernelGGL(reward_final_kernel, dim3(n_env), dim3(64), 0, s, (const StatAcc*)parts, stats, nparts);
    return launch_status();
}
int64_t ssa_reward_stats_workspace_bytes(int32_t n_env)
{
    return (int64_t)(n_env < 1 ? 1 : n_env) * 256 * (int64_t)sizeof(StatAcc) + 64;
}

int ssa_propagate_f64(const double* x_in, double* x_out, int64_t n, double dt, int32_t propagator, void* stream)
{
    if (n == 0) return SSA_OK;
    if (!x_in || !x_out || n < 0) return SSA_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (propagator == SSA_PROP_FG) hipLaunchKernelGGL(propagate_kernel<1>, dim3(nblk(n, 64)), dim3(64), 0, s, x_in, x_out, n, dt);
    else if (propagator == SSA_PROP_ELEMENTS) hipLaunchKernelGGL(propagate_kernel<0>, dim3(nblk(n, 64)), dim3(64), 0, s, x_in, x_out, n, dt);
    else if (propagator == SSA_PROP_HYBRID) hipLaunchKernelGGL(propagate_hybrid_kernel, dim3(nblk(n, 64)), dim3(64), 0, s, x_in, x_out, n, dt);
    else return SSA_E_INVALID;
    return launch_status();
}

int ssa_propagate_j2_f64(const double* x_in, double* x_out, int64_t n, double dt, double j2, double r_eq, int32_t substeps,
                         void* stream)
{
    if (n == 0) return SSA_OK;
    if (!x_in || !x_out || n < 0 || substeps < 1 || substeps > 4096) return SSA_E_INVALID;
    J2Params q = {j2, r_eq, substeps};
    hipLaunchKernelGGL(propagate_j2_kernel, dim3(nblk(n, 64)), dim3(64), 0, (hipStream_t)stream, x_in, x_out, n, dt, q);
    return launch_status();
}

int ssa_kepler_elements_f64(const double* x_in, double* coe, int64_t n, double dt, void* stream)
{
    if (!x_in || !coe || n < 0) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(elements_kernel, dim3(nblk(n, 64)), dim3(64), 0, (hipStream_t)stream, x_in, coe, n, dt);
    return launch_status();
}

int ssa_robust_cholesky6_f64(const double* A, double* U, int32_t* rung, int64_t n, void* stream)
{
    if (!A || !U || !rung || n < 0) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(cholesky_kernel, dim3(nblk(n, 64)), dim3(64), 0, (hipStream_t)stream, A, U, rung, n);
    return launch_status();
}

int ssa_ladder_probe_f64(const double* A, double scale, int32_t* rung, int32_t* mask, double* U, int64_t n, void* stream)
{
    if (!A || !rung || !mask || n < 0) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(ladder_probe_kernel, dim3(nblk(n, OBJ_PER_WAVE)), dim3(64), 0, (hipStream_t)stream, A, scale, rung, mask, U, n);
    return launch_status();
}

int ssa_sigma_points_f64(const double* x, const double* P, double scale, double* sig, int32_t* fail, int64_t n, void* stream)
{
    if (!x || !P || !sig || !fail || n < 0) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(sigma_points_kernel, dim3(nblk(n, 64)), dim3(64), 0, (hipStream_t)stream, x, P, scale, sig, fail, n);
    return launch_status();
}

int ssa_hx_aer_f64(const double* x, int64_t x_stride, const double* M, const ssa_consts* c, double* z, int64_t n, void* stream)
{
    if (!x || !M || !c || !z || n < 0 || x_stride < 3) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(hx_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x_stride, M, make_geo(c), z, n);
    return launch_status();
}

int ssa_mean_z_uvw_f64(const double* sigmas, const ssa_consts* c, double* zp, int64_t n, void* stream)
{
    if (!sigmas || !c || !zp || n < 0) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(mean_z_kernel, dim3(nblk(n, 64)), dim3(64), 0, (hipStream_t)stream, sigmas, make_geo(c), zp, n);
    return launch_status();
}

int ssa_residual_z_aer_f64(const double* a, const double* b, double* c, int64_t n, void* stream)
{
    if (!a || !b || !c || n < 0) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(residual_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, n);
    return launch_status();
}

int ssa_visible_mask_f64(const double* x_true, const double* M, const ssa_consts* c, uint8_t* mask, double* el, int64_t n,
                         void* stream)
{
    if (!x_true || !M || !c || !mask || n < 0) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(visible_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, x_true, M, make_geo(c), mask, el, n,
                       (const int32_t*)nullptr, 0, 0);
    return launch_status();
}
int ssa_visible_mask_at_f64(const double* x_true, const double* trans, const int32_t* env_time, int32_t time_offset, int32_t n_time,
                            const ssa_consts* c, uint8_t* mask, double* el, int64_t n, void* stream)
{
    if (!x_true || !trans || !env_time || n_time <= 0 || !c || !mask || n < 0) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(visible_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, x_true, trans, make_geo(c), mask, el, n,
                       env_time, time_offset, n_time);
    return launch_status();
}

int ssa_observe_f64(const double* x_true, const double* x, const double* P, double* obs, double* metrics, int64_t n, void* stream)
{
    if (!x_true || !x || !P || !obs || !metrics || n < 0) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(observe_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, x_true, x, P, obs, metrics, n);
    return launch_status();
}

int ssa_aer_obs_f64(const double* x, const double* P, const double* M, const ssa_consts* c, double* out, int64_t n, void* stream)
{
    if (!x || !P || !M || !c || !out || n < 0) return SSA_E_INVALID;
    if (n == 0) return SSA_OK;
    hipLaunchKernelGGL(aer_obs_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, x, P, M, make_geo(c), out, n);
    return launch_status();
}

int ssa_agent_scores_f64(const double* x_true, const double* x_cur, const double* P_cur, const double* P_prev, const double* M,
                         const ssa_consts* c, double* scores, uint8_t* mask, int64_t n, void* stream)
{
    if (n == 0) return SSA_OK;
    if (!x_true || !x_cur || !P_cur || !scores || !c || n < 0 || (mask && !M)) return SSA_E_INVALID;
    hipLaunchKernelGGL(agent_scores_kernel, dim3(nblk(n, 64)), dim3(64), 0, (hipStream_t)stream, x_true, x_cur, P_cur, P_prev, M,
                       make_geo(c), scores, mask, n, (const int32_t*)nullptr, 0, 0);
    return launch_status();
}
int ssa_agent_scores_at_f64(const double* x_true, const double* x_cur, const double* P_cur, const double* P_prev, const double* trans,
                            const int32_t* env_time, int32_t time_offset, int32_t n_time, const ssa_consts* c, double* scores, uint8_t* mask,
                            int64_t n, void* stream)
{
    if (n == 0) return SSA_OK;
    if (!x_true || !x_cur || !P_cur || !scores || !c || n < 0 || !trans || !env_time || n_time <= 0 || !mask) return SSA_E_INVALID;
    hipLaunchKernelGGL(agent_scores_kernel, dim3(nblk(n, 64)), dim3(64), 0, (hipStream_t)stream, x_true, x_cur, P_cur, P_prev, trans,
                       make_geo(c), scores, mask, n, env_time, time_offset, n_time);
    return launch_status();
}

int ssa_nees_f64(const double* x_true, const double* x, const double* P, double* nees, int64_t n, void* stream)
{
    if (n == 0) return SSA_OK;
    if (!x_true || !x || !P || !nees || n < 0) return SSA_E_INVALID;
    hipLaunchKernelGGL(nees_kernel, dim3(nblk(n, 64)), dim3(64), 0, (hipStream_t)stream, x_true, x, P, nees, n);
    return launch_status();
}
int ssa_nis_f64(const double* y, const double* S, double* nis, int64_t n, void* stream)
{
    if (n == 0) return SSA_OK;
    if (!y || !S || !nis || n < 0) return SSA_E_INVALID;
    hipLaunchKernelGGL(nis_kernel, dim3(nblk(n, 64)), dim3(64), 0, (hipStream_t)stream, y, S, nis, n);
    return launch_status();
}

int ssa_chi2_contained_f64(const double* v, int64_t n, double lo, double hi, int64_t* counts, void* stream)
{
    if (!counts || n < 0 || (n > 0 && !v)) return SSA_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(chi2_zero_kernel, dim3(1), dim3(64), 0, s, (unsigned long long*)counts);
    if (n > 0) {
        int64_t nb = (n + 256 * 8 - 1) / (256 * 8);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(chi2_contained_kernel, dim3((unsigned)nb), dim3(256), 0, s, v, n, lo, hi, (unsigned long long*)counts);
    }
    return launch_status();
}

int64_t ssa_agent_select_workspace_bytes(int64_t n_obj, int32_t n_env)
{
    if (n_obj <= 0 || n_env <= 0) return 0;
    return (int64_t)n_env * ((n_obj + AGENT_T - 1) / AGENT_T) * (int64_t)sizeof(AgentPart);
}
int ssa_agent_select_f64(const ssa_consts* c, int32_t kind, const double* x_true, const double* x_cur, const double* P_cur,
                         const double* P_prev, const double* trans, const int32_t* env_time, int32_t time_offset, int32_t n_time,
                         const int32_t* fallback, void* workspace, int32_t* action_out, int64_t* pick_out, int64_t n_obj,
                         int32_t n_env, void* stream)
{
    return ssa_agent_select_ids_f64(c, kind, x_true, x_cur, P_cur, P_prev, trans, env_time, time_offset, n_time, fallback, workspace, action_out,
                                    pick_out, n_obj, n_env, nullptr, stream);
}
int ssa_agent_select_ids_f64(const ssa_consts* c, int32_t kind, const double* x_true, const double* x_cur, const double* P_cur,
                             const double* P_prev, const double* trans, const int32_t* env_time, int32_t time_offset, int32_t n_time,
                             const int32_t* fallback, void* workspace, int32_t* action_out, int64_t* pick_out, int64_t n_obj,
                             int32_t n_env, const int32_t* obj_ids, void* stream)
{
    if (!c || !x_true || !x_cur || !P_cur || !trans || !env_time || !workspace || !action_out || n_obj <= 0 || n_env <= 0)
        return SSA_E_INVALID;
    if (obj_ids && n_env != 1) return SSA_E_UNSUPPORTED;
    const int nparts = (int)((n_obj + AGENT_T - 1) / AGENT_T);
    const dim3 grid(nparts, n_env), block(AGENT_T);
    hipStream_t s = (hipStream_t)stream;
    AgentPart* parts = (AgentPart*)workspace;
    const GeoK g = make_geo(c);
#define SSA_AGENT_LAUNCH(K) hipLaunchKernelGGL(agent_partial_kernel<K>, grid, block, 0, s, x_true, x_cur, P_cur, P_prev, trans, env_time, \
                                               time_offset, n_time, g, parts, n_obj, obj_ids)
    switch (kind) {
        case SSA_AGENT_NAIVE_GREEDY: SSA_AGENT_LAUNCH(SSA_AGENT_NAIVE_GREEDY); break;
        case SSA_AGENT_VISIBLE_GREEDY: SSA_AGENT_LAUNCH(SSA_AGENT_VISIBLE_GREEDY); break;
        case SSA_AGENT_SHANNON: SSA_AGENT_LAUNCH(SSA_AGENT_SHANNON); break;
        case SSA_AGENT_POS_ERROR: SSA_AGENT_LAUNCH(SSA_AGENT_POS_ERROR); break;
        case SSA_AGENT_VEL_ERROR: SSA_AGENT_LAUNCH(SSA_AGENT_VEL_ERROR); break;
        default: return SSA_E_INVALID;
    }
#undef SSA_AGENT_LAUNCH
    hipLaunchKernelGGL(agent_final_kernel, dim3(n_env), dim3(64), 0, s, (const AgentPart*)parts, nparts, fallback, action_out, pick_out);
    return launch_status();
}

int ssa_masked_argmax_f64(const double* score, const uint8_t* mask, int64_t n, int64_t* out, void* stream)
{
    if (!out || n < 0 || (n > 0 && !score)) return SSA_E_INVALID;
    hipLaunchKernelGGL(masked_argmax_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, score, mask, n, out);
    return launch_status();
}

int64_t ssa_masked_argmax_workspace_bytes(int64_t n)
{
    if (n < 0) return SSA_E_INVALID;
    return 64 + 16 * ((n + AMAX_T * AMAX_Q - 1) / (AMAX_T * AMAX_Q) + 1);
}
int ssa_masked_argmax_ws_f64(const double* score, const uint8_t* mask, int64_t n, int64_t* out, void* workspace, int64_t workspace_bytes,
                             void* stream)
{
    if (!out || n < 0 || (n > 0 && !score)) return SSA_E_INVALID;
    const int64_t nb = (n + AMAX_T * AMAX_Q - 1) / (AMAX_T * AMAX_Q);
    if (nb <= 1 || !workspace)      // (one workgroup's worth, or no workspace: the single-workgroup kernel)
        return ssa_masked_argmax_f64(score, mask, n, out, stream);
    if (workspace_bytes < ssa_masked_argmax_workspace_bytes(n) || nb > 0x7fffffff) return SSA_E_INVALID;
    hipLaunchKernelGGL(masked_argmax_blocks_kernel, dim3((unsigned)nb), dim3(AMAX_T), 0, (hipStream_t)stream, score, mask, n, out,
                       (unsigned long long*)workspace);
    return launch_status();
}

// ---- all-gather by direct peer stores (include/ssa_hip.h): no collective, plain kernels
int ssa_peer_push_f64(const double* src, int64_t n_words, double* const* dst, uint64_t* const* flag, int32_t n_peer,
                      const uint64_t* seq_base, uint64_t seq_off, const uint64_t* prev_flags, int64_t timeout_ticks, int32_t* error,
                      uint32_t* tickets, void* stream)
{
    if (n_peer < 0 || n_words < 0 || !seq_base || (n_peer > 0 && (!dst || !flag || !tickets || (n_words > 0 && !src)))) return SSA_E_INVALID;
    if (n_peer == 0) return SSA_OK;
    hipLaunchKernelGGL(peer_push_kernel, dim3(n_peer, PEER_SPLIT), dim3(256), 0, (hipStream_t)stream, src, n_words, dst,
                       (unsigned long long* const*)flag, (const unsigned long long*)seq_base, (unsigned long long)seq_off,
                       (const unsigned long long*)prev_flags, (long long)(timeout_ticks > 0 ? timeout_ticks : 200000000ll), error, tickets);
    return launch_status();
}
int ssa_peer_wait(const uint64_t* flags, int32_t n_src, const uint64_t* seq_base, uint64_t seq_off, int64_t timeout_ticks, int32_t* error,
                  void* stream)
{
    if (n_src < 0 || !seq_base || (n_src > 0 && !flags)) return SSA_E_INVALID;
    if (n_src == 0) return SSA_OK;
    hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)flags, n_src,
                       (const unsigned long long*)seq_base, (unsigned long long)seq_off,
                       (long long)(timeout_ticks > 0 ? timeout_ticks : 200000000ll), error);
    return launch_status();
}

}  // extern "C"
