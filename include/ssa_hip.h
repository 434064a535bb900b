/*
 * ssa_hip.h -- C ABI of libssa_hip.so: the MI355X (gfx950) implementation of
 * ssa-gym's per-step hot path (propagate -> UKF predict -> one UKF update ->
 * observation / error metrics / reward statistics).
 *
 * The reference (AshHarvey/ssa-gym) is pure Python and has no FFI of its own;
 * the boundary it exposes is the gym.Env class plus the operator callables in
 * its env_config dict (envs/__init__.py:23-28).  This library sits directly
 * below that class: every entry point replaces the per-object Python loop or
 * numba/filterpy/LAPACK call cited next to it (file:line under the reference
 * root).  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (hipMalloc / torch.Tensor.data_ptr())
 *     unless the parameter name ends in _host;
 *   - arrays are float64, C-contiguous, array-of-structures exactly as the
 *     reference's numpy arrays: states [n][6] (m, m/s, GCRS), covariances
 *     [n][6][6], observations [n][12], measurements [n][3];
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     launches are asynchronous, nothing synchronises;
 *   - every function returns 0 on success or a negative SSA_E_* code; no
 *     exceptions, no allocation, no host<->device copies -> all entry points
 *     are hipGraph-capturable.
 */
#ifndef SSA_HIP_H
#define SSA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSA_ABI_VERSION 22
#define SSA_INLINE_ENVS 8

/* error codes */
#define SSA_OK 0
#define SSA_E_INVALID (-1) /* bad size / null pointer / unknown flag */
#define SSA_E_LAUNCH (-2)  /* hipGetLastError() != hipSuccess after launch */
#define SSA_E_UNSUPPORTED (-3) /* valid arguments, but this entry point does not cover the configuration (the caller takes the
                                  general one: e.g. ssa_env_closed_loop_f64 with more objects than wavefronts resident at once) */

/* per-object filter status (int32), persists across steps.
 * ssa_tasker_simple_2.py:271-285, 300-313, 369-382 (filter_error): a failed filter is
 * overwritten with the sentinels x_failed / P_failed (:157-158) and skipped afterwards. */
#define SSA_ST_OK 0
#define SSA_ST_PREDICT_NAN 1    /* ', predict returned nan. ' */
#define SSA_ST_PREDICT_LINALG 2 /* robust_cholesky ladder exhausted -> LinAlgError */
#define SSA_ST_UPDATE_NAN 3     /* ', update returned nan. ' */
#define SSA_ST_UPDATE_LINALG 4  /* inv(S) singular -> LinAlgError */

/* observation model of the update (env_config['obs_type'], ssa_tasker_simple_2.py:98-107) */
#define SSA_OBS_AER 0 /* hx_aer_erfa + mean_z_uvw + residual_z_aer (dynamics.py:219,343,260) */
#define SSA_OBS_XYZ 1 /* hx_xyz + mean_xyz + residual_xyz/np.subtract (dynamics.py:207,276,271) */

/* two-body propagator variant; both evaluate envs/farnocchia.py:1010 farnocchia() */
#define SSA_PROP_ELEMENTS 0 /* rv2coe -> delta_t_from_nu -> nu_from_delta_t -> coe2rv, operation by operation; with
                               SSA_FLAG_REFERENCE_COV the BEHAVIOUR-FAITHFUL variant (the reference's episode-level filter failures) */
#define SSA_PROP_FG 1       /* every conic branch of farnocchia() through ONE equation: Kepler's equation in universal variables
                               (Stumpff series + Halley steps; closed-form Stumpff functions + Laguerre-Conway iterations for
                               long steps), state by the Lagrange coefficients f, g.  More accurate than the reference's chain on
                               diverged (hyperbolic) states, which is why its filters do not fail where the reference's do */
#define SSA_PROP_J2_RK4 2   /* EXTENSION without reference counterpart: two-body + J2 zonal
                               acceleration, classical RK4 with ssa_consts.rk4_substeps sub-steps per dt */

#define SSA_PROP_HYBRID 3   /* the behaviour-faithful variant at speed: the universal-variable series solver of SSA_PROP_FG on
                               strong-elliptic states (ecc < 1 - 1e-2, farnocchia.py:871: there the reference's chain agrees with it
                               to 1e-14); the reference's own STRONG-HYPERBOLIC chain (ecc > 1 + 1e-2, farnocchia.py:909-912,
                               1001-1004: F from the elements, F -> nu -> F, the hyperbolic Newton solve, F -> nu) -- operation by
                               operation, NaN by NaN -- on the states where its filters go wrong: a diverged filter's sigma points, which
                               that chain propagates metres to kilometres off; and (round 4) the universal-variable solver again on
                               the bands in between (|ecc - 1| <= 1e-2, elliptic orbits beyond the series' range), where the reference
                               is accurate to its Newton tolerance and the two agree to ~1e-12.  rv2coe's special orientations
                               (circular, equatorial) take the complete restatement.  With SSA_FLAG_REFERENCE_COV the same
                               episode-level failure statistics as SSA_PROP_ELEMENTS and the oracle (tests/test_episode_failures.py:
                               pooled counts, failed-set overlap, first-failure-step distributions over five workloads) */

/* flags of ssa_step_params.flags */
#define SSA_FLAG_RESAMPLE 1u /* predict() ends by redrawing the sigma points from the prior, for every filter (the
                                predict() of filterpy's development branch); default off = the propagated points are
                                kept for update() (believed to be what the released filterpy 1.4.5 does).  Which
                                of the two the reference's unpinned `filterpy` requirement resolved to cannot be checked
                                offline (the package is absent): the default is PARITY-UNPINNED, both are tested. */

#define SSA_FLAG_REFERENCE_COV 2u /* the prior covariance in the reference's own arithmetic: P = sum_i (sigma_i' - x)(Wc_i (sigma_i'
                                - x))^T + Q over all 13 points (filterpy's unscented_transform, call ssa_tasker_simple_2.py:275), whose
                                Wc_0 ~ -2e8 term cancels the other twelve.  For filters that have diverged late in a predict-mostly
                                episode that cancellation makes P indefinite and the NEXT predict fails with LinAlgError -- the
                                reference loses 2-3 % of its filters per 480-step episode this way (:271-285, 369-382).  Default
                                off: the same matrix expanded around sigma_0' (no cancellation, those filters survive).  On = the
                                BEHAVIOUR-FAITHFUL form; the env selects it together with SSA_PROP_ELEMENTS. */

/* layout of the per-env update record written by ssa_env_step_f64 (doubles) */
#define SSA_UPD_STRIDE 64
#define SSA_UPD_OBS_TAKEN 0 /* 1.0 if filters[a].update() ran (obs_taken[i], :305) */
#define SSA_UPD_Z_TRUE 1    /* [3] hx(x_true[i][a])                      (:298) */
#define SSA_UPD_Y 4         /* [3] innovation                            (:302) */
#define SSA_UPD_S 7         /* [3][3] innovation covariance              (:303) */
#define SSA_UPD_SIGMAS_H 16 /* [13][3] measurement sigma points          (:304) */
#define SSA_UPD_VISIBLE 55  /* 1.0 if object_visible([a])                (:299) */
#define SSA_UPD_ACTION 56   /* the action this record belongs to (as double), -1 = no update attempted */

/* layout of the per-env reward statistics written by ssa_reward_stats_f64 (doubles) */
#define SSA_STAT_SHARDS 128 /* accumulator shards per env of the atomics-based statistics path */
#define SSA_STAT_SHARD_WORDS 16 /* uint64 words between shards: every shard owns a 128-byte line (words 0..2 used); with 64 shards
                                  packed four words apart, 5000 wavefronts x 2 atomics met on 16 lines: +2 us per 20 000-object step */
#define SSA_STAT_STRIDE 8
#define SSA_STAT_MAX_DPOS 0   /* np.max(delta_pos[i])  (NaN-propagating)        (:325-343) */
#define SSA_STAT_CNT_LT_1E4 1 /* count(delta_pos < 1e4)  } results.py:432        */
#define SSA_STAT_CNT_LT_1E7 2 /* count(delta_pos < 1e7)  } reward_proportional_trinary_true */
#define SSA_STAT_ARGMAX_SPOS 3 /* np.argmax(sigma_pos[i]) (first max; 'shaped' uses it next step, :346) */
#define SSA_STAT_N_FAILED 4   /* number of objects with status != 0 */
#define SSA_STAT_MAX_SPOS 5   /* np.max(sigma_pos[i]) */

/* Constants of one environment family; built on the host, passed by value. */
typedef struct ssa_consts {
    double Q[36];       /* process noise, Q_discrete_white_noise (ssa_tasker_simple_2.py:110) */
    double R[9];        /* measurement noise (:128-131) */
    double Wm0, Wc0, Wi; /* Merwe weights: Wm[0], Wc[0], Wm[i]=Wc[i] (i>=1) (filterpy, :211-214) */
    double sum_wm_m1;   /* sum(Wm) - 1 evaluated exactly from the double weights */
    double sum_wc;      /* sum(Wc) evaluated exactly from the double weights */
    double scale;       /* n + lambda */
    double dt;          /* time_step [s] */
    double obs_limit;   /* elevation mask [rad] (:86) */
    double enu[9];      /* ecef2aer's trans_uvw_ecef matrix for the observer (transformations.py:341-343) */
    double obs_itrs[3]; /* lla2ecef(observer) (transformations.py:217) */
    int32_t obs_type;   /* SSA_OBS_* */
    int32_t propagator; /* SSA_PROP_* */
    uint32_t flags;     /* SSA_FLAG_* */
    int32_t update_interval; /* env_config['update_interval'] (:292) */
    double j2, r_eq;    /* SSA_PROP_J2_RK4: zonal coefficient and equatorial radius [m] (j2 = 0 -> two-body) */
    int32_t rk4_substeps; /* SSA_PROP_J2_RK4: RK4 steps per dt (>= 1) */
    int32_t reserved0;
} ssa_consts;

/* One env step for n_env independent environments of n_obj objects each
 * (object g = e * n_obj + j).  Replaces SSA_Tasker_Env.step() lines 265-322. */
typedef struct ssa_step_params {
    int64_t n_obj; /* m */
    int32_t n_env; /* E (1 for the plain gym env) */
    int32_t time_offset; /* added to env_time[e] (lets a captured / looped launch advance time) */
    const double *x_true_in; double *x_true_out; /* [E*m][6]  x_true[i-1] -> x_true[i]   (:265-266) */
    const double *x_in;      double *x_out;      /* [E*m][6]  filters[j].x               (:275,286) */
    const double *P_in;      double *P_out;      /* [E*m][6][6] filters[j].P             (:275,287) */
    int32_t *status;                             /* [E*m] in/out, SSA_ST_*               (:272,293) */
    double *obs;     /* [E*m][12]  observations(x, P)                     (results.py:61)  */
    double *metrics; /* [E][4][m]  delta_pos | delta_vel | sigma_pos | sigma_vel (results.py:37) */
    double *upd;     /* [E][SSA_UPD_STRIDE] update record, may be NULL */
    const double *trans;       /* [n_time][3][3] GCRS->ITRS matrices (trans_matrix, :137) */
    const int32_t *env_time;   /* [E] time index i of this step per env (after the increment of :259) */
    const int32_t *actions;    /* [E] object chosen per env, < 0 = no update */
    const double *z_noise;     /* measurement noise [3] of (env e, time i, object a) at
                                  z_noise + e*zn_stride_env + i*zn_stride_time + a*zn_stride_obj   (:219-221);
                                  the reference layout z_noise[n][m][3] has strides (0, 3m, 3) */
    int64_t zn_stride_env, zn_stride_time, zn_stride_obj;
    int32_t n_time;            /* rows in `trans` / time rows of `z_noise` (index = i % n_time) */
    uint32_t launch_mask;      /* 0 = everything; diagnostic: 1 common-path kernel, 2 post kernel, 4 final;
                                  SSA_LAUNCH_DEFER_FOLD (8): see stat_shards_prev */
    double *stats;             /* [E][SSA_STAT_STRIDE] reward statistics of this step (O3), may be NULL */
    int32_t *work;             /* unused since ABI 12 (the exception queue is gone); may be NULL */
    void *stat_ws;             /* ssa_reward_stats_workspace_bytes(): per-block statistics partials */
    uint64_t *stat_shards;     /* [E][SSA_STAT_SHARDS][SSA_STAT_SHARD_WORDS] zero-initialised device words, or NULL.  When given the
                                  step kernel accumulates max delta_pos / trinary counts / failures itself with sharded
                                  atomics (and writes aer_out, if asked, in its epilogue); a one-wave fold kernel writes
                                  `stats` and clears the words: two launches per step instead of three.  arg-max
                                  sigma_pos (only the 'shaped' reward needs it) is then computed only when spos_tiles is
                                  given; without: stats[SSA_STAT_ARGMAX_SPOS] = -1, stats[SSA_STAT_MAX_SPOS] = NaN. */
    uint64_t *stat_shards_prev;/* deferred fold (with SSA_LAUNCH_DEFER_FOLD in launch_mask): the shard set the PREVIOUS step
                                  accumulated into, or NULL.  The step kernel then carries n_env extra wavefronts that fold
                                  it into stats_prev and clear it while the objects of THIS step are being advanced, and
                                  this step's own statistics stay in stat_shards (no fold launch: ONE launch per step) until
                                  the next step passes them here or ssa_stats_fold_f64() folds them.  Alternate two sets. */
    double *stats_prev;        /* [E][SSA_STAT_STRIDE] destination of that fold */
    double *aer_out;           /* [E*m][4] aer_obs() of the NEW state (O4: az, el, range, trace P; NaN/inf -> 0.001;
                                  ssa_tasker_simple_2.py:834-840), may be NULL.  With stat_shards it is written by the step
                                  kernel's epilogue from the on-chip tiles (no extra launch, no second pass over x / P);
                                  without, by the post kernel: the 'aer' observation / the sharded all-gather payload. */
    uint64_t *stat_shards_clear; /* [E][SSA_STAT_SHARDS][SSA_STAT_SHARD_WORDS] or NULL: a shard set this launch ZEROES (nothing reads or adds to it
                                  during the launch).  Lets a consumer that takes the statistics as RAW shards -- the sharded
                                  multi-GPU step sends its rank's shard words in the all-gather payload and every rank
                                  folds all ranks' words itself -- rotate payload buffers without a fold / clear
                                  launch: step k accumulates into buffer k % N and clears buffer (k + 1) % N (N = 2 when the
                                  all-gather runs in the step's stream, 3 when it overlaps the next step on another one). */
    int32_t aer_cols;          /* columns of aer_out per object: 0 or 4 = (az, el, range, trace P); 1 = trace P only, [E*m] --
                                  the "per-object covariance-trace observation" of the sharded 160 000-object configuration:
                                  a quarter of the all-gather payload and no inverse trigonometry in the epilogue */
    int32_t action0;           /* with SSA_LAUNCH_INLINE_ACTION in launch_mask (one env): the env's action BY VALUE -- `actions` is then not
                                  read and may be NULL.  For callers whose action is born on the host every step (a gym env): a word
                                  in host-mapped memory would be fetched over PCIe by every wavefront of the launch (5 000 reads at
                                  20 000 objects: the step took 28 us instead of 16), a host-to-device copy costs a copy-engine pass */
    double *obs_mirror;        /* [E*m][12] or NULL: a SECOND destination of the observation rows (O1), written by the same lanes as
                                  `obs`.  Meant for host-mapped pinned memory: the 'flatten' observation of a gym-style caller then
                                  reaches the host from inside the kernel, overlapped with the other wavefronts' arithmetic, instead
                                  of through a copy-engine pass after it (1.92 MB at 20 000 objects: 44 us) */
    int32_t inline_time[SSA_INLINE_ENVS];   /* with SSA_LAUNCH_INLINE_ENVS (n_env <= SSA_INLINE_ENVS): env e's time index and action BY VALUE; */
    int32_t inline_action[SSA_INLINE_ENVS]; /* `env_time` / `actions` are then not read (env_time must still be non-NULL).  A vector env whose
                                  actions are born on the host every step saves the host-to-device copy in front of the launch */
    uint64_t *spos_tiles;      /* [ceil(E*m / 4)][2] device words or NULL (with stat_shards): arg-max of sigma_pos ON THE ONE-LAUNCH PATHS -- what the
                                  'shaped' reward reads, np.argmax(sigma_pos[i - 1]) (ssa_tasker_simple_2.py:339-352).  Every wavefront of the step
                                  kernel leaves (ordered bits of its tile's largest sigma_pos, index in the env of the first object that holds it) in
                                  its tile's slot -- no atomics -- and whoever folds the statistics shards (the fold kernel, the deferred fold's
                                  wavefront in the next launch, the last wavefront with SSA_LAUNCH_FOLD_INSIDE) reduces the slots with np.argmax's
                                  semantics (first maximum; the first NaN wins): stats[SSA_STAT_ARGMAX_SPOS] and stats[SSA_STAT_MAX_SPOS] are then
                                  exact on these paths too.  Needs whole tiles per env: n_env == 1 or n_obj % 4 == 0 (SSA_E_UNSUPPORTED otherwise:
                                  take the three-launch path, stat_shards = NULL).  Contents need no initialisation. */
    uint64_t *spos_tiles_prev; /* with SSA_LAUNCH_DEFER_FOLD: the slots the PREVIOUS step wrote (folded with stat_shards_prev), or NULL */
    double *fail_log;          /* [fail_cap][SSA_FAIL_STRIDE] or NULL: the filter_error() bookkeeping (ssa_tasker_simple_2.py:369-382) written BY THE KERNEL.
                                  A filter that fails in this step appends one record -- env, object, SSA_ST_* code, the step's time index and
                                  error_failed() of the state it failed from (:376-378: |x - x_true| and sqrt(sum diag P) of position and velocity, taken
                                  from the step's inputs) -- at index atomicAdd(fail_count, 1).  Meant for host-mapped pinned memory: the host reads the new
                                  records (their number = the rise of stats[SSA_STAT_N_FAILED]) right after the step's synchronisation, with no
                                  further copy -- the reference loses 2-3 % of its filters per episode, a few per step late in an episode, and a
                                  status read-back plus gathers per step doubled the cost of a gym-style step (round 4) */
    uint32_t *fail_count;      /* device word: records appended so far (the caller zeroes it when it resets its episode); NULL with fail_log NULL */
    int32_t fail_cap;          /* capacity of fail_log in records (a record beyond it is counted but not written) */
    int32_t reserved1;
    const int32_t *obj_ids;    /* [4 ceil(m / 4)] device words or NULL (one env: ssa_env_step_f64 with stat_shards, ssa_env_rollout_f64,
                                  ssa_env_closed_loop_f64 together with ssa_closed_loop_params.slot_of; SEVERAL envs, ssa_env_step_f64 only:
                                  [n_env][m] words, m % 4 == 0, every env its own permutation of 0 .. m - 1 -- indices within the env, as the
                                  actions are; obs_mirror / aer_out row e m + obj_ids[e][i]).  A LAYOUT: the caller stores its objects in another order than it numbers them --
                                  position i of every array of this struct holds the object the caller calls obj_ids[i] (round 4: objects of one orbit
                                  regime share wavefronts; late in an episode the diverged filters are the LEO objects, and packed they cost the launch
                                  10 % less: DESIGN.md section 6).  The kernel then speaks the CALLER's indices wherever an index leaves it or enters it:
                                  `actions` / action0 select obj_ids[i] == action, z_noise is indexed by the action as before, fail_log records and
                                  stats[SSA_STAT_ARGMAX_SPOS] carry obj_ids values (np.argmax's first maximum = the lowest such index), and the
                                  HOST-FACING copies of the observation -- obs_mirror rows, aer_out rows -- are written at row obj_ids[i].  x / P / x_true /
                                  status / obs / metrics stay in storage order.  An object's arithmetic does not depend on its position. */
} ssa_step_params;
#define SSA_FAIL_STRIDE 8
#define SSA_FAIL_ENV 0
#define SSA_FAIL_OBJ 1      /* index within the env */
#define SSA_FAIL_STATUS 2
#define SSA_FAIL_TIME 3     /* the step's time index i (env_time + time_offset) */
#define SSA_FAIL_ERR 4      /* [4] delta_pos, delta_vel, sigma_pos, sigma_vel of the state the filter failed from */

/* ---------------------------------------------------------------- fused hot path
 * One env step for every object of every env, every propagator:
 * the step kernel advances 4 objects per wavefront with complete semantics (robust_cholesky's jitter ladder inline,
 * conic branches beyond the strong-elliptic one as out-of-line calls), the update and -- with stat_shards -- the
 * reward statistics included.  Launches per step:
 *   stat_shards given : step kernel (aer_out, if asked, in its epilogue) + a one-wave fold; nothing more with
 *       SSA_LAUNCH_DEFER_FOLD (the next step's launch folds): 2 / 1;
 *   stat_shards NULL  : step kernel + post kernel (exact statistics per block, arg-max sigma_pos included, and the
 *       payload) + a one-wave fold of those: 3 launches. */
int ssa_env_step_f64(const ssa_consts *c_host, const ssa_step_params *p_host, void *stream);
/* Same step, with the dominant launch (the common-path kernel) bracketed by the event pair `slot`
 * (0 <= slot < SSA_PROFILE_SLOTS) bound to that dispatch: ssa_env_step_profile_ms() then returns the kernel's
 * duration from its own begin/end timestamps -- what rocprofv3 --kernel-trace reports -- without the queue having
 * been drained between launches.  Measurement aid of bench.py's roofline line (an event pair recorded around the
 * call would add the queue latency of the records, ~10 us). */
#define SSA_PROFILE_SLOTS 1024
#define SSA_LAUNCH_DEFER_FOLD 8u
#define SSA_LAUNCH_INLINE_ACTION 16u /* the env's action is ssa_step_params.action0 (n_env == 1) */
#define SSA_LAUNCH_FOLD_INSIDE 32u   /* with stat_shards + stats, no deferral: the statistics of env e are folded by the LAST wavefront of the
                                        step kernel that adds to them (it counts the tiles behind the shards' sums, words 3 / 4 of the env's
                                        shard lines) instead of by a fold kernel launched behind the step: ONE launch per step with the
                                        statistics available when it completes -- for callers that read them on the host after every step.
                                        Launches with more than one tile per wavefront (> 20 480 objects in all) get the fold kernel behind
                                        the step instead: counting a tile costs such a wavefront a memory round trip per tile */
#define SSA_LAUNCH_MIRROR_F32 128u    /* EXTENSION (round 4; the reference's observations are float64): the HOST-FACING copies of the observation -- obs_mirror
                                         rows and, on the one-launch statistics path, aer_out rows -- are written in SINGLE precision (obs_mirror / aer_out
                                         then point at float arrays of the same shape): half the bytes over PCIe for consumers that cast to float32
                                         anyway (every RL framework does).  `obs` (the device-resident rows) stays double. */
#define SSA_LAUNCH_INLINE_ENVS 64u   /* time indices and actions of all envs are ssa_step_params.inline_time / inline_action (n_env <= 8);
                                        time_offset is still added */
int ssa_env_step_profiled_f64(const ssa_consts *c_host, const ssa_step_params *p_host, void *stream, int32_t slot);
/* waits for slot's kernel and writes its duration in milliseconds */
int ssa_env_step_profile_ms(int32_t slot, float *kernel_ms);
/* ---------------------------------------------------------------- rollout: K steps in one launch
 * For open-loop action schedules (the reference's round-robin / random agents, agents.py, and filter-only runs): the
 * K calls ssa_env_step_f64 would make, with every step's outputs written to that step's slot of the history rings and
 * results bit-identical to them, but each wavefront keeps its objects' state in LDS across the steps (an object's
 * trajectory depends on no other object).  `first` is the parameter block of the FIRST step (n_obj, n_env, time_offset,
 * status, trans, env_time, z_noise and strides, n_time; its in/out/stats pointers are ignored).  Two launches: the rollout kernel and a fold of the per-step statistics. */
typedef struct ssa_rollout_params {
    int32_t n_steps;           /* K >= 1 */
    int32_t history;           /* H >= 2: depth of the rings below */
    int32_t slot_out;          /* step k (0-based) reads slot (slot_out + k - 1) mod H and writes slot (slot_out + k) mod H */
    int32_t reserved;
    double *x_true_ring;       /* [H][E*m][6]   slot 0 of each ring */
    double *x_ring;            /* [H][E*m][6] */
    double *P_ring;            /* [H][E*m][6][6] */
    double *obs_ring;          /* [H][E*m][12] */
    double *metrics_ring;      /* [H][E][4][m] */
    double *upd_ring;          /* [H][E][SSA_UPD_STRIDE] or NULL */
    double *stats_ring;        /* [H][E][SSA_STAT_STRIDE]; only the last H steps' statistics survive, as in any ring */
    const int32_t *actions;    /* [K][E] */
    uint64_t *stat_shards;     /* [K][E][SSA_STAT_SHARDS][SSA_STAT_SHARD_WORDS] zero-initialised; cleared again by the fold */
    uint64_t *spos_tiles;      /* [K][ceil(E*m / 4)][2] or NULL: per-step arg-max slots (ssa_step_params.spos_tiles): the statistics of every
                                  step then carry np.argmax(sigma_pos) -- the 'shaped' reward over a rollout */
} ssa_rollout_params;
int ssa_env_rollout_f64(const ssa_consts *c_host, const ssa_step_params *first, const ssa_rollout_params *r, void *stream);

/* folds a shard set into stats[n_env][SSA_STAT_STRIDE] and clears it: the last step of a deferred-fold sequence,
 * or whenever the host wants the statistics of the step just launched */
int ssa_stats_fold_f64(uint64_t *stat_shards, double *stats, int32_t n_env, void *stream);
/* the same with the arg-max slots of that step (ssa_step_params.spos_tiles; NULL = as ssa_stats_fold_f64) */
int ssa_stats_fold_spos_f64(uint64_t *stat_shards, const uint64_t *spos_tiles, double *stats, int64_t n_obj, int32_t n_env, void *stream);
/* historical: size of the `work` buffer (now unused); returns a token size */
int64_t ssa_env_step_work_bytes(int64_t n_obj, int32_t n_env);

/* O3: per-env reductions over metrics[E][4][m] and status -> stats[E][SSA_STAT_STRIDE]
 * (ssa_tasker_simple_2.py:324-354, results.py:432). */
int ssa_reward_stats_f64(const double *metrics, const int32_t *status, double *stats, void *workspace,
                         int64_t n_obj, int32_t n_env, void *stream);
/* bytes of device workspace ssa_reward_stats_f64 needs for n_env environments */
int64_t ssa_reward_stats_workspace_bytes(int32_t n_env);

/* ----------------------------------------------------- single operators (rows of SURVEY 8a) */
/* P1-P5  fx_xyz_farnocchia(x, dt) for n states (farnocchia.py:1054). */
int ssa_propagate_f64(const double *x_in, double *x_out, int64_t n, double dt, int32_t propagator, void *stream);
/* the J2 + RK4 extension propagator on its own (SSA_PROP_J2_RK4) */
int ssa_propagate_j2_f64(const double *x_in, double *x_out, int64_t n, double dt, double j2, double r_eq,
                         int32_t substeps, void *stream);
/* P2-P4 diagnostics: coe[n][8] = p, ecc, inc, raan, argp, nu0, delta_t0, nu(dt)  (farnocchia.py:165,847,925). */
int ssa_kepler_elements_f64(const double *x_in, double *coe, int64_t n, double dt, void *stream);
/* U2  robust_cholesky(A) for n 6x6 matrices: U upper (zeros below), rung[n] = -1 (no jitter),
 * 0..15 (10^(rung-6) added), 16 = LinAlgError (dynamics.py:402). */
int ssa_robust_cholesky6_f64(const double *A, double *U, int32_t *rung, int64_t n, void *stream);
/* U2 as the FUSED step kernels run it (the row-distributed ladder of robust_chol_row_lds, four matrices per wavefront): rung[n] as above
 * from the kernel's one-pass ladder, mask[n] = bit i set when rung i (jitter 10^(i-6)) factorises in the kernel's arithmetic (bit 16: the
 * plain attempt) -- all sixteen rungs are actually tried, so rung[n] must be the lowest set bit of mask[n] whatever the pattern.  (On the
 * numerically rank-one (n + lambda) P of a diverged filter that round 3 took for a non-monotone case -- tests/golden/
 * ladder_illconditioned_tile.npz -- the mask is 0111...1, and so it is on 4 096 one-ulp perturbations: DESIGN.md section 6, Round 4.) */
int ssa_ladder_probe_f64(const double *A, double scale, int32_t *rung, int32_t *mask, double *U, int64_t n, void *stream);
/* U1  MerweScaledSigmaPoints.sigma_points(x, P): sig[n][13][6], fail[n] (0 / SSA_ST_PREDICT_LINALG). */
int ssa_sigma_points_f64(const double *x, const double *P, double scale, double *sig, int32_t *fail,
                         int64_t n, void *stream);
/* H1  hx_aer_erfa for n states sharing one matrix M[9] (device pointer) (dynamics.py:219). */
int ssa_hx_aer_f64(const double *x, int64_t x_stride, const double *M, const ssa_consts *c_host, double *z,
                   int64_t n, void *stream);
/* H3  mean_z_uvw(sigmas[n][13][3], Wm) with the Merwe weights of c_host (dynamics.py:343). */
int ssa_mean_z_uvw_f64(const double *sigmas, const ssa_consts *c_host, double *zp, int64_t n, void *stream);
/* H4  residual_z_aer(a[n][3], b[n][3]) (dynamics.py:260). */
int ssa_residual_z_aer_f64(const double *a, const double *b, double *c, int64_t n, void *stream);
/* V1  object_visibility(): mask[n] = elevation(x_true) >= obs_limit; el[n] optional (:418-434). */
int ssa_visible_mask_f64(const double *x_true, const double *M, const ssa_consts *c_host, uint8_t *mask,
                         double *el, int64_t n, void *stream);
/* O1/O2  observations() + error() for n objects of one env (results.py:61,37); metrics[4][n]. */
int ssa_observe_f64(const double *x_true, const double *x, const double *P, double *obs, double *metrics,
                    int64_t n, void *stream);
/* O4  aer_obs(): out[n][4] = hx(x_filter), trace(P); NaN/inf -> 0.001 (ssa_tasker_simple_2.py:834). */
int ssa_aer_obs_f64(const double *x, const double *P, const double *M, const ssa_consts *c_host, double *out,
                    int64_t n, void *stream);

/* ------------------------------------------------ device-side agent primitives
 * Per-object scores the reference's heuristic agents compute in Python loops (agents.py:7-81):
 *   scores[0][j] = trace(P_cur[j])                        agent_naive_greedy / agent_visible_greedy
 *   scores[1][j] = log(det P_cur[j] / det P_prev[j])      agent_shannon        (NaN if either det <= 0)
 *   scores[2][j] = |x_cur[j][:3] - x_true[j][:3]|         agent_pos_error_greedy
 *   scores[3][j] = |x_cur[j][3:] - x_true[j][3:]|         agent_vel_error_greedy
 *   mask[j]      = elevation(x_true[j]) >= obs_limit      visible_objects()    (:410-425)
 * P_prev may be NULL (scores[1] = NaN). */
int ssa_agent_scores_f64(const double *x_true, const double *x_cur, const double *P_cur, const double *P_prev,
                         const double *M, const ssa_consts *c_host, double *scores, uint8_t *mask, int64_t n,
                         void *stream);
/* The two primitives above for callers whose time index lives ON THE DEVICE (a policy evaluated inside a captured hipGraph: the graph
 * advances the index between replays): the GCRS->ITRS matrix is row (env_time[0] + time_offset) % n_time of the table `trans`, as in the step. */
int ssa_visible_mask_at_f64(const double *x_true, const double *trans, const int32_t *env_time, int32_t time_offset, int32_t n_time,
                            const ssa_consts *c_host, uint8_t *mask, double *el, int64_t n, void *stream);
int ssa_agent_scores_at_f64(const double *x_true, const double *x_cur, const double *P_cur, const double *P_prev, const double *trans,
                            const int32_t *env_time, int32_t time_offset, int32_t n_time, const ssa_consts *c_host, double *scores,
                            uint8_t *mask, int64_t n, void *stream);
/* np.argmax(score[mask]) mapped back to object indices: out[0] = index of the first maximum of `score`
 * over entries with mask != 0 (mask may be NULL = all), or -1 when no entry is selected / n == 0;
 * NaN entries are skipped (the reference's agents run under np.errstate and np.argmax would return a
 * NaN's index; skipping is the documented deviation).  out[1] = the maximum (as double bits). */
int ssa_masked_argmax_f64(const double *score, const uint8_t *mask, int64_t n, int64_t *out, void *stream);
/* The same fold spread over the chip, for callers that own a workspace -- the arg-max head of a device-side policy between two step
 * launches (SSA_Tasker_Env.run_policy, PolicyView.argmax): one workgroup reads 20 000 entries in 8.6 us (a single CU's share of the memory
 * system), ceil(n / 2048) workgroups and a last-arrival fold take about 3.  `workspace`: ssa_masked_argmax_workspace_bytes(n) bytes of
 * device memory, ZERO before the first call (the kernel leaves it zero-ticketed for the next), used by one call at a time (calls in one
 * stream are).  NULL workspace or n <= 2048: the single-workgroup kernel. */
int ssa_masked_argmax_ws_f64(const double *score, const uint8_t *mask, int64_t n, int64_t *out, void *workspace, int64_t workspace_bytes,
                             void *stream);
int64_t ssa_masked_argmax_workspace_bytes(int64_t n);

/* ------------------------------------------------ closed loop on the device
 * The agent's choice for the NEXT step, made on the GPU from the state the step just wrote and stored into the int32
 * action word(s) ssa_step_params.actions of that next step points at -- enqueue  step, select, step, select, ...  in one
 * stream and a greedy agent runs at kernel rate with no host round trip (the reference's loop is
 * `a = agent(obs, env); obs, r, done, _ = env.step(a)`, run_environment.py / compare_agents.py).
 *   kind                        score (first maximum wins, NaN scores skipped)              objects considered
 *   SSA_AGENT_NAIVE_GREEDY      trace(P_cur[j])                              agents.py:7    all
 *   SSA_AGENT_VISIBLE_GREEDY    trace(P_cur[j])                              agents.py:36   visible (elevation of x_true >= obs_limit)
 *   SSA_AGENT_SHANNON           log(det P_cur[j] / det P_prev[j])            agents.py:15   visible
 *   SSA_AGENT_POS_ERROR         |x_cur[j][:3] - x_true[j][:3]|               agents.py:66   visible
 *   SSA_AGENT_VEL_ERROR         |x_cur[j][3:] - x_true[j][3:]|               agents.py:75   visible
 * Per env e (objects e*n_obj .. +n_obj): action_out[e] = index of the first maximum, or fallback[e] when no object
 * qualifies (the reference calls action_space.sample() there; the caller supplies that draw), or -1 with fallback NULL.
 * The GCRS->ITRS matrix is row (env_time[e] + time_offset) % n_time of `trans`, as in the step.  pick_out (optional,
 * [E][2] int64): the arg-max (-1 = none) and the winning score's bit pattern.  workspace: ssa_agent_select_workspace_bytes().
 * P_prev may be NULL (no previous step yet): the Shannon score is then NaN for every object -> fallback, as agents.py at i = 0.
 * Two small launches (per-block first maxima over all CUs, then one wavefront per env). */
#define SSA_AGENT_NAIVE_GREEDY 0
#define SSA_AGENT_VISIBLE_GREEDY 1
#define SSA_AGENT_SHANNON 2
#define SSA_AGENT_POS_ERROR 3
#define SSA_AGENT_VEL_ERROR 4
int ssa_agent_select_f64(const ssa_consts *c_host, int32_t kind, const double *x_true, const double *x_cur,
                         const double *P_cur, const double *P_prev, const double *trans, const int32_t *env_time,
                         int32_t time_offset, int32_t n_time, const int32_t *fallback, void *workspace,
                         int32_t *action_out, int64_t *pick_out, int64_t n_obj, int32_t n_env, void *stream);
/* the same with a storage layout (ssa_step_params.obj_ids, one env): the state tensors are in storage order, the action word and the pick
 * name the object as the CALLER numbers it, and the first maximum is the candidate with the lowest such index.  obj_ids NULL: identical to
 * ssa_agent_select_f64. */
int ssa_agent_select_ids_f64(const ssa_consts *c_host, int32_t kind, const double *x_true, const double *x_cur,
                             const double *P_cur, const double *P_prev, const double *trans, const int32_t *env_time,
                             int32_t time_offset, int32_t n_time, const int32_t *fallback, void *workspace,
                             int32_t *action_out, int64_t *pick_out, int64_t n_obj, int32_t n_env, const int32_t *obj_ids, void *stream);
int64_t ssa_agent_select_workspace_bytes(int64_t n_obj, int32_t n_env);

/* ------------------------------------------------ consistency diagnostics
 * NEES  nees[k] = d^T inv(P[k]) d, d = x_true[k] - x[k]   for n (step, object) pairs  (anees(), ssa_tasker_simple_2.py:436-446;
 *       fitness_test() :764-768) -- LU with partial pivoting per lane, NaN where P is singular;
 * NIS   nis[k] = y[k]^T inv(S[k]) y[k]                     (fitness_test() :750-754). */
int ssa_nees_f64(const double *x_true, const double *x, const double *P, double *nees, int64_t n, void *stream);
int ssa_nis_f64(const double *y, const double *S, double *nis, int64_t n, void *stream);
/* chi-square containment of fitness_test() (ssa_tasker_simple_2.py:757-760 NIS, :770-771 NEES): counts[0] = number of v[k] with
 * lo < v[k] < hi (lo, hi = stats.chi2.ppf([alpha/2, 1 - alpha/2], df), computed by the caller), counts[1] = number of non-NaN
 * v[k] (the reference drops NaN NIS values before the mean and keeps NaN NEES values in it).  counts: device int64[2]. */
int ssa_chi2_contained_f64(const double *v, int64_t n, double lo, double hi, int64_t *counts, void *stream);

/* ---------------------------------------------------------------- closed loop: K steps AND their K decisions in one launch
 * The loop `a = agent(obs, env); obs, r, done, _ = env.step(a)` of the reference's drivers (run_environment.py:26-29,
 * compare_agents.py:41-42) for one of its greedy agents (agents.py:7-81, SSA_AGENT_*): what K x [ssa_env_step_f64 +
 * ssa_agent_select_f64] enqueue, with bit-identical states and identical actions, but as ONE persistent launch -- every
 * wavefront keeps its objects in LDS across the steps (as ssa_env_rollout_f64), and the wavefronts agree on the next action
 * among themselves through agent-scope atomics while the next step's predicts (which do not depend on it) already run.
 * One env; every wavefront must be resident at once -- one per four objects plus one service wavefront per 64 of those plus
 * one (they fold the wavefronts' scores into the decision): n_obj <= 20 160 on MI355X; SSA_E_UNSUPPORTED otherwise -- use the
 * per-step calls.  `first` as for ssa_env_rollout_f64.
 *   actions[0]   in : the action of the first step (e.g. from ssa_agent_select_f64 on the state the launch starts from)
 *   actions[k]   out: the agent's choice for step k (0-based), k = 1 .. K  (actions[K]: the decision after the last step)
 *   fallback[k]     : used for actions[k] when no object qualifies at that decision (the reference samples at random,
 *                     agents.py:40-42: the caller supplies the draws); NULL -> -1 (no update)
 *   stats_out[k]    : [SSA_STAT_STRIDE] reward statistics of step k (arg-max sigma_pos only with SSA_LOOP_ARGMAX_SPOS; else -1 / NaN)
 *   upd_out[k]      : [SSA_UPD_STRIDE] update record of step k, or NULL
 *   picks[k]        : optional [K+1][2] int64: arg-max (-1 = fallback used) and the winning score's bits of decision k >= 1
 *   error           : optional device-visible int32, set to 1 if a wavefront waited 2 s for a decision and the launch gave up
 *                     (every wait is bounded; outputs are then incomplete)
 * The rings hold the last `history` steps as after K per-step calls. */
typedef struct ssa_closed_loop_params {
    int32_t n_steps;           /* K >= 1 */
    int32_t history;           /* H >= 2 */
    int32_t slot_out;          /* step k reads slot (slot_out + k - 1) mod H, writes slot (slot_out + k) mod H */
    int32_t agent;             /* SSA_AGENT_* */
    double *x_true_ring, *x_ring, *P_ring, *obs_ring, *metrics_ring;   /* as ssa_rollout_params */
    double *upd_out;           /* [K][SSA_UPD_STRIDE] or NULL */
    double *stats_out;         /* [K][SSA_STAT_STRIDE] */
    int32_t *actions;          /* [K + 1] */
    const int32_t *fallback;   /* [K + 1] or NULL */
    int64_t *picks;            /* [K + 1][2] or NULL */
    int32_t *error;            /* or NULL */
    void *workspace;           /* ssa_closed_loop_workspace_bytes() bytes of device memory (contents irrelevant) */
    int64_t workspace_bytes;
    int64_t wait_ticks;        /* bound of every wait inside the launch, in ticks of the 100 MHz wall clock; 0 = the default (2 s).  A wavefront
                                  that waits longer for a decision publishes the abort generation, every wavefront leaves, `error` is set */
    uint32_t flags;            /* SSA_LOOP_* */
    uint32_t reserved;
    const int32_t *slot_of;    /* with ssa_step_params.obj_ids (a storage layout): [m] device words, slot_of[j] = the position at which the object the
                                  caller calls j is stored (the inverse of obj_ids) -- the wavefronts agree through it on whose tile holds the selected
                                  object; NULL without a layout.  Actions, picks, failure records and the arg-max of sigma_pos speak the caller's
                                  indices (first maximum = the lowest such index), as in the per-step launches */
} ssa_closed_loop_params;
#define SSA_LOOP_ARGMAX_SPOS 1u   /* stats_out[k][SSA_STAT_ARGMAX_SPOS / SSA_STAT_MAX_SPOS] = np.argmax / np.max of sigma_pos after step k (the
                                     'shaped' reward): two more words per wavefront in the exchange */
#define SSA_LOOP_DEBUG_WITHHOLD 2u /* DIAGNOSTIC: the deciding wavefront never publishes -- every wait runs into its bound.  For the test of the
                                     give-up path (tests/test_env_gpu.py): the grid must drain, `error` must be set, later launches must work */
int ssa_env_closed_loop_f64(const ssa_consts *c_host, const ssa_step_params *first, const ssa_closed_loop_params *r, void *stream);
int64_t ssa_closed_loop_workspace_bytes(int64_t n_obj, int32_t n_env);

/* ------------------------------------------------ all-gather by direct peer stores (SURVEY 8e "Collective")
 * The reference runs one env per process and has no exchange step; the sharded env of this library (one env's objects spread over the
 * GPUs of a node) reassembles every step's observation block + statistics words on every rank.  These two entry points do that without a
 * collective library: after the step kernel of step k a rank PUSHES its payload (n_words doubles at src) into the slot it owns in each of
 * n_peer receive buffers -- dst[r], device pointers the host obtained once by mapping every peer's buffer over hipIpc (its own buffer
 * included: dst of the own rank is a local pointer) -- and then stores the step number into flag[r] (release, system scope).  A consumer
 * WAITS until the n_src flag words of its own buffer have reached the step number.  The step number of a launch is seq_base[0] + seq_off:
 * seq_base lives in device memory so that a replayed hipGraph can advance it (as ssa_step_params.env_time does for the time index).
 * Plain kernels: capturable at any world size, no rendezvous inside the launch, no RCCL communicator.  The wait is bounded: a source that
 * has not arrived after timeout_ticks of the 100 MHz wall clock (0 = 2 s) sets error[0] = 1 + its index (if error != NULL) and the kernel
 * ends.  Flags only grow; the caller rotates >= 2 receive buffers (ssa-gym_amd/parallel.py: 3) and pushes step k to a peer only after that
 * peer's payload of step k - 1 has arrived (its readers of the slot being overwritten have then passed, in its stream order): pass the
 * rank's own flag words of the previous step's buffer as prev_flags and the push waits for exactly that, per peer, inside its launch
 * (bounded like ssa_peer_wait; NULL = the caller has waited).  tickets: n_peer zeroed 32-bit device words owned by this exchange (a push
 * is several workgroups per peer; the last one to finish raises the flag and leaves the ticket at zero). */
int ssa_peer_push_f64(const double *src, int64_t n_words, double *const *dst, uint64_t *const *flag, int32_t n_peer,
                      const uint64_t *seq_base, uint64_t seq_off, const uint64_t *prev_flags, int64_t timeout_ticks, int32_t *error,
                      uint32_t *tickets, void *stream);
int ssa_peer_wait(const uint64_t *flags, int32_t n_src, const uint64_t *seq_base, uint64_t seq_off, int64_t timeout_ticks, int32_t *error,
                  void *stream);

/* library identification */
int ssa_abi_version(void);
const char *ssa_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* SSA_HIP_H */
