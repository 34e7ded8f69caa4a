/*
 * stmpc.h -- C-ABI of the MI355X-native ST ("MPC") lattice solver.
 *
 * This is the drop-in boundary for the reference's trajectory-search hot path.
 * Each entry point names the reference interface it replaces (paths relative to
 * the reference checkout jlubars/RL-MPC-LaneMerging):
 *
 *   stmpc_solve_grid          <- st_cy.solve_s_t_path_fast           st_cy.pyx:315-399 (call site st.py:740-746)
 *   stmpc_solve_grid_no_jerk  <- st_cy.solve_s_t_path_no_jerk_fast / _djikstra   st_cy.pyx:96-312 (call site st.py:751)
 *   stmpc_build_grid          <- st.find_s_t_obstacles_from_state    st.py:25-70
 *   stmpc_solve_batch[_device]<- st.get_appropriate_base_st_path_and_obstacles (st.py:726-754) applied to N
 *                                independent HighwayState's (prediction.py:9-20), plus the path-distance probe of
 *                                st.test_guaranteed_crash_from_state (st.py:790-802)
 *   stmpc_predict_batch       <- HighwayState.predict_step_with_ego / predict_step_without_ego (prediction.py:22-105)
 *   stmpc_finer_fit_batch     <- st.finer_fit                        st.py:584-723 (QP via cvxopt.solvers.qp, st.py:16-17,722)
 *   stmpc_st_control_batch[_device] <- st.do_st_control              st.py:757-783 applied to N independent states
 *   stmpc_rollout_step_device / stmpc_combined_decide_device <- dqn.RLAgent.do_combined_control  dqn.py:117-200 (the policy
 *                                network stays the caller's; everything around it runs here)
 *   stmpc_policy_features_device <- dqn.get_state_vector_from_base_state  dqn.py:389-446 (+ the float32 cast and TimeFeature input of ddpg.py:41,84)
 *   stmpc_actor_eval_device   <- DDPGAgent.get_control                ddpg.py:83-87 (state vector + the pretrained policy network, one launch)
 *   stmpc_ego_s               <- control.get_ego_s                   control.py:373-380
 *   stmpc_num_s / stmpc_num_t <- the np.arange sizes at st.py:31-32
 *
 * Plain pointers and sizes only; no torch / numpy types.  All floating point is
 * IEEE fp64.  Functions return 0 on success or a negative STMPC_E* code;
 * stmpc_last_error() returns a thread-local message for the last failure.
 * The library has no CPU fallback: without a HIP device every compute entry
 * returns STMPC_ENODEV.
 *
 * Concurrency: a context owns one set of scratch buffers, work counters and overflow queues; AT MOST ONE batched call may be
 * in flight per context (calls on one stream are naturally ordered; calls on different streams, or from different host
 * threads, need a context each).  The "_device" entries are asynchronous with respect to the host -- with these exceptions: the first
 * wide-lattice solve with a set of dynamics / cost parameters builds a small table on the host (a few ms; its upload is queued on the call's
 * stream, so that call cannot be captured in a hipGraph) -- the context keeps the tables of the four most recent parameter sets, so alternating
 * sets neither rebuild nor wait, and only a fifth set waits for the device (hipDeviceSynchronize) before it replaces the least recently used;
 * and stmpc_combined_decide_device with sparse_control (one integer comes back to the host, see stmpc_combined_cfg).
 * Environment knobs (STMPC_*) are read once, in stmpc_create.
 * Scratch: back-pointers (one byte per lattice cell of a window and time layer -- the distance to the predecessor -- when no step of the dynamics
 * exceeds 255 cells, else two) live per RESIDENT workgroup (84 MB for the first window's 1024 workgroups at H = 40).  Only a search that overflows
 * the first window keeps anything of its own: an entry of a pool (an eighth of the batch, at least 256 entries; 104 KB each at H = 40: 53 MB for
 * 4096 episodes) that receives its back-pointer rows and the layer it continues from in the next window.  The pool is taken only while it is at most
 * a quarter of the device memory that is free at the time (hipMemGetInfo); without it, or beyond its capacity, overflowing searches start over in
 * the wider window instead of continuing (stmpc_stats: resume_refused, pool_exhausted) -- results are the same bits.
 *
 * Arithmetic contract.  Every operation of the reference's search (st_cy.pyx:34-93) is evaluated as one IEEE-754 fp64 operation in the
 * reference's order; the library is built with FP contraction off.  Two kinds of division are formed without the hardware's division
 * sequence, and both return the correctly rounded quotient, i.e. the same bits as `x / d`:
 *   - by the lattice step (per episode): q = x*r, two residual corrections q += fma(-q, d, x)*r with r = RN(1/d) (Markstein);
 *   - by dt, dt^2, dt^3 (per launch): fma(x, zh, x*zl) with zh = RN(1/d), zl = RN(1/d - zh), used only after stmpc_fastdiv2_check has
 *     enumerated every significand of x whose quotient lies within the sequence's error (2^-105 relative) of a rounding midpoint and
 *     has run the sequence on each of them; a divisor that fails (about one in a hundred; 0.3, 0.09, 0.027 pass) gets the kernels
 *     that divide.  Precondition of that proof: x*zl must not underflow, |x| >~ 1e-290; the dividends here are lattice coordinate
 *     differences (multiples of ~1e-2 m up to a few hundred metres, or exactly 0, for which both forms give 0), so this cannot occur.
 *   Results are bit-identical whichever form runs (STMPC_FASTDIV=0 forces the dividing kernels; the test suite runs both).
 * The bounding pre-pass that precedes the exact search works in single precision; it only supplies an upper bound that the exact
 * pass re-checks (a bound that turns out too low is raised and the pass repeated), so it cannot influence any output bit.
 * Limits the reference does not have: STMPC_QP_NMAX = 64 fine samples in st.finer_fit / st.do_st_control -- the fine grid has
 * floor((len - 1) * dt / tick) + 1 samples, so e.g. H = 40 with dt / tick = 1.5 (59 samples) is accepted and H = 44 at that ratio
 * (65) is refused: fine_len = -1 for that state, reported by stmpc_check_error / stmpc_combined_read_state as STMPC_EINVAL;
 * STMPC_KMAX_LIMIT = 32 vehicles per state; STMPC_S_LIMIT lattice cells (16-bit back-pointers).
 *
 * Errors detected on the device (an impossible solver state, a refused QP) are latched in the context: the asynchronous `_device`
 * entries cannot return them, so call stmpc_check_error at the next point where the host synchronises anyway (stmpc_get_stats,
 * stmpc_combined_read_state and the host-pointer entries do it themselves).
 */
#ifndef STMPC_H
#define STMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STMPC_ABI_VERSION 6   /* bumped whenever an exported signature, a struct layout or the accepted values of a field change (6: stmpc_sim_cfg.yield_overlap must be 2); see stmpc_abi_version() */

#define STMPC_OK        0
#define STMPC_EINVAL   -1   /* bad argument (NULL, size, Kmax/H/S out of range) */
#define STMPC_ENODEV   -2   /* no HIP device / device init failed */
#define STMPC_EHIP     -3   /* a HIP runtime call failed (message has hipGetErrorString) */
#define STMPC_ENOMEM   -4   /* device allocation failed */
#define STMPC_EINTERNAL -5  /* solver reported an impossible state (should not happen) */

#define STMPC_KMAX_LIMIT 32   /* max other vehicles per state */
#define STMPC_H_LIMIT    64   /* max time layers */
#define STMPC_S_LIMIT    65000 /* max position cells (back-pointers are 16-bit) */

/* Settings.* that parameterise the path (config.py:30-37,94-110,143,150).  Field
 * order = st.py:727-734 (grid) then the 11 tunables in st_cy.solve_s_t_path_fast's
 * positional order (st.py:740-746) then grid/predictor constants. */
typedef struct stmpc_params {
    double future_s;        /* Settings.FUTURE_S */
    double ds;              /* Settings.S_DISCRETIZATION */
    double dt;              /* Settings.T_DISCRETIZATION */
    double future_t;        /* Settings.FUTURE_T */
    double start_unc;       /* Settings.START_UNCERTAINTY */
    double unc_per_s;       /* Settings.UNCERTAINTY_PER_SECOND */
    double d_w, v_w, a_w, j_w;      /* D/V/A/J_WEIGHT */
    double v_des;           /* DESIRED_SPEED */
    double v_max;           /* MAX_SPEED */
    double a_min, a_max;    /* MAX_NEGATIVE_ACCELERATION, MAX_POSITIVE_ACCELERATION */
    double j_min, j_max;    /* MINIMUM_NEGATIVE_JERK, MAXIMUM_POSITIVE_JERK */
    double min_allowed;     /* MIN_ALLOWED_DISTANCE */
    double car_length;      /* CAR_LENGTH */
    double crash_min_s;     /* CRASH_MIN_S */
    double max_pred_decel;  /* MAX_PREDICTED_DECELERATION */
    double follow_gap;      /* literal 30 at prediction.py:85 */
    double react_thr;       /* HighwayState.ego_reaction_threshold (prediction.py:11) */
    double crash_thr;       /* HighwayState.ego_crash_threshold (prediction.py:12) */
    double comb_min_dist;   /* COMBINATION_MIN_DISTANCE */
} stmpc_params;

/* Per-launch solver statistics (filled by stmpc_get_stats after a batch solve). */
typedef struct stmpc_stats {
    int64_t episodes;        /* N of the last batch */
    int64_t fast_path;       /* episodes finished by the LDS-resident kernel */
    int64_t fallback;        /* episodes whose reachable span overflowed the first LDS window (re-solved in a larger one) */
    int64_t hbm_tier;        /* of those, episodes that ended in the HBM-scratch tier */
    int64_t retries;         /* exact passes repeated because the pre-pass bound was below the reference's terminal cost */
    int64_t nodes_exact;     /* lattice nodes expanded by the exact passes (all tiers, incl. repeated work) */
    int64_t nodes_bound;     /* lattice nodes expanded by the bounding pre-passes */
    double  solve_ms;        /* device time of the last batch (HIP events on the launch stream) */
    double  dp_kernel_ms;    /* device time of the LDS lattice-DP kernel launches (all LDS tiers) */
    int64_t guided;          /* episodes whose bound came from the guided attempt (a tube around the unobstructed optimum) */
    int64_t resume_refused;  /* 1: the batch wanted per-episode back-pointers + checkpoints (an overflowing search then continues in the wider window instead of
                                starting over) and the memory rule turned them down -- they are taken only while they are at most a quarter of the device memory
                                free at that moment (a process shared with torch / RCCL).  Same results, another schedule; re-priced every 64th call. */
    int64_t pool_exhausted;  /* overflowing searches that found the checkpoint pool (an eighth of the batch, at least 256 entries) exhausted and started over in the
                                wider window instead of continuing: same results, more work */
} stmpc_stats;

/* Totals over the launches issued between stmpc_profile(ctx, 1, ..) and stmpc_profile(ctx, 0, &totals). */
typedef struct stmpc_profile_totals {
    int64_t launches;        /* stmpc_solve_batch_device calls */
    int64_t episodes;        /* sum of N */
    double  solve_ms;        /* sum of device time, predictor + DP + second tier (HIP events on the launch stream) */
    double  dp_kernel_ms;    /* sum of device time of the LDS lattice-DP kernel launches (all LDS tiers) */
} stmpc_profile_totals;

typedef struct stmpc_ctx stmpc_ctx;

/* Library / device identification, e.g. "stmpc 0.1 hip gfx950 AMD Instinct MI355X cu=256". */
const char *stmpc_backend_info(void);
const char *stmpc_last_error(void);
/* STMPC_ABI_VERSION the library was built with: a binding compares it with the header it was written against before calling anything else. */
int stmpc_abi_version(void);

/* Context = one HIP device + its staging/scratch buffers.  device < 0 -> current device. */
int  stmpc_create(stmpc_ctx **out, int device);
void stmpc_destroy(stmpc_ctx *ctx);

/* Host-side exact helpers (same libm calls as the reference's Python). */
double stmpc_ego_s(double x, double y);                               /* control.py:373-380 */
int    stmpc_num_s(const stmpc_params *p, double start_s);            /* len(np.arange(...)) st.py:31 */
int    stmpc_num_t(const stmpc_params *p);                            /* len(np.arange(...)) st.py:32 */
double stmpc_path_mean_abs_jerk(const double *s_sequence, int n, double v0, double a0, double dt); /* st.py:274-288 */

/*
 * Batched solve, DEVICE pointers (the hot path; inputs already resident in HBM).
 *   ego      [N][5]     x, y, speed, acceleration, start_s (= get_ego_s(x,y), see stmpc_ego_s)
 *   k_count  [N]        number of other vehicles of each state (<= Kmax), front->back order
 *   other_x  [N][Kmax]  other_xs   (prediction.py:138-141 ordering), entries >= k_count ignored
 *   other_v  [N][Kmax]  other_speeds
 * outputs (any of path_dist / crash may be NULL):
 *   path_idx [N][H]     s index per layer, -1 past best_t   (s_sequence[t] = s_values[path_idx[t]], 0.0 past best_t)
 *   best_t   [N]        deepest layer reached (H-1 = success)            st_cy.pyx:365-369
 *   cost     [N]        accumulated cost of the terminal node (dropped by the reference at st_cy.pyx:393-399)
 *   path_dist[N][H]     distances[t, int((s_t - s_0)/delta_s)] along the path, NaN past best_t   st.py:797-799
 *   crash    [N]        st.test_guaranteed_crash_from_state's verdict    st.py:790-802
 * stream is a hipStream_t (NULL = default stream).  Asynchronous w.r.t. the host.
 */
int stmpc_solve_batch_device(stmpc_ctx *ctx, const stmpc_params *p, int N, int Kmax,
                             const double *d_ego, const int32_t *d_k_count,
                             const double *d_other_x, const double *d_other_v,
                             int32_t *d_path_idx, int32_t *d_best_t, double *d_cost,
                             double *d_path_dist, int32_t *d_crash, void *stream);

/* The same solve with one more output for the multi-GPU gather: action_cost [N][2] fp64 = (path_idx[i][1] as a double -- the cell
 * of the first step, the "action", -1.0 if the path ends at the start cell --, cost[i]) written by the solver's back-track itself, so that the
 * sharded step hands ONE buffer to ONE collective without a packing pass.  d_action_cost may be NULL (= stmpc_solve_batch_device). */
int stmpc_solve_batch_device_ac(stmpc_ctx *ctx, const stmpc_params *p, int N, int Kmax,
                                const double *d_ego, const int32_t *d_k_count,
                                const double *d_other_x, const double *d_other_v,
                                int32_t *d_path_idx, int32_t *d_best_t, double *d_cost,
                                double *d_path_dist, int32_t *d_crash, double *d_action_cost, void *stream);

/* Same with HOST pointers: stages H2D, solves, copies back, synchronises. */
int stmpc_solve_batch(stmpc_ctx *ctx, const stmpc_params *p, int N, int Kmax,
                      const double *ego, const int32_t *k_count,
                      const double *other_x, const double *other_v,
                      int32_t *path_idx, int32_t *best_t, double *cost,
                      double *path_dist, int32_t *crash);

int stmpc_get_stats(stmpc_ctx *ctx, stmpc_stats *out);

/* Synchronises the device and returns (and clears) what the kernels of earlier calls on this context flagged: STMPC_EINTERNAL if a
 * solver kernel met an impossible state (e.g. a queue entry that never appeared: that episode's output rows are then stale),
 * STMPC_EINVAL if st.do_st_control could not re-sample a path (more than STMPC_QP_NMAX fine samples: that state's commanded speed is
 * not valid), else STMPC_OK. */
int stmpc_check_error(stmpc_ctx *ctx);

/* enable != 0: start timing every subsequent stmpc_solve_batch_device launch with its own HIP events (no host
 * synchronisation is added to the launches).  enable == 0: wait for those launches, sum their device times into
 * *out (may be NULL) and stop.  While profiling, stmpc_get_stats reports nothing new. */
int stmpc_profile(stmpc_ctx *ctx, int enable, stmpc_profile_totals *out);

/*
 * Single-episode entry with materialised grids: st_cy.solve_s_t_path_fast's exact
 * argument meaning (st_cy.pyx:315).  HOST pointers.  obstacles [H][S] bytes
 * (non-zero = blocked), distances [H][S], s_values [S], t_values [H];
 * s_sequence_out [H] receives s along the path, 0.0 past the deepest layer reached.
 * Requires H >= 2, S >= 2 (the reference reads s_values[1], t_indices[1] unconditionally).
 */
int stmpc_solve_grid(stmpc_ctx *ctx, const uint8_t *obstacles, const double *s_values, int S,
                     const double *t_values, int H, double ego_start_speed,
                     double ego_start_acceleration, const double *distances,
                     double d_weight, double v_weight, double a_weight, double j_weight,
                     double desired_speed, double max_speed, double negative_acceleration_limit,
                     double positive_acceleration_limit, double negative_jerk_limit,
                     double positive_jerk_limit, double min_allowed_distance,
                     double *s_sequence_out);

/*
 * The reference's two non-production solvers on materialised grids (HOST pointers), same grid arguments as stmpc_solve_grid:
 *   variant 0 <- st_cy.solve_s_t_path_no_jerk_fast      st_cy.pyx:209-312   (node = (t, s))
 *   variant 1 <- st_cy.solve_s_t_path_no_jerk_djikstra  st_cy.pyx:96-206    (node = (t, s, s_prev); the USE_FAST_ST_SOLVER = False
 *                                                                           dispatch of st.py:749-753; needs H*S*S <= 2^28)
 * Both use the constants compiled into the reference's st_cy module (st_cy.pyx:21-31), not Settings.
 */
int stmpc_solve_grid_no_jerk(stmpc_ctx *ctx, int variant, const uint8_t *obstacles, const double *s_values, int S,
                             const double *t_values, int H, double ego_start_speed, const double *distances,
                             double *s_sequence_out);

/*
 * st.find_s_t_obstacles_from_state for one state (HOST pointers): fills
 * obstacles [H][S] (0/1), distances [H][S], s_values [S], t_values [H] with
 * S = stmpc_num_s(p, start_s), H = stmpc_num_t(p).  state5 = x, y, v, a, start_s.
 */
int stmpc_build_grid(stmpc_ctx *ctx, const stmpc_params *p, const double *state5, int k,
                     const double *other_x, const double *other_v,
                     uint8_t *obstacles, double *distances, double *s_values, double *t_values);

/*
 * Batched one-step traffic prediction (HOST pointers), prediction.py:22-105.
 *   mode 0: predict_step_with_ego(selected_speed[i], dt, min_crash_distance)
 *   mode 1: predict_step_without_ego(dt, min_crash_distance)   (selected_speed ignored, may be NULL)
 * state layout as in stmpc_solve_batch but ego is [N][4] = x, y, v, a.  Outputs have the same
 * shapes; crashed [N] receives the crash flag.  stmpc_predict_batch_acc additionally returns other_a_out [N][Kmax] (may be NULL),
 * the new_other_accelerations of prediction.py:86-89,97 (the deceleration applied to a following vehicle, else 0) -- what the RL
 * state vector reads (dqn.get_state_vector_from_base_state, dqn.py:400).  (ABI 2 had given stmpc_predict_batch itself that
 * trailing parameter; ABI 3 restores its original signature and adds the _acc entry.)
 */
int stmpc_predict_batch(stmpc_ctx *ctx, const stmpc_params *p, int mode, int N, int Kmax,
                        const double *ego4, const int32_t *k_count, const double *other_x,
                        const double *other_v, const double *selected_speed, double dt,
                        double min_crash_distance, double *ego4_out, double *other_x_out,
                        double *other_v_out, int32_t *crashed);
int stmpc_predict_batch_acc(stmpc_ctx *ctx, const stmpc_params *p, int mode, int N, int Kmax,
                            const double *ego4, const int32_t *k_count, const double *other_x,
                            const double *other_v, const double *selected_speed, double dt,
                            double min_crash_distance, double *ego4_out, double *other_x_out,
                            double *other_v_out, int32_t *crashed, double *other_a_out);

#define STMPC_QP_NMAX     64   /* max fine samples of st.finer_fit (one wavefront lane per sample) */
#define STMPC_QP_MAXITERS 10   /* solvers.options['maxiters'] = 10, st.py:17 */

/*
 * st.finer_fit (st.py:584-723), batched, HOST pointers: re-samples coarse ST paths (planning step coarse_delta_t)
 * to the simulator tick delta_t by the reference's QP: minimise |x - interp(s)|^2 subject to x_0 = s_0 and speed /
 * acceleration / jerk limits written as finite differences (limits from p: v_max, a_max, a_min, j_max, j_min),
 * solved with cvxopt's coneqp iteration capped at maxiters (the reference uses STMPC_QP_MAXITERS).
 *   s_seq   [N][Hs]  coarse paths, row i valid for len[i] entries (1 <= len[i] <= Hs <= 64)
 *   v0, a0  [N]      start_speed, start_acceleration
 *   bac     [N][4] or NULL: before_s, before_speed, after_s, after_speed (st.py:672-702; +-inf = no such car)
 *   out     [N][n_max] fine paths;  out_len [N] their lengths: 1 when len[i] == 1 (returned as is, st.py:587-588),
 *           -1 when the fine grid would have more than STMPC_QP_NMAX samples (nothing written)
 *   iters   [N] or NULL: iterations done, negated when the cap was reached without meeting cvxopt's tolerances
 */
int stmpc_finer_fit_batch(stmpc_ctx *ctx, const stmpc_params *p, double delta_t, double coarse_delta_t, int maxiters,
                          int N, int Hs, const double *s_seq, const int32_t *len, const double *v0, const double *a0,
                          const double *bac, int n_max, double *out, int32_t *out_len, int32_t *iters);

/*
 * st.do_st_control (st.py:757-783) for N states: lattice search, trailing-zero trim, QP re-sampling when
 * tick_length < p->dt (st.py:771-772), commanded speed (x_1 - x_0) / tick_length, or the current speed when the
 * path has a single point (st.py:775-777).  Inputs as stmpc_solve_batch.  Outputs: speed [N]; best_t [N]
 * (best_t < H-1 <=> "ST Solver finds crash inevitable", st.py:765-766); optional path_idx [N][H], cost [N],
 * fine [N][STMPC_QP_NMAX] + fine_len [N] (the re-sampled path).  The _device form takes device pointers
 * (path_idx, best_t, cost required as scratch) and is asynchronous on `stream`.
 */
int stmpc_st_control_batch(stmpc_ctx *ctx, const stmpc_params *p, double tick_length, int N, int Kmax,
                           const double *ego, const int32_t *k_count, const double *other_x, const double *other_v,
                           double *speed, int32_t *best_t, int32_t *path_idx, double *cost, double *fine,
                           int32_t *fine_len);
int stmpc_st_control_batch_device(stmpc_ctx *ctx, const stmpc_params *p, double tick_length, int N, int Kmax,
                                  const double *d_ego, const int32_t *d_k_count, const double *d_other_x,
                                  const double *d_other_v, int32_t *d_path_idx, int32_t *d_best_t, double *d_cost,
                                  double *d_speed, double *d_fine, int32_t *d_fine_len, void *stream);

/*
 * Combined RL + ST controller, dqn.RLAgent.do_combined_control (dqn.py:117-200), for N independent states, DEVICE pointers.
 * The policy network is the caller's: it proposes one jerk per live episode and rollout step (dqn.py:119,132).  Protocol per tick:
 *   for step = 1 .. max(rollout_length, 1):
 *       action[N] = policy(current state arrays)              (caller, on the device; step 1 uses first_action)
 *       stmpc_rollout_step_device(.., step, ..)               speed from jerk (control.py:160-171), predict_step_with_ego with
 *                                                             COMBINATION_MIN_DISTANCE (dqn.py:136), history / probe-state bookkeeping;
 *                                                             episodes whose rollout ended (crash predicted, x > STOP_X) are left untouched
 *   stmpc_combined_decide_device(..)                          feasibility probe of the rolled-out state (one batched solve), controller solve of
 *                                                             the start state (lattice search + QP re-sampling), decision rules of dqn.py:144-200
 * Outputs: takeover [N] (what the reference appends to takeover_history), reason [N] (0 policy kept, 1 crash predicted, 2 policy too fast,
 * 3 probe rejects the rolled-out state, 4 ST path deemed better), speed [N] (the commanded speed: st.do_st_control's for reasons 1-3, the first
 * re-sampled step for 4, control.get_ego_speed_from_jerk(first_action) for 0).  cur_* arrays are updated in place by the rollout steps;
 * cur_oa [N][Kmax] (may be NULL) receives the other vehicles' accelerations for the policy's state vector.  The rollout history's
 * s coordinates of predicted positions come from the device map of control.get_ego_s (squares as x*x where CPython calls pow).
 */
#define STMPC_ROLLOUT_LIMIT 63
typedef struct stmpc_combined_cfg {
    double tick_length;              /* Settings.TICK_LENGTH */
    double stop_x;                   /* Settings.STOP_X */
    int32_t rollout_length;          /* Settings.ROLLOUT_LENGTH */
    int32_t st_test_rollouts;        /* Settings.ST_TEST_ROLLOUTS */
    int32_t check_rollout_crash;     /* Settings.CHECK_ROLLOUT_CRASH */
    int32_t limit_dqn_speed;         /* Settings.LIMIT_DQN_SPEED */
    int32_t test_rollout_state;      /* Settings.TEST_ROLLOUT_STATE */
    int32_t test_st_strictly_better; /* Settings.TEST_ST_STRICTLY_BETTER */
    int32_t remember_last_choice;    /* Settings.REMEMBER_LAST_CHOICE_FOR_SWITCHING_COMBINED */
    int32_t sparse_control;          /* not a reference flag.  1: like the reference, st.do_st_control(start_state) is solved only for the states whose decision hands
                                        control over (dqn.py:144-155) -- the decide call then makes ONE host round trip (the number of those states) and is no longer
                                        capturable in a hipGraph; ignored (all states are solved) when test_st_strictly_better needs every state's path.  0: all states,
                                        fully asynchronous.  Same decisions and commands either way. */
} stmpc_combined_cfg;
int stmpc_rollout_step_device(stmpc_ctx *ctx, const stmpc_params *p, const stmpc_combined_cfg *cfg, int N, int Kmax, int step,
                              const double *d_ego5_start, double *d_cur_ego4, const int32_t *d_k_count, double *d_cur_other_x,
                              double *d_cur_other_v, double *d_cur_other_a, const double *d_action, void *stream);
int stmpc_combined_decide_device(stmpc_ctx *ctx, const stmpc_params *p, const stmpc_combined_cfg *cfg, int N, int Kmax,
                                 const double *d_ego5_start, const int32_t *d_k_count, const double *d_other_x_start,
                                 const double *d_other_v_start, const double *d_cur_ego4, const double *d_cur_other_x,
                                 const double *d_cur_other_v, const double *d_first_action, const int32_t *d_last_choice_rl,
                                 int32_t *d_takeover, int32_t *d_reason, double *d_speed, void *stream);
/* Decisions taken by stmpc_combined_decide_device on this context and controller solves run for them (equal unless sparse_control). */
int stmpc_combined_counts(stmpc_ctx *ctx, int64_t *decisions, int64_t *control_solves, int reset);

/*
 * The policy network's input for N states on the device: dqn.get_state_vector_from_base_state (dqn.py:389-446: the nearest cars_ahead /
 * cars_behind vehicles in list order as (acceleration, speed [difference], gap, 1), zero tuples where there are fewer, then ego speed,
 * acceleration, x, y; normalised by 9, MAX_SPEED, SENSOR_RADIUS, 300, 100), cast to float32 as the gym wrapper of the reference's RL
 * library hands it to the network (ddpg.py:84: env._make_state), plus, if time_feature, the 21st input of ddpg.py:41 (TimeFeature of
 * autonomous-learning-library 0.5.3): time_scale x d_evals[i] (policy evaluations of that episode so far), after which d_evals[i] counts
 * one up -- for step > 1 only where the context's rollout (stmpc_rollout_step_device) is still going on, because the reference only asks
 * its policy for those states.  The cast and the time feature restate a library that is absent from the reference checkout (parity
 * unpinned); the other 20 entries are pinned to the reference's own function (tests/golden/golden_combined_real.npz).
 * d_feat [N][feat_stride] float32, feat_stride >= stmpc_policy_features_len(cfg).  d_cur_oa may be NULL (accelerations read as 0).
 */
typedef struct stmpc_policy_features_cfg {
    double max_speed;                /* Settings.MAX_SPEED */
    double sensor_radius;            /* Settings.SENSOR_RADIUS */
    double time_scale;               /* TimeFeature.scale: 0.001 */
    int32_t cars_ahead, cars_behind; /* Settings.CARS_AHEAD, Settings.CARS_BEHIND */
    int32_t use_acceleration;        /* Settings.USE_ACCELERATION_OF_OTHER_CARS */
    int32_t use_speed_difference;    /* Settings.USE_SPEED_DIFFERENCE */
    int32_t normalize;               /* Settings.NORMALIZE_VECTOR_INPUT */
    int32_t time_feature;            /* 1: append the TimeFeature input (DDPG agents, ddpg.py:41) */
} stmpc_policy_features_cfg;
int stmpc_policy_features_len(const stmpc_policy_features_cfg *cfg);
int stmpc_policy_features_device(stmpc_ctx *ctx, const stmpc_policy_features_cfg *cfg, int N, int Kmax, int step, const double *d_cur_ego4,
                                 const int32_t *d_k_count, const double *d_cur_other_x, const double *d_cur_other_v, const double *d_cur_other_a,
                                 int32_t *d_evals, float *d_feat, int feat_stride, void *stream);

/*
 * The policy network itself on the device (optional -- a caller may keep the network in its own framework and only take the input vectors from
 * stmpc_policy_features_device): the reference's DDPGAgent.get_control (ddpg.py:83-87) for N states in ONE launch = the state vector above ->
 * Linear(n_in, h1) -> ReLU -> Linear(h1, h2) -> ReLU -> Linear(h2, 1) -> tanh * tanh_scale + tanh_mean (the `all` library's fc_deterministic_policy
 * behind ddpg.py:29-41; 21 -> 400 -> 300 -> 1, scale 5, mean 0 for the reference's pretrained_models/), float32 throughout like the reference's
 * torch modules, hidden layers on the matrix cores (v_mfma_f32_16x16x4_f32: exact f32 fused multiply-adds in a fixed order).
 * stmpc_actor_create takes HOST pointers to row-major weights as torch stores them (w0 [h1][n_in], w1 [h2][h1], w2 [h2], biases) and packs
 * them for the kernel; n_in <= 32, h1, h2 <= 1024 and small enough for one workgroup's LDS.  stmpc_actor_eval_device: arguments as
 * stmpc_policy_features_device (d_feat may be NULL) plus d_jerk [N] fp64, the proposed jerk per state.
 */
typedef struct stmpc_actor stmpc_actor;
int  stmpc_actor_create(stmpc_ctx *ctx, int n_in, int h1, int h2, const float *w0, const float *b0, const float *w1, const float *b1,
                        const float *w2, const float *b2, double tanh_scale, double tanh_mean, stmpc_actor **out);
void stmpc_actor_destroy(stmpc_actor *actor);
int  stmpc_actor_eval_device(stmpc_ctx *ctx, const stmpc_actor *actor, const stmpc_policy_features_cfg *cfg, int N, int Kmax, int step,
                             const double *d_cur_ego4, const int32_t *d_k_count, const double *d_cur_other_x, const double *d_cur_other_v,
                             const double *d_cur_other_a, int32_t *d_evals, float *d_feat, int feat_stride, double *d_jerk, void *stream);

/* Host copies of the context's rollout bookkeeping and intermediate results (synchronises; any pointer may be NULL):
 * live, hist_len, crash_pred, have_test [N]; sel_speed [N]; rollout_s [N][rollout_length + 1]; test_ego4 [N][4], test_ox / test_ov [N][Kmax];
 * probe_crash [N], st_speed [N], fine [N][STMPC_QP_NMAX], fine_len [N] (valid after stmpc_combined_decide_device; with sparse_control st_speed is NaN and
 * fine_len 0 for the states whose decision did not call the controller). */
int stmpc_combined_read_state(stmpc_ctx *ctx, int N, int32_t *live, int32_t *hist_len, int32_t *crash_pred, double *sel_speed,
                              double *rollout_s, int32_t *have_test, double *test_ego4, double *test_ox, double *test_ov,
                              int32_t *probe_crash, double *st_speed, double *fine, int32_t *fine_len);

/*
 * Batched SUMO-free merge episodes: stands in for control.run_episode / control.step (control.py:207-340) so that whole
 * episodes can be run for N environments in lock-step on the device.  The world restates what the reference configures SUMO to do
 * (the "simple traffic distribution", config.py:39): highway vehicles of vType "normal" follow SUMO's Krauss model (Euler form,
 * sigma 0) behind the vehicle ahead and behind the ego once that is on the junction, they enter as control.py:215-226 adds them,
 * the ego obeys its acceleration limits only (speed mode 22, control.py:43) and moves along `ego_route` -- the centre line of its lanes in
 * the reference's network (ramp_0 and the junction's internal lane, merge.net.xml:42,52), whose (x, y) pairs are what TraCI reports and what
 * the reference's planner and trained actors see -- or, without a route, along the straight lines the planner assumes (prediction.py:46-59).
 * It is not SUMO: episode statistics compare with the reference's reports as distributions only.
 *   stmpc_sim_init_device   traffic in its stationary state, ego at the ramp start with control.get_ego_start_speed's draw
 *   stmpc_sim_view_device   planner inputs of every environment (the layout stmpc_solve_batch_device takes; vehicles within the
 *                           sensor radius, front to back; other_a may be NULL)
 *   stmpc_sim_step_device   one tick with the commanded speeds (limited by the vehicle's acceleration limits); finished environments idle
 *   stmpc_sim_read          host copies: status [N] (0 running, 1 arrived = "merged", 2 crashed, 3 out of time), ticks [N], acc [N][STMPC_SIM_NACC] =
 *                           sum of speeds, max speed, sum |jerk|, (internal), samples, closest distance past CRASH_MIN_S, sum and count of those
 *                           distances; then the reference's "disruption" columns (control.py:289-304, stats.py:64-68: deceleration of the nearest
 *                           vehicle behind the ego while ego_s > disruption_min_s): sum, maximum, samples, samples with a non-zero value; and ego [N][4]
 */
#define STMPC_SIM_NACC 12
typedef struct stmpc_sim_cfg {
    double tick_length;              /* Settings.TICK_LENGTH */
    double other_car_speed;          /* Settings.OTHER_CAR_SPEED */
    double base_traffic_interval;    /* Settings.BASE_TRAFFIC_INTERVAL */
    double spawn_x, despawn_x;       /* ends of the highway edges: -250, 100 (merge.net.xml:45-49) */
    double ego_start_x, ego_start_y; /* departPos 40 on the ramp (control.py:42) */
    double arrive_x;                 /* arrivalPos 50 on highwayahead = x 51.5 (control.py:42) */
    double sensor_radius;            /* Settings.SENSOR_RADIUS */
    double start_speed, start_speed_std, min_start_speed, max_start_speed;   /* control.py:198-204 */
    /* the highway vehicles' SUMO vType (merge_impossible.rou.xml:3 "normal": Krauss, accel 4.5, decel 6.0, minGap 1, tau 0.5, length 5; SUMO's
     * defaults: emergencyDecel 9, width 1.8); speed_dev = deviation of the per-vehicle speed factor (0 in the simple distribution) */
    double veh_accel, veh_decel, veh_min_gap, veh_tau, veh_emergency_decel, veh_length, veh_width, speed_dev;
    int32_t vary_traffic_start_times, randomize_start_speed, max_ticks;
    int32_t yield_overlap;           /* must be 2 (anything else: STMPC_EINVAL).  What a highway vehicle does about an ego that is on the junction but whose rear (plus
                                        minGap) is not yet ahead of the vehicle's front -- the ego "laps in": SUMO's link-leader rule as its effect shows in the
                                        reference's "disruption" columns: a vehicle whose front is behind the ego's front is asked to STOP while the ego laps in
                                        (emergency braking, whatever the ego's speed).  Rounds 3-5 compared four other rules under the values 0, 1, 3, 4 (DESIGN.md
                                        section 9); round 6 froze the world at this one and removed them */
    uint64_t seed;
    const double *ego_route_xy;      /* HOST [ego_route_n][2]: polyline of the ego's lane centre line from the ramp's start to the junction exit, x strictly
                                        increasing (beyond its last point the ego keeps that point's y); read by stmpc_sim_init_device, which keeps a device
                                        copy for the following view / step calls.  NULL / fewer than 2 points: straight lines towards (1.5, -1.5) */
    int32_t ego_route_n;
    int32_t reserved0;
    double disruption_min_s;         /* Settings.MERGE_POINT_X (-50): the "disruption" columns are recorded while the ego's s exceeds it (control.py:289) */
} stmpc_sim_cfg;
int stmpc_sim_init_device(stmpc_ctx *ctx, const stmpc_sim_cfg *cfg, int N, void *stream);
int stmpc_sim_view_device(stmpc_ctx *ctx, const stmpc_sim_cfg *cfg, int N, int Kmax, double *d_ego5, int32_t *d_k_count,
                          double *d_other_x, double *d_other_v, double *d_other_a, void *stream);
int stmpc_sim_step_device(stmpc_ctx *ctx, const stmpc_params *p, const stmpc_sim_cfg *cfg, int N, const double *d_cmd_speed, void *stream);
int stmpc_sim_read(stmpc_ctx *ctx, int N, int32_t *status, int32_t *ticks, double *acc /* [N][STMPC_SIM_NACC] */, double *ego4);
/* status [N] into a DEVICE array, asynchronously on `stream`: lets a controller loop mask its own per-environment statistics to the
 * environments that are still running without a host round trip. */
int stmpc_sim_status_device(stmpc_ctx *ctx, int N, int32_t *d_status, void *stream);

/* Device arithmetic probe used by the parity tests: out[i] = a[i] op b[i] evaluated on the GPU
 * with the kernels' compile flags. op: 0 div, 1 sqrt(a), 2 mul, 3 add, 4 fma(a,a,b*b), 5 the five-operation
 * quotient a/b of the FASTDIV kernels, 6 their two-operation quotient a/b.  HOST pointers. */
int stmpc_probe_arith(stmpc_ctx *ctx, int op, const double *a, const double *b, double *out, int n);

/* The check the solver applies to dt, dt^2 and dt^3 (st_cy.pyx:46-50 divides by them once per edge) before it launches the kernels
 * that form these quotients as fma(x, zh, x*zl): returns 1 if that equals x / d for every double x (quotient in the normal range),
 * 0 if some x would round differently or the check cannot be completed -- the solver then divides with the ordinary IEEE sequence.
 * *zl (may be NULL) receives RN(1/d - RN(1/d)).  Host-only, no context. */
int stmpc_fastdiv2_check(double d, double *zl);

#ifdef __cplusplus
}
#endif
#endif /* STMPC_H */
