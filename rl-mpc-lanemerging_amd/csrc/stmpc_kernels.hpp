// stmpc_kernels.hpp -- CDNA4 (gfx950) device code of the ST lattice solver.
//
// Design:
//   k_predict : one workgroup of two wavefronts per 32 / KMAX merge episodes, one LANE per vehicle.  Wavefront 0 runs the
//               reference's traffic predictor (prediction.py:22-105) H-1 times (the follower chain as repeated one-lane
//               DPP shifts); wavefront 1 follows a layer group behind and emits, per time layer, the list of
//               vehicles that obstruct the lattice (st.py:44-65): front/back edge and the
//               blocked index window.  The H x S obstacle / distance grids of the reference
//               are never materialised.
//   k_solve   : one WORKGROUP (4 or 8 wavefronts) per episode, persistent workgroups pulling episodes
//               from a device work counter.  Layer-synchronous forward DP that is equivalent to the
//               reference's heap Dijkstra (st_cy.pyx:315-399).  Per layer the waves compact the nodes
//               to expand into a descending list and consume it in rounds of 64 sources per wave;
//               each lane relaxes its node's candidate cells into the next layer with an exact
//               (cost, predecessor-index) minimum: ds_min_rtn_u64 on the fp64 bit pattern, then --
//               separated by workgroup barriers -- the unique first setter of a cell's final value
//               writes the predecessor key and equal-cost candidates ds_min_u32 it.  One circular LDS
//               window of W cells holds both the layer being expanded and the one being built
//               (sources are consumed top-down and every edge points upward).  A cheap banded
//               pre-pass bounds the terminal cost; the exact pass then expands only nodes within the
//               bound and evaluates only candidates that can stay within it.  An episode whose live
//               span exceeds W is queued, with its bound, for a tier with a larger window (LDS up to
//               8192 cells, else HBM scratch; same code).
//               Scheduling: the first launch hands out the bounding and the exact pass
//               of an episode as separate tasks; the launch of the second window runs alongside it on a
//               side stream and consumes the overflow queue as it fills; an exact pass that cannot build a
//               layer in the first window checkpoints the finished layer and is continued, not restarted,
//               in the second.
//
// Everything is fp64 and compiled with -ffp-contract=off: one IEEE op per reference op.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef unsigned long long u64;
typedef unsigned short u16;

#define STMPC_MAXH 64

// Phase profile of the lattice pass (analysis builds only, -DSTMPC_PHASE_PROF): thread 0 of every workgroup accumulates
// the shader clock between the marks below (waits at a barrier count towards the phase that ends with it) and adds
// the totals to SolveArgs::phase_prof[pass][phase] when the pass ends.  Compiled out of the product build.
#ifdef STMPC_PHASE_PROF
#define STMPC_NPH 16
#define STMPC_PH_DECL unsigned long long ph_acc_[STMPC_NPH] = {0}; unsigned long long ph_t_ = __builtin_readcyclecounter(); unsigned long long ph_cand_ = 0, ph_slots_ = 0; \
                      unsigned long long bw_acc_[8] = {0}; const unsigned long long bw_t0_ = __builtin_readcyclecounter();
// rows 2 + pass of phase_prof: what lane 0 of EVERY wave waited at the barrier with this id (0 S1, 1 B1, 2 B2, 3 B3, 4 B4, 5 S2), slot 15 = the waves' total time
#define STMPC_BARW(id) do { const unsigned long long b0_ = __builtin_readcyclecounter(); M::barrier(); if ((threadIdx.x & 63) == 0) bw_acc_[id] += __builtin_readcyclecounter() - b0_; } while (0)
#define STMPC_PH(k) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); ph_acc_[k] += n_ - ph_t_; ph_t_ = n_; } } while (0)
#ifndef STMPC_PHASE_TIER
#define STMPC_PHASE_TIER -1     /* >= 0: only this tier's passes are accumulated */
#endif
#define STMPC_PH_COUNT(k) do { if (threadIdx.x == 0) ph_acc_[k] += 1ull; } while (0)      /* slots 12 / 13: rounds, candidate batches */
#define STMPC_PH_FLUSH(mode) do { if (STMPC_PHASE_TIER >= 0 && a.tier != STMPC_PHASE_TIER) break; \
                                   if (threadIdx.x == 0 && a.phase_prof) for (int k_ = 0; k_ < 14; ++k_) atomicAdd(&a.phase_prof[(mode) * STMPC_NPH + k_], ph_acc_[k_]); \
                                   if ((threadIdx.x & 63) == 0 && a.phase_prof) { atomicAdd(&a.phase_prof[(mode) * STMPC_NPH + 14], ph_cand_); atomicAdd(&a.phase_prof[(mode) * STMPC_NPH + 15], ph_slots_); \
                                       for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&a.phase_prof[(2 + (mode)) * STMPC_NPH + k_], bw_acc_[k_]); \
                                       atomicAdd(&a.phase_prof[(2 + (mode)) * STMPC_NPH + 15], __builtin_readcyclecounter() - bw_t0_); } } while (0)
// slots 14 / 15 of a pass: candidate evaluations that were offered to a cell / lane-slots the waves executed (64 per slot)
#define STMPC_PH_CAND(okflag) do { ph_cand_ += (unsigned long long)__popcll(__ballot(okflag)); ph_slots_ += 64ull; } while (0)
#else
#define STMPC_PH_DECL
#define STMPC_PH(k) do { } while (0)
#define STMPC_PH_FLUSH(mode) do { } while (0)
#define STMPC_PH_CAND(okflag) do { } while (0)
#define STMPC_PH_COUNT(k) do { } while (0)
#define STMPC_BARW(id) M::barrier()
#endif

namespace stmpc {

static constexpr u64 INF_BITS = 0x7FF0000000000000ull;

// Launch-constant parameters (kernel argument, lives in SGPRs / kernarg segment).
struct DevP {
    double future_s, ds, dt, dt2, dt3;
    double d_w, v_w, a_w, j_w, v_des, v_max, a_min, a_max, j_min, j_max, min_allowed;
    double car_length, obst_min_s /* crash_min_s - min_allowed, st.py:46 */;
    double max_pred_decel, follow_gap, react_thr, crash_thr, crash_dist_thr /* comb_min_dist - car_length, st.py:800 */;
    // prediction.py:24,64-66 compare get_ego_s = +-sqrt(q) of the predicted ego with react_thr.  sqrt is correctly rounded and monotone, so each
    // comparison is a comparison of q itself with a double the host finds by bisection (make_devp): no square root in k_predict's recurrence.
    //   ramp side of the merge point (es = +sqrt q):  es > thr <=> q > q_gt_pos    es < thr <=> q < q_lt_pos
    //   before the ramp's end point  (es = -sqrt q):  es > thr <=> q < q_gt_neg    es < thr <=> q > q_lt_neg
    double q_gt_pos, q_lt_pos, q_gt_neg, q_lt_neg;
    double unc[STMPC_MAXH];     // start_unc + unc_per_s * t_values[t]   (st.py:40)
    int    dunc[STMPC_MAXH];    // int(unc / ds)                         (st.py:41)
    int    H;
    int    dlen;                // int(CAR_LENGTH / ds)                  (st.py:37)
};

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ double dmin_py(double a, double b) { return (b < a) ? b : a; }   // Python/Cython min(a,b)
__device__ __forceinline__ double dmax_py(double a, double b) { return (b > a) ? b : a; }   // Python/Cython max(a,b)
// The same for operands that are never NaN, as ONE v_min_f64 / v_max_f64 instead of compare + two selects + constant moves: equal to
// Python's min / max except for the sign of a zero result, which no consumer here can see (it is multiplied by dt and added to a coordinate).
__device__ __forceinline__ double dmin1(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double dmax1(double a, double b) { return __builtin_fmax(a, b); }

// Edge cost of st_cy.pyx:46-50 without its gap term, as a function of the candidate's offset u = s_n - s from the source (metres):
//   k_v (u - A)^2 + k_a (u - d1)^2 + k_j (u - (2 d1 - d2))^2,   A = v_des dt,  d1 = s - s_1,  d2 = s_1 - s_2
// = K (u - m)^2 + emin with  m = (k_v A + (k_a + 2 k_j) d1 - k_j d2) / K  and  emin = k_v A^2 + k_a d1^2 + k_j (2 d1 - d2)^2 - K m^2.
// Used by the candidate filters only (never for a cost that is output): FMAs are fine here.
struct EdgeQuad {
    double kv, ka, kj, K, invK, w0, w1, w2, kvA2;
    bool ok;            // the cost has a quadratic part (K > 0)
};
__device__ __forceinline__ EdgeQuad edge_quad(const DevP &p) {
    EdgeQuad q;
    q.kv = p.v_w / p.dt2; q.ka = p.a_w / (p.dt2 * p.dt2); q.kj = p.j_w / (p.dt3 * p.dt3);
    q.K = q.kv + q.ka + q.kj;
    q.invK = 1.0 / q.K;
    q.ok = q.K > 0.0 && q.invK < 1e300;
    const double A = p.v_des * p.dt;
    q.w0 = q.kv * A * q.invK; q.w1 = (q.ka + 2.0 * q.kj) * q.invK; q.w2 = -q.kj * q.invK;
    q.kvA2 = q.kv * A * A;
    return q;
}
// m and emin of a source; *mag receives t + K m^2 (the magnitude the cancellation in emin is relative to)
__device__ __forceinline__ void edge_quad_source(const EdgeQuad &q, double d1, double d2, double &m, double &emin, double *mag = nullptr) {
    const double cj = __builtin_fma(2.0, d1, -d2);
    m = __builtin_fma(q.w1, d1, __builtin_fma(q.w2, d2, q.w0));
    const double t = __builtin_fma(q.kj * cj, cj, __builtin_fma(q.ka * d1, d1, q.kvA2));
    const double km2 = q.K * m * m;
    emin = t - km2;
    if (mag) *mag = t + km2;
}

// Wave-wide integer min / max on the DPP network (no LDS round trips): row_shr 1,2,4,8 leave each
// 16-lane row's result in its last lane, row_bcast:15 / row_bcast:31 fold the rows, lane 63 holds the result.
template <bool IS_MIN>
__device__ __forceinline__ int wave_reduce_i(int v) {
    const int ident = IS_MIN ? 0x7fffffff : (int)0x80000000;
#define STMPC_DPP_STEP(CTRL, RMASK)                                                              \
    { int t_ = __builtin_amdgcn_update_dpp(ident, v, CTRL, RMASK, 0xf, false);                   \
      v = IS_MIN ? (t_ < v ? t_ : v) : (t_ > v ? t_ : v); }
    STMPC_DPP_STEP(0x111, 0xf)   // row_shr:1
    STMPC_DPP_STEP(0x112, 0xf)   // row_shr:2
    STMPC_DPP_STEP(0x114, 0xf)   // row_shr:4
    STMPC_DPP_STEP(0x118, 0xf)   // row_shr:8
    STMPC_DPP_STEP(0x142, 0xa)   // row_bcast:15 -> rows 1,3
    STMPC_DPP_STEP(0x143, 0xc)   // row_bcast:31 -> rows 2,3
#undef STMPC_DPP_STEP
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i(int v) { return wave_reduce_i<true>(v); }
__device__ __forceinline__ int wave_max_i(int v) { return wave_reduce_i<false>(v); }
// Wave-wide minimum of a 64-bit key on the DPP network (same steps as wave_reduce_i: no LDS crossbar round trips -- six dependent
// ds_bpermute pairs cost ~900 cycles of a layer's critical path, these thirty VALU instructions ~150)
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#define STMPC_DPP_STEP64(CTRL, RMASK)                                                                           \
    { const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)v, CTRL, RMASK, 0xf, false);   \
      const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)(v >> 32), CTRL, RMASK, 0xf, false); \
      const u64 t_ = ((u64)hi_ << 32) | lo_; v = t_ < v ? t_ : v; }
    STMPC_DPP_STEP64(0x111, 0xf)   // row_shr:1
    STMPC_DPP_STEP64(0x112, 0xf)   // row_shr:2
    STMPC_DPP_STEP64(0x114, 0xf)   // row_shr:4
    STMPC_DPP_STEP64(0x118, 0xf)   // row_shr:8
    STMPC_DPP_STEP64(0x142, 0xa)   // row_bcast:15 -> rows 1,3
    STMPC_DPP_STEP64(0x143, 0xc)   // row_bcast:31 -> rows 2,3
#undef STMPC_DPP_STEP64
    return ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
}
// control.py:373-380 get_ego_s with x*x for the squares (host code uses libm pow exactly like
// the reference; on the device this only feeds the >8 / >11 threshold tests, prediction.py:64-66).
__device__ __forceinline__ double dev_ego_s(double x, double y) {
    const double mpx = -50.9, mpy = 1.72, mp2x = 1.5, mp3x = -51.0;
    const double common_s = mp2x - mp3x;
    double dx = x - mpx, dy = y - mpy;
    if (x < mpx) return -sqrt(dx * dx + dy * dy);
    else if (x < mp2x) return sqrt(dx * dx + dy * dy);
    else return x - mp2x + common_s;
}

// np.arange length, st.py:31 (numpy: ceil((stop - start) / step))
__device__ __forceinline__ int dev_num_s(const DevP &p, double start_s) {
    double stop = start_s + p.future_s + p.ds;
    return (int)ceil((stop - start_s) / p.ds);
}

// ---------------------------------------------------------------- traffic predictor
template <int KMAX>
struct DState {
    double ex, ey, ev, ea;
    int k;
    double xs[KMAX], vs[KMAX];
};

// prediction.py:46-105 (in place). Returns the crash flag.
template <int KMAX>
__device__ __forceinline__ bool dev_predict_with_ego(const DevP &p, DState<KMAX> &s, double sel,
                                                     double dt, double min_crash_distance, double *acc_out = nullptr) {
    const double mp2x = 1.5, mp2y = -1.5;
    double cx = s.ex, cy = s.ey, px, py;
    if (cx < mp2x) {
        double d0 = mp2x - cx, d1 = mp2y - cy;
        // np.linalg.norm of the 2-vector as this image's BLAS evaluates it: fma(d1,d1,d0*d0)
        double nrm = sqrt(__builtin_fma(d1, d1, d0 * d0));
        d0 /= nrm; d1 /= nrm;
        double step = sel * dt;
        d0 *= step; d1 *= step;
        px = cx + d0; py = cy + d1;
        if (py < -1.6) py = -1.6;
    } else {
        py = cy; px = cx + sel * dt;
    }
    double next_acc = (sel - s.ev) / dt;
    double es = dev_ego_s(px, py);
    bool can_crash = es > p.crash_thr;
    bool merged = es > p.react_thr;
    double last_x = __builtin_inf(), last_speed = 0.0;
    bool enc = false;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        if (i < s.k) {
            double ov = s.vs[i], ox = s.xs[i];
            if (ox < px && !enc) {
                enc = true;
                if (merged) { last_x = px; last_speed = sel; }
            }
            double sd = last_speed - ov;
            double xd = last_x - ox;
            double nv;
            double acc = 0.0;                                     // new_other_acceleration, prediction.py:86,89
            if (sd < 0 && xd < p.follow_gap) {
                acc = dmax_py(sd, p.max_pred_decel);
                nv = ov + acc * dt;
            } else nv = ov;
            if (acc_out) acc_out[i] = acc;
            double nx = ox + nv * dt;
            last_x = nx; last_speed = nv;
            s.xs[i] = nx; s.vs[i] = nv;
        }
    }
    bool crashed = false;
    double cdd = dmax_py(p.car_length, min_crash_distance);
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
        if (i < s.k) { if (fabs(s.xs[i] - px) < cdd && can_crash) crashed = true; }
    s.ex = px; s.ey = py; s.ev = sel; s.ea = next_acc;
    return crashed;
}

// prediction.py:22-44 (in place)
template <int KMAX>
__device__ __forceinline__ bool dev_predict_without_ego(const DevP &p, DState<KMAX> &s, double dt,
                                                        double min_crash_distance, double *acc_out = nullptr) {
    double ego_s = dev_ego_s(s.ex, s.ey);
    double ego_x = s.ex;
    double sel = 0.0;
    if (ego_s < p.react_thr || s.k == 0) {
        sel = 0.0;
    } else if (s.xs[0] < ego_x) {
        s.ex = -20.0; s.ey = -10.0; s.ev = 0.0; s.ea = 0.0; sel = 0.0;
    } else {
        double last_speed = 0.0, last_x = 0.0;
        bool found = false;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            if (i < s.k && !found) {
                if (s.xs[i] < ego_x) found = true;
                else { last_speed = s.vs[i]; last_x = s.xs[i]; }
            }
        }
        if (found) { s.ex = last_x - p.car_length - 5; s.ev = last_speed; s.ea = 0.0; }
        sel = last_speed;
    }
    return dev_predict_with_ego<KMAX>(p, s, sel, dt, min_crash_distance, acc_out);
}

// Per-(episode, layer) list of obstructing vehicles.
struct CarTab {
    double *edge;     // [N][H][Kmax][2]  front = o-L-u, back = o+L+u          (st.py:52-53)
    int    *win;      // [N][H][Kmax][2]  blocked index window [imin, imax)     (st.py:60-65)
    int    *nact;     // [N][H]
    int    *num_s;    // [N]   S of the episode
};

// k_predict: one workgroup of TWO wavefronts per E = 32 / KMAX episodes.
//   wavefront 0  the predictor recurrence (prediction.py:22-105 applied H-1 times, st.py:42-43), one LANE PER VEHICLE: the only serial part of the
//                path, and what the whole launch waits for.  A lone wavefront pays 3-5 ns per instruction whatever it does (measured:
//                scripts/lab/exp/issue_latency.hip), so the layer's instruction count is what is minimised:
//                * the follower rule (prediction.py:72-97) is a chain over the vehicles -- each reacts to its leader's NEW speed and position.
//                  Every lane applies the rule to its own vehicle against its left neighbour's current values (one DPP shift) and the sweep is
//                  repeated until no lane changes: the chain's unique solution, exact, after at most k sweeps and typically two (start:
//                  everyone keeps its speed);
//                * which vehicle takes the ego as its leader, and which vehicle the phantom ego tails in the next layer, are ballots;
//                * get_ego_s of the predicted ego feeds two comparisons only: comparisons of the squared distance with precomputed doubles
//                  (DevP::q_*), no square root; an ego that is told to stand still does not move, so the unit vector of prediction.py:49-54 is
//                  formed only for a moving ego on the ramp; the ego's speed and acceleration are never read back, crash flag and vehicle
//                  accelerations are not needed here.
//                Lanes 32-63 repeat lanes 0-31 (a wavefront with half of its lanes disabled issues more slowly, same measurement).
//   wavefront 1  follows one group of layers behind: the obstructing-vehicle list of st.py:44-65 for every (layer, vehicle) -- front / back edge,
//                blocked index window, the break / continue rules as ballots over the layer's lanes, entries compacted in vehicle order, one table
//                row per lane group (coalesced) -- and the guide's cell of the layer checked against each listed vehicle.
// (Round 3 ran all of it as one thread per episode: 119 us of serial prefix per step of 4096 episodes.)
// The vehicle table is written by k_predict and only read by the solve kernels: its rows are wave-uniform data, fetched through the scalar
// cache (s_load) by reading them in the constant address space -- no vector-memory instruction, no per-lane address arithmetic.
typedef const double __attribute__((address_space(4))) stmpc_cdouble;
typedef const int __attribute__((address_space(4))) stmpc_cint;
__device__ __forceinline__ stmpc_cdouble *as_const(const double *q) { return (stmpc_cdouble *)(unsigned long long)q; }
__device__ __forceinline__ stmpc_cint *as_const(const int *q) { return (stmpc_cint *)(unsigned long long)q; }

template <int KMAX> struct PredShape { static constexpr int E = 32 / KMAX; };

// lane i <- lane i-1 within an aligned group of GL lanes (the group's first lane gets something it never uses): a DPP row shift where
// the group lies within a 16-lane row, else a permute through the LDS crossbar
template <int GL>
__device__ __forceinline__ double left_neighbour(double v, int lane) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (GL <= 16)
        return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xf, 0xf, false));
    else
        return __shfl(v, lane > 0 ? lane - 1 : 0, 64);
}

template <int KMAX>
__global__ void __launch_bounds__(128) k_predict(DevP p, int N, int Kmax, const double *__restrict__ ego,
                                                 const int *__restrict__ k_count,
                                                 const double *__restrict__ other_x,
                                                 const double *__restrict__ other_v, CarTab tab,
                                                 unsigned *counters /* [64], zeroed here */, u64 *ubound /* [N] or null, zeroed here */,
                                                 int *queue1 /* [N] or null: first overflow queue, preset to -1 (empty slots) */,
                                                 unsigned *proxy0 /* [N] or null: zeroed here (split tasks) */,
                                                 int *resume_t /* [N] or null: zeroed here */,
                                                 unsigned char *prio_key /* [N] or null: static weight class of the episode, 0 = heaviest */,
                                                 unsigned *sticky /* [2] or null: error flags that survive until stmpc_check_error reads them */,
                                                 const unsigned char *__restrict__ guide_tab /* [(imax+1)*(2D+1)][H-1] steps of the unobstructed optimum, or null */,
                                                 int guide_imax, int guide_D, u16 *guide /* [N][H] out */, int recurrence_only = 0 /* analysis: no table rows */) {
    static_assert(KMAX == 8 || KMAX == 16 || KMAX == 32, "lanes per episode");
    constexpr int E = PredShape<KMAX>::E;
    constexpr int LPI = 64 / KMAX;                       // layers per sweep of wavefront 1
    __shared__ double xs_l[E][STMPC_MAXH][KMAX];         // vehicle positions per layer (wavefront 0 -> wavefront 1)
    __shared__ double unc_l[STMPC_MAXH];
    __shared__ int dunc_l[STMPC_MAXH];
    __shared__ double ep_start[E], ep_slast[E];
    __shared__ int ep_S[E], ep_k[E], gok_l[E];
    __shared__ int gcell_l[E][STMPC_MAXH];
    __shared__ int prog;                                 // layers whose positions are in xs_l
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x * 128 + tid;                // (the grid has at least N threads)
    // the previous solve's error flag is latched before the counters are reused (same wavefront: the read precedes lane 63's store)
    if (g == 0 && sticky && counters[63 /* STMPC_CNT_ERR */]) atomicOr(&sticky[0], 1u);
    if (g < 64) counters[g] = 0u;
    if (g < N) {
        if (ubound) ubound[g] = 0ull;
        if (queue1) queue1[g] = -1;
        if (proxy0) proxy0[g] = 0u;
        if (resume_t) resume_t[g] = 0;
    }
    if (tid < p.H) { unc_l[tid] = p.unc[tid]; dunc_l[tid] = p.dunc[tid]; }
    if (tid == 0) prog = 0;
    const int e0 = blockIdx.x * E;
    // ---- both wavefronts: per-episode constants (wavefront 0 keeps them in registers, wavefront 1 reads them from LDS)
    const int l32 = lane & 31;
    const int el0 = l32 / KMAX, c0 = l32 & (KMAX - 1);
    int k = 0, S = 0;
    double start_s = 0.0, delta = 1.0, ex = 0.0, ey = 0.0, ev0 = 0.0, ea0 = 0.0;
    const bool live = e0 + el0 < N;
    if (live) {
        const int e = e0 + el0;
        ex = ego[e * 5 + 0]; ey = ego[e * 5 + 1]; ev0 = ego[e * 5 + 2]; ea0 = ego[e * 5 + 3]; start_s = ego[e * 5 + 4];
        k = k_count[e];
        k = k < 0 ? 0 : (k > KMAX ? KMAX : k);
        k = k > Kmax ? Kmax : k;                 // the table rows hold Kmax entries (device-pointer callers are not validated on the host)
        S = dev_num_s(p, start_s);
        const double s1 = start_s + p.ds;        // s_values[S-1] per numpy's arange fill
        delta = s1 - start_s;
        if (wave == 0 && lane < 32 && c0 == 0) {
            tab.num_s[e] = S;
            ep_start[el0] = start_s;
            ep_slast[el0] = (S - 1 == 1) ? s1 : start_s + (double)(S - 1) * delta;
            ep_S[el0] = S; ep_k[el0] = k;
        }
    } else if (wave == 0 && lane < 32 && c0 == 0) { ep_S[el0] = 0; ep_k[el0] = 0; ep_start[el0] = 0.0; ep_slast[el0] = 0.0; gok_l[el0] = 0; }
    __syncthreads();
    if (wave == 0) {
        // ================= wavefront 0: the recurrence, lane = (episode el0, vehicle c0) =================
        const int c = c0, el = el0;
        const int grp = lane & ~(KMAX - 1);
        const u64 gmask = (1ull << KMAX) - 1ull;
        const bool wr = lane < 32 && live;
        const bool in = c < k;
        double x = 0.0, v = 0.0;
        if (live && in && c < Kmax) { x = other_x[(size_t)(e0 + el) * Kmax + c]; v = other_v[(size_t)(e0 + el) * Kmax + c]; }
        const int kw = wave_max_i(k);                // most vehicles of any of the wavefront's episodes: sweeps that settle every chain
        // Guide of the guided bounding attempt (SolveArgs::guide): the unobstructed optimum from the lattice state nearest to the episode's start
        // (cells covered per layer at the start speed, and one layer earlier), usable if it stays on the lattice and clear of every vehicle and of
        // its penalty zone (checked by wavefront 1).  Approximate on purpose: it only centres a search whose result is checked like any other bound.
        const unsigned char *grow = nullptr;
        int g_cell = 0;
        bool g_ok = false;
        if (guide_tab && live) {
            const double cps = p.dt / delta;
            int i1 = (int)rint(ev0 * cps), i2 = (int)rint((ev0 - ea0 * p.dt) * cps);
            i1 = i1 < 0 ? 0 : (i1 > guide_imax ? guide_imax : i1);
            int d = i1 - i2; d = d < -guide_D ? -guide_D : (d > guide_D ? guide_D : d);
            if (i1 - d < 0) d = i1;
            if (i1 - d > guide_imax) d = i1 - guide_imax;
            grow = guide_tab + ((size_t)i1 * (2 * guide_D + 1) + (size_t)(d + guide_D)) * (size_t)(p.H - 1);
            g_ok = true;
        }
        if (prio_key && wr && c == 0) {
            // the slower the ego starts, the more of the lattice stays below the cost bound (rank correlation of the exact pass's
            // time with the start speed: -0.57 on the benchmark states)
            double f = ev0 / (p.v_max > 0.0 ? p.v_max : 1.0);
            f = f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f);
            prio_key[e0 + el] = (unsigned char)(f * 255.0);
        }
        // The loop's launch constants, made opaque so that they stay in registers: left to itself the compiler re-loads them from the kernel
        // argument segment inside the loop (DevP is large), and a scalar load + wait in the serial chain costs hundreds of cycles.
        double c_carlen = p.car_length, c_thr = p.react_thr, c_qgn = p.q_gt_neg, c_qgp = p.q_gt_pos, c_qln = p.q_lt_neg, c_qlp = p.q_lt_pos;
        double c_dt = p.dt, c_gap = p.follow_gap, c_dcl = p.max_pred_decel;
        int c_H = p.H;
        asm volatile("" : "+s"(c_carlen), "+s"(c_thr), "+s"(c_qgn), "+s"(c_qgp), "+s"(c_qln), "+s"(c_qlp), "+s"(c_dt), "+s"(c_gap), "+s"(c_dcl), "+s"(c_H));
        const bool dcl_nonpos = c_dcl <= 0.0;
        // control.py:373-380 of a position as the two comparisons with react_thr that prediction.py makes (x*x for the squares, as dev_ego_s);
        // straight-line mask logic, no short-circuit branches
        const double mpx = -50.9, mpy = 1.72, mp2x = 1.5, mp2y = -1.5, mp3x = -51.0;
        const double common_s = mp2x - mp3x;
        auto zone = [&](double zx, double zy, bool &gt, bool &lt) {
            const double dx = zx - mpx, dy = zy - mpy;
            const double q = dx * dx + dy * dy;
            const double lin = zx - mp2x + common_s;
            const bool neg = zx < mpx, pos = zx < mp2x;                     // (neg implies pos)
            const bool mid = pos & !neg, hwy = !pos;
            gt = (neg & (q < c_qgn)) | (mid & (q > c_qgp)) | (hwy & (lin > c_thr));
            lt = (neg & (q > c_qln)) | (mid & (q < c_qlp)) | (hwy & (lin < c_thr));
        };
        // What predict_step_without_ego decides from a state (prediction.py:24-44): stand still / park / tail the vehicle ahead of the first one
        // that is behind the ego.  A ballot over the episode's lanes: is anyone behind, is the front vehicle behind, who is the first.
        bool lt, found, park;
        double lx, lv;                           // position and speed of the vehicle to tail (the last one if nobody is behind)
        auto who_is_behind = [&](double egox) {
            const u64 m = (__ballot(in && x < egox) >> grp) & gmask;
            found = m != 0ull;
            park = (m & 1ull) != 0ull;                                      // prediction.py:28 (the front vehicle; k > 0 is part of `in`)
            const int fb = found ? __ffsll((long long)m) - 1 : k;           // first vehicle behind, or one past the last
            const int src = grp + (fb > 0 ? fb - 1 : 0);
            lx = __shfl(x, src, 64); lv = __shfl(v, src, 64);
        };
        { bool gt_unused; zone(ex, ey, gt_unused, lt); }
        who_is_behind(ex);
        if (wr) xs_l[el][0][c] = x;
        if (wr && c == 0) gcell_l[el][0] = 0;
        int g_next = grow ? (int)grow[0] : 0;          // the row's steps are fetched one layer ahead: the load's latency hides behind the layer's work
        for (int t = 1; t < c_H; ++t) {
            // -- predict_step_without_ego's choice (prediction.py:24-44)
            const bool still = lt || k == 0;
            const bool tail = !still && !park;
            double sel = 0.0;
            if (!still && park) { ex = -20.0; ey = -10.0; }
            if (tail) { if (found) ex = lx - c_carlen - 5; sel = lv; }
            // -- the ego's step (prediction.py:47-59).  An ego that stands still keeps its position bit for bit: cx + (d0 / nrm) * (0 * dt) = cx.
            double px = ex, py = ey;
            const bool ramp = ex < mp2x;
            const bool curved = ramp && sel != 0.0;
            if (__ballot(curved) != 0ull) {
                if (curved) {
                    double d0 = mp2x - ex, d1 = mp2y - ey;
                    // np.linalg.norm of the 2-vector as this image's BLAS evaluates it: fma(d1,d1,d0*d0)
                    const double nrm = sqrt(__builtin_fma(d1, d1, d0 * d0));
                    d0 /= nrm; d1 /= nrm;
                    const double step = sel * c_dt;
                    d0 *= step; d1 *= step;
                    px = ex + d0; py = ey + d1;
                }
            }
            if (!ramp) px = ex + sel * c_dt;
            if (ramp && py < -1.6) py = -1.6;
            bool merged;
            zone(px, py, merged, lt);                  // prediction.py:64-66 (and the next layer's prediction.py:24)
            // -- the vehicles (prediction.py:72-97).  Leader of the front vehicle: nobody (inf, 0); of the first vehicle behind the ego, once
            // the ego has merged: the ego (prediction.py:78-82, old positions against the ego's new one); of everyone else: the left neighbour.
            const u64 mo = (__ballot(in && x < px) >> grp) & gmask;
            const bool lead_ego = merged && mo != 0ull && c == __ffsll((long long)mo) - 1;
            const bool fixed = c == 0 || lead_ego;
            const double fix_x = lead_ego ? px : __builtin_inf(), fix_v = lead_ego ? sel : 0.0;
            //   nv = ov + acc * dt,  acc = (sd < 0 && xd < gap) ? max(sd, decel) : 0   (prediction.py:83-91; ov + 0 * dt = ov), the maximum for
            //   sd < 0 and else 0 taken as min(max(sd, decel), 0) for the usual decel <= 0
            double gv = v, gx = x + v * c_dt;                                      // first guess: everyone keeps its speed
            for (int sweep = 0; sweep < kw; ++sweep) {
                const double nbx = left_neighbour<KMAX>(gx, lane), nbv = left_neighbour<KMAX>(gv, lane);
                const double Lx = fixed ? fix_x : nbx, Lv = fixed ? fix_v : nbv;
                const double sd = Lv - v, xd = Lx - x;
                double acc = __builtin_fmin(__builtin_fmax(sd, c_dcl), 0.0);
                if (!dcl_nonpos) acc = (sd < 0) ? dmax_py(sd, c_dcl) : 0.0;      // (launch-uniform: a positive "deceleration" takes the literal form)
                acc = (xd < c_gap) ? acc : 0.0;
                const double nv = v + acc * c_dt;
                const double nx = x + nv * c_dt;
                const bool changed = in && (nx != gx || nv != gv);
                gx = nx; gv = nv;
                if (__ballot(changed) == 0ull) break;                              // every lane reproduced its value from settled leaders
            }
            x = gx; v = gv;
            ex = px; ey = py;
            who_is_behind(px);                         // the next layer's prediction.py:28-40, on the new positions
            if (wr) xs_l[el][t][c] = x;
            if (grow) {
                const int stp = g_next;
                if (t < c_H - 1) g_next = (int)grow[t];
                g_ok = g_ok && stp != 255;
                g_cell += stp;
                g_ok = g_ok && g_cell < S;
            }
            if (wr && c == 0) gcell_l[el][t] = g_cell;
            if ((t & (LPI - 1)) == LPI - 1 && lane == 0) __hip_atomic_store(&prog, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (wr && c == 0) gok_l[el] = g_ok ? 1 : 0;
        if (lane == 0) __hip_atomic_store(&prog, c_H, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    // ================= wavefront 1: the obstructing-vehicle list of every layer, KMAX lanes per layer =================
    if (recurrence_only) return;
    const int c = lane & (KMAX - 1);
    const int grp = lane & ~(KMAX - 1);
    const u64 gmask = KMAX >= 64 ? ~0ull : ((1ull << KMAX) - 1ull);
    bool bad[E];
#pragma unroll
    for (int el = 0; el < E; ++el) bad[el] = false;
    for (int tb = 0; tb < p.H; tb += LPI) {
        const int need = tb + LPI < p.H ? tb + LPI : p.H;
        while (__hip_atomic_load(&prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(8);
        const int t = tb + (lane / KMAX);
        const bool tin = t < p.H;
        const int tt = tin ? t : 0;
#pragma unroll
        for (int el = 0; el < E; ++el) {
            const int e = e0 + el;
            if (e >= N) break;                                                     // (wave-uniform)
            const int ke = ep_k[el], Se = ep_S[el];
            const double st_s = ep_start[el], s_last = ep_slast[el];
            const double dl = (st_s + p.ds) - st_s;
            const double o = xs_l[el][tt][c] - (-51.0);                            // control.py:388-389
            const bool inb = tin && c < ke;
            const bool below = inb && o < p.obst_min_s;                            // st.py:46-47 break: this vehicle and every later one
            const u64 bm = (__ballot(below) >> grp) & gmask;
            const bool stop = (bm & ((2ull << c) - 1ull)) != 0ull;
            const bool valid = inb && !stop && !(o > s_last + p.car_length);       // st.py:48-49 continue
            const u64 vm = (__ballot(valid) >> grp) & gmask;
            const int slot = __popcll(vm & ((1ull << c) - 1ull));
            const size_t rowbase = ((size_t)e * p.H + tt) * Kmax;
            if (valid) {
                const double unc = unc_l[tt];
                const int dunc = dunc_l[tt];
                const double front = o - p.car_length - unc;
                const double back = o + p.car_length + unc;
                const int i0 = (int)((o - st_s) / p.ds);                           // st.py:60 (trunc toward 0)
                int imin = i0 - p.dlen - dunc; imin = imin < 0 ? 0 : imin;
                int imax = i0 + p.dlen + dunc; imax = imax > Se ? Se : imax;
                if (!(imin < Se && imax > 0)) { imin = 0; imax = 0; }              // st.py:63
                double2 ed; ed.x = front; ed.y = back;
                int2 wd; wd.x = imin; wd.y = imax;
                *(double2 *)&tab.edge[(rowbase + slot) * 2] = ed;
                *(int2 *)&tab.win[(rowbase + slot) * 2] = wd;
                if (guide_tab && tt != 0) {                                        // the guide's cell of this layer against this vehicle
                    const int gc = gcell_l[el][tt];
                    const double g_sn = st_s + (double)gc * dl;
                    const double gd = __builtin_fmin(fabs(g_sn - front), fabs(g_sn - back));
                    if ((gc >= imin && gc < imax) || gd < p.min_allowed) bad[el] = true;
                }
            }
            if (tin && c == 0) {
                tab.nact[(size_t)e * p.H + tt] = __popcll(vm);
                if (guide_tab && tt != 0) { const int gc = gcell_l[el][tt]; guide[(size_t)e * p.H + tt] = (u16)(gc < 65535 ? gc : 65535); }
            }
        }
    }
    // (prog == H: wavefront 0 has written gok_l)
#pragma unroll
    for (int el = 0; el < E; ++el) {
        const int e = e0 + el;
        if (e >= N || !guide_tab) break;
        const bool anybad = __ballot(bad[el]) != 0ull;
        if (lane == 0) {
            const bool ok = gok_l[el] != 0 && !anybad;
            guide[(size_t)e * p.H] = ok ? (u16)0 : (u16)0xffff;
            // Episodes whose unobstructed optimum is clear of the traffic behave: their bounds are tight, their exact passes do not overflow the window (offline:
            // none of 245).  The others hold every long chain of a step (poor bound -> wide search -> window overflow -> second window), so they go first.
            if (prio_key) prio_key[e] = (unsigned char)((ok ? 128 : 0) + (prio_key[e] >> 1));
        }
    }
}

// One-step prediction exposed through the C-ABI (stmpc_predict_batch).
template <int KMAX>
__global__ void __launch_bounds__(64) k_predict_step(DevP p, int mode, int N, int Kmax,
                                                     const double *__restrict__ ego4, const int *__restrict__ k_count,
                                                     const double *__restrict__ other_x, const double *__restrict__ other_v,
                                                     const double *__restrict__ sel, double dt, double mcd,
                                                     double *ego4_out, double *ox_out, double *ov_out, int *crashed,
                                                     double *oa_out /* [N][Kmax] or null: new_other_accelerations */) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    DState<KMAX> s;
    s.ex = ego4[e * 4 + 0]; s.ey = ego4[e * 4 + 1]; s.ev = ego4[e * 4 + 2]; s.ea = ego4[e * 4 + 3];
    int k = k_count[e];
    k = k < 0 ? 0 : (k > KMAX ? KMAX : k);
    k = k > Kmax ? Kmax : k;                 // never beyond the row stride of the caller's arrays
    s.k = k;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        bool in = (i < k) && (i < Kmax);
        s.xs[i] = in ? other_x[(size_t)e * Kmax + i] : 0.0;
        s.vs[i] = in ? other_v[(size_t)e * Kmax + i] : 0.0;
    }
    double acc[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) acc[i] = 0.0;
    bool cr = (mode == 0) ? dev_predict_with_ego<KMAX>(p, s, sel[e], dt, mcd, acc)
                          : dev_predict_without_ego<KMAX>(p, s, dt, mcd, acc);
    ego4_out[e * 4 + 0] = s.ex; ego4_out[e * 4 + 1] = s.ey; ego4_out[e * 4 + 2] = s.ev; ego4_out[e * 4 + 3] = s.ea;
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
        if (i < k && i < Kmax) { ox_out[(size_t)e * Kmax + i] = s.xs[i]; ov_out[(size_t)e * Kmax + i] = s.vs[i]; if (oa_out) oa_out[(size_t)e * Kmax + i] = acc[i]; }
    crashed[e] = cr ? 1 : 0;
}

// ---------------------------------------------------------------- lattice DP
// st_cy.pyx:34-38 distance_penalty, pre-multiplied by d_weight exactly as st_cy.pyx:50 does
__device__ __forceinline__ double dev_weighted_penalty(double d, double min_allowed, double d_w) {
    // one IEEE division either way: 1000000.0 / max(d, 1.0) or 1 / d
    const bool close = d < min_allowed;
    const double num = close ? 1000000.0 : 1.0;
    const double den = close ? dmax_py(d, 1.0) : d;
    return d_w * (num / den);
}

// Correctly rounded x / d for a divisor whose correctly rounded reciprocal r = RN(1/d) is known:
// q0 = RN(x*r) is within 1.5 ulp; one residual step makes it faithful, the second (Markstein's
// theorem: faithful q, exact residual by FMA, r = RN(1/d)) gives RN(x/d).  5 ops instead of the
// ~12 of the generic v_div_scale/v_rcp/v_div_fmas/v_div_fixup expansion.  The FMAs here are
// explicit; nothing else in this file may contract (-ffp-contract=off).
// Cell offsets of the candidate filters, made safe for the conversion to int: clamped to +-1e9, a NaN reads as "no limit" on its side.
__device__ __forceinline__ double cell_lo(double x) { return __builtin_fmin(__builtin_fmax(x, -1e9), 1e9); }
__device__ __forceinline__ double cell_hi(double x) { return __builtin_fmax(__builtin_fmin(x, 1e9), -1e9); }

template <bool FASTDIV>
__device__ __forceinline__ double divc(double x, double d, double r) {
    if constexpr (FASTDIV) {
        double q = x * r;
        double e = __builtin_fma(-q, d, x);
        q = __builtin_fma(e, r, q);
        e = __builtin_fma(-q, d, x);
        q = __builtin_fma(e, r, q);
        return q;
    } else {
        return x / d;
    }
}

// x / d for one of the launch's CONSTANT divisors (dt, dt^2, dt^3) in two operations: zh = RN(1/d), zl = RN(1/d - zh).
// fma(x, zh, RN(x*zl)) equals x/d to 2^-105 relative before its rounding, i.e. it is RN(x/d) unless x/d lies within 2^-52 ulp
// of a midpoint; for a given d only a few dozen significands x can come that close, and the host tries every one of them
// (fastdiv2_ok in stmpc.hip; the argument and its exhaustive replay in small formats: tests/div2_check.py).
// A divisor that fails the check, or was not checked, never gets here: FASTDIV is then false and x / d is used.
template <bool FASTDIV>
__device__ __forceinline__ double divk(double x, double d, double zh, double zl) {
    if constexpr (FASTDIV) return __builtin_fma(x, zh, x * zl);
    else return x / d;
}

#define STMPC_MAX_TIERS 6
#define STMPC_PROXY_MOVED 0xffffffffu   // split tasks: the bounding task sent the episode to the next tier
#define STMPC_CNT_CONSUMED 32   // + 2*tier + {0 unbounded, 1 bounded}: entries of the tier's queue handed out
#define STMPC_CNT_RESIDENT 48   // workgroups of the first launch that have started
#define STMPC_CNT_FINISHED 49   // ... and that have exited
#define STMPC_CNT_ERR 63
#define STMPC_CNT_RETRY 62
#define STMPC_CNT_NODES_EXACT 61
#define STMPC_CNT_NODES_BOUND 60
#define STMPC_CNT_GUIDED 58      // bounds that came from the guided attempt
#define STMPC_CNT_POOL 56        // checkpoint pool entries handed out in this step (first window -> second window, see SolveArgs::pool_bp)
#define STMPC_CNT_POOL_FULL 57   // overflowing searches that found the pool exhausted (they start over in the wider window)

struct SolveArgs {
    DevP p;
    int N;
    int Kmax;
    int W;                 // window cells of this tier (power of two)
    int PW;                // cells of the circular penalty buffer (power of two, <= W)
    int tier;              // 0: pull episodes 0..N-1 from counters[0]; k>=1: walk list k
    int last_tier;         // overflow here is an internal error
    int prune;             // 1: bound the exact pass by a banded pre-pass (dp_pass PASS_BOUND)
    double band;           // cost band of the pre-pass
    int band_cap;          // > 0: the pre-pass steers its band towards this many expanded nodes per layer
    double band2_mult;     // the second pre-pass attempt (penalty zone allowed) uses band * band2_mult
    // table mode inputs
    const double *ego;     // [N][5]
    CarTab tab;
    // grid mode inputs (single episode, materialised grids; st_cy.pyx:315 semantics)
    const uint8_t *obstacles;   // [H][S]
    const double *distances;    // [H][S]
    const double *s_values;     // [S]
    int S_grid;
    double v0_grid, a0_grid;
    // scratch
    u16 *bp;               // [blocks][H][W] back-pointers of this tier: the predecessor's cell (2 bytes), or, with bp_rel8, the distance to it in cells (1 byte)
    int bp_rel8;           // every step of the dynamics is at most 255 cells: back-pointers are stored as one-byte distances (half the scratch and the traffic)
    unsigned char *gscratch;   // HBM-storage variant: [blocks][20*W] bytes
    unsigned *counters;    // [0] tier-0 work counter; [4k] list-k length, [4k+1] list-k work counter; [63] error flag
    int *lists;            // [STMPC_MAX_TIERS][N] episode ids queued for tier k
    u64 *ubound;           // [N] cost bound of an episode: written by the bound-only phase / on overflow of a tier
    int force_general;     // test hook: route every episode to the last tier as if its lattice were not affine
    int phase;             // 0: bound + exact in one go; 1: bounding pre-passes only (writes ubound, proxy); 2: exact, bound from ubound
    const int *order;      // tier-0 episode order for phase 2 (heaviest first) or null
    // Checkpoint / resume of the exact pass across window tiers: when the layer about to be built cannot fit the
    // first window, the pass saves the finished layer (costs + histories of its live cells) and the episode moves
    // on; the wider window loads it and continues from that layer instead of starting over.  Back-pointers of the
    // finished layers stay where tier 0 wrote them (one region per episode).
    // Round 5: only the searches that DO overflow keep anything per episode.  The first window writes its back-pointers per resident workgroup
    // like every other tier; a search that checkpoints takes an entry of a POOL (pool_cap entries, one atomic ticket per step), copies the
    // back-pointer rows of its finished layers there (<= H x W0 bytes, once) and saves the layer next to them; resume_t[e] = layer | entry << 8.
    // An exhausted pool costs that search its resume (it starts over in the wider window), never a result.
    unsigned char *ckpt;   // [pool_cap][ckpt_stride] or null: header {layer, wlo, whi, flags} + cost[W0] + hist[W0]
    size_t ckpt_stride;
    int *resume_t;         // [N] layer to resume at (0 = start over) | pool entry << 8; written before the episode is queued
    unsigned char *pool_bp;   // [pool_cap][H][W0] back-pointer rows of checkpointed searches (element size as bp), or null
    int pool_cap;
    int W0;                // first window (cells)
    int maxshift;          // upper bound of (target cell - source cell) + rounding slack of the interval bookkeeping
    unsigned long long *phase_prof;   // analysis builds: [2][16] clock totals per pass and phase, else null
    int gsh_max;           // a sparse layer spreads a source over up to 2^gsh_max lanes (dp_pass)
    double zl_dt, zl_dt2, zl_dt3;   // low words of 1/dt, 1/dt^2, 1/dt^3 (divk); meaningful when the FASTDIV kernels are launched
    int split;             // tier 0 hands out 2N tasks: the bounding pre-passes of all episodes, then their exact passes
    int concurrent;        // this launch runs alongside the previous tier's and waits for its queue to fill (see k_solve)
    int feeds_concurrent;  // this launch's overflow queue is being consumed while it runs: publish entries with release stores
    int prev_grid;         // workgroups of the producing launch (concurrent consumer: all must be resident, all must finish)
    int always_wait;       // concurrent consumer on compute units the producers cannot use (CU-masked streams): waiting is always safe
    // Handing compute units over to the wider window while the first launch is still running: once fewer than retire_left tasks are
    // left, the workgroups of the units ranked retire_from and above take no more tasks and exit, so that those units become free
    // WHOLE (a second-window workgroup needs a unit's whole LDS) while the queue of overflowing episodes is still filling, instead
    // of quarter by quarter in the launch's tail.  cu_tab: [512] unit keys, [512] rank + 1, [1] units seen.
    // Guided bounding pass (round 3): for episodes whose unobstructed optimum -- looked up in a table the host builds per parameter set: the
    // optimal step sequence of the obstacle-free problem from every lattice state (speed, speed one step earlier) -- meets no vehicle, the first
    // bounding attempt only expands cells within tube_w of that path.  The guide only centres the search: a poor one costs a short wasted pass.
    int band_dense;        // the ordinary bounding attempts of the standard first window run as band_pass (dense windows) before dp_pass is tried
    int tube_dense;        // the guided attempt of the standard first window runs as tube_pass (lane = cell) instead of dp_pass under a tube
    const u16 *guide;      // [N][H] guide cells per layer written by k_predict ([0] = 0xffff: no usable guide), or null = off
    int tube_w;
    int prio_mode;         // (experiment: which estimate prio_thr is compared with)
    int prio_thr;          // first window: an overflowing search with more than this much left (layers x nodes) joins the front class of the next queue; 0 = off
    int retry_move;        // first window: after this many failed exact passes of an episode the next one runs in the second window (0 = never)
    double bound_infl;     // a bounding pass's single-precision total times this is the bound (its rounding alone needs 1.00002; see stmpc.hip)
    double last_infl;      // exact pass: factor on the bound in the candidate filter of the step INTO the last layer (>= 1; see dp_pass, vmin_bits)
    double retry_mult[3];  // growth of a bound that turned out to be below the reference's terminal cost: first, second, third repeat (then unbounded)
    unsigned *cu_tab;      // null = off
    int retire_from;
    long long retire_left;
    unsigned long long wait_ticks;   // concurrent consumer: give up waiting after this many 100 MHz ticks
    unsigned *proxy;       // [N] work estimate written by phase 1 (nodes the pre-passes expanded)
    // outputs
    int *path_idx;         // [N][H]
    int *best_t;           // [N]
    double *cost;          // [N]
    double *path_dist;     // [N][H] or null
    int *crash;            // [N] or null
    int *host_overflow;    // null or TWO words in MAPPED HOST memory: the batch's last launch stores the number of episodes that overflowed the first window / that were queued for the last tier there
    double *action_cost;   // [N][2] or null: (cell of the first step as a double, -1 if the path has none; cost) -- the fused row the multi-GPU gather moves
    double *s_sequence;    // grid mode: [H]
};

template <bool USE_LDS>
struct Mem {
    static constexpr int SCOPE = USE_LDS ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT;
    static __device__ __forceinline__ u64 ld64(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
    static __device__ __forceinline__ void st64(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }
    static __device__ __forceinline__ u64 min64(u64 *p, u64 v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, SCOPE); }
    static __device__ __forceinline__ unsigned ld32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
    static __device__ __forceinline__ void st32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }
    static __device__ __forceinline__ unsigned min32(unsigned *p, unsigned v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, SCOPE); }
    static __device__ __forceinline__ u16 ld16(const u16 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
    static __device__ __forceinline__ void st16(u16 *p, u16 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }
    static __device__ __forceinline__ double ldf(const double *p) {
        return __longlong_as_double((long long)__hip_atomic_load((const u64 *)p, __ATOMIC_RELAXED, SCOPE));
    }
    static __device__ __forceinline__ void stf(double *p, double v) {
        __hip_atomic_store((u64 *)p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, SCOPE);
    }
    // Workgroup barrier.  LDS tiers: wait only for this wave's LDS traffic (lgkmcnt) -- outstanding global
    // back-pointer stores need not be acknowledged here (they are read once, after a full fence).
    static __device__ __forceinline__ void barrier() {
        if constexpr (USE_LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
    }
};

// Per-episode constants shared by the passes of one solve.
struct Ep {
    double start_s, s1, delta, v0, a0;
    double r_dt, r_dt2, r_dt3, r_delta, est_prev, est_second;
    int S, e;
    bool s1_plain;
    u16 *bp;
    int pool;          // second window: pool entry of the checkpoint this search continues from
};

struct PassOut {
    int best_t, best_n;
    u64 best_bits;
    bool pruned;       // some reached node was not expanded (bound / band) or some cell was withheld
    int nodes;         // expanded nodes
    int maxspan;       // widest live span (cells) the pass needed
    int pool;          // first window, return code 2: pool entry that holds the saved layer and the back-pointer rows
    u64 vmin_bits;     // exact pass that ended without a terminal within its bound: cheapest node of the LAST layer (all of them above the bound), ~0 = none
};

// Workgroup-shared scratch of one episode (LDS in every variant).
#define STMPC_MAXWAVES 8
struct WgShared {
    int red[STMPC_MAXWAVES * 4];          // per-wave (min lo, max hi, max fan) of a round
    int agg[8];                           // (-min lo, max hi, max fan, -lowest source) over the whole workgroup (one LDS atomic per wave; reset after every round that used them); the
                                          // bounding pass alternates between two sets (its rounds have no barrier after the candidates)
    u64 best_bits[2][STMPC_MAXWAVES];     // per-wave cheapest node of the layer, by layer parity: the waves' entries are only combined when a pass
    int best_n[2][STMPC_MAXWAVES];        // ends (the deepest non-empty layer's set is then still in place), not once per layer
    u64 min_tot[STMPC_MAXWAVES];          // per-wave cheapest relaxed candidate (PASS_BOUND)
    int cnt[STMPC_MAXWAVES];              // per-wave number of selected cells of the layer
    int flags;                            // bit 0: a reached node was not expanded
    int nlist;
    int work;                             // episode id broadcast / -1 = queue drained
    int rc;
    int path[STMPC_MAXH];                 // back-track of the exact pass; guide of a guided bounding pass
};

enum { PASS_EXACT = 0, PASS_BOUND = 1 };

// Reset of a set of workgroup-wide round figures (min lo, max hi, max fan).  The constants are made opaque where they are used: left to itself
// the compiler materialises the triple once per kernel, runs out of registers, and reloads it from SCRATCH in front of every reset -- a global
// memory round trip for the one wave every other wave then waits for at the next barrier.
__device__ __forceinline__ void agg_reset(int *g) {      // (-min lo, max hi, max fan, -min source): all folded with ds_max_i32
    int nbig = -0x7fffffff, zero = 0;
    asm volatile("" : "+v"(nbig), "+v"(zero));
    g[0] = nbig; g[1] = zero; g[2] = zero; g[3] = nbig;
}

// One forward sweep over the layers by one WORKGROUP of NW wavefronts.  Returns 0 ok, 1 window overflow
// (workgroup-uniform).
//
// PASS_EXACT: the reference's search restricted to nodes whose cost is <= ubits.  Every node with true
//   cost <= the bound gets exactly the reference's (cost, predecessor, history): all of its ancestors
//   cost less, so none of them was cut, and any cut node could only offer candidates above the bound.
//   With ubits = +inf bits this is the full layered DP.
// PASS_BOUND: cheap search for an upper bound of the terminal cost: expands only nodes within `band` of
//   the cheapest node of their layer (the band narrows when a layer expands more than band_cap nodes) and
//   (hardsoft) treats cells closer than min_allowed to a vehicle as blocked; no back-pointers, no tie repair;
//   instead of overflowing the window it drops a layer's lowest sources.  Any complete path it finds is a
//   valid bound (ties between equal offers are broken by arrival, so its node counts may vary from run to run;
//   the exact pass does not depend on them).
//
// Storage per lattice cell (circular, slot = cell & (W-1)):
//   cost[]  u64  fp64 bits of the accumulated cost of the node, +inf bits = not reached
//   hist[]  u32  (index of s_{t-1}) << 16 | (index of s_{t-2}) of the winning chain
//   pen[]   f64  d_weight * distance_penalty of the cell in the layer being relaxed into, < 0 = blocked
//   list[]  u16  cells of the current layer selected for expansion, highest first
// ONE cost/hist array serves both the layer being expanded and the layer being built: the listed sources
// are consumed in DESCENDING rounds of 64*NW cells, each round first loads its sources into registers
// (barrier), and every edge goes to a cell index >= its source (speeds are >= 0), so a target never lands
// on a source that is still unread.
//
// Exact (cost, predecessor) minimum across waves, per round:
//   A  every candidate: old = atomic_min(cost[cell], bits(total));  remember bits, old>bits, old==bits
//   -- barrier --
//   B  candidates that strictly improved the cell at their time and still hold it (cost[cell]==bits) are the
//      unique first setter of the final value: hist[cell] = key
//   -- barrier --
//   C  candidates that met an equal value and still match the final cost: atomic_min(hist[cell], key);
//      key has the predecessor index in the high half, so the smaller predecessor wins (st_cy.pyx:388 order)
// Checkpoint of a finished layer (see SolveArgs::ckpt).  Kept out of line: these run once per overflowing episode and
// must not cost the lattice loop any registers.
template <bool USE_LDS>
__device__ __attribute__((noinline)) void ckpt_save(unsigned char *ck, int W0, const u64 *cost, const unsigned *hist, int WM,
                                                    int t, int wlo, int whi, int flags) {
    typedef Mem<USE_LDS> M;
    u64 *ck_cost = (u64 *)(ck + 16);
    unsigned *ck_hist = (unsigned *)(ck + 16 + (size_t)W0 * 8);
    for (int n = wlo + (int)threadIdx.x; n < whi; n += (int)blockDim.x) {
        ck_cost[n - wlo] = M::ld64(&cost[n & WM]);
        ck_hist[n - wlo] = M::ld32(&hist[n & WM]);
    }
    if (threadIdx.x == 0) { int *hd = (int *)ck; hd[0] = t; hd[1] = wlo; hd[2] = whi; hd[3] = flags; }
}
template <bool USE_LDS>
__device__ __attribute__((noinline)) void ckpt_load(const unsigned char *ck, int W0, u64 *cost, unsigned *hist, int WM,
                                                    int *wlo_out, int *whi_out, int *flags_out) {
    typedef Mem<USE_LDS> M;
    const int *hd = (const int *)ck;
    const int wlo = hd[1], whi = hd[2];
    const u64 *ck_cost = (const u64 *)(ck + 16);
    const unsigned *ck_hist = (const unsigned *)(ck + 16 + (size_t)W0 * 8);
    for (int n = wlo + (int)threadIdx.x; n < whi; n += (int)blockDim.x) {
        M::st64(&cost[n & WM], ck_cost[n - wlo]);
        M::st32(&hist[n & WM], ck_hist[n - wlo]);
    }
    *wlo_out = wlo; *whi_out = whi; *flags_out = hd[3];
}

// RES: 0 no checkpointing, 1 this tier saves a layer it cannot build (first window), 2 this tier may continue from one
// NWX selects the workgroup shape an instantiation is compiled for: 8 = any (up to 8 waves, window sizes from SolveArgs);
// 4 = the standard first window (exactly 4 waves, 2048 cells, 1024 penalty entries); 88 = the standard second window (exactly
// 8 waves, 8192 cells, 4096 penalty entries).  The fixed shapes have their sizes as immediates and search the list segments
// with one compare per wave they really have.
template <int NWX> struct WgShape {
    static constexpr bool fixed = (NWX == 4 || NWX == 88);
    static constexpr int waves = (NWX == 4) ? 4 : 8;                 // (most, or exactly when fixed)
    static constexpr int W = (NWX == 4) ? 2048 : 8192, PW = (NWX == 4) ? 1024 : 4096;
};
template <bool USE_LDS, bool GRID, bool FASTDIV, int KT, int MODE, int FANMAX, bool S1GEN, int RES = 0, int NWX = STMPC_MAXWAVES>
__device__ int dp_pass(const SolveArgs &a, const Ep &ep, WgShared &sh, u64 *cost, unsigned *hist, double *pen,
                       u16 *list, int *chunk_cnt, const double *ltab_e, const int *ltab_w, const int *ltab_n,
                       u64 ubits, double band, bool hardsoft, PassOut &out, const int t_start = 0, const int tube_w = 0) {
    typedef Mem<USE_LDS> M;
    const DevP &p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    typedef WgShape<NWX> SH;
    const int NW = SH::fixed ? SH::waves : (int)(blockDim.x >> 6);
    const int per = NW * 64;
    const int W = SH::fixed ? SH::W : a.W, WM = W - 1;
    const int PW = SH::fixed ? SH::PW : a.PW, PWM = PW - 1;
    const int H = p.H, S = ep.S, e = __builtin_amdgcn_readfirstlane(ep.e);      // (uniform by construction; said so that the vehicle rows are fetched by scalar loads)
    const double start_s = ep.start_s, delta = ep.delta, s1 = ep.s1;
    const double dt = p.dt, dt2 = p.dt2, dt3 = p.dt3;
    const double r_dt = ep.r_dt, r_dt2 = ep.r_dt2, r_dt3 = ep.r_dt3, r_delta = ep.r_delta;
    const double zl_dt = a.zl_dt, zl_dt2 = a.zl_dt2, zl_dt3 = a.zl_dt3;
    const bool s1_plain = ep.s1_plain;
    auto sval = [&](int n) -> double {
        if constexpr (GRID) return a.s_values[n];
        else {
            double v = start_s + (double)n * delta;      // numpy arange: a[i>=2] = start + i*delta
            if constexpr (S1GEN) { if (!s1_plain) { if (n == 1) v = s1; } }   //  a[1] = start + step (S1GEN=false: caller guarantees they coincide)
            return v;
        }
    };
    u16 *bp = ep.bp;
    const EdgeQuad nk = edge_quad(p);     // coefficients of the edge-cost quadratic (see the candidate filter); no filter when the cost has no quadratic part

    STMPC_PH_DECL
    M::barrier();                       // previous users of the arrays are done
    int wlo = 0, whi = 1;
    if (RES == 2 && t_start > 0) {      // continue a pass checkpointed by the first window (see SolveArgs::ckpt)
        int fl = 0;
        ckpt_load<USE_LDS>(a.ckpt + (size_t)ep.pool * a.ckpt_stride, a.W0, cost, hist, WM, &wlo, &whi, &fl);
        if (tid == 0) sh.flags = fl;
    } else if (tid == 0) { M::st64(&cost[0], 0ull); M::st32(&hist[0], 0u); sh.flags = 0; }
    if (tid == 0) { agg_reset(&sh.agg[0]); agg_reset(&sh.agg[4]); }
    M::barrier();
    out.best_t = 0; out.best_n = 0; out.best_bits = 0ull; out.pruned = false; out.nodes = 0; out.maxspan = 0;
    u64 lmin = 0ull;               // cheapest node of the layer being expanded (PASS_BOUND)
    double bandt = band;           // PASS_BOUND: the band in force, steered towards a.band_cap expanded nodes per layer
    int total_nodes = 0;
    int maxspan = 0;
    int last_t = -1;               // PASS_EXACT: deepest layer that had nodes to expand
    int died_t = -1;               // layer whose scan found nothing to expand
    // Dense layers (narrow-lattice kernels, unbounded exact pass: the reference's own lattice).  98 % of the cells between the lowest and the highest
    // reached cell of a layer ARE reached there (measured on the benchmark batch: 420 nodes in a span of 429), so the list of cells to expand is the
    // span itself: no scan, no compaction, no list lookup in front of every source's cost and history -- a fifth of this pass's time.  A cell of the
    // span that was not reached simply leaves its lane idle.  [d_lo, d_hi) = extent of the candidates offered to the layer being expanded.
    constexpr bool DENSE_OK = (MODE == PASS_EXACT) && (FANMAX == 9) && (RES == 0);
    const bool dense = DENSE_OK && ubits == INF_BITS;
    int d_lo = 0, d_hi = 1, nd_lo = 0x7fffffff, nd_hi = 0;
    int wave_nodes = 0;            // dense layers: sources this wave expanded (the list's length is not the node count there)

    const int last_src_layer = (MODE == PASS_BOUND) ? H - 2 : H - 1;
    STMPC_PH(0);                        // 0: pass set-up
    for (int t = t_start; t <= last_src_layer; ++t) {
        const bool relax = t < H - 1;
        int ilo = 0, ihi = 0;
        bool first = true;
        u64 my_best = ~0ull;
        int my_best_n = 0x7fffffff;
        u64 my_min_tot = ~0ull;
        u64 thr = ubits;
        // bound the candidate filter works with: the pass's bound, but last_infl times that for the step into the last layer, so that terminals just
        // above the bound are still recorded: if the pass then ends without a terminal within its bound, the cheapest of them bounds the repeat
        // (out.vmin_bits) -- each is the cost of a real path of exact nodes, hence >= what the reference's search holds for that cell, hence >= its terminal
        const double filt_u = (MODE == PASS_EXACT && ubits < INF_BITS) ? __longlong_as_double((long long)ubits) * ((t == H - 2) ? a.last_infl : 1.0) : 0.0;
        int tube_c = 0, tube_lo = 0, tube_hi = 0x7fffffff;       // guided bounding pass: centre of this layer's tube, cell range of the next layer's
        if constexpr (MODE == PASS_BOUND) {
            if (tube_w > 0) {
                tube_c = sh.path[t];
                const int c1 = sh.path[t + 1 < H ? t + 1 : t];
                tube_lo = c1 - tube_w; tube_hi = c1 + tube_w + 1;
            }
        }
        if constexpr (MODE == PASS_BOUND) {
            // (cells of this pass hold (fp32 cost) << 32 | history, see below: unsigned order = cost order)
            const float lim = __uint_as_float((unsigned)(lmin >> 32)) + (float)bandt;
            thr = ((u64)__float_as_uint(lim) << 32) | 0xffffffffull;
        }
        // obstructing vehicles of layer t+1 (wave-uniform): kept in scalar registers when KT > 0
        int nact = 0;
        double cfront[KT > 0 ? KT : 1], cback[KT > 0 ? KT : 1];
        int cimin[KT > 0 ? KT : 1], cimax[KT > 0 ? KT : 1];
        stmpc_cdouble *cedge = nullptr;
        stmpc_cint *cwin = nullptr;
        if constexpr (!GRID) {
            if (relax) {
                size_t row = (size_t)e * H + (t + 1);
                if constexpr (KT > 0) {
                    // staged in LDS by solve_episode: lane c picks vehicle c, then broadcast to scalars
                    nact = ltab_n[t + 1];
                    double f = 0.0, b = 0.0; int i0 = 0, i1 = 0;
                    if (lane < KT) {
                        f = ltab_e[((t + 1) * KT + lane) * 2]; b = ltab_e[((t + 1) * KT + lane) * 2 + 1];
                        i0 = ltab_w[((t + 1) * KT + lane) * 2]; i1 = ltab_w[((t + 1) * KT + lane) * 2 + 1];
                    }
#pragma unroll
                    for (int c = 0; c < KT; ++c) {
                        cfront[c] = __longlong_as_double(((long long)__builtin_amdgcn_readlane(__double2hiint(f), c) << 32) |
                                                         (unsigned)__builtin_amdgcn_readlane(__double2loint(f), c));
                        cback[c] = __longlong_as_double(((long long)__builtin_amdgcn_readlane(__double2hiint(b), c) << 32) |
                                                        (unsigned)__builtin_amdgcn_readlane(__double2loint(b), c));
                        cimin[c] = __builtin_amdgcn_readlane(i0, c);
                        cimax[c] = __builtin_amdgcn_readlane(i1, c);
                    }
                } else {
                    nact = as_const(a.tab.nact)[row];
                    cedge = as_const(a.tab.edge) + row * a.Kmax * 2;
                    cwin = as_const(a.tab.win) + row * a.Kmax * 2;
                }
            }
        }
        // penalty of cell n in layer t+1 (< 0 = blocked)
        auto cell_penalty = [&](int n) -> double {
            double pv;
            if constexpr (GRID) {
                pv = -1.0;
                if (n < S) {
                    size_t at = (size_t)(t + 1) * S + n;
                    if (!a.obstacles[at]) pv = dev_weighted_penalty(a.distances[at], p.min_allowed, p.d_w);
                }
            } else {
                const double sn = sval(n);
                double d = 1e10;                                         // st.py:34-35
                bool blocked = false;
                if constexpr (KT > 0) {
#pragma unroll
                    for (int c = 0; c < KT; ++c) {
                        if (c < nact) {
                            d = __builtin_fmin(d, fabs(sn - cfront[c]));  // st.py:56-57 (operands are >= 0, never NaN)
                            d = __builtin_fmin(d, fabs(sn - cback[c]));
                            blocked |= (n >= cimin[c]) & (n < cimax[c]); // st.py:64
                        }
                    }
                } else {
                    for (int c = 0; c < nact; ++c) {
                        d = __builtin_fmin(d, fabs(sn - cedge[c * 2 + 0]));
                        d = __builtin_fmin(d, fabs(sn - cedge[c * 2 + 1]));
                        blocked |= (n >= cwin[c * 2 + 0]) & (n < cwin[c * 2 + 1]);
                    }
                }
                if constexpr (MODE == PASS_BOUND) {
                    // single-precision penalty (the gap itself and the zone test stay exact): one v_rcp_f32 instead of an IEEE division
                    if (hardsoft && d < p.min_allowed) blocked = true;
                    const bool close = d < p.min_allowed;
                    const float den = (float)(close ? dmax_py(d, 1.0) : d);
                    const float pf = (float)p.d_w * ((close ? 1000000.0f : 1.0f) * __builtin_amdgcn_rcpf(den));
                    pv = blocked ? -1.0 : (double)pf;
                } else
                pv = blocked ? -1.0 : dev_weighted_penalty(d, p.min_allowed, p.d_w);
            }
            return pv;
        };
        float *const penf = (float *)pen;                     // the bounding pass keeps its penalties in single precision (same buffer, PW entries)
        auto st_pen = [&](int n, double v) {
            if constexpr (MODE == PASS_BOUND) __hip_atomic_store(&penf[n & PWM], (float)v, __ATOMIC_RELAXED, M::SCOPE);
            else M::stf(&pen[n & PWM], v);
        };
        // pen[] is a circular buffer of PW cells; [pv_lo, pv_hi) is the range of cells whose entry is current
        int pv_lo = 0, pv_hi = 0;
        auto note_written = [&](int from, int to) {
            if (to - from > PW) { pv_lo = pv_hi = 0; return; }            // the range aliased itself
            if (pv_hi > pv_lo && from <= pv_hi && to >= pv_lo) {          // touches the current range
                int lo2 = from < pv_lo ? from : pv_lo, hi2 = to > pv_hi ? to : pv_hi;
                if (hi2 - lo2 > PW) { if (from < pv_lo) hi2 = lo2 + PW; else lo2 = hi2 - PW; }   // far end evicted
                pv_lo = lo2; pv_hi = hi2;
            } else { pv_lo = from; pv_hi = to; }
        };
        auto init_cells = [&](int from, int to) {            // new next-layer cells: penalty + "not reached"
            for (int n = from + tid; n < to; n += per) {
                st_pen(n, cell_penalty(n));
                M::st64(&cost[n & WM], INF_BITS);
                if constexpr (MODE == PASS_EXACT) M::st32(&hist[n & WM], 0xFFFFFFFFu);      // (above every key: see the merged winner stage)
            }
            note_written(from, to);
        };
        auto repen = [&](int from, int to) {                 // penalty entries that were evicted and are needed again
            for (int n = from + tid; n < to; n += per) st_pen(n, cell_penalty(n));
            note_written(from, to);
        };

        // ---- scan: compact the cells of layer t that get expanded (cost <= thr), highest first.
        // 64-cell chunks, chunk j = cells [top0-64(j+1), top0-64j); wave w owns the contiguous chunks
        // [w*cpw, (w+1)*cpw) and writes its cells into its own segment of list[] (capacity cpw*64), so one pass
        // and one barrier suffice; a global list index is mapped to (segment, offset) with the segment counts.
        STMPC_PH(1);                    // 1: layer set-up (threshold, vehicle rows)
        const int top0 = (whi + 63) & ~63;
        const int nch = (top0 - (wlo & ~63)) >> 6;
        const int cpw = (nch + NW - 1) / NW;
        if (!dense) {
            bool pruned_l = false;
            int wn = 0;
            const int jend = (wave + 1) * cpw < nch ? (wave + 1) * cpw : nch;
            for (int jc = wave * cpw; jc < jend; ++jc) {
                const int i = top0 - 64 * (jc + 1) + lane;
                const u64 cb = ((i >= wlo) & (i < whi)) ? M::ld64(&cost[i & WM]) : INF_BITS;
                const bool reached = cb < INF_BITS;
                bool act = cb <= thr && reached;
                if constexpr (MODE == PASS_BOUND) { if (tube_w > 0) { const int dg = i - tube_c; act = act && dg <= tube_w && dg >= -tube_w; } }
                pruned_l |= reached && !act;
                const u64 amask = __ballot(act);
                if (act) {
                    if (cb < my_best || (cb == my_best && i < my_best_n)) { my_best = cb; my_best_n = i; }
                    const int above = (lane == 63) ? 0 : __popcll(amask >> (lane + 1));
                    M::st16(&list[wave * cpw * 64 + wn + above], (u16)i);
                }
                wn += __popcll(amask);
            }
            if (__ballot(pruned_l) && lane == 0) atomicOr(&sh.flags, 1);
            if constexpr (MODE == PASS_EXACT) {      // the wave's cheapest selected node, smallest cell among equals
                const u64 wb = wave_min_u64(my_best);
                const int wbn = wave_min_i(my_best == wb ? my_best_n : 0x7fffffff);
                my_best = wb; my_best_n = wbn;
            }
            if (lane == 0) {
                sh.cnt[wave] = wn;
                if constexpr (MODE == PASS_EXACT) { sh.best_bits[t & 1][wave] = my_best; sh.best_n[t & 1][wave] = my_best_n; }
            }
        }
        STMPC_BARW(0);       // S1
        STMPC_PH(2);                    // 2: scan + S1
        int segbase[SH::waves + 1];
        segbase[0] = 0;
#pragma unroll
        for (int w = 0; w < SH::waves; ++w)           // wave-uniform: keep the boundaries in scalar registers
            segbase[w + 1] = segbase[w] + (dense ? 0 : __builtin_amdgcn_readfirstlane(w < NW ? sh.cnt[w] : 0));
        const int nlist = dense ? (d_hi > d_lo ? d_hi - d_lo : 0) : segbase[SH::waves];
        auto list_at = [&](int g) -> int {                   // g-th selected cell of the layer, descending
            if constexpr (DENSE_OK) { if (dense) return d_hi - 1 - g; }
            int w = 0;
#pragma unroll
            for (int k = 1; k < SH::waves; ++k) w += (g >= segbase[k]) ? 1 : 0;
            int basew = segbase[0];
#pragma unroll
            for (int k = 1; k < SH::waves; ++k) basew = (w == k) ? segbase[k] : basew;
            return (int)M::ld16(&list[w * cpw * 64 + (g - basew)]);
        };
        if (!dense) total_nodes += nlist;
        if constexpr (MODE == PASS_BOUND) {
            // beam-like control of the band: a layer that expanded more than band_cap nodes narrows the next one in
            // proportion, a thin one lets it recover (square root), never beyond the nominal band
            if (a.band_cap > 0 && nlist > 0) {
                const double f = (double)a.band_cap / (double)nlist;
                bandt = bandt * (f < 1.0 ? f : sqrt(f));
                bandt = bandt > band ? band : (bandt < 0.05 * band ? 0.05 * band : bandt);
            }
        }
        if (nlist == 0) { died_t = t; break; }      // nothing to expand in layer t: the deepest layer reached is t-1
        const double rad_b = (MODE == PASS_BOUND && nk.ok) ? (double)__builtin_amdgcn_sqrtf((float)(bandt * nk.invK)) : 0.0;   // half-width of the bounding pass's candidate interval (metres)
        const int smin = __builtin_amdgcn_readfirstlane(list_at(nlist - 1));      // lowest source of the layer (list[] is final since S1)
        if constexpr (MODE == PASS_EXACT) {
            last_t = t;                     // deepest non-empty layer so far (its cheapest node is looked up after the loop)
            if (RES == 1 && relax && t > 0) {
                // will layer t+1 fit?  Every target lies within maxshift cells above its source, so the live span of
                // the layer about to be built is at most (highest source + maxshift) - lowest source.
                const int top_src = list_at(0), low_src = smin;
                if (top_src + a.maxshift - low_src > W) {
                    // a pool entry for the saved layer and the back-pointer rows written so far (layers 1 .. t-1: layer t's are written
                    // when it is expanded, i.e. by the window that continues)
                    if (tid == 0) {
                        int idx = (int)atomicAdd(&a.counters[STMPC_CNT_POOL], 1u);
                        if (idx >= a.pool_cap) { idx = -1; atomicAdd(&a.counters[STMPC_CNT_POOL_FULL], 1u); }
                        sh.work = idx;
                    }
                    M::barrier();
                    const int idx = sh.work;
                    M::barrier();
                    if (idx < 0) return 1;      // no room: an ordinary overflow, the wider window starts this search over
                    ckpt_save<USE_LDS>(a.ckpt + (size_t)idx * a.ckpt_stride, a.W0, cost, hist, WM, t, wlo, whi, (int)sh.flags);
                    {
                        const size_t esz = a.bp_rel8 ? 1 : 2, row = (size_t)W * esz;          // (W == W0 in the window that saves)
                        const uint4 *src = (const uint4 *)((const unsigned char *)bp + row);
                        uint4 *dst = (uint4 *)(a.pool_bp + (size_t)idx * H * row + row);
                        const size_t n16 = (size_t)(t - 1) * row / 16;
                        for (size_t x = tid; x < n16; x += blockDim.x) dst[x] = src[x];
                    }
                    out.best_t = t;             // the layer that was saved
                    out.nodes = nlist;          // (nodes of the layer that was saved: what is left to do scales with it, see solve_episode)
                    out.pool = idx;
                    return 2;
                }
            }
        }

        STMPC_PH(3);                    // 3: list boundaries, layer's best node, checkpoint test
        // ---- expand: rounds of listed sources, highest cells first.  A sparse layer spreads each source over G = 2^gsh
        // adjacent lanes (lane `sub` of a group takes the candidates sub, sub + G, ...), so that a layer with few nodes
        // still fills the workgroup's lanes: the candidate loop runs ceil(fan / G) slots instead of fan.  All lanes of a
        // group load the same source (LDS broadcast) and derive the same range; which lane evaluates a candidate
        // does not matter to the staged minimum below.
        for (int r0 = 0, rstep = per, rpar = 0; r0 < nlist && (relax || MODE == PASS_EXACT); r0 += rstep, ++rpar) {
            const int ab = (MODE == PASS_BOUND) ? ((rpar & 1) << 2) : 0;      // set of workgroup-wide figures this round uses
            // (chosen per round: the short last round of a wide layer is spread as well)
            int gsh = 0;
            if (relax) { while (gsh < a.gsh_max && ((nlist - r0) << (gsh + 1)) <= per) ++gsh; }
#ifdef STMPC_EXP_GSH_MIN      /* experiment (round 6): every source on at least 2^STMPC_EXP_GSH_MIN lanes through range, filter and candidates */
            if (relax && MODE == PASS_EXACT && gsh < STMPC_EXP_GSH_MIN) gsh = STMPC_EXP_GSH_MIN;
#endif
            const int sub = tid & ((1 << gsh) - 1);
            rstep = per >> gsh;
            const int srcidx = r0 + (tid >> gsh);
            bool inlist = srcidx < nlist;
            const int i = inlist ? list_at(srcidx) : 0;
            u64 cb = INF_BITS;
            unsigned h = 0u;
            if constexpr (DENSE_OK) { if (dense && inlist) { cb = M::ld64(&cost[i & WM]); inlist = cb < INF_BITS; } }
            if (inlist) {
                if (!dense) cb = M::ld64(&cost[i & WM]);
                if constexpr (MODE == PASS_BOUND) h = (unsigned)cb;       // the bounding pass keeps the history in the cell's low word
                else h = M::ld32(&hist[i & WM]);
            }
            double sv = 0.0, p1 = 0.0, p2 = 0.0;
            const double C = __longlong_as_double((long long)cb);            // (PASS_EXACT)
            const float Cf = __uint_as_float((unsigned)(cb >> 32));           // (PASS_BOUND)
            double q_smin = 0.0;      // PASS_BOUND: minimiser of the edge-cost quadratic of this source and
            float q_base = 0.0f;      //             its cost + the quadratic's minimum, in single precision
            int lo = 0, hi = 0;
            unsigned key = 0u;        // what a target won by this source stores: i << 16 | p1 index
            int pr = 0;
            bool cut_l = false;
            if (inlist) {
                sv = sval(i);
                if (t == 0) { p1 = ep.est_prev; p2 = ep.est_second; key = 0u; }      // st_cy.pyx:342
                else {
                    pr = (int)(h >> 16);
                    const int pp = (int)(h & 0xFFFFu);
                    p1 = sval(pr);
                    p2 = (t == 1) ? ep.est_prev : sval(pp);
                    key = ((unsigned)i << 16) | (unsigned)pr;
                }
                if (relax) {
                    // st_cy.pyx:65-75
                    double prev_v = divk<FASTDIV>(p1 - p2, dt, r_dt, zl_dt);
                    double v = divk<FASTDIV>(sv - p1, dt, r_dt, zl_dt);
                    double acc = divk<FASTDIV>(v - prev_v, dt, r_dt, zl_dt);
                    double min_a = dmax1(acc + p.j_min * dt, p.a_min);
                    double max_a = dmin1(acc + p.j_max * dt, p.a_max);
                    double min_v = dmax1(v + min_a * dt, 0.0);
                    double max_v = dmin1(v + max_a * dt, p.v_max);
                    double min_s = sv + min_v * dt;
                    double max_s = sv + max_v * dt;
                    // st_cy.pyx:78-93
                    double x = divc<FASTDIV>(min_s - start_s, delta, r_delta);
                    int mi = (int)x;
                    int ma = (int)divc<FASTDIV>(max_s - start_s, delta, r_delta);
                    if (mi < x) mi += 1;
                    lo = mi; hi = ma + 1;
                    if (hi > S) hi = S;                                      // st_cy.pyx:379
                    if (lo < i) lo = i;                                      // cannot happen (min_v >= 0); keeps the in-place invariant
                    if constexpr (MODE == PASS_EXACT) {
                        // Candidates that cannot stay within the bound are not evaluated.  Without the gap penalty (>= 0)
                        // the edge cost is a quadratic in the candidate position s_n (st_cy.pyx:46-50):
                        //   k_v (s_n - c_v)^2 + k_a (s_n - c_a)^2 + k_j (s_n - c_j)^2  <=  U - C
                        // holds on an interval around its minimiser.  This is a conservative filter (slack inflated by 1e-9, relative
                        // and absolute, against the rounding of the evaluated cost, which is below 1e-12 relative), so
                        // every candidate it drops has total cost > U; whatever it keeps is evaluated exactly as before.
                        if (nk.ok && ubits < INF_BITS && hi > lo) {
                            // (in the source's own frame: u = s_n - sv, see EdgeQuad; d1, d2 are the differences the range above already formed)
                            double m_, emin, mag;
                            edge_quad_source(nk, sv - p1, p1 - p2, m_, emin, &mag);
                            // slack inflated by 1e-9 relative and absolute against the rounding of the evaluated cost, and by 1e-12 of the
                            // magnitude emin is a difference of (its own rounding: a few 1e-16 of that)
                            const double slack = __builtin_fma(filt_u - C, 1.0 + 1e-9, __builtin_fma(mag, 1e-12, 1e-9));
                            const double room = slack - emin;
                            if (!(room >= 0.0)) { if (hi > lo) cut_l = true; lo = 0; hi = 0; }
                            else {
                                // radius from the hardware's single-precision square root (1 ulp), nudged up by 1e-5: an over-estimate of at most
                                // 0.05 cell even for a radius of 5000 cells.  Cells outside [m - rad, m + rad] cost more than the inflated slack;
                                // 0.01 cell covers the rounding of the lattice coordinates and of this interval.
                                const double rad = (double)(__builtin_amdgcn_sqrtf((float)(room * nk.invK)) * 1.00001f);
                                // (clamped before the conversion: an over-sized or non-finite radius -- a huge bound against a tiny K -- must degenerate to
                                // "no filtering", not to a saturated conversion that wraps i + INT_MAX + 1 negative and drops every candidate)
                                const int nlo_ = i + (int)cell_lo(ceil(__builtin_fma(m_ - rad, r_delta, -0.01)));
                                const int nhi_ = i + (int)cell_hi(floor(__builtin_fma(m_ + rad, r_delta, 0.01))) + 1;
                                if (nlo_ > lo || nhi_ < hi) cut_l = true;
                                lo = nlo_ > lo ? nlo_ : lo; hi = nhi_ < hi ? nhi_ : hi;
                            }
                        }
                    }
                    if constexpr (MODE == PASS_BOUND) {
                        // Heuristic counterpart for the pre-pass: a child whose quadratic cost exceeds the source's best by
                        // more than the band would fall outside the next layer's band anyway (the pre-pass may drop
                        // anything; it only has to find some complete path).
                        if (nk.ok && hi > lo) {
                            // the three quadratic terms of st_cy.pyx:46-50 as ONE quadratic K (s_n - smin_)^2 + emin: what the candidate loop evaluates
                            double m_, emin;
                            edge_quad_source(nk, sv - p1, p1 - p2, m_, emin);
                            q_smin = sv + m_;
                            q_base = Cf + (float)emin;
                            // (no spare cells beyond the interval: the band is a heuristic)
                            const int nlo_ = i + (int)cell_lo(ceil((m_ - rad_b) * r_delta));
                            const int nhi_ = i + (int)cell_hi(floor((m_ + rad_b) * r_delta)) + 1;
                            // never drop everything: a source whose whole window lies off the minimiser keeps its nearest end
                            const int a_ = nlo_ > lo ? nlo_ : lo, b_ = nhi_ < hi ? nhi_ : hi;
                            if (a_ < b_) { lo = a_; hi = b_; }
                            else if (nlo_ >= hi) lo = hi - 1;
                            else hi = lo + 1;
                        } else { q_smin = sv; q_base = Cf; }      // (no quadratic part: the edge cost is the gap penalty alone)
                        if (tube_w > 0) { lo = lo > tube_lo ? lo : tube_lo; hi = hi < tube_hi ? hi : tube_hi; }      // targets outside the next layer's tube would not be selected anyway
                    }
                    if (lo >= hi) { lo = 0; hi = 0; }
                }
            }
            if constexpr (MODE == PASS_EXACT) { if (__ballot(cut_l) && lane == 0) atomicOr(&sh.flags, 1); }
            STMPC_PH(4);                // 4: source load, range, candidate filter
            STMPC_PH_COUNT(12);
            if (relax) {
                const int clo_w = wave_min_i(hi > lo ? lo : 0x7fffffff), chi_w = wave_max_i(hi);
                const int fan_w = wave_max_i(hi - lo);
                // lowest source of the wave's share of the round: the list is descending, so it is the last listed lane's
                const u64 im = __ballot(inlist);
                const int low_w = im != 0ull ? __builtin_amdgcn_readlane(i, 63 - __builtin_clzll(im)) : 0x7fffffff;
                // four lanes, one LDS write and one LDS atomic: the wave's figures, and their fold into the workgroup-wide ones -- all as
                // maxima (the two minima negated), so that a single ds_max_i32 with four addresses does it
                if (lane < 4) {
                    const int raw = lane == 0 ? clo_w : (lane == 1 ? chi_w : (lane == 2 ? fan_w : low_w));
                    sh.red[wave * 4 + lane] = raw;
                    atomicMax(&sh.agg[ab + lane], (lane == 0 || lane == 3) ? -raw : raw);
                }
            }
            STMPC_BARW(1);    // B1: the round's sources are in registers: their cells may now be overwritten
            STMPC_PH(5);                // 5: wave reductions + B1
            // The round takes the leading kw waves' sources: as many as keep the targets within the penalty buffer -- nearly always all
            // of them, which the workgroup-wide figures show at once; the per-wave walk is the rare fallback.
            int kw = NW, clo = 0x7fffffff, chi = 0, fan = 0, low_all = 0;
            bool agg_used = false;
            if (relax) {
                clo = -sh.agg[ab + 0]; chi = sh.agg[ab + 1]; fan = sh.agg[ab + 2]; low_all = -sh.agg[ab + 3];
                agg_used = chi > clo || low_all != 0x7fffffff;      // (untouched initial values otherwise: nothing to reset; a round whose sources have
                                                                    // no candidate at all still folded its lowest source in)
                if (chi > clo && (((chi + 63) & ~63) - (clo & ~63)) > PW) {
                    kw = 0; clo = 0x7fffffff; chi = 0; fan = 0;
                    for (int w = 0; w < NW; ++w) {
                        const int l_ = sh.red[w * 4 + 0], h_ = sh.red[w * 4 + 1], f_ = sh.red[w * 4 + 2];
                        const int l2 = l_ < clo ? l_ : clo, h2 = h_ > chi ? h_ : chi;
                        if (w > 0 && h2 > l2 && (((h2 + 63) & ~63) - (l2 & ~63)) > PW) break;
                        clo = l2; chi = h2; fan = f_ > fan ? f_ : fan; kw = w + 1;
                    }
                }
                rstep = (64 * kw) >> gsh;
            }
            const bool act = inlist && wave < kw;
            if constexpr (DENSE_OK) {
                if (dense) {                               // what the scan does for a listed layer: the nodes' count, the layer's cheapest node
                    const bool mine = act && sub == 0;
                    wave_nodes += __popcll(__ballot(mine));
                    if (mine && (cb < my_best || (cb == my_best && i < my_best_n))) { my_best = cb; my_best_n = i; }
                }
            }
            if constexpr (MODE == PASS_EXACT) {
                if (act && t > 0 && sub == 0) {
                    if (a.bp_rel8) ((unsigned char *)bp)[(size_t)t * W + (i & WM)] = (unsigned char)(i - pr);
                    else bp[(size_t)t * W + (i & WM)] = (u16)pr;
                }
            }
            if (!act) { lo = 0; hi = 0; }
            if (!relax) continue;
            // lowest source of this round (uniform): the workgroup-wide figure when every wave's sources are taken, a list lookup otherwise
            int a_k = low_all;
            if (kw != NW) { const int rlast = (r0 + rstep < nlist ? r0 + rstep : nlist) - 1; a_k = list_at(rlast); }
            if (clo >= chi) {                                 // (keeps sh.red / sh.agg stable until everyone has read them)
                M::barrier();
                if (agg_used) { if (tid == 0) agg_reset(&sh.agg[ab]); M::barrier(); }     // (fallback walk that took an empty wave only)
                continue;
            }
            const int need_lo = clo, need_hi = chi;              // cells this round's candidates can touch
            if constexpr (DENSE_OK) { nd_lo = need_lo < nd_lo ? need_lo : nd_lo; nd_hi = need_hi > nd_hi ? need_hi : nd_hi; }
            // the interval of initialised next-layer cells grows in 64-cell blocks where that is safe: never
            // below a_k, the lowest source of this round (lower cells may hold sources that are still unread;
            // cells >= a_k are either in registers or were not selected for expansion)
            clo &= ~63; if (clo < a_k) clo = a_k;
            chi = (chi + 63) & ~63;
            if (first) { ilo = ihi = clo; first = false; }
            const int nlo2 = clo < ilo ? clo : ilo, nhi2 = chi > ihi ? chi : ihi;
            if (nhi2 - smin > maxspan) maxspan = nhi2 - smin;
            bool last_round = false;
            if (nhi2 - smin > W) {                                           // live cells exceed the circular window
                // the bounding pre-pass may drop anything: it gives up the layer's remaining (lowest, slowest) sources
                // instead of the window, as long as the layer being built fits on its own
                if (MODE == PASS_BOUND && nhi2 - a_k <= W) last_round = true;
                else return 1;
            }
            if (chi - clo > PW) return 1;                                    // 64 sources' targets exceed the penalty buffer
            STMPC_PH(6);                // 6: round geometry (interval bookkeeping)
            if (clo < ilo) init_cells(clo, ilo);
            if (chi > ihi) init_cells(ihi, chi);
            ilo = nlo2; ihi = nhi2;
            // every cell this round can touch must have a current penalty entry
            if (!(pv_hi > pv_lo) || need_hi <= pv_lo || need_lo >= pv_hi) repen(need_lo, need_hi);
            else {
                if (need_lo < pv_lo) repen(need_lo, pv_lo);
                if (need_hi > pv_hi) repen(pv_hi, need_hi);
            }
            STMPC_BARW(2);    // B2: next-layer cells of this round are initialised
            // (everyone has read the workgroup-wide figures; the next round's atomics come after this round's B3 -- the bounding pass has
            // no B3: its next round uses the other set, which was reset a whole round ago)
            if (agg_used && tid == 0) agg_reset(&sh.agg[ab]);
            STMPC_PH(7);                // 7: cell initialisation (penalties) + B2

            auto cand = [&](int slot) -> int { return lo + sub + (slot << gsh); };      // cell of this lane's slot-th candidate
            if constexpr (MODE == PASS_BOUND) {
                // Bounding pass: any complete path is a bound, so nothing here has to reproduce the reference's arithmetic.  A cell holds
                // (fp32 accumulated cost) << 32 | key of the offering source, and ONE ds_min_u64 per candidate keeps the cheapest offer together
                // with its history -- no read-back, no first-setter stage, no barrier after the candidates (the next round's sources lie below
                // every cell this round can touch, its new cells outside the initialised interval, and its penalty entries are rewritten only
                // after its B1).  The edge cost is the single quadratic K d^2 + (C + emin) + penalty, d = distance of the candidate from the
                // source's minimiser: ~12 instructions per candidate instead of ~70.
                const float Kf = (float)nk.K;
                const float stepf = (float)(delta * (double)(1 << gsh));
                float dcur = (hi > lo) ? (float)(sval(lo + sub) - q_smin) : 0.0f;
                const u64 keyw = (u64)key;
                for (int cbase = 0; (cbase << gsh) < fan; cbase += 4) {
                    if (__ballot(cand(cbase) < hi)) {
                        float pn[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) pn[u] = __hip_atomic_load(&penf[cand(cbase + u) & PWM], __ATOMIC_RELAXED, M::SCOPE);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int n = cand(cbase + u);
                            const float dd = dcur + (float)u * stepf;
                            const float tot = __builtin_fmaf(Kf * dd, dd, q_base) + pn[u];
                            const bool ok = (n < hi) & (pn[u] >= 0.0f);
                            STMPC_PH_CAND(ok);
                            if (ok) {
                                const u64 val = ((u64)__float_as_uint(tot) << 32) | keyw;
                                if (val < my_min_tot) my_min_tot = val;
                                (void)M::min64(&cost[n & WM], val);
                            }
                        }
                    }
                    dcur += 4.0f * stepf;
                }
                STMPC_PH(8);
            } else {
            const double two_sv = 2 * sv, three_sv = 3 * sv, three_p1 = 3 * p1;
            const bool one_batch = (MODE == PASS_EXACT) && (fan <= (FANMAX << gsh));      // (workgroup-uniform: both forms of a batch have their own barriers)
            for (int cbase = 0; (cbase << gsh) < fan; cbase += FANMAX) {
                // A batch whose candidate cells do not wrap around the circular arrays (and whose lanes own one source each) addresses them as
                // one base per array plus the slot number -- immediate offsets of the LDS instructions -- instead of (cell & mask) * size per
                // access: eight address instructions less per slot.  Chosen per wave and batch; both forms execute the same barriers.
                const int c0 = cand(cbase);
                const bool fastb = USE_LDS && gsh == 0 && __ballot(((c0 & WM) + FANMAX > W) | ((c0 & PWM) + FANMAX > PW)) == 0ull;
                auto batch = [&](auto fast_tag) {
                constexpr bool FAST = decltype(fast_tag)::value;
                u64 *const cbp = &cost[c0 & WM]; double *const pbp = &pen[c0 & PWM]; unsigned *const hbp = &hist[c0 & WM];
                auto cost_at = [&](int j) -> u64 * { if constexpr (FAST) return cbp + j; else return &cost[cand(cbase + j) & WM]; };
                auto pen_at = [&](int j) -> double * { if constexpr (FAST) return pbp + j; else return &pen[cand(cbase + j) & PWM]; };
                auto hist_at = [&](int j) -> unsigned * { if constexpr (FAST) return hbp + j; else return &hist[cand(cbase + j) & WM]; };
                u64 tb[FANMAX];
                unsigned improved = 0u, tied = 0u;
                STMPC_PH_COUNT(13);
                // stage A: candidates in batches of UB so that the LDS round trips of a batch overlap
                constexpr int UB = (FANMAX % 4 == 0) ? 4 : (FANMAX % 3 == 0 ? 3 : FANMAX);   // small batches: slots beyond a wave's widest range are skipped
#pragma unroll
                for (int ub = 0; ub < FANMAX; ub += UB) {
                    if (__ballot(cand(cbase + ub) < hi)) {
                        // branch-free: lanes without a candidate at this offset compute on whatever the slot holds and
                        // then offer ~0, which a min never takes
                        double pn[UB];
#pragma unroll
                        for (int u = 0; u < UB; ++u) pn[u] = M::ldf(pen_at(ub + u));
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const int n = FAST ? c0 + (ub + u) : cand(cbase + ub + u);
                            const double sn = sval(n);
                            // st_cy.pyx:46-50 cost_with_jerk(next, s, p1, p2)
                            const double v = divk<FASTDIV>(sn - sv, dt, r_dt, zl_dt);
                            const double aa = divk<FASTDIV>(sn - two_sv + p1, dt2, r_dt2, zl_dt2);
                            const double jj = divk<FASTDIV>(sn - three_sv + three_p1 - p2, dt3, r_dt3, zl_dt3);
                            const double dv = v - p.v_des;
                            const double ec = p.v_w * (dv * dv) + p.a_w * (aa * aa) + p.j_w * (jj * jj) + pn[u];
                            double tot = C + ec;                             // st_cy.pyx:388
#ifndef STMPC_NO_ILP88
                            // second window: the workgroup is alone on its unit (two waves per SIMD), so what costs time is the latency of the
                            // dependent fp64 chain, not issue slots: evaluate every slot of a batch unconditionally (the compiler otherwise sinks
                            // the evaluation into a per-slot exec-mask region and runs the slots one after the other), four chains in flight
                            if constexpr (NWX == 88) asm volatile("" : "+v"(tot));
#endif
                            const bool ok = (n < hi) & (pn[u] >= 0.0);       // st_cy.pyx:379,383
                            tb[ub + u] = ok ? (u64)__double_as_longlong(tot) : ~0ull;
                            STMPC_PH_CAND(ok);
                            if constexpr (MODE == PASS_BOUND) { if (tb[ub + u] < my_min_tot) my_min_tot = tb[ub + u]; }
                        }
                        if (one_batch) {
#pragma unroll
                            for (int u = 0; u < UB; ++u) (void)M::min64(cost_at(ub + u), tb[ub + u]);
                        } else {
#pragma unroll
                            for (int u = 0; u < UB; ++u) {
                                const u64 old = M::min64(cost_at(ub + u), tb[ub + u]);
                                if (old > tb[ub + u]) improved |= 1u << (ub + u);
                                else if (old == tb[ub + u]) tied |= 1u << (ub + u);
                            }
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < UB; ++u) tb[ub + u] = ~0ull;
                    }
                }
                STMPC_BARW(3);    // B3: every min of this round is in
                STMPC_PH(8);            // 8: stage A (candidate costs, atomic minima) + B3
                if (one_batch) {
                    // Winner stage of a round whose candidates all sit in ONE batch (the common case: the filter leaves 7 of a source's 21): every
                    // candidate that equals the cell's value after B3 offers its key to ds_min_u32 -- no first-setter stage, no barrier before the
                    // tie stage, no results of the cost atomics.  Why a minimum suffices: the rounds of a layer take its sources in DESCENDING order
                    // and the key's high half is the source cell, so whatever an earlier round left in hist[] (or the initial ~0) is above every key
                    // of this round; whether this round lowered the cell or only met its value, the smallest key among its equal offers is the
                    // reference's (cost, predecessor) winner (st_cy.pyx:388).  (Not so across the batches of one round -- a later batch may lower a
                    // cell with a LARGER source than an earlier batch's winner -- hence the staged form below for those.)
#pragma unroll
                    for (int ub = 0; ub < FANMAX; ub += 4) {
                        if (__ballot(cand(cbase + ub) < hi)) {
                            u64 cur[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) if (ub + u < FANMAX) cur[u] = M::ld64(cost_at(ub + u));
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (ub + u < FANMAX) { if (cur[u] == tb[ub + u]) M::min32(hist_at(ub + u), key); }
                        }
                    }
                    STMPC_PH(9);
                    return;
                }
                // stage B: the unique first setter of a cell's final value records the predecessor.
                // Read-backs are issued in groups of 4 (unconditionally) so their LDS latencies overlap.
#pragma unroll
                for (int ub = 0; ub < FANMAX; ub += 4) {
                    if (__ballot(((improved >> ub) & 0xFu) != 0u)) {
                        u64 cur[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (ub + u < FANMAX) cur[u] = M::ld64(cost_at(ub + u));
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (ub + u < FANMAX) { if (((improved >> (ub + u)) & 1u) && cur[u] == tb[ub + u]) M::st32(hist_at(ub + u), key); }
                    }
                }
                STMPC_BARW(4);    // B4
                STMPC_PH(9);            // 9: stage B (first setter writes the predecessor) + B4
                if constexpr (MODE == PASS_EXACT) {
                    // stage C: equal total cost -> the smaller predecessor index wins (heap tuple order, st_cy.pyx:388)
                    if (__ballot(tied != 0u)) {
#pragma unroll
                        for (int u = 0; u < FANMAX; ++u) {
                            if ((tied >> u) & 1u) {
                                if (M::ld64(cost_at(u)) == tb[u]) M::min32(hist_at(u), key);
                            }
                        }
                    }
                }
                STMPC_PH(10);           // 10: stage C (tie repair)
                };
                if (fastb) batch(std::true_type{}); else batch(std::false_type{});
            }
            }
            if (last_round) break;
        }

        // ---- end of layer.  PASS_EXACT needs no barrier here: the next scan only reads cost[] (final since the
        // last round's B3) and its own S1 orders everything else.  PASS_BOUND publishes the cheapest candidate.
        if constexpr (MODE == PASS_BOUND) {
            my_min_tot = wave_min_u64(my_min_tot);
            if (lane == 0) sh.min_tot[wave] = my_min_tot;
            STMPC_BARW(5);   // S2
            u64 mt = ~0ull;
            for (int w = 0; w < NW; ++w) { const u64 m_ = sh.min_tot[w]; mt = m_ < mt ? m_ : mt; }
            lmin = mt;
            if (mt < INF_BITS) { out.best_t = t + 1; out.best_bits = mt; }
        }
        STMPC_PH(11);                   // 11: end of layer
        if constexpr (DENSE_OK) {
            if (dense) {                // the layer's cheapest node per wave (a listed layer's scan leaves it there)
                const u64 wb = wave_min_u64(my_best);
                const int wbn = wave_min_i(my_best == wb ? my_best_n : 0x7fffffff);
                if (lane == 0) { sh.best_bits[t & 1][wave] = wb; sh.best_n[t & 1][wave] = wbn; }
                d_lo = nd_lo; d_hi = nd_hi; nd_lo = 0x7fffffff; nd_hi = 0;
            }
        }
        if (!relax) break;
        if (first) { wlo = 0; whi = 0; d_lo = 0; d_hi = 0; } else { wlo = ilo; whi = ihi; }
    }
    M::barrier();
    if constexpr (DENSE_OK) {
        if (dense) {
            // node count over the workgroup; and a last layer whose span held candidates' cells but no reached node (all of them blocked) is not
            // the deepest layer: the other parity still holds the layer before it
            if (tid == 0) sh.nlist = 0;
            M::barrier();
            if (lane == 0 && wave_nodes) atomicAdd(&sh.nlist, wave_nodes);
            M::barrier();
            total_nodes = sh.nlist;
            if (last_t > 0) {
                u64 any = ~0ull;
                for (int w = 0; w < NW; ++w) { const u64 b_ = sh.best_bits[last_t & 1][w]; any = b_ < any ? b_ : any; }
                if (any == ~0ull) last_t -= 1;
            }
            M::barrier();
        }
    }
    STMPC_PH_FLUSH(MODE);
    out.vmin_bits = ~0ull;
    if constexpr (MODE == PASS_EXACT && !GRID) {
        // A bounded pass that built the last layer and found every node of it above the bound: their cheapest (one sweep over the layer's
        // cells, outside every hot loop) is the bound of the repeat (solve_episode) -- never below the reference's terminal, and equal to it
        // whenever the reference's path was cut in its last step only, which is the usual way for a bound 0.00-0.2 % too low to fail.
        if (died_t == H - 1 && ubits < INF_BITS && whi > wlo) {
            u64 m = ~0ull;
            for (int n = wlo + tid; n < whi; n += per) { const u64 c_ = M::ld64(&cost[n & WM]); m = c_ < m ? c_ : m; }
            m = wave_min_u64(m);
            if (lane == 0) sh.min_tot[wave] = m;
            M::barrier();
            u64 mt = ~0ull;
            for (int w = 0; w < NW; ++w) { const u64 m_ = sh.min_tot[w]; mt = m_ < mt ? m_ : mt; }
            if (mt < INF_BITS) out.vmin_bits = mt;
            M::barrier();
        }
    }
    if constexpr (MODE == PASS_EXACT) {
        // the terminal of the search (st_cy.pyx:365-369): cheapest node of the deepest non-empty layer, smallest cell among equals
        if (last_t >= 0) {
            u64 bb = ~0ull; int bn = 0x7fffffff;
            for (int w = 0; w < NW; ++w) {
                const u64 b_ = sh.best_bits[last_t & 1][w]; const int n_ = sh.best_n[last_t & 1][w];
                if (b_ < bb || (b_ == bb && n_ < bn)) { bb = b_; bn = n_; }
            }
            out.best_t = last_t; out.best_n = bn; out.best_bits = bb;
        }
    }
    if constexpr (MODE == PASS_BOUND) {
        // the bound: the path's cost as accumulated in single precision (error below 1e-5 relative: <= 2^-23 per operation, ~4 operations
        // per layer, H layers), inflated beyond that -- any value is safe, the exact pass re-checks
        const double ub = (double)__uint_as_float((unsigned)(out.best_bits >> 32)) * a.bound_infl;
        out.best_bits = (u64)__double_as_longlong(ub);
    }
    out.pruned = (sh.flags & 1) != 0;
    out.nodes = total_nodes;
    out.maxspan = maxspan;
    return 0;
}

// Guided bounding attempt of a four-wave workgroup, dense: lane j of the workgroup IS cell (guide cell of the layer) - tube_w + j.  The tube around the
// unobstructed optimum holds 2 * tube_w + 1 <= 256 cells per layer, nearly all of them reached, so the general pass's machinery -- compacting the
// reached cells into a list, rounds of sources, the interval bookkeeping of an in-place window -- buys nothing here and costs most of the pass's
// instructions.  Two 256-cell layers ping-pong in LDS, every lane relaxes its own cell into the next layer with the same arithmetic, filters and
// packed (single-precision cost, history) cells as dp_pass<PASS_BOUND> under a tube, and a layer costs two barriers.  Any complete path is a valid
// bound; what this pass must share with the general one is the reference's one-history-per-cell competition inside the tube, which is what makes
// its bound the reference's own cost in most episodes.
// cells: [2][256] u64, penf: [256] float (LDS).  sh.path holds the guide.  Returns true with out.best_bits / out.nodes set if a complete path was found.
template <bool FASTDIV, bool S1GEN>
__device__ __forceinline__ bool tube_pass(const SolveArgs &a, const Ep &ep, WgShared &sh, u64 *cells, float *penf, PassOut &out) {
    typedef Mem<true> M;
    const DevP &p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, S = ep.S, e = __builtin_amdgcn_readfirstlane(ep.e), w = a.tube_w, TW = 2 * w + 1;
    const double start_s = ep.start_s, delta = ep.delta, dt = p.dt;
    const double r_dt = ep.r_dt, r_delta = ep.r_delta, zl_dt = a.zl_dt;
    const bool s1_plain = ep.s1_plain;
    auto sval = [&](int n) -> double {
        double v = start_s + (double)n * delta;
        if constexpr (S1GEN) { if (!s1_plain) { if (n == 1) v = ep.s1; } }
        return v;
    };
    const EdgeQuad nk = edge_quad(p);
    const double K = nk.K, invK = nk.invK;
    const bool quad = nk.ok;
    const float Kf = (float)K, stepf = (float)delta;
    const float bandf = (float)a.band;
    const double rad = quad ? (double)__builtin_amdgcn_sqrtf((float)(a.band * invK)) : 0.0;       // (the band is constant here: a tube layer never exceeds band_cap nodes)
    u64 *cur = cells, *nxt = cells + 256;
    M::barrier();                        // previous users of the arrays are done
    M::st64(&cur[tid], (tid == w) ? 0ull : INF_BITS);          // layer 0: cell 0 (the guide's cell of layer 0) at cost 0, no history
    if (tid == 0) sh.nlist = 0;
    u64 lmin = 0ull;
    int wn = 0;
    bool complete = true;
    for (int t = 0; t < H - 1; ++t) {
        const int base = sh.path[t] - w, base1 = sh.path[t + 1] - w;
        // ---- next layer: my cell's penalty, "not reached"
        {
            const int n1 = base1 + tid;
            float pf = -1.0f;
            if (tid < TW && n1 >= 0 && n1 < S) {
                const size_t row = (size_t)e * H + (t + 1);
                int nact = as_const(a.tab.nact)[row];
                stmpc_cdouble *cedge = as_const(a.tab.edge) + row * a.Kmax * 2;
                stmpc_cint *cwin = as_const(a.tab.win) + row * a.Kmax * 2;
                const double sn = sval(n1);
                double d = 1e10;
                bool blocked = false;
                for (int c0 = 0; c0 < nact; c0 += 4) {           // (rows in groups of four, loads first; a row taken twice changes nothing)
                    double vf[4], vb[4]; int w0[4], w1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + u < nact ? c0 + u : nact - 1;
                        vf[u] = cedge[2 * c]; vb[u] = cedge[2 * c + 1]; w0[u] = cwin[2 * c]; w1[u] = cwin[2 * c + 1];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        d = __builtin_fmin(d, fabs(sn - vf[u]));
                        d = __builtin_fmin(d, fabs(sn - vb[u]));
                        blocked |= (n1 >= w0[u]) & (n1 < w1[u]);
                    }
                }
                if (!blocked && !(d < p.min_allowed)) pf = (float)p.d_w * __builtin_amdgcn_rcpf((float)d);     // (the penalty zone counts as blocked in this attempt)
            }
            __hip_atomic_store(&penf[tid], pf, __ATOMIC_RELAXED, M::SCOPE);
            M::st64(&nxt[tid], INF_BITS);
        }
        M::barrier();                    // A: the next layer is initialised (and everyone is past the previous layer's candidates)
        // ---- my cell of layer t
        u64 my_min = ~0ull;
        {
            const int i = base + tid;
            const u64 cb = M::ld64(&cur[tid]);
            const float Cf = __uint_as_float((unsigned)(cb >> 32));
            const float lim = __uint_as_float((unsigned)(lmin >> 32)) + bandf;
            const bool act = tid < TW && cb < INF_BITS && Cf <= lim;
            wn += __popcll(__ballot(act));
            int lo = 0, hi = 0;
            double q_smin = 0.0; float q_base = 0.0f;
            unsigned key = 0u;
            if (act) {
                const unsigned h = (unsigned)cb;
                const double sv = sval(i);
                double p1, p2;
                if (t == 0) { p1 = ep.est_prev; p2 = ep.est_second; }
                else {
                    const int pr = (int)(h >> 16), pp = (int)(h & 0xFFFFu);
                    p1 = sval(pr);
                    p2 = (t == 1) ? ep.est_prev : sval(pp);
                    key = ((unsigned)i << 16) | (unsigned)pr;
                }
                // st_cy.pyx:65-93, as in dp_pass
                const double prev_v = divk<FASTDIV>(p1 - p2, dt, r_dt, zl_dt);
                const double v = divk<FASTDIV>(sv - p1, dt, r_dt, zl_dt);
                const double acc = divk<FASTDIV>(v - prev_v, dt, r_dt, zl_dt);
                const double min_a = dmax1(acc + p.j_min * dt, p.a_min);
                const double max_a = dmin1(acc + p.j_max * dt, p.a_max);
                const double min_v = dmax1(v + min_a * dt, 0.0);
                const double max_v = dmin1(v + max_a * dt, p.v_max);
                const double min_s = sv + min_v * dt, max_s = sv + max_v * dt;
                const double x = divc<FASTDIV>(min_s - start_s, delta, r_delta);
                int mi = (int)x;
                const int ma = (int)divc<FASTDIV>(max_s - start_s, delta, r_delta);
                if (mi < x) mi += 1;
                lo = mi; hi = ma + 1;
                if (hi > S) hi = S;
                if (lo < i) lo = i;
                if (quad && hi > lo) {
                    double m_, emin;
                    edge_quad_source(nk, sv - p1, p1 - p2, m_, emin);
                    q_smin = sv + m_;
                    q_base = Cf + (float)emin;
                    const int nlo_ = i + (int)cell_lo(ceil((m_ - rad) * r_delta));
                    const int nhi_ = i + (int)cell_hi(floor((m_ + rad) * r_delta)) + 1;
                    const int a_ = nlo_ > lo ? nlo_ : lo, b_ = nhi_ < hi ? nhi_ : hi;
                    if (a_ < b_) { lo = a_; hi = b_; }
                    else if (nlo_ >= hi) lo = hi - 1;
                    else hi = lo + 1;
                } else { q_smin = sv; q_base = Cf; }
                lo = lo > base1 ? lo : base1; hi = hi < base1 + TW ? hi : base1 + TW;      // the next layer's tube
                if (lo >= hi) { lo = 0; hi = 0; }
            }
            float dcur = (hi > lo) ? (float)(sval(lo) - q_smin) : 0.0f;
            const u64 keyw = (u64)key;
            for (int k = 0; __ballot(lo + k < hi) != 0ull; k += 4) {
                float pn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) pn[u] = __hip_atomic_load(&penf[(lo + k + u - base1) & 255], __ATOMIC_RELAXED, M::SCOPE);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int n = lo + k + u;
                    const float dd = dcur + (float)u * stepf;
                    const float tot = __builtin_fmaf(Kf * dd, dd, q_base) + pn[u];
                    if ((n < hi) & (pn[u] >= 0.0f)) {
                        const u64 val = ((u64)__float_as_uint(tot) << 32) | keyw;
                        my_min = val < my_min ? val : my_min;
                        (void)M::min64(&nxt[n - base1], val);
                    }
                }
                dcur += 4.0f * stepf;
            }
        }
        my_min = wave_min_u64(my_min);
        if (lane == 0) sh.min_tot[wave] = my_min;
        M::barrier();                    // B: every offer of this layer is in
        u64 mt = sh.min_tot[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) { const u64 m_ = sh.min_tot[k]; mt = m_ < mt ? m_ : mt; }
        if (mt >= INF_BITS) { complete = false; break; }          // nothing reached the next layer (workgroup-uniform)
        lmin = mt;
        u64 *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (lane == 0) atomicAdd(&sh.nlist, wn);
    M::barrier();
    out.nodes = sh.nlist;
    out.best_t = complete ? H - 1 : 0;
    if (complete) out.best_bits = (u64)__double_as_longlong((double)__uint_as_float((unsigned)(lmin >> 32)) * a.bound_infl);
    M::barrier();                        // (sh.nlist, the arrays: free for the next user)
    return complete;
}

// The ordinary bounding attempts of a four-wave workgroup, dense: the layer being expanded and the layer being built are two windows of
// STMPC_BAND_W consecutive cells in LDS (ping-pong), lane tid owns cells tid, tid + 256, ... of each, and a layer costs three barriers -- no list
// of reached cells, no rounds, no interval bookkeeping of an in-place window (dp_pass<PASS_BOUND> pays for those with most of its instructions:
// the band-limited layers hold a few hundred nodes spread over up to a thousand cells).  Same selection (cost within the band of the layer's
// cheapest node, the band steered towards band_cap nodes), same per-source range, quadratic filter and packed single-precision candidates as
// dp_pass, so the bound is the same wherever dp_pass would not have dropped sources; a layer pair that does not fit the window returns 2 and
// the caller runs the general pass instead.
// Returns 0: complete path found (out.best_bits / out.nodes), 1: no complete path under this band, 2: window too small.
#define STMPC_BAND_W 1536
#define STMPC_BAND_SLOTS (STMPC_BAND_W / 256)
template <bool FASTDIV, bool S1GEN>
__device__ __forceinline__ int band_pass(const SolveArgs &a, const Ep &ep, WgShared &sh, unsigned char *lds /* >= 30 KB */, const double band, const bool hardsoft, PassOut &out) {
    typedef Mem<true> M;
    const DevP &p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, S = ep.S, e = __builtin_amdgcn_readfirstlane(ep.e);      // (uniform by construction; said so that the vehicle rows are fetched by scalar loads)
    const double start_s = ep.start_s, delta = ep.delta, dt = p.dt;
    const double r_dt = ep.r_dt, r_delta = ep.r_delta, zl_dt = a.zl_dt;
    const bool s1_plain = ep.s1_plain;
    auto sval = [&](int n) -> double {
        double v = start_s + (double)n * delta;
        if constexpr (S1GEN) { if (!s1_plain) { if (n == 1) v = ep.s1; } }
        return v;
    };
    const EdgeQuad nk = edge_quad(p);
    const double K = nk.K, invK = nk.invK;
    const bool quad = nk.ok;
    const float Kf = (float)K, stepf = (float)delta;
    const int mshift = a.maxshift - 62;          // every target lies below (its source) + mshift (SolveArgs::maxshift carries 66 cells of alignment slack)
    u64 *cur = (u64 *)lds, *nxt = cur + STMPC_BAND_W;
    float *penf = (float *)(nxt + STMPC_BAND_W);
    M::barrier();                        // previous users of the arrays are done
#pragma unroll 1
    for (int s_ = 0; s_ < STMPC_BAND_SLOTS; ++s_) M::st64(&cur[s_ * 256 + tid], (s_ == 0 && tid == 0) ? 0ull : INF_BITS);     // layer 0: cell 0 at cost 0
    int base = 0;
    u64 lmin = 0ull;
    double bandt = band;
    int total_nodes = 0;
    int rc = 0;
    M::barrier();
    for (int t = 0; t < H - 1; ++t) {
        // ---- which of my cells are expanded; the layer's extent and node count
        const float lim = __uint_as_float((unsigned)(lmin >> 32)) + (float)bandt;
        unsigned am = 0u;
        int my_lo = 0x7fffffff, my_hi = -1, cnt = 0;
#pragma unroll
        for (int s_ = 0; s_ < STMPC_BAND_SLOTS; ++s_) {
            const u64 cb = M::ld64(&cur[s_ * 256 + tid]);
            const bool act = cb < INF_BITS && __uint_as_float((unsigned)(cb >> 32)) <= lim;
            if (act) { am |= 1u << s_; const int i = base + s_ * 256 + tid; my_lo = i < my_lo ? i : my_lo; my_hi = i > my_hi ? i : my_hi; }
            cnt += __popcll(__ballot(act));
        }
        my_lo = wave_min_i(my_lo); my_hi = wave_max_i(my_hi);
        if (lane == 0) { sh.red[wave * 4 + 0] = my_lo; sh.red[wave * 4 + 1] = my_hi; sh.cnt[wave] = cnt; }
        M::barrier();                    // C
        int slo = 0x7fffffff, shi = -1, nlist = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int l_ = sh.red[k * 4 + 0], h_ = sh.red[k * 4 + 1]; slo = l_ < slo ? l_ : slo; shi = h_ > shi ? h_ : shi; nlist += sh.cnt[k]; }
        total_nodes += nlist;
        if (a.band_cap > 0 && nlist > 0) {                  // (dp_pass's steering of the band)
            const double f = (double)a.band_cap / (double)nlist;
            bandt = bandt * (f < 1.0 ? f : sqrt(f));
            bandt = bandt > band ? band : (bandt < 0.05 * band ? 0.05 * band : bandt);
        }
        if (nlist == 0) { rc = 1; break; }
        int thi = shi + mshift; thi = thi > S ? S : thi;
        if (thi - slo > STMPC_BAND_W) { rc = 2; break; }    // (workgroup-uniform)
        const int base1 = slo;
        const double rad = (double)__builtin_amdgcn_sqrtf((float)(bandt * invK));
        // ---- next layer: "not reached" everywhere, penalties where a target can land
        {
            const size_t row = (size_t)e * H + (t + 1);
            const int nact = as_const(a.tab.nact)[row];
            stmpc_cdouble *cedge = as_const(a.tab.edge) + row * a.Kmax * 2;
            stmpc_cint *cwin = as_const(a.tab.win) + row * a.Kmax * 2;
#pragma unroll 1
            for (int s_ = 0; s_ < STMPC_BAND_SLOTS; ++s_) {
                const int j = s_ * 256 + tid, n1 = base1 + j;
                M::st64(&nxt[j], INF_BITS);
                if (n1 < thi) {
                    const double sn = sval(n1);
                    double d = 1e10;
                    bool blocked = false;
                    for (int c0 = 0; c0 < nact; c0 += 4) {       // (rows in groups of four, loads first; a row taken twice changes nothing)
                        double vf[4], vb[4]; int w0[4], w1[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = c0 + u < nact ? c0 + u : nact - 1;
                            vf[u] = cedge[2 * c]; vb[u] = cedge[2 * c + 1]; w0[u] = cwin[2 * c]; w1[u] = cwin[2 * c + 1];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            d = __builtin_fmin(d, fabs(sn - vf[u]));
                            d = __builtin_fmin(d, fabs(sn - vb[u]));
                            blocked |= (n1 >= w0[u]) & (n1 < w1[u]);
                        }
                    }
                    const bool close = d < p.min_allowed;
                    if (hardsoft && close) blocked = true;
                    const float den = (float)(close ? dmax_py(d, 1.0) : d);
                    const float pf = (float)p.d_w * ((close ? 1000000.0f : 1.0f) * __builtin_amdgcn_rcpf(den));
                    __hip_atomic_store(&penf[j], blocked ? -1.0f : pf, __ATOMIC_RELAXED, M::SCOPE);
                }
            }
        }
        M::barrier();                    // A: the next layer is initialised
        // ---- my cells of layer t
        u64 my_min = ~0ull;
#pragma unroll 1
        for (int s_ = 0; s_ < STMPC_BAND_SLOTS; ++s_) {
            const bool act = (am >> s_) & 1u;
            if (__ballot(act) == 0ull) continue;
            const int i = base + s_ * 256 + tid;
            int lo = 0, hi = 0;
            double q_smin = 0.0; float q_base = 0.0f;
            unsigned key = 0u;
            if (act) {
                const u64 cb = M::ld64(&cur[s_ * 256 + tid]);
                const float Cf = __uint_as_float((unsigned)(cb >> 32));
                const unsigned h = (unsigned)cb;
                const double sv = sval(i);
                double p1, p2;
                if (t == 0) { p1 = ep.est_prev; p2 = ep.est_second; }
                else {
                    const int pr = (int)(h >> 16), pp = (int)(h & 0xFFFFu);
                    p1 = sval(pr);
                    p2 = (t == 1) ? ep.est_prev : sval(pp);
                    key = ((unsigned)i << 16) | (unsigned)pr;
                }
                // st_cy.pyx:65-93, as in dp_pass
                const double prev_v = divk<FASTDIV>(p1 - p2, dt, r_dt, zl_dt);
                const double v = divk<FASTDIV>(sv - p1, dt, r_dt, zl_dt);
                const double acc = divk<FASTDIV>(v - prev_v, dt, r_dt, zl_dt);
                const double min_a = dmax1(acc + p.j_min * dt, p.a_min);
                const double max_a = dmin1(acc + p.j_max * dt, p.a_max);
                const double min_v = dmax1(v + min_a * dt, 0.0);
                const double max_v = dmin1(v + max_a * dt, p.v_max);
                const double min_s = sv + min_v * dt, max_s = sv + max_v * dt;
                const double x = divc<FASTDIV>(min_s - start_s, delta, r_delta);
                int mi = (int)x;
                const int ma = (int)divc<FASTDIV>(max_s - start_s, delta, r_delta);
                if (mi < x) mi += 1;
                lo = mi; hi = ma + 1;
                if (hi > S) hi = S;
                if (lo < i) lo = i;
                if (quad && hi > lo) {
                    double m_, emin;
                    edge_quad_source(nk, sv - p1, p1 - p2, m_, emin);
                    q_smin = sv + m_;
                    q_base = Cf + (float)emin;
                    const int nlo_ = i + (int)cell_lo(ceil((m_ - rad) * r_delta));
                    const int nhi_ = i + (int)cell_hi(floor((m_ + rad) * r_delta)) + 1;
                    const int a_ = nlo_ > lo ? nlo_ : lo, b_ = nhi_ < hi ? nhi_ : hi;
                    if (a_ < b_) { lo = a_; hi = b_; }
                    else if (nlo_ >= hi) lo = hi - 1;
                    else hi = lo + 1;
                } else { q_smin = sv; q_base = Cf; }
                if (hi > thi) hi = thi;              // (cannot happen: thi bounds every target; keeps the window safe)
                if (lo >= hi) { lo = 0; hi = 0; }
            }
            float dcur = (hi > lo) ? (float)(sval(lo) - q_smin) : 0.0f;
            const u64 keyw = (u64)key;
            for (int k = 0; __ballot(lo + k < hi) != 0ull; k += 4) {
                float pn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { int jx = lo + k + u - base1; jx = jx < 0 ? 0 : (jx >= STMPC_BAND_W ? STMPC_BAND_W - 1 : jx); pn[u] = __hip_atomic_load(&penf[jx], __ATOMIC_RELAXED, M::SCOPE); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int n = lo + k + u;
                    const float dd = dcur + (float)u * stepf;
                    const float tot = __builtin_fmaf(Kf * dd, dd, q_base) + pn[u];
                    if ((n < hi) & (pn[u] >= 0.0f)) {
                        const u64 val = ((u64)__float_as_uint(tot) << 32) | keyw;
                        my_min = val < my_min ? val : my_min;
                        (void)M::min64(&nxt[n - base1], val);
                    }
                }
                dcur += 4.0f * stepf;
            }
        }
        my_min = wave_min_u64(my_min);
        if (lane == 0) sh.min_tot[wave] = my_min;
        M::barrier();                    // B: every offer of this layer is in
        u64 mt = sh.min_tot[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) { const u64 m_ = sh.min_tot[k]; mt = m_ < mt ? m_ : mt; }
        if (mt >= INF_BITS) { rc = 1; break; }              // nothing reached the next layer
        lmin = mt;
        u64 *tmp = cur; cur = nxt; nxt = tmp;
        base = base1;
    }
    out.nodes = total_nodes;
    out.best_t = rc == 0 ? H - 1 : 0;
    if (rc == 0) out.best_bits = (u64)__double_as_longlong((double)__uint_as_float((unsigned)(lmin >> 32)) * a.bound_infl);
    M::barrier();                        // the arrays are free for the next user
    return rc;
}

// Solve one episode with one workgroup.  Returns 0 ok, 1 window overflow (workgroup-uniform).
template <bool USE_LDS, bool GRID, bool FASTDIV, int KT, int FANMAX, bool S1GEN, int RES = 0, int NWX = STMPC_MAXWAVES>
__device__ int solve_episode(const SolveArgs &a, int e, int slot, WgShared &sh, u64 *cost, unsigned *hist,
                             double *pen, u16 *list, int *chunk_cnt, double *ltab_e, int *ltab_w, int *ltab_n, const int phase) {
    const DevP &p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int W = WgShape<NWX>::fixed ? WgShape<NWX>::W : a.W, WM = W - 1;
    const int H = p.H;
    Ep ep;
    if constexpr (GRID) {
        ep.start_s = a.s_values[0]; ep.s1 = a.s_values[1]; ep.delta = ep.s1 - ep.start_s;      // st_cy.pyx:318,320
        ep.v0 = a.v0_grid; ep.a0 = a.a0_grid; ep.S = a.S_grid;
    } else {
        ep.start_s = a.ego[(size_t)e * 5 + 4]; ep.v0 = a.ego[(size_t)e * 5 + 2]; ep.a0 = a.ego[(size_t)e * 5 + 3];
        ep.s1 = ep.start_s + p.ds; ep.delta = ep.s1 - ep.start_s; ep.S = a.tab.num_s[e];
    }
    ep.e = e;
    const double dt = p.dt;
    ep.r_dt = 1.0 / dt; ep.r_dt2 = 1.0 / p.dt2; ep.r_dt3 = 1.0 / p.dt3; ep.r_delta = 1.0 / ep.delta;
    // st_cy.pyx:329-330 virtual history
    ep.est_prev = ep.start_s - ep.v0 * dt;
    ep.est_second = ep.est_prev - dt * (ep.v0 - ep.a0 * dt);
    ep.s1_plain = (ep.start_s + 1.0 * ep.delta == ep.s1);
    if constexpr (!S1GEN) {
        // this variant evaluates s_values[n] as start + n*delta for every n; the (very rare) episode whose second
        // lattice point start+step differs from that goes to the last tier, which is compiled with the general form
        if ((!ep.s1_plain || a.force_general) && phase != 1) return 1;      // (a bound-only phase tolerates the ulp-level difference: bounds are re-checked)
    }
    // (per resident workgroup in every tier; element size 1 or 2 bytes, see SolveArgs::bp_rel8.  A first-window search that checkpoints copies its rows to the pool.)
    ep.bp = (u16 *)((unsigned char *)a.bp + (size_t)slot * H * W * (a.bp_rel8 ? 1 : 2));
    ep.pool = 0;
    const double start_s = ep.start_s, delta = ep.delta;
    const int S = ep.S;
    auto sval = [&](int n) -> double {
        if constexpr (GRID) return a.s_values[n];
        else {
            double v = start_s + (double)n * delta;
            if (!ep.s1_plain) { if (n == 1) v = ep.s1; }
            return v;
        }
    };

    if constexpr (!GRID && KT > 0) {
        // stage the episode's table of obstructing vehicles (k_predict output) in LDS: one coalesced sweep
        // instead of two dependent global round trips per layer and pass
        __syncthreads();
        const size_t rowbase = (size_t)e * H;
        for (int x = tid; x < H * KT * 2; x += blockDim.x) {
            const int t_ = x / (KT * 2), r_ = x - t_ * (KT * 2);
            const bool in = r_ < a.Kmax * 2;
            ltab_e[x] = in ? a.tab.edge[(rowbase + t_) * a.Kmax * 2 + r_] : 0.0;
            ltab_w[x] = in ? a.tab.win[(rowbase + t_) * a.Kmax * 2 + r_] : 0;
        }
        for (int x = tid; x < H; x += blockDim.x) ltab_n[x] = a.tab.nact[rowbase + x];
        __syncthreads();
    }
    PassOut out;
    u64 ubits = INF_BITS;
    bool have_bound = false;
    if constexpr (!GRID) {
        if (a.prune && (a.tier > 0 || phase == 2)) {   // bounded earlier (bound-only phase, or the tier it overflowed)
            const u64 ub = a.ubound[e];
            if (ub != 0ull) { ubits = ub; have_bound = true; }
        }
        if (a.prune && !have_bound) {
            // upper bound of the terminal cost from a cheap banded search (two attempts), see dp_pass
            // (one call site in a loop: a second inlined copy of the pass would add a third to the kernel's code size)
            int rc = 0, bn = 0;
            // attempt -1 (guided): only where the unobstructed optimum is clear of the traffic, inside a tube around it
            bool guided = false;
            if (a.guide != nullptr) {                         // (written by k_predict: cells per layer, first entry 0xffff = not usable)
                __syncthreads();
                if (tid < H) sh.path[tid] = (int)a.guide[(size_t)e * H + tid];
                __syncthreads();
                guided = sh.path[0] != 0xffff;
            }
            if constexpr (USE_LDS && NWX == 4 && FANMAX != 9) {
                // (the standard first window: four waves = 256 lanes, one per cell of the tube; not compiled into the narrow-lattice kernels, whose
                // search is not bounded by default and whose register allocation has no room to spare)
                if (guided && a.tube_dense && 2 * a.tube_w + 1 <= 256) {
                    const bool found = tube_pass<FASTDIV, S1GEN>(a, ep, sh, cost, (float *)pen, out);
                    bn += out.nodes;
                    if (found) { ubits = out.best_bits; if (tid == 0) atomicAdd(&a.counters[STMPC_CNT_GUIDED], 1u); }
                    guided = false;
                }
            }
            for (int att = guided ? -1 : 0; att < 2 && ubits == INF_BITS; ++att) {
                if constexpr (USE_LDS && NWX == 4 && FANMAX != 9) {
                    if (att >= 0 && a.band_dense) {
                        const int rb = band_pass<FASTDIV, S1GEN>(a, ep, sh, (unsigned char *)cost, att == 0 ? a.band : a.band * a.band2_mult, att == 0, out);
                        bn += out.nodes;
                        if (rb == 0) { ubits = out.best_bits; break; }
                        if (rb == 1) continue;             // no complete path under this band: next attempt
                        // (rb == 2: the layers do not fit the dense window: the general pass takes this attempt)
                    }
                }
                rc = dp_pass<USE_LDS, GRID, FASTDIV, KT, PASS_BOUND, FANMAX, S1GEN, 0, NWX>(a, ep, sh, cost, hist, pen, list, chunk_cnt, ltab_e, ltab_w, ltab_n, INF_BITS,
                                                                                   att <= 0 ? a.band : a.band * a.band2_mult, att <= 0, out, 0, att < 0 ? a.tube_w : 0);
                bn += out.nodes;
                if (rc != 0) { if (att < 0) { rc = 0; continue; } break; }      // (a tube that does not fit the window: go on with the ordinary attempts)
                if (out.best_t == H - 1) { ubits = out.best_bits; if (att < 0 && tid == 0) atomicAdd(&a.counters[STMPC_CNT_GUIDED], 1u); break; }
            }
            if (tid == 0) atomicAdd(&a.counters[STMPC_CNT_NODES_BOUND], (unsigned)bn);
            if (rc != 0 && !a.last_tier) {
                if (phase == 1 && a.split && tid == 0)        // tell this episode's exact task that the episode has moved on
                    __hip_atomic_store(&a.proxy[e], STMPC_PROXY_MOVED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                return rc;
            }
            if (phase == 1) {                                 // bound-only task: publish the bound and a work estimate
                if (tid == 0) {
                    a.proxy[e] = (ubits == INF_BITS) ? 0x3fffffffu : (unsigned)bn;     // unbounded episodes are the heaviest
                    __hip_atomic_store(&a.ubound[e], (ubits == 0ull) ? 1ull : ubits, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                return 0;
            }
        }
    }
    int t_res = 0;                      // layer the successful pass started from (checkpoint of the first window) or 0
    if constexpr (RES == 2) { if (have_bound) { const int rt = a.resume_t[e]; t_res = rt & 0xff; ep.pool = rt >> 8; } }
    for (int attempt = 0;; ++attempt) {
        if (attempt > 0) t_res = 0;     // a relaxed bound invalidates the checkpoint: start over
        int rc = dp_pass<USE_LDS, GRID, FASTDIV, KT, PASS_EXACT, FANMAX, S1GEN, RES, NWX>(a, ep, sh, cost, hist, pen, list, chunk_cnt, ltab_e, ltab_w, ltab_n, ubits, 0.0, false, out, t_res);
        if (rc != 0) {
            if constexpr (!GRID) {
                if (tid == 0 && a.ubound) a.ubound[e] = (ubits == 0ull) ? 1ull : ubits;    // 0 is reserved for "unknown"
                if constexpr (RES == 1) { if (tid == 0) a.resume_t[e] = (rc == 2) ? (out.best_t | (out.pool << 8)) : 0; }   // rc 2: layer out.best_t is saved in pool entry out.pool
                __threadfence();        // checkpoint, back-pointers, bound: visible before the episode is queued
                __syncthreads();
                // Longest first in the next window: what is left of this search is roughly (layers left) x (nodes of the saved layer); above
                // prio_thr the episode joins the queue's front class (bit 2 of the return code), which the consumers drain first.  The step
                // ends with the second window's last episode, so the long ones must not start last.
                if constexpr (RES == 1) {
                    if (rc == 2 && a.prio_thr > 0) {
                        const int left = H - 1 - out.best_t;
                        const int key = a.prio_mode == 1 ? left * 1000 : (a.prio_mode == 2 ? out.nodes * 20 : (a.prio_mode == 3 ? left * left * out.nodes / 16 : left * out.nodes));
                        if (key > a.prio_thr) return 6;
                    }
                }
            }
            return rc;
        }
        if (tid == 0) atomicAdd(&a.counters[STMPC_CNT_NODES_EXACT], (unsigned)out.nodes);
        if (out.best_t == H - 1 || !out.pruned) break;
        // the bound was below the reference's terminal cost (its search is not globally optimal): relax it
        // (growth factors: SolveArgs::retry_mult)
        if (out.vmin_bits != ~0ull && out.vmin_bits > ubits) ubits = out.vmin_bits;      // a complete path's exact cost, >= the reference's terminal: this repeat cannot fail (dp_pass)
        else if (attempt >= 3) ubits = INF_BITS;
        else ubits = (u64)__double_as_longlong(__longlong_as_double((long long)ubits) * (attempt == 0 ? a.retry_mult[0] : (attempt == 1 ? a.retry_mult[1] : a.retry_mult[2])));
        if (tid == 0) atomicAdd(&a.counters[STMPC_CNT_RETRY], 1u);
        if constexpr (!GRID && RES == 1) {
            // A search that has failed retry_move times already is one of the step's longest chains (three passes, each larger than the last):
            // its next pass goes to the second window, whose workgroup has a compute unit to itself and runs an episode about twice as fast.
            if (a.retry_move > 0 && attempt + 1 >= a.retry_move && !a.last_tier) {
                if (tid == 0) { a.ubound[e] = (ubits == 0ull) ? 1ull : ubits; a.resume_t[e] = 0; }
                __threadfence();
                __syncthreads();
                return 1;
            }
        }
    }
    const int best_t = out.best_t, best_n = out.best_n;
    const u64 best_bits = out.best_bits;
    u16 *bp = ep.bp;

    // back-track (st_cy.pyx:391-398); the back-pointers were written by all waves
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    if (tid == 0) {
        int n = best_n;
        for (int t = best_t; t > 0; --t) {
            sh.path[t] = n;
            if (a.bp_rel8) {
                const unsigned char *b8 = (RES != 2 || t >= t_res) ? (const unsigned char *)bp + (size_t)t * W + (n & WM)
                                                                   : (const unsigned char *)a.pool_bp + ((size_t)ep.pool * H + t) * a.W0 + (n & (a.W0 - 1));
                n -= (int)__hip_atomic_load(b8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (RES != 2 || t >= t_res) n = __hip_atomic_load(&bp[(size_t)t * W + (n & WM)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else n = __hip_atomic_load(&((const u16 *)a.pool_bp)[((size_t)ep.pool * H + t) * a.W0 + (n & (a.W0 - 1))], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sh.path[0] = n;
    }
    __syncthreads();

    // outputs; path distance probe of st.py:797-800 (first wave)
    if (tid < 64) {
        bool crash_l = false;
        if (lane < H) {
            const int t = lane;
            int n = (t <= best_t) ? sh.path[t] : -1;
            double pd = __builtin_nan("");
            if (n >= 0) {
                double s_t = sval(n);
                int qi = (int)((s_t - start_s) / delta);                          // st.py:798 -> st.py:20-22
                double d;
                if constexpr (GRID) {
                    d = a.distances[(size_t)t * S + qi];
                } else {
                    size_t row = (size_t)e * H + t;
                    int na = a.tab.nact[row];
                    const double *ce = a.tab.edge + row * a.Kmax * 2;
                    const int *cw = a.tab.win + row * a.Kmax * 2;
                    double sq = sval(qi);
                    d = 1e10;
                    bool blocked = false;
                    for (int c = 0; c < na; ++c) {
                        double f = fabs(sq - ce[c * 2 + 0]);
                        double b = fabs(sq - ce[c * 2 + 1]);
                        d = (f < d) ? f : d; d = (b < d) ? b : d;
                        blocked |= (qi >= cw[c * 2 + 0]) & (qi < cw[c * 2 + 1]);
                    }
                    if (blocked) d = 0.0;
                }
                pd = d;
                crash_l = d < p.crash_dist_thr;
            }
            if constexpr (GRID) {
                a.s_sequence[t] = (n >= 0) ? sval(n) : 0.0;                       // st_cy.pyx:393-398
            } else {
                a.path_idx[(size_t)e * H + t] = n;
                if (a.path_dist) a.path_dist[(size_t)e * H + t] = pd;
                if (t == 1 && a.action_cost) a.action_cost[(size_t)e * 2] = (double)n;
            }
        }
        const bool any_crash = __ballot(crash_l) != 0ull;
        if constexpr (!GRID) {
            if (lane == 0) {
                a.best_t[e] = best_t;
                a.cost[e] = __longlong_as_double((long long)best_bits);
                if (a.action_cost) a.action_cost[(size_t)e * 2 + 1] = __longlong_as_double((long long)best_bits);
                if (a.crash) a.crash[e] = (best_t != H - 1 || any_crash) ? 1 : 0;
            }
        }
    }
    return 0;
}

// Rank of the compute unit this workgroup runs on, in order of first appearance in this launch (0 = first).  The key is the unit's
// hardware position: XCC_ID and the SE / SH / CU fields of HW_ID.
__device__ __forceinline__ int cu_rank(unsigned *tab) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned key1 = (((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu)) + 1u;
    unsigned h = (key1 * 2654435761u) >> 23;
    for (int probe = 0; probe < 512; ++probe, h = (h + 1u) & 511u) {
        const unsigned old = atomicCAS(&tab[h], 0u, key1);
        if (old == 0u) {
            const unsigned r = atomicAdd(&tab[1024], 1u);
            __hip_atomic_store(&tab[512 + h], r + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return (int)r;
        }
        if (old == key1) {
            unsigned r1;
            do { r1 = __hip_atomic_load(&tab[512 + h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while (r1 == 0u);
            return (int)(r1 - 1u);
        }
    }
    return 0;
}

#define STMPC_CELL_BYTES 14     // cost 8 + hist 4 + list 2 per window cell; plus 8 B per penalty-buffer cell
#define STMPC_LIST_SLACK (64 * (STMPC_MAXWAVES + 1) * 2)   // bytes: per-wave list segments are rounded up to whole chunks
// dynamic LDS of a tier: the cell arrays (LDS tiers only) + one int per 64-cell chunk
__host__ __device__ inline size_t stmpc_chunk_ints(int W) { return (size_t)(W / 64 + 8); }
// LDS bytes of the staged vehicle table: H*KT*(2 doubles + 2 ints) + H ints (8-byte aligned)
__host__ __device__ inline size_t stmpc_tab_bytes(int H, int KT) { return (size_t)H * KT * 24 + (((size_t)H * 4 + 7) & ~(size_t)7); }

// minimum resident waves per SIMD the register allocation is made for (4 = 128 VGPRs; A/B builds override it)
#ifndef STMPC_MIN_WAVES
#define STMPC_MIN_WAVES 4
#endif
// Persistent kernel: workgroups of NW waves pull episodes until the tier's queue is drained.
template <bool USE_LDS, bool GRID, bool FASTDIV, int KT, int FANMAX, bool S1GEN, int RES = 0, int NWX = STMPC_MAXWAVES>
// (the standard second window -- NWX 88: eight waves, 147 KB of LDS -- is alone on its compute unit, two waves per SIMD: it may use 256 VGPRs)
__global__ void __launch_bounds__(512, ((FANMAX <= 12 && NWX != 88) ? STMPC_MIN_WAVES : 2)) k_solve(SolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ WgShared sh;
    const int tid = threadIdx.x;
    const int W = WgShape<NWX>::fixed ? WgShape<NWX>::W : a.W;
    const int PWc = WgShape<NWX>::fixed ? WgShape<NWX>::PW : a.PW;
    unsigned char *base;
    int *chunk_cnt;
    // dynamic LDS: [vehicle table][chunk counters][cell arrays (LDS tiers only)]
    const int H = a.p.H;
    double *ltab_e = (double *)smem;
    int *ltab_w = (int *)(ltab_e + (size_t)H * (KT > 0 ? KT : 0) * 2);
    int *ltab_n = ltab_w + (size_t)H * (KT > 0 ? KT : 0) * 2;
    unsigned char *after_tab = smem + stmpc_tab_bytes(H, KT > 0 ? KT : 0);
    chunk_cnt = (int *)after_tab;
    unsigned char *cells = after_tab + ((stmpc_chunk_ints(W) * sizeof(int) + 15) & ~(size_t)15);
    if constexpr (USE_LDS) base = cells;
    else base = a.gscratch + (size_t)blockIdx.x * ((size_t)W * STMPC_CELL_BYTES + STMPC_LIST_SLACK + (size_t)PWc * 8);
    u64 *cost = (u64 *)base;
    double *pen = (double *)(cost + W);
    unsigned *hist = (unsigned *)(pen + PWc);
    u16 *list = (u16 *)(hist + W);

    if constexpr (GRID) {
        int rc = solve_episode<USE_LDS, true, FASTDIV, 0, FANMAX, true>(a, 0, 0, sh, cost, hist, pen, list, chunk_cnt, ltab_e, ltab_w, ltab_n, 0);
        if (rc != 0 && tid == 0) atomicExch(&a.counters[STMPC_CNT_ERR], 1u);
        return;
    } else {
        // Queue of an overflow tier: episodes without a cost bound (the expensive ones) grow from the front of the
        // tier's list, the others from the back; unbounded ones are handed out first to keep the launch's tail short.
        // A concurrent launch (a.concurrent) starts on CUs the previous tier's persistent workgroups have already
        // left and consumes the queue while it is still being filled: entries are published with release stores
        // into slots preset to -1, tickets are taken by compare-and-swap only when an entry exists, and the launch
        // ends when the producers have all exited and the queue is drained.  It never waits unless every producer
        // workgroup is resident (so the producers cannot be waiting for this launch's CUs), and not longer than
        // a.wait_ticks; whatever it leaves is picked up by the ordinary launch of the tier that follows.
        auto claim = [&](int which) -> int {
            unsigned *consumed = &a.counters[STMPC_CNT_CONSUMED + 2 * a.tier + which];
            unsigned *total = &a.counters[4 * a.tier + 2 + which];
            for (;;) {
                const unsigned c = __hip_atomic_load(consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned t = __hip_atomic_load(total, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (c >= t) return -1;
                if (atomicCAS(consumed, c, c + 1u) != c) continue;
                int *slot = &a.lists[(size_t)a.tier * a.N + (which == 0 ? c : (unsigned)a.N - 1u - c)];
                int e_;
                // (a producer publishes its entry right after taking the slot: the wait is two instructions long, but bounded all
                // the same -- if the entry does not appear within wait_ticks the launch raises the error flag and the call fails
                // with STMPC_EINTERNAL instead of hanging)
                const unsigned long long t_spin = a.concurrent ? wall_clock64() : 0ull;
                do { e_ = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
                while (e_ < 0 && a.concurrent && wall_clock64() - t_spin <= a.wait_ticks);
                if (e_ < 0 && a.concurrent) atomicExch(&a.counters[STMPC_CNT_ERR], 1u);
                return e_;
            }
        };
        if (a.last_tier && !a.concurrent && a.host_overflow && blockIdx.x == 0 && tid == 0) {    // (every earlier launch of the batch has finished: the counts are final)
            __hip_atomic_store(a.host_overflow, (int)__hip_atomic_load(&a.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // [1]: episodes queued for THIS (the last) tier -- the host sizes the next batch's clean-up launch by it
            __hip_atomic_store(a.host_overflow + 1, a.tier > 0 ? (int)__hip_atomic_load(&a.counters[4 * a.tier], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned long long t_begin = a.concurrent ? wall_clock64() : 0ull;
        if (a.tier == 0 && a.feeds_concurrent && tid == 0) atomicAdd(&a.counters[STMPC_CNT_RESIDENT], 1u);
        int my_rank = 0;
        if (a.tier == 0 && a.cu_tab && tid == 0) my_rank = cu_rank(a.cu_tab);
        bool may_wait = false;
        if (a.concurrent && tid == 0)
            may_wait = a.always_wait || __hip_atomic_load(&a.counters[STMPC_CNT_RESIDENT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)a.prev_grid;
        for (;;) {
            __syncthreads();
            if (tid == 0) {
                int e = -1;
                int task_phase = a.phase;
                if (a.tier == 0 && a.split) {
                    // 2N tasks: bounding pre-passes of all episodes first, then the exact passes in the same order --
                    // finer-grained tasks pack the persistent workgroups better.  An exact task is handed out at
                    // least N tasks after its bounding task, which is therefore long finished; the wait below only
                    // covers the pathological case.
                    unsigned w = 2u * (unsigned)a.N;
                    bool retire = false;
                    if (my_rank > 0) {
                        const long long left = 2ll * a.N - (long long)__hip_atomic_load(&a.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        retire = my_rank >= a.retire_from && left < a.retire_left;
                    }
                    if (!retire) w = atomicAdd(&a.counters[0], 1u);
                    if (w < (unsigned)a.N) { e = a.order ? a.order[w] : (int)w; task_phase = 1; }
                    else if (w < 2u * (unsigned)a.N) {
                        e = a.order ? a.order[w - (unsigned)a.N] : (int)(w - (unsigned)a.N); task_phase = 2;
                        for (;;) {
                            const unsigned px = __hip_atomic_load(&a.proxy[e], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                            if (px == STMPC_PROXY_MOVED) { task_phase = 3; break; }      // nothing left to do here
                            if (__hip_atomic_load(&a.ubound[e], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
                            __builtin_amdgcn_s_sleep(32);
                        }
                    }
                    sh.rc = task_phase;
                } else if (a.tier == 0) {
                    unsigned w = atomicAdd(&a.counters[a.phase == 1 ? 2 : 0], 1u);
                    if (w < (unsigned)a.N) e = a.order ? a.order[w] : (int)w;
                    sh.rc = a.phase;
                } else {
                    for (;;) {
                        e = claim(0);
                        if (e < 0) e = claim(1);
                        if (e >= 0 || !a.concurrent || !may_wait) break;
                        if (__hip_atomic_load(&a.counters[STMPC_CNT_FINISHED], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)a.prev_grid) {
                            e = claim(0);                          // producers are gone: whatever is queued now is final
                            if (e < 0) e = claim(1);
                            break;
                        }
                        // (a consumer lives for the first launch's tail, a few ms: a tenth of the bound on a published entry -- 20 ms -- is
                        // ample, and caps what a mis-scheduled side launch can cost when several batches are in flight)
                        if (wall_clock64() - t_begin > a.wait_ticks / 10) break;
                        __builtin_amdgcn_s_sleep(64);
                    }
                }
                sh.work = e;
            }
            __syncthreads();
            const int e = sh.work;
            if (e < 0) break;
            const int task_phase = (a.tier == 0) ? sh.rc : a.phase;
            if (task_phase == 3) continue;
            int rc = solve_episode<USE_LDS, false, FASTDIV, KT, FANMAX, S1GEN, RES, NWX>(a, e, blockIdx.x, sh, cost, hist, pen, list, chunk_cnt, ltab_e, ltab_w, ltab_n, task_phase);
            if (rc != 0 && tid == 0) {
                if (!a.last_tier) {
                    atomicAdd(&a.counters[4 * (a.tier + 1)], 1u);
                    const u64 ub = a.ubound ? a.ubound[e] : 0ull;          // written by this thread in solve_episode
                    const bool heavy = (ub == 0ull || ub == INF_BITS) || (rc & 4) != 0;
                    const unsigned pos = atomicAdd(&a.counters[4 * (a.tier + 1) + (heavy ? 2 : 3)], 1u);
                    int *slot = &a.lists[(size_t)(a.tier + 1) * a.N + (heavy ? pos : (unsigned)a.N - 1u - pos)];
                    __hip_atomic_store(slot, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // after ubound[e]
                } else {
                    atomicExch(&a.counters[STMPC_CNT_ERR], 1u);   // the last tier's window covers all S cells
                }
            }
        }
        if (a.tier == 0 && a.feeds_concurrent && tid == 0)
            __hip_atomic_fetch_add(&a.counters[STMPC_CNT_FINISHED], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Heaviest-first order of the episodes for the exact phase (LPT): counting sort of the work estimates into
// 64 logarithmic buckets, one workgroup.
__global__ void __launch_bounds__(1024) k_order(int N, const unsigned *__restrict__ proxy, int *__restrict__ order) {
    __shared__ unsigned cnt[64], off[64];
    const int tid = threadIdx.x;
    if (tid < 64) cnt[tid] = 0u;
    __syncthreads();
    auto bucket = [](unsigned v) -> int {       // 2 buckets per octave, descending work
        v += 1u;
        const int l = 31 - __clz((int)v);
        const int half = (l > 0) ? (int)((v >> (l - 1)) & 1u) : 0;
        int b = 2 * l + half;
        b = b > 63 ? 63 : b;
        return 63 - b;
    };
    for (int e = tid; e < N; e += blockDim.x) atomicAdd(&cnt[bucket(proxy[e])], 1u);
    __syncthreads();
    if (tid == 0) { unsigned acc = 0; for (int b = 0; b < 64; ++b) { off[b] = acc; acc += cnt[b]; } }
    __syncthreads();
    for (int e = tid; e < N; e += blockDim.x) { const unsigned pos = atomicAdd(&off[bucket(proxy[e])], 1u); order[pos] = e; }
}

// Episodes in ascending order of a one-byte key (counting sort, one workgroup): the static heavy-first order of the tasks.
__global__ void __launch_bounds__(1024) k_order8(int N, const unsigned char *__restrict__ key, int *__restrict__ order) {
    __shared__ unsigned cnt[256], off[256];
    const int tid = threadIdx.x;
    if (tid < 256) cnt[tid] = 0u;
    __syncthreads();
    for (int e = tid; e < N; e += blockDim.x) atomicAdd(&cnt[key[e]], 1u);
    __syncthreads();
    if (tid == 0) { unsigned acc = 0; for (int b = 0; b < 256; ++b) { off[b] = acc; acc += cnt[b]; } }
    __syncthreads();
    for (int e = tid; e < N; e += blockDim.x) { const unsigned pos = atomicAdd(&off[key[e]], 1u); order[pos] = e; }
}

// Materialise the reference's grids for one state (st.py:25-70), from the car table of episode 0.
__global__ void k_build_grid(DevP p, CarTab tab, int Kmax, double start_s, int S, uint8_t *obstacles,
                             double *distances, double *s_values) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    int t = blockIdx.y;
    if (n >= S) return;
    double s1 = start_s + p.ds, delta = s1 - start_s;
    double sn = (n == 1) ? s1 : start_s + (double)n * delta;
    if (t == 0) s_values[n] = sn;
    size_t row = (size_t)t;
    int na = tab.nact[row];
    const double *ce = tab.edge + row * Kmax * 2;
    const int *cw = tab.win + row * Kmax * 2;
    double d = 0.0 + 1e10;
    bool blocked = false;
    for (int c = 0; c < na; ++c) {
        double f = fabs(sn - ce[c * 2 + 0]);
        double b = fabs(sn - ce[c * 2 + 1]);
        d = (f < d) ? f : d; d = (b < d) ? b : d;
        blocked |= (n >= cw[c * 2 + 0]) & (n < cw[c * 2 + 1]);
    }
    obstacles[(size_t)t * S + n] = blocked ? 1 : 0;
    distances[(size_t)t * S + n] = blocked ? 0.0 : d;
}

__global__ void k_probe(int op, const double *a, const double *b, double *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b ? b[i] : 0.0, r;
    switch (op) {
        case 0: r = x / y; break;
        case 1: r = sqrt(x); break;
        case 2: r = x * y; break;
        case 3: r = x + y; break;
        case 5: r = divc<true>(x, y, 1.0 / y); break;
        case 6: { const double zh = 1.0 / y; r = divk<true>(x, y, zh, __builtin_fma(-y, zh, 1.0) / y); break; }   // (zl as fastdiv2_ok computes it)
        default: r = __builtin_fma(x, x, y * y); break;
    }
    out[i] = r;
}

}  // namespace stmpc
