// stmpc_kernels.hpp -- CDNA4 (gfx950) device code of the ST lattice solver.
//
// Design:
//   k_predict : one THREAD per merge episode.  Runs the reference's traffic predictor
//               (prediction.py:22-105) H-1 times and emits, per time layer, the list of
//               vehicles that obstruct the lattice (st.py:44-65): front/back edge and the
//               blocked index window.  The H x S obstacle / distance grids of the reference
//               are never materialised.
//   k_solve   : one WAVEFRONT (64 lanes) per episode, persistent blocks pulling episodes from
//               a device work counter.  Layer-synchronous forward DP that is equivalent to the
//               reference's heap Dijkstra (st_cy.pyx:315-399): lanes stripe over the source
//               nodes of a layer, each lane relaxes its node's <=A candidate cells into the
//               next layer with an exact (cost, predecessor-index) minimum built from
//               ds_min_rtn_u64 on the fp64 bit pattern plus an in-order fix-up of the
//               predecessor (valid because one wave owns the episode and LDS is in-order per
//               wave).  Layers live in circular LDS windows of W cells; an episode whose
//               reachable span exceeds W is queued for the HBM-scratch variant of the same code.
//
// Everything is fp64 and compiled with -ffp-contract=off: one IEEE op per reference op.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned short u16;

#define STMPC_WAVE 64
#define STMPC_MAXH 64

namespace stmpc {

static constexpr u64 INF_BITS = 0x7FF0000000000000ull;

// Launch-constant parameters (kernel argument, lives in SGPRs / kernarg segment).
struct DevP {
    double future_s, ds, dt, dt2, dt3;
    double d_w, v_w, a_w, j_w, v_des, v_max, a_min, a_max, j_min, j_max, min_allowed;
    double car_length, obst_min_s /* crash_min_s - min_allowed, st.py:46 */;
    double max_pred_decel, follow_gap, react_thr, crash_thr, crash_dist_thr /* comb_min_dist - car_length, st.py:800 */;
    double unc[STMPC_MAXH];     // start_unc + unc_per_s * t_values[t]   (st.py:40)
    int    dunc[STMPC_MAXH];    // int(unc / ds)                         (st.py:41)
    int    H;
    int    dlen;                // int(CAR_LENGTH / ds)                  (st.py:37)
};

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ double dmin_py(double a, double b) { return (b < a) ? b : a; }   // Python/Cython min(a,b)
__device__ __forceinline__ double dmax_py(double a, double b) { return (b > a) ? b : a; }   // Python/Cython max(a,b)

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
// lexicographic min of (bits, n) across the wave
__device__ __forceinline__ void wave_min_key(u64 &bits, int &n) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        u64 ob = __shfl_xor(bits, o);
        int on = __shfl_xor(n, o);
        if (ob < bits || (ob == bits && on < n)) { bits = ob; n = on; }
    }
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
                                                     double dt, double min_crash_distance) {
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
            if (sd < 0 && xd < p.follow_gap) {
                double acc = dmax_py(sd, p.max_pred_decel);
                nv = ov + acc * dt;
            } else nv = ov;
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
                                                        double min_crash_distance) {
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
    return dev_predict_with_ego<KMAX>(p, s, sel, dt, min_crash_distance);
}

// Per-(episode, layer) list of obstructing vehicles.
struct CarTab {
    double *edge;     // [N][H][Kmax][2]  front = o-L-u, back = o+L+u          (st.py:52-53)
    int    *win;      // [N][H][Kmax][2]  blocked index window [imin, imax)     (st.py:60-65)
    int    *nact;     // [N][H]
    int    *num_s;    // [N]   S of the episode
};

template <int KMAX>
__global__ void __launch_bounds__(64) k_predict(DevP p, int N, int Kmax, const double *__restrict__ ego,
                                                const int *__restrict__ k_count,
                                                const double *__restrict__ other_x,
                                                const double *__restrict__ other_v, CarTab tab,
                                                unsigned *counters /* [64], zeroed here */) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 64) counters[e] = 0u;
    if (e >= N) return;
    DState<KMAX> s;
    s.ex = ego[e * 5 + 0]; s.ey = ego[e * 5 + 1]; s.ev = ego[e * 5 + 2]; s.ea = ego[e * 5 + 3];
    double start_s = ego[e * 5 + 4];
    int k = k_count[e];
    k = k < 0 ? 0 : (k > KMAX ? KMAX : k);
    s.k = k;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        bool in = (i < k) && (i < Kmax);
        s.xs[i] = in ? other_x[(size_t)e * Kmax + i] : 0.0;
        s.vs[i] = in ? other_v[(size_t)e * Kmax + i] : 0.0;
    }
    int S = dev_num_s(p, start_s);
    tab.num_s[e] = S;
    // s_values[S-1] per numpy's arange fill
    double s1 = start_s + p.ds;
    double delta = s1 - start_s;
    double s_last = (S - 1 == 1) ? s1 : start_s + (double)(S - 1) * delta;
    for (int t = 0; t < p.H; ++t) {
        if (t != 0) dev_predict_without_ego<KMAX>(p, s, p.dt, 5.0);            // st.py:42-43
        double unc = p.unc[t];
        int dunc = p.dunc[t];
        size_t rowbase = ((size_t)e * p.H + t) * Kmax;
        int na = 0;
        bool stop = false;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            if (c < k && !stop) {
                double o = s.xs[c] - (-51.0);                                  // control.py:388-389
                if (o < p.obst_min_s) stop = true;                             // st.py:46-47 break
                else if (o > s_last + p.car_length) { /* continue */ }         // st.py:48-49
                else {
                    double front = o - p.car_length - unc;
                    double back = o + p.car_length + unc;
                    int i0 = (int)((o - start_s) / p.ds);                      // st.py:60 (trunc toward 0)
                    int imin = i0 - p.dlen - dunc; imin = imin < 0 ? 0 : imin;
                    int imax = i0 + p.dlen + dunc; imax = imax > S ? S : imax;
                    if (!(imin < S && imax > 0)) { imin = 0; imax = 0; }       // st.py:63
                    tab.edge[(rowbase + na) * 2 + 0] = front;
                    tab.edge[(rowbase + na) * 2 + 1] = back;
                    tab.win[(rowbase + na) * 2 + 0] = imin;
                    tab.win[(rowbase + na) * 2 + 1] = imax;
                    ++na;
                }
            }
        }
        tab.nact[(size_t)e * p.H + t] = na;
    }
}

// One-step prediction exposed through the C-ABI (stmpc_predict_batch).
template <int KMAX>
__global__ void __launch_bounds__(64) k_predict_step(DevP p, int mode, int N, int Kmax,
                                                     const double *__restrict__ ego4, const int *__restrict__ k_count,
                                                     const double *__restrict__ other_x, const double *__restrict__ other_v,
                                                     const double *__restrict__ sel, double dt, double mcd,
                                                     double *ego4_out, double *ox_out, double *ov_out, int *crashed) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    DState<KMAX> s;
    s.ex = ego4[e * 4 + 0]; s.ey = ego4[e * 4 + 1]; s.ev = ego4[e * 4 + 2]; s.ea = ego4[e * 4 + 3];
    int k = k_count[e];
    k = k < 0 ? 0 : (k > KMAX ? KMAX : k);
    s.k = k;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        bool in = (i < k) && (i < Kmax);
        s.xs[i] = in ? other_x[(size_t)e * Kmax + i] : 0.0;
        s.vs[i] = in ? other_v[(size_t)e * Kmax + i] : 0.0;
    }
    bool cr = (mode == 0) ? dev_predict_with_ego<KMAX>(p, s, sel[e], dt, mcd)
                          : dev_predict_without_ego<KMAX>(p, s, dt, mcd);
    ego4_out[e * 4 + 0] = s.ex; ego4_out[e * 4 + 1] = s.ey; ego4_out[e * 4 + 2] = s.ev; ego4_out[e * 4 + 3] = s.ea;
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
        if (i < k && i < Kmax) { ox_out[(size_t)e * Kmax + i] = s.xs[i]; ov_out[(size_t)e * Kmax + i] = s.vs[i]; }
    crashed[e] = cr ? 1 : 0;
}

// ---------------------------------------------------------------- lattice DP
// st_cy.pyx:34-38 distance_penalty, pre-multiplied by d_weight exactly as st_cy.pyx:50 does
__device__ __forceinline__ double dev_weighted_penalty(double d, double min_allowed, double d_w) {
    double pen = (d < min_allowed) ? (1000000.0 / dmax_py(d, 1.0)) : (1 / d);
    return d_w * pen;
}

// Correctly rounded x / d for a divisor whose correctly rounded reciprocal r = RN(1/d) is known:
// q0 = RN(x*r) is within 1.5 ulp; one residual step makes it faithful, the second (Markstein's
// theorem: faithful q, exact residual by FMA, r = RN(1/d)) gives RN(x/d).  5 ops instead of the
// ~12 of the generic v_div_scale/v_rcp/v_div_fmas/v_div_fixup expansion.  The FMAs here are
// explicit; nothing else in this file may contract (-ffp-contract=off).
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

#define STMPC_MAX_TIERS 4
#define STMPC_CNT_ERR 63

struct SolveArgs {
    DevP p;
    int N;
    int Kmax;
    int W;                 // window cells of this tier (power of two)
    int tier;              // 0: pull episodes 0..N-1 from counters[0]; k>=1: walk list k
    int last_tier;         // overflow here is an internal error
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
    u16 *bp;               // [blocks][H][W] back-pointers of this tier
    unsigned char *gscratch;   // HBM-storage variant: [blocks][20*W] bytes
    unsigned *counters;    // [0] tier-0 work counter; [4k] list-k length, [4k+1] list-k work counter; [63] error flag
    int *lists;            // [STMPC_MAX_TIERS][N] episode ids queued for tier k
    // outputs
    int *path_idx;         // [N][H]
    int *best_t;           // [N]
    double *cost;          // [N]
    double *path_dist;     // [N][H] or null
    int *crash;            // [N] or null
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
    static __device__ __forceinline__ double ldf(const double *p) {
        return __longlong_as_double((long long)__hip_atomic_load((const u64 *)p, __ATOMIC_RELAXED, SCOPE));
    }
    static __device__ __forceinline__ void stf(double *p, double v) {
        __hip_atomic_store((u64 *)p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, SCOPE);
    }
    // order this wave's own accesses: LDS is in-order per wave (a compiler barrier suffices);
    // the HBM variant drains the vector-memory queue.
    static __device__ __forceinline__ void order() {
        if constexpr (USE_LDS) __atomic_signal_fence(__ATOMIC_SEQ_CST);
        else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
};

// Solve one episode with one wavefront.  Returns 0 ok, 1 window overflow.
//
// Storage per lattice cell (circular, slot = cell & (W-1)):
//   cost[]  u64  fp64 bits of the accumulated cost of the node, +inf bits = not reached
//   hist[]  u32  (index of s_{t-1}) | (index of s_{t-2}) << 16 of the winning chain
//   pen[]   f64  d_weight * distance_penalty of the cell in the layer being relaxed into, < 0 = blocked
// ONE cost/hist array serves both the layer being expanded and the layer being built: sources are
// consumed in DESCENDING chunks of 64 cells (loaded to registers first) and every edge goes to a
// cell index >= its source (speeds are >= 0), so a target never lands on a source that is still
// unread.  Live cells at any time lie in [wlo, ihi), which must fit in W.
template <bool USE_LDS, bool GRID, bool FASTDIV, int KT>
__device__ int solve_episode(const SolveArgs &a, int e, int slot, u64 *cost, unsigned *hist, double *pen,
                             int *path_lds) {
    typedef Mem<USE_LDS> M;
    const DevP &p = a.p;
    const int lane = threadIdx.x & 63;
    const int W = a.W, WM = a.W - 1;
    const int H = p.H;
    double start_s, v0, a0, s1, delta;
    int S;
    if constexpr (GRID) {
        start_s = a.s_values[0]; s1 = a.s_values[1]; delta = s1 - start_s;      // st_cy.pyx:318,320
        v0 = a.v0_grid; a0 = a.a0_grid; S = a.S_grid;
    } else {
        start_s = a.ego[(size_t)e * 5 + 4]; v0 = a.ego[(size_t)e * 5 + 2]; a0 = a.ego[(size_t)e * 5 + 3];
        s1 = start_s + p.ds; delta = s1 - start_s; S = a.tab.num_s[e];
    }
    const double dt = p.dt, dt2 = p.dt2, dt3 = p.dt3;
    const double r_dt = 1.0 / dt, r_dt2 = 1.0 / dt2, r_dt3 = 1.0 / dt3, r_delta = 1.0 / delta;
    // st_cy.pyx:329-330 virtual history
    const double est_prev = start_s - v0 * dt;
    const double est_second = est_prev - dt * (v0 - a0 * dt);
    const bool s1_plain = (start_s + 1.0 * delta == s1);      // numpy arange: a[1] = start+step, a[i>=2] = start+i*delta
    auto sval = [&](int n) -> double {
        if constexpr (GRID) return a.s_values[n];
        else {
            double v = start_s + (double)n * delta;
            if (!s1_plain) { if (n == 1) v = s1; }
            return v;
        }
    };
    u16 *bp = a.bp + (size_t)slot * H * W;

    if (lane == 0) { M::st64(&cost[0], 0ull); M::st32(&hist[0], 0u); }
    M::order();
    int wlo = 0, whi = 1;
    int best_t = 0, best_n = 0;
    u64 best_bits = 0ull;

    for (int t = 0; t < H; ++t) {
        const bool relax = t < H - 1;
        int ilo = 0, ihi = 0;
        bool first = true;
        u64 my_best = ~0ull;
        int my_best_n = 0x7fffffff;
        int n_active = 0;
        // obstructing vehicles of layer t+1 (wave-uniform): kept in SGPRs when KT > 0
        int nact = 0;
        double cfront[KT > 0 ? KT : 1], cback[KT > 0 ? KT : 1];
        int cimin[KT > 0 ? KT : 1], cimax[KT > 0 ? KT : 1];
        const double *cedge = nullptr;
        const int *cwin = nullptr;
        if constexpr (!GRID) {
            if (relax) {
                size_t row = (size_t)e * H + (t + 1);
                nact = a.tab.nact[row];
                cedge = a.tab.edge + row * a.Kmax * 2;
                cwin = a.tab.win + row * a.Kmax * 2;
                if constexpr (KT > 0) {
                    // lane c loads vehicle c, then broadcast to scalars
                    double f = 0.0, b = 0.0; int i0 = 0, i1 = 0;
                    if (lane < nact) { f = cedge[lane * 2]; b = cedge[lane * 2 + 1]; i0 = cwin[lane * 2]; i1 = cwin[lane * 2 + 1]; }
#pragma unroll
                    for (int c = 0; c < KT; ++c) {
                        cfront[c] = __shfl(f, c); cback[c] = __shfl(b, c);
                        cimin[c] = __shfl(i0, c); cimax[c] = __shfl(i1, c);
                    }
                }
            }
        }
        auto init_cells = [&](int from, int to) {
            for (int n = from + lane; n < to; n += 64) {
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
                                double f = fabs(sn - cfront[c]);
                                double b = fabs(sn - cback[c]);
                                d = (f < d) ? f : d; d = (b < d) ? b : d;    // st.py:56-57
                                blocked |= (n >= cimin[c]) & (n < cimax[c]); // st.py:64
                            }
                        }
                    } else {
                        for (int c = 0; c < nact; ++c) {
                            double f = fabs(sn - cedge[c * 2 + 0]);
                            double b = fabs(sn - cedge[c * 2 + 1]);
                            d = (f < d) ? f : d; d = (b < d) ? b : d;
                            blocked |= (n >= cwin[c * 2 + 0]) & (n < cwin[c * 2 + 1]);
                        }
                    }
                    pv = blocked ? -1.0 : dev_weighted_penalty(d, p.min_allowed, p.d_w);
                }
                M::stf(&pen[n & WM], pv);
                M::st64(&cost[n & WM], INF_BITS);
            }
            M::order();
        };

        // sources of layer t, highest cells first
        for (int top = (whi + 63) & ~63; top > wlo; top -= 64) {      // chunks aligned to multiples of 64 cells
            const int i = top - 64 + lane;
            const bool valid = (i >= wlo) & (i < whi);
            const u64 cb = valid ? M::ld64(&cost[i & WM]) : INF_BITS;
            const bool act = cb < INF_BITS;
            const u64 amask = __ballot(act);
            if (!amask) continue;
            n_active += __popcll(amask);
            double sv = 0.0, p1 = 0.0, p2 = 0.0;
            const double C = __longlong_as_double((long long)cb);
            int lo = 0, hi = 0;
            unsigned myh = 0u;        // what a target won by this source stores: i | (p1 index << 16)
            if (act) {
                if (cb < my_best || (cb == my_best && i < my_best_n)) { my_best = cb; my_best_n = i; }
                sv = sval(i);
                if (t == 0) { p1 = est_prev; p2 = est_second; myh = 0u; }      // st_cy.pyx:342
                else {
                    const unsigned h = M::ld32(&hist[i & WM]);
                    const int pr = (int)(h & 0xFFFFu), pp = (int)(h >> 16);
                    bp[(size_t)t * W + (i & WM)] = (u16)pr;
                    p1 = sval(pr);
                    p2 = (t == 1) ? est_prev : sval(pp);
                    myh = (unsigned)i | ((unsigned)pr << 16);
                }
                if (relax) {
                    // st_cy.pyx:65-75
                    double prev_v = divc<FASTDIV>(p1 - p2, dt, r_dt);
                    double v = divc<FASTDIV>(sv - p1, dt, r_dt);
                    double acc = divc<FASTDIV>(v - prev_v, dt, r_dt);
                    double min_a = dmax_py(acc + p.j_min * dt, p.a_min);
                    double max_a = dmin_py(acc + p.j_max * dt, p.a_max);
                    double min_v = dmax_py(v + min_a * dt, 0.0);
                    double max_v = dmin_py(v + max_a * dt, p.v_max);
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
                    if (lo >= hi) { lo = 0; hi = 0; }
                }
            }
            M::order();      // this chunk's cost/hist are in registers: its cells may now be overwritten
            if (!relax) continue;
            int clo = wave_min_i(hi > lo ? lo : 0x7fffffff), chi = wave_max_i(hi);
            if (clo >= chi) continue;
            // the interval of initialised next-layer cells grows in whole 64-cell blocks: clo >= top-64 (a
            // multiple of 64), so rounding down never touches a source that is still unread
            clo &= ~63; chi = (chi + 63) & ~63;
            if (first) { ilo = ihi = clo; first = false; }
            const int nlo2 = clo < ilo ? clo : ilo, nhi2 = chi > ihi ? chi : ihi;
            if (nhi2 - (wlo & ~63) > W) return 1;                            // live cells exceed the circular window
            if (clo < ilo) init_cells(clo, ilo);
            if (chi > ihi) init_cells(ihi, chi);
            ilo = nlo2; ihi = nhi2;

            const double two_sv = 2 * sv, three_sv = 3 * sv, three_p1 = 3 * p1;
            for (int c = 0;; ++c) {
                const int n = lo + c;
                if (!__ballot(n < hi)) break;                                // inactive lanes have lo = hi = 0
                bool tie = false;
                int sl = 0;
                if (n < hi) {
                    sl = n & WM;
                    const double pn = M::ldf(&pen[sl]);
                    if (pn >= 0.0) {                                         // st_cy.pyx:383 obstacle skip
                        const double sn = sval(n);
                        // st_cy.pyx:46-50 cost_with_jerk(next, s, p1, p2)
                        const double v = divc<FASTDIV>(sn - sv, dt, r_dt);
                        const double aa = divc<FASTDIV>(sn - two_sv + p1, dt2, r_dt2);
                        const double jj = divc<FASTDIV>(sn - three_sv + three_p1 - p2, dt3, r_dt3);
                        const double dv = v - p.v_des;
                        const double ec = p.v_w * (dv * dv) + p.a_w * (aa * aa) + p.j_w * (jj * jj) + pn;
                        const double tot = C + ec;                           // st_cy.pyx:388
                        const u64 tb = (u64)__double_as_longlong(tot);
                        const u64 old = M::min64(&cost[sl], tb);
                        M::order();
                        const u64 cur = M::ld64(&cost[sl]);
                        if (cur == tb) {
                            if (old > tb) M::st32(&hist[sl], myh);           // unique first setter of this value
                            else tie = true;                                 // equal cost already present
                        }
                    }
                }
                M::order();
                // equal total cost: the smaller predecessor index wins (heap tuple order, st_cy.pyx:388)
                u64 tm = __ballot(tie);
                while (tm) {
                    const int l = __ffsll((long long)tm) - 1;
                    tm &= tm - 1;
                    if (lane == l) { unsigned q = M::ld32(&hist[sl]); if ((unsigned)i < (q & 0xFFFFu)) M::st32(&hist[sl], myh); }
                    M::order();
                }
            }
        }
        if (n_active == 0) break;            // layer t is empty: the deepest layer reached is t-1
        wave_min_key(my_best, my_best_n);
        best_t = t; best_n = my_best_n; best_bits = my_best;
        if (!relax) break;
        if (first) { wlo = 0; whi = 0; } else { wlo = ilo; whi = ihi; }
    }

    // back-track (st_cy.pyx:391-398)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (lane == 0) {
        int n = best_n;
        for (int t = best_t; t > 0; --t) {
            path_lds[t] = n;
            n = __hip_atomic_load(&bp[(size_t)t * W + (n & WM)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        path_lds[0] = n;
    }
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);   // lane 0's LDS writes visible to the wave
    __atomic_signal_fence(__ATOMIC_SEQ_CST);

    // outputs; path distance probe of st.py:797-800
    bool crash_l = false;
    if (lane < H) {
        const int t = lane;
        int n = (t <= best_t) ? *(volatile int *)&path_lds[t] : -1;
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
        }
    }
    const bool any_crash = __ballot(crash_l) != 0ull;
    if constexpr (!GRID) {
        if (lane == 0) {
            a.best_t[e] = best_t;
            a.cost[e] = __longlong_as_double((long long)best_bits);
            if (a.crash) a.crash[e] = (best_t != H - 1 || any_crash) ? 1 : 0;
        }
    }
    return 0;
}

#define STMPC_CELL_BYTES 20     // cost 8 + pen 8 + hist 4

// Persistent kernel: blocks of one wave pull episodes until the tier's queue is drained.
template <bool USE_LDS, bool GRID, bool FASTDIV, int KT>
__global__ void __launch_bounds__(64) k_solve(SolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int path_lds[STMPC_MAXH];
    const int lane = threadIdx.x;
    const int W = a.W;
    unsigned char *base;
    if constexpr (USE_LDS) base = smem;
    else base = a.gscratch + (size_t)blockIdx.x * (size_t)W * STMPC_CELL_BYTES;
    u64 *cost = (u64 *)base;
    double *pen = (double *)(cost + W);
    unsigned *hist = (unsigned *)(pen + W);

    if constexpr (GRID) {
        int rc = solve_episode<USE_LDS, true, FASTDIV, 0>(a, 0, 0, cost, hist, pen, path_lds);
        if (rc != 0 && lane == 0) atomicExch(&a.counters[STMPC_CNT_ERR], 1u);
        return;
    } else {
        for (;;) {
            int e;
            if (a.tier == 0) {
                unsigned w = 0;
                if (lane == 0) w = atomicAdd(&a.counters[0], 1u);
                w = __builtin_amdgcn_readfirstlane(w);
                if (w >= (unsigned)a.N) break;
                e = (int)w;
            } else {
                unsigned w = 0, cnt = 0;
                if (lane == 0) {
                    w = atomicAdd(&a.counters[4 * a.tier + 1], 1u);
                    cnt = __hip_atomic_load(&a.counters[4 * a.tier], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                w = __builtin_amdgcn_readfirstlane(w);
                cnt = __builtin_amdgcn_readfirstlane(cnt);
                if (w >= cnt) break;
                e = a.lists[(size_t)a.tier * a.N + w];
            }
            int rc = solve_episode<USE_LDS, false, FASTDIV, KT>(a, e, blockIdx.x, cost, hist, pen, path_lds);
            if (rc != 0 && lane == 0) {
                if (!a.last_tier) {
                    unsigned pos = atomicAdd(&a.counters[4 * (a.tier + 1)], 1u);
                    a.lists[(size_t)(a.tier + 1) * a.N + pos] = e;
                } else {
                    atomicExch(&a.counters[STMPC_CNT_ERR], 1u);   // the last tier's window covers all S cells
                }
            }
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
        }
    }
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
        default: r = __builtin_fma(x, x, y * y); break;
    }
    out[i] = r;
}

}  // namespace stmpc
