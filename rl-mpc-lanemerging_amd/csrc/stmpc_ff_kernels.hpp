// stmpc_ff_kernels.hpp -- QP re-sampling of a coarse ST path to the simulator tick: st.finer_fit (st.py:584-723).
//
// The reference builds a dense QP (<= 64 variables, 6(n-1) difference constraints, one equality) and hands it to
// cvxopt.solvers.qp with maxiters = 10 (st.py:16-17,722).  Here one wavefront solves one QP entirely in registers:
// lane i owns variable x_i and the (up to 8) constraint rows that start at i -- speed >= 0, speed <= v_max,
// acceleration <= a_max, >= a_min, jerk <= j_max, >= j_min (rows i of V_1, V_2, A_3, A_4, J_5, J_6, st.py:609-668)
// and the optional position bounds of C_7 (st.py:670-702).  Each row touches x_{i-2}..x_{i+1}, so G x and G' z are
// three lane shuffles, the normal matrix P + G' D G has 7 diagonals, and its L D L' factor is a 64-step lane-serial
// recurrence on v_readlane broadcasts.  The iteration is cvxopt's coneqp restricted to the componentwise cone
// (Mehrotra predictor-corrector, same starting point, step rule, centering exponent, tolerances and iteration cap;
// tests/test_finer_fit.py checks this kernel bit for bit against the CPU restatement of the same iteration).
#pragma once
#include <hip/hip_runtime.h>

namespace stmpc {

struct FFConst {
    double dt, cdt;                         // delta_t (simulator tick), coarse_delta_t (planning step)
    double cv, ca1, ca2, cj1, cj2, cj3;     // 1/dt, 1/dt^2, 2/dt^2, 1/dt^3, 2/dt^3, 3/dt^3, computed on the host as the reference does
    double dt2;                             // dt ** 2 (libm pow on the host)
    double v_max, a_max, a_min, j_max, j_min, car_length;
    int maxiters;
};

struct FFArgs {
    FFConst k;
    int N, Hs, n_max;
    // input A: explicit coarse paths
    const double *s_seq;      // [N][Hs]
    const int *len;           // [N]
    const double *v0, *a0;    // [N]
    const double *bac;        // [N][4] or null
    // input B: lattice paths straight from the DP (do_st_control, st.py:757-772): s_t = start_s + idx_t * delta
    const int *path_idx;      // [N][Hs]
    const int *best_t;        // [N]
    const double *ego;        // [N][5]
    double ds;
    int use_qp;               // 0: TICK_LENGTH >= T_DISCRETIZATION, the coarse path is used as is (st.py:771)
    // outputs (any may be null)
    double *out;              // [N][n_max]
    int *out_len;             // [N]; -1: more fine samples than the launch's group width (at most 64 are supported)
    int *iters;               // [N]; iterations, negated when the cap was hit without convergence
    double *speed;            // [N]  (x_1 - x_0) / dt, or the current speed when the path has one point (st.py:774-783)
    unsigned *refused;        // or null: word that is OR-ed with 1 when a path cannot be re-sampled (more fine samples than the group holds);
                              // the controller entries point it at the context's sticky flag that stmpc_check_error reports as STMPC_EINVAL
};

// A problem with at most GW (16, 32 or 64) fine samples occupies an aligned group of GW lanes, so a wavefront carries
// 64 / GW problems.  Lanes past a problem's last sample hold zeros, which makes the narrower xor butterflies below
// produce the same bits as the 64-wide one: results do not depend on how problems are packed.
template <int GW>
__device__ __forceinline__ double ff_at(double v, int gbase, int idx) {   // value of the group's lane idx, 0 outside the group
    const double r = __shfl(v, gbase + (idx & (GW - 1)), 64);
    return (idx >= 0 && idx < GW) ? r : 0.0;
}
// Value of the group's lane i (i uniform across the wavefront).  A full-width group broadcasts through v_readlane
// (no LDS round trip on the lane-serial recurrences; EXEC is ignored); packed groups need a different source lane per
// group, where one ds_bpermute beats 2-4 readlanes plus selects (measured).
template <int GW>
__device__ __forceinline__ double ff_bcast(double v, int gbase, int i) {
    if constexpr (GW == 64) {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        return __hiloint2double(__builtin_amdgcn_readlane(hi, i), __builtin_amdgcn_readlane(lo, i));
    } else {
        return __shfl(v, gbase + i, 64);
    }
}
template <int GW>
__device__ __forceinline__ double ff_wave_sum(double v) {          // xor butterfly: every lane of the group ends with the same bits
#pragma unroll
    for (int off = GW / 2; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}
template <int GW>
__device__ __forceinline__ double ff_wave_max(double v) {
#pragma unroll
    for (int off = GW / 2; off >= 1; off >>= 1) { const double o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

template <int NF, int GW>
struct FFLane {
    double cV[4], cA[4], cJ[4];
    double l1, l2, l3, invd;
    int lane, n, gbase;            // lane = index inside the group
    int nmax;                      // largest n among the wavefront's groups: a wave-uniform bound for the descending sweeps

    __device__ __forceinline__ void gx(double x, double &gV, double &gA, double &gJ) const {
        const double xm2 = ff_at<GW>(x, gbase, lane - 2), xm1 = ff_at<GW>(x, gbase, lane - 1), xp1 = ff_at<GW>(x, gbase, lane + 1);
        gV = ((cV[0] * xm2 + cV[1] * xm1) + cV[2] * x) + cV[3] * xp1;
        gA = ((cA[0] * xm2 + cA[1] * xm1) + cA[2] * x) + cA[3] * xp1;
        gJ = ((cJ[0] * xm2 + cJ[1] * xm1) + cJ[2] * x) + cJ[3] * xp1;
    }
    // G' u for per-row values u (inactive rows hold 0)
    __device__ __forceinline__ double gt(const double (&u)[NF]) const {
        const double wV = u[0] - u[1], wA = u[2] - u[3], wJ = u[4] - u[5];
        double t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = (cV[k] * wV + cA[k] * wA) + cJ[k] * wJ;
        const double a0 = ff_at<GW>(t[0], gbase, lane + 2), a1 = ff_at<GW>(t[1], gbase, lane + 1), a3 = ff_at<GW>(t[3], gbase, lane - 1);
        const double bnd = (NF == 8) ? (u[NF - 1] - u[NF - 2]) : 0.0;
        return (((a0 + a1) + t[2]) + a3) + bnd;
    }
    // L D L' of S = 2 I + G' diag(D) G, rows in order
    __device__ __forceinline__ void factor(const double (&D)[NF]) {
        const double DV = D[0] + D[1], DA = D[2] + D[3], DJ = D[4] + D[5];
        double T[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int l = 0; l <= k; ++l)
                T[k][l] = (((DV * cV[k]) * cV[l]) + ((DA * cA[k]) * cA[l])) + ((DJ * cJ[k]) * cJ[l]);
        double S0 = 0.0, S1 = 0.0, S2 = 0.0, S3 = 0.0;
        S0 = S0 + ff_at<GW>(T[0][0], gbase, lane + 2); S0 = S0 + ff_at<GW>(T[1][1], gbase, lane + 1); S0 = S0 + T[2][2]; S0 = S0 + ff_at<GW>(T[3][3], gbase, lane - 1);
        S1 = S1 + ff_at<GW>(T[1][0], gbase, lane + 1); S1 = S1 + T[2][1]; S1 = S1 + ff_at<GW>(T[3][2], gbase, lane - 1);
        S2 = S2 + T[2][0]; S2 = S2 + ff_at<GW>(T[3][1], gbase, lane - 1);
        S3 = S3 + ff_at<GW>(T[3][0], gbase, lane - 1);
        const double bd = (NF == 8) ? (D[NF - 2] + D[NF - 1]) : 0.0;
        S0 = (2.0 + S0) + bd;
        l1 = 0.0; l2 = 0.0; l3 = 0.0; invd = 0.5;         // lanes >= n: S = 2 I
        for (int i = 0; i < n; ++i) {
            const double l1_m2 = (i >= 2) ? ff_bcast<GW>(l1, gbase, i - 2) : 0.0;
            const double l1_m1 = (i >= 1) ? ff_bcast<GW>(l1, gbase, i - 1) : 0.0;
            const double l2_m1 = (i >= 1) ? ff_bcast<GW>(l2, gbase, i - 1) : 0.0;
            const double id_m1 = (i >= 1) ? ff_bcast<GW>(invd, gbase, i - 1) : 0.0;
            const double id_m2 = (i >= 2) ? ff_bcast<GW>(invd, gbase, i - 2) : 0.0;
            const double id_m3 = (i >= 3) ? ff_bcast<GW>(invd, gbase, i - 3) : 0.0;
            const double e3 = S3;
            const double e2 = S2 - e3 * l1_m2;
            const double e1 = (S1 - e3 * l2_m1) - e2 * l1_m1;
            const double n3 = e3 * id_m3, n2 = e2 * id_m2, n1 = e1 * id_m1;
            const double d = ((S0 - e3 * n3) - e2 * n2) - e1 * n1;
            if (lane == i) { l1 = n1; l2 = n2; l3 = n3; invd = 1.0 / d; }
        }
    }
    // S u = rhs for two right-hand sides in one pair of lane-serial sweeps (same arithmetic per right-hand side)
    __device__ __forceinline__ void solve2(double rhsA, double rhsB, double &outA, double &outB) const {
        double yA = 0.0, yB = 0.0;
        for (int i = 0; i < n; ++i) {
            const double a1 = (i >= 1) ? ff_bcast<GW>(yA, gbase, i - 1) : 0.0, b1 = (i >= 1) ? ff_bcast<GW>(yB, gbase, i - 1) : 0.0;
            const double a2 = (i >= 2) ? ff_bcast<GW>(yA, gbase, i - 2) : 0.0, b2 = (i >= 2) ? ff_bcast<GW>(yB, gbase, i - 2) : 0.0;
            const double a3 = (i >= 3) ? ff_bcast<GW>(yA, gbase, i - 3) : 0.0, b3 = (i >= 3) ? ff_bcast<GW>(yB, gbase, i - 3) : 0.0;
            const double vA = ((rhsA - l1 * a1) - l2 * a2) - l3 * a3;
            const double vB = ((rhsB - l1 * b1) - l2 * b2) - l3 * b3;
            if (lane == i) { yA = vA; yB = vB; }
        }
        yA = yA * invd; yB = yB * invd;
        if (lane >= n) { yA = 0.0; yB = 0.0; }
        const double u1 = ff_at<GW>(l1, gbase, lane + 1), u2 = ff_at<GW>(l2, gbase, lane + 2), u3 = ff_at<GW>(l3, gbase, lane + 3);
        double uA = 0.0, uB = 0.0;
        for (int i = nmax - 1; i >= 0; --i) {      // uniform i (v_readlane needs it); rows i >= n stay 0
            const double a1 = (i + 1 < GW) ? ff_bcast<GW>(uA, gbase, i + 1) : 0.0, b1 = (i + 1 < GW) ? ff_bcast<GW>(uB, gbase, i + 1) : 0.0;
            const double a2 = (i + 2 < GW) ? ff_bcast<GW>(uA, gbase, i + 2) : 0.0, b2 = (i + 2 < GW) ? ff_bcast<GW>(uB, gbase, i + 2) : 0.0;
            const double a3 = (i + 3 < GW) ? ff_bcast<GW>(uA, gbase, i + 3) : 0.0, b3 = (i + 3 < GW) ? ff_bcast<GW>(uB, gbase, i + 3) : 0.0;
            const double vA = ((yA - u1 * a1) - u2 * a2) - u3 * a3;
            const double vB = ((yB - u1 * b1) - u2 * b2) - u3 * b3;
            if (lane == i) { uA = vA; uB = vB; }
        }
        outA = uA; outB = uB;
    }
    __device__ __forceinline__ double solve(double rhs) const {
        double y = 0.0;
        for (int i = 0; i < n; ++i) {
            const double y1 = (i >= 1) ? ff_bcast<GW>(y, gbase, i - 1) : 0.0;
            const double y2 = (i >= 2) ? ff_bcast<GW>(y, gbase, i - 2) : 0.0;
            const double y3 = (i >= 3) ? ff_bcast<GW>(y, gbase, i - 3) : 0.0;
            const double v = ((rhs - l1 * y1) - l2 * y2) - l3 * y3;
            if (lane == i) y = v;
        }
        y = y * invd;
        if (lane >= n) y = 0.0;
        const double u1 = ff_at<GW>(l1, gbase, lane + 1), u2 = ff_at<GW>(l2, gbase, lane + 2), u3 = ff_at<GW>(l3, gbase, lane + 3);
        double u = 0.0;
        for (int i = nmax - 1; i >= 0; --i) {
            const double x1 = (i + 1 < GW) ? ff_bcast<GW>(u, gbase, i + 1) : 0.0;
            const double x2 = (i + 2 < GW) ? ff_bcast<GW>(u, gbase, i + 2) : 0.0;
            const double x3 = (i + 3 < GW) ? ff_bcast<GW>(u, gbase, i + 3) : 0.0;
            const double v = ((y - u1 * x1) - u2 * x2) - u3 * x3;
            if (lane == i) u = v;
        }
        return u;
    }
};

// numpy.interp on the coarse path held one sample per lane (scipy interp1d linear delegates to it), st.py:597-598
__device__ __forceinline__ double ff_interp_lane(double sc, int gbase, int len, double cdt, double x) {
    const double t_last = (double)(len - 1) * cdt;
    int j = (int)(x / cdt);
    j = j < 0 ? 0 : (j > len - 2 ? len - 2 : j);
#pragma unroll
    for (int it = 0; it < 2; ++it) { if (j + 1 <= len - 2 && (double)(j + 1) * cdt <= x) ++j; }
#pragma unroll
    for (int it = 0; it < 2; ++it) { if (j > 0 && (double)j * cdt > x) --j; }
    const double f0 = __shfl(sc, gbase + j, 64), f1 = __shfl(sc, gbase + j + 1, 64), fl = __shfl(sc, gbase + len - 1, 64);
    const double t0 = (double)j * cdt, t1 = (double)(j + 1) * cdt;
    if (x >= t_last) return fl;
    if (t0 == x) return f0;
    const double slope = (f1 - f0) / (t1 - t0);
    return slope * (x - t0) + f0;
}

template <int NF, int GW>
__global__ void __launch_bounds__(64) k_finer_fit(FFArgs a) {
    const int lane = threadIdx.x & (GW - 1), gbase = threadIdx.x - lane;     // lane: index inside the problem's group
    const int e = blockIdx.x * (64 / GW) + threadIdx.x / GW;
    if (e >= a.N) return;
    const FFConst &k = a.k;
    // ---- coarse path, one sample per lane
    int len;
    double sc = 0.0, v0, a0;
    if (a.path_idx) {
        len = a.best_t[e] + 1;                                        // trailing zeros trimmed, st.py:762-768
        const double start_s = a.ego[(size_t)e * 5 + 4];
        const double delta = (start_s + a.ds) - start_s;              // numpy arange fill: start + n * delta
        if (lane < len) sc = start_s + (double)a.path_idx[(size_t)e * a.Hs + lane] * delta;
        v0 = a.ego[(size_t)e * 5 + 2]; a0 = a.ego[(size_t)e * 5 + 3];
    } else {
        len = a.len[e];
        if (lane < len) sc = a.s_seq[(size_t)e * a.Hs + lane];
        v0 = a.v0[e]; a0 = a.a0[e];
    }
    const double s_first = __shfl(sc, gbase, 64), s_second = __shfl(sc, gbase + 1, 64);   // shuffles stay outside divergent code
    if (len <= 1 || !a.use_qp) {                                      // st.py:587-588 / st.py:771, then st.py:774-783
        if (a.out && lane < len && lane < a.n_max) a.out[(size_t)e * a.n_max + lane] = sc;
        if (lane == 0) {
            if (a.out_len) a.out_len[e] = len;
            if (a.iters) a.iters[e] = 0;
            if (a.speed) a.speed[e] = (len <= 1) ? v0 : (s_second - s_first) / k.dt;
        }
        return;
    }
    // ---- fine grid length, st.py:590-595
    const double t_last = (double)(len - 1) * k.cdt;
    int n = (int)rint(t_last / k.dt + 1.0);
    if ((double)(n - 1) * k.dt > t_last) n -= 1;
    if (n < 2 || n > GW) {
        if (lane == 0) {
            if (a.out_len) a.out_len[e] = -1; if (a.iters) a.iters[e] = 0; if (a.speed) a.speed[e] = __builtin_nan("");
            if (a.refused) atomicOr(a.refused, 1u);
        }
        return;
    }
    const double bi_all = ff_interp_lane(sc, gbase, len, k.cdt, (double)lane * k.dt);   // all lanes: the shuffles inside need them
    const double bi = (lane < n) ? bi_all : 0.0;
    const double qv = -2.0 * bi;
    const double beq = s_first;

    FFLane<NF, GW> L;
    L.lane = lane; L.n = n; L.gbase = gbase;
    {
        // largest n among the groups of this wavefront that are still here (a group whose path had a single point has
        // returned: its lanes are inactive and neither hold nor forward anything, so no butterfly -- read each active
        // group's n directly; v_readlane ignores EXEC, the ballot says which groups count)
        const unsigned long long alive = __ballot(true);
        int nm = 0;
#pragma unroll
        for (int g = 0; g < 64 / GW; ++g) {
            const int ng = __builtin_amdgcn_readlane(n, g * GW);
            if ((alive >> (g * GW)) & 1ull) nm = ng > nm ? ng : nm;
        }
        L.nmax = nm;
    }
    const int r = lane;
    const bool row = r < n - 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) { L.cV[c] = 0.0; L.cA[c] = 0.0; L.cJ[c] = 0.0; }
    if (row) {                                                        // st.py:609-660
        L.cV[2] = k.cv; L.cV[3] = -k.cv;
        if (r == 0) { L.cA[2] = -k.ca1; L.cA[3] = k.ca1; }
        else { L.cA[1] = k.ca1; L.cA[2] = -k.ca2; L.cA[3] = k.ca1; }
        if (r == 0) { L.cJ[2] = -k.cj1; L.cJ[3] = k.cj1; }
        else if (r == 1) { L.cJ[1] = k.cj2; L.cJ[2] = -k.cj3; L.cJ[3] = k.cj1; }
        else { L.cJ[0] = -k.cj1; L.cJ[1] = k.cj3; L.cJ[2] = -k.cj3; L.cJ[3] = k.cj1; }
    }
    bool act[NF];
    double h[NF];
#pragma unroll
    for (int f = 0; f < 6; ++f) act[f] = row;
    h[0] = 0.0; h[1] = k.v_max;                                       // st.py:617-622
    h[2] = (r == 0) ? k.a_max + v0 / k.dt : k.a_max;                  // st.py:629,634
    h[3] = (r == 0) ? -k.a_min - v0 / k.dt : -k.a_min;                // st.py:639-641
    h[4] = (r == 0) ? k.j_max + a0 / k.dt + v0 / k.dt2 : ((r == 1) ? k.j_max - v0 / k.dt2 : k.j_max);       // st.py:648,653,659
    h[5] = (r == 0) ? -k.j_min - a0 / k.dt - v0 / k.dt2 : ((r == 1) ? -k.j_min + v0 / k.dt2 : -k.j_min);    // st.py:664-668
    if constexpr (NF == 8) {                                          // st.py:670-702
        act[6] = false; act[7] = false; h[6] = 0.0; h[7] = 0.0;
        if (a.bac && lane < n) {
            const double before_s = a.bac[(size_t)e * 4 + 0], before_v = a.bac[(size_t)e * 4 + 1];
            const double after_s = a.bac[(size_t)e * 4 + 2], after_v = a.bac[(size_t)e * 4 + 3];
            const double ti = (double)lane * k.dt;
            if (!isinf(before_s)) { const double pr = before_s + ti * before_v; if (!(pr < -k.car_length)) { act[6] = true; h[6] = -pr - k.car_length; } }
            if (!isinf(after_s)) { const double pr = after_s + ti * after_v; if (!(pr < -k.car_length)) { act[7] = true; h[7] = pr - k.car_length; } }
        }
    }
    int mloc = 0;
#pragma unroll
    for (int f = 0; f < NF; ++f) { if (!act[f]) h[f] = 0.0; mloc += act[f] ? 1 : 0; }
    int m = mloc;
#pragma unroll
    for (int off = GW / 2; off >= 1; off >>= 1) m += __shfl_xor(m, off, 64);

    const double STEP = 0.99, ABSTOL = 1e-7, RELTOL = 1e-6, FEASTOL = 1e-7;   // cvxopt defaults
    double resx0 = sqrt(ff_wave_sum<GW>(qv * qv)); resx0 = resx0 > 1.0 ? resx0 : 1.0;
    const double resy0 = fabs(beq) > 1.0 ? fabs(beq) : 1.0;
    double hh = 0.0;
#pragma unroll
    for (int f = 0; f < NF; ++f) hh = hh + h[f] * h[f];
    double resz0 = sqrt(ff_wave_sum<GW>(hh)); resz0 = resz0 > 1.0 ? resz0 : 1.0;
    const double e0 = (lane == 0) ? 1.0 : 0.0;

    // ---- starting point: [P A' G'; A 0 0; G 0 -I] [x; y; z] = [-q; b; h], s = -z, shifted into the cone
    double D[NF], uu[NF], s[NF], z[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) { D[f] = act[f] ? 1.0 : 0.0; uu[f] = h[f]; }
    L.factor(D);
    double tmp = L.gt(uu);
    tmp = (lane < n) ? (-qv + tmp) : 0.0;
    double u, v;
    L.solve2(tmp, e0, u, v);
    double y = (__shfl(u, gbase, 64) - beq) / __shfl(v, gbase, 64);
    double x = (lane < n) ? u - v * y : 0.0;
    {
        double gV, gA, gJ;
        L.gx(x, gV, gA, gJ);
        double g[NF];
        g[0] = gV; g[1] = -gV; g[2] = gA; g[3] = -gA; g[4] = gJ; g[5] = -gJ;
        if constexpr (NF == 8) { g[6] = -x; g[7] = x; }
        double nn = 0.0, ms = -__builtin_inf(), mz = -__builtin_inf();
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (act[f]) {
                z[f] = g[f] - h[f]; s[f] = -z[f];
                nn = nn + s[f] * s[f];
                ms = -s[f] > ms ? -s[f] : ms; mz = -z[f] > mz ? -z[f] : mz;
            } else { s[f] = 1.0; z[f] = 0.0; }
        }
        const double nrm = sqrt(ff_wave_sum<GW>(nn));
        const double ts = ff_wave_max<GW>(ms), tz = ff_wave_max<GW>(mz);
        const double thr = -1e-8 * (nrm > 1.0 ? nrm : 1.0);
#pragma unroll
        for (int f = 0; f < NF; ++f) if (act[f]) {
            if (ts >= thr) s[f] = s[f] + (1.0 + ts);
            if (tz >= thr) z[f] = z[f] + (1.0 + tz);
        }
    }
    double gp = 0.0;
#pragma unroll
    for (int f = 0; f < NF; ++f) if (act[f]) gp = gp + s[f] * z[f];
    double gap = ff_wave_sum<GW>(gp);

    int iters = 0;
    bool converged = false;
    for (iters = 0; iters <= k.maxiters; ++iters) {
        // residuals: rx = P x + q + A' y + G' z, ry = A x - b, rz = s + G x - h
        double rz[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) uu[f] = act[f] ? z[f] : 0.0;
        tmp = L.gt(uu);
        const double px = 2.0 * x + qv;
        const double f0p = x * px + x * qv;
        const double rx = (lane < n) ? (px + (lane == 0 ? y : 0.0)) + tmp : 0.0;
        const double f0 = 0.5 * ff_wave_sum<GW>(f0p);
        const double resx = sqrt(ff_wave_sum<GW>(rx * rx));
        const double ry = __shfl(x, gbase, 64) - beq;
        const double resy = fabs(ry);
        double rzn = 0.0, rzz = 0.0;
        {
            double gV, gA, gJ;
            L.gx(x, gV, gA, gJ);
            double g[NF];
            g[0] = gV; g[1] = -gV; g[2] = gA; g[3] = -gA; g[4] = gJ; g[5] = -gJ;
            if constexpr (NF == 8) { g[6] = -x; g[7] = x; }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                rz[f] = act[f] ? (s[f] - h[f]) + g[f] : 0.0;
                rzn = rzn + rz[f] * rz[f];
                rzz = rzz + (act[f] ? z[f] * rz[f] : 0.0);
            }
        }
        const double resz = sqrt(ff_wave_sum<GW>(rzn));
        const double pcost = f0, dcost = ((f0 + y * ry) + ff_wave_sum<GW>(rzz)) - gap;
        bool have_rel = false;
        double relgap = 0.0;
        if (pcost < 0.0) { relgap = gap / -pcost; have_rel = true; }
        else if (dcost > 0.0) { relgap = gap / dcost; have_rel = true; }
        const double pres = resy / resy0 > resz / resz0 ? resy / resy0 : resz / resz0;
        const double dres = resx / resx0;
        if (pres <= FEASTOL && dres <= FEASTOL && (gap <= ABSTOL || (have_rel && relgap <= RELTOL))) { converged = true; break; }
        if (iters == k.maxiters) break;

#pragma unroll
        for (int f = 0; f < NF; ++f) D[f] = act[f] ? z[f] / s[f] : 0.0;
        L.factor(D);
        double v_0 = 0.0;
        const double mu = gap / (double)m;
        double sigma = 0.0, step = 1.0, dx = 0.0, dy = 0.0;
        double ds[NF], dz[NF], dsdz_a[NF];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (!act[f]) { uu[f] = 0.0; continue; }
                double bs = -(s[f] * z[f]);
                if (pass == 1) bs = (bs - dsdz_a[f]) + sigma * mu;
                uu[f] = (bs + z[f] * rz[f]) / s[f];
            }
            tmp = L.gt(uu);
            tmp = (lane < n) ? -rx - tmp : 0.0;
            if (pass == 0) { L.solve2(tmp, e0, u, v); v_0 = __shfl(v, gbase, 64); }     // v = S^-1 A' rides along with the predictor
            else u = L.solve(tmp);
            dy = (__shfl(u, gbase, 64) + ry) / v_0;
            dx = (lane < n) ? u - v * dy : 0.0;
            double gV, gA, gJ;
            L.gx(dx, gV, gA, gJ);
            double g[NF];
            g[0] = gV; g[1] = -gV; g[2] = gA; g[3] = -gA; g[4] = gJ; g[5] = -gJ;
            if constexpr (NF == 8) { g[6] = -dx; g[7] = dx; }
            double pd = 0.0, ms = -__builtin_inf(), mz = -__builtin_inf();
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (!act[f]) { ds[f] = 0.0; dz[f] = 0.0; continue; }
                dz[f] = uu[f] + D[f] * g[f];
                ds[f] = -rz[f] - g[f];
                pd = pd + ds[f] * dz[f];
                const double qs = -ds[f] / s[f], qz = -dz[f] / z[f];
                ms = qs > ms ? qs : ms; mz = qz > mz ? qz : mz;
            }
            const double dsdz = ff_wave_sum<GW>(pd);
            double t = ff_wave_max<GW>(ms);
            const double tz = ff_wave_max<GW>(mz);
            t = t > tz ? t : tz; t = t > 0.0 ? t : 0.0;
            if (t == 0.0) step = 1.0;
            else if (pass == 0) step = 1.0 / t < 1.0 ? 1.0 / t : 1.0;
            else step = STEP / t < 1.0 ? STEP / t : 1.0;
            if (pass == 0) {
                double c = (1.0 - step) + (dsdz / gap) * (step * step);
                c = c > 0.0 ? c : 0.0; c = c < 1.0 ? c : 1.0;
                sigma = (c * c) * c;
#pragma unroll
                for (int f = 0; f < NF; ++f) dsdz_a[f] = ds[f] * dz[f];
            }
        }
        x = x + step * dx;
        y = y + step * dy;
        gp = 0.0;
#pragma unroll
        for (int f = 0; f < NF; ++f) if (act[f]) {
            s[f] = s[f] + step * ds[f];
            z[f] = z[f] + step * dz[f];
            gp = gp + s[f] * z[f];
        }
        gap = ff_wave_sum<GW>(gp);
    }
    if (a.out && lane < n && lane < a.n_max) a.out[(size_t)e * a.n_max + lane] = x;
    const double x1 = __shfl(x, gbase + 1, 64), x0 = __shfl(x, gbase, 64);
    if (lane == 0) {
        if (a.out_len) a.out_len[e] = n;
        if (a.iters) a.iters[e] = converged ? iters : -iters;
        if (a.speed) a.speed[e] = (x1 - x0) / k.dt;                   // st.py:780-781
    }
}

}  // namespace stmpc
