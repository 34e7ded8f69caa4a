// stmpc_nj_kernels.hpp -- the reference's two non-production lattice solvers:
//   st_cy.solve_s_t_path_no_jerk_fast      st_cy.pyx:209-312   node (t, s)
//   st_cy.solve_s_t_path_no_jerk_djikstra  st_cy.pyx:96-206    node (t, s, s_prev)
// No shipped configuration selects them (USE_FAST_ST_SOLVER = False), they exist for API completeness on small lattices:
// the (t, s, s_prev) form needs S^2 bytes per layer.  Their result depends on the heap's LIFO tie rule (entry_order counts
// down, st_cy.pyx:131-132), so the search itself is the reference's: one WAVEFRONT per problem, lane 0 owns the binary heap
// (in HBM scratch), the 64 lanes evaluate the candidate cells of a popped node side by side and lane 0 pushes the survivors
// in ascending cell order.  Module constants st_cy.pyx:21-31.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stmpc {

struct NjItem { double cost; long long order; unsigned short t, s, prev, second; };      // 24 bytes

struct NjArgs {
    int triple, S, H;
    double v0;
    const uint8_t *obstacles;     // [H][S]
    const double *distances;      // [H][S]
    const double *s_values;       // [S]
    double dt;
    uint8_t *enc;                 // [H][S] or [H][S][S], zeroed
    int *prev;                    // same shape
    NjItem *heap;
    unsigned long long cap;
    double *s_sequence;           // [H]
    int *status;                  // 0 ok, 1 heap capacity exceeded, 2 seeding range leaves the grid (the reference raises IndexError)
};

__device__ __forceinline__ double nj_cost(double s, double s1, double s2, double dt, double d) {      // st_cy.pyx:34-44
    const double v = (s - s1) / dt;
    const double a = (s - 2 * s1 + s2) / (dt * dt);
    const double pen = d < 7.5 ? 1000000.0 / (d > 1.0 ? d : 1.0) : 1 / d;
    return 0.5 * ((v - 30.0) * (v - 30.0)) + 1.0 * (a * a) + 1000.0 * pen;
}
__device__ __forceinline__ void nj_range(double s, double prev_s, double dt, double start_s, double ds, int &lo, int &hi) {   // st_cy.pyx:56-62,78-93
    const double v = (s - prev_s) / dt;
    double min_v = v + -6.0 * dt; if (!(min_v > 0)) min_v = 0;
    double max_v = v + 4.5 * dt; if (!(max_v < 40.0)) max_v = 40.0;
    const double min_s = s + min_v * dt, max_s = s + max_v * dt;
    const double x = (min_s - start_s) / ds;
    int mi = (int)x; const int ma = (int)((max_s - start_s) / ds);
    if (mi < x) mi += 1;
    lo = mi; hi = ma + 1;
}
__device__ __forceinline__ bool nj_less(const NjItem &x, const NjItem &y) { return x.cost < y.cost || (x.cost == y.cost && x.order < y.order); }

__global__ void __launch_bounds__(64) k_nojerk(NjArgs a) {
    const int lane = threadIdx.x;
    const int S = a.S, H = a.H;
    const double ds = a.s_values[1] - a.s_values[0], dt = a.dt, start_s = a.s_values[0];
    const double est_prev = start_s - a.v0 * dt;
    NjItem *hp = a.heap;
    unsigned long long hn = 0;            // heap size (lane 0's copy is authoritative; broadcast where needed)
    long long order = 0;
    auto push = [&](const NjItem &it) -> bool {       // lane 0 only
        if (hn >= a.cap) return false;
        unsigned long long i = hn++;
        while (i > 0) { const unsigned long long p = (i - 1) >> 1; const NjItem pv = hp[p]; if (!nj_less(it, pv)) break; hp[i] = pv; i = p; }
        hp[i] = it;
        return true;
    };
    auto pop = [&]() -> NjItem {                       // lane 0 only
        const NjItem top = hp[0], last = hp[--hn];
        unsigned long long i = 0;
        for (;;) {
            unsigned long long c = 2 * i + 1;
            if (c >= hn) break;
            NjItem cv = hp[c];
            if (c + 1 < hn) { const NjItem c2 = hp[c + 1]; if (nj_less(c2, cv)) { cv = c2; c++; } }
            if (!nj_less(cv, last)) break;
            hp[i] = cv; i = c;
        }
        if (hn) hp[i] = last;
        return top;
    };
    auto at_of = [&](int t, int s, int prev) -> size_t { return a.triple ? ((size_t)t * S + s) * S + prev : (size_t)t * S + s; };
    // expansion of one node: the wave evaluates the cells [lo, hi) 64 at a time, lane 0 pushes the valid ones in ascending order
    int status = 0;
    auto expand = [&](int nt, double sv, double pv, double base_cost, int src, int srcprev, int lo, int hi, bool check_enc) {
        for (int b = lo; b < hi && b < S; b += 64) {
            const int n = b + lane;
            bool ok = n < hi && n < S;
            double c = 0.0;
            if (ok) {
                if (check_enc && a.enc[at_of(nt, n, src)]) ok = false;
                else if (a.obstacles[(size_t)nt * S + n]) ok = false;
                else c = base_cost + nj_cost(a.s_values[n], sv, pv, dt, a.distances[(size_t)nt * S + n]);
            }
            unsigned long long m = __ballot(ok);
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const double cl = __longlong_as_double(((long long)__builtin_amdgcn_readlane(__double2hiint(c), l) << 32) | (unsigned)__builtin_amdgcn_readlane(__double2loint(c), l));
                if (lane == 0) {
                    NjItem it; it.cost = cl; it.order = order; it.t = (unsigned short)nt; it.s = (unsigned short)(b + l); it.prev = (unsigned short)src; it.second = (unsigned short)srcprev;
                    if (!push(it)) status = 1;
                }
                order -= 1;
            }
        }
    };
    int lo, hi;
    nj_range(start_s, est_prev, dt, start_s, ds, lo, hi);
    if (hi > S || lo < 0) { if (lane == 0) *a.status = 2; return; }
    expand(1, start_s, est_prev, 0.0, 0, 0, lo, hi, false);                  // st_cy.pyx:123-129 / 236-242 (no encountered test when seeding)
    int best_t = 0, best_s = 0, best_p = 0;
    for (;;) {
        // lane 0 pops until it finds a node that is not settled yet; the node is broadcast to the wave
        int t = -1, s = 0, pr = 0, sec = 0;
        double cst = 0.0;
        if (lane == 0) {
            while (hn > 0) {
                const NjItem it = pop();
                const size_t at = at_of(it.t, it.s, it.prev);
                if (a.enc[at]) continue;
                a.enc[at] = 1; a.prev[at] = a.triple ? (int)it.second : (int)it.prev;
                t = it.t; s = it.s; pr = it.prev; sec = it.second; cst = it.cost;
                break;
            }
        }
        __threadfence_block();
        t = __builtin_amdgcn_readfirstlane(t);
        if (t < 0 || __builtin_amdgcn_readfirstlane(status)) break;
        s = __builtin_amdgcn_readfirstlane(s); pr = __builtin_amdgcn_readfirstlane(pr); sec = __builtin_amdgcn_readfirstlane(sec);
        cst = __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane(__double2hiint(cst)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(cst)));
        if (t == H - 1) { best_t = H - 1; best_s = s; best_p = pr; break; }
        else if (t > best_t) { best_t = t; best_s = s; best_p = pr; }
        const double sv = a.s_values[s], pv = a.s_values[pr];
        nj_range(sv, pv, dt, start_s, ds, lo, hi);
        expand(t + 1, sv, pv, cst, s, pr, lo, hi, true);
    }
    if (lane == 0) {
        *a.status = status;
        for (int t = 0; t < H; ++t) a.s_sequence[t] = 0.0;
        if (a.triple) {                                                      // st_cy.pyx:193-204
            int bs = best_s, bp = best_p;
            for (int t = best_t; t > 1; --t) { a.s_sequence[t] = a.s_values[bs]; const int second = a.prev[at_of(t, bs, bp)]; bs = bp; bp = second; }
            a.s_sequence[0] = a.s_values[bp];
            if (H > 1) a.s_sequence[1] = a.s_values[bs];
        } else {                                                             // st_cy.pyx:303-310
            int bs = best_s;
            for (int t = best_t; t > 0; --t) { a.s_sequence[t] = a.s_values[bs]; bs = a.prev[(size_t)t * S + bs]; }
            a.s_sequence[0] = a.s_values[bs];
        }
    }
}

}  // namespace stmpc
