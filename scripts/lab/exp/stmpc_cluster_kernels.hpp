// stmpc_cluster_kernels.hpp -- several compute units for ONE heavy search (round 5).
//
// The searches that end a step at N = 4096 are the dozen second-window episodes under a bound of 1e6 and more: every reachable cell outside the
// penalty zones is expanded, ~3000 sources x 21 candidates per layer, 1.1-1.2 ms of fp64 issue on the one unit a search can use -- while 200
// units idle.  k_cluster spreads such a search over G workgroups on G units ("a cluster") with the exact pass in PULL form:
//
//   layer t + 1, cell n:  cost[n] = min over the sources i of layer t whose range [lo_i, hi_i) holds n of  C_i + edge(i -> n),
//
// sources taken in ascending order with a strict comparison, so that among equal totals the smallest predecessor wins -- the reference's heap
// order (cost, predecessor), st_cy.pyx:355-388 -- without atomics and without the first-setter and tie stages of the push form (dp_pass).  The
// thread that owns cell n then treats it as a source of the next layer at once (bound test, back-pointer, range st_cy.pyx:65-93), so a layer costs
// ONE barrier across the cluster (1.0-1.3 us, scripts/lab/exp/cluster_barrier.hip).  Cells, histories and ranges of a layer live in global
// memory (three small arrays per cluster, written once and read by the neighbours through the cluster's L2: its workgroups sit on one XCD); each
// workgroup stages the window of sources its 512 cells can be reached from into LDS.  The arithmetic of a range and of an edge is dp_pass's,
// statement by statement; the candidate filter of the push form is not needed (it only drops offers whose total exceeds the bound: here such a
// cell is simply reached-but-not-expanded, and raises the same "pruned" flag).
//
// The kernel consumes its own queue -- the searches that leave the first window with the most left to do (SolveArgs::cluster_thr) and the first
// window's repeated passes -- alongside the ordinary second-window launch, always from layer 0.  Whatever it does not get to is picked up by
// the ordinary launch that follows.
#pragma once
#include "stmpc_kernels.hpp"

namespace stmpc {

constexpr int CL_THREADS = 512;
constexpr int CL_GMAX = 16;

struct ClusterCtrl {                       // one per cluster (global memory)
    unsigned bar;                          // barrier arrivals so far
    int ep;                                // the episode the leader has claimed (-1: none left)
    int dead;                              // a member gave up waiting: everybody leaves
    unsigned flags;                        // bit 0: a reached cell was not expanded (the bound cut something)
    int pad0[60];
};
struct ClusterLayer {                      // statistics of one layer of the search in progress (global memory, [H] per cluster)
    int nodes, tlo, thi, pad;              // cells expanded; range of next-layer cells their candidates can touch (min lo, max hi)
    unsigned long long best_bits[CL_GMAX]; // per member: cheapest expanded cell ...
    int best_n[CL_GMAX];                   // ... and the smallest such cell
};

struct ClusterArgs {
    int G;                                 // workgroups per cluster
    int SP;                                // cells per row of the scratch arrays (>= S of every episode)
    unsigned char *scratch;                // [clusters][stride]: C[2][SP] u64, HK[2][SP] u32, RG[2][SP] u32, BP[H][SP] u8, ClusterLayer[H]
    size_t stride;
    size_t layer_off;                      // offset of the ClusterLayer array inside a cluster's scratch
    ClusterCtrl *ctrl;                     // [clusters]
};

// cluster c = workgroups {b : b % 8 == c % 8, b / (8 G) == c / 8}: all on one XCD as workgroups are dealt today (speed only: nothing depends on it)
__device__ __forceinline__ void cluster_of(int b, int G, int &c, int &m) { c = (b / (8 * G)) * 8 + (b & 7); m = (b >> 3) % G; }

template <bool FASTDIV>
__global__ void __launch_bounds__(CL_THREADS, 4) k_cluster(SolveArgs a, ClusterArgs ca) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_smem[];
    __shared__ int sh_i[12];               // 1 nodes, 2 tlo, 3 thi, 4 flags, 5 dead, 6-8 the layer's totals after the barrier
    __shared__ unsigned long long sh_best[CL_GMAX];
    __shared__ int sh_bestn[CL_GMAX];
    __shared__ int sh_path[STMPC_MAXH];
    const DevP &p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = ca.G, SP = ca.SP, H = p.H;
    int cl, m;
    cluster_of((int)blockIdx.x, G, cl, m);
    ClusterCtrl *ctl = ca.ctrl + cl;
    unsigned char *sc = ca.scratch + (size_t)cl * ca.stride;
    u64 *Cg = (u64 *)sc;                                   // [2][SP]
    unsigned *HKg = (unsigned *)(Cg + 2 * (size_t)SP);     // [2][SP]
    unsigned *RGg = HKg + 2 * (size_t)SP;                  // [2][SP]
    unsigned char *BPg = (unsigned char *)(RGg + 2 * (size_t)SP);   // [H][SP]
    ClusterLayer *LS = (ClusterLayer *)(sc + ca.layer_off);         // [H]
    const int MSW = (a.maxshift + 63) & ~63;               // sources of cell n lie in [n - MSW, n]
    const int NWD = (MSW + 1 + 31) >> 5;                   // words of a cell's mask of offering sources
    // LDS: window of sources [MSW + CL_THREADS] x (cost 8, history 4, range 4); masks [CL_THREADS][NWD]
    u64 *Cl = (u64 *)cl_smem;
    unsigned *HKl = (unsigned *)(Cl + (MSW + CL_THREADS));
    unsigned *RGl = HKl + (MSW + CL_THREADS);
    unsigned *HM = RGl + (MSW + CL_THREADS);
    unsigned round = 0;                                    // barriers passed (thread 0)
#ifdef STMPC_CL_DEBUG
    unsigned long long dbg_bar = 0, dbg_ep = 0, dbg_t0 = 0, dbg_first = 0, dbg_last = 0, dbg_ph[6] = {0, 0, 0, 0, 0, 0}, dbg_tp = 0; int dbg_n = 0;
#define CLPH(k) do { if (m == 0 && tid == 0) { const unsigned long long t_ = wall_clock64(); dbg_ph[k] += t_ - dbg_tp; dbg_tp = t_; } } while (0)
#else
#define CLPH(k) do { } while (0)
#endif
    const double dt = p.dt, dt2 = p.dt2, dt3 = p.dt3;
    const double zl_dt = a.zl_dt, zl_dt2 = a.zl_dt2, zl_dt3 = a.zl_dt3;

    // Barrier across the cluster; false if somebody gave up (then everybody leaves).  The hand-off recipe of the gfx950 notes: plain payload
    // stores, every wave drains them, one agent-scope release by lane 0, a relaxed arrival; relaxed polls, ONE agent-scope acquire (it
    // invalidates this unit's L1), plain loads after the workgroup barrier.
    auto cbar = [&]() -> bool {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            ++round;
#ifdef STMPC_CL_DEBUG
            const unsigned long long tb0 = wall_clock64();
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&ctl->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = wall_clock64();
            int dead = 0;
            while (__hip_atomic_load(&ctl->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G * round) {
                if (__hip_atomic_load(&ctl->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { dead = 1; break; }
                if (wall_clock64() - t0 > a.wait_ticks) {
                    __hip_atomic_store(&ctl->dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicExch(&a.counters[STMPC_CNT_ERR], 1u);
                    dead = 1; break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#ifdef STMPC_CL_DEBUG
            if (dbg_t0) dbg_bar += wall_clock64() - tb0;
#endif
            sh_i[5] = dead;
        }
        __syncthreads();
        return sh_i[5] == 0;
    };
    auto ldi = [&](const int *q) -> int { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

    // the clusters' queue (see k_solve: what leaves the first window with the most left to do, and its repeated passes)
    auto claim0 = [&]() -> int {
        unsigned *consumed = &a.counters[STMPC_CNT_CLQ + 1];
        unsigned *total = &a.counters[STMPC_CNT_CLQ];
        for (;;) {
            const unsigned c = __hip_atomic_load(consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned t = __hip_atomic_load(total, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (c >= t) return -1;
            if (atomicCAS(consumed, c, c + 1u) != c) continue;
            int *slot = &a.cl_queue[c];
            int e_;
            const unsigned long long t_spin = wall_clock64();
            do { e_ = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while (e_ < 0 && wall_clock64() - t_spin <= a.wait_ticks);
            if (e_ < 0) atomicExch(&a.counters[STMPC_CNT_ERR], 1u);
            return e_;
        }
    };
    const unsigned long long t_begin = wall_clock64();
    bool may_wait = false;
    if (m == 0 && tid == 0) may_wait = __hip_atomic_load(&a.counters[STMPC_CNT_RESIDENT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)a.prev_grid;

    for (;;) {
        // ---- next episode: the leader claims it and publishes it
        if (m == 0 && tid == 0) {
            int e = -1;
            for (;;) {
                e = claim0();
                if (e >= 0 || !may_wait) break;
                if (__hip_atomic_load(&a.counters[STMPC_CNT_FINISHED], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)a.prev_grid) { e = claim0(); break; }
                if (wall_clock64() - t_begin > a.wait_ticks / 10) break;
                __builtin_amdgcn_s_sleep(64);
            }
            __hip_atomic_store(&ctl->ep, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!cbar()) return;
        const int e = __builtin_amdgcn_readfirstlane(ldi(&ctl->ep));
#ifdef STMPC_CL_DEBUG
        if (e < 0 && m == 0 && tid == 0 && dbg_n)
            printf("cluster %2d: %d episodes, %.0f us in them, %.0f us of that in barriers; staging %.0f masks %.0f cells %.0f fold %.0f barrier %.0f totals %.0f\n", cl, dbg_n,
                   dbg_ep * 0.01, dbg_bar * 0.01, dbg_ph[0] * 0.01, dbg_ph[1] * 0.01, dbg_ph[2] * 0.01, dbg_ph[3] * 0.01, dbg_ph[4] * 0.01, dbg_ph[5] * 0.01);
        if (m == 0 && tid == 0) { dbg_t0 = wall_clock64(); if (!dbg_first) dbg_first = dbg_t0; }
#endif
        if (e < 0) return;

        // ---- the episode (solve_episode's set-up)
        const double start_s = a.ego[(size_t)e * 5 + 4], v0 = a.ego[(size_t)e * 5 + 2], a0 = a.ego[(size_t)e * 5 + 3];
        const double s1 = start_s + p.ds, delta = s1 - start_s;
        const int S = a.tab.num_s[e];
        const double r_dt = 1.0 / dt, r_dt2 = 1.0 / dt2, r_dt3 = 1.0 / dt3, r_delta = 1.0 / delta;
        const double est_prev = start_s - v0 * dt;                              // st_cy.pyx:329-330
        const double est_second = est_prev - dt * (v0 - a0 * dt);
        const bool s1_plain = (start_s + 1.0 * delta == s1);
        auto sval = [&](int n) -> double {
            double v = start_s + (double)n * delta;
            if (!s1_plain) { if (n == 1) v = s1; }
            return v;
        };
        u64 ubits = INF_BITS;
        if (a.prune) { const u64 ub = a.ubound[e]; if (ub != 0ull) ubits = ub; }

        int best_t = 0, best_n = 0;
        u64 best_bits = 0ull;
        for (int attempt = 0;; ++attempt) {
            // the layers' statistics of this attempt (leader workgroup, one thread per layer)
            if (m == 0) {
                if (tid < H) {
                    ClusterLayer *L = LS + tid;
                    L->nodes = 0; L->tlo = 0x7fffffff; L->thi = 0; L->pad = 0;
                    for (int g = 0; g < CL_GMAX; ++g) { L->best_bits[g] = ~0ull; L->best_n[g] = 0x7fffffff; }
                }
                if (tid == 0) __hip_atomic_store(&ctl->flags, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!cbar()) return;

            // A cell of layer `tt` that has its value: expand it as a source of the next layer (what dp_pass's scan and source stage do).
            auto as_source = [&](int tt, int n, u64 cb, unsigned key, int &acc_nodes, int &acc_lo, int &acc_hi, bool &acc_pruned,
                                 u64 &my_best, int &my_best_n) -> unsigned {
                const bool reached = cb < INF_BITS;
                const bool act = reached && cb <= ubits;
                if (reached && !act) acc_pruned = true;
                if (!act) return 0u;
                acc_nodes += 1;
                if (cb < my_best || (cb == my_best && n < my_best_n)) { my_best = cb; my_best_n = n; }
                const int pr = (int)(key >> 16), pp = (int)(key & 0xFFFFu);
                if (tt > 0) BPg[(size_t)tt * SP + n] = (unsigned char)(n - pr);
                if (tt >= H - 1) return 0u;
                const double sv = sval(n);
                double p1, p2;
                if (tt == 0) { p1 = est_prev; p2 = est_second; }                 // st_cy.pyx:342
                else { p1 = sval(pr); p2 = (tt == 1) ? est_prev : sval(pp); }
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
                int lo = mi, hi = ma + 1;
                if (hi > S) hi = S;                                              // st_cy.pyx:379
                if (lo < n) lo = n;
                if (lo >= hi) return 0u;
                acc_lo = lo < acc_lo ? lo : acc_lo; acc_hi = hi > acc_hi ? hi : acc_hi;
                return (unsigned)lo | ((unsigned)hi << 16);
            };
            // fold a workgroup's statistics of layer tt into the cluster's
            auto fold = [&](int tt, int acc_nodes, int acc_lo, int acc_hi, bool acc_pruned, u64 my_best, int my_best_n) {
                if (tid == 0) { sh_i[1] = 0; sh_i[2] = 0x7fffffff; sh_i[3] = 0; sh_i[4] = 0; }
                __syncthreads();
                const u64 wb = wave_min_u64(my_best);
                const int wbn = wave_min_i(my_best == wb ? my_best_n : 0x7fffffff);
                if (lane == 0) { sh_best[wave] = wb; sh_bestn[wave] = wbn; }
                // (one LDS atomic per WAVE: hundreds of lanes on one address serialise)
                int wn = acc_nodes;
                for (int off = 1; off < 64; off <<= 1) wn += __shfl_xor(wn, off);
                const int wlo_ = wave_min_i(acc_hi > acc_lo ? acc_lo : 0x7fffffff), whi_ = wave_max_i(acc_hi > acc_lo ? acc_hi : 0);
                const bool wpr = __ballot(acc_pruned) != 0ull;
                if (lane == 0) {
                    if (wn) atomicAdd(&sh_i[1], wn);
                    if (whi_ > wlo_) { atomicMin(&sh_i[2], wlo_); atomicMax(&sh_i[3], whi_); }
                    if (wpr) atomicOr(&sh_i[4], 1);
                }
                __syncthreads();
                if (tid == 0) {
                    ClusterLayer *L = LS + tt;
                    u64 bb = ~0ull; int bn = 0x7fffffff;
                    for (int w = 0; w < CL_THREADS / 64; ++w) { if (sh_best[w] < bb || (sh_best[w] == bb && sh_bestn[w] < bn)) { bb = sh_best[w]; bn = sh_bestn[w]; } }
                    __hip_atomic_store(&L->best_bits[m], bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&L->best_n[m], bn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (sh_i[1]) __hip_atomic_fetch_add(&L->nodes, sh_i[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (sh_i[3] > sh_i[2]) {
                        __hip_atomic_fetch_min(&L->tlo, sh_i[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_max(&L->thi, sh_i[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (sh_i[4]) __hip_atomic_fetch_or(&ctl->flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            // after the barrier: a layer's totals, read once per workgroup (sc1 loads: the words were updated by atomics)
            auto totals = [&](int tt, int &nn, int &tlo, int &thi, u64 &bb, int &bn) {
                const ClusterLayer *L = LS + tt;
                if (tid < G) {
                    sh_best[tid] = __hip_atomic_load(&L->best_bits[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sh_bestn[tid] = ldi(&L->best_n[tid]);
                }
                if (tid == 64) { sh_i[6] = ldi(&L->nodes); sh_i[7] = ldi(&L->tlo); sh_i[8] = ldi(&L->thi); }
                __syncthreads();
                nn = sh_i[6]; tlo = sh_i[7]; thi = sh_i[8];
                bb = ~0ull; bn = 0x7fffffff;
                for (int g = 0; g < G; ++g) { if (sh_best[g] < bb || (sh_best[g] == bb && sh_bestn[g] < bn)) { bb = sh_best[g]; bn = sh_bestn[g]; } }
                __syncthreads();
            };

            // ---- layer 0: the start cell
            {
                int acc_nodes = 0, acc_lo = 0x7fffffff, acc_hi = 0; bool acc_pruned = false; u64 mb = ~0ull; int mbn = 0x7fffffff;
                if (m == 0 && tid == 0) {
                    const unsigned rg = as_source(0, 0, 0ull, 0u, acc_nodes, acc_lo, acc_hi, acc_pruned, mb, mbn);
                    Cg[0] = 0ull; HKg[0] = 0u; RGg[0] = rg;
                }
                fold(0, acc_nodes, acc_lo, acc_hi, acc_pruned, mb, mbn);
            }
            if (!cbar()) return;
            int slo = 0, shi = 1;                  // span of layer t's cells
            int last_t = -1, total_nodes = 0;
            u64 lb = ~0ull; int lbn = 0x7fffffff;  // best node of the deepest non-empty layer
            int tlo = 0, thi = 0;                  // cells of layer t + 1 that layer t's sources can touch
            {
                int nn; u64 bb; int bn;
                totals(0, nn, tlo, thi, bb, bn);
                if (nn > 0) { last_t = 0; total_nodes = nn; lb = bb; lbn = bn; }
            }
            for (int t = 0; t < H - 1 && last_t == t; ++t) {
                int acc_nodes = 0, acc_lo = 0x7fffffff, acc_hi = 0; bool acc_pruned = false; u64 mb = ~0ull; int mbn = 0x7fffffff;
                if (thi > tlo) {
                    // vehicle rows of layer t + 1 (uniform)
                    const size_t row = (size_t)e * H + (t + 1);
                    const int nact = as_const(a.tab.nact)[row];
                    stmpc_cdouble *cedge = as_const(a.tab.edge) + row * a.Kmax * 2;
                    stmpc_cint *cwin = as_const(a.tab.win) + row * a.Kmax * 2;
                    const u64 *Cs = Cg + (size_t)(t & 1) * SP; const unsigned *HKs = HKg + (size_t)(t & 1) * SP, *RGs = RGg + (size_t)(t & 1) * SP;
                    u64 *Cd = Cg + (size_t)((t + 1) & 1) * SP; unsigned *HKd = HKg + (size_t)((t + 1) & 1) * SP, *RGd = RGg + (size_t)((t + 1) & 1) * SP;
                    const int t64 = tlo & ~63;
                    // lanes per cell: a layer narrower than the cluster spreads the offers of a cell over 2 or 4 adjacent lanes (each walks the
                    // cell's mask and takes every LPC-th offer; the group's minimum by (cost, source) follows) -- the offers of a cell are a
                    // dependent fp64 chain, and in a bounded layer most lanes would idle
                    const int span = thi - t64;
                    const int lsh = (span * 4 <= G * CL_THREADS) ? 2 : ((span * 2 <= G * CL_THREADS) ? 1 : 0);
                    const int LPC = 1 << lsh, cpb = CL_THREADS >> lsh;            // cells per workgroup and pass
                    const int cell = tid >> lsh, sub = tid & (LPC - 1);
                    for (int base = t64 + m * cpb; base < thi; base += G * cpb) {
                        // sources this workgroup's cells [base, base + cpb) can be reached from: [base - MSW, base + cpb) within the layer's span
                        const int w0 = base - MSW;
                        __syncthreads();
                        CLPH(5);
                        for (int x = tid; x < MSW + cpb; x += CL_THREADS) {
                            const int i = w0 + x;
                            const bool in = i >= slo && i < shi;
                            Cl[x] = in ? Cs[i] : INF_BITS;
                            HKl[x] = in ? HKs[i] : 0u;
                            RGl[x] = in ? RGs[i] : 0u;
                        }
                        for (int w = tid; w < cpb * NWD; w += CL_THREADS) HM[w] = 0u;
                        __syncthreads();
                        CLPH(0);
                        // which sources offer which cell: every source marks itself in the masks of the cells of its range (bit = its
                        // distance from the low end of the cell's window, so ascending bits are ascending sources)
                        for (int x = tid; x < MSW + cpb; x += CL_THREADS) {
                            const unsigned rg = RGl[x];
                            int lo = (int)(rg & 0xFFFFu), hi = (int)(rg >> 16);
                            if (lo < base) lo = base;
                            if (hi > base + cpb) hi = base + cpb;
                            const int i = w0 + x;
                            for (int n = lo; n < hi; ++n) {
                                const int bit = i - n + MSW;                        // in [0, MSW]: i <= n <= i + maxshift
                                atomicOr(&HM[(n - base) * NWD + (bit >> 5)], 1u << (bit & 31));
                            }
                        }
                        __syncthreads();
                        CLPH(1);
                        const int n = base + cell;
                        const bool mine = n >= tlo && n < thi;
                        u64 bestv = INF_BITS; unsigned bestkey = 0u;
                        if (mine) {
                            // penalty of cell n in layer t + 1 (dp_pass::cell_penalty, table form)
                            const double sn = sval(n);
                            double d = 1e10;                                         // st.py:34-35
                            bool blocked = false;
                            for (int c = 0; c < nact; ++c) {
                                d = __builtin_fmin(d, fabs(sn - cedge[c * 2 + 0]));
                                d = __builtin_fmin(d, fabs(sn - cedge[c * 2 + 1]));
                                blocked |= (n >= cwin[c * 2 + 0]) & (n < cwin[c * 2 + 1]);
                            }
                            if (!blocked) {
                                const double pn = dev_weighted_penalty(d, p.min_allowed, p.d_w);
                                int k = 0;
                                for (int w = 0; w < NWD; ++w) {
                                    unsigned mw = HM[cell * NWD + w];
                                    while (mw) {
                                        const int b = __builtin_ctz(mw);
                                        mw &= mw - 1u;
                                        if (((k++) & (LPC - 1)) != sub) continue;
                                        const int x = cell + w * 32 + b, i = w0 + x;    // source i = (n - MSW) + bit
                                        const u64 cb = Cl[x];
                                        const unsigned h = HKl[x];
                                        const double C = __longlong_as_double((long long)cb);
                                        const double sv = sval(i);
                                        double p1, p2; unsigned key;
                                        if (t == 0) { p1 = est_prev; p2 = est_second; key = 0u; }
                                        else {
                                            const int pr = (int)(h >> 16), pp = (int)(h & 0xFFFFu);
                                            p1 = sval(pr); p2 = (t == 1) ? est_prev : sval(pp);
                                            key = ((unsigned)i << 16) | (unsigned)pr;
                                        }
                                        const double two_sv = 2 * sv, three_sv = 3 * sv, three_p1 = 3 * p1;
                                        // st_cy.pyx:46-50 cost_with_jerk(next, s, p1, p2)
                                        const double v = divk<FASTDIV>(sn - sv, dt, r_dt, zl_dt);
                                        const double aa = divk<FASTDIV>(sn - two_sv + p1, dt2, r_dt2, zl_dt2);
                                        const double jj = divk<FASTDIV>(sn - three_sv + three_p1 - p2, dt3, r_dt3, zl_dt3);
                                        const double dv = v - p.v_des;
                                        const double ec = p.v_w * (dv * dv) + p.a_w * (aa * aa) + p.j_w * (jj * jj) + pn;
                                        const double tot = C + ec;                       // st_cy.pyx:388
                                        const u64 tb = (u64)__double_as_longlong(tot);
                                        if (tb < bestv) { bestv = tb; bestkey = key; }
                                    }
                                }
                            }
                        }
                        // the group's minimum: cost first, then the smaller source (the key's high half) -- the reference's heap order
                        for (int off = 1; off < LPC; off <<= 1) {
                            const unsigned olo = __shfl_xor((unsigned)bestv, off), ohi = __shfl_xor((unsigned)(bestv >> 32), off);
                            const unsigned okey = __shfl_xor(bestkey, off);
                            const u64 ov = ((u64)ohi << 32) | olo;
                            if (ov < bestv || (ov == bestv && okey < bestkey)) { bestv = ov; bestkey = okey; }
                        }
                        if (mine && sub == 0) {
                            const unsigned rgn = as_source(t + 1, n, bestv, bestkey, acc_nodes, acc_lo, acc_hi, acc_pruned, mb, mbn);
                            Cd[n] = bestv; HKd[n] = bestkey; RGd[n] = rgn;
                        }
                    }
                }
                CLPH(2);
                fold(t + 1, acc_nodes, acc_lo, acc_hi, acc_pruned, mb, mbn);
                CLPH(3);
                if (!cbar()) return;
                CLPH(4);
                slo = tlo; shi = thi;
                int nn; u64 bb; int bn;
                totals(t + 1, nn, tlo, thi, bb, bn);
                if (nn > 0) { last_t = t + 1; total_nodes += nn; lb = bb; lbn = bn; }
            }
            const bool pruned = (__hip_atomic_load(&ctl->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) != 0u;
            best_t = last_t < 0 ? 0 : last_t; best_n = lbn; best_bits = lb;
            if (m == 0 && tid == 0) atomicAdd(&a.counters[STMPC_CNT_NODES_EXACT], (unsigned)total_nodes);
            if (best_t == H - 1 || !pruned) break;
            // the bound was below the reference's terminal cost: relax it (solve_episode's ladder)
            if (attempt >= 3) ubits = INF_BITS;
            else ubits = (u64)__double_as_longlong(__longlong_as_double((long long)ubits) * (attempt == 0 ? a.retry_mult[0] : (attempt == 1 ? a.retry_mult[1] : a.retry_mult[2])));
            if (m == 0 && tid == 0) atomicAdd(&a.counters[STMPC_CNT_RETRY], 1u);
        }

        // ---- back-track (st_cy.pyx:391-398) and outputs (solve_episode's), by the leader workgroup
        if (m == 0) {
            if (tid == 0) {
                int n = best_n;
                for (int t = best_t; t > 0; --t) {
                    sh_path[t] = n;
                    n -= (int)BPg[(size_t)t * SP + n];
                }
                sh_path[0] = n;
            }
            __syncthreads();
            if (tid < 64) {
                bool crash_l = false;
                if (lane < H) {
                    const int t = lane;
                    const int n = (t <= best_t) ? sh_path[t] : -1;
                    double pd = __builtin_nan("");
                    if (n >= 0) {
                        const double s_t = sval(n);
                        const int qi = (int)((s_t - start_s) / delta);                    // st.py:798 -> st.py:20-22
                        const size_t row = (size_t)e * H + t;
                        const int na = a.tab.nact[row];
                        const double *ce = a.tab.edge + row * a.Kmax * 2;
                        const int *cw = a.tab.win + row * a.Kmax * 2;
                        const double sq = sval(qi);
                        double d = 1e10;
                        bool blocked = false;
                        for (int c = 0; c < na; ++c) {
                            const double f = fabs(sq - ce[c * 2 + 0]);
                            const double b = fabs(sq - ce[c * 2 + 1]);
                            d = (f < d) ? f : d; d = (b < d) ? b : d;
                            blocked |= (qi >= cw[c * 2 + 0]) & (qi < cw[c * 2 + 1]);
                        }
                        if (blocked) d = 0.0;
                        pd = d;
                        crash_l = d < p.crash_dist_thr;
                    }
                    a.path_idx[(size_t)e * H + t] = n;
                    if (a.path_dist) a.path_dist[(size_t)e * H + t] = pd;
                    if (t == 1 && a.action_cost) a.action_cost[(size_t)e * 2] = (double)n;
                }
                const bool any_crash = __ballot(crash_l) != 0ull;
                if (lane == 0) {
                    a.best_t[e] = best_t;
                    a.cost[e] = __longlong_as_double((long long)best_bits);
                    if (a.action_cost) a.action_cost[(size_t)e * 2 + 1] = __longlong_as_double((long long)best_bits);
                    if (a.crash) a.crash[e] = (best_t != H - 1 || any_crash) ? 1 : 0;
                }
            }
            __syncthreads();
        }
#ifdef STMPC_CL_DEBUG
        if (m == 0 && tid == 0) { dbg_last = wall_clock64(); dbg_ep += dbg_last - dbg_t0; dbg_n += 1; dbg_t0 = 0; }
#endif
        // (the next episode's first barrier orders the scratch arrays' reuse)
    }
}

}  // namespace stmpc
