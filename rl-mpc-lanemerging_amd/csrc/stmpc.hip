// stmpc.hip -- C-ABI (include/stmpc.h) over the gfx950 kernels in stmpc_kernels.hpp.
// Host side: context, device buffers, launch sequencing on the caller's HIP stream.
// No CPU fallback: every compute entry needs a HIP device.
#include "stmpc_kernels.hpp"
#include "stmpc_ff_kernels.hpp"
#include "stmpc_cc_kernels.hpp"
#include "stmpc_nj_kernels.hpp"
#include "stmpc_actor_kernels.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "stmpc.h"

using namespace stmpc;

namespace {

thread_local std::string g_err;
std::string g_info;

int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(_e == hipErrorOutOfMemory ? STMPC_ENOMEM : STMPC_EHIP,                     \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                        \
    } while (0)

// libm pow through a volatile pointer so clang cannot fold pow(x,2.0)/pow(x,3.0): the reference's
// Python evaluates float**int with libm pow (control.py:38) and Cython's dt**3 likewise (st_cy.pyx:49).
double (*volatile host_pow)(double, double) = pow;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return STMPC_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(STMPC_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
        cap = want;
        return STMPC_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

int next_pow2(int v) { int w = 1; while (w < v) w <<= 1; return w; }

// divc<true> needs RN(1/d) to be usable by Markstein's theorem: excludes divisors whose significand is all ones
bool fastdiv_ok(double d) {
    uint64_t b; memcpy(&b, &d, 8);
    return (d > 1e-100 && d < 1e100) && ((b & 0xFFFFFFFFFFFFFull) != 0xFFFFFFFFFFFFFull);
}

// Two-operation division by d (divk in stmpc_kernels.hpp): true, with *zl = RN(1/d - RN(1/d)), if fma(x, zh, x*zl) == x/d for
// EVERY double x whose quotient neither overflows nor falls into the subnormals.  The sequence can only round wrongly when x/d
// lies within 2^-52 ulp of a midpoint, i.e. |2^t X - (2M+1) D| < 8 D 2^-52 (four-fold safety) for the significands X of x and D of d
// (odd part, L bits), t in {L-1, L, L+1}: each residue r has at most a few X in [2^52, 2^53).  All of them, their neighbours, both
// signs and three binades are run through the real arithmetic here (the CPU's fma and division are IEEE like the GPU's).
// tests/div2_check.py replays the argument exhaustively in 8-10 bit formats.
bool fastdiv2_ok(double d, double *zl_out) {
    if (!(d > 1e-100 && d < 1e100)) return false;
    const double zh = 1.0 / d;
    const double zl = std::fma(-d, zh, 1.0) / d;
    *zl_out = zl;
    int e2;
    const double m = std::frexp(d, &e2);
    uint64_t D = (uint64_t)std::ldexp(m, 53);
    while ((D & 1) == 0) D >>= 1;
    auto two = [&](double x) { volatile double u1 = x * zl; return std::fma(x, zh, (double)u1); };
    auto good = [&](double x) { return two(x) == x / d && two(-x) == -x / d; };
    if (D > 1) {
        int L = 0; while ((D >> L) != 0) ++L;
        const long long R = (long long)(((unsigned __int128)D * 8) >> 52) + 1;
        for (int t = L - 1; t <= L + 1; ++t) {
            // 2^t mod D, then its inverse (extended Euclid)
            unsigned __int128 pw = 1; for (int k = 0; k < t; ++k) pw = (pw * 2) % D;
            long long a0 = (long long)D, a1 = (long long)(uint64_t)pw, x0 = 0, x1 = 1;
            while (a1 != 0) { const long long q = a0 / a1, a2 = a0 - q * a1, x2 = x0 - q * x1; a0 = a1; a1 = a2; x0 = x1; x1 = x2; }
            if (a0 != 1) return false;                               // (cannot happen: D is odd)
            const uint64_t inv = (uint64_t)((x0 % (long long)D + (long long)D) % (long long)D);
            for (long long r = -R; r <= R; ++r) {
                if (r == 0) continue;
                const uint64_t rm = (uint64_t)(((r % (long long)D) + (long long)D) % (long long)D);
                uint64_t X = (uint64_t)(((unsigned __int128)rm * inv) % D);
                const uint64_t lo = 1ull << 52, hi = 1ull << 53;
                if (X < lo) X += ((lo - X + D - 1) / D) * D;
                int n = 0;
                for (; X < hi; X += D) {
                    if (++n > 4096) return false;                    // too many close calls to try: use ordinary division
                    for (int dx = -1; dx <= 1; ++dx) {
                        const double xb = (double)(X + dx);
                        if (!good(xb) || !good(std::ldexp(xb, -40)) || !good(std::ldexp(xb, 30))) return false;
                    }
                }
            }
        }
    }
    // (not part of the argument: a smoke test of the arithmetic on ordinary values)
    uint64_t st_ = 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < 4096; ++k) {
        st_ ^= st_ << 13; st_ ^= st_ >> 7; st_ ^= st_ << 17;
        const double x = (double)(st_ >> 11) * (1.0 / 9007199254740992.0) * 2.0e4 - 1.0e4;
        if (two(x) != x / d) return false;
    }
    return true;
}

}  // namespace

struct stmpc_ctx {
    int device = 0;
    int num_cu = 256;
    int lds_per_block = 65536;
    // scratch
    DevBuf tab_edge, tab_win, tab_nact, tab_nums, counters, lists, ubound, proxy, order, gscratch, bp_tier[STMPC_MAX_TIERS];
    // staging for the host-pointer API
    DevBuf s_ego, s_k, s_ox, s_ov, s_path, s_bt, s_cost, s_pd, s_crash, s_misc0, s_misc1, s_misc2, s_misc3;
    DevBuf ckpt, pool_bp, resume_t, phase_prof, prio_key;
    int pool_cap_override = 0;     // STMPC_POOL=n: checkpoint pool entries (tests: a tiny pool must only cost speed)
    // combined controller (stmpc_rollout_step_device / stmpc_combined_decide_device): rollout bookkeeping and probe / controller outputs
    DevBuf cc_live, cc_hist_len, cc_crash_pred, cc_have_test, cc_sel, cc_rollout_s, cc_test_ego, cc_test_ox, cc_test_ov, cc_probe_ego, cc_probe_ox, cc_probe_ov,
        cc_path, cc_bt, cc_cost, cc_pcrash, cc_speed, cc_fine, cc_fine_len, cc_err,
        cc_sel_idx, cc_sel_count, cc_c_ego, cc_c_k, cc_c_ox, cc_c_ov, cc_c_speed, cc_c_fine, cc_c_fine_len;      // sparse controller solve: the states that need st.do_st_control
    int cc_N = 0, cc_K = 0, cc_R = 0;
    int *cc_host_count = nullptr;  // pinned host word for the number of those states
    int64_t cc_ticks = 0, cc_control_solves = 0;      // decisions taken / controller solves run for them (stmpc_combined_counts)
    // batched episode simulator (stmpc_sim_*)
    DevBuf sim_ego, sim_nveh, sim_vx, sim_vv, sim_va, sim_vc, sim_delay, sim_status, sim_ticks, sim_rng, sim_acc, sim_route;
    int sim_N = 0, sim_route_n = 0;
    DevBuf f_seq, f_len, f_v0, f_a0, f_bac, f_out, f_olen, f_iters, f_speed;   // finer_fit / st_control staging
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    stmpc_stats stats{};
    bool stats_pending = false;
    // profiling pool: one (begin, dp-begin, dp-end, end) event quad per launch while enabled
    bool profiling = false;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;          // events used (multiple of 4)
    double acc_solve_ms = 0, acc_dp_ms = 0;
    int64_t acc_launches = 0, acc_fallback = 0, acc_episodes = 0;
    int lds_tier_W[STMPC_MAX_TIERS] = {2048, 4096, 8192, 0, 0, 0};   // LDS windows (cells), increasing
    int n_lds_tiers = 0;          // 0 = automatic: {2048, smallest window covering every cell (<= 8192)}
    int pen_cells[STMPC_MAX_TIERS] = {0, 0, 0, 0, 0, 0};   // STMPC_PEN_CELLS="a,b,c": penalty-buffer cells per LDS tier (0 = min(W, 4096))
    int max_waves_per_cu = 16;
    int lds_headroom = 1024;       // STMPC_LDS_HEADROOM: bytes added to a workgroup's dynamic LDS when counting workgroups per CU
    int waves_override = 0;       // STMPC_NW=n or "a,b,c": waves per workgroup (episode), all tiers or per LDS tier
    int waves_tier[STMPC_MAX_TIERS] = {0, 0, 0, 0, 0, 0};
    bool tiers_from_env = false;
    bool allow_fastdiv = true;
    double fd2_dt = 0, fd2_dt2 = 0, fd2_dt3 = 0, fd2_zl[3] = {0, 0, 0}; bool fd2_ok = false;      // fastdiv2_ok results for the current dt
    int prune = -1;               // -1 auto (bounded search only when the fan-out is large), 0 off, 1 on
    double band_override = 0.0;
    int band_dense = 1;            // STMPC_BAND_DENSE=0/1: dense ordinary bounding attempts (band_pass)
    int tube_dense = 1;            // STMPC_TUBE_DENSE=0/1: dense guided attempt (tube_pass)
    int band_cap = 450;            // STMPC_BAND_CAP: nodes per layer the pre-pass steers its band towards (0 = fixed band); 300 until the pre-pass moved to
                                   // packed single precision (round 3): with candidates at a fifth of their former cost a wider pre-pass pays for itself in
                                   // tighter bounds (10 state seeds at N=4096: 375-600 all within 2 % of each other and 5 % ahead of 300)
    double band2_mult = 0.0;       // STMPC_BAND2_MULT (0 = default: 4 with the node cap, 5 with a fixed band)
    bool force_general = false;    // STMPC_FORCE_GENERAL=1 (tests)
    bool two_phase = false;        // STMPC_TWO_PHASE=1: bound all episodes first, then solve heaviest-first (measured 6 % slower at N=4096)
    bool allow_stage_tab = false;  // STMPC_STAGE_TAB=1: stage the vehicle table in LDS + scalar registers (costs the 4th workgroup per CU)
    int last_nt = 0;
    bool last_has_hbm = true;
    // STMPC_OVERLAP=0/1: start the second LDS tier on its own stream while the first is still running (see k_solve)
    bool resume = true;            // STMPC_RESUME=0/1: the wider window continues a checkpointed exact pass instead of starting over
    bool heavy_first = false;      // STMPC_HEAVY_FIRST=1: split tasks are handed out slow starters first (measured: 6.76-6.82 vs 6.81-6.82 ms at N=4096, 13.0 vs 12.2 ms at N=8192 -- long searches side by side slow each other down; off)
    int gsh_max = 4;               // STMPC_GSH=0..4: lanes per source of sparse layers, log2 (0 = one lane per source)
    bool split = true;             // STMPC_SPLIT=0/1: bounding and exact pass of an episode are separate tasks of the first launch (-4 % at N=4096)
    int overlap = -1;              // -1 auto: with the bounded (wide fan-out) search, where overflow is common
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // STMPC_CU_RESERVE=n (multiple of 8, experiment): n compute units are kept out of the first window's launch and host the second
    // window's workgroups from the start of the step (CU-masked streams); 0 = off
    int cu_reserve = 0;
    int side_grid = 0;             // STMPC_SIDE_GRID=n: workgroups of the second window's side launch (0 = automatic: the tier's full grid with the bounded
                                   // search, 32 on the narrow lattice where a handful of episodes overflow)
    int *h_overflow = nullptr, *d_overflow = nullptr;   // mapped pinned word: episodes that overflowed the first window in the batch before this one (stored by the
                                   // batch's last launch straight into host memory -- no copy, no stall --, read, possibly one batch late, when the next one
                                   // is set up): the narrow lattice starts its second window alongside the first only when there was something for it to do
    double last_infl = 1.005;      // STMPC_LAST_INFL: the exact pass's candidate filter lets terminals up to this factor above the bound through (SolveArgs::last_infl); 1 = off
    double bound_infl = 1.00002;   // STMPC_BOUND_INFL: factor on a bounding pass's single-precision path cost (>= 1.00002, the rounding of that total)
    int qp_maxiters = STMPC_QP_MAXITERS;   // STMPC_QP_ITERS (experiment: the iteration cap of st.do_st_control's QP; the reference's is 10, st.py:17)
    int tube_w = 96;               // STMPC_TUBE=w: half-width (cells) of the guided bounding attempt, 0 = off (see SolveArgs::guide_tab)
    // guide tables, one per parameter set (dynamics + cost weights), least recently used replaced: a caller that alternates parameter sets
    // (two controllers on one context) neither rebuilds nor waits
    struct GuideSlot { double key[12] = {0}; bool valid = false, ok = false; int imax = 0, D = 0; std::vector<unsigned char> host; DevBuf dev; uint64_t last_use = 0; };
    GuideSlot guides[4]; uint64_t guide_clock = 0;
    DevBuf guide_cells;
    int prio_thr = 32000;          // STMPC_PRIO=t (0 = off): an overflowing search with more than t (layers left x nodes of the saved layer) ahead of it is served first
                                   // by the second window (SolveArgs::prio_thr): 4.60 -> 4.46 ms over 12 seeds at N = 4096, flat from 25000 to 35000
    int prio_mode = 0;             // STMPC_PRIO_MODE (experiment: which estimate prio_thr is compared with)
    bool bp16 = false;             // STMPC_BP16=1: two-byte back-pointers even where one byte would do
    unsigned resume_refused_calls = 0;
    bool last_resume_refused = false;   // the last batch wanted checkpoint / resume and did not get it (stmpc_stats::resume_refused)
    int64_t last_hbm_tier_count = 0;    // episodes the last batch whose statistics were read sent to the clean-up tier
    size_t resume_refused_for = 0; // back-pointer bytes of the last request the quarter-of-free-memory rule turned down (not asked again until the request changes)
    int retry_move = 0;            // STMPC_RETRY_MOVE=k: see SolveArgs::retry_move
    double retry_mult[3] = {1.05, 1.3, 4.0};    // STMPC_RETRY="a,b,c": growth of a bound that turned out to be below the reference's terminal cost.  Round 2 grew gently
                                                // (1.02, 1.08, 1.3): most failures need less than 0.2 %, but the rare search that fails twice is three ever larger passes
                                                // in a row and ends the step; over 16 state seeds (1.05, 1.3, 4) has the same median and no 5.4-5.8 ms outliers
    int retire_cus = 0, retire_at = 75;   // STMPC_RETIRE_CUS=k, STMPC_RETIRE_AT=percent of N: k compute units leave the first launch once fewer than that many tasks are left (see SolveArgs::cu_tab)
    DevBuf cu_tab;
    DevBuf sticky;                 // [2] error flags that outlive a call: [0] solver internal error, [1] QP re-sampling refused a path (read and cleared by stmpc_check_error)
    hipStream_t main_masked = nullptr, aux_reserved = nullptr;
    hipEvent_t ev_join0 = nullptr, ev_join_r = nullptr;
};

extern "C" {

const char *stmpc_last_error(void) { return g_err.c_str(); }

#ifndef STMPC_SRC_HASH
#define STMPC_SRC_HASH "unknown"      /* build.py passes sha256[:16] of csrc/ + include/stmpc.h */
#endif
const char *stmpc_backend_info(void) {
    if (!g_info.empty()) return g_info.c_str();
    int n = 0;
    char buf[512];
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        g_info = "stmpc 0.1 hip (no device) src=" STMPC_SRC_HASH;
        return g_info.c_str();
    }
    hipDeviceProp_t pr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) { g_info = "stmpc 0.1 hip (device query failed) src=" STMPC_SRC_HASH; return g_info.c_str(); }
    snprintf(buf, sizeof buf, "stmpc 0.1 hip %s %s cu=%d lds=%zu devices=%d src=%s", pr.gcnArchName, pr.name,
             pr.multiProcessorCount, (size_t)pr.sharedMemPerBlock, n, STMPC_SRC_HASH);
    g_info = buf;
    return g_info.c_str();
}

int stmpc_create(stmpc_ctx **out, int device) {
    if (!out) return fail(STMPC_EINVAL, "stmpc_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(STMPC_ENODEV, "no HIP device available (stmpc has no CPU fallback)");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return fail(STMPC_ENODEV, "hipGetDevice failed"); }
    if (device >= n) return fail(STMPC_EINVAL, "device index out of range");
    HIPCHK(hipSetDevice(device));
    stmpc_ctx *c = new stmpc_ctx();
    c->device = device;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) == hipSuccess) {
        c->num_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
        c->lds_per_block = (int)pr.sharedMemPerBlock;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreate(&c->ev2) != hipSuccess || hipEventCreate(&c->ev3) != hipSuccess) {
        delete c;
        return fail(STMPC_EHIP, "hipEventCreate failed");
    }
    // experiment knobs: STMPC_TIERS="512,2048" (LDS windows), STMPC_WAVES_PER_CU, STMPC_FASTDIV=0
    if (const char *w = getenv("STMPC_TIERS")) {
        int n = 0; const char *q = w;
        while (*q && n < STMPC_MAX_TIERS - 1) {
            int v = atoi(q);
            if (v >= 64 && v <= 8192 && (v & (v - 1)) == 0 && (n == 0 || v > c->lds_tier_W[n - 1])) c->lds_tier_W[n++] = v;
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
        if (n > 0) c->n_lds_tiers = n;
        c->tiers_from_env = n > 0;
    }
    if (const char *w = getenv("STMPC_WAVES_PER_CU")) { int v = atoi(w); if (v >= 1 && v <= 32) c->max_waves_per_cu = v; }
    if (const char *w = getenv("STMPC_LDS_HEADROOM")) { int v = atoi(w); if (v >= 600 && v <= 8192) c->lds_headroom = v; }
    if (const char *w = getenv("STMPC_PEN_CELLS")) {
        int n = 0; const char *q = w;
        while (*q && n < STMPC_MAX_TIERS) {
            int v = atoi(q);
            if (v >= 128 && v <= 8192 && (v & (v - 1)) == 0) c->pen_cells[n] = v;
            ++n;
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    if (const char *w = getenv("STMPC_NW")) {
        if (strchr(w, ',')) {
            int n = 0; const char *q = w;
            while (*q && n < STMPC_MAX_TIERS) {
                int v = atoi(q);
                if (v >= 1 && v <= STMPC_MAXWAVES) c->waves_tier[n] = v;
                ++n;
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        } else { int v = atoi(w); if (v >= 1 && v <= STMPC_MAXWAVES) c->waves_override = v; }
    }
    if (const char *w = getenv("STMPC_FASTDIV")) c->allow_fastdiv = atoi(w) != 0;
    if (const char *w = getenv("STMPC_PRUNE")) c->prune = atoi(w) != 0 ? 1 : 0;
    if (const char *w = getenv("STMPC_BAND")) c->band_override = atof(w);
    if (const char *w = getenv("STMPC_FORCE_GENERAL")) c->force_general = atoi(w) != 0;
    if (const char *w = getenv("STMPC_TWO_PHASE")) c->two_phase = atoi(w) != 0;
    if (const char *w = getenv("STMPC_BAND2_MULT")) { double v = atof(w); if (v >= 1.0) c->band2_mult = v; }
    if (const char *w = getenv("STMPC_STAGE_TAB")) c->allow_stage_tab = atoi(w) != 0;
    if (const char *w = getenv("STMPC_OVERLAP")) c->overlap = atoi(w) != 0 ? 1 : 0;
    if (const char *w = getenv("STMPC_BAND_DENSE")) c->band_dense = atoi(w) != 0;
    if (const char *w = getenv("STMPC_TUBE_DENSE")) c->tube_dense = atoi(w) != 0;
    if (const char *w = getenv("STMPC_BAND_CAP")) { int v = atoi(w); if (v >= 0) c->band_cap = v; }
    if (const char *w = getenv("STMPC_SPLIT")) c->split = atoi(w) != 0;
    if (const char *w = getenv("STMPC_HEAVY_FIRST")) c->heavy_first = atoi(w) != 0;
    if (const char *w = getenv("STMPC_GSH")) { int v = atoi(w); if (v >= 0 && v <= 4) c->gsh_max = v; }
    if (const char *w = getenv("STMPC_RESUME")) c->resume = atoi(w) != 0;
    if (const char *w = getenv("STMPC_LAST_INFL")) { double v = atof(w); if (v >= 1.0 && v <= 4.0) c->last_infl = v; }
    if (const char *w = getenv("STMPC_BOUND_INFL")) { double v = atof(w); if (v >= 1.00002 && v <= 2.0) c->bound_infl = v; }
    if (const char *w = getenv("STMPC_POOL")) { int v = atoi(w); if (v >= 1) c->pool_cap_override = v; }
    if (const char *w = getenv("STMPC_QP_ITERS")) { int v = atoi(w); if (v >= 0 && v <= 1000) c->qp_maxiters = v; }
    // the side stream gets the highest priority: priority levels have their own hardware queues, so its launch
    // cannot end up queued behind the main stream's in a process that owns many streams (torch + RCCL)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess) {
        (void)hipGetLastError();
        c->aux_stream = nullptr;
        if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess) c->aux_stream = nullptr;
    }
    if (!c->aux_stream) c->overlap = 0;          // no side stream: the tiers simply run one after the other
    if (c->sticky.ensure(2 * sizeof(unsigned)) || hipMemset(c->sticky.p, 0, 2 * sizeof(unsigned)) != hipSuccess) { stmpc_destroy(c); return fail(STMPC_ENOMEM, "device allocation failed"); }
    if (const char *w = getenv("STMPC_TUBE")) { int v = atoi(w); if (v >= 0 && v <= 4096) c->tube_w = v; }
    if (const char *w = getenv("STMPC_PRIO")) { int v = atoi(w); if (v >= 0) c->prio_thr = v; }
    if (const char *w = getenv("STMPC_PRIO_MODE")) c->prio_mode = atoi(w);
    if (getenv("STMPC_BP16")) c->bp16 = true;
    if (const char *w = getenv("STMPC_RETRY_MOVE")) { int v = atoi(w); if (v >= 0 && v <= 4) c->retry_move = v; }
    if (const char *w = getenv("STMPC_RETRY")) { double x[3]; if (sscanf(w, "%lf,%lf,%lf", &x[0], &x[1], &x[2]) == 3 && x[0] > 1.0 && x[1] > 1.0 && x[2] > 1.0) for (int i = 0; i < 3; ++i) c->retry_mult[i] = x[i]; }
    if (const char *w = getenv("STMPC_RETIRE_CUS")) { int v = atoi(w); if (v >= 0 && v < 256) c->retire_cus = v; }
    if (const char *w = getenv("STMPC_RETIRE_AT")) { int v = atoi(w); if (v >= 1 && v <= 200) c->retire_at = v; }
    if (const char *w = getenv("STMPC_SIDE_GRID")) { int v = atoi(w); if (v >= 1) c->side_grid = v; }
    if (const char *w = getenv("STMPC_CU_RESERVE")) {
        // Reserved compute units: bit 32a + a + 8j (a = 0..7, j < n/8) of the CU mask.  Whether the driver numbers the mask bits
        // XCD by XCD or round-robin over the XCDs, every XCD gives up n/8 units and keeps the rest (a queue whose mask leaves an XCD
        // without units would never get the workgroups the dispatcher assigns to that XCD).
        int v = atoi(w);
        if (c->aux_stream && v >= 8 && v <= 128 && v % 8 == 0 && c->num_cu == 256) {
            uint32_t res[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rest[8];
            for (int a = 0; a < 8; ++a) for (int j = 0; j < v / 8; ++j) { const int bit = 32 * a + ((a + 8 * j) & 31); res[bit >> 5] |= 1u << (bit & 31); }
            for (int i = 0; i < 8; ++i) rest[i] = ~res[i];
            if (hipExtStreamCreateWithCUMask(&c->main_masked, 8, rest) == hipSuccess && hipExtStreamCreateWithCUMask(&c->aux_reserved, 8, res) == hipSuccess &&
                hipEventCreateWithFlags(&c->ev_join0, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_join_r, hipEventDisableTiming) == hipSuccess)
                c->cu_reserve = v;
            else (void)hipGetLastError();
        }
    }
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        stmpc_destroy(c);
        return fail(STMPC_EHIP, "stream/event creation failed");
    }
    *out = c;
    return STMPC_OK;
}

void stmpc_destroy(stmpc_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (auto &g : c->guides) g.dev.release();
    DevBuf *all[] = {&c->guide_cells, &c->sticky, &c->cu_tab, &c->tab_edge, &c->tab_win, &c->tab_nact, &c->tab_nums, &c->counters, &c->lists, &c->ubound, &c->proxy, &c->order, &c->bp_tier[0],
                     &c->bp_tier[1], &c->bp_tier[2], &c->bp_tier[3], &c->bp_tier[4], &c->bp_tier[5], &c->gscratch, &c->s_ego, &c->s_k, &c->s_ox, &c->s_ov, &c->s_path, &c->s_bt, &c->s_cost,
                     &c->s_pd, &c->s_crash, &c->s_misc0, &c->s_misc1, &c->s_misc2, &c->s_misc3,
                     &c->ckpt, &c->pool_bp, &c->resume_t, &c->phase_prof, &c->prio_key, &c->cc_live, &c->cc_hist_len, &c->cc_crash_pred, &c->cc_have_test, &c->cc_sel, &c->cc_rollout_s, &c->cc_test_ego,
                     &c->cc_test_ox, &c->cc_test_ov, &c->cc_probe_ego, &c->cc_probe_ox, &c->cc_probe_ov, &c->cc_path, &c->cc_bt, &c->cc_cost, &c->cc_pcrash, &c->cc_speed,
                     &c->cc_fine, &c->cc_fine_len, &c->cc_err, &c->sim_ego, &c->sim_nveh, &c->sim_vx, &c->sim_vv, &c->sim_va, &c->sim_vc, &c->sim_delay, &c->sim_status, &c->sim_ticks,
                     &c->sim_rng, &c->sim_acc, &c->sim_route, &c->f_seq, &c->f_len, &c->f_v0, &c->f_a0, &c->f_bac, &c->f_out, &c->f_olen, &c->f_iters, &c->f_speed,
                     &c->cc_sel_idx, &c->cc_sel_count, &c->cc_c_ego, &c->cc_c_k, &c->cc_c_ox, &c->cc_c_ov, &c->cc_c_speed, &c->cc_c_fine, &c->cc_c_fine_len};
    for (DevBuf *b : all) b->release();
    if (c->cc_host_count) (void)hipHostFree(c->cc_host_count);
    if (c->h_overflow) (void)hipHostFree(c->h_overflow);
    if (c->main_masked) (void)hipStreamDestroy(c->main_masked);
    if (c->aux_reserved) (void)hipStreamDestroy(c->aux_reserved);
    if (c->ev_join0) (void)hipEventDestroy(c->ev_join0);
    if (c->ev_join_r) (void)hipEventDestroy(c->ev_join_r);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev2) (void)hipEventDestroy(c->ev2);
    if (c->ev3) (void)hipEventDestroy(c->ev3);
    for (hipEvent_t ev : c->pool) (void)hipEventDestroy(ev);
    delete c;
}

// ---- host helpers ------------------------------------------------------------------------
double stmpc_ego_s(double x, double y) {
    // control.py:366-380 with control.distance (control.py:37-38): math.sqrt(dx**2 + dy**2)
    const double mpx = -50.9, mpy = 1.72, mp2x = 1.5, mp3x = -51.0;
    const double common_s = mp2x - mp3x;
    if (x < mpx) return -sqrt(host_pow(x - mpx, 2.0) + host_pow(y - mpy, 2.0));
    else if (x < mp2x) return sqrt(host_pow(x - mpx, 2.0) + host_pow(y - mpy, 2.0));
    else return x - mp2x + common_s;
}

int stmpc_num_s(const stmpc_params *p, double start_s) {
    if (!p) return STMPC_EINVAL;
    double stop = start_s + p->future_s + p->ds;
    return (int)ceil((stop - start_s) / p->ds);
}

int stmpc_num_t(const stmpc_params *p) {
    if (!p) return STMPC_EINVAL;
    double stop = p->future_t + p->dt;
    return (int)ceil((stop - 0.0) / p->dt);
}

double stmpc_path_mean_abs_jerk(const double *s, int n, double v0, double a0, double dt) {
    // st.py:274-288
    double prev_a = a0, prev_v = v0, acc = 0.0;
    for (int i = 1; i < n; ++i) {
        double v = (s[i] - s[i - 1]) / dt;
        double a = (v - prev_v) / dt;
        double j = (a - prev_a) / dt;
        prev_v = v; prev_a = a;
        acc += fabs(j);
    }
    return acc / (double)(n - 1);
}

}  // extern "C"

namespace {

// np.arange(0, future_t + dt, dt) as numpy fills it (st.py:32)
void host_t_values(const stmpc_params *p, int H, double *t) {
    if (H > 0) t[0] = 0.0;
    if (H > 1) t[1] = 0.0 + p->dt;
    if (H > 2) { double d = t[1] - t[0]; for (int i = 2; i < H; ++i) t[i] = 0.0 + (double)i * d; }
}

// Largest double q >= 0 with RN(sqrt(q)) <= m (-1 if there is none), and smallest double q >= 0 with RN(sqrt(q)) >= m (+inf if none).
// sqrt is correctly rounded on the host and on the device and monotone, so bisection over the bit patterns of the non-negative doubles is exact.
double (*volatile host_sqrt)(double) = sqrt;
double q_largest_sqrt_le(double m) {
    if (!(m >= 0.0)) return -1.0;
    uint64_t lo = 0, hi = 0x7FF0000000000000ull;            // invariant: sqrt(lo) <= m; answer in [lo, hi]
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        double q; memcpy(&q, &mid, 8);
        if (host_sqrt(q) <= m) lo = mid; else hi = mid - 1;
    }
    double q; memcpy(&q, &lo, 8);
    return q;
}
double q_smallest_sqrt_ge(double m) {
    if (!(m > 0.0)) return 0.0;                               // sqrt(0) = 0 >= m
    uint64_t lo = 0, hi = 0x7FF0000000000000ull;            // invariant: sqrt(hi) >= m (sqrt(inf) = inf)
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        double q; memcpy(&q, &mid, 8);
        if (host_sqrt(q) >= m) hi = mid; else lo = mid + 1;
    }
    double q; memcpy(&q, &hi, 8);
    return q;
}

int make_devp(const stmpc_params *p, DevP *d) {
    if (!p) return fail(STMPC_EINVAL, "params is NULL");
    if (!(p->ds > 0) || !(p->dt > 0)) return fail(STMPC_EINVAL, "ds and dt must be positive");
    int H = stmpc_num_t(p);
    if (H < 2 || H > STMPC_H_LIMIT) return fail(STMPC_EINVAL, "number of time layers must be in [2, 64]");
    memset(d, 0, sizeof *d);
    d->future_s = p->future_s; d->ds = p->ds;
    double tv[STMPC_MAXH];
    host_t_values(p, H, tv);
    d->dt = tv[1] - tv[0];                       // st_cy.pyx:319 delta_t = t_indices[1] - t_indices[0]
    d->dt2 = d->dt * d->dt;                      // delta_t**2 (compiled to x*x)
    d->dt3 = host_pow(d->dt, 3.0);               // delta_t**3 -> libm pow
    d->d_w = p->d_w; d->v_w = p->v_w; d->a_w = p->a_w; d->j_w = p->j_w; d->v_des = p->v_des; d->v_max = p->v_max;
    d->a_min = p->a_min; d->a_max = p->a_max; d->j_min = p->j_min; d->j_max = p->j_max; d->min_allowed = p->min_allowed;
    d->car_length = p->car_length;
    d->obst_min_s = p->crash_min_s - p->min_allowed;
    d->max_pred_decel = p->max_pred_decel; d->follow_gap = p->follow_gap; d->react_thr = p->react_thr;
    d->crash_thr = p->crash_thr; d->crash_dist_thr = p->comb_min_dist - p->car_length;
    // (see DevP) es = +sqrt(q): es > thr <=> q > largest q with sqrt(q) <= thr; es < thr <=> q < smallest q with sqrt(q) >= thr;
    //            es = -sqrt(q): es > thr <=> sqrt(q) < -thr;                    es < thr <=> sqrt(q) > -thr
    d->q_gt_pos = q_largest_sqrt_le(p->react_thr); d->q_lt_pos = q_smallest_sqrt_ge(p->react_thr);
    d->q_gt_neg = q_smallest_sqrt_ge(-p->react_thr); d->q_lt_neg = q_largest_sqrt_le(-p->react_thr);
    d->H = H;
    d->dlen = (int)(p->car_length / p->ds);      // st.py:37
    for (int t = 0; t < H; ++t) {
        d->unc[t] = p->start_unc + p->unc_per_s * tv[t];   // st.py:40
        d->dunc[t] = (int)(d->unc[t] / p->ds);             // st.py:41
    }
    return STMPC_OK;
}


// Table of the guided bounding attempt (SolveArgs::guide_tab): the optimal step sequence of the OBSTACLE-FREE problem from every lattice state
// (i1 = cells covered in the last layer, d = i1 - cells covered in the layer before), by backward dynamic programming over the true state
// (speed, acceleration) -- not the reference's history-collapsed search, whose answer it only approximates: the table centres a search, it
// never supplies a cost.  Ranges are st_cy.pyx:65-75 in cell units, shrunk by 1e-6 cell; costs st_cy.pyx:46-50 without the gap term.
// Returns false (no table: the attempt is skipped) for parameter sets whose state space does not fit a byte per step.
bool build_guide_table(const DevP &dp, std::vector<unsigned char> &tab, int &imax_out, int &D_out) {
    const int H = dp.H;
    const double ds = dp.ds, dt = dp.dt, u = ds / dt;
    if (H < 3 || !(ds > 0) || !(dt > 0)) return false;
    const int imax = (int)floor(dp.v_max * dt / ds + 1e-9);
    const int D = (int)ceil(fmax(fabs(dp.a_min), fabs(dp.a_max)) * dt * dt / ds) + 1;
    if (imax < 1 || imax > 254 || D > 60) return false;
    const int nd = 2 * D + 1, nS = (imax + 1) * nd, R = H - 1;
    std::vector<double> Fp((size_t)nS, 0.0), Fc((size_t)nS);
    std::vector<unsigned char> pol((size_t)(R + 1) * nS, 255);
    for (int r = 1; r <= R; ++r) {
        for (int i1 = 0; i1 <= imax; ++i1) for (int dd = -D; dd <= D; ++dd) {
            const int st = i1 * nd + dd + D, i2 = i1 - dd;
            Fc[st] = INFINITY;
            if (i2 < 0 || i2 > imax) continue;
            const double v = i1 * u, pv = i2 * u, a = (v - pv) / dt;
            const double lo_a = fmax(a + dp.j_min * dt, dp.a_min), hi_a = fmin(a + dp.j_max * dt, dp.a_max);
            const double lo_v = fmax(v + lo_a * dt, 0.0), hi_v = fmin(v + hi_a * dt, dp.v_max);
            int lo = (int)ceil(lo_v * dt / ds + 1e-6), hi = (int)floor(hi_v * dt / ds - 1e-6);
            lo = lo < i1 - D ? i1 - D : lo; lo = lo < 0 ? 0 : lo;
            hi = hi > i1 + D ? i1 + D : hi; hi = hi > imax ? imax : hi;
            double best = INFINITY; int arg = 255;
            for (int i0 = lo; i0 <= hi; ++i0) {
                const double vv = i0 * u - dp.v_des, aa = (i0 - i1) * u / dt, jj = (i0 - 2 * i1 + i2) * u / (dt * dt);
                const double tot = dp.v_w * vv * vv + dp.a_w * aa * aa + dp.j_w * jj * jj + Fp[(size_t)i0 * nd + (i0 - i1 + D)];
                if (tot < best) { best = tot; arg = i0; }
            }
            Fc[st] = best; pol[(size_t)r * nS + st] = (unsigned char)arg;
        }
        Fp.swap(Fc);
    }
    tab.assign((size_t)nS * R, 255);
    for (int i1 = 0; i1 <= imax; ++i1) for (int dd = -D; dd <= D; ++dd) {
        int c1 = i1, c2 = i1 - dd;
        if (c2 < 0 || c2 > imax) continue;
        unsigned char *row = &tab[(size_t)(i1 * nd + dd + D) * R];
        for (int t = 1; t <= R; ++t) {
            const int dcur = c1 - c2;
            if (dcur < -D || dcur > D) break;
            const int i0 = pol[(size_t)(R - t + 1) * nS + c1 * nd + dcur + D];
            if (i0 == 255) break;
            row[t - 1] = (unsigned char)i0;
            c2 = c1; c1 = i0;
        }
    }
    imax_out = imax; D_out = D;
    return true;
}

// Synchronous entries that reuse the counters: an error flag raised by an earlier asynchronous solve and not yet seen by
// stmpc_get_stats / stmpc_check_error is moved to the context's sticky word first (k_predict does the same on the device).
int latch_solver_error(stmpc_ctx *c) {
    if (!c->counters.p) return STMPC_OK;
    HIPCHK(hipDeviceSynchronize());
    unsigned cur = 0;
    HIPCHK(hipMemcpy(&cur, (const unsigned *)c->counters.p + STMPC_CNT_ERR, sizeof cur, hipMemcpyDeviceToHost));
    if (cur) { const unsigned one = 1u; HIPCHK(hipMemcpy(c->sticky.p, &one, sizeof one, hipMemcpyHostToDevice)); }
    return STMPC_OK;
}

template <int KMAX>
void launch_predict(const DevP &dp, int N, int Kmax, const double *ego, const int *k, const double *ox, const double *ov,
                    CarTab tab, unsigned *counters, u64 *ubound, int *queue1, unsigned *proxy0, int *resume_t, unsigned char *prio_key, hipStream_t st,
                    unsigned *sticky = nullptr, const unsigned char *guide_tab = nullptr, int guide_imax = 0, int guide_D = 0, u16 *guide = nullptr) {
    constexpr int E = PredShape<KMAX>::E;          // episodes per wavefront (k_predict)
    int blocks = (N + E - 1) / E;
    hipLaunchKernelGGL(k_predict<KMAX>, dim3(blocks), dim3(128), 0, st, dp, N, Kmax, ego, k, ox, ov, tab, counters, ubound, queue1, proxy0, resume_t, prio_key, sticky,
                       guide_tab, guide_imax, guide_D, guide, 0);
}

}  // namespace

extern "C" {

int stmpc_solve_batch_device(stmpc_ctx *c, const stmpc_params *p, int N, int Kmax, const double *d_ego,
                             const int32_t *d_k, const double *d_ox, const double *d_ov, int32_t *d_path,
                             int32_t *d_bt, double *d_cost, double *d_pd, int32_t *d_crash, void *stream) {
    return stmpc_solve_batch_device_ac(c, p, N, Kmax, d_ego, d_k, d_ox, d_ov, d_path, d_bt, d_cost, d_pd, d_crash, nullptr, stream);
}

int stmpc_solve_batch_device_ac(stmpc_ctx *c, const stmpc_params *p, int N, int Kmax, const double *d_ego,
                                const int32_t *d_k, const double *d_ox, const double *d_ov, int32_t *d_path,
                                int32_t *d_bt, double *d_cost, double *d_pd, int32_t *d_crash, double *d_action_cost, void *stream) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N < 0 || Kmax < 0 || Kmax > STMPC_KMAX_LIMIT) return fail(STMPC_EINVAL, "N or Kmax out of range");
    if (N == 0) return STMPC_OK;
    if (!d_ego || !d_k || !d_path || !d_bt || !d_cost) return fail(STMPC_EINVAL, "NULL device pointer");
    if (Kmax > 0 && (!d_ox || !d_ov)) return fail(STMPC_EINVAL, "NULL device pointer (other_x/other_v)");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    DevP dp;
    int rc = make_devp(p, &dp);
    if (rc) return rc;
    const int H = dp.H;
    const int S_nom = stmpc_num_s(p, 0.0);
    if (S_nom < 2 || S_nom + 2 > STMPC_S_LIMIT) return fail(STMPC_EINVAL, "number of position cells out of range");
    const int Kalloc = Kmax > 0 ? Kmax : 1;
    // FASTDIV kernels: Markstein's five-operation quotient for the lattice step (per-episode value), the two-operation one for
    // dt, dt^2, dt^3 once each of them has passed fastdiv2_ok (cached per context: the check costs a few hundred divisions)
    if (c->fd2_dt != dp.dt || c->fd2_dt2 != dp.dt2 || c->fd2_dt3 != dp.dt3) {
        c->fd2_dt = dp.dt; c->fd2_dt2 = dp.dt2; c->fd2_dt3 = dp.dt3;
        c->fd2_ok = fastdiv2_ok(dp.dt, &c->fd2_zl[0]) && fastdiv2_ok(dp.dt2, &c->fd2_zl[1]) && fastdiv2_ok(dp.dt3, &c->fd2_zl[2]);
    }
    const bool fastdiv = c->allow_fastdiv && fastdiv_ok(dp.dt) && fastdiv_ok(dp.dt2) && fastdiv_ok(dp.dt3) && c->fd2_ok;

    // scratch
    if ((rc = c->tab_edge.ensure((size_t)N * H * Kalloc * 2 * sizeof(double)))) return rc;
    if ((rc = c->tab_win.ensure((size_t)N * H * Kalloc * 2 * sizeof(int)))) return rc;
    if ((rc = c->tab_nact.ensure((size_t)N * H * sizeof(int)))) return rc;
    if ((rc = c->tab_nums.ensure((size_t)N * sizeof(int)))) return rc;
    if (!c->counters.p) { if ((rc = c->counters.ensure(64 * sizeof(unsigned)))) return rc; HIPCHK(hipMemsetAsync(c->counters.p, 0, 64 * sizeof(unsigned), st)); }
    if ((rc = c->lists.ensure((size_t)STMPC_MAX_TIERS * N * sizeof(int)))) return rc;
    if ((rc = c->ubound.ensure((size_t)N * sizeof(u64)))) return rc;
    if ((rc = c->proxy.ensure((size_t)N * sizeof(unsigned)))) return rc;
    if ((rc = c->order.ensure((size_t)N * sizeof(int)))) return rc;

    // widest fan-out the dynamics allow (st_cy.pyx:65-93): acceleration- or jerk-limited window, +2 for rounding
    const double fan_acc = (dp.a_max - dp.a_min) * dp.dt2 / dp.ds, fan_jerk = (dp.j_max - dp.j_min) * dp.dt3 / dp.ds;
    const double fan_bound = (fan_acc < fan_jerk ? fan_acc : fan_jerk) + 2.0;
    const bool small_fan = fan_bound <= 9.0;
    // the scalar-register vehicle table costs ~48 SGPRs/VGPRs: only with the small-fan kernel (the wide one would spill)
    const bool stage_tab = c->allow_stage_tab && small_fan && Kalloc <= 8 && stmpc_tab_bytes(H, 8) <= 4096;

    // tiers: LDS windows in increasing size, then one HBM-scratch tier whose window covers every cell
    const int Wg = next_pow2(S_nom + 2 + 128);   // covers every cell plus the 64-cell alignment slack
    int tierW[STMPC_MAX_TIERS]; int tierPW[STMPC_MAX_TIERS]; bool tierLds[STMPC_MAX_TIERS]; int tierGrid[STMPC_MAX_TIERS]; int tierNW[STMPC_MAX_TIERS];
    size_t tierLdsBytes[STMPC_MAX_TIERS];
    int nt = 0;
    int auto_W[2] = {2048, Wg < 8192 ? Wg : 8192};
    int n_auto = 2;
    if (auto_W[1] <= auto_W[0]) { auto_W[0] = auto_W[1]; n_auto = 1; }      // one window already covers the lattice
    const int n_lds = c->tiers_from_env ? c->n_lds_tiers : n_auto;
    for (int k = 0; k < n_lds && nt < STMPC_MAX_TIERS - 1; ++k) {
        int W = c->tiers_from_env ? c->lds_tier_W[k] : auto_W[k];
        if (W > Wg && nt > 0) break;
        const int nw = c->waves_tier[k] > 0 ? c->waves_tier[k] : (c->waves_override > 0 ? c->waves_override : (W <= 2048 ? 4 : 8));
        // penalty buffer: 1024 cells for the first (4-wave) tier -- with the 14 B/cell arrays that is 38 KB per
        // workgroup, i.e. 4 workgroups = 16 waves per CU -- and up to 4096 cells for the wider tiers
        int PW = c->pen_cells[k] > 0 ? c->pen_cells[k] : (k == 0 && W <= 2048 ? 1024 : 4096);
        if (PW > W) PW = W;
        const size_t lds = (size_t)W * STMPC_CELL_BYTES + STMPC_LIST_SLACK + (size_t)PW * 8 + ((stmpc_chunk_ints(W) * sizeof(int) + 15) & ~(size_t)15) +
                           stmpc_tab_bytes(H, stage_tab ? 8 : 0);
        if (lds + 2048 > (size_t)c->lds_per_block) break;
        tierW[nt] = W; tierPW[nt] = PW; tierLds[nt] = true; tierLdsBytes[nt] = lds;
        tierNW[nt] = nw;
        int per_cu = (int)((size_t)(c->lds_per_block) / (lds + c->lds_headroom));      // (+ the kernel's static LDS and allocation granularity)
        int by_waves = c->max_waves_per_cu / tierNW[nt];
        if (per_cu > by_waves) per_cu = by_waves;
        if (per_cu < 1) per_cu = 1;
        tierGrid[nt] = c->num_cu * per_cu;
        ++nt;
    }
    // an LDS tier whose window covers every cell cannot overflow: the HBM-scratch tier is only needed beyond that
    const bool need_hbm_tier = (nt == 0) || tierW[nt - 1] < Wg || tierPW[nt - 1] < tierW[nt - 1];
    if (need_hbm_tier) {
        // Clean-up launch: when the last LDS window already covers every cell, the only episodes that can reach this tier are those whose
        // lattice is not start + n*delta (the LDS kernels are compiled for that form) and rounds whose 64 sources' targets do not fit the
        // penalty buffer -- none in 4096 x 16 benchmark batches.  The launch then exists for correctness only and is sized accordingly: a
        // full persistent grid costs 13 us per step to start and leave on an empty queue (and its spill prologue writes 9 MB), 16 workgroups 3.
        const bool cleanup_only = nt > 0 && tierW[nt - 1] >= Wg && !c->tiers_from_env && !c->force_general;
        tierW[nt] = Wg; tierPW[nt] = Wg; tierLds[nt] = false; tierNW[nt] = c->waves_override > 0 ? c->waves_override : 8;
        tierLdsBytes[nt] = ((stmpc_chunk_ints(Wg) * sizeof(int) + 15) & ~(size_t)15) + 16;
        // (a batch that sent more than a handful of episodes there -- e.g. identical reset states whose second lattice point is not start + step --
        // gets the full grid from the next step on)
        // (the count comes from a mapped host word the batch's last launch stores itself, [1] of h_overflow: a caller that never reads statistics --
        // EpisodeRunner, decide_batch_device -- gets the full grid as well)
        const int64_t sent_last = c->h_overflow && c->last_has_hbm ? (int64_t)c->h_overflow[1] : 0;
        tierGrid[nt] = (cleanup_only && c->last_hbm_tier_count <= 16 && sent_last <= 16) ? 16 : c->num_cu * (c->max_waves_per_cu / tierNW[nt] > 0 ? c->max_waves_per_cu / tierNW[nt] : 1); ++nt;
    }
    const int prune_on = c->prune < 0 ? (small_fan ? 0 : 1) : c->prune;
    // checkpoint / resume across the first two LDS windows (SolveArgs::ckpt, ::pool_bp): a search that cannot build a layer in the first window
    // saves that layer and the back-pointer rows written so far in an entry of a pool and continues in the second window from there.
    // back-pointers: one byte (distance to the predecessor) when no step of the dynamics exceeds 255 cells, else two (its cell)
    const bool bp_rel8 = ceil(dp.v_max * dp.dt / dp.ds) + 4.0 <= 255.0 && !c->bp16;
    const size_t bp_elem = bp_rel8 ? 1 : sizeof(u16);
    const size_t ckpt_stride = 16 + (size_t)tierW[0] * 12;
    // Pool: an eighth of the batch (5 % of the benchmark's searches overflow), at least 256 entries, of H x W0 back-pointers + one saved layer
    // (104 KB at H = 40): 53 MB for 4096 episodes, 0.85 GB for 65536 -- round 4 kept both for EVERY episode (0.43 GB / 6.9 GB).  A search that
    // finds the pool exhausted starts over in the wider window (stmpc_stats::pool_exhausted counts them).
    int pool_cap = c->pool_cap_override > 0 ? c->pool_cap_override : (N / 8 > 256 ? N / 8 : 256);
    if (pool_cap > N) pool_cap = N;
    if (pool_cap > (1 << 22)) pool_cap = 1 << 22;            // (the entry number shares a word with the layer)
    const size_t pool_bytes = (size_t)pool_cap * ((size_t)H * tierW[0] * bp_elem + ckpt_stride);
    bool resume = c->resume && prune_on && !small_fan && !stage_tab && nt >= 2 && tierLds[0] && tierLds[1];      // (compiled for the wide-fan kernels only)
    if (resume && c->pool_bp.cap + c->ckpt.cap < pool_bytes) {
        // a growing request: only while it is at most a quarter of what the device has free right now (a process shared with torch / RCCL).
        // A request that was turned down is priced again every 64th call: memory another tenant held at that moment may be free by now.
        if (c->resume_refused_for == pool_bytes && (++c->resume_refused_calls & 63) != 0) resume = false;
        else {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            if (pool_bytes > (free_b + c->pool_bp.cap + c->ckpt.cap) / 4) { resume = false; c->resume_refused_for = pool_bytes; }
            else c->resume_refused_for = 0;
        }
    }
    const bool resume_wanted = c->resume && prune_on && !small_fan && !stage_tab && nt >= 2 && tierLds[0] && tierLds[1];
    // reserved compute units (experiment, STMPC_CU_RESERVE): the first window's persistent grid covers the remaining units only
    const bool reserve_cfg = c->cu_reserve > 0 && prune_on && nt >= 2 && tierLds[0] && tierLds[1] && !c->two_phase;
    if (reserve_cfg) tierGrid[0] = tierGrid[0] / c->num_cu * (c->num_cu - c->cu_reserve);
    for (int k = 0; k < nt; ++k) if (tierGrid[k] > N) tierGrid[k] = N;
    if (resume) {
        // the pool: if the device cannot spare it, overflowing searches restart in the wider window instead of continuing
        if (c->pool_bp.ensure((size_t)pool_cap * H * tierW[0] * bp_elem) || c->ckpt.ensure((size_t)pool_cap * ckpt_stride) || c->resume_t.ensure((size_t)N * sizeof(int))) {
            (void)hipGetLastError();
            c->pool_bp.release(); c->ckpt.release();
            resume = false;
        }
    }
    for (int k = 0; k < nt; ++k)                 // back-pointers of a tier: per resident workgroup
        if ((rc = c->bp_tier[k].ensure((size_t)tierGrid[k] * H * tierW[k] * bp_elem))) return rc;
    c->last_resume_refused = resume_wanted && !resume;
    int *resume_t = resume ? c->resume_t.as<int>() : nullptr;
    if (need_hbm_tier && (rc = c->gscratch.ensure((size_t)tierGrid[nt - 1] * ((size_t)Wg * STMPC_CELL_BYTES + STMPC_LIST_SLACK + (size_t)Wg * 8)))) return rc;

    CarTab tab{c->tab_edge.as<double>(), c->tab_win.as<int>(), c->tab_nact.as<int>(), c->tab_nums.as<int>()};
    unsigned *counters = c->counters.as<unsigned>();

    hipEvent_t e0 = c->ev0, e1 = c->ev1, e2 = c->ev2, e3 = c->ev3;
    if (c->profiling) {
        if (c->pool_used + 4 > c->pool.size()) {
            for (int i = 0; i < 4; ++i) { hipEvent_t ev; HIPCHK(hipEventCreate(&ev)); c->pool.push_back(ev); }
        }
        e0 = c->pool[c->pool_used]; e1 = c->pool[c->pool_used + 1]; e2 = c->pool[c->pool_used + 2]; e3 = c->pool[c->pool_used + 3];
        c->pool_used += 4;
    }
    // second LDS tier started alongside the first (see k_solve): only where overflow is common enough to pay for the
    // extra launch, and not with the two-phase schedule (its first launch of tier 0 only bounds)
    // (narrow lattice, round 5: three of 4096 benchmark states overflow the first window; run after the first launch they cost one search's latency,
    // 0.14 of a 1.0 ms step; alongside it, on a small grid, they are done when it ends -- but a side launch that finds nothing to do costs 40 us of
    // stream hand-overs, so it is started only when the previous batch on this context overflowed)
    if (!c->h_overflow) {
        HIPCHK(hipHostMalloc((void **)&c->h_overflow, 2 * sizeof(int), hipHostMallocMapped)); c->h_overflow[0] = 0; c->h_overflow[1] = 0;
        HIPCHK(hipHostGetDevicePointer((void **)&c->d_overflow, c->h_overflow, 0));
    }
    const bool overlap_auto = prune_on != 0 || (small_fan && *c->h_overflow > 0);
    const bool overlap = nt >= 2 && tierLds[1] && !(prune_on && c->two_phase) && N > tierGrid[0] &&
                         (c->overlap < 0 ? overlap_auto : c->overlap != 0);
    const int side_grid_auto = c->side_grid > 0 ? c->side_grid : (small_fan ? 32 : 0);
    const bool reserve = reserve_cfg && overlap;
    int *queue1 = overlap ? c->lists.as<int>() + (size_t)N : nullptr;
    const bool split = prune_on && !c->two_phase && c->split && N >= 2 * tierGrid[0];
    unsigned *proxy0 = split ? c->proxy.as<unsigned>() : nullptr;
    const bool heavy_first = split && c->heavy_first;
    if (heavy_first && (rc = c->prio_key.ensure((size_t)N))) return rc;
    unsigned char *prio_key = heavy_first ? c->prio_key.as<unsigned char>() : nullptr;
    const unsigned char *g_tab = nullptr; u16 *g_cells = nullptr; int g_imax = 0, g_D = 0;
    if (prune_on && c->tube_w > 0) {
        // guided bounding attempt: the table depends on the dynamics and the cost weights only; rebuilt when they change (a few ms on the host)
        const double key[12] = {dp.ds, dp.dt, dp.v_w, dp.a_w, dp.j_w, dp.v_des, dp.v_max, dp.a_min, dp.a_max, dp.j_min, dp.j_max, (double)H};
        stmpc_ctx::GuideSlot *slot = nullptr;
        for (auto &g : c->guides) if (g.valid && memcmp(key, g.key, sizeof key) == 0) { slot = &g; break; }
        if (!slot) {
            // A parameter set not seen before (or replaced since): build its table on the host (a few ms) and queue the upload on THIS call's
            // stream, ahead of the kernels that read it.  Nothing waits for the device unless a table has to be replaced (a fifth parameter set):
            // kernels of earlier calls may still read the one that goes.
            for (auto &g : c->guides) if (!g.valid) { slot = &g; break; }
            if (!slot) {
                slot = &c->guides[0];
                for (auto &g : c->guides) if (g.last_use < slot->last_use) slot = &g;
                slot->valid = false;
                HIPCHK(hipDeviceSynchronize());
            }
            slot->ok = build_guide_table(dp, slot->host, slot->imax, slot->D);
            if (slot->ok) {
                if ((rc = slot->dev.ensure(slot->host.size()))) return rc;
                HIPCHK(hipMemcpyAsync(slot->dev.p, slot->host.data(), slot->host.size(), hipMemcpyHostToDevice, st));
            }
            memcpy(slot->key, key, sizeof key);
            slot->valid = true;                 // (only after the upload has been queued successfully)
        }
        slot->last_use = ++c->guide_clock;
        if (slot->ok) { if ((rc = c->guide_cells.ensure((size_t)N * H * sizeof(u16)))) return rc; g_tab = slot->dev.as<unsigned char>(); g_cells = c->guide_cells.as<u16>(); g_imax = slot->imax; g_D = slot->D; }
    }
    HIPCHK(hipEventRecord(e0, st));
    if (Kalloc <= 8) launch_predict<8>(dp, N, Kalloc, d_ego, d_k, d_ox, d_ov, tab, counters, c->ubound.as<u64>(), queue1, proxy0, resume_t, prio_key, st, c->sticky.as<unsigned>(), g_tab, g_imax, g_D, g_cells);
    else if (Kalloc <= 16) launch_predict<16>(dp, N, Kalloc, d_ego, d_k, d_ox, d_ov, tab, counters, c->ubound.as<u64>(), queue1, proxy0, resume_t, prio_key, st, c->sticky.as<unsigned>(), g_tab, g_imax, g_D, g_cells);
    else launch_predict<32>(dp, N, Kalloc, d_ego, d_k, d_ox, d_ov, tab, counters, c->ubound.as<u64>(), queue1, proxy0, resume_t, prio_key, st, c->sticky.as<unsigned>(), g_tab, g_imax, g_D, g_cells);

    SolveArgs a;
    memset(&a, 0, sizeof a);
    a.p = dp; a.N = N; a.Kmax = Kalloc;
    a.ego = d_ego; a.tab = tab;
    a.counters = counters; a.lists = c->lists.as<int>(); a.ubound = c->ubound.as<u64>();
    a.prune = prune_on;
    // band of the bounding pre-pass.  Nominal: half the per-step cost of standing still (225 with the reference's
    // weights).  With the node cap (default) the pass starts from 8x that and narrows the band whenever a layer expands
    // more than band_cap nodes (dp_pass): wide where few alternatives exist, beam-like where many do -- 15 % fewer
    // expanded nodes in total than the best fixed band (sweeps on the H=40 workload: fixed 60..1200, capped 225..8000 x
    // 150..550).  Any value is safe (the exact pass re-checks); it only trades pre-pass work for tightness of the bound.
    const double band_nominal = fmax(1.0, 0.5 * dp.v_w * dp.v_des * dp.v_des);
    a.band = c->band_override > 0 ? c->band_override : (c->band_cap > 0 ? 8.0 * band_nominal : band_nominal);
    // second attempt (penalty zone allowed): a wider band, but kept well below the cost of one worst-case step
    // (j_w * j_max^2 ~ 12 k on the benchmark lattice): a band that admits those steps keeps everything, and single
    // episodes then take several times longer (measured cliff at 44x the nominal band; 32x is used)
    a.band2_mult = c->band2_mult > 0 ? c->band2_mult : (c->band_cap > 0 && c->band_override <= 0 ? 4.0 : 5.0);
    if (c->band2_mult <= 0 && c->band_cap > 0) {
        const double dv = fmax(dp.v_des, dp.v_max - dp.v_des), da = fmax(fabs(dp.a_min), fabs(dp.a_max)), dj = fmax(fabs(dp.j_min), fabs(dp.j_max));
        const double step_max = dp.v_w * dv * dv + dp.a_w * da * da + dp.j_w * dj * dj;      // dearest single step, penalties aside
        if (step_max > 0 && a.band * a.band2_mult > 0.7 * step_max) a.band2_mult = fmax(1.0, 0.7 * step_max / a.band);
    }
    a.band_cap = c->band_cap;
    for (int i = 0; i < 3; ++i) a.retry_mult[i] = c->retry_mult[i];
    a.bound_infl = c->bound_infl; a.last_infl = c->last_infl;
    a.guide = g_cells; a.tube_w = c->tube_w; a.tube_dense = c->tube_dense; a.band_dense = c->band_dense;
    a.retry_move = resume ? c->retry_move : 0;
    a.prio_thr = c->prio_thr; a.prio_mode = c->prio_mode;
    a.bp_rel8 = bp_rel8 ? 1 : 0;
    a.force_general = c->force_general ? 1 : 0;
    a.gsh_max = c->gsh_max;
    a.zl_dt = c->fd2_zl[0]; a.zl_dt2 = c->fd2_zl[1]; a.zl_dt3 = c->fd2_zl[2];
#ifdef STMPC_PHASE_PROF
    if ((rc = c->phase_prof.ensure(4 * STMPC_NPH * sizeof(unsigned long long)))) return rc;
    HIPCHK(hipMemsetAsync(c->phase_prof.p, 0, 4 * STMPC_NPH * sizeof(unsigned long long), st));
    a.phase_prof = c->phase_prof.as<unsigned long long>();
#endif
    a.ckpt = resume ? c->ckpt.as<unsigned char>() : nullptr; a.ckpt_stride = ckpt_stride; a.resume_t = resume_t;
    a.pool_bp = resume ? c->pool_bp.as<unsigned char>() : nullptr; a.pool_cap = resume ? pool_cap : 0;
    a.W0 = tierW[0];
    a.maxshift = (int)ceil(dp.v_max * dp.dt / dp.ds) + 2 + 66;     // st_cy.pyx:65-93: v <= v_max; + interval rounding to 64-cell blocks
    a.proxy = c->proxy.as<unsigned>();
    const bool two_phase = a.prune && c->two_phase;      // bound all episodes first, then solve them heaviest-first
    a.path_idx = d_path; a.best_t = d_bt; a.cost = d_cost; a.path_dist = d_pd; a.crash = d_crash; a.action_cost = d_action_cost;
    a.host_overflow = c->d_overflow;

    if (overlap && split && c->retire_cus > 0 && c->retire_cus < c->num_cu) {
        if ((rc = c->cu_tab.ensure(1025 * sizeof(unsigned)))) return rc;
        HIPCHK(hipMemsetAsync(c->cu_tab.p, 0, 1025 * sizeof(unsigned), st));
        a.cu_tab = c->cu_tab.as<unsigned>(); a.retire_from = c->num_cu - c->retire_cus; a.retire_left = (long long)N * c->retire_at / 100;
    }
    HIPCHK(hipEventRecord(e1, st));
    if (overlap) HIPCHK(hipEventRecord(c->ev_fork, st));      // the vehicle table and the preset queue are ready

    // one launch of tier k: phase 0 = bound + exact, 1 = bounding pre-passes only, 2 = exact with the stored bounds;
    // side = on the side stream, consuming tier 0's overflow queue while tier 0 is still running
    auto launch_tier = [&](int k, int phase, bool side, bool on_reserved = false) -> int {
        hipStream_t lst = side ? (on_reserved ? c->aux_reserved : c->aux_stream) : ((reserve && k == 0) ? c->main_masked : st);
        a.concurrent = side ? 1 : 0;
        a.always_wait = on_reserved ? 1 : 0;
        a.split = (split && k == 0) ? 1 : 0;
        a.feeds_concurrent = (overlap && k == 0) ? 1 : 0;
        a.prev_grid = side ? tierGrid[0] : 0;
        a.wait_ticks = side ? 20000000ull : 0ull;             // 0.2 s of the 100 MHz clock
        a.phase = phase;
        a.order = ((phase == 2 || heavy_first) && k == 0) ? c->order.as<int>() : nullptr;
        a.W = tierW[k]; a.PW = tierPW[k]; a.tier = k; a.last_tier = (k == nt - 1);
        a.bp = c->bp_tier[k].as<u16>();
        a.gscratch = tierLds[k] ? nullptr : c->gscratch.as<unsigned char>();
        const size_t lds = tierLdsBytes[k];
        const bool std_shape = tierNW[k] == 4 && tierW[k] == 2048 && tierPW[k] == 1024;      // the kernels compiled with these as constants
        const bool std_shape2 = tierNW[k] == 8 && tierW[k] == 8192 && tierPW[k] == 4096;
        const int side_g = (side && !on_reserved && side_grid_auto > 0 && side_grid_auto < tierGrid[k]) ? side_grid_auto : tierGrid[k];
        const dim3 grid(on_reserved ? (tierGrid[k] / c->num_cu > 0 ? tierGrid[k] / c->num_cu : 1) * c->cu_reserve : side_g), block(64 * tierNW[k]);
#define STMPC_LAUNCH_R(L, FD, KT_, FM, SG, RS)                                                                \
        do {                                                                                                  \
            if (lds > 48 * 1024)                                                                              \
                HIPCHK(hipFuncSetAttribute((const void *)k_solve<L, false, FD, KT_, FM, SG, RS>,              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
            hipLaunchKernelGGL((k_solve<L, false, FD, KT_, FM, SG, RS>), grid, block, lds, lst, a);           \
        } while (0)
        // (the first window's four-wave workgroups: list segments searched with 3 compares instead of 7)
#define STMPC_LAUNCH_R4(L, FD, KT_, FM, SG, RS)                                                               \
        do {                                                                                                  \
            if (lds > 48 * 1024)                                                                              \
                HIPCHK(hipFuncSetAttribute((const void *)k_solve<L, false, FD, KT_, FM, SG, RS, 4>,           \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
            hipLaunchKernelGGL((k_solve<L, false, FD, KT_, FM, SG, RS, 4>), grid, block, lds, lst, a);        \
        } while (0)
#define STMPC_LAUNCH_R88(L, FD, KT_, FM, SG, RS)                                                              \
        do {                                                                                                  \
            if (lds > 48 * 1024)                                                                              \
                HIPCHK(hipFuncSetAttribute((const void *)k_solve<L, false, FD, KT_, FM, SG, RS, 88>,          \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
            hipLaunchKernelGGL((k_solve<L, false, FD, KT_, FM, SG, RS, 88>), grid, block, lds, lst, a);       \
        } while (0)
#ifndef STMPC_FAN1
#define STMPC_FAN1 8        /* candidate slots per barrier pair of the wide-lattice kernels outside the standard second window (128 VGPRs); 7 / 11 / 12 measured in round 5, 12 again in round 6 */
#endif
#ifndef STMPC_FAN88
#define STMPC_FAN88 24      /* candidate slots per barrier pair in the standard second window (it has the registers: 256 VGPRs); 12 / 16 / 21 / 24 measured, EXPERIMENTS.md */
#endif
#define STMPC_LAUNCH_R0(L, FD, KT_, FM, SG)                                                                   \
        do {                                                                                                  \
            if constexpr (L) { if (std_shape) STMPC_LAUNCH_R4(L, FD, KT_, FM, SG, 0); else STMPC_LAUNCH_R(L, FD, KT_, FM, SG, 0); } \
            else STMPC_LAUNCH_R(L, FD, KT_, FM, SG, 0);                                                       \
        } while (0)
        // checkpointing variants only where they are used: the first window saves, the second continues
#define STMPC_LAUNCH_S(L, FD, KT_, FM, SG)                                                                    \
        do {                                                                                                  \
            if constexpr (L && FM == STMPC_FAN1 && KT_ == 0) {                                                        \
                if (resume && k == 0 && std_shape) STMPC_LAUNCH_R4(L, FD, KT_, FM, SG, 1);               \
                else if (resume && k == 0) STMPC_LAUNCH_R(L, FD, KT_, FM, SG, 1);                             \
                else if (resume && k == 1 && std_shape2) STMPC_LAUNCH_R88(L, FD, KT_, STMPC_FAN88, SG, 2);    \
                else if (resume && k == 1) STMPC_LAUNCH_R(L, FD, KT_, FM, SG, 2);                             \
                else STMPC_LAUNCH_R0(L, FD, KT_, FM, SG);                                                     \
            } else STMPC_LAUNCH_R0(L, FD, KT_, FM, SG);                                                       \
        } while (0)
        // only the last tier carries the general lattice-coordinate form (see solve_episode)
#define STMPC_LAUNCH(L, FD, KT_, FM) do { if (a.last_tier) STMPC_LAUNCH_S(L, FD, KT_, FM, true); else STMPC_LAUNCH_S(L, FD, KT_, FM, false); } while (0)
#define STMPC_LAUNCH_FM(L, FD, KT_) do { if (small_fan) STMPC_LAUNCH(L, FD, KT_, 9); else STMPC_LAUNCH(L, FD, KT_, STMPC_FAN1); } while (0)
        if (tierLds[k]) {
            if (stage_tab) { if (fastdiv) STMPC_LAUNCH_FM(true, true, 8); else STMPC_LAUNCH_FM(true, false, 8); }
            else { if (fastdiv) STMPC_LAUNCH_FM(true, true, 0); else STMPC_LAUNCH_FM(true, false, 0); }
        } else {
            if (fastdiv) STMPC_LAUNCH_FM(false, true, 0); else STMPC_LAUNCH_FM(false, false, 0);
        }
#undef STMPC_LAUNCH_FM
#undef STMPC_LAUNCH
#undef STMPC_LAUNCH_S
#undef STMPC_LAUNCH_R
#undef STMPC_LAUNCH_R4
#undef STMPC_LAUNCH_R88
#undef STMPC_LAUNCH_R0
        return STMPC_OK;
    };

    if (heavy_first) hipLaunchKernelGGL(k_order8, dim3(1), dim3(1024), 0, st, N, (const unsigned char *)prio_key, c->order.as<int>());
    if (two_phase) {                                           // bound every episode, order them heaviest-first
        if ((rc = launch_tier(0, 1, false))) return rc;
        hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, st, N, (const unsigned *)c->proxy.as<unsigned>(), c->order.as<int>());
    }
    for (int k = 0; k < nt; ++k) {
        if (reserve && k == 0) HIPCHK(hipStreamWaitEvent(c->main_masked, c->ev_fork, 0));
        if ((rc = launch_tier(k, two_phase ? 2 : 0, false))) return rc;
        if (reserve && k == 0) {
            // the reserved units host second-window workgroups from the start of the step; the masked streams partition the device, so
            // these consumers may always wait for the queue (they cannot be holding a unit a producer needs)
            HIPCHK(hipEventRecord(c->ev_join0, c->main_masked));
            HIPCHK(hipStreamWaitEvent(c->aux_reserved, c->ev_fork, 0));
            if ((rc = launch_tier(1, 0, true, true))) return rc;
            HIPCHK(hipEventRecord(c->ev_join_r, c->aux_reserved));
            HIPCHK(hipStreamWaitEvent(st, c->ev_join0, 0));
            HIPCHK(hipStreamWaitEvent(st, c->ev_join_r, 0));
        }
        if (overlap && k == 0) {
            // tier 1 alongside tier 0: queued on the side stream behind the predictor only; its workgroups start when
            // tier 0's persistent workgroups begin to leave CUs.  The main stream then waits for it, and the ordinary
            // launch of tier 1 that follows picks up whatever it left (normally nothing).
            HIPCHK(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
            if ((rc = launch_tier(1, 0, true))) return rc;
            HIPCHK(hipEventRecord(c->ev_join, c->aux_stream));
            HIPCHK(hipStreamWaitEvent(st, c->ev_join, 0));
        }
        if ((need_hbm_tier && k == nt - 2) || (!need_hbm_tier && k == nt - 1) || nt == 1) HIPCHK(hipEventRecord(e2, st));   // after the last LDS tier
    }
    HIPCHK(hipEventRecord(e3, st));
    HIPCHK(hipGetLastError());
    c->stats.episodes = N;
    c->last_nt = nt; c->last_has_hbm = need_hbm_tier;
    c->stats_pending = !c->profiling;
    if (c->profiling) { c->acc_launches += 1; c->acc_episodes += N; }
    return STMPC_OK;
}

int stmpc_get_stats(stmpc_ctx *c, stmpc_stats *out) {
    if (!c || !out) return fail(STMPC_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    if (c->stats_pending) {
        HIPCHK(hipEventSynchronize(c->ev3));
        unsigned cnt[64];
        HIPCHK(hipMemcpy(cnt, c->counters.p, sizeof cnt, hipMemcpyDeviceToHost));
        float ms_all = 0.f, ms_dp = 0.f;
        HIPCHK(hipEventElapsedTime(&ms_all, c->ev0, c->ev3));
        HIPCHK(hipEventElapsedTime(&ms_dp, c->ev1, c->ev2));
#ifdef STMPC_PHASE_PROF
        if (const char *f = getenv("STMPC_PHASE_DUMP")) {
            unsigned long long pp[4 * STMPC_NPH];
            HIPCHK(hipMemcpy(pp, c->phase_prof.p, sizeof pp, hipMemcpyDeviceToHost));
            FILE *fp = fopen(f, "w");
            if (fp) { for (int m = 0; m < 4; ++m) { for (int k = 0; k < STMPC_NPH; ++k) fprintf(fp, "%llu ", pp[m * STMPC_NPH + k]); fprintf(fp, "\n"); } fclose(fp); }
        }
#endif
        c->stats.fallback = cnt[4];                       // episodes that overflowed the first LDS window
        c->stats.hbm_tier = (c->last_has_hbm && c->last_nt >= 2) ? cnt[4 * (c->last_nt - 1)] : 0;
        c->stats.fast_path = c->stats.episodes - cnt[4];
        c->last_hbm_tier_count = c->stats.hbm_tier;
        c->stats.resume_refused = c->last_resume_refused ? 1 : 0;
        c->stats.pool_exhausted = cnt[STMPC_CNT_POOL_FULL];
        c->stats.retries = cnt[STMPC_CNT_RETRY];
        c->stats.guided = cnt[STMPC_CNT_GUIDED];
        c->stats.nodes_exact = cnt[STMPC_CNT_NODES_EXACT];
        c->stats.nodes_bound = cnt[STMPC_CNT_NODES_BOUND];
        c->stats.solve_ms = ms_all;
        c->stats.dp_kernel_ms = ms_dp;
        c->stats_pending = false;
        if (cnt[STMPC_CNT_ERR]) {
            // reported here, once: cleared so that the next k_predict does not latch it again and blame a later batch
            HIPCHK(hipMemset((unsigned *)c->counters.p + STMPC_CNT_ERR, 0, sizeof(unsigned)));
            *out = c->stats;
            return fail(STMPC_EINTERNAL, "solver error flag set on device");
        }
    }
    *out = c->stats;
    return STMPC_OK;
}

int stmpc_profile(stmpc_ctx *c, int enable, stmpc_profile_totals *out) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    if (enable) {
        c->profiling = true; c->pool_used = 0;
        c->acc_solve_ms = c->acc_dp_ms = 0; c->acc_launches = c->acc_fallback = c->acc_episodes = 0;
        return STMPC_OK;
    }
    // disable: drain the pool
    for (size_t i = 0; i + 3 < c->pool_used; i += 4) {
        HIPCHK(hipEventSynchronize(c->pool[i + 3]));
        float a = 0.f, b = 0.f;
        HIPCHK(hipEventElapsedTime(&a, c->pool[i], c->pool[i + 3]));
        HIPCHK(hipEventElapsedTime(&b, c->pool[i + 1], c->pool[i + 2]));
        c->acc_solve_ms += a; c->acc_dp_ms += b;
    }
    c->pool_used = 0;
    c->profiling = false;
    if (out) {
        out->launches = c->acc_launches; out->episodes = c->acc_episodes;
        out->solve_ms = c->acc_solve_ms; out->dp_kernel_ms = c->acc_dp_ms;
    }
    return STMPC_OK;
}

int stmpc_solve_batch(stmpc_ctx *c, const stmpc_params *p, int N, int Kmax, const double *ego, const int32_t *k,
                      const double *ox, const double *ov, int32_t *path, int32_t *bt, double *cost, double *pd,
                      int32_t *crash) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N < 0 || Kmax < 0 || Kmax > STMPC_KMAX_LIMIT) return fail(STMPC_EINVAL, "N or Kmax out of range");
    if (N == 0) return STMPC_OK;
    if (!ego || !k || !path || !bt || !cost) return fail(STMPC_EINVAL, "NULL host pointer");
    if (Kmax > 0 && (!ox || !ov)) return fail(STMPC_EINVAL, "NULL host pointer (other_x/other_v)");
    for (int i = 0; i < N; ++i) if (k[i] < 0 || k[i] > Kmax) return fail(STMPC_EINVAL, "k_count[i] outside [0, Kmax]");
    HIPCHK(hipSetDevice(c->device));
    int H = stmpc_num_t(p);
    if (H < 2 || H > STMPC_H_LIMIT) return fail(STMPC_EINVAL, "number of time layers must be in [2, 64]");
    int rc;
    const int Kalloc = Kmax > 0 ? Kmax : 1;
    if ((rc = c->s_ego.ensure((size_t)N * 5 * 8))) return rc;
    if ((rc = c->s_k.ensure((size_t)N * 4))) return rc;
    if ((rc = c->s_ox.ensure((size_t)N * Kalloc * 8))) return rc;
    if ((rc = c->s_ov.ensure((size_t)N * Kalloc * 8))) return rc;
    if ((rc = c->s_path.ensure((size_t)N * H * 4))) return rc;
    if ((rc = c->s_bt.ensure((size_t)N * 4))) return rc;
    if ((rc = c->s_cost.ensure((size_t)N * 8))) return rc;
    if ((rc = c->s_pd.ensure((size_t)N * H * 8))) return rc;
    if ((rc = c->s_crash.ensure((size_t)N * 4))) return rc;
    HIPCHK(hipMemcpy(c->s_ego.p, ego, (size_t)N * 5 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->s_k.p, k, (size_t)N * 4, hipMemcpyHostToDevice));
    if (Kmax > 0) {
        HIPCHK(hipMemcpy(c->s_ox.p, ox, (size_t)N * Kmax * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->s_ov.p, ov, (size_t)N * Kmax * 8, hipMemcpyHostToDevice));
    }
    rc = stmpc_solve_batch_device(c, p, N, Kmax, c->s_ego.as<double>(), c->s_k.as<int32_t>(), c->s_ox.as<double>(),
                                  c->s_ov.as<double>(), c->s_path.as<int32_t>(), c->s_bt.as<int32_t>(),
                                  c->s_cost.as<double>(), c->s_pd.as<double>(), c->s_crash.as<int32_t>(), nullptr);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(path, c->s_path.p, (size_t)N * H * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(bt, c->s_bt.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cost, c->s_cost.p, (size_t)N * 8, hipMemcpyDeviceToHost));
    if (pd) HIPCHK(hipMemcpy(pd, c->s_pd.p, (size_t)N * H * 8, hipMemcpyDeviceToHost));
    if (crash) HIPCHK(hipMemcpy(crash, c->s_crash.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    stmpc_stats s;
    return stmpc_get_stats(c, &s);
}

int stmpc_solve_grid(stmpc_ctx *c, const uint8_t *obstacles, const double *s_values, int S, const double *t_values,
                     int H, double v0, double a0, const double *distances, double d_w, double v_w, double a_w,
                     double j_w, double v_des, double v_max, double a_min, double a_max, double j_min, double j_max,
                     double min_allowed, double *s_sequence_out) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (!obstacles || !s_values || !t_values || !distances || !s_sequence_out) return fail(STMPC_EINVAL, "NULL host pointer");
    if (H < 2 || H > STMPC_H_LIMIT) return fail(STMPC_EINVAL, "num_t must be in [2, 64]");
    if (S < 2 || S > STMPC_S_LIMIT) return fail(STMPC_EINVAL, "num_s must be in [2, 65000]");
    HIPCHK(hipSetDevice(c->device));
    DevP dp;
    memset(&dp, 0, sizeof dp);
    dp.dt = t_values[1] - t_values[0];           // st_cy.pyx:319
    if (dp.dt == 0.0) return fail(STMPC_EINVAL, "float division by zero (delta_t == 0)");   // ZeroDivisionError in the reference
    if (s_values[1] - s_values[0] == 0.0) return fail(STMPC_EINVAL, "float division by zero (delta_s == 0)");
    dp.dt2 = dp.dt * dp.dt; dp.dt3 = host_pow(dp.dt, 3.0);
    dp.d_w = d_w; dp.v_w = v_w; dp.a_w = a_w; dp.j_w = j_w; dp.v_des = v_des; dp.v_max = v_max; dp.a_min = a_min;
    dp.a_max = a_max; dp.j_min = j_min; dp.j_max = j_max; dp.min_allowed = min_allowed;
    dp.crash_dist_thr = -1.0; dp.H = H;
    int rc;
    const size_t cells = (size_t)H * S;
    if ((rc = c->s_misc0.ensure(cells))) return rc;
    if ((rc = c->s_misc1.ensure(cells * 8))) return rc;
    if ((rc = c->s_misc2.ensure((size_t)S * 8))) return rc;
    if ((rc = c->s_misc3.ensure((size_t)H * 8))) return rc;
    if ((rc = c->counters.ensure(64 * sizeof(unsigned)))) return rc;
    const int Wg = next_pow2(S + 2 + 128);
    if ((rc = c->gscratch.ensure((size_t)Wg * STMPC_CELL_BYTES + STMPC_LIST_SLACK + (size_t)Wg * 8))) return rc;
    if ((rc = c->bp_tier[STMPC_MAX_TIERS - 1].ensure((size_t)H * Wg * sizeof(u16)))) return rc;
    HIPCHK(hipMemcpy(c->s_misc0.p, obstacles, cells, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->s_misc1.p, distances, cells * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->s_misc2.p, s_values, (size_t)S * 8, hipMemcpyHostToDevice));
    if ((rc = latch_solver_error(c))) return rc;          // an earlier asynchronous call's flag survives the reset below
    HIPCHK(hipMemset(c->counters.p, 0, 64 * sizeof(unsigned)));
    SolveArgs a;
    memset(&a, 0, sizeof a);
    a.p = dp; a.N = 1; a.Kmax = 1; a.W = Wg; a.PW = Wg; a.last_tier = 1;
    for (int i = 0; i < 3; ++i) a.retry_mult[i] = c->retry_mult[i];
    a.bound_infl = c->bound_infl; a.last_infl = c->last_infl;
    a.obstacles = c->s_misc0.as<uint8_t>(); a.distances = c->s_misc1.as<double>(); a.s_values = c->s_misc2.as<double>();
    a.S_grid = S; a.v0_grid = v0; a.a0_grid = a0; a.gsh_max = c->gsh_max;
    a.bp = c->bp_tier[STMPC_MAX_TIERS - 1].as<u16>(); a.gscratch = c->gscratch.as<unsigned char>(); a.counters = c->counters.as<unsigned>();
    a.s_sequence = c->s_misc3.as<double>();
    hipLaunchKernelGGL((k_solve<false, true, false, 0, 16, true>), dim3(1), dim3(256), ((stmpc_chunk_ints(Wg) * sizeof(int) + 15) & ~(size_t)15) + 16, nullptr, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(s_sequence_out, c->s_misc3.p, (size_t)H * 8, hipMemcpyDeviceToHost));
    unsigned cnt[64];
    HIPCHK(hipMemcpy(cnt, c->counters.p, sizeof cnt, hipMemcpyDeviceToHost));
    if (cnt[STMPC_CNT_ERR]) {
        HIPCHK(hipMemset((unsigned *)c->counters.p + STMPC_CNT_ERR, 0, sizeof(unsigned)));
        return fail(STMPC_EINTERNAL, "grid solver reported a window overflow");
    }
    return STMPC_OK;
}

int stmpc_build_grid(stmpc_ctx *c, const stmpc_params *p, const double *state5, int k, const double *ox,
                     const double *ov, uint8_t *obstacles, double *distances, double *s_values, double *t_values) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (!state5 || !obstacles || !distances || !s_values || !t_values) return fail(STMPC_EINVAL, "NULL host pointer");
    if (k < 0 || k > STMPC_KMAX_LIMIT || (k > 0 && (!ox || !ov))) return fail(STMPC_EINVAL, "bad vehicle count / arrays");
    HIPCHK(hipSetDevice(c->device));
    DevP dp;
    int rc = make_devp(p, &dp);
    if (rc) return rc;
    const int H = dp.H;
    const double start_s = state5[4];
    const int S = stmpc_num_s(p, start_s);
    if (S < 2 || S > STMPC_S_LIMIT) return fail(STMPC_EINVAL, "number of position cells out of range");
    const int Kalloc = k > 0 ? k : 1;
    if ((rc = c->tab_edge.ensure((size_t)H * Kalloc * 2 * 8))) return rc;
    if ((rc = c->tab_win.ensure((size_t)H * Kalloc * 2 * 4))) return rc;
    if ((rc = c->tab_nact.ensure((size_t)H * 4))) return rc;
    if ((rc = c->tab_nums.ensure(4))) return rc;
    if ((rc = c->counters.ensure(64 * sizeof(unsigned)))) return rc;
    if ((rc = c->s_ego.ensure(5 * 8))) return rc;
    if ((rc = c->s_k.ensure(4))) return rc;
    if ((rc = c->s_ox.ensure((size_t)Kalloc * 8))) return rc;
    if ((rc = c->s_ov.ensure((size_t)Kalloc * 8))) return rc;
    const size_t cells = (size_t)H * S;
    if ((rc = c->s_misc0.ensure(cells))) return rc;
    if ((rc = c->s_misc1.ensure(cells * 8))) return rc;
    if ((rc = c->s_misc2.ensure((size_t)S * 8))) return rc;
    HIPCHK(hipMemcpy(c->s_ego.p, state5, 5 * 8, hipMemcpyHostToDevice));
    int32_t kk = k;
    HIPCHK(hipMemcpy(c->s_k.p, &kk, 4, hipMemcpyHostToDevice));
    if (k > 0) {
        HIPCHK(hipMemcpy(c->s_ox.p, ox, (size_t)k * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->s_ov.p, ov, (size_t)k * 8, hipMemcpyHostToDevice));
    }
    CarTab tab{c->tab_edge.as<double>(), c->tab_win.as<int>(), c->tab_nact.as<int>(), c->tab_nums.as<int>()};
    unsigned *counters = c->counters.as<unsigned>();
    if (Kalloc <= 8) launch_predict<8>(dp, 1, Kalloc, c->s_ego.as<double>(), c->s_k.as<int>(), c->s_ox.as<double>(), c->s_ov.as<double>(), tab, counters, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, c->sticky.as<unsigned>());
    else if (Kalloc <= 16) launch_predict<16>(dp, 1, Kalloc, c->s_ego.as<double>(), c->s_k.as<int>(), c->s_ox.as<double>(), c->s_ov.as<double>(), tab, counters, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, c->sticky.as<unsigned>());
    else launch_predict<32>(dp, 1, Kalloc, c->s_ego.as<double>(), c->s_k.as<int>(), c->s_ox.as<double>(), c->s_ov.as<double>(), tab, counters, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, c->sticky.as<unsigned>());
    dim3 grid((S + 255) / 256, H);
    hipLaunchKernelGGL(k_build_grid, grid, dim3(256), 0, nullptr, dp, tab, Kalloc, start_s, S, c->s_misc0.as<uint8_t>(),
                       c->s_misc1.as<double>(), c->s_misc2.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(obstacles, c->s_misc0.p, cells, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(distances, c->s_misc1.p, cells * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(s_values, c->s_misc2.p, (size_t)S * 8, hipMemcpyDeviceToHost));
    host_t_values(p, H, t_values);
    return STMPC_OK;
}

int stmpc_predict_batch(stmpc_ctx *c, const stmpc_params *p, int mode, int N, int Kmax, const double *ego4,
                        const int32_t *k, const double *ox, const double *ov, const double *sel, double dt,
                        double mcd, double *ego4_out, double *ox_out, double *ov_out, int32_t *crashed) {
    return stmpc_predict_batch_acc(c, p, mode, N, Kmax, ego4, k, ox, ov, sel, dt, mcd, ego4_out, ox_out, ov_out, crashed, nullptr);
}

int stmpc_abi_version(void) { return STMPC_ABI_VERSION; }

int stmpc_check_error(stmpc_ctx *c) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    unsigned flags[2] = {0, 0}, cur = 0;
    HIPCHK(hipMemcpy(flags, c->sticky.p, sizeof flags, hipMemcpyDeviceToHost));
    if (c->counters.p) HIPCHK(hipMemcpy(&cur, (const unsigned *)c->counters.p + STMPC_CNT_ERR, sizeof cur, hipMemcpyDeviceToHost));
    if (flags[0] || flags[1] || cur) {
        HIPCHK(hipMemset(c->sticky.p, 0, sizeof flags));
        if (cur) HIPCHK(hipMemset((unsigned *)c->counters.p + STMPC_CNT_ERR, 0, sizeof cur));
    }
    if (flags[0] || cur) return fail(STMPC_EINTERNAL, "solver error flag set on device (an episode of an earlier batch may not have been solved)");
    if (flags[1]) return fail(STMPC_EINVAL, "finer_fit: a fine grid longer than STMPC_QP_NMAX samples is not supported (the commanded speed of that state is not valid)");
    return STMPC_OK;
}

int stmpc_predict_batch_acc(stmpc_ctx *c, const stmpc_params *p, int mode, int N, int Kmax, const double *ego4,
                            const int32_t *k, const double *ox, const double *ov, const double *sel, double dt,
                            double mcd, double *ego4_out, double *ox_out, double *ov_out, int32_t *crashed, double *oa_out) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N < 0 || Kmax < 0 || Kmax > STMPC_KMAX_LIMIT || (mode != 0 && mode != 1)) return fail(STMPC_EINVAL, "bad N/Kmax/mode");
    if (N == 0) return STMPC_OK;
    if (!ego4 || !k || !ego4_out || !crashed || (mode == 0 && !sel)) return fail(STMPC_EINVAL, "NULL host pointer");
    if (Kmax > 0 && (!ox || !ov || !ox_out || !ov_out)) return fail(STMPC_EINVAL, "NULL host pointer (vehicles)");
    for (int i = 0; i < N; ++i) if (k[i] < 0 || k[i] > Kmax) return fail(STMPC_EINVAL, "k_count[i] outside [0, Kmax]");
    HIPCHK(hipSetDevice(c->device));
    DevP dp;
    int rc = make_devp(p, &dp);
    if (rc) return rc;
    const int Kalloc = Kmax > 0 ? Kmax : 1;
    if ((rc = c->s_ego.ensure((size_t)N * 5 * 8))) return rc;
    if ((rc = c->s_k.ensure((size_t)N * 4))) return rc;
    if ((rc = c->s_ox.ensure((size_t)N * Kalloc * 8))) return rc;
    if ((rc = c->s_ov.ensure((size_t)N * Kalloc * 8))) return rc;
    if ((rc = c->s_misc0.ensure((size_t)N * 8))) return rc;
    if ((rc = c->s_misc1.ensure((size_t)N * 4 * 8))) return rc;
    if ((rc = c->s_misc2.ensure((size_t)N * Kalloc * 8))) return rc;
    if ((rc = c->s_misc3.ensure((size_t)N * Kalloc * 8))) return rc;
    if ((rc = c->s_crash.ensure((size_t)N * 4))) return rc;
    if (oa_out && (rc = c->s_pd.ensure((size_t)N * Kalloc * 8))) return rc;
    if (oa_out) HIPCHK(hipMemset(c->s_pd.p, 0, (size_t)N * Kalloc * 8));
    HIPCHK(hipMemcpy(c->s_ego.p, ego4, (size_t)N * 4 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->s_k.p, k, (size_t)N * 4, hipMemcpyHostToDevice));
    if (Kmax > 0) {
        HIPCHK(hipMemcpy(c->s_ox.p, ox, (size_t)N * Kmax * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->s_ov.p, ov, (size_t)N * Kmax * 8, hipMemcpyHostToDevice));
    }
    if (mode == 0) HIPCHK(hipMemcpy(c->s_misc0.p, sel, (size_t)N * 8, hipMemcpyHostToDevice));
    int blocks = (N + 63) / 64;
#define STMPC_LAUNCH_STEP(KM)                                                                                         \
    hipLaunchKernelGGL(k_predict_step<KM>, dim3(blocks), dim3(64), 0, nullptr, dp, mode, N, Kalloc, c->s_ego.as<double>(), \
                       c->s_k.as<int>(), c->s_ox.as<double>(), c->s_ov.as<double>(), c->s_misc0.as<double>(), dt, mcd,   \
                       c->s_misc1.as<double>(), c->s_misc2.as<double>(), c->s_misc3.as<double>(), c->s_crash.as<int>(), \
                       oa_out ? c->s_pd.as<double>() : (double *)nullptr)
    if (Kalloc <= 8) STMPC_LAUNCH_STEP(8);
    else if (Kalloc <= 16) STMPC_LAUNCH_STEP(16);
    else STMPC_LAUNCH_STEP(32);
#undef STMPC_LAUNCH_STEP
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(ego4_out, c->s_misc1.p, (size_t)N * 4 * 8, hipMemcpyDeviceToHost));
    if (Kmax > 0) {
        HIPCHK(hipMemcpy(ox_out, c->s_misc2.p, (size_t)N * Kmax * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ov_out, c->s_misc3.p, (size_t)N * Kmax * 8, hipMemcpyDeviceToHost));
    }
    HIPCHK(hipMemcpy(crashed, c->s_crash.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (oa_out && Kmax > 0) HIPCHK(hipMemcpy(oa_out, c->s_pd.p, (size_t)N * Kmax * 8, hipMemcpyDeviceToHost));
    return STMPC_OK;
}

int stmpc_fastdiv2_check(double d, double *zl) {
    double z = 0.0;
    const bool ok = fastdiv2_ok(d, &z);
    if (zl) *zl = z;
    return ok ? 1 : 0;
}

int stmpc_probe_arith(stmpc_ctx *c, int op, const double *a, const double *b, double *out, int n) {
    if (!c || !a || !out || n < 0) return fail(STMPC_EINVAL, "bad argument");
    if (n == 0) return STMPC_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = c->s_misc0.ensure((size_t)n * 8))) return rc;
    if ((rc = c->s_misc1.ensure((size_t)n * 8))) return rc;
    if ((rc = c->s_misc2.ensure((size_t)n * 8))) return rc;
    HIPCHK(hipMemcpy(c->s_misc0.p, a, (size_t)n * 8, hipMemcpyHostToDevice));
    if (b) HIPCHK(hipMemcpy(c->s_misc1.p, b, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3((n + 255) / 256), dim3(256), 0, nullptr, op, c->s_misc0.as<double>(),
                       b ? c->s_misc1.as<double>() : nullptr, c->s_misc2.as<double>(), n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, c->s_misc2.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return STMPC_OK;
}

}  // extern "C"

namespace {
int make_ffconst(const stmpc_params *p, double dt, double cdt, int maxiters, FFConst *k) {
    if (!p) return fail(STMPC_EINVAL, "params is NULL");
    if (!(dt > 0) || !(cdt > 0)) return fail(STMPC_EINVAL, "delta_t and coarse_delta_t must be positive");
    if (maxiters < 0) return fail(STMPC_EINVAL, "maxiters must be >= 0");
    memset(k, 0, sizeof *k);
    const double dt2 = host_pow(dt, 2.0), dt3 = host_pow(dt, 3.0);   // delta_t ** 2, delta_t ** 3 (st.py:626,644)
    k->dt = dt; k->cdt = cdt; k->dt2 = dt2;
    k->cv = 1.0 / dt;                                                // st.py:613
    k->ca1 = 1.0 / dt2; k->ca2 = 2.0 / dt2;                          // st.py:629-633
    k->cj1 = 1.0 / dt3; k->cj2 = 2.0 / dt3; k->cj3 = 3.0 / dt3;      // st.py:646-658
    k->v_max = p->v_max; k->a_max = p->a_max; k->a_min = p->a_min; k->j_max = p->j_max; k->j_min = p->j_min;
    k->car_length = p->car_length;
    k->maxiters = maxiters;
    return STMPC_OK;
}

// Lanes per problem: the smallest of 16/32/64 that holds the longest fine grid a path of Hs samples can produce
// (st.py:590-595); a wavefront then carries 64/GW problems.  The result bits do not depend on the choice.
int ff_group_width(int Hs, double dt, double cdt) {
    const double t_last = (double)(Hs - 1) * cdt;
    int n = (int)rint(t_last / dt + 1.0);
    if ((double)(n - 1) * dt > t_last) n -= 1;
    if (Hs > 32 || n > 32) return 64;       // the coarse path itself is held one sample per lane
    if (Hs > 16 || n > 16) return 32;
    return 16;
}

void launch_finer_fit(const FFArgs &a, bool bounds, int gw, hipStream_t st) {
    const dim3 block(64), grid((a.N + (64 / gw) - 1) / (64 / gw));
#define STMPC_FF(NF_, GW_) hipLaunchKernelGGL((k_finer_fit<NF_, GW_>), grid, block, 0, st, a)
    if (bounds) { if (gw == 16) STMPC_FF(8, 16); else if (gw == 32) STMPC_FF(8, 32); else STMPC_FF(8, 64); }
    else { if (gw == 16) STMPC_FF(6, 16); else if (gw == 32) STMPC_FF(6, 32); else STMPC_FF(6, 64); }
#undef STMPC_FF
}
}  // namespace

extern "C" {

int stmpc_finer_fit_batch(stmpc_ctx *c, const stmpc_params *p, double dt, double cdt, int maxiters, int N, int Hs,
                          const double *s_seq, const int32_t *len, const double *v0, const double *a0, const double *bac,
                          int n_max, double *out, int32_t *out_len, int32_t *iters) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N < 0 || Hs < 1 || Hs > 64 || n_max < 1) return fail(STMPC_EINVAL, "N, Hs (1..64) or n_max out of range");
    if (N == 0) return STMPC_OK;
    if (!s_seq || !len || !v0 || !a0 || !out || !out_len) return fail(STMPC_EINVAL, "NULL host pointer");
    for (int i = 0; i < N; ++i) if (len[i] < 1 || len[i] > Hs) return fail(STMPC_EINVAL, "len[i] outside [1, Hs]");
    FFArgs a;
    memset(&a, 0, sizeof a);
    int rc = make_ffconst(p, dt, cdt, maxiters, &a.k);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = c->f_seq.ensure((size_t)N * Hs * 8))) return rc;
    if ((rc = c->f_len.ensure((size_t)N * 4))) return rc;
    if ((rc = c->f_v0.ensure((size_t)N * 8))) return rc;
    if ((rc = c->f_a0.ensure((size_t)N * 8))) return rc;
    if ((rc = c->f_out.ensure((size_t)N * n_max * 8))) return rc;
    if ((rc = c->f_olen.ensure((size_t)N * 4))) return rc;
    if ((rc = c->f_iters.ensure((size_t)N * 4))) return rc;
    HIPCHK(hipMemcpy(c->f_seq.p, s_seq, (size_t)N * Hs * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->f_len.p, len, (size_t)N * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->f_v0.p, v0, (size_t)N * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->f_a0.p, a0, (size_t)N * 8, hipMemcpyHostToDevice));
    if (bac) {
        if ((rc = c->f_bac.ensure((size_t)N * 32))) return rc;
        HIPCHK(hipMemcpy(c->f_bac.p, bac, (size_t)N * 32, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemset(c->f_out.p, 0, (size_t)N * n_max * 8));
    a.N = N; a.Hs = Hs; a.n_max = n_max; a.use_qp = 1;
    a.s_seq = c->f_seq.as<double>(); a.len = c->f_len.as<int>(); a.v0 = c->f_v0.as<double>(); a.a0 = c->f_a0.as<double>();
    a.bac = bac ? c->f_bac.as<double>() : nullptr;
    a.out = c->f_out.as<double>(); a.out_len = c->f_olen.as<int>(); a.iters = c->f_iters.as<int>();
    launch_finer_fit(a, bac != nullptr, ff_group_width(Hs, dt, cdt), nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, c->f_out.p, (size_t)N * n_max * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_len, c->f_olen.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (iters) HIPCHK(hipMemcpy(iters, c->f_iters.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    return STMPC_OK;
}

// st.do_st_control on device buffers.  `refused`: the word a path that cannot be re-sampled raises (the public entry: the context's sticky flag,
// so that stmpc_check_error reports STMPC_EINVAL; the combined controller: null -- there k_cc_decide raises it only when that speed is used).
static int st_control_device(stmpc_ctx *c, const stmpc_params *p, double tick, int N, int Kmax, const double *d_ego,
                             const int32_t *d_k, const double *d_ox, const double *d_ov, int32_t *d_path,
                             int32_t *d_bt, double *d_cost, double *d_speed, double *d_fine, int32_t *d_fine_len,
                             void *stream, unsigned *refused) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N == 0) return STMPC_OK;
    if (!d_speed) return fail(STMPC_EINVAL, "NULL device pointer (speed)");
    int rc = stmpc_solve_batch_device(c, p, N, Kmax, d_ego, d_k, d_ox, d_ov, d_path, d_bt, d_cost, nullptr, nullptr, stream);
    if (rc) return rc;
    FFArgs a;
    memset(&a, 0, sizeof a);
    const int H = stmpc_num_t(p);
    double tv[STMPC_MAXH];
    host_t_values(p, H, tv);
    // finer_fit is called with (TICK_LENGTH, T_DISCRETIZATION) = the settings, not the arange spacing (st.py:771-772)
    if ((rc = make_ffconst(p, tick, p->dt, c->qp_maxiters, &a.k))) return rc;
    a.N = N; a.Hs = H; a.n_max = STMPC_QP_NMAX; a.use_qp = (tick < p->dt) ? 1 : 0;
    a.path_idx = d_path; a.best_t = d_bt; a.ego = d_ego; a.ds = p->ds;
    a.out = d_fine; a.out_len = d_fine_len; a.speed = d_speed;
    a.refused = refused;
    launch_finer_fit(a, false, a.use_qp ? ff_group_width(H, tick, p->dt) : 64, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return STMPC_OK;
}

int stmpc_st_control_batch_device(stmpc_ctx *c, const stmpc_params *p, double tick, int N, int Kmax, const double *d_ego,
                                  const int32_t *d_k, const double *d_ox, const double *d_ov, int32_t *d_path,
                                  int32_t *d_bt, double *d_cost, double *d_speed, double *d_fine, int32_t *d_fine_len,
                                  void *stream) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    // a path that cannot be re-sampled (its speed is NaN): stmpc_check_error returns STMPC_EINVAL
    return st_control_device(c, p, tick, N, Kmax, d_ego, d_k, d_ox, d_ov, d_path, d_bt, d_cost, d_speed, d_fine, d_fine_len, stream, c->sticky.as<unsigned>() + 1);
}

int stmpc_st_control_batch(stmpc_ctx *c, const stmpc_params *p, double tick, int N, int Kmax, const double *ego,
                           const int32_t *k, const double *ox, const double *ov, double *speed, int32_t *bt,
                           int32_t *path, double *cost, double *fine, int32_t *fine_len) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N < 0 || Kmax < 0 || Kmax > STMPC_KMAX_LIMIT) return fail(STMPC_EINVAL, "N or Kmax out of range");
    if (N == 0) return STMPC_OK;
    if (!ego || !k || !speed || !bt) return fail(STMPC_EINVAL, "NULL host pointer");
    if (Kmax > 0 && (!ox || !ov)) return fail(STMPC_EINVAL, "NULL host pointer (other_x/other_v)");
    for (int i = 0; i < N; ++i) if (k[i] < 0 || k[i] > Kmax) return fail(STMPC_EINVAL, "k_count[i] outside [0, Kmax]");
    HIPCHK(hipSetDevice(c->device));
    const int H = stmpc_num_t(p);
    if (H < 2 || H > STMPC_H_LIMIT) return fail(STMPC_EINVAL, "number of time layers must be in [2, 64]");
    int rc;
    const int Kalloc = Kmax > 0 ? Kmax : 1;
    if ((rc = c->s_ego.ensure((size_t)N * 5 * 8))) return rc;
    if ((rc = c->s_k.ensure((size_t)N * 4))) return rc;
    if ((rc = c->s_ox.ensure((size_t)N * Kalloc * 8))) return rc;
    if ((rc = c->s_ov.ensure((size_t)N * Kalloc * 8))) return rc;
    if ((rc = c->s_path.ensure((size_t)N * H * 4))) return rc;
    if ((rc = c->s_bt.ensure((size_t)N * 4))) return rc;
    if ((rc = c->s_cost.ensure((size_t)N * 8))) return rc;
    if ((rc = c->f_speed.ensure((size_t)N * 8))) return rc;
    if ((rc = c->f_out.ensure((size_t)N * STMPC_QP_NMAX * 8))) return rc;
    if ((rc = c->f_olen.ensure((size_t)N * 4))) return rc;
    HIPCHK(hipMemcpy(c->s_ego.p, ego, (size_t)N * 5 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->s_k.p, k, (size_t)N * 4, hipMemcpyHostToDevice));
    if (Kmax > 0) {
        HIPCHK(hipMemcpy(c->s_ox.p, ox, (size_t)N * Kmax * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->s_ov.p, ov, (size_t)N * Kmax * 8, hipMemcpyHostToDevice));
    }
    if (fine) HIPCHK(hipMemset(c->f_out.p, 0, (size_t)N * STMPC_QP_NMAX * 8));
    rc = stmpc_st_control_batch_device(c, p, tick, N, Kmax, c->s_ego.as<double>(), c->s_k.as<int32_t>(), c->s_ox.as<double>(),
                                       c->s_ov.as<double>(), c->s_path.as<int32_t>(), c->s_bt.as<int32_t>(), c->s_cost.as<double>(),
                                       c->f_speed.as<double>(), c->f_out.as<double>(), c->f_olen.as<int>(), nullptr);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(speed, c->f_speed.p, (size_t)N * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(bt, c->s_bt.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (path) HIPCHK(hipMemcpy(path, c->s_path.p, (size_t)N * H * 4, hipMemcpyDeviceToHost));
    if (cost) HIPCHK(hipMemcpy(cost, c->s_cost.p, (size_t)N * 8, hipMemcpyDeviceToHost));
    if (fine) HIPCHK(hipMemcpy(fine, c->f_out.p, (size_t)N * STMPC_QP_NMAX * 8, hipMemcpyDeviceToHost));
    if (fine_len) HIPCHK(hipMemcpy(fine_len, c->f_olen.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    stmpc_stats s;
    if ((rc = stmpc_get_stats(c, &s))) return rc;
    // (this entry is synchronous: a refused re-sampling is its own error, not left for a later stmpc_check_error)
    unsigned refused = 0;
    HIPCHK(hipMemcpy(&refused, c->sticky.as<unsigned>() + 1, sizeof refused, hipMemcpyDeviceToHost));
    if (refused) {
        HIPCHK(hipMemset(c->sticky.as<unsigned>() + 1, 0, sizeof refused));
        return fail(STMPC_EINVAL, "finer_fit: a fine grid longer than STMPC_QP_NMAX samples is not supported (speed = NaN, fine_len = -1 for those states)");
    }
    return STMPC_OK;
}

}  // extern "C"

namespace {
int make_ccfg(const stmpc_params *p, const stmpc_combined_cfg *g, CCfg *c) {
    if (!p || !g) return fail(STMPC_EINVAL, "params / combined cfg is NULL");
    if (!(g->tick_length > 0)) return fail(STMPC_EINVAL, "tick_length must be positive");
    memset(c, 0, sizeof *c);
    c->tick = g->tick_length; c->comb_min_dist = p->comb_min_dist; c->stop_x = g->stop_x;
    c->a_max = p->a_max; c->a_min = p->a_min; c->v_max = p->v_max; c->desired_speed = p->v_des;
    c->rollout_length = g->rollout_length > 1 ? g->rollout_length : 1;            // max(ROLLOUT_LENGTH, 1), dqn.py:129
    c->st_test_rollouts = g->st_test_rollouts; c->check_rollout_crash = g->check_rollout_crash; c->limit_speed = g->limit_dqn_speed;
    c->test_rollout_state = g->test_rollout_state; c->strictly_better = g->test_st_strictly_better; c->remember_last = g->remember_last_choice;
    if (c->rollout_length > STMPC_ROLLOUT_LIMIT) return fail(STMPC_EINVAL, "rollout_length above STMPC_ROLLOUT_LIMIT");
    return STMPC_OK;
}
int cc_ensure(stmpc_ctx *c, int N, int K, int R) {
    int rc;
    if ((rc = c->cc_live.ensure((size_t)N * 4))) return rc;
    if ((rc = c->cc_hist_len.ensure((size_t)N * 4))) return rc;
    if ((rc = c->cc_crash_pred.ensure((size_t)N * 4))) return rc;
    if ((rc = c->cc_have_test.ensure((size_t)N * 4))) return rc;
    if ((rc = c->cc_sel.ensure((size_t)N * 8))) return rc;
    if ((rc = c->cc_rollout_s.ensure((size_t)N * (R + 1) * 8))) return rc;
    if ((rc = c->cc_test_ego.ensure((size_t)N * 4 * 8))) return rc;
    if ((rc = c->cc_test_ox.ensure((size_t)N * K * 8))) return rc;
    if ((rc = c->cc_test_ov.ensure((size_t)N * K * 8))) return rc;
    c->cc_N = N; c->cc_K = K; c->cc_R = R;
    return STMPC_OK;
}
CCState cc_state(stmpc_ctx *c) {
    return CCState{c->cc_live.as<int>(), c->cc_hist_len.as<int>(), c->cc_crash_pred.as<int>(), c->cc_have_test.as<int>(), c->cc_sel.as<double>(),
                   c->cc_rollout_s.as<double>(), c->cc_test_ego.as<double>(), c->cc_test_ox.as<double>(), c->cc_test_ov.as<double>()};
}
}  // namespace

extern "C" {

int stmpc_rollout_step_device(stmpc_ctx *c, const stmpc_params *p, const stmpc_combined_cfg *g, int N, int Kmax, int step,
                              const double *d_ego5_start, double *d_cur_ego4, const int32_t *d_k, double *d_cur_ox, double *d_cur_ov,
                              double *d_cur_oa, const double *d_action, void *stream) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N < 0 || Kmax < 0 || Kmax > STMPC_KMAX_LIMIT || step < 1) return fail(STMPC_EINVAL, "N, Kmax or step out of range");
    if (N == 0) return STMPC_OK;
    if (!d_ego5_start || !d_cur_ego4 || !d_k || !d_action) return fail(STMPC_EINVAL, "NULL device pointer");
    if (Kmax > 0 && (!d_cur_ox || !d_cur_ov)) return fail(STMPC_EINVAL, "NULL device pointer (vehicles)");
    HIPCHK(hipSetDevice(c->device));
    DevP dp; CCfg cc;
    int rc = make_devp(p, &dp);
    if (rc) return rc;
    if ((rc = make_ccfg(p, g, &cc))) return rc;
    const int Kalloc = Kmax > 0 ? Kmax : 1;
    if (step == 1) { if ((rc = cc_ensure(c, N, Kalloc, cc.rollout_length))) return rc; }
    else if (c->cc_N != N || c->cc_K != Kalloc || c->cc_R != cc.rollout_length) return fail(STMPC_EINVAL, "rollout step > 1 does not continue the rollout begun with step 1");
    CCState st = cc_state(c);
    const int blocks = (N + 63) / 64;
#define STMPC_RS(KM) hipLaunchKernelGGL(k_rollout_step<KM>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, dp, cc, N, Kalloc, step, d_ego5_start, d_cur_ego4, d_k, d_cur_ox, d_cur_ov, d_cur_oa, d_action, st)
    if (Kalloc <= 8) STMPC_RS(8); else if (Kalloc <= 16) STMPC_RS(16); else STMPC_RS(32);
#undef STMPC_RS
    HIPCHK(hipGetLastError());
    return STMPC_OK;
}

int stmpc_policy_features_device(stmpc_ctx *c, const stmpc_policy_features_cfg *f, int N, int Kmax, int step, const double *d_cur_ego4, const int32_t *d_k,
                                 const double *d_cur_ox, const double *d_cur_ov, const double *d_cur_oa, int32_t *d_evals, float *d_feat, int feat_stride, void *stream) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (!f) return fail(STMPC_EINVAL, "features cfg is NULL");
    if (N < 0 || Kmax < 0 || Kmax > STMPC_KMAX_LIMIT || step < 1) return fail(STMPC_EINVAL, "N, Kmax or step out of range");
    if (f->cars_ahead < 0 || f->cars_behind < 0 || f->cars_ahead > STMPC_KMAX_LIMIT || f->cars_behind > STMPC_KMAX_LIMIT) return fail(STMPC_EINVAL, "cars_ahead / cars_behind out of range");
    const int len = stmpc_policy_features_len(f);
    if (feat_stride < len) return fail(STMPC_EINVAL, "feat_stride is shorter than the feature vector");
    if (f->normalize && (!(f->max_speed > 0) || !(f->sensor_radius > 0))) return fail(STMPC_EINVAL, "max_speed and sensor_radius must be positive");
    if (N == 0) return STMPC_OK;
    if (!d_cur_ego4 || !d_k || !d_feat) return fail(STMPC_EINVAL, "NULL device pointer");
    if (Kmax > 0 && (!d_cur_ox || !d_cur_ov)) return fail(STMPC_EINVAL, "NULL device pointer (vehicles)");
    if (f->time_feature && !d_evals) return fail(STMPC_EINVAL, "time_feature needs the evaluation counters");
    const int *live = nullptr;
    if (step > 1) {                // later rollout steps: only states whose rollout is still going on are evaluated by the reference (dqn.py:129-133)
        if (c->cc_N != N) return fail(STMPC_EINVAL, "step > 1 without a rollout of this size in the context (stmpc_rollout_step_device)");
        live = c->cc_live.as<int>();
    }
    HIPCHK(hipSetDevice(c->device));
    FeatCfg fc;
    fc.max_speed = f->max_speed; fc.sensor_radius = f->sensor_radius; fc.time_scale = (float)f->time_scale;
    fc.cars_ahead = f->cars_ahead; fc.cars_behind = f->cars_behind; fc.use_accel = f->use_acceleration != 0; fc.use_speed_diff = f->use_speed_difference != 0;
    fc.normalize = f->normalize != 0; fc.time_feature = f->time_feature != 0;
    hipLaunchKernelGGL(k_policy_features, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, fc, N, Kmax, d_cur_ego4, d_k, d_cur_ox, d_cur_ov, d_cur_oa, live,
                       d_evals, d_feat, feat_stride);
    HIPCHK(hipGetLastError());
    return STMPC_OK;
}

int stmpc_policy_features_len(const stmpc_policy_features_cfg *f) {
    if (!f) return 0;
    return (f->cars_ahead + f->cars_behind) * (f->use_acceleration ? 4 : 3) + 4 + (f->time_feature ? 1 : 0);
}

}  // extern "C"

// ---- the policy network itself (optional: the caller may keep it in its own framework and only use stmpc_policy_features_device) ----
struct stmpc_actor {
    int device = 0;
    DevBuf p0, b0, p1, b1, w2;
    ActorDev dev{};
    size_t lds = 0;
};

namespace {
// [column tile][k block][lane = j + 16 kk][4]: W[n0 + j][k0 + 4 kk + s], zero outside [rows) x [cols)
std::vector<float> pack_layer(const float *W, int rows, int cols, int rows_p, int cols_p) {
    std::vector<float> out((size_t)rows_p * cols_p, 0.f);
    const int kblocks = cols_p / 16;
    for (int nt = 0; nt < rows_p / 16; ++nt)
        for (int kb = 0; kb < kblocks; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int q = 0; q < 4; ++q) {
                    const int n = nt * 16 + (lane & 15), k = kb * 16 + 4 * (lane >> 4) + q;
                    if (n < rows && k < cols) out[(((size_t)nt * kblocks + kb) * 64 + lane) * 4 + q] = W[(size_t)n * cols + k];
                }
    return out;
}
int upload(DevBuf &b, const std::vector<float> &v) {
    int rc = b.ensure(v.size() * sizeof(float));
    if (rc) return rc;
    HIPCHK(hipMemcpy(b.p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return STMPC_OK;
}
}  // namespace

extern "C" {

int stmpc_actor_create(stmpc_ctx *c, int n_in, int h1, int h2, const float *w0, const float *b0, const float *w1, const float *b1, const float *w2,
                       const float *b2, double tanh_scale, double tanh_mean, stmpc_actor **out) {
    if (!c || !out) return fail(STMPC_EINVAL, "NULL argument");
    *out = nullptr;
    if (!w0 || !b0 || !w1 || !b1 || !w2 || !b2) return fail(STMPC_EINVAL, "NULL weight pointer");
    if (n_in < 1 || n_in > AT_KIN || h1 < 1 || h1 > 1024 || h2 < 1 || h2 > 1024) return fail(STMPC_EINVAL, "actor shape out of range (n_in <= 32, hidden widths <= 1024)");
    HIPCHK(hipSetDevice(c->device));
    const int h1p = (h1 + 15) & ~15, h2p = (h2 + 15) & ~15;
    const size_t lds = ((size_t)AT_TM * AT_KIN + (size_t)AT_TM * (h1p + 4) + (size_t)AT_TM * (h2p + 4)) * sizeof(float);
    if (lds + 1024 > (size_t)c->lds_per_block) return fail(STMPC_EINVAL, "actor too wide for one workgroup's LDS");
    stmpc_actor *a = new stmpc_actor();
    a->device = c->device; a->lds = lds;
    std::vector<float> vb0(h1p, 0.f), vb1(h2p, 0.f), vw2(h2p, 0.f);
    for (int i = 0; i < h1; ++i) vb0[i] = b0[i];
    for (int i = 0; i < h2; ++i) { vb1[i] = b1[i]; vw2[i] = w2[i]; }
    int rc;
    if ((rc = upload(a->p0, pack_layer(w0, h1, n_in, h1p, AT_KIN))) || (rc = upload(a->b0, vb0)) || (rc = upload(a->p1, pack_layer(w1, h2, h1, h2p, h1p))) ||
        (rc = upload(a->b1, vb1)) || (rc = upload(a->w2, vw2))) { stmpc_actor_destroy(a); return rc; }
    a->dev.p0 = a->p0.as<float>(); a->dev.b0 = a->b0.as<float>(); a->dev.p1 = a->p1.as<float>(); a->dev.b1 = a->b1.as<float>(); a->dev.w2 = a->w2.as<float>();
    a->dev.b2 = b2[0]; a->dev.scale = (float)tanh_scale; a->dev.mean = (float)tanh_mean; a->dev.n_in = n_in; a->dev.h1p = h1p; a->dev.h2p = h2p;
    // (the attribute belongs to the kernel, not to this actor: only ever raised, so that a narrower actor created later does not take the
    // dynamic LDS away from a wider one that is still in use)
    static size_t actor_lds_max[16] = {0};
    size_t &lds_max = actor_lds_max[(unsigned)c->device & 15u];
    if (lds > 48 * 1024 && lds > lds_max) {
        if (hipFuncSetAttribute((const void *)k_actor_eval, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            stmpc_actor_destroy(a);
            return fail(STMPC_EHIP, "hipFuncSetAttribute(k_actor_eval, dynamic LDS) failed");
        }
        lds_max = lds;
    }
    *out = a;
    return STMPC_OK;
}

void stmpc_actor_destroy(stmpc_actor *a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    a->p0.release(); a->b0.release(); a->p1.release(); a->b1.release(); a->w2.release();
    delete a;
}

int stmpc_actor_eval_device(stmpc_ctx *c, const stmpc_actor *a, const stmpc_policy_features_cfg *f, int N, int Kmax, int step, const double *d_cur_ego4,
                            const int32_t *d_k, const double *d_cur_ox, const double *d_cur_ov, const double *d_cur_oa, int32_t *d_evals, float *d_feat,
                            int feat_stride, double *d_jerk, void *stream) {
    if (!c || !a || !f) return fail(STMPC_EINVAL, "NULL argument");
    if (a->device != c->device) return fail(STMPC_EINVAL, "actor and context are on different devices");
    if (N < 0 || Kmax < 0 || Kmax > STMPC_KMAX_LIMIT || step < 1) return fail(STMPC_EINVAL, "N, Kmax or step out of range");
    if (f->cars_ahead < 0 || f->cars_behind < 0 || f->cars_ahead > STMPC_KMAX_LIMIT || f->cars_behind > STMPC_KMAX_LIMIT) return fail(STMPC_EINVAL, "cars_ahead / cars_behind out of range");
    if (stmpc_policy_features_len(f) != a->dev.n_in) return fail(STMPC_EINVAL, "the actor's input width is not the length of this state vector");
    if (d_feat && feat_stride < a->dev.n_in) return fail(STMPC_EINVAL, "feat_stride is shorter than the feature vector");
    if (f->normalize && (!(f->max_speed > 0) || !(f->sensor_radius > 0))) return fail(STMPC_EINVAL, "max_speed and sensor_radius must be positive");
    if (N == 0) return STMPC_OK;
    if (!d_cur_ego4 || !d_k || !d_jerk) return fail(STMPC_EINVAL, "NULL device pointer");
    if (Kmax > 0 && (!d_cur_ox || !d_cur_ov)) return fail(STMPC_EINVAL, "NULL device pointer (vehicles)");
    if (f->time_feature && !d_evals) return fail(STMPC_EINVAL, "time_feature needs the evaluation counters");
    const int *live = nullptr;
    if (step > 1) {
        if (c->cc_N != N) return fail(STMPC_EINVAL, "step > 1 without a rollout of this size in the context (stmpc_rollout_step_device)");
        live = c->cc_live.as<int>();
    }
    HIPCHK(hipSetDevice(c->device));
    FeatCfg fc;
    fc.max_speed = f->max_speed; fc.sensor_radius = f->sensor_radius; fc.time_scale = (float)f->time_scale;
    fc.cars_ahead = f->cars_ahead; fc.cars_behind = f->cars_behind; fc.use_accel = f->use_acceleration != 0; fc.use_speed_diff = f->use_speed_difference != 0;
    fc.normalize = f->normalize != 0; fc.time_feature = f->time_feature != 0;
    hipLaunchKernelGGL(k_actor_eval, dim3((N + AT_TM - 1) / AT_TM), dim3(AT_THREADS), a->lds, (hipStream_t)stream, fc, a->dev, N, Kmax, d_cur_ego4, d_k, d_cur_ox, d_cur_ov,
                       d_cur_oa, live, d_evals, d_feat, feat_stride, d_jerk);
    HIPCHK(hipGetLastError());
    return STMPC_OK;
}

int stmpc_combined_counts(stmpc_ctx *c, int64_t *decisions, int64_t *control_solves, int reset) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (decisions) *decisions = c->cc_ticks;
    if (control_solves) *control_solves = c->cc_control_solves;
    if (reset) { c->cc_ticks = 0; c->cc_control_solves = 0; }
    return STMPC_OK;
}

int stmpc_combined_decide_device(stmpc_ctx *c, const stmpc_params *p, const stmpc_combined_cfg *g, int N, int Kmax,
                                 const double *d_ego5_start, const int32_t *d_k, const double *d_ox_start, const double *d_ov_start,
                                 const double *d_cur_ego4, const double *d_cur_ox, const double *d_cur_ov, const double *d_first_action,
                                 const int32_t *d_last_choice_rl, int32_t *d_takeover, int32_t *d_reason, double *d_speed, void *stream) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (!p || !g) return fail(STMPC_EINVAL, "NULL parameter struct");
    if (N < 0 || Kmax < 0 || Kmax > STMPC_KMAX_LIMIT) return fail(STMPC_EINVAL, "N or Kmax out of range");
    if (N == 0) return STMPC_OK;
    if (!d_ego5_start || !d_k || !d_cur_ego4 || !d_first_action || !d_takeover || !d_reason || !d_speed) return fail(STMPC_EINVAL, "NULL device pointer");
    if (Kmax > 0 && (!d_ox_start || !d_ov_start || !d_cur_ox || !d_cur_ov)) return fail(STMPC_EINVAL, "NULL device pointer (vehicles)");
    HIPCHK(hipSetDevice(c->device));
    CCfg cc;
    int rc = make_ccfg(p, g, &cc);
    if (rc) return rc;
    const int Kalloc = Kmax > 0 ? Kmax : 1;
    if (c->cc_N != N || c->cc_K != Kalloc || c->cc_R != cc.rollout_length) return fail(STMPC_EINVAL, "no rollout of this shape in the context (call stmpc_rollout_step_device first)");
    const int H = stmpc_num_t(p);
    if (H < 2 || H > STMPC_H_LIMIT) return fail(STMPC_EINVAL, "number of time layers must be in [2, 64]");
    hipStream_t st_ = (hipStream_t)stream;
    if ((rc = c->cc_probe_ego.ensure((size_t)N * 5 * 8))) return rc;
    if ((rc = c->cc_probe_ox.ensure((size_t)N * Kalloc * 8))) return rc;
    if ((rc = c->cc_probe_ov.ensure((size_t)N * Kalloc * 8))) return rc;
    if ((rc = c->cc_path.ensure((size_t)N * H * 4))) return rc;
    if ((rc = c->cc_bt.ensure((size_t)N * 4))) return rc;
    if ((rc = c->cc_cost.ensure((size_t)N * 8))) return rc;
    if ((rc = c->cc_pcrash.ensure((size_t)N * 4))) return rc;
    if ((rc = c->cc_speed.ensure((size_t)N * 8))) return rc;
    if ((rc = c->cc_fine.ensure((size_t)N * STMPC_QP_NMAX * 8))) return rc;
    if ((rc = c->cc_fine_len.ensure((size_t)N * 4))) return rc;
    CCState st = cc_state(c);
    const int blocks = (N + 63) / 64;
    HIPCHK(hipMemsetAsync(c->cc_pcrash.p, 0, (size_t)N * 4, st_));
    // 1. feasibility probe of the rolled-out state (st.test_guaranteed_crash_from_state, dqn.py:152): one batched solve
    if (cc.test_rollout_state) {
        hipLaunchKernelGGL(k_cc_probe_state, dim3(blocks), dim3(64), 0, st_, N, Kalloc, Kmax, d_k, d_cur_ego4, d_cur_ox, d_cur_ov, st, c->cc_probe_ego.as<double>(),
                           c->cc_probe_ox.as<double>(), c->cc_probe_ov.as<double>());
        if ((rc = stmpc_solve_batch_device(c, p, N, Kmax, c->cc_probe_ego.as<double>(), d_k, c->cc_probe_ox.as<double>(), c->cc_probe_ov.as<double>(),
                                           c->cc_path.as<int32_t>(), c->cc_bt.as<int32_t>(), c->cc_cost.as<double>(), nullptr, c->cc_pcrash.as<int32_t>(), stream))) return rc;
    }
    // 2. the controller on the start state (st.do_st_control; also the path of the strictly-better comparison, dqn.py:157-164)
    HIPCHK(hipMemsetAsync(c->cc_fine.p, 0, (size_t)N * STMPC_QP_NMAX * 8, st_));
    c->cc_ticks += N;
    if (g->sparse_control && !cc.strictly_better) {
        // The reference solves the start state only when a branch of dqn.py:144-155 hands control over (2-5 % of the ticks under the shipped
        // configs); here: ordered compaction of those states, ONE host round trip for their number, the controller on the compact batch, scatter.
        if ((rc = c->cc_sel_idx.ensure((size_t)N * 4))) return rc;
        if ((rc = c->cc_sel_count.ensure(4))) return rc;
        if (!c->cc_host_count) HIPCHK(hipHostMalloc((void **)&c->cc_host_count, 4, hipHostMallocDefault));
        HIPCHK(hipMemsetAsync(c->cc_speed.p, 0xFF, (size_t)N * 8, st_));            // NaN: no controller command exists for a state the policy keeps
        HIPCHK(hipMemsetAsync(c->cc_fine_len.p, 0, (size_t)N * 4, st_));
        hipLaunchKernelGGL(k_cc_select, dim3(1), dim3(1024), 0, st_, cc, N, st, (const int *)c->cc_pcrash.as<int>(), c->cc_sel_idx.as<int>(), c->cc_sel_count.as<int>());
        HIPCHK(hipMemcpyAsync(c->cc_host_count, c->cc_sel_count.p, 4, hipMemcpyDeviceToHost, st_));
        HIPCHK(hipStreamSynchronize(st_));
        const int M = *c->cc_host_count;
        if (M < 0 || M > N) return fail(STMPC_EINTERNAL, "combined controller: selection count out of range");
        c->cc_control_solves += M;
        if (M > 0) {
            if ((rc = c->cc_c_ego.ensure((size_t)M * 5 * 8))) return rc;
            if ((rc = c->cc_c_k.ensure((size_t)M * 4))) return rc;
            if ((rc = c->cc_c_ox.ensure((size_t)M * Kalloc * 8))) return rc;
            if ((rc = c->cc_c_ov.ensure((size_t)M * Kalloc * 8))) return rc;
            if ((rc = c->cc_c_speed.ensure((size_t)M * 8))) return rc;
            if ((rc = c->cc_c_fine.ensure((size_t)M * STMPC_QP_NMAX * 8))) return rc;
            if ((rc = c->cc_c_fine_len.ensure((size_t)M * 4))) return rc;
            const int mb = (M + 63) / 64;
            hipLaunchKernelGGL(k_cc_gather, dim3(mb), dim3(64), 0, st_, M, Kalloc, Kmax, (const int *)c->cc_sel_idx.as<int>(), d_ego5_start, d_k, d_ox_start, d_ov_start,
                               c->cc_c_ego.as<double>(), c->cc_c_k.as<int>(), c->cc_c_ox.as<double>(), c->cc_c_ov.as<double>());
            HIPCHK(hipMemsetAsync(c->cc_c_fine.p, 0, (size_t)M * STMPC_QP_NMAX * 8, st_));
            if ((rc = st_control_device(c, p, g->tick_length, M, Kalloc, c->cc_c_ego.as<double>(), c->cc_c_k.as<int32_t>(), c->cc_c_ox.as<double>(), c->cc_c_ov.as<double>(),
                                        c->cc_path.as<int32_t>(), c->cc_bt.as<int32_t>(), c->cc_cost.as<double>(), c->cc_c_speed.as<double>(), c->cc_c_fine.as<double>(),
                                        c->cc_c_fine_len.as<int32_t>(), stream, nullptr))) return rc;
            hipLaunchKernelGGL(k_cc_scatter, dim3(mb), dim3(64), 0, st_, M, (const int *)c->cc_sel_idx.as<int>(), (const double *)c->cc_c_speed.as<double>(),
                               (const double *)c->cc_c_fine.as<double>(), (const int *)c->cc_c_fine_len.as<int>(), STMPC_QP_NMAX, c->cc_speed.as<double>(),
                               c->cc_fine.as<double>(), c->cc_fine_len.as<int>());
        }
    } else {
        c->cc_control_solves += N;
        if ((rc = st_control_device(c, p, g->tick_length, N, Kmax, d_ego5_start, d_k, d_ox_start, d_ov_start, c->cc_path.as<int32_t>(), c->cc_bt.as<int32_t>(),
                                    c->cc_cost.as<double>(), c->cc_speed.as<double>(), c->cc_fine.as<double>(), c->cc_fine_len.as<int32_t>(), stream, nullptr))) return rc;
    }
    // 3. the decision
    hipLaunchKernelGGL(k_cc_decide, dim3(blocks), dim3(64), 0, st_, cc, N, d_ego5_start, d_first_action, d_last_choice_rl, st, (const int *)c->cc_pcrash.as<int>(),
                       (const double *)c->cc_speed.as<double>(), (const double *)c->cc_fine.as<double>(), (const int *)c->cc_fine_len.as<int>(), STMPC_QP_NMAX,
                       d_takeover, d_reason, d_speed, c->sticky.as<unsigned>() + 1);
    HIPCHK(hipGetLastError());
    return STMPC_OK;
}

int stmpc_combined_read_state(stmpc_ctx *c, int N, int32_t *live, int32_t *hist_len, int32_t *crash_pred, double *sel_speed, double *rollout_s,
                              int32_t *have_test, double *test_ego4, double *test_ox, double *test_ov, int32_t *probe_crash, double *st_speed,
                              double *fine, int32_t *fine_len) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N != c->cc_N) return fail(STMPC_EINVAL, "no rollout of this size in the context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    const size_t K = c->cc_K, R1 = c->cc_R + 1;
    if (live) HIPCHK(hipMemcpy(live, c->cc_live.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (hist_len) HIPCHK(hipMemcpy(hist_len, c->cc_hist_len.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (crash_pred) HIPCHK(hipMemcpy(crash_pred, c->cc_crash_pred.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (sel_speed) HIPCHK(hipMemcpy(sel_speed, c->cc_sel.p, (size_t)N * 8, hipMemcpyDeviceToHost));
    if (rollout_s) HIPCHK(hipMemcpy(rollout_s, c->cc_rollout_s.p, (size_t)N * R1 * 8, hipMemcpyDeviceToHost));
    if (have_test) HIPCHK(hipMemcpy(have_test, c->cc_have_test.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (test_ego4) HIPCHK(hipMemcpy(test_ego4, c->cc_test_ego.p, (size_t)N * 4 * 8, hipMemcpyDeviceToHost));
    if (test_ox) HIPCHK(hipMemcpy(test_ox, c->cc_test_ox.p, (size_t)N * K * 8, hipMemcpyDeviceToHost));
    if (test_ov) HIPCHK(hipMemcpy(test_ov, c->cc_test_ov.p, (size_t)N * K * 8, hipMemcpyDeviceToHost));
    if (probe_crash && c->cc_pcrash.p) HIPCHK(hipMemcpy(probe_crash, c->cc_pcrash.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (st_speed && c->cc_speed.p) HIPCHK(hipMemcpy(st_speed, c->cc_speed.p, (size_t)N * 8, hipMemcpyDeviceToHost));
    if (fine && c->cc_fine.p) HIPCHK(hipMemcpy(fine, c->cc_fine.p, (size_t)N * STMPC_QP_NMAX * 8, hipMemcpyDeviceToHost));
    if (fine_len && c->cc_fine_len.p) HIPCHK(hipMemcpy(fine_len, c->cc_fine_len.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    return stmpc_check_error(c);         // (device already synchronised: just the flags)
}

int stmpc_solve_grid_no_jerk(stmpc_ctx *c, int variant, const uint8_t *obstacles, const double *s_values, int S, const double *t_values,
                             int H, double ego_start_speed, const double *distances, double *s_sequence_out) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (variant != 0 && variant != 1) return fail(STMPC_EINVAL, "variant must be 0 (no_jerk_fast) or 1 (no_jerk_djikstra)");
    if (!obstacles || !s_values || !t_values || !distances || !s_sequence_out) return fail(STMPC_EINVAL, "NULL host pointer");
    if (H < 2 || H > STMPC_H_LIMIT) return fail(STMPC_EINVAL, "num_t must be in [2, 64]");
    if (S < 2 || S > STMPC_S_LIMIT) return fail(STMPC_EINVAL, "num_s must be in [2, 65000]");
    if (variant == 1 && (size_t)H * S * S > ((size_t)1 << 28)) return fail(STMPC_EINVAL, "no_jerk_djikstra keeps H*S*S node flags: lattice too large (H*S*S > 2^28)");
    if (t_values[1] - t_values[0] == 0.0) return fail(STMPC_EINVAL, "float division by zero (delta_t == 0)");
    if (s_values[1] - s_values[0] == 0.0) return fail(STMPC_EINVAL, "float division by zero (delta_s == 0)");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    const size_t cells = (size_t)H * S, states = variant ? cells * S : cells;
    size_t cap = states * 8;
    if (cap < ((size_t)1 << 20)) cap = (size_t)1 << 20;
    if (cap > ((size_t)1 << 25)) cap = (size_t)1 << 25;               // 32 M entries = 768 MB at most
    if ((rc = c->s_misc0.ensure(cells))) return rc;
    if ((rc = c->s_misc1.ensure(cells * 8))) return rc;
    if ((rc = c->s_misc2.ensure((size_t)S * 8))) return rc;
    if ((rc = c->s_misc3.ensure((size_t)H * 8 + 16))) return rc;
    if ((rc = c->s_pd.ensure(states))) return rc;
    if ((rc = c->s_path.ensure(states * 4))) return rc;
    if ((rc = c->gscratch.ensure(cap * sizeof(NjItem)))) return rc;
    HIPCHK(hipMemcpy(c->s_misc0.p, obstacles, cells, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->s_misc1.p, distances, cells * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->s_misc2.p, s_values, (size_t)S * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(c->s_pd.p, 0, states));
    HIPCHK(hipMemset(c->s_path.p, 0, states * 4));
    NjArgs a;
    memset(&a, 0, sizeof a);
    a.triple = variant; a.S = S; a.H = H; a.v0 = ego_start_speed;
    a.obstacles = c->s_misc0.as<uint8_t>(); a.distances = c->s_misc1.as<double>(); a.s_values = c->s_misc2.as<double>();
    a.dt = t_values[1] - t_values[0];
    a.enc = c->s_pd.as<uint8_t>(); a.prev = c->s_path.as<int>(); a.heap = c->gscratch.as<NjItem>(); a.cap = cap;
    a.s_sequence = c->s_misc3.as<double>(); a.status = (int *)(c->s_misc3.as<double>() + H);
    hipLaunchKernelGGL(k_nojerk, dim3(1), dim3(64), 0, nullptr, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    int status = 0;
    HIPCHK(hipMemcpy(s_sequence_out, c->s_misc3.p, (size_t)H * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&status, (char *)c->s_misc3.p + (size_t)H * 8, 4, hipMemcpyDeviceToHost));
    if (status == 1) return fail(STMPC_ENOMEM, "no-jerk solver: heap capacity exceeded");
    if (status == 2) return fail(STMPC_EINVAL, "index out of bounds: the first layer's reachable cells leave the grid (IndexError in the reference)");
    return STMPC_OK;
}

}  // extern "C"

namespace {
int make_simcfg(const stmpc_sim_cfg *g, sim::Cfg *c) {
    if (!g) return fail(STMPC_EINVAL, "sim cfg is NULL");
    if (!(g->tick_length > 0) || !(g->other_car_speed > 0) || !(g->base_traffic_interval > 0)) return fail(STMPC_EINVAL, "tick_length, other_car_speed and base_traffic_interval must be positive");
    memset(c, 0, sizeof *c);
    c->tick = g->tick_length; c->other_speed = g->other_car_speed; c->base_interval = g->base_traffic_interval;
    c->spawn_x = g->spawn_x; c->despawn_x = g->despawn_x; c->ego_start_x = g->ego_start_x; c->ego_start_y = g->ego_start_y; c->arrive_x = g->arrive_x;
    c->sensor_radius = g->sensor_radius; c->start_speed = g->start_speed; c->start_speed_std = g->start_speed_std;
    c->min_start_speed = g->min_start_speed; c->max_start_speed = g->max_start_speed;
    if (!(g->veh_accel > 0) || !(g->veh_decel > 0) || !(g->veh_tau >= 0) || !(g->veh_length > 0) || !(g->veh_emergency_decel >= g->veh_decel) || g->veh_min_gap < 0 || g->speed_dev < 0)
        return fail(STMPC_EINVAL, "vehicle type parameters (accel, decel, tau, length, emergency decel, minGap, speed_dev) out of range");
    c->veh_accel = g->veh_accel; c->veh_decel = g->veh_decel; c->veh_min_gap = g->veh_min_gap; c->veh_tau = g->veh_tau; c->veh_emergency_decel = g->veh_emergency_decel;
    c->veh_length = g->veh_length; c->veh_width = g->veh_width; c->speed_dev = g->speed_dev;
    c->vary_interval = g->vary_traffic_start_times; c->randomize_start_speed = g->randomize_start_speed; c->max_ticks = g->max_ticks; c->seed = g->seed;
    if (g->yield_overlap != 2) return fail(STMPC_EINVAL, "stmpc_sim_cfg.yield_overlap must be 2 (the one junction rule since ABI v6)");
    c->route = nullptr; c->route_n = 0;          // (the device copy of the route belongs to the context: sim_route_of)
    c->disruption_min_s = g->disruption_min_s;
    return STMPC_OK;
}
void sim_route_of(stmpc_ctx *c, sim::Cfg *sc) {
    if (c->sim_route_n >= 2) { sc->route = c->sim_route.as<double>(); sc->route_n = c->sim_route_n; }
}
sim::State sim_state(stmpc_ctx *c) {
    return sim::State{c->sim_ego.as<double>(), c->sim_nveh.as<int>(), c->sim_vx.as<double>(), c->sim_vv.as<double>(), c->sim_va.as<double>(), c->sim_vc.as<double>(), c->sim_delay.as<double>(),
                      c->sim_status.as<int>(), c->sim_ticks.as<int>(), c->sim_rng.as<unsigned>(), c->sim_acc.as<double>()};
}
}  // namespace

extern "C" {

int stmpc_sim_init_device(stmpc_ctx *c, const stmpc_sim_cfg *g, int N, void *stream) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N < 1) return fail(STMPC_EINVAL, "N must be positive");
    sim::Cfg sc;
    int rc = make_simcfg(g, &sc);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    const size_t KS = sim::KS;
    if ((rc = c->sim_ego.ensure((size_t)N * 4 * 8))) return rc;
    if ((rc = c->sim_nveh.ensure((size_t)N * 4))) return rc;
    if ((rc = c->sim_vx.ensure((size_t)N * KS * 8))) return rc;
    if ((rc = c->sim_vv.ensure((size_t)N * KS * 8))) return rc;
    if ((rc = c->sim_va.ensure((size_t)N * KS * 8))) return rc;
    if ((rc = c->sim_vc.ensure((size_t)N * KS * 8))) return rc;
    if ((rc = c->sim_delay.ensure((size_t)N * 8))) return rc;
    if ((rc = c->sim_status.ensure((size_t)N * 4))) return rc;
    if ((rc = c->sim_ticks.ensure((size_t)N * 4))) return rc;
    if ((rc = c->sim_rng.ensure((size_t)N * 4))) return rc;
    if ((rc = c->sim_acc.ensure((size_t)N * sim::NACC * 8))) return rc;
    c->sim_N = N;
    c->sim_route_n = 0;
    if (g->ego_route_xy && g->ego_route_n >= 2) {
        const int n = g->ego_route_n;
        if (n > 4096) return fail(STMPC_EINVAL, "ego_route_n out of range (at most 4096 points)");
        std::vector<double> xy((size_t)2 * n);
        for (int i = 0; i < n; ++i) {
            xy[i] = g->ego_route_xy[2 * i]; xy[n + i] = g->ego_route_xy[2 * i + 1];
            if (i && !(xy[i] > xy[i - 1])) return fail(STMPC_EINVAL, "ego_route_xy: x must be strictly increasing");
        }
        if ((rc = c->sim_route.ensure(xy.size() * 8))) return rc;
        HIPCHK(hipMemcpyAsync(c->sim_route.p, xy.data(), xy.size() * 8, hipMemcpyHostToDevice, (hipStream_t)stream));
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));       // (xy is a local)
        c->sim_route_n = n;
    }
    sim_route_of(c, &sc);
    hipLaunchKernelGGL(sim::k_sim_init, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, sc, N, sim_state(c));
    HIPCHK(hipGetLastError());
    return STMPC_OK;
}

int stmpc_sim_view_device(stmpc_ctx *c, const stmpc_sim_cfg *g, int N, int Kmax, double *d_ego5, int32_t *d_k, double *d_ox, double *d_ov, double *d_oa, void *stream) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N != c->sim_N || Kmax < 1 || Kmax > STMPC_KMAX_LIMIT) return fail(STMPC_EINVAL, "N does not match stmpc_sim_init_device, or Kmax out of range");
    if (!d_ego5 || !d_k || !d_ox || !d_ov) return fail(STMPC_EINVAL, "NULL device pointer");
    sim::Cfg sc;
    int rc = make_simcfg(g, &sc);
    if (rc) return rc;
    sim_route_of(c, &sc);
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(sim::k_sim_view, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, sc, N, Kmax, sim_state(c), d_ego5, d_k, d_ox, d_ov, d_oa);
    HIPCHK(hipGetLastError());
    return STMPC_OK;
}

int stmpc_sim_step_device(stmpc_ctx *c, const stmpc_params *p, const stmpc_sim_cfg *g, int N, const double *d_cmd_speed, void *stream) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N != c->sim_N || !d_cmd_speed) return fail(STMPC_EINVAL, "N does not match stmpc_sim_init_device, or NULL device pointer");
    sim::Cfg sc;
    DevP dp;
    int rc = make_simcfg(g, &sc);
    if (rc) return rc;
    sim_route_of(c, &sc);
    if ((rc = make_devp(p, &dp))) return rc;
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(sim::k_sim_step, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, dp, sc, N, sim_state(c), d_cmd_speed, p->crash_min_s);
    HIPCHK(hipGetLastError());
    return STMPC_OK;
}

int stmpc_sim_status_device(stmpc_ctx *c, int N, int32_t *d_status, void *stream) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N != c->sim_N || !d_status) return fail(STMPC_EINVAL, "N does not match stmpc_sim_init_device, or NULL pointer");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(d_status, c->sim_status.p, (size_t)N * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return STMPC_OK;
}

int stmpc_sim_read(stmpc_ctx *c, int N, int32_t *status, int32_t *ticks, double *acc, double *ego4) {
    if (!c) return fail(STMPC_EINVAL, "ctx is NULL");
    if (N != c->sim_N) return fail(STMPC_EINVAL, "N does not match stmpc_sim_init_device");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    if (status) HIPCHK(hipMemcpy(status, c->sim_status.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (ticks) HIPCHK(hipMemcpy(ticks, c->sim_ticks.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    if (acc) HIPCHK(hipMemcpy(acc, c->sim_acc.p, (size_t)N * sim::NACC * 8, hipMemcpyDeviceToHost));
    if (ego4) HIPCHK(hipMemcpy(ego4, c->sim_ego.p, (size_t)N * 32, hipMemcpyDeviceToHost));
    return STMPC_OK;
}

}  // extern "C"
