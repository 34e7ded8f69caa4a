// stmpc_cc_kernels.hpp -- device side of the combined RL + ST controller (reference dqn.RLAgent.do_combined_control,
// dqn.py:117-200), batched: one thread per episode.  The policy network stays the caller's (evaluated on the device
// between rollout steps); these kernels do what the reference does around it -- the rollout of the proposed jerks
// through the traffic predictor, the bookkeeping of the rollout history and probe state, and the final decision from the
// feasibility probe and the controller solve of the start state.
#pragma once
#include "stmpc_kernels.hpp"

namespace stmpc {

struct CCfg {
    double tick, comb_min_dist, stop_x, a_max, a_min, v_max, desired_speed;
    int rollout_length, st_test_rollouts, check_rollout_crash, limit_speed, test_rollout_state, strictly_better, remember_last;
};

enum { CC_RL = 0, CC_CRASH = 1, CC_SPEED = 2, CC_ROLLOUT = 3, CC_ST_BETTER = 4 };

// control.get_ego_speed_from_jerk, control.py:160-171
__device__ __forceinline__ double dev_speed_from_jerk(const CCfg &c, double v, double a, double jerk) {
    double na = a + jerk * c.tick;
    if (na > c.a_max) na = c.a_max;
    if (na < c.a_min) na = c.a_min;
    double nv = v + na * c.tick;
    if (nv > c.v_max) nv = c.v_max;
    if (nv < 0) nv = 0;
    return nv;
}

// st.get_path_mean_abs_jerk, st.py:274-288
__device__ __forceinline__ double dev_mean_abs_jerk(const double *s, int n, double v0, double a0, double dt) {
    double prev_a = a0, prev_v = v0, acc = 0.0;
    for (int i = 1; i < n; ++i) {
        const double v = (s[i] - s[i - 1]) / dt;
        const double a = (v - prev_v) / dt;
        const double j = (a - prev_a) / dt;
        prev_v = v; prev_a = a;
        acc += fabs(j);
    }
    return acc / (double)(n - 1);
}

struct CCState {              // per-context bookkeeping of one rollout (device arrays, [N] unless noted)
    int *live, *hist_len, *crash_pred, *have_test;
    double *sel_speed;
    double *rollout_s;        // [N][R + 1]
    double *test_ego4, *test_ox, *test_ov;      // [N][4], [N][Kmax], [N][Kmax]
};

// One rollout step, dqn.py:129-141.  step is 1-based; step 1 also resets the bookkeeping.
template <int KMAX>
__global__ void __launch_bounds__(64) k_rollout_step(DevP p, CCfg c, int N, int Kmax, int step, const double *__restrict__ ego5_start,
                                                     double *ego4, const int *__restrict__ k_count, double *ox, double *ov, double *oa,
                                                     const double *__restrict__ action, CCState st) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int R1 = c.rollout_length + 1;
    if (step == 1) {
        st.live[e] = 1; st.hist_len[e] = 1; st.crash_pred[e] = 0; st.have_test[e] = 0; st.sel_speed[e] = 0.0;
        st.rollout_s[(size_t)e * R1] = ego5_start[(size_t)e * 5 + 4];          // control.get_ego_s(start_state.ego_position), dqn.py:121
    }
    if (!st.live[e]) return;
    DState<KMAX> s;
    s.ex = ego4[e * 4 + 0]; s.ey = ego4[e * 4 + 1]; s.ev = ego4[e * 4 + 2]; s.ea = ego4[e * 4 + 3];
    int k = k_count[e];
    k = k < 0 ? 0 : (k > KMAX ? KMAX : k);
    k = k > Kmax ? Kmax : k;
    s.k = k;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        const bool in = (i < k) && (i < Kmax);
        s.xs[i] = in ? ox[(size_t)e * Kmax + i] : 0.0;
        s.vs[i] = in ? ov[(size_t)e * Kmax + i] : 0.0;
    }
    const double sel = dev_speed_from_jerk(c, s.ev, s.ea, action[e]);          // dqn.py:133-135
    double acc[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) acc[i] = 0.0;
    const bool cr = dev_predict_with_ego<KMAX>(p, s, sel, c.tick, c.comb_min_dist, acc);      // dqn.py:136
    ego4[e * 4 + 0] = s.ex; ego4[e * 4 + 1] = s.ey; ego4[e * 4 + 2] = s.ev; ego4[e * 4 + 3] = s.ea;
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
        if (i < k && i < Kmax) { ox[(size_t)e * Kmax + i] = s.xs[i]; ov[(size_t)e * Kmax + i] = s.vs[i]; if (oa) oa[(size_t)e * Kmax + i] = acc[i]; }
    st.sel_speed[e] = sel;
    st.crash_pred[e] = cr ? 1 : 0;
    if (step == c.st_test_rollouts) {                                           // dqn.py:137-138
        st.test_ego4[e * 4 + 0] = s.ex; st.test_ego4[e * 4 + 1] = s.ey; st.test_ego4[e * 4 + 2] = s.ev; st.test_ego4[e * 4 + 3] = s.ea;
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i < k && i < Kmax) { st.test_ox[(size_t)e * Kmax + i] = s.xs[i]; st.test_ov[(size_t)e * Kmax + i] = s.vs[i]; }
        st.have_test[e] = 1;
    }
    const int hl = st.hist_len[e];
    if (hl < R1) { st.rollout_s[(size_t)e * R1 + hl] = dev_ego_s(s.ex, s.ey); st.hist_len[e] = hl + 1; }     // dqn.py:139
    st.live[e] = (!cr && !(s.ex > c.stop_x)) ? 1 : 0;                          // loop condition + dqn.py:140-141
}

// Probe state of the episodes whose rollout ended before step ST_TEST_ROLLOUTS (dqn.py:142-143), as the 5-column state
// the solver takes (start_s of the probe state from the device map of control.get_ego_s).
__global__ void __launch_bounds__(64) k_cc_probe_state(int N, int Kmax /* row stride */, int Kcopy /* vehicles per row the caller's arrays hold (0: none, pointers may be null) */,
                                                       const int *__restrict__ k_count, const double *__restrict__ cur_ego4, const double *__restrict__ cur_ox,
                                                       const double *__restrict__ cur_ov, CCState st, double *probe_ego5, double *probe_ox, double *probe_ov) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const bool ht = st.have_test[e] != 0;
    const double *src4 = ht ? st.test_ego4 + (size_t)e * 4 : cur_ego4 + (size_t)e * 4;
    for (int q = 0; q < 4; ++q) probe_ego5[(size_t)e * 5 + q] = src4[q];
    probe_ego5[(size_t)e * 5 + 4] = dev_ego_s(src4[0], src4[1]);
    int k = k_count[e];
    k = k < 0 ? 0 : (k > Kcopy ? Kcopy : k);                                    // only the episode's own vehicles are copied; the rest of the row is zero
    for (int i = 0; i < Kmax; ++i) {
        double x = 0.0, v = 0.0;
        if (i < k) { x = ht ? st.test_ox[(size_t)e * Kmax + i] : cur_ox[(size_t)e * Kmax + i]; v = ht ? st.test_ov[(size_t)e * Kmax + i] : cur_ov[(size_t)e * Kmax + i]; }
        probe_ox[(size_t)e * Kmax + i] = x; probe_ov[(size_t)e * Kmax + i] = v;
    }
}

// The decision, dqn.py:144-200.  probe_crash = st.test_guaranteed_crash_from_state(test_state); st_speed / fine / fine_len =
// st.do_st_control's command and (trimmed, re-sampled) path for the START state.
__global__ void __launch_bounds__(64) k_cc_decide(CCfg c, int N, const double *__restrict__ ego5_start, const double *__restrict__ first_action,
                                                  const int *__restrict__ last_choice_rl, CCState st, const int *__restrict__ probe_crash,
                                                  const double *__restrict__ st_speed, const double *__restrict__ fine, const int *__restrict__ fine_len,
                                                  int fine_stride, int *takeover, int *reason_out, double *speed_out, unsigned *err) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int R1 = c.rollout_length + 1;
    const double v0 = ego5_start[(size_t)e * 5 + 2], a0 = ego5_start[(size_t)e * 5 + 3];
    int reason = CC_RL;
    double speed = dev_speed_from_jerk(c, v0, a0, first_action[e]);            // control.set_ego_jerk(first_action), control.py:174-178
    if (c.check_rollout_crash && st.crash_pred[e]) reason = CC_CRASH;
    else if (c.limit_speed && st.sel_speed[e] > c.desired_speed) reason = CC_SPEED;
    else if (c.test_rollout_state && probe_crash[e]) reason = CC_ROLLOUT;
    else if (c.strictly_better) {
        const int m = fine_len[e];
        if (m < 0) atomicOr(err, 1u);                                           // fine grid longer than the QP kernel supports
        else if (m > 1) {                                                      // dqn.py:167-169: a single point = nothing to compare
            const int hl = st.hist_len[e];
            const int ml = m < hl ? m : hl;
            const double *fs = fine + (size_t)e * fine_stride, *hs = st.rollout_s + (size_t)e * R1;
            const double st_jerk = dev_mean_abs_jerk(fs, ml, v0, a0, c.tick), rl_jerk = dev_mean_abs_jerk(hs, ml, v0, a0, c.tick);
            const double st_dist = fs[ml - 1] - fs[0], rl_dist = hs[ml - 1] - hs[0];
            const bool last_rl = last_choice_rl ? last_choice_rl[e] != 0 : true;
            bool choose;
            if (last_rl || !c.remember_last) choose = (st_jerk < rl_jerk && st_dist > rl_dist) || rl_dist == 0;
            else choose = !(rl_jerk < st_jerk && rl_dist > st_dist);
            if (choose) { reason = CC_ST_BETTER; speed = (fs[1] - fs[0]) / c.tick; }        // dqn.py:180-181,192-193
        }
    }
    if (reason == CC_CRASH || reason == CC_SPEED || reason == CC_ROLLOUT) {
        speed = st_speed[e];                                                    // st.do_st_control(start_state)
        if (fine_len[e] < 0) atomicOr(err, 1u);                                 // ... which could not re-sample its path: the command is not the reference's
    }
    takeover[e] = reason != CC_RL;
    reason_out[e] = reason;
    speed_out[e] = speed;
}

// The policy's input vector for N states: dqn.get_state_vector_from_base_state (dqn.py:389-446), then what the reference's RL library does
// between that function and the network in DDPGAgent.get_control (ddpg.py:83-87) -- GymEnvironment._make_state's cast to the observation
// space's float32, and TimeFeature (ddpg.py:41), which appends 0.001 x (policy evaluations since the episode began) and counts one up.
// The library (`all` 0.5.3) is absent from the reference checkout: the cast and the time feature are restated from its published source
// and are parity-unpinned; the 20 entries before them are the reference's own function and are pinned by golden_combined_real.npz.
// One thread per state.  Vehicles are taken in LIST order as the reference takes them (front list reversed, dqn.py:417), not sorted.
struct FeatCfg {
    double max_speed, sensor_radius;      // Settings.MAX_SPEED, Settings.SENSOR_RADIUS
    float time_scale;                     // TimeFeature.scale (0.001)
    int cars_ahead, cars_behind, use_accel, use_speed_diff, normalize, time_feature;
};
__device__ __forceinline__ void dev_policy_features(const FeatCfg &f, int e, int Kmax, const double *__restrict__ ego4, const int *__restrict__ k_count,
                                                    const double *__restrict__ ox, const double *__restrict__ ov, const double *__restrict__ oa,
                                                    const int *__restrict__ live /* null: every state is evaluated */, int *evals, float *row) {
    const double ex = ego4[(size_t)e * 4 + 0], ey = ego4[(size_t)e * 4 + 1], ev = ego4[(size_t)e * 4 + 2], ea = ego4[(size_t)e * 4 + 3];
    int k = k_count[e];
    k = k < 0 ? 0 : (k > Kmax ? Kmax : k);
    const int tw = f.use_accel ? 4 : 3;
    const int nveh = (f.cars_ahead + f.cars_behind) * tw;
    for (int q = 0; q < nveh; ++q) row[q] = 0.0f;                              // buffer tuples (dqn.py:418-425)
    const double *xs = ox + (size_t)e * Kmax, *vs = ov + (size_t)e * Kmax, *as = oa ? oa + (size_t)e * Kmax : nullptr;
    auto put = [&](int slot, int i) {
        float *t = row + slot * tw;
        int o = 0;
        if (f.use_accel) { const double a = as ? as[i] : 0.0; t[o++] = (float)(f.normalize ? a / 9.0 : a); }
        const double dv = f.use_speed_diff ? vs[i] - ev : vs[i];
        t[o++] = (float)(f.normalize ? dv / f.max_speed : dv);
        const double dx = xs[i] - ex;
        t[o++] = (float)(f.normalize ? dx / f.sensor_radius : dx);
        t[o] = 1.0f;
    };
    int nf = 0, nb = 0;
    for (int i = k - 1; i >= 0 && nf < f.cars_ahead; --i) if (xs[i] > ex) put(nf++, i);                // front_cars reversed, first CARS_AHEAD
    for (int i = 0; i < k && nb < f.cars_behind; ++i) if (!(xs[i] > ex)) put(f.cars_ahead + nb++, i);  // back_cars, first CARS_BEHIND
    float *t = row + nveh;
    t[0] = (float)(f.normalize ? ev / f.max_speed : ev);
    t[1] = (float)(f.normalize ? ea / 9.0 : ea);
    t[2] = (float)(f.normalize ? ex / 300.0 : ex);
    t[3] = (float)(f.normalize ? ey / 100.0 : ey);
    if (f.time_feature) {
        const int n = evals[e];
        t[4] = f.time_scale * (float)n;                                       // scale * timestep, single precision as torch evaluates it
        if (!live || live[e]) evals[e] = n + 1;                               // only the states whose rollout goes on ask the policy again (dqn.py:129-133)
    }
}
__global__ void __launch_bounds__(64) k_policy_features(FeatCfg f, int N, int Kmax, const double *__restrict__ ego4, const int *__restrict__ k_count,
                                                        const double *__restrict__ ox, const double *__restrict__ ov, const double *__restrict__ oa,
                                                        const int *__restrict__ live, int *evals, float *feat, int stride) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    dev_policy_features(f, e, Kmax, ego4, k_count, ox, ov, oa, live, evals, feat + (size_t)e * stride);
}

// Ordered list of the states whose decision needs st.do_st_control(start_state) (dqn.py:144-155; every branch but the
// strictly-better comparison, which needs the controller's path for every state).  One workgroup, block-wide ordered compaction.
__global__ void __launch_bounds__(1024) k_cc_select(CCfg c, int N, CCState st, const int *__restrict__ probe_crash, int *sel_idx, int *sel_count) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < N; start += 1024) {
        const int e = start + tid;
        bool need = false;
        if (e < N) need = (c.check_rollout_crash && st.crash_pred[e]) || (c.limit_speed && st.sel_speed[e] > c.desired_speed) || (c.test_rollout_state && probe_crash[e]);
        const unsigned long long b = __ballot(need);
        if (lane == 0) wsum[w] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int i = 0; i < w; ++i) off += wsum[i];
        if (need) sel_idx[off + __popcll(b & ((1ull << lane) - 1ull))] = e;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int i = 0; i < 16; ++i) t += wsum[i]; base += t; }
        __syncthreads();
    }
    if (tid == 0) *sel_count = base;
}
__global__ void __launch_bounds__(64) k_cc_gather(int M, int Kmax, int Kcopy, const int *__restrict__ sel_idx, const double *__restrict__ ego5, const int *__restrict__ k_count,
                                                  const double *__restrict__ ox, const double *__restrict__ ov, double *c_ego5, int *c_k, double *c_ox, double *c_ov) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const int e = sel_idx[j];
    for (int q = 0; q < 5; ++q) c_ego5[(size_t)j * 5 + q] = ego5[(size_t)e * 5 + q];
    c_k[j] = k_count[e];
    for (int i = 0; i < Kmax; ++i) {
        c_ox[(size_t)j * Kmax + i] = i < Kcopy ? ox[(size_t)e * Kcopy + i] : 0.0;
        c_ov[(size_t)j * Kmax + i] = i < Kcopy ? ov[(size_t)e * Kcopy + i] : 0.0;
    }
}
__global__ void __launch_bounds__(64) k_cc_scatter(int M, const int *__restrict__ sel_idx, const double *__restrict__ c_speed, const double *__restrict__ c_fine,
                                                   const int *__restrict__ c_fine_len, int fine_stride, double *speed, double *fine, int *fine_len) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const int e = sel_idx[j];
    speed[e] = c_speed[j];
    fine_len[e] = c_fine_len[j];
    for (int q = 0; q < fine_stride; ++q) fine[(size_t)e * fine_stride + q] = c_fine[(size_t)j * fine_stride + q];
}

// ------------------------------------------------------------------------------------------------------------------
// SUMO-free batched merge episodes (stands in for control.run_episode / control.step,
// control.py:215-340).  The traffic is the reference's "simple traffic distribution" (config.py:39, sumo.py:43-58): vehicles of
// SUMO's vType "normal" of merge_impossible.rou.xml:3 -- car-following model Krauss, accel 4.5, decel 6.0, minGap 1, tau 0.5,
// sigma 0, speedFactor 1 / speedDev 0, maxSpeed = OTHER_CAR_SPEED -- added at the start of the highway every
// BASE_TRAFFIC_INTERVAL (+ U(0,1)) seconds (control.py:215-226), and the ego under speed mode 22 (control.py:43: it obeys its
// acceleration limits and nothing else) moving along the centre line of its lanes (Cfg::route: ramp_0 and the junction's internal lane of
// merge.net.xml -- the positions TraCI reports and the reference's planner and actors see).  The Krauss follow speed is SUMO's Euler form
// (MSCFModel::maximumSafeStopSpeedEuler / maximumSafeFollowSpeed), a highway vehicle regards as its leader the vehicle ahead and, once the
// ego is on the junction's internal lane or beyond, the ego if that is ahead of it; while the ego laps in, a vehicle whose front is behind the
// ego's front is asked to stop (the one junction rule since round 6; rounds 3-5 compared five); bodies collide
// when they overlap along the lane while the converging lanes are less than a vehicle width apart.  This is a restatement of
// SUMO's documented models, not SUMO: episode statistics compare with the reference's reports as DISTRIBUTIONS -- and are labelled so.
// One thread per environment; vehicles live in [N][KS] arrays, front to back.
namespace sim {
constexpr int KS = 64;             // vehicle slots per environment
struct Cfg {
    double tick, other_speed, base_interval, spawn_x, despawn_x, ego_start_x, ego_start_y, arrive_x, sensor_radius;
    double start_speed, start_speed_std, min_start_speed, max_start_speed;
    double veh_accel, veh_decel, veh_min_gap, veh_tau, veh_emergency_decel, veh_length, veh_width, speed_dev;
    int vary_interval, randomize_start_speed, max_ticks;
    unsigned long long seed;
    const double *route;           // device [2][route_n]: x then y of the ego's lane centre line (nullptr: straight lines)
    int route_n;
    double disruption_min_s;
};
constexpr int NACC = 12;           // statistics per environment (STMPC_SIM_NACC)
struct State {                     // device arrays
    double *ego4;                  // [N][4] x, y, v, a
    int *nveh;                     // [N]
    double *vx, *vv, *va, *vc;     // [N][KS] position (front bumper, as traci reports it), speed, acceleration, desired speed of each vehicle
    double *delay;                 // [N] time to the next highway vehicle
    int *status;                   // [N] 0 running, 1 arrived ("merged"), 2 crashed, 3 out of time
    int *ticks;                    // [N] controlled ticks so far
    unsigned *rng;                 // [N] draws so far
    double *acc;                   // [N][NACC]: sum speed, max speed, sum |jerk|, previous acceleration, samples, min gap (s > CRASH_MIN_S), sum gap, gap samples,
                                   //            disruption (deceleration of the nearest vehicle behind, s > disruption_min_s): sum, max, samples, non-zero samples
};
__device__ __forceinline__ double uniform01(unsigned long long seed, int env, unsigned &ctr) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)env * 0x100000001ull + (unsigned long long)(ctr++) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;      // splitmix64
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
// desired speed of a vehicle: OTHER_CAR_SPEED * speedFactor, speedFactor ~ normc(1, speed_dev) clipped to [0.2, 2] (0 = exactly 1, the simple distribution)
__device__ __forceinline__ double cruise_speed(const Cfg &c, int env, unsigned &ctr) {
    if (!(c.speed_dev > 0.0)) return c.other_speed;
    const double u1 = uniform01(c.seed, env, ctr), u2 = uniform01(c.seed, env, ctr);
    double f = 1.0 + c.speed_dev * sqrt(-2.0 * log(u1 > 1e-300 ? u1 : 1e-300)) * cos(6.283185307179586 * u2);
    f = f < 0.2 ? 0.2 : (f > 2.0 ? 2.0 : f);
    return c.other_speed * f;
}
// distance covered while braking from `speed` with `decel` in whole steps (MSCFModel::brakeGapEuler, headway 0)
__device__ __forceinline__ double brake_gap(double speed, double decel, double ts) {
    const double red = decel * ts;
    const int steps = (int)(speed / red);
    return ts * ((double)steps * speed - red * (double)steps * (double)(steps + 1) / 2.0);
}
// largest speed from which the vehicle can still stop within `gap` when it brakes with `decel` after its reaction time (MSCFModel::maximumSafeStopSpeedEuler)
__device__ __forceinline__ double safe_stop_speed(double gap, double decel, double tau, double ts) {
    const double g = gap - 0.001;
    if (g < 0.0) return 0.0;
    const double b = decel * ts, t = tau, s_ = ts;
    const double n = floor(0.5 - ((t + (sqrt((s_ * s_) + (4.0 * ((s_ * (2.0 * g / b - t)) + (t * t)))) * -0.5)) / s_));
    const double h = 0.5 * n * (n - 1.0) * b * s_ + n * b * t;
    const double r = (g - h) / (n * s_ + t);
    const double x = n * b + r;
    return x > 0.0 ? x : 0.0;
}
// Krauss: the speed that stays safe behind a leader `gap` ahead (net of minGap) that may brake with the same deceleration (MSCFModel::maximumSafeFollowSpeed)
__device__ __forceinline__ double krauss_follow(const Cfg &c, double gap, double lead_speed) {
    return safe_stop_speed(gap + brake_gap(lead_speed, c.veh_decel, c.tick), c.veh_decel, c.veh_tau, c.tick);
}
// position of the ego along the highway lane once it is on the junction's internal lane: both internal lanes start at x = -50.6, are 52.2 m long and
// end at x = 1.5, and the ego's lane differs in x extent from its length by 0.2 %: the x coordinate itself, which is also what the planner
// compares with the vehicles' (prediction.py:78)
__device__ __forceinline__ double ego_lane_pos(double x, double /*y*/) { return x; }
__global__ void __launch_bounds__(64) k_sim_init(Cfg c, int N, State s) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    unsigned ctr = 0;
    // highway traffic in its stationary state (what the reference gets from its warm-up ticks, control.py:255-258): every vehicle at its desired
    // speed, spaced by the insertion intervals; an insertion that is not yet safe is postponed, as SUMO does
    int n = 0;
    const double head0 = c.base_interval + (c.vary_interval ? uniform01(c.seed, e, ctr) : 0.0);
    double x = c.despawn_x - c.other_speed * uniform01(c.seed, e, ctr) * head0;
    const double min_space = c.veh_length + c.veh_min_gap + c.other_speed * c.veh_tau;
    while (x > c.spawn_x && n < KS) {
        const double vc_ = cruise_speed(c, e, ctr);
        s.vx[(size_t)e * KS + n] = x; s.vv[(size_t)e * KS + n] = vc_ < c.other_speed ? vc_ : c.other_speed; s.va[(size_t)e * KS + n] = 0.0; s.vc[(size_t)e * KS + n] = vc_; ++n;
        const double step = c.other_speed * (c.base_interval + (c.vary_interval ? uniform01(c.seed, e, ctr) : 0.0));
        x -= step > min_space ? step : min_space;
    }
    s.nveh[e] = n;
    s.delay[e] = (x - c.spawn_x) / (c.other_speed > 0 ? -c.other_speed : -1.0);      // (x < spawn_x here: the next vehicle enters after that time)
    double v0 = c.start_speed;                                                        // control.get_ego_start_speed, control.py:198-204
    if (c.randomize_start_speed) {
        const double u1 = uniform01(c.seed, e, ctr), u2 = uniform01(c.seed, e, ctr);
        v0 = c.start_speed + c.start_speed_std * sqrt(-2.0 * log(u1 > 1e-300 ? u1 : 1e-300)) * cos(6.283185307179586 * u2);
        v0 = v0 < c.min_start_speed ? c.min_start_speed : (v0 > c.max_start_speed ? c.max_start_speed : v0);
    }
    s.ego4[e * 4 + 0] = c.ego_start_x; s.ego4[e * 4 + 1] = c.ego_start_y; s.ego4[e * 4 + 2] = v0; s.ego4[e * 4 + 3] = 0.0;
    s.status[e] = 0; s.ticks[e] = 0; s.rng[e] = ctr;
    for (int q = 0; q < NACC; ++q) s.acc[(size_t)e * NACC + q] = 0.0;
    s.acc[(size_t)e * NACC + 5] = 1e300;
}
// The planner's view of each environment (HighwayState.from_sumo, prediction.py:112-142): the vehicles within the sensor radius of the ego (plane
// distance; the highway lane runs at y = -1.6), front to back, and the ego with its s coordinate.
__global__ void __launch_bounds__(64) k_sim_view(Cfg c, int N, int Kmax, State s, double *ego5, int *k_count, double *ox, double *ov, double *oa) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const double ex = s.ego4[e * 4 + 0], ey = s.ego4[e * 4 + 1];
    for (int q = 0; q < 4; ++q) ego5[(size_t)e * 5 + q] = s.ego4[e * 4 + q];
    ego5[(size_t)e * 5 + 4] = dev_ego_s(ex, ey);
    int k = 0;
    const int n = s.nveh[e];
    for (int i = 0; i < n && k < Kmax; ++i) {
        const double x = s.vx[(size_t)e * KS + i];
        const double dx = x - ex, dy = -1.6 - ey;
        if (sqrt(dx * dx + dy * dy) < c.sensor_radius) { ox[(size_t)e * Kmax + k] = x; ov[(size_t)e * Kmax + k] = s.vv[(size_t)e * KS + i]; if (oa) oa[(size_t)e * Kmax + k] = s.va[(size_t)e * KS + i]; ++k; }
    }
    for (int i = k; i < Kmax; ++i) { ox[(size_t)e * Kmax + i] = 0.0; ov[(size_t)e * Kmax + i] = 0.0; if (oa) oa[(size_t)e * Kmax + i] = 0.0; }
    k_count[e] = k;
}
// One simulator tick with the commanded ego speed.
__global__ void __launch_bounds__(64) k_sim_step(DevP p, Cfg c, int N, State s, const double *__restrict__ cmd_speed, double crash_min_s) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N || s.status[e] != 0) return;
    const double dt = c.tick;
    const double cx = s.ego4[e * 4 + 0], cy = s.ego4[e * 4 + 1];
    const double v_prev = s.ego4[e * 4 + 2];
    double sel = cmd_speed[e];
    if (!(sel == sel)) sel = v_prev;                                              // (a NaN command keeps the speed)
    // speed mode 22: the ego obeys its vType's acceleration limits (accel 4.5 / decel 6.0, merge_impossible.rou.xml:2) and its lane's speed limit only
    const double hi = v_prev + p.a_max * dt, lo = v_prev + p.a_min * dt;
    sel = sel > hi ? hi : (sel < lo ? lo : sel);
    sel = sel < 0 ? 0 : (sel > p.v_max ? p.v_max : sel);
    // ego motion along its route: the lane centre line of the reference's network, or the straight lines the planner assumes (prediction.py:46-59)
    double px, py;
    if (c.route && cx < c.route[c.route_n - 1]) {
        const double *RX = c.route, *RY = c.route + c.route_n;
        int i = 0;
        while (i + 2 < c.route_n && RX[i + 1] <= cx) ++i;                        // segment [i, i + 1] holds the ego (x strictly increasing)
        double left = sel * dt;
        px = cx; py = cy;
        while (left > 0.0 && i + 1 < c.route_n) {
            const double ex = RX[i + 1] - px, ey = RY[i + 1] - py;
            const double seg = sqrt(__builtin_fma(ey, ey, ex * ex));
            if (left < seg) { px += ex / seg * left; py += ey / seg * left; left = 0.0; break; }
            left -= seg; px = RX[i + 1]; py = RY[i + 1]; ++i;
        }
        px += left;                                                               // past the junction exit: along the highway lane
    } else if (c.route) {
        py = cy; px = cx + sel * dt;
    } else if (cx < 1.5) {
        double d0 = 1.5 - cx, d1 = -1.5 - cy;
        const double nrm = sqrt(__builtin_fma(d1, d1, d0 * d0));
        d0 /= nrm; d1 /= nrm;
        px = cx + d0 * (sel * dt); py = cy + d1 * (sel * dt);
        if (py < -1.6) py = -1.6;
    } else { py = cy; px = cx + sel * dt; }
    const double acc_ego = (sel - v_prev) / dt;
    const double es = dev_ego_s(px, py);
    // Highway vehicles (Krauss, front to back).  Every vehicle plans on the positions at the START of the step, as SUMO's planMovements does.
    const bool ego_on_lane = cx >= -50.58;                                       // on the junction's internal lane or beyond: a leader for those behind it
    const double ego_pos0 = ego_lane_pos(cx, cy);
    const int n = s.nveh[e];
    double lead_x = __builtin_inf(), lead_v = 0.0;                               // the vehicle ahead at the start of the step
    bool crashed = false;
    double gap_ahead = 100.0, gap_behind = 100.0;                                // distances to the nearest vehicles in front of / behind the ego (control.py:289-303)
    double behind_acc = 0.0; bool have_behind = false;                           // acceleration of the nearest vehicle behind the ego (get_closest_cars, prediction.py:162-182)
    const double ego_pos1 = ego_lane_pos(px, py);
    // lateral distance of the converging lanes at the ego's position: bodies can touch only where it is below a vehicle width
    const double lat = px < 1.5 ? 3.31 * (1.5 - px) / 52.08 : 0.0;
    for (int i = 0; i < n; ++i) {
        const double ox_ = s.vx[(size_t)e * KS + i], ov_ = s.vv[(size_t)e * KS + i];
        double vnext = ov_ + c.veh_accel * dt;                                    // maxNextSpeed
        const double vdes = s.vc[(size_t)e * KS + i];
        vnext = vnext < vdes ? vnext : vdes;
        if (lead_x < __builtin_inf()) {
            const double vs = krauss_follow(c, lead_x - c.veh_length - ox_ - c.veh_min_gap, lead_v);
            vnext = vs < vnext ? vs : vnext;
        }
        // the ego is a leader too (it may be closer than the vehicle ahead) once it is on the junction
        if (ego_on_lane) {
            // SUMO's link leader on merging internal lanes, restated from its observable effect (the reference's "disruption" columns: the vehicle
            // behind an ego that cuts in brakes at its EMERGENCY deceleration for ~3 ticks however fast the ego is; MSVehicle::getSafeFollowSpeed's
            // branch for a negative gap asks for a stop, not for a follow speed): with room behind the ego's rear (net of minGap) the vehicle
            // follows it like any leader; while the ego laps in, a vehicle whose front is behind the ego's front is asked to stop, i.e. brakes
            // as hard as it can until the ego's rear is clear.  (The one rule kept: "rule 2" of round 5's comparison, frozen in round 6.)
            const double g_net = ego_pos0 - c.veh_length - ox_ - c.veh_min_gap;
            double vs = __builtin_inf();
            if (g_net >= 0.0) vs = krauss_follow(c, g_net, v_prev);
            else if (ego_pos0 > ox_) vs = 0.0;
            vnext = vs < vnext ? vs : vnext;
        }
        const double vmin = ov_ - c.veh_emergency_decel * dt;                     // (a vehicle never brakes harder than its emergency deceleration)
        vnext = vnext < vmin ? vmin : vnext;
        vnext = vnext < 0.0 ? 0.0 : vnext;
        const double nx = ox_ + vnext * dt;                                       // Euler update: the new speed for the whole step
        lead_x = ox_; lead_v = ov_;
        s.vx[(size_t)e * KS + i] = nx; s.vv[(size_t)e * KS + i] = vnext; s.va[(size_t)e * KS + i] = (vnext - ov_) / dt;
        // collision: the bodies [x - length, x] overlap along the lane while the lanes are less than a vehicle width apart
        if (px >= -50.58 && lat < c.veh_width && fabs(nx - ego_pos1) < c.veh_length) crashed = true;
        const double d = fabs(nx - px);                                           // (the reference's closest-vehicle metric uses raw x differences)
        if (nx >= px) gap_ahead = d < gap_ahead ? d : gap_ahead; else gap_behind = d < gap_behind ? d : gap_behind;
        if (!have_behind && nx < px) {                                            // (front to back: the first one behind; seen only within the sensor radius)
            have_behind = true;
            const double ddx = nx - px, ddy = -1.6 - py;
            behind_acc = sqrt(ddx * ddx + ddy * ddy) < c.sensor_radius ? (vnext - ov_) / dt : 0.0;
        }
    }
    const double gap = gap_ahead < gap_behind ? gap_ahead : gap_behind;
    // vehicles leaving at the end of the highway (front of the list) and entering at its start, control.py:215-226
    int nn = n, drop = 0;
    while (drop < nn && s.vx[(size_t)e * KS + drop] > c.despawn_x) ++drop;
    if (drop) { for (int i = drop; i < nn; ++i) { s.vx[(size_t)e * KS + i - drop] = s.vx[(size_t)e * KS + i]; s.vv[(size_t)e * KS + i - drop] = s.vv[(size_t)e * KS + i]; s.va[(size_t)e * KS + i - drop] = s.va[(size_t)e * KS + i]; s.vc[(size_t)e * KS + i - drop] = s.vc[(size_t)e * KS + i]; } nn -= drop; }
    double delay = s.delay[e];
    unsigned ctr = s.rng[e];
    if (delay <= 0) {
        // SUMO inserts with departSpeed = OTHER_CAR_SPEED when that speed is safe behind the last vehicle, else postpones the insertion
        bool room = nn < KS;
        if (room && nn > 0) room = krauss_follow(c, s.vx[(size_t)e * KS + nn - 1] - c.veh_length - c.spawn_x - c.veh_min_gap, s.vv[(size_t)e * KS + nn - 1]) >= c.other_speed;
        if (room) {
            const double vc_ = cruise_speed(c, e, ctr);
            s.vx[(size_t)e * KS + nn] = c.spawn_x; s.vv[(size_t)e * KS + nn] = vc_ < c.other_speed ? vc_ : c.other_speed; s.va[(size_t)e * KS + nn] = 0.0; s.vc[(size_t)e * KS + nn] = vc_; ++nn;
            delay = (c.vary_interval ? uniform01(c.seed, e, ctr) : 0.0) + c.base_interval;
        }
    }
    delay -= dt;
    s.delay[e] = delay; s.rng[e] = ctr; s.nveh[e] = nn;
    s.ego4[e * 4 + 0] = px; s.ego4[e * 4 + 1] = py; s.ego4[e * 4 + 2] = sel; s.ego4[e * 4 + 3] = acc_ego;
    // episode statistics, control.py:276-305 / stats.py:43-75
    double *a = s.acc + (size_t)e * NACC;
    const int tk = s.ticks[e];
    a[0] += sel; a[1] = sel > a[1] ? sel : a[1];
    if (tk > 0) a[2] += fabs((acc_ego - a[3]) / dt);
    a[3] = acc_ego; a[4] += 1.0;
    if (es > crash_min_s) { a[5] = gap < a[5] ? gap : a[5]; a[6] += gap; a[7] += 1.0; }
    if (es > c.disruption_min_s) {
        const double dis = behind_acc < 0.0 ? -behind_acc : 0.0;
        a[8] += dis; a[9] = dis > a[9] ? dis : a[9]; a[10] += 1.0; if (dis != 0.0) a[11] += 1.0;
    }
    s.ticks[e] = tk + 1;
    // (a vehicle that reaches its arrival position leaves SUMO's network inside the move, before the step's collision check sees it)
    if (px >= c.arrive_x) s.status[e] = 1;
    else if (crashed) s.status[e] = 2;
    else if (tk + 1 >= c.max_ticks) s.status[e] = 3;
}
}  // namespace sim
}  // namespace stmpc
