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

// ------------------------------------------------------------------------------------------------------------------
// SUMO-free batched merge episodes (stands in for control.run_episode / control.step,
// control.py:215-340).  The world IS the planner's own model: the ego moves as prediction.py:46-59 says, the highway
// vehicles follow prediction.py:75-97 (constant speed, braking behind a slower leader within 30 m, reacting to the ego once
// it has merged), vehicles enter the highway as control.step adds them (control.py:215-226) and leave at its end.  Nothing
// here reproduces SUMO's Krauss model, so episode statistics are comparable with the reference's reports only as
// distributions -- and are labelled so.  One thread per environment; vehicles live in [N][KS] arrays, front to back.
namespace sim {
constexpr int KS = 64;             // vehicle slots per environment
struct Cfg {
    double tick, other_speed, base_interval, spawn_x, despawn_x, ego_start_x, ego_start_y, arrive_x, sensor_radius;
    double start_speed, start_speed_std, min_start_speed, max_start_speed;
    int vary_interval, randomize_start_speed, max_ticks;
    unsigned long long seed;
};
struct State {                     // device arrays
    double *ego4;                  // [N][4] x, y, v, a
    int *nveh;                     // [N]
    double *vx, *vv, *va, *vc;     // [N][KS] position, speed, acceleration, cruise speed of each vehicle
    double *delay;                 // [N] time to the next highway vehicle
    int *status;                   // [N] 0 running, 1 arrived ("merged"), 2 crashed, 3 out of time
    int *ticks;                    // [N] controlled ticks so far
    unsigned *rng;                 // [N] draws so far
    double *acc;                   // [N][8]: sum speed, max speed, sum |jerk|, previous acceleration, samples, min gap (s > CRASH_MIN_S), sum gap, gap samples
};
__device__ __forceinline__ double uniform01(unsigned long long seed, int env, unsigned &ctr) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)env * 0x100000001ull + (unsigned long long)(ctr++) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;      // splitmix64
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
// SUMO's vType "normal" (merge.rou.xml:28-29, maxSpeed set to OTHER_CAR_SPEED by sumo.py:58): each vehicle cruises at
// OTHER_CAR_SPEED * speedFactor, speedFactor ~ normc(1.0, 0.1) clipped to [0.2, 2]; a vehicle is inserted only when the one
// ahead leaves the IDM desired gap minGap + v * tau + length = 2.5 + v + 5 free (SUMO postpones unsafe insertions).
__device__ __forceinline__ double cruise_speed(const Cfg &c, int env, unsigned &ctr) {
    const double u1 = uniform01(c.seed, env, ctr), u2 = uniform01(c.seed, env, ctr);
    double f = 1.0 + 0.1 * sqrt(-2.0 * log(u1 > 1e-300 ? u1 : 1e-300)) * cos(6.283185307179586 * u2);
    f = f < 0.2 ? 0.2 : (f > 2.0 ? 2.0 : f);
    return c.other_speed * f;
}
__global__ void __launch_bounds__(64) k_sim_init(Cfg c, int N, State s) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    unsigned ctr = 0;
    // highway traffic in its stationary state (what the reference gets from 20 s of warm-up ticks, control.py:255-256)
    int n = 0;
    double x = c.despawn_x - c.other_speed * uniform01(c.seed, e, ctr) * (c.base_interval + (c.vary_interval ? 1.0 : 0.0));
    const double min_gap = 2.5 + c.other_speed + 5.0;
    while (x > c.spawn_x && n < KS) {
        s.vx[(size_t)e * KS + n] = x; s.vv[(size_t)e * KS + n] = c.other_speed; s.va[(size_t)e * KS + n] = 0.0; s.vc[(size_t)e * KS + n] = cruise_speed(c, e, ctr); ++n;
        const double step = c.other_speed * (c.base_interval + (c.vary_interval ? uniform01(c.seed, e, ctr) : 0.0));
        x -= step > min_gap ? step : min_gap;
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
    for (int q = 0; q < 8; ++q) s.acc[(size_t)e * 8 + q] = 0.0;
    s.acc[(size_t)e * 8 + 5] = 1e300;
}
// The planner's view of each environment: the vehicles within the sensor radius, front to back, and the ego with its s coordinate.
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
        if (fabs(x - ex) <= c.sensor_radius) { ox[(size_t)e * Kmax + k] = x; ov[(size_t)e * Kmax + k] = s.vv[(size_t)e * KS + i]; if (oa) oa[(size_t)e * Kmax + k] = s.va[(size_t)e * KS + i]; ++k; }
    }
    for (int i = k; i < Kmax; ++i) { ox[(size_t)e * Kmax + i] = 0.0; ov[(size_t)e * Kmax + i] = 0.0; if (oa) oa[(size_t)e * Kmax + i] = 0.0; }
    k_count[e] = k;
}
// One simulator tick with the commanded ego speed.
__global__ void __launch_bounds__(64) k_sim_step(DevP p, Cfg c, int N, State s, const double *__restrict__ cmd_speed, double crash_min_s) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N || s.status[e] != 0) return;
    const double dt = c.tick;
    double cx = s.ego4[e * 4 + 0], cy = s.ego4[e * 4 + 1];
    const double v_prev = s.ego4[e * 4 + 2];
    double sel = cmd_speed[e];
    if (!(sel == sel)) sel = v_prev;                                              // (a NaN command keeps the speed)
    // the vehicle cannot change its speed faster than its acceleration limits (SUMO vType accel 4.5 / decel 6.0, merge.rou.xml:2)
    const double hi = v_prev + p.a_max * dt, lo = v_prev + p.a_min * dt;
    sel = sel > hi ? hi : (sel < lo ? lo : sel);
    sel = sel < 0 ? 0 : (sel > p.v_max ? p.v_max : sel);
    // ego motion, prediction.py:46-59
    double px, py;
    if (cx < 1.5) {
        double d0 = 1.5 - cx, d1 = -1.5 - cy;
        const double nrm = sqrt(__builtin_fma(d1, d1, d0 * d0));
        d0 /= nrm; d1 /= nrm;
        px = cx + d0 * (sel * dt); py = cy + d1 * (sel * dt);
        if (py < -1.6) py = -1.6;
    } else { py = cy; px = cx + sel * dt; }
    const double acc_ego = (sel - v_prev) / dt;
    const double es = dev_ego_s(px, py);
    const bool can_crash = es > p.crash_thr, merged = es > p.react_thr;
    // highway vehicles, prediction.py:75-97
    const int n = s.nveh[e];
    double last_x = __builtin_inf(), last_speed = 0.0;
    bool enc = false, crashed = false;
    double gap = 100.0;
    for (int i = 0; i < n; ++i) {
        const double ox_ = s.vx[(size_t)e * KS + i], ov_ = s.vv[(size_t)e * KS + i];
        if (ox_ < px && !enc) { enc = true; if (merged) { last_x = px; last_speed = sel; } }
        const double sd = last_speed - ov_, xd = last_x - ox_;
        double a_ = 0.0, nv = ov_;
        if (sd < 0 && xd < p.follow_gap) { a_ = sd > p.max_pred_decel ? sd : p.max_pred_decel; nv = ov_ + a_ * dt; }
        else {                                                                    // free road: towards the vehicle's own cruise speed
            const double vc_ = s.vc[(size_t)e * KS + i];
            if (ov_ < vc_) { a_ = (vc_ - ov_) / dt; a_ = a_ > 2.6 ? 2.6 : a_; nv = ov_ + a_ * dt; }
            else if (ov_ > vc_) { a_ = (vc_ - ov_) / dt; a_ = a_ < -2.0 ? -2.0 : a_; nv = ov_ + a_ * dt; }
        }
        const double nx = ox_ + nv * dt;
        last_x = nx; last_speed = nv;
        s.vx[(size_t)e * KS + i] = nx; s.vv[(size_t)e * KS + i] = nv; s.va[(size_t)e * KS + i] = a_;
        const double d = fabs(nx - px);
        if (d < p.car_length && can_crash) crashed = true;
        gap = d < gap ? d : gap;
    }
    // vehicles leaving at the end of the highway (front of the list) and entering at its start, control.py:215-226
    int nn = n, drop = 0;
    while (drop < nn && s.vx[(size_t)e * KS + drop] > c.despawn_x) ++drop;
    if (drop) { for (int i = drop; i < nn; ++i) { s.vx[(size_t)e * KS + i - drop] = s.vx[(size_t)e * KS + i]; s.vv[(size_t)e * KS + i - drop] = s.vv[(size_t)e * KS + i]; s.va[(size_t)e * KS + i - drop] = s.va[(size_t)e * KS + i]; s.vc[(size_t)e * KS + i - drop] = s.vc[(size_t)e * KS + i]; } nn -= drop; }
    double delay = s.delay[e];
    unsigned ctr = s.rng[e];
    if (delay <= 0) {
        const bool room = nn == 0 || s.vx[(size_t)e * KS + nn - 1] - c.spawn_x >= 2.5 + c.other_speed + 5.0;       // SUMO postpones an unsafe insertion
        if (room && nn < KS) {
            s.vx[(size_t)e * KS + nn] = c.spawn_x; s.vv[(size_t)e * KS + nn] = c.other_speed; s.va[(size_t)e * KS + nn] = 0.0; s.vc[(size_t)e * KS + nn] = cruise_speed(c, e, ctr); ++nn;
            delay = (c.vary_interval ? uniform01(c.seed, e, ctr) : 0.0) + c.base_interval;
        }
    }
    delay -= dt;
    s.delay[e] = delay; s.rng[e] = ctr; s.nveh[e] = nn;
    s.ego4[e * 4 + 0] = px; s.ego4[e * 4 + 1] = py; s.ego4[e * 4 + 2] = sel; s.ego4[e * 4 + 3] = acc_ego;
    // episode statistics, control.py:276-283 / stats.py:43-75
    double *a = s.acc + (size_t)e * 8;
    const int tk = s.ticks[e];
    a[0] += sel; a[1] = sel > a[1] ? sel : a[1];
    if (tk > 0) a[2] += fabs((acc_ego - a[3]) / dt);
    a[3] = acc_ego; a[4] += 1.0;
    if (es > crash_min_s) { a[5] = gap < a[5] ? gap : a[5]; a[6] += gap; a[7] += 1.0; }
    s.ticks[e] = tk + 1;
    if (crashed) s.status[e] = 2;
    else if (px >= c.arrive_x) s.status[e] = 1;
    else if (tk + 1 >= c.max_ticks) s.status[e] = 3;
}
}  // namespace sim
}  // namespace stmpc
