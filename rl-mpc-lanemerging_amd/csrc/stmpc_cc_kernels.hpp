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
__global__ void __launch_bounds__(64) k_cc_probe_state(int N, int Kmax, const double *__restrict__ cur_ego4, const double *__restrict__ cur_ox,
                                                       const double *__restrict__ cur_ov, CCState st, double *probe_ego5, double *probe_ox, double *probe_ov) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const bool ht = st.have_test[e] != 0;
    const double *src4 = ht ? st.test_ego4 + (size_t)e * 4 : cur_ego4 + (size_t)e * 4;
    const double *sx = ht ? st.test_ox + (size_t)e * Kmax : cur_ox + (size_t)e * Kmax;
    const double *sv = ht ? st.test_ov + (size_t)e * Kmax : cur_ov + (size_t)e * Kmax;
    for (int q = 0; q < 4; ++q) probe_ego5[(size_t)e * 5 + q] = src4[q];
    probe_ego5[(size_t)e * 5 + 4] = dev_ego_s(src4[0], src4[1]);
    for (int i = 0; i < Kmax; ++i) { probe_ox[(size_t)e * Kmax + i] = sx[i]; probe_ov[(size_t)e * Kmax + i] = sv[i]; }
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
        if (m < 0) atomicExch(err, 1u);                                         // fine grid longer than the QP kernel supports
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
    if (reason == CC_CRASH || reason == CC_SPEED || reason == CC_ROLLOUT) speed = st_speed[e];   // st.do_st_control(start_state)
    takeover[e] = reason != CC_RL;
    reason_out[e] = reason;
    speed_out[e] = speed;
}

}  // namespace stmpc
