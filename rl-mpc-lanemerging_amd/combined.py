"""Combined RL + ST controller: the decision logic of the reference's
``dqn.RLAgent.do_combined_control`` (dqn.py:117-200), batched.

The RL policy stays the caller's (``get_control(state) -> jerk``; in the reference a DDPG/DQN network);
what runs on the GPU here is everything the reference does around it on the CPU: the rollout of the
proposed action through the traffic predictor (``predict_step_with_ego`` with
``COMBINATION_MIN_DISTANCE``, dqn.py:129-141), the feasibility probe of the rolled-out state
(``st.test_guaranteed_crash_from_state``, dqn.py:152) and, on takeover, the ST solve of the start state
(``st.do_st_control``, dqn.py:147,155).  SURVEY section 8 row f2.
"""
import numpy as np

from . import _capi
from . import control
from . import st
from .config import Settings
from .prediction import HighwayState, pack_states


def get_ego_speed_from_jerk(current_speed, current_acceleration, jerk):
    # control.py:160-171
    new_acceleration = current_acceleration + jerk * Settings.TICK_LENGTH
    if new_acceleration > Settings.MAX_POSITIVE_ACCELERATION:
        new_acceleration = Settings.MAX_POSITIVE_ACCELERATION
    if new_acceleration < Settings.MAX_NEGATIVE_ACCELERATION:
        new_acceleration = Settings.MAX_NEGATIVE_ACCELERATION
    new_speed = current_speed + new_acceleration * Settings.TICK_LENGTH
    if new_speed > Settings.MAX_SPEED:
        new_speed = Settings.MAX_SPEED
    if new_speed < 0:
        new_speed = 0
    return new_speed


REASON_RL, REASON_CRASH, REASON_SPEED, REASON_ROLLOUT, REASON_ST_BETTER = 0, 1, 2, 3, 4
REASON_NAMES = {REASON_RL: "rl", REASON_CRASH: "crash predicted", REASON_SPEED: "too fast",
                REASON_ROLLOUT: "st solver not happy with rollout state", REASON_ST_BETTER: "st path deemed better"}


def _unpack(ego4, k, ox, ov, i):
    kk = int(k[i])
    return HighwayState((float(ego4[i, 0]), float(ego4[i, 1])), float(ego4[i, 2]), float(ego4[i, 3]),
                        [float(x) for x in ox[i, :kk]], [float(x) for x in ov[i, :kk]], [0.0] * kk)


def decide_batch(states, get_control, ctx=None, last_choice_rl=None):
    """Decision part of ``do_combined_control`` for a list of states.

    Returns a dict: ``takeover[n]`` (bool), ``reason[n]`` (REASON_*), ``first_action[n]``, ``selected_speed[n]``
    (speed of the last rollout step), ``crash_predicted[n]``, ``test_states`` (list of HighwayState, the
    state the feasibility probe ran on), ``st_speed[n]`` (NaN unless the ST path was chosen by the
    strictly-better comparison, dqn.py:156-197; then the speed to command).  ``get_control`` is called once per
    live state per rollout step, in state order, exactly as the reference calls its policy.
    ``last_choice_rl[n]``: whether the previous tick of that episode was left to the policy (dqn.py:124-126;
    default True = empty takeover history).
    """
    ctx = ctx or _capi.default_context()
    params = _capi.Params.from_settings(Settings)
    n = len(states)
    ego5, k, ox, ov = pack_states(states)
    ego4 = np.ascontiguousarray(ego5[:, :4])
    first_action = np.array([get_control(s) for s in states], dtype=np.float64)
    crash_predicted = np.zeros(n, dtype=bool)
    selected_speed = np.zeros(n, dtype=np.float64)
    live = np.ones(n, dtype=bool)
    rollout_s = [[float(ego5[j, 4])] for j in range(n)]                            # dqn.py:121
    test_ego = ego4.copy(); test_ox = ox.copy(); test_ov = ov.copy(); have_test = np.zeros(n, dtype=bool)
    cur_ego, cur_ox, cur_ov = ego4.copy(), ox.copy(), ov.copy()
    steps = max(Settings.ROLLOUT_LENGTH, 1)
    for i in range(1, steps + 1):                                                 # dqn.py:129-141
        idx = np.nonzero(live)[0]
        if idx.size == 0:
            break
        if i == 1:
            action = first_action[idx]
        else:
            action = np.array([get_control(_unpack(cur_ego, k, cur_ox, cur_ov, j)) for j in idx], dtype=np.float64)
        sel = np.array([get_ego_speed_from_jerk(float(cur_ego[j, 2]), float(cur_ego[j, 3]), float(a))
                        for j, a in zip(idx, action)], dtype=np.float64)
        eo, xo, vo, cr = ctx.predict_batch(params, 0, cur_ego[idx], k[idx], cur_ox[idx], cur_ov[idx], sel,
                                           Settings.TICK_LENGTH, Settings.COMBINATION_MIN_DISTANCE)
        cur_ego[idx], cur_ox[idx], cur_ov[idx] = eo, xo, vo
        selected_speed[idx] = sel
        crash_predicted[idx] = cr.astype(bool)
        if i == Settings.ST_TEST_ROLLOUTS:
            test_ego[idx], test_ox[idx], test_ov[idx] = eo, xo, vo
            have_test[idx] = True
        for r, j in enumerate(idx):                                                # dqn.py:138
            rollout_s[j].append(control.get_ego_s((float(eo[r, 0]), float(eo[r, 1]))))
        live[idx] = ~crash_predicted[idx] & ~(eo[:, 0] > Settings.STOP_X)
    nt = ~have_test                                                               # dqn.py:142-143
    test_ego[nt], test_ox[nt], test_ov[nt] = cur_ego[nt], cur_ox[nt], cur_ov[nt]

    reason = np.full(n, REASON_RL, dtype=np.int32)
    if Settings.CHECK_ROLLOUT_CRASH:
        reason[crash_predicted] = REASON_CRASH
    if getattr(Settings, "LIMIT_DQN_SPEED", False):
        m = (reason == REASON_RL) & (selected_speed > Settings.DESIRED_SPEED)
        reason[m] = REASON_SPEED
    test_states = [_unpack(test_ego, k, test_ox, test_ov, j) for j in range(n)]
    if Settings.TEST_ROLLOUT_STATE:
        idx = np.nonzero(reason == REASON_RL)[0]
        if idx.size:
            _, res = st.solve_states([test_states[j] for j in idx], ctx=ctx)       # one batched probe (dqn.py:152)
            reason[idx[res["crash"].astype(bool)]] = REASON_ROLLOUT
    st_speed = np.full(n, np.nan)
    if getattr(Settings, "TEST_ST_STRICTLY_BETTER", False):                        # dqn.py:156-197
        idx = np.nonzero(reason == REASON_RL)[0]
        if idx.size:
            last_rl = np.ones(n, dtype=bool) if last_choice_rl is None else np.asarray(last_choice_rl, dtype=bool)
            res = ctx.st_control_batch(params, Settings.TICK_LENGTH, ego5[idx], k[idx], ox[idx], ov[idx], want_paths=True)
            for r, j in enumerate(idx):
                m = int(res["fine_len"][r])
                if m < 0:
                    raise ValueError("finer_fit: more than %d fine samples are not supported" % _capi.QP_NMAX)
                s_seq = res["fine"][r, :m]                                          # trimmed, QP-resampled when TICK < T_DISCRETIZATION
                if m <= 1:                                                         # dqn.py:167-169
                    continue
                hist = rollout_s[j]
                ml = min(m, len(hist))
                v0, a0 = float(ego5[j, 2]), float(ego5[j, 3])
                st_jerk = st.get_path_mean_abs_jerk(s_seq[:ml], v0, a0, Settings.TICK_LENGTH)
                rl_jerk = st.get_path_mean_abs_jerk(hist[:ml], v0, a0, Settings.TICK_LENGTH)
                st_distance = s_seq[ml - 1] - s_seq[0]
                rl_distance = hist[ml - 1] - hist[0]
                if last_rl[j] or not Settings.REMEMBER_LAST_CHOICE_FOR_SWITCHING_COMBINED:
                    choose_st = (st_jerk < rl_jerk and st_distance > rl_distance) or rl_distance == 0
                else:
                    choose_st = not (rl_jerk < st_jerk and rl_distance > st_distance)
                if choose_st:
                    reason[j] = REASON_ST_BETTER
                    st_speed[j] = (s_seq[1] - s_seq[0]) / Settings.TICK_LENGTH     # dqn.py:180-181,192-193
    return {"takeover": reason != REASON_RL, "reason": reason, "first_action": first_action,
            "selected_speed": selected_speed, "crash_predicted": crash_predicted, "test_states": test_states,
            "st_speed": st_speed, "rollout_s": rollout_s}


class CombinedController:
    """Stateful single-state wrapper with the reference's call shape (``control_function(state)``)."""

    def __init__(self, get_control):
        self.get_control = get_control
        self.takeover_history = []

    def do_combined_control(self, state):
        last_choice_rl = not (len(self.takeover_history) > 0 and self.takeover_history[-1])     # dqn.py:124-126
        d = decide_batch([state], self.get_control, last_choice_rl=[last_choice_rl])
        take = bool(d["takeover"][0])
        self.takeover_history.append(take)
        if int(d["reason"][0]) == REASON_ST_BETTER:
            if last_choice_rl or not Settings.REMEMBER_LAST_CHOICE_FOR_SWITCHING_COMBINED:
                print("ST Path deemed better")
            speed = float(d["st_speed"][0])
            control.set_ego_speed(speed)
            return speed
        if take:
            print({REASON_CRASH: "Crash predicted", REASON_SPEED: "DDPG going too fast",
                   REASON_ROLLOUT: "ST solver not happy with rollout state"}[int(d["reason"][0])])
            return st.do_st_control(state)
        new_speed = get_ego_speed_from_jerk(state.ego_speed, state.ego_acceleration, float(d["first_action"][0]))
        control.set_ego_speed(new_speed)                                          # control.set_ego_jerk, control.py:174-178
        return new_speed
