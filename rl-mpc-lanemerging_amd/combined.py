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


REASON_RL, REASON_CRASH, REASON_SPEED, REASON_ROLLOUT = 0, 1, 2, 3
REASON_NAMES = {REASON_RL: "rl", REASON_CRASH: "crash predicted", REASON_SPEED: "too fast",
                REASON_ROLLOUT: "st solver not happy with rollout state"}


def _unpack(ego4, k, ox, ov, i):
    kk = int(k[i])
    return HighwayState((float(ego4[i, 0]), float(ego4[i, 1])), float(ego4[i, 2]), float(ego4[i, 3]),
                        [float(x) for x in ox[i, :kk]], [float(x) for x in ov[i, :kk]], [0.0] * kk)


def decide_batch(states, get_control, ctx=None):
    """Decision part of ``do_combined_control`` for a list of states.

    Returns a dict: ``takeover[n]`` (bool), ``reason[n]`` (REASON_*), ``first_action[n]``, ``selected_speed[n]``
    (speed of the last rollout step), ``crash_predicted[n]``, ``test_states`` (list of HighwayState, the
    state the feasibility probe ran on).  ``get_control`` is called once per live state per rollout step,
    in state order, exactly as the reference calls its policy.
    """
    if getattr(Settings, "TEST_ST_STRICTLY_BETTER", False):
        raise NotImplementedError("the TEST_ST_STRICTLY_BETTER branch (dqn.py:156-197, the 'b' combined configs) is not built; "
                                  "use the non-'b' combined configs")
    ctx = ctx or _capi.default_context()
    params = _capi.Params.from_settings(Settings)
    n = len(states)
    ego5, k, ox, ov = pack_states(states)
    ego4 = np.ascontiguousarray(ego5[:, :4])
    first_action = np.array([get_control(s) for s in states], dtype=np.float64)
    crash_predicted = np.zeros(n, dtype=bool)
    selected_speed = np.zeros(n, dtype=np.float64)
    live = np.ones(n, dtype=bool)
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
    return {"takeover": reason != REASON_RL, "reason": reason, "first_action": first_action,
            "selected_speed": selected_speed, "crash_predicted": crash_predicted, "test_states": test_states}


class CombinedController:
    """Stateful single-state wrapper with the reference's call shape (``control_function(state)``)."""

    def __init__(self, get_control):
        self.get_control = get_control
        self.takeover_history = []

    def do_combined_control(self, state):
        d = decide_batch([state], self.get_control)
        take = bool(d["takeover"][0])
        self.takeover_history.append(take)
        if take:
            print({REASON_CRASH: "Crash predicted", REASON_SPEED: "DDPG going too fast",
                   REASON_ROLLOUT: "ST solver not happy with rollout state"}[int(d["reason"][0])])
            return st.do_st_control(state)
        new_speed = get_ego_speed_from_jerk(state.ego_speed, state.ego_acceleration, float(d["first_action"][0]))
        control.set_ego_speed(new_speed)                                          # control.set_ego_jerk, control.py:174-178
        return new_speed
