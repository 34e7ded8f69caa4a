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
    """Speed after one tick of constant jerk, acceleration and speed clamped to their limits (the host twin of the
    first lines of ``k_rollout_step``; same result as control.py:160-171)."""
    tick = Settings.TICK_LENGTH
    acc = min(max(current_acceleration + jerk * tick, Settings.MAX_NEGATIVE_ACCELERATION), Settings.MAX_POSITIVE_ACCELERATION)
    return min(max(current_speed + acc * tick, 0), Settings.MAX_SPEED)


REASON_RL, REASON_CRASH, REASON_SPEED, REASON_ROLLOUT, REASON_ST_BETTER = 0, 1, 2, 3, 4
REASON_NAMES = {REASON_RL: "rl", REASON_CRASH: "crash predicted", REASON_SPEED: "too fast",
                REASON_ROLLOUT: "st solver not happy with rollout state", REASON_ST_BETTER: "st path deemed better"}


def _unpack(ego4, k, ox, ov, oa, i):
    kk = int(k[i])
    return HighwayState((float(ego4[i, 0]), float(ego4[i, 1])), float(ego4[i, 2]), float(ego4[i, 3]),
                        [float(x) for x in ox[i, :kk]], [float(x) for x in ov[i, :kk]], [float(x) for x in oa[i, :kk]])


def decide_batch_device(ctx, params, cfg, d_ego5, d_k, d_ox, d_ov, policy, d_last_choice_rl=None, stream=0, d_oa=None):
    """One tick of ``do_combined_control`` for N states held in device tensors (torch, fp64 / int32), nothing leaves the GPU
    (with ``cfg.sparse_control`` one integer does: the number of states whose decision calls the controller).
    ``d_oa``: the other vehicles' accelerations of the start states (``HighwayState.other_accelerations``; read by the policy's state vector
    only, dqn.py:399-400); zeros if not given.

    ``policy(step, cur_ego4, k, cur_ox, cur_ov, cur_oa) -> jerk[N]`` (fp64 device tensor) is the caller's network; it is asked once
    per rollout step for every episode (entries of episodes whose rollout has ended are ignored, dqn.py:129-141).  Returns device
    tensors ``takeover``, ``reason`` (REASON_*), ``speed`` (the command) and ``first_action``; the rollout bookkeeping stays in the
    context (``ctx.combined_read_state``)."""
    import torch
    n, K = d_ego5.shape[0], d_ox.shape[1]
    cur_ego4 = d_ego5[:, :4].clone().contiguous()      # (a copy: with one row the slice is already contiguous and would alias the start state)
    cur_ox, cur_ov = d_ox.clone(), d_ov.clone()
    cur_oa = d_oa.clone() if d_oa is not None else torch.zeros_like(d_ox)
    first_action = None
    for step in range(1, max(int(cfg.rollout_length), 1) + 1):
        action = policy(step, cur_ego4, d_k, cur_ox, cur_ov, cur_oa).to(torch.float64).contiguous()
        if step == 1:
            first_action = action.clone()
        ctx.rollout_step_device(params, cfg, n, K, step, d_ego5.data_ptr(), cur_ego4.data_ptr(), d_k.data_ptr(), cur_ox.data_ptr(),
                                cur_ov.data_ptr(), cur_oa.data_ptr(), action.data_ptr(), stream)
    takeover = torch.empty(n, dtype=torch.int32, device=d_ego5.device)
    reason = torch.empty(n, dtype=torch.int32, device=d_ego5.device)
    speed = torch.empty(n, dtype=torch.float64, device=d_ego5.device)
    ctx.combined_decide_device(params, cfg, n, K, d_ego5.data_ptr(), d_k.data_ptr(), d_ox.data_ptr(), d_ov.data_ptr(), cur_ego4.data_ptr(),
                               cur_ox.data_ptr(), cur_ov.data_ptr(), first_action.data_ptr(),
                               d_last_choice_rl.data_ptr() if d_last_choice_rl is not None else 0,
                               takeover.data_ptr(), reason.data_ptr(), speed.data_ptr(), stream)
    return {"takeover": takeover, "reason": reason, "speed": speed, "first_action": first_action,
            "cur_ego4": cur_ego4, "cur_ox": cur_ox, "cur_ov": cur_ov, "cur_oa": cur_oa}      # the last rolled-out state of every episode


def decide_batch(states, get_control, ctx=None, last_choice_rl=None):
    """Decision part of ``do_combined_control`` for a list of states, with the reference's per-state policy callback.

    The rollout, the feasibility probe, the controller solve and the decision run on the GPU (``decide_batch_device``); only the
    policy callback needs the rolled-out states on the host.  Returns a dict: ``takeover[n]`` (bool), ``reason[n]`` (REASON_*),
    ``first_action[n]``, ``selected_speed[n]`` (speed of the last rollout step), ``crash_predicted[n]``, ``test_states`` (list of
    HighwayState, the state the feasibility probe ran on), ``st_speed[n]`` (NaN unless the ST path was chosen by the
    strictly-better comparison, dqn.py:156-197; then the speed to command), ``speed[n]`` (the command in every case), ``rollout_s``.
    ``get_control`` is called once per live state per rollout step, in state order, exactly as the reference calls its policy,
    on states that carry the predictor's ``other_accelerations``.  ``last_choice_rl[n]``: whether the previous tick of that
    episode was left to the policy (dqn.py:124-126; default True = empty takeover history).
    """
    import torch
    ctx = ctx or _capi.default_context()
    params = _capi.Params.from_settings(Settings)
    cfg = _capi.CombinedCfg.from_settings(Settings)
    n = len(states)
    ego5, k, ox, ov = pack_states(states)
    K = ox.shape[1]
    dev = torch.device("cuda", torch.cuda.current_device())
    d_ego5, d_k = torch.as_tensor(ego5, device=dev), torch.as_tensor(k, device=dev)
    d_ox, d_ov = torch.as_tensor(ox, device=dev), torch.as_tensor(ov, device=dev)
    R = max(int(Settings.ROLLOUT_LENGTH), 1)

    def policy(step, cur_ego4, d_k_, cur_ox, cur_ov, cur_oa):
        if step == 1:
            act = np.array([get_control(s) for s in states], dtype=np.float64)
        else:
            live = ctx.combined_read_state(n, K, R, after_decide=False)["live"].astype(bool)
            e4, xo, vo, ao = cur_ego4.cpu().numpy(), cur_ox.cpu().numpy(), cur_ov.cpu().numpy(), cur_oa.cpu().numpy()
            act = np.zeros(n)
            for j in np.nonzero(live)[0]:
                act[j] = get_control(_unpack(e4, k, xo, vo, ao, j))
        return torch.as_tensor(act, device=dev)

    d_last = None
    if last_choice_rl is not None:
        d_last = torch.as_tensor(np.asarray(last_choice_rl, dtype=bool).astype(np.int32), device=dev)
    d = decide_batch_device(ctx, params, cfg, d_ego5, d_k, d_ox, d_ov, policy, d_last)
    torch.cuda.synchronize()
    st_ = ctx.combined_read_state(n, K, R, after_decide=True)
    reason = d["reason"].cpu().numpy()
    speed = d["speed"].cpu().numpy()
    hist = [list(st_["rollout_s"][j, :st_["hist_len"][j]]) for j in range(n)]
    # the probe state: the state after ST_TEST_ROLLOUTS steps, else the last rolled-out state (dqn.py:137-143)
    test_states = []
    fin = [d[q].cpu().numpy() for q in ("cur_ego4", "cur_ox", "cur_ov", "cur_oa")]
    for j in range(n):
        if st_["have_test"][j]:
            test_states.append(_unpack(st_["test_ego4"], k, st_["test_ox"], st_["test_ov"], np.zeros_like(ox), j))
        else:                                           # rollout ended before ST_TEST_ROLLOUTS: the last rolled-out state (dqn.py:142-143)
            test_states.append(_unpack(fin[0], k, fin[1], fin[2], fin[3], j))
    st_speed = np.where(reason == REASON_ST_BETTER, speed, np.nan)
    return {"takeover": reason != REASON_RL, "reason": reason, "first_action": d["first_action"].cpu().numpy(),
            "selected_speed": st_["sel_speed"], "crash_predicted": st_["crash_pred"].astype(bool), "test_states": test_states,
            "st_speed": st_speed, "speed": speed, "rollout_s": hist}


class CombinedController:
    """Stateful single-state wrapper with the reference's call shape (``control_function(state)``)."""

    def __init__(self, get_control):
        self.get_control = get_control
        self.takeover_history = []

    def do_combined_control(self, state):
        last_choice_rl = not (len(self.takeover_history) > 0 and self.takeover_history[-1])     # dqn.py:124-126
        d = decide_batch([state], self.get_control, last_choice_rl=[last_choice_rl])
        take = bool(d["takeover"][0])
        reason = int(d["reason"][0])
        self.takeover_history.append(take)
        speed = float(d["speed"][0])
        if reason == REASON_ST_BETTER:
            if last_choice_rl or not Settings.REMEMBER_LAST_CHOICE_FOR_SWITCHING_COMBINED:
                print("ST Path deemed better")
        elif take:
            print({REASON_CRASH: "Crash predicted", REASON_SPEED: "DDPG going too fast",
                   REASON_ROLLOUT: "ST solver not happy with rollout state"}[reason])
        control.set_ego_speed(speed)           # st.do_st_control / control.set_ego_speed / control.set_ego_jerk all end here (st.py:782, control.py:174-178)
        return speed
