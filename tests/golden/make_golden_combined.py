#!/usr/bin/env python3
"""Golden vectors for the combined RL+ST controller's decision logic (dqn.RLAgent.do_combined_control,
dqn.py:117-200), generated from the reference itself with a deterministic stand-in policy.

Build-container only (needs /root/reference).  Inert stand-ins: traci, cvxopt and torch.utils.tensorboard
(logging) -- none is called on this path.  st.do_st_control and control.set_ego_jerk (TraCI side effects) are
replaced by recorders so that only the DECISION is captured; the QP-resampled speed is not (cvxopt absent).

Second part ("b_*" arrays): the TEST_ST_STRICTLY_BETTER branch (dqn.py:156-197, the 'b' configs), which compares the
QP-resampled ST path with the policy's rollout.  The reference's own lattice solver, rollout, jerk/distance
comparison and switching rules run unmodified; only `st.finer_fit`'s call into cvxopt is replaced -- by this repo's
restatement of cvxopt's iteration (oracle/ff_oracle.c), since cvxopt is not installed.  These vectors therefore pin
the branch's logic around the QP, not the QP solve itself.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
from make_golden import import_reference, REF      # noqa: E402


def stub_policy(state):
    """Deterministic stand-in for the actor network: a jerk in [-5, 5] from plain float arithmetic on the state."""
    gap = 100.0
    for x in state.other_xs:
        d = x - state.ego_position[0]
        if 0.0 <= d < gap:
            gap = d
    j = 0.4 * (18.0 - state.ego_speed) - 0.8 * state.ego_acceleration - 25.0 / (gap + 5.0) + 1.0
    return max(-5.0, min(5.0, j))


def main():
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    S, control, prediction, st, st_cy = import_reference()
    import dqn                                                  # noqa: E402  (the reference's)
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import synth
    S.load_from_file(os.path.join(REF, "configs", "combined_medium_1.json"))     # BASELINE config 3
    for k_, v_ in pkg.REFERENCE_DEFAULT.items():
        setattr(S, k_, v_)

    class Agent(dqn.RLAgent):
        @classmethod
        def load(cls, path): pass
        @classmethod
        def train(cls, num_frames): pass
        @classmethod
        def resume_training(cls, path, num_frames): pass
        def get_control(self, state):
            return stub_policy(state)

    calls = {}
    msgs = []
    dqn.print = lambda *a, **kw: msgs.append(str(a[0]))         # the takeover messages of dqn.py:145,149,153
    st.do_st_control = lambda state: calls.setdefault("st", True) and "ST"
    control.set_ego_jerk = lambda jerk: calls.setdefault("rl", jerk) and "RL"
    # the reference looks these up as module attributes at call time (dqn.py:147,155,200)

    ego, k, ox, ov = synth.generate_states(600, k=6, kmax=8, seed=31, vary_k=True, dt=0.2, blocked_quota=0.02)
    # closer, faster traffic so that all three outcomes occur
    ego[:, 0] = np.random.default_rng(5).uniform(-120.0, 40.0, ego.shape[0])
    ego[:, 1] = synth.road_y(ego[:, 0])
    takeover = np.zeros(len(k), dtype=np.int32)
    reason = np.zeros(len(k), dtype=np.int32)
    codes = {"Crash predicted": 1, "DDPG going too fast": 2, "ST solver not happy with rollout state": 3}
    for i in range(len(k)):
        kk = int(k[i])
        state = prediction.HighwayState((float(ego[i, 0]), float(ego[i, 1])), float(ego[i, 2]), float(ego[i, 3]),
                                        [float(x) for x in ox[i, :kk]], [float(x) for x in ov[i, :kk]], [0.0] * kk)
        ego[i, 4] = control.get_ego_s(state.ego_position)
        agent = Agent()
        calls.clear()
        msgs.clear()
        agent.do_combined_control(state)
        reason[i] = codes[msgs[0]] if msgs else 0
        assert len(agent.takeover_history) == 1
        takeover[i] = int(agent.takeover_history[0])
        assert ("st" in calls) == bool(takeover[i])
    keys = ["ROLLOUT_LENGTH", "ST_TEST_ROLLOUTS", "COMBINATION_MIN_DISTANCE", "STOP_X", "TICK_LENGTH"]
    np.savez_compressed(os.path.join(HERE, "golden_combined.npz"), ego=ego, k_count=k, other_x=ox, other_v=ov,
                        takeover=takeover, reason=reason, setting_keys=np.array(keys), setting_vals=np.array([float(getattr(S, q)) for q in keys]),
                        flags=np.array([int(S.CHECK_ROLLOUT_CRASH), int(S.LIMIT_DQN_SPEED), int(S.TEST_ROLLOUT_STATE),
                                        int(S.TEST_ST_STRICTLY_BETTER)]))
    print("combined: %d states, %d takeovers, reasons %s" % (len(k), int(takeover.sum()), np.bincount(reason).tolist()))

    # ---- part b: TEST_ST_STRICTLY_BETTER
    from oracle import ff_oracle as ff
    FS = ff.settings(S.MAX_SPEED, S.MAX_POSITIVE_ACCELERATION, S.MAX_NEGATIVE_ACCELERATION, S.MAXIMUM_POSITIVE_JERK,
                     S.MINIMUM_NEGATIVE_JERK, S.CAR_LENGTH)

    def finer_fit(s_sequence, delta_t, coarse_delta_t, start_speed, start_acceleration, before_after_cars=None):
        return ff.finer_fit(np.asarray(s_sequence, dtype=np.float64), delta_t, coarse_delta_t, start_speed, start_acceleration, FS,
                            before_after_cars)[0]
    st.finer_fit = finer_fit
    S.TEST_ST_STRICTLY_BETTER = True
    nb = 360
    b_takeover = np.zeros(nb, dtype=np.int32); b_reason = np.zeros(nb, dtype=np.int32)
    b_speed = np.full(nb, np.nan); b_last_rl = np.ones(nb, dtype=np.int32); b_remember = np.zeros(nb, dtype=np.int32)
    codes["ST Path deemed better"] = 4
    codes["RL path deemed better"] = 0
    sent = []
    control.set_ego_speed = lambda v: sent.append(float(v))
    for i in range(nb):
        kk = int(k[i])
        state = prediction.HighwayState((float(ego[i, 0]), float(ego[i, 1])), float(ego[i, 2]), float(ego[i, 3]),
                                        [float(x) for x in ox[i, :kk]], [float(x) for x in ov[i, :kk]], [0.0] * kk)
        agent = Agent()
        b_remember[i] = int(i >= nb // 2)
        S.REMEMBER_LAST_CHOICE_FOR_SWITCHING_COMBINED = bool(b_remember[i])
        if i % 3 == 1:
            agent.takeover_history.append(True)                 # previous tick was an ST takeover
            b_last_rl[i] = 0
        calls.clear(); msgs.clear(); sent.clear()
        agent.do_combined_control(state)
        b_takeover[i] = int(agent.takeover_history[-1])
        if "st" in calls:                                        # crash / rollout-probe takeovers: do_st_control(start_state)
            b_reason[i] = codes[msgs[0]]
        elif sent:                                               # ST path chosen by the comparison
            b_reason[i] = 4
            b_speed[i] = sent[0]
        else:
            b_reason[i] = 0
        assert b_takeover[i] == int(b_reason[i] != 0)
    S.REMEMBER_LAST_CHOICE_FOR_SWITCHING_COMBINED = False
    S.TEST_ST_STRICTLY_BETTER = False
    np.savez_compressed(os.path.join(HERE, "golden_combined_b.npz"), n=np.array(nb), b_takeover=b_takeover, b_reason=b_reason,
                        b_speed=b_speed, b_last_rl=b_last_rl, b_remember=b_remember)
    print("combined b: %d states, reasons %s (remember off) %s (remember on)" % (
        nb, np.bincount(b_reason[:nb // 2], minlength=5).tolist(), np.bincount(b_reason[nb // 2:], minlength=5).tolist()))


if __name__ == "__main__":
    main()
