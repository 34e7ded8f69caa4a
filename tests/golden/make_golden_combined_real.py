#!/usr/bin/env python3
"""Golden vectors for the combined RL+ST controller with the reference's REAL pretrained actor
(BASELINE configs[2]: configs/combined_medium_1.json -> runs/ddpg_medium1_extended).

Build-container only (needs /root/reference).  What runs is the reference's own code:
  * dqn.RLAgent.do_combined_control (dqn.py:117-200), unmodified, under combined_medium_1.json;
  * its get_control is DDPGAgent.get_control (ddpg.py:83-87) with the two pieces of the absent ``all`` library restated:
      - dqn.get_state_vector_from_base_state (dqn.py:389-446, the reference's own function) -> 20 doubles,
      - GymEnvironment._make_state: cast to the observation space's float32,
      - TimeFeature (all/bodies/time.py of all 0.5.3, restated -- PARITY UNPINNED, the library is absent): append
        0.001 * (number of evaluations since the episode began) as the 21st input, then count one up.  Note that under the
        combined controller the counter advances once per rollout step, not once per tick, because every rollout step
        calls get_control (dqn.py:132-133);
      - the pretrained network itself: the tensors of policy.pt in a torch fp32 nn.Sequential, 5 * tanh (see rl-mpc-lanemerging_amd/data/make_actor_weights.py).
  * st.do_st_control / control.set_ego_jerk are recorders (TraCI side effects), as in make_golden_combined.py.

Recorded per state: the inputs (incl. other vehicles' accelerations and the evaluation counter at the tick's start), every policy
evaluation of the rollout (the 21-float input vector and the jerk), the decision and its reason.
Re-run:  python tests/golden/make_golden_combined_real.py
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


def main():
    import torch
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb
    S, control, prediction, st, st_cy = import_reference()
    import dqn                                                  # noqa: E402  (the reference's)
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import synth
    S.load_from_file(os.path.join(REF, "configs", "combined_medium_1.json"))     # BASELINE configs[2]
    for k_, v_ in pkg.REFERENCE_DEFAULT.items():
        setattr(S, k_, v_)
    assert S.MODEL_NAME == "runs/ddpg_medium1_extended"
    w = np.load(os.path.join(HERE, "..", "..", "rl-mpc-lanemerging_amd", "data", "actor_ddpg_medium1.npz"))
    net = torch.nn.Sequential(torch.nn.Linear(21, 400), torch.nn.ReLU(), torch.nn.Linear(400, 300), torch.nn.ReLU(), torch.nn.Linear(300, 1))
    with torch.no_grad():
        for layer, (wk, bk) in zip((net[0], net[2], net[4]), (("w0", "b0"), ("w1", "b1"), ("w2", "b2"))):
            layer.weight.copy_(torch.from_numpy(w[wk]))
            layer.bias.copy_(torch.from_numpy(w[bk]))
    scale, mean = float(w["tanh_scale"]), float(w["tanh_mean"])

    class Agent(dqn.RLAgent):
        def __init__(self, evaluations):
            super().__init__()
            self.timestep = evaluations          # TimeFeature.timestep
            self.vectors, self.jerks = [], []
        @classmethod
        def load(cls, path): pass
        @classmethod
        def train(cls, num_frames): pass
        @classmethod
        def resume_training(cls, path, num_frames): pass
        def get_control(self, state):                                       # ddpg.py:83-87
            vector_state = dqn.get_state_vector_from_base_state(state)      # the reference's own
            obs = torch.from_numpy(np.array(vector_state, dtype=np.float32)).unsqueeze(0)
            feat = torch.cat((obs, 0.001 * torch.tensor([float(self.timestep)]).view(-1, 1)), dim=1)
            self.timestep += 1
            with torch.no_grad():
                out = torch.tanh(net(feat.float())) * scale + mean
            self.vectors.append(feat[0].numpy().copy())
            self.jerks.append(out.item())
            return out.item()

    calls, msgs = {}, []
    dqn.print = lambda *a, **kw: msgs.append(str(a[0]))
    st.do_st_control = lambda state: calls.setdefault("st", True) and "ST"
    control.set_ego_jerk = lambda jerk: calls.setdefault("rl", jerk) and "RL"

    n = 1600
    ego, k, ox, ov = synth.generate_states(n, k=6, kmax=8, seed=77, vary_k=True, dt=0.2, blocked_quota=0.02)
    rng = np.random.default_rng(78)
    ego[:, 0] = rng.uniform(-200.0, 55.0, n)                     # the whole ramp and the first 50 m of the highway
    ego[n // 2:, 0] = rng.uniform(-100.0, 20.0, n - n // 2)      # half of them where the decisions are made: the approach and the merge zone
    ego[:, 1] = synth.road_y(ego[:, 0])
    ego[:, 2] = np.clip(rng.normal(12.0, 4.0, n), 0.5, 24.0)     # what the trained policy drives (saved_data.csv: mean 10-12 m/s)
    ego[:, 3] = np.clip(rng.normal(0.0, 1.0, n), -3.0, 3.0)
    oa = np.where(rng.random(ox.shape) < 0.7, 0.0, rng.uniform(-3.0, 1.0, ox.shape))       # Krauss vehicles: cruising (0) or adjusting
    evals0 = rng.integers(0, 5 * 150, n).astype(np.int32)        # evaluations since the episode began (5 per tick, up to 30 s)
    R = int(S.ROLLOUT_LENGTH)
    takeover = np.zeros(n, dtype=np.int32)
    reason = np.zeros(n, dtype=np.int32)
    jerks = np.full((n, R), np.nan)
    vectors = np.zeros((n, R, 21), dtype=np.float32)
    n_evals = np.zeros(n, dtype=np.int32)
    codes = {"Crash predicted": 1, "DDPG going too fast": 2, "ST solver not happy with rollout state": 3}
    for i in range(n):
        kk = int(k[i])
        oa[i, kk:] = 0.0
        state = prediction.HighwayState((float(ego[i, 0]), float(ego[i, 1])), float(ego[i, 2]), float(ego[i, 3]),
                                        [float(x) for x in ox[i, :kk]], [float(x) for x in ov[i, :kk]], [float(x) for x in oa[i, :kk]])
        ego[i, 4] = control.get_ego_s(state.ego_position)
        agent = Agent(int(evals0[i]))
        calls.clear()
        msgs.clear()
        agent.do_combined_control(state)
        reason[i] = codes[msgs[0]] if msgs else 0
        takeover[i] = int(agent.takeover_history[0])
        assert ("st" in calls) == bool(takeover[i])
        n_evals[i] = len(agent.jerks)
        jerks[i, :n_evals[i]] = agent.jerks
        vectors[i, :n_evals[i]] = np.array(agent.vectors)
    keys = ["ROLLOUT_LENGTH", "ST_TEST_ROLLOUTS", "COMBINATION_MIN_DISTANCE", "STOP_X", "TICK_LENGTH"]
    np.savez_compressed(os.path.join(HERE, "golden_combined_real.npz"), ego=ego, k_count=k, other_x=ox, other_v=ov, other_a=oa, evals0=evals0,
                        jerks=jerks, vectors=vectors, n_evals=n_evals, takeover=takeover, reason=reason, actor=np.array("ddpg_medium1"),
                        setting_keys=np.array(keys), setting_vals=np.array([float(getattr(S, q)) for q in keys]),
                        flags=np.array([int(S.CHECK_ROLLOUT_CRASH), int(S.LIMIT_DQN_SPEED), int(S.TEST_ROLLOUT_STATE),
                                        int(S.TEST_ST_STRICTLY_BETTER)]))
    print("combined, real actor: %d states, %d takeovers, reasons %s, rollouts cut short %d, |jerk| mean %.3f, saturated %d"
          % (n, int(takeover.sum()), np.bincount(reason, minlength=4).tolist(), int((n_evals < R).sum()), float(np.nanmean(np.abs(jerks))),
             int((np.abs(jerks) > 4.99).sum())))


if __name__ == "__main__":
    main()
