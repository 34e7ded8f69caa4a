"""The reference's pretrained DDPG actor as the combined controller's policy (BASELINE configs[2]: ``configs/combined_medium_1.json``
-> ``MODEL_NAME runs/ddpg_medium1_extended``), evaluated on the GPU for N states at once.

``DDPGAgent.get_control`` (ddpg.py:83-87) is, per state:

    vector = dqn.get_state_vector_from_base_state(state)        dqn.py:389-446: 20 doubles
    state  = env._make_state(vector, False)                     the RL library's gym wrapper: cast to float32
    action = TimeFeature(GreedyAgent(policy)).eval(state)       ddpg.py:40-41: 21st input = 0.001 x evaluations so far; the network;
                                                                DeterministicPolicyNetwork: tanh(.) * 5 + 0

Here the network is 21 -> 400 -> ReLU -> 300 -> ReLU -> 1 with the tensors of the reference's ``pretrained_models/ddpg_*_extended/policy.pt``
(exported as data by ``data/make_actor_weights.py`` next to this file, into ``data/``), float32 like the reference's, squash ``tanh * tanh_scale + tanh_mean``, in
two interchangeable engines:

* ``engine="hip"`` (default): ONE launch per evaluation, ``k_actor_eval`` (``stmpc_actor_eval_device``): state vector, both hidden layers on
  the matrix cores (``v_mfma_f32_16x16x4_f32``, weights pre-packed in lane order), output layer and squash; activations stay in LDS;
* ``engine="torch"``: the input vectors from ``k_policy_features`` (``stmpc_policy_features_device``), the network on PyTorch-ROCm (three
  rocBLAS GEMMs + elementwise kernels) -- BASELINE configs[2]'s literal wording, and the float32 reference the fused kernel is tested against.

Parity: the 20 state-vector entries and the network's weights are the reference's own (pinned by ``golden_combined_real.npz``); the
float32 cast and the TimeFeature input restate ``autonomous-learning-library`` 0.5.3 (requirements.txt:10), which is absent from the
reference checkout -- parity unpinned for those two steps, and the suite's ``test_actor.py`` measures how much the decisions depend on them.
"""
import os

import numpy as np

from . import _capi

ACTOR_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")      # package data: the reference's pretrained_models/*/policy.pt as tensors
#: MODEL_NAME of the shipped evaluation configs -> exported tensor file (configs/combined_<traffic>_1.json:4)
PRETRAINED = ("low1", "medium1", "default1", "moderate1", "fast1")


def weights_path(name):
    """``name``: "medium1", "ddpg_medium1", "runs/ddpg_medium1_extended" (a config's MODEL_NAME) or a path to an .npz file."""
    if os.path.exists(name):
        return name
    short = os.path.basename(name)
    if short.startswith("ddpg_"):
        short = short[5:]
    if short.endswith("_extended"):
        short = short[:-9]
    path = os.path.join(ACTOR_DIR, "actor_ddpg_%s.npz" % short)
    if not os.path.exists(path):
        raise FileNotFoundError("no exported actor for %r (have: %s); data/make_actor_weights.py writes them" % (name, ", ".join(PRETRAINED)))
    return path


def load_weights(name):
    with np.load(weights_path(name)) as z:
        w = {k: np.array(z[k]) for k in ("w0", "b0", "w1", "b1", "w2", "b2")}
        w["tanh_scale"], w["tanh_mean"] = float(z["tanh_scale"]), float(z["tanh_mean"])
    assert w["w0"].shape[0] == w["b0"].shape[0] == w["w1"].shape[1] and w["w1"].shape[0] == w["w2"].shape[1] and w["w2"].shape[0] == 1
    return w


def state_vector_host(S, ego4, xs, vs, accs, evaluations=None):
    """Host twin of ``k_policy_features`` for ONE state (numpy, same operations): dqn.get_state_vector_from_base_state, the float32 cast and,
    if ``evaluations`` is given, the TimeFeature input.  Used by the CPU suite; the product path is the kernel."""
    fc = _capi.FeaturesCfg.from_settings(S, time_feature=evaluations is not None)
    tw = 4 if fc.use_acceleration else 3
    out = np.zeros((fc.cars_ahead + fc.cars_behind) * tw + 4 + (1 if fc.time_feature else 0), dtype=np.float32)
    ex, ey, ev, ea = (float(q) for q in ego4)

    def put(slot, i):
        t = []
        if fc.use_acceleration:
            t.append(accs[i] / 9.0 if fc.normalize else accs[i])
        dv = vs[i] - ev if fc.use_speed_difference else vs[i]
        t.append(dv / fc.max_speed if fc.normalize else dv)
        dx = xs[i] - ex
        t.append(dx / fc.sensor_radius if fc.normalize else dx)
        t.append(1.0)
        out[slot * tw:(slot + 1) * tw] = np.array(t, dtype=np.float64).astype(np.float32)
    front = [i for i in range(len(xs)) if xs[i] > ex][::-1][:fc.cars_ahead]
    back = [i for i in range(len(xs)) if not xs[i] > ex][:fc.cars_behind]
    for s_, i in enumerate(front):
        put(s_, i)
    for s_, i in enumerate(back):
        put(fc.cars_ahead + s_, i)
    base = (fc.cars_ahead + fc.cars_behind) * tw
    ego = [ev / fc.max_speed, ea / 9.0, ex / 300.0, ey / 100.0] if fc.normalize else [ev, ea, ex, ey]
    out[base:base + 4] = np.array(ego, dtype=np.float64).astype(np.float32)
    if fc.time_feature:
        out[base + 4] = np.float32(fc.time_scale) * np.float32(evaluations)
    return out


class DDPGActor:
    """``policy(step, cur_ego4, k, cur_ox, cur_ov, cur_oa) -> jerk[N]`` for ``combined.decide_batch_device``.

    Holds the per-episode evaluation counters of the reference's TimeFeature wrapper (``evals``, int32 [N] on the device;
    ``reset(mask)`` zeroes them where an episode ends, as ``end_episode_callback`` does, ddpg.py:89-90).
    ``engine``: "hip" (one fused launch) or "torch" (see the module text).  ``dtype`` (torch engine only): torch.float32 evaluates the network
    as the reference does; torch.float64 is offered for sensitivity measurements.  ``time_feature=False`` feeds 0 as the 21st input (the
    sensitivity test's other arm; the evaluation counters still run)."""

    def __init__(self, name, n, ctx, S, device=None, dtype=None, time_feature=True, engine="hip"):
        import torch
        self.torch = torch
        self.name = name
        self.ctx = ctx
        self.n = int(n)
        if engine not in ("hip", "torch"):
            raise ValueError("engine must be 'hip' or 'torch'")
        if dtype is not None and dtype != torch.float32 and engine == "hip":
            engine = "torch"                      # the fused kernel is float32 only
        self.engine = engine
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dtype = dtype or torch.float32
        w = load_weights(name)
        t = lambda a: torch.as_tensor(a, device=dev).to(self.dtype)
        self.w0t, self.b0 = t(w["w0"]).t().contiguous(), t(w["b0"])
        self.w1t, self.b1 = t(w["w1"]).t().contiguous(), t(w["b1"])
        self.w2t, self.b2 = t(w["w2"]).t().contiguous(), t(w["b2"])
        self.scale, self.mean = w["tanh_scale"], w["tanh_mean"]
        self.fcfg = _capi.FeaturesCfg.from_settings(S, time_feature=True)
        if not time_feature:
            self.fcfg.time_scale = 0.0            # the input reads 0 x evaluations
        self.flen = (self.fcfg.cars_ahead + self.fcfg.cars_behind) * (4 if self.fcfg.use_acceleration else 3) + 5
        if self.flen != self.w0t.shape[0]:
            raise ValueError("the actor takes %d inputs, the state vector of these settings has %d" % (self.w0t.shape[0], self.flen))
        self.evals = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.feat = torch.empty(self.n, self.flen, dtype=torch.float32, device=dev)
        self.jerk = torch.empty(self.n, dtype=torch.float64, device=dev)
        self.keep_features = False                # hip engine: also write the input vectors to ``self.feat`` (parity checks)
        self.handle = ctx.actor_create(w) if engine == "hip" else None

    def __del__(self):
        if getattr(self, "handle", None) is not None:
            try:
                self.ctx.actor_destroy(self.handle)
            except Exception:
                pass
            self.handle = None

    def reset(self, mask=None):
        if mask is None:
            self.evals.zero_()
        else:
            self.evals.masked_fill_(mask, 0)

    def _check_rows(self, cur_ego4, k, cur_ox, cur_ov, cur_oa):
        # the kernels are launched for self.n states and write self.jerk / self.feat / self.evals of that length
        for name, t_ in (("cur_ego4", cur_ego4), ("k", k), ("cur_ox", cur_ox), ("cur_ov", cur_ov), ("cur_oa", cur_oa)):
            if t_ is not None and t_.shape[0] != self.n:
                raise ValueError("%s has %d rows, this actor was built for %d states" % (name, t_.shape[0], self.n))

    def features(self, step, cur_ego4, k, cur_ox, cur_ov, cur_oa, stream=None):
        torch = self.torch
        self._check_rows(cur_ego4, k, cur_ox, cur_ov, cur_oa)
        stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
        self.ctx.policy_features_device(self.fcfg, self.n, cur_ox.shape[1], step, cur_ego4.data_ptr(), k.data_ptr(), cur_ox.data_ptr(), cur_ov.data_ptr(),
                                        cur_oa.data_ptr() if cur_oa is not None else 0, self.evals.data_ptr(), self.feat.data_ptr(), self.flen, stream)
        return self.feat

    def forward(self, feat):
        torch = self.torch
        x = feat.to(self.dtype)
        h = torch.relu(torch.addmm(self.b0, x, self.w0t))
        h = torch.relu(torch.addmm(self.b1, h, self.w1t))
        return torch.tanh(torch.addmm(self.b2, h, self.w2t)).squeeze(1) * self.scale + self.mean

    def __call__(self, step, cur_ego4, k, cur_ox, cur_ov, cur_oa):
        if self.engine == "hip":
            self._check_rows(cur_ego4, k, cur_ox, cur_ov, cur_oa)
            stream = self.torch.cuda.current_stream().cuda_stream
            self.ctx.actor_eval_device(self.handle, self.fcfg, self.n, cur_ox.shape[1], step, cur_ego4.data_ptr(), k.data_ptr(), cur_ox.data_ptr(),
                                       cur_ov.data_ptr(), cur_oa.data_ptr() if cur_oa is not None else 0, self.evals.data_ptr(),
                                       self.feat.data_ptr() if self.keep_features else 0, self.flen, self.jerk.data_ptr(), stream)
            return self.jerk
        with self.torch.no_grad():
            feat = self.features(step, cur_ego4, k, cur_ox, cur_ov, cur_oa)
            return self.forward(feat).to(self.torch.float64)


def forward_host(w, feat, dtype=np.float32):
    """The network on the host in numpy (CPU suite): feat [n, 21] -> jerk [n]."""
    x = np.asarray(feat, dtype=dtype)
    h = np.maximum(x @ w["w0"].T.astype(dtype) + w["b0"].astype(dtype), 0)
    h = np.maximum(h @ w["w1"].T.astype(dtype) + w["b1"].astype(dtype), 0)
    y = h @ w["w2"].T.astype(dtype) + w["b2"].astype(dtype)
    return np.tanh(y[:, 0]) * dtype(w["tanh_scale"]) + dtype(w["tanh_mean"])
