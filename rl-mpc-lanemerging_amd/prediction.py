"""``HighwayState`` -- the value type the ST path consumes (reference ``prediction.py:9-20``).

Same constructor and attribute names as the reference.  The two predictor methods
(``prediction.py:22-105``) run on the GPU through ``stmpc_predict_batch``; inside the
batched solver the same device code runs fused with the lattice build, so these methods
exist for callers that use the predictor on its own (the combined controller's rollouts,
``dqn.py:129-141``).  ``from_sumo`` (TraCI) is out of scope.
"""
import numpy as np

from . import _capi
from . import control
from .config import Settings


class HighwayState:

    ego_reaction_threshold = 8
    ego_crash_threshold = 11

    def __init__(self, ego_position, ego_speed, ego_acceleration, other_xs, other_speeds, other_accelerations):
        self.ego_position = ego_position
        self.ego_speed = ego_speed
        self.ego_acceleration = ego_acceleration
        self.other_xs = other_xs
        self.other_speeds = other_speeds
        self.other_accelerations = other_accelerations

    # -- GPU predictor -----------------------------------------------------------------------
    def _step(self, mode, selected_speed, delta_t, min_crash_distance):
        params = _capi.Params.from_settings(Settings)
        params.react_thr = float(self.ego_reaction_threshold)
        params.crash_thr = float(self.ego_crash_threshold)
        k = len(self.other_xs)
        ego4 = np.array([[self.ego_position[0], self.ego_position[1], self.ego_speed, self.ego_acceleration]],
                        dtype=np.float64)
        ox = np.asarray(self.other_xs, dtype=np.float64).reshape(1, k)
        ov = np.asarray(self.other_speeds, dtype=np.float64).reshape(1, k)
        sel = np.array([selected_speed], dtype=np.float64)
        eo, xo, vo, cr, ao = _capi.default_context().predict_batch(params, mode, ego4, np.array([k], np.int32), ox, ov, sel,
                                                                   delta_t, min_crash_distance, want_acc=True)
        # new_other_accelerations (prediction.py:86-89,97): the deceleration applied to a following vehicle, else 0 -- the
        # RL policy's state vector reads them (dqn.get_state_vector_from_base_state, dqn.py:400), so they come from the predictor itself
        st = HighwayState((float(eo[0, 0]), float(eo[0, 1])), float(eo[0, 2]), float(eo[0, 3]),
                          [float(x) for x in xo[0]], [float(v) for v in vo[0]], [float(a) for a in ao[0]])
        return st, bool(cr[0])

    def predict_step_without_ego(self, delta_t, min_crash_distance=5):
        """prediction.py:22-44."""
        return self._step(1, 0.0, delta_t, min_crash_distance)

    def predict_step_with_ego(self, selected_speed, delta_t, min_crash_distance=5):
        """prediction.py:46-105."""
        return self._step(0, selected_speed, delta_t, min_crash_distance)

    @classmethod
    def empty_state(cls):
        return cls(0, 0, 0, [], [], [])

    def get_closest_cars(self):
        """Nearest vehicle ahead of and behind the ego as ``(x, speed, acceleration)`` or ``None`` (same result as
        the reference's ``prediction.py:162-182``; vehicles are ordered front to back)."""
        ego_x = self.ego_position[0]
        triple = lambda i: (self.other_xs[i], self.other_speeds[i], self.other_accelerations[i])
        behind = next((i for i, x in enumerate(self.other_xs) if x < ego_x), None)
        n_front = behind if behind is not None else len(self.other_xs)
        return (triple(n_front - 1) if n_front > 0 else None), (triple(behind) if behind is not None else None)


def pack_states(states, kmax=None):
    """Lay a list of ``HighwayState`` out as the SoA batch the C-ABI takes.

    Returns ``ego[N,5]`` (x, y, v, a, start_s), ``k_count[N]``, ``other_x[N,K]``, ``other_v[N,K]``.
    ``start_s`` is ``control.get_ego_s`` evaluated on the host exactly as the reference does
    (st.py:28): CPython's ``**`` is a libm ``pow`` call and is not always ``x*x``.
    """
    n = len(states)
    k_count = np.array([len(s.other_xs) for s in states], dtype=np.int32)
    K = int(k_count.max()) if n else 0
    if kmax is not None:
        if K > kmax:
            raise ValueError("a state has more vehicles than kmax")
        K = kmax
    K = max(K, 1)
    ego = np.zeros((n, 5), dtype=np.float64)
    ox = np.zeros((n, K), dtype=np.float64)
    ov = np.zeros((n, K), dtype=np.float64)
    for i, s in enumerate(states):
        ego[i, 0], ego[i, 1] = s.ego_position
        ego[i, 2] = s.ego_speed
        ego[i, 3] = s.ego_acceleration
        ego[i, 4] = control.get_ego_s(s.ego_position)
        k = k_count[i]
        ox[i, :k] = s.other_xs
        ov[i, :k] = s.other_speeds
    return ego, k_count, ox, ov
