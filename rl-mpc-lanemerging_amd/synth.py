"""Seeded synthetic merge states for the suite and ``bench.py``.

Not taken from the reference (it has no generator: states come from SUMO); the
distributions follow the reference's scenario constants -- ramp/highway geometry
(control.py:366-370, merge.net.xml:38-52), start speed N(15,5) clipped to [5,25]
(config.py:49-54), traffic speed 7/11/15 m/s at 1.2/1.8/2.4 s headway (configs/*.json).
"""
import numpy as np

from . import control


def road_y(x):
    """y of the ego lane centre line: ramp -> merge zone -> highway lane at y = -1.6."""
    x = np.asarray(x, dtype=np.float64)
    ramp = 1.72 + 0.134 * (-50.9 - x)
    zone = 1.72 + (x + 50.9) / 52.4 * (-1.6 - 1.72)
    return np.where(x < -50.9, ramp, np.where(x < 1.5, zone, -1.6))


def generate_states(n, k=6, kmax=None, seed=0, dt=0.3, blocked_quota=0.05, vary_k=False):
    """Return ``ego[n,5]`` (x, y, v, a, start_s), ``k_count[n]``, ``other_x[n,K]``, ``other_v[n,K]``.

    Vehicles are ordered front->back (descending x) as ``HighwayState.from_sumo`` orders them
    (prediction.py:138-141).  ``vary_k`` draws k uniformly from 0..k per state.
    """
    rng = np.random.default_rng(seed)
    K = kmax or max(k, 1)
    ego = np.zeros((n, 5))
    k_count = np.zeros(n, dtype=np.int32)
    ox = np.zeros((n, K))
    ov = np.zeros((n, K))
    blocked_left = int(blocked_quota * n)
    i = 0
    while i < n:
        ex = rng.uniform(-250.0, 60.0)
        ey = float(road_y(ex))
        if rng.random() < 0.25:
            ev = rng.uniform(0.0, 30.0)
        else:
            ev = float(np.clip(rng.normal(15.0, 5.0), 5.0, 25.0))
        ea = rng.uniform(-6.0, 4.5)
        if ev - ea * dt < 0.0:
            ea = ev / dt
        kk = int(rng.integers(0, k + 1)) if vary_k else k
        v_car = float(rng.choice([7.0, 11.0, 15.0]))
        interval = float(rng.choice([1.2, 1.8, 2.4]))
        xs = np.zeros(kk)
        vs = np.zeros(kk)
        x = ex + rng.uniform(-20.0, 60.0)
        for c in range(kk):
            xs[c] = x
            vs[c] = v_car - (rng.uniform(0.0, 4.0) if rng.random() < 0.2 else 0.0)
            x -= v_car * (interval + rng.uniform(0.0, 1.0))
        start_s = control.get_ego_s((ex, ey))
        # ego already inside a vehicle's blocked window at t = 0 (st.py:60-65)?
        o = xs + 51.0
        inside = bool(np.any((o >= 15.0) & (np.abs(o - start_s) < 5.0)))
        if inside:
            if blocked_left <= 0:
                continue
            blocked_left -= 1
        ego[i] = (ex, ey, ev, ea, start_s)
        k_count[i] = kk
        ox[i, :kk] = xs
        ov[i, :kk] = vs
        i += 1
    return ego, k_count, ox, ov
