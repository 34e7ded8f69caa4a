#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference, Cython, gcc).  It
  * compiles the reference's st_cy.pyx, unmodified, into a temporary directory,
  * imports the reference's config / control / prediction / st modules from /root/reference
    (with inert stand-ins for the two modules the image lacks and the hot path never calls:
    `traci` (SUMO RPC) and `cvxopt` (used only by st.finer_fit)),
  * feeds them seeded synthetic states from this repo's generator, and
  * stores inputs + the reference's outputs as .npz data files.

Nothing of the reference (source, bytecode, binaries) is written into the repo: the outputs
are numbers only.  Re-run:  python tests/golden/make_golden.py
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def import_reference():
    tmp = tempfile.mkdtemp(prefix="stcy_ref_")
    shutil.copy(os.path.join(REF, "st_cy.pyx"), tmp)
    shutil.copy(os.path.join(REF, "setup.py"), tmp)
    subprocess.run([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=tmp, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    os.environ.setdefault("SUMO_HOME", tmp)
    os.environ["MPLBACKEND"] = "Agg"
    traci = types.ModuleType("traci")          # never called on the path
    cvxopt = types.ModuleType("cvxopt")
    cvxopt.solvers = types.SimpleNamespace(options={})
    cvxopt.matrix = lambda x: x
    sys.modules["traci"] = traci
    sys.modules["cvxopt"] = cvxopt
    sys.path.insert(0, REF)
    sys.path.insert(0, tmp)
    import config  # noqa
    import st_cy  # noqa
    import control  # noqa
    import prediction  # noqa
    import st  # noqa
    return config.Settings, control, prediction, st, st_cy


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def replay_cost(s_seq, best_t, v0, a0, dt, distances, s_values, S):
    """Accumulated cost along the returned path in st_cy's operation order (st_cy.pyx:46-50, 388)."""
    s1 = s_values[0] - v0 * dt
    s2 = s1 - dt * (v0 - a0 * dt)
    s0 = s_values[0]
    total = 0.0
    idx = [int(np.where(s_values == s_seq[t])[0][0]) for t in range(best_t + 1)]
    for t in range(1, best_t + 1):
        s = s_seq[t]
        d = distances[t, idx[t]]
        v = (s - s0) / dt
        a = (s - 2 * s0 + s1) / (dt * dt)
        j = (s - 3 * s0 + 3 * s1 - s2) / (dt ** 3)
        if d < S.MIN_ALLOWED_DISTANCE:
            pen = 1000000.0 / max(d, 1.0)
        else:
            pen = 1 / d
        c = S.V_WEIGHT * ((v - S.DESIRED_SPEED) * (v - S.DESIRED_SPEED)) + S.A_WEIGHT * (a * a) + S.J_WEIGHT * (j * j) + S.D_WEIGHT * pen
        total = total + c
        s2, s1, s0 = s1, s0, s
    return total, idx


def run_states(refmods, overrides, ego, k_count, ox, ov, n_full_grids=0, rng=None):
    S, control, prediction, st, st_cy = refmods
    for k_, v_ in overrides.items():
        setattr(S, k_, v_)
    n = ego.shape[0]
    out = {}
    H = None
    rows = []
    grids = []
    for i in range(n):
        k = int(k_count[i])
        state = prediction.HighwayState((float(ego[i, 0]), float(ego[i, 1])), float(ego[i, 2]), float(ego[i, 3]),
                                        [float(x) for x in ox[i, :k]], [float(x) for x in ov[i, :k]], [0.0] * k)
        start_s = control.get_ego_s(state.ego_position)
        s_seq, obstacles, s_values, t_values, distances = st.get_appropriate_base_st_path_and_obstacles(state)
        H = len(t_values)
        best_t = H - 1
        while best_t > 0 and s_seq[best_t] == 0:
            best_t -= 1
        dt = t_values[1] - t_values[0]
        cost, idx = replay_cost(s_seq, best_t, state.ego_speed, state.ego_acceleration, dt, distances, s_values, S)
        path_idx = np.full(H, -1, dtype=np.int32)
        path_idx[:best_t + 1] = idx
        crash = st.test_guaranteed_crash_from_state(state)
        pdist = np.full(H, np.nan)
        for t in range(best_t + 1):
            qi = st.get_range_index(s_values[0], s_values[1] - s_values[0], s_seq[t])
            pdist[t] = distances[t, qi]
        # predicted traffic per layer (st.py:42-43)
        tab = np.full((H, ox.shape[1]), np.nan)
        ps = state
        for t in range(H):
            if t != 0:
                ps, _ = ps.predict_step_without_ego(dt)
            tab[t, :k] = ps.other_xs
        rows.append(dict(start_s=start_s, num_s=len(s_values), s_seq=s_seq, best_t=best_t, cost=cost,
                         path_idx=path_idx, crash=int(crash), pdist=pdist, tab=tab,
                         ob_sha=sha(obstacles), di_sha=sha(distances), sv_sha=sha(s_values)))
        if i < n_full_grids:
            grids.append((np.packbits(obstacles, axis=None), distances.copy(), s_values.copy(), t_values.copy()))
    out["ego"] = ego.copy()
    out["ego"][:, 4] = [r["start_s"] for r in rows]
    out["k_count"] = k_count
    out["other_x"] = ox
    out["other_v"] = ov
    out["num_s"] = np.array([r["num_s"] for r in rows], dtype=np.int32)
    out["s_sequence"] = np.stack([r["s_seq"] for r in rows])
    out["best_t"] = np.array([r["best_t"] for r in rows], dtype=np.int32)
    out["cost"] = np.array([r["cost"] for r in rows])
    out["path_idx"] = np.stack([r["path_idx"] for r in rows])
    out["crash"] = np.array([r["crash"] for r in rows], dtype=np.int32)
    out["path_dist"] = np.stack([r["pdist"] for r in rows])
    out["pred_x"] = np.stack([r["tab"] for r in rows])
    out["obstacles_sha256"] = np.stack([r["ob_sha"] for r in rows])
    out["distances_sha256"] = np.stack([r["di_sha"] for r in rows])
    out["s_values_sha256"] = np.stack([r["sv_sha"] for r in rows])
    out["t_values"] = np.asarray(t_values)
    for gi, (obp, di, sv, tv) in enumerate(grids):
        out["grid%d_obstacles_packed" % gi] = obp
        out["grid%d_distances" % gi] = di
        out["grid%d_s_values" % gi] = sv
    out["n_full_grids"] = np.array(len(grids))
    keys = sorted(overrides)
    out["override_keys"] = np.array(keys)
    out["override_vals"] = np.array([float(overrides[k_]) for k_ in keys])
    return out


def run_predictor_steps(refmods, ego, k_count, ox, ov, rng):
    """One-step predictor goldens: with_ego at a random speed / min distance, and without_ego."""
    S, control, prediction, st, st_cy = refmods
    n = ego.shape[0]
    K = ox.shape[1]
    sel = rng.uniform(0.0, 30.0, n)
    mcd = rng.choice([5.0, 5.1, 3.0, 7.5], n)
    dts = rng.choice([0.2, 0.3], n)
    w_ego = np.zeros((n, 4)); w_x = np.zeros((n, K)); w_v = np.zeros((n, K)); w_cr = np.zeros(n, np.int32)
    wo_ego = np.zeros((n, 4)); wo_x = np.zeros((n, K)); wo_v = np.zeros((n, K)); wo_cr = np.zeros(n, np.int32)
    for i in range(n):
        k = int(k_count[i])
        mk = lambda: prediction.HighwayState((float(ego[i, 0]), float(ego[i, 1])), float(ego[i, 2]), float(ego[i, 3]),
                                             [float(x) for x in ox[i, :k]], [float(x) for x in ov[i, :k]], [0.0] * k)
        s1, c1 = mk().predict_step_with_ego(float(sel[i]), float(dts[i]), float(mcd[i]))
        w_ego[i] = (s1.ego_position[0], s1.ego_position[1], s1.ego_speed, s1.ego_acceleration)
        w_x[i, :k] = s1.other_xs; w_v[i, :k] = s1.other_speeds; w_cr[i] = int(c1)
        s2, c2 = mk().predict_step_without_ego(float(dts[i]), float(mcd[i]))
        wo_ego[i] = (s2.ego_position[0], s2.ego_position[1], s2.ego_speed, s2.ego_acceleration)
        wo_x[i, :k] = s2.other_xs; wo_v[i, :k] = s2.other_speeds; wo_cr[i] = int(c2)
    return dict(ego=ego, k_count=k_count, other_x=ox, other_v=ov, sel=sel, mcd=mcd, dt=dts,
                with_ego=w_ego, with_x=w_x, with_v=w_v, with_crash=w_cr,
                without_ego=wo_ego, without_x=wo_x, without_v=wo_v, without_crash=wo_cr)


def run_raw_grids(refmods, rng, n_cases=40):
    """st_cy.solve_s_t_path_fast on arbitrary (non-state) grids, incl. exact-tie constructions."""
    S, control, prediction, st, st_cy = refmods
    cases = {}
    for c in range(n_cases):
        H = int(rng.integers(2, 12))
        Sn = int(rng.integers(2, 400))
        ds = float(rng.choice([0.05, 0.1, 0.25, 0.5]))
        dt = float(rng.choice([0.2, 0.3, 0.5]))
        s0 = float(rng.uniform(-50, 50)) if c % 3 else 0.0
        s_values = np.arange(s0, s0 + Sn * ds - 1e-9, ds)[:Sn]
        if s_values.size < 2:
            s_values = np.array([s0, s0 + ds])
        Sn = s_values.size
        t_values = np.arange(0, H * dt - 1e-9, dt)[:H]
        H = t_values.size
        mode = c % 4
        obstacles = rng.random((H, Sn)) < (0.0 if mode == 0 else 0.15)
        if mode == 2:   # walls: whole bands blocked -> failures
            for t in range(1, H):
                if rng.random() < 0.3:
                    obstacles[t, :] = True
        if mode == 3:   # constant distances and zero weights -> many exact cost ties
            distances = np.full((H, Sn), 4.0)
        else:
            distances = rng.uniform(0.0, 60.0, (H, Sn))
            distances[rng.random((H, Sn)) < 0.1] = 0.0
        v0 = float(rng.uniform(0, 25)); a0 = float(rng.uniform(-3, 3))
        if mode == 3:
            tun = (1.0, 0.0, 0.0, 0.0, 30.0, 30.0, -6.0, 4.5, -5.0, 5.0, 5.0)
        else:
            tun = (float(rng.choice([10.0, 1.0, 1000.0])), 0.5, float(rng.choice([10.0, 1.0])), float(rng.choice([10.0, 0.0])),
                   float(rng.choice([30.0, 15.0])), float(rng.choice([30.0, 40.0])), -6.0, float(rng.choice([4.5, 5.2])),
                   float(rng.choice([-5.0, -35.0])), float(rng.choice([5.0, 35.0])), float(rng.choice([5.0, 7.5])))
        seq = st_cy.solve_s_t_path_fast(obstacles, s_values, t_values, v0, a0, distances, *tun)
        if c < 12:   # python twin agrees where its cost order coincides is not guaranteed; record only st_cy
            pass
        cases["c%d_obstacles" % c] = obstacles
        cases["c%d_distances" % c] = distances
        cases["c%d_s_values" % c] = s_values
        cases["c%d_t_values" % c] = t_values
        cases["c%d_v0a0" % c] = np.array([v0, a0])
        cases["c%d_tunables" % c] = np.array(tun)
        cases["c%d_s_sequence" % c] = np.asarray(seq)
    cases["n_cases"] = np.array(n_cases)
    return cases


def main():
    sys.path.insert(0, REPO)
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import synth
    refmods = import_reference()
    S = refmods[0]
    S.load_from_file(os.path.join(REF, "configs", "st_low.json"))     # BASELINE config 1
    default_over = dict(pkg.REFERENCE_DEFAULT)
    rng = np.random.default_rng(2024)

    # 1) default parameters: 320 states, K in 0..8 (Kmax 8), incl. failure cases
    ego, k, ox, ov = synth.generate_states(200, k=6, kmax=8, seed=11)
    ego2, k2, ox2, ov2 = synth.generate_states(120, k=8, kmax=8, seed=12, vary_k=True, blocked_quota=0.15)
    ego = np.concatenate([ego, ego2]); k = np.concatenate([k, k2]); ox = np.concatenate([ox, ox2]); ov = np.concatenate([ov, ov2])
    g = run_states(refmods, default_over, ego, k, ox, ov, n_full_grids=3)
    np.savez_compressed(os.path.join(HERE, "golden_default.npz"), **g)
    print("default: %d states, %d failures, %d crash_guaranteed" % (len(k), int((g["best_t"] < g["t_values"].size - 1).sum()), int(g["crash"].sum())))

    # 2) predictor one-step goldens on the same states
    pr = run_predictor_steps(refmods, g["ego"], k, ox, ov, rng)
    np.savez_compressed(os.path.join(HERE, "golden_predictor.npz"), **pr)

    # 3) synthetic H=40 / A=21 mapping: 6 states (the Cython reference needs up to ~1 s each)
    ego3, k3, ox3, ov3 = synth.generate_states(6, k=6, kmax=8, seed=13)
    g3 = run_states(refmods, dict(pkg.SYNTHETIC_H40A21), ego3, k3, ox3, ov3, n_full_grids=0)
    np.savez_compressed(os.path.join(HERE, "golden_h40a21.npz"), **g3)
    print("h40a21: H=%d S=%d failures %d" % (g3["t_values"].size, int(g3["num_s"][0]), int((g3["best_t"] < g3["t_values"].size - 1).sum())))

    # 4) uncertainty > 0 and other non-default grid parameters (exercise st.py:40-41,61-62)
    over4 = dict(default_over)
    over4.update(START_UNCERTAINTY=0.5, UNCERTAINTY_PER_SECOND=0.4, CRASH_MIN_S=12, S_DISCRETIZATION=0.1, FUTURE_S=120.0,
                 T_DISCRETIZATION=0.2, FUTURE_T=4.0, MIN_ALLOWED_DISTANCE=7.5)
    ego4, k4, ox4, ov4 = synth.generate_states(48, k=7, kmax=8, seed=14, vary_k=True, dt=0.2)
    g4 = run_states(refmods, over4, ego4, k4, ox4, ov4, n_full_grids=1)
    np.savez_compressed(os.path.join(HERE, "golden_uncertainty.npz"), **g4)
    print("uncertainty: H=%d S=%d failures %d" % (g4["t_values"].size, int(g4["num_s"][0]), int((g4["best_t"] < g4["t_values"].size - 1).sum())))
    for k_, v_ in default_over.items():
        setattr(S, k_, v_)

    # 5) raw-grid solver cases
    rg = run_raw_grids(refmods, rng)
    np.savez_compressed(os.path.join(HERE, "golden_rawgrid.npz"), **rg)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
