#!/usr/bin/env python3
"""More reference-held vectors at the benchmark lattice (H = 40, fan-out 21) and at the predictor's thresholds.

Same rules as make_golden.py (build container only; the reference is imported from /root/reference, its st_cy.pyx
compiled unmodified in a temporary directory; only numbers are stored):
  golden_h40a21.npz      72 states through the real st_cy at H=40/S=7201: the 6 states of the first edition, then
                         K in {0, 3, 6, 8}, a raised quota of blocked starts, >= 8 failure cases
  golden_h40a21_unc.npz  8 states at H=40 with START_UNCERTAINTY / UNCERTAINTY_PER_SECOND > 0
  golden_thresholds.npz  one-step predictor cases whose predicted ego lands within 1e-9 (relative) of the reaction (8)
                         and crash (11) thresholds of prediction.py:64-66, on both sides
Re-run:  python tests/golden/make_golden_h40.py
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import make_golden as mg  # noqa: E402


def threshold_cases(refmods, rng, n_per=24):
    S, control, prediction, st, st_cy = refmods
    rows = []
    for thr in (8.0, 11.0):
        made = 0
        while made < n_per:
            # ego on the ramp-to-lane segment (x < 1.5), heading to merge_point2; choose the step so that the predicted
            # position has arclength coordinate thr * (1 + eps)
            cx = rng.uniform(-50.0, -44.0)
            cy = 1.72 + (cx + 50.9) / 52.4 * (-1.6 - 1.72)
            dt = float(rng.choice([0.2, 0.3]))
            eps = float(rng.choice([-1e-9, 1e-9, -3e-10, 3e-10, -1e-12, 1e-12]))
            target = thr * (1.0 + eps)
            lo, hi = 0.0, 40.0
            k = int(rng.integers(1, 7))
            xs = sorted([float(cx + rng.uniform(-25, 25)) for _ in range(k)], reverse=True)
            vs = [float(rng.choice([7.0, 11.0, 15.0])) for _ in range(k)]
            mk = lambda: prediction.HighwayState((cx, cy), 10.0, 0.0, list(xs), list(vs), [0.0] * k)
            for _ in range(200):
                mid = 0.5 * (lo + hi)
                s1, _c = mk().predict_step_with_ego(mid, dt, 5.0)
                es = control.get_ego_s(s1.ego_position)
                if es < target: lo = mid
                else: hi = mid
            sel = hi if eps > 0 else lo
            s1, c1 = mk().predict_step_with_ego(sel, dt, 5.0)
            es = control.get_ego_s(s1.ego_position)
            if not (abs(es - thr) <= 2e-9 * thr and (es > thr) == (eps > 0)):
                continue
            row = dict(ego=(cx, cy, 10.0, 0.0), k=k, xs=xs + [0.0] * (8 - k), vs=vs + [0.0] * (8 - k), sel=sel, dt=dt, mcd=5.0,
                       out_ego=(s1.ego_position[0], s1.ego_position[1], s1.ego_speed, s1.ego_acceleration),
                       out_x=list(s1.other_xs) + [0.0] * (8 - k), out_v=list(s1.other_speeds) + [0.0] * (8 - k), crash=int(c1), es=es, thr=thr)
            rows.append(row)
            made += 1
    return dict(ego=np.array([r["ego"] for r in rows]), k_count=np.array([r["k"] for r in rows], dtype=np.int32),
                other_x=np.array([r["xs"] for r in rows]), other_v=np.array([r["vs"] for r in rows]),
                sel=np.array([r["sel"] for r in rows]), dt=np.array([r["dt"] for r in rows]), mcd=np.array([r["mcd"] for r in rows]),
                with_ego=np.array([r["out_ego"] for r in rows]), with_x=np.array([r["out_x"] for r in rows]),
                with_v=np.array([r["out_v"] for r in rows]), with_crash=np.array([r["crash"] for r in rows], dtype=np.int32),
                ego_s=np.array([r["es"] for r in rows]), threshold=np.array([r["thr"] for r in rows]))


def main():
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import synth
    refmods = mg.import_reference()
    S = refmods[0]
    S.load_from_file(os.path.join(mg.REF, "configs", "st_low.json"))
    rng = np.random.default_rng(4040)
    parts = [synth.generate_states(6, k=6, kmax=8, seed=13)]                       # the first edition's states, unchanged
    for kk, seed, quota in ((0, 131, 0.05), (3, 132, 0.2), (6, 133, 0.3), (8, 134, 0.3)):
        parts.append(synth.generate_states(16 if kk else 6, k=kk, kmax=8, seed=seed, blocked_quota=quota))
    parts.append(synth.generate_states(12, k=8, kmax=8, seed=135, vary_k=True, blocked_quota=0.5))
    ego = np.concatenate([p[0] for p in parts]); k = np.concatenate([p[1] for p in parts])
    ox = np.concatenate([p[2] for p in parts]); ov = np.concatenate([p[3] for p in parts])
    over = dict(pkg.REFERENCE_DEFAULT); over.update(pkg.SYNTHETIC_H40A21)
    g = mg.run_states(refmods, over, ego, k, ox, ov, n_full_grids=0)
    np.savez_compressed(os.path.join(HERE, "golden_h40a21.npz"), **g)
    H = g["t_values"].size
    print("h40a21: %d states, H=%d S=%d, failures %d, crash %d, K values %s" % (len(k), H, int(g["num_s"][0]), int((g["best_t"] < H - 1).sum()), int(g["crash"].sum()), sorted(set(k.tolist()))))

    over2 = dict(over); over2.update(START_UNCERTAINTY=0.4, UNCERTAINTY_PER_SECOND=0.25)
    e2, k2, ox2, ov2 = synth.generate_states(8, k=6, kmax=8, seed=136, blocked_quota=0.25)
    g2 = mg.run_states(refmods, over2, e2, k2, ox2, ov2, n_full_grids=0)
    np.savez_compressed(os.path.join(HERE, "golden_h40a21_unc.npz"), **g2)
    print("h40a21_unc: 8 states, failures %d" % int((g2["best_t"] < H - 1).sum()))
    for k_, v_ in pkg.REFERENCE_DEFAULT.items():
        setattr(S, k_, v_)
    setattr(S, "START_UNCERTAINTY", 0.0); setattr(S, "UNCERTAINTY_PER_SECOND", 0.0)

    th = threshold_cases(refmods, rng)
    np.savez_compressed(os.path.join(HERE, "golden_thresholds.npz"), **th)
    print("thresholds: %d cases; |es-thr|/thr max %.2e; above %d below %d" % (len(th["sel"]), np.max(np.abs(th["ego_s"] - th["threshold"]) / th["threshold"]),
                                                                               int((th["ego_s"] > th["threshold"]).sum()), int((th["ego_s"] <= th["threshold"]).sum())))


if __name__ == "__main__":
    main()
