#!/usr/bin/env python3
"""Golden vectors for the reference's two non-production solvers (st_cy.solve_s_t_path_no_jerk_fast / _djikstra,
st_cy.pyx:96-312), from the reference's own compiled st_cy on small grids (the (t, s, s_prev) variant needs S^2 memory).
Build-container only; numbers only.  Re-run: python tests/golden/make_golden_nojerk.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference      # noqa: E402


def main():
    S, control, prediction, st, st_cy = import_reference()
    rng = np.random.default_rng(777)
    cases = {}
    c = 0
    while c < 48:
        H = int(rng.integers(3, 14))
        ds = float(rng.choice([0.25, 0.5, 1.0]))
        dt = float(rng.choice([0.3, 0.5]))
        Sn = int(rng.integers(120, 200))
        s0 = float(rng.uniform(-20, 20)) if c % 3 else 0.0
        s_values = s0 + ds * np.arange(Sn)
        t_values = dt * np.arange(H)
        mode = c % 4
        obstacles = rng.random((H, Sn)) < (0.0 if mode == 0 else 0.12)
        if mode == 2:
            for t in range(2, H):
                if rng.random() < 0.35:
                    obstacles[t, :] = True                       # walls -> failures
        if mode == 3:
            distances = np.full((H, Sn), 9.0)                    # constant penalty -> cost ties between symmetric choices
        else:
            distances = rng.uniform(0.0, 60.0, (H, Sn))
            distances[rng.random((H, Sn)) < 0.1] = 0.0
        v0 = float(rng.uniform(0, min(30.0, 0.8 * (Sn * ds) / max((H - 1) * dt, 1e-9))))
        try:
            fast = np.asarray(st_cy.solve_s_t_path_no_jerk_fast(obstacles, s_values, t_values, v0, distances))
            dj = np.asarray(st_cy.solve_s_t_path_no_jerk_djikstra(obstacles, s_values, t_values, v0, distances))
        except IndexError:
            continue
        cases["c%d_obstacles" % c] = obstacles; cases["c%d_distances" % c] = distances
        cases["c%d_s_values" % c] = s_values; cases["c%d_t_values" % c] = t_values; cases["c%d_v0" % c] = np.array(v0)
        cases["c%d_fast" % c] = fast; cases["c%d_djikstra" % c] = dj
        c += 1
    cases["n_cases"] = np.array(c)
    np.savez_compressed(os.path.join(HERE, "golden_nojerk.npz"), **cases)
    fails = sum(1 for i in range(c) if cases["c%d_fast" % i][-1] == 0)
    differ = sum(1 for i in range(c) if not np.array_equal(cases["c%d_fast" % i], cases["c%d_djikstra" % i]))
    print("nojerk: %d cases, %d with truncated paths, fast != djikstra in %d" % (c, fails, differ))


if __name__ == "__main__":
    main()
