#!/usr/bin/env python3
"""Golden values of the reference's st.get_path_mean_abs_jerk (st.py:274-288) for the paths already held by the state goldens.

Build container only (imports the reference through make_golden.import_reference).  Output: golden_jerk.npz with, per case, the
golden file and row the path comes from, the trimmed path length, v0, a0, dt and the reference's return value.
Re-run:  python tests/golden/make_golden_jerk.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402

FILES = ["golden_default.npz", "golden_uncertainty.npz", "golden_h40a21.npz"]


def main():
    S, control, prediction, st, st_cy = import_reference()
    rows = {k: [] for k in ("file_index", "row", "length", "v0", "a0", "dt", "mean_abs_jerk")}
    for fi, f in enumerate(FILES):
        g = np.load(os.path.join(HERE, f), allow_pickle=False)
        dt = float(g["t_values"][1] - g["t_values"][0])
        for i in range(g["ego"].shape[0]):
            bt = int(g["best_t"][i])
            if bt < 1:
                continue                                  # a single point: the reference divides by len - 1 = 0
            seq = [float(x) for x in g["s_sequence"][i, :bt + 1]]
            for dt_ in (dt, 0.2):                         # the planning step and the simulator tick (dqn.py:172-173 uses the tick)
                val = st.get_path_mean_abs_jerk(seq, float(g["ego"][i, 2]), float(g["ego"][i, 3]), dt_)
                for k, v in zip(rows, (fi, i, bt + 1, g["ego"][i, 2], g["ego"][i, 3], dt_, val)):
                    rows[k].append(v)
    out = {k: np.asarray(v, dtype=(np.int32 if k in ("file_index", "row", "length") else np.float64)) for k, v in rows.items()}
    out["files"] = np.array(FILES)
    np.savez_compressed(os.path.join(HERE, "golden_jerk.npz"), **out)
    print("golden_jerk.npz: %d cases" % len(rows["row"]))


if __name__ == "__main__":
    main()
