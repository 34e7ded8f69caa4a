"""Scores row f3's world per junction rule / ego route from profiles/<round>/world_rules.txt (scripts/lab/world_rules.sh): mean absolute relative
deviation from the reference's reported rows over the core columns (mean / max speed, time to merge, mean |jerk|, closest distance) and over the
disruption columns (max, total), for the pure ST rows and the combined-controller rows; the combined controller's ST share (percentage points);
crashes + time-outs summed over those seven rows; and the actor-alone rows' excess collision rate.  usage: world_score.py [file]"""
import ast
import re
import sys

import numpy as np

REFST = {2.4: dict(mean_speed=10.416, max_speed=23.612, time_to_merge=25.659, mean_abs_jerk=1.074, closest_distance=10.110, max_disruption=3.222, total_disruption=2.254),
         1.8: dict(mean_speed=9.297, max_speed=23.296, time_to_merge=28.645, mean_abs_jerk=1.262, closest_distance=10.273, max_disruption=6.490, total_disruption=6.949),
         1.2: dict(mean_speed=8.919, max_speed=23.149, time_to_merge=29.838, mean_abs_jerk=1.105, closest_distance=10.153, max_disruption=6.638, total_disruption=6.902)}
CORE = ("mean_speed", "max_speed", "time_to_merge", "mean_abs_jerk", "closest_distance")
DIS = ("max_disruption", "total_disruption")
txt = open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r5/world_rules.txt").read()
print("%-38s | %-22s | %-38s | %-18s | %s" % ("world", "pure ST: core, disruption", "combined: core, disruption, ST share", "crashes + time-outs", "actor alone: excess collisions"))
for b in re.split(r"^== ", txt, flags=re.M)[1:]:
    st_dev, st_dis, cb_dev, cb_dis, cb_st, ddpg, bad, rows = [], [], [], [], [], [], 0.0, {}
    for line in b.splitlines()[1:]:
        if line.startswith("interval"):
            iv, d = float(line.split()[1]), ast.literal_eval(line[line.index("{"):])
            st_dev += [abs(d[k] / REFST[iv][k] - 1) for k in CORE]; st_dis += [abs(d[k] / REFST[iv][k] - 1) for k in DIS]
            bad += d["crashed"] + d["timed_out"]
        else:
            m = re.match(r"(\S+)\s+(here|reference)\s+(.*)", line)
            if m:
                v = m.group(3).split()
                rows.setdefault(m.group(1), {})[m.group(2)] = {v[i]: float(v[i + 1]) for i in range(0, len(v), 2)}
    for name, r in rows.items():
        h, ref = r["here"], r["reference"]
        if name.startswith("combined"):
            cb_dev += [abs(h[k] / ref[k] - 1) for k in CORE]; cb_dis += [abs(h[k] / ref[k] - 1) for k in DIS]
            cb_st.append(abs(h["percent_st"] - ref["percent_st"])); bad += h["crashed"] + (1 - h["crashed"] - h["merged"])
        else:
            ddpg.append(h["crashed"] - ref["crashed"])
    print("%-38s | %4.1f %%, %3.0f %%            | %4.1f %%, %3.0f %%, %.2f pp                  | %.4f             | %.3f" % (
        b.splitlines()[0], 100 * np.mean(st_dev), 100 * np.mean(st_dis), 100 * np.mean(cb_dev), 100 * np.mean(cb_dis), 100 * np.mean(cb_st), bad, np.mean(ddpg)))
