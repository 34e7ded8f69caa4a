"""Row f3 with the combined controller and the reference's pretrained actors: episode statistics against the reference's reported rows
(experiment_data/saved_data.csv: combined_low_1 / combined_medium_1 / combined_default_1 / combined_moderate_1, numbers copied as data).
STATISTICAL comparison: the world is a restatement of the SUMO scenario, not SUMO.  usage: combined_episodes.py [n] [out.json]"""
import json
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, actor, combined_bench, episodes

REF = {   # config: (actor, interval, speed, reference row: crashed, merged, mean_speed, max_speed, mean_abs_jerk, closest_distance, time_to_merge, percent_st)
    "combined_low_1": ("low1", 2.4, 7.0, dict(crashed=0.0, merged=1.0, mean_speed=11.408, max_speed=19.497, mean_abs_jerk=0.654, closest_distance=6.106, time_to_merge=23.532, percent_st=0.0323, mean_disruption=0.184, max_disruption=5.244, total_disruption=3.031, disruption_time=2.195)),
    "combined_medium_1": ("medium1", 1.8, 7.0, dict(crashed=0.0, merged=1.0, mean_speed=10.400, max_speed=18.149, mean_abs_jerk=0.809, closest_distance=7.246, time_to_merge=25.904, percent_st=0.0238, mean_disruption=0.387, max_disruption=6.526, total_disruption=7.313, disruption_time=3.117)),
    "combined_default_1": ("default1", 1.2, 7.0, dict(crashed=0.0, merged=1.0, mean_speed=9.482, max_speed=16.643, mean_abs_jerk=0.775, closest_distance=6.330, time_to_merge=28.780, percent_st=0.0349, mean_disruption=0.334, max_disruption=6.615, total_disruption=6.585, disruption_time=3.052)),
    "combined_moderate_1": ("moderate1", 1.2, 11.0, dict(crashed=0.0, merged=1.0, mean_speed=13.814, max_speed=19.521, mean_abs_jerk=0.689, closest_distance=5.883, time_to_merge=20.251, percent_st=0.0374, mean_disruption=0.352, max_disruption=5.878, total_disruption=4.516, disruption_time=1.662)),
}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device("cuda", 0)
    ctx = _capi.default_context()
    report = {}
    for name, (act, interval, speed, ref) in REF.items():
        pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
        pkg.apply_overrides(combined_bench.COMBINED_MEDIUM_1)           # the combined_*_1 configs share these flags
        pkg.apply_overrides(dict(BASE_TRAFFIC_INTERVAL=interval, OTHER_CAR_SPEED=speed))
        row = {}
        for label, kw in (("", {}), ("_no_time_feature", {"time_feature": False})):
            pol = actor.DDPGActor(act, n, ctx, pkg.Settings, dev, **kw)
            st = episodes.run_episodes(n, seed=21, controller="combined", policy=pol, ctx=ctx, kmax=16)
            s = episodes.summary(st)
            row["here" + label] = {k: round(s[k], 4) for k in ref if k in s}
        row["reference"] = ref
        report[name] = row
        print(name, json.dumps(row), flush=True)
    if out_path:
        json.dump(report, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
