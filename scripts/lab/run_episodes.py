import sys, time; sys.path.insert(0, '.')
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import episodes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for interval, speed in ((2.4, 7.0), (1.8, 7.0), (1.2, 7.0), (1.2, 11.0)):
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(dict(BASE_TRAFFIC_INTERVAL=interval, OTHER_CAR_SPEED=speed))
    t = time.perf_counter(); st = episodes.run_episodes(n, seed=1, controller="st"); dt = time.perf_counter() - t
    s = episodes.summary(st)
    print("interval %.1f speed %.0f: %s  (%.2f s wall, %d env-ticks, %.0f env-ticks/s)" % (interval, speed, {k: round(v, 3) for k, v in s.items()}, dt, st["ticks"].sum(), st["ticks"].sum() / dt))
