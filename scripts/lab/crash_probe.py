"""Where do the batched episodes crash?  Final ego states of crashed environments (analysis tool)."""
import sys; sys.path.insert(0, '.')
import numpy as np
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import episodes, _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
REF = {2.4: "ref st_low: time 25.66 speed 10.42 jerk 1.074 closest 10.11 disruption mean 0.113 max 3.22 total 2.25 time 0.47",
       1.8: "ref st_medium: time 28.64 speed 9.30 jerk 1.262 closest 10.27 disruption mean 0.305 max 6.49 total 6.95 time 1.40",
       1.2: "ref st_default: time 29.84 speed 8.92 jerk 1.105 closest 10.15 disruption mean 0.288 max 6.64 total 6.90 time 1.36"}
for interval in (2.4, 1.8, 1.2):
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(dict(BASE_TRAFFIC_INTERVAL=interval, OTHER_CAR_SPEED=7.0))
    ctx = _capi.default_context()
    st = episodes.run_episodes(n, seed=3, controller="st", ctx=ctx)
    status, ticks, acc, ego4 = ctx.sim_read(n)
    cr = np.nonzero(status == 2)[0]
    print("interval", interval, "crashed", cr.size, "of", n, {k: round(float(np.nanmean(v)), 3) for k, v in st.items() if k not in ("ticks", "status", "ego4")})
    print("   ", REF[interval])
    for i in cr[:20]:
        print("   env %4d tick %3d  x %.2f y %.2f v %.2f a %.2f  mean_speed %.2f closest %.2f" % (i, ticks[i], ego4[i, 0], ego4[i, 1], ego4[i, 2], ego4[i, 3], st["mean_speed"][i], st["closest_distance"][i]))
