"""Trace of the actor-alone episodes (controller="policy") up to a crash (analysis tool).
usage: policy_trace.py <actor> <interval> <speed> <n> [which crashing env, default 0] [yield_overlap]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import episodes, _capi, actor
act, interval, speed, n = sys.argv[1], float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
which = int(sys.argv[5]) if len(sys.argv) > 5 else 0
yo = int(sys.argv[6]) if len(sys.argv) > 6 else None
pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(dict(BASE_TRAFFIC_INTERVAL=interval, OTHER_CAR_SPEED=speed))
ctx = _capi.default_context(); dev = torch.device("cuda", 0)
pol = actor.DDPGActor(act, n, ctx, pkg.Settings, dev)
r = episodes.EpisodeRunner(n, 21, "policy", pol, ctx, 16)
if yo is not None:
    r.cfg.yield_overlap = yo
    ctx.sim_init(r.cfg, n)
hist = []
for tick in range(r.cfg.max_ticks + 1):
    ctx.sim_view(r.cfg, n, r.kmax, r.d_ego5.data_ptr(), r.d_k.data_ptr(), r.d_ox.data_ptr(), r.d_ov.data_ptr(), r.d_oa.data_ptr())
    torch.cuda.synchronize()
    snap = (r.d_ego5.cpu().numpy().copy(), r.d_k.cpu().numpy().copy(), r.d_ox.cpu().numpy().copy(), r.d_ov.cpu().numpy().copy(), r.d_oa.cpu().numpy().copy())
    r.tick()
    torch.cuda.synchronize()
    hist.append(snap + (pol.jerk.cpu().numpy().copy(), r.status().copy()))
    if (hist[-1][-1] != 0).all():
        break
status = hist[-1][-1]
print("crashed %.3f merged %.3f timeout %.3f of %d" % ((status == 2).mean(), (status == 1).mean(), (status == 3).mean(), n))
crash_x = []
for e in np.nonzero(status == 2)[0]:
    t = next(i for i, h in enumerate(hist) if h[-1][e] == 2)
    crash_x.append(hist[t][0][e][0])
print("ego x at the last view before the crash: quantiles", np.round(np.quantile(crash_x, [0, .1, .25, .5, .75, .9, 1]), 1))
envs = np.nonzero(status == 2)[0]
e = envs[which]
t_end = next(i for i, h in enumerate(hist) if h[-1][e] == 2)
for t in range(max(0, t_end - 30), t_end + 1):
    ego5, k, ox, ov, oa, jerk, st = hist[t]
    kk = k[e]; dx = ox[e, :kk] - ego5[e, 0]
    order = np.argsort(np.abs(dx))[:4]; order = order[np.argsort(-dx[order])]
    print("t %3d ego x %7.2f y %5.2f v %5.2f a %5.2f | jerk %5.2f | " % (t, ego5[e, 0], ego5[e, 1], ego5[e, 2], ego5[e, 3], jerk[e]) +
          " ".join("(dx %6.1f v %4.1f a %4.1f)" % (dx[i], ov[e, i], oa[e, i]) for i in order))
