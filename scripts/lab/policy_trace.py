"""Trace of the actor-alone episodes (controller="policy"; STMPC_TRACE_CONTROLLER=combined: the combined controller) up to a crash (analysis tool).
usage: policy_trace.py <actor> <interval> <speed> <n> [which crashing env, default 0] [yield_overlap] [remap: 1 = feed the actor SUMO's lane y for the ego's x]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import episodes, _capi, actor
act, interval, speed, n = sys.argv[1], float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
which = int(sys.argv[5]) if len(sys.argv) > 5 else 0
yo = int(sys.argv[6]) if len(sys.argv) > 6 and sys.argv[6] != "-" else None
remap = len(sys.argv) > 7 and sys.argv[7] == "1"
import os
controller = os.environ.get("STMPC_TRACE_CONTROLLER", "policy")
pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
if controller == "combined":
    from rl_mpc_lanemerging_amd import combined_bench
    pkg.apply_overrides(combined_bench.COMBINED_MEDIUM_1)
pkg.apply_overrides(dict(BASE_TRAFFIC_INTERVAL=interval, OTHER_CAR_SPEED=speed))
ctx = _capi.default_context(); dev = torch.device("cuda", 0)
pol = actor.DDPGActor(act, n, ctx, pkg.Settings, dev)
if remap:
    from rl_mpc_lanemerging_amd import scenario
    lx, ly, _ = scenario.lane_polyline()
    lx, ly = torch.tensor(lx, device=dev), torch.tensor(ly, device=dev)
    inner = pol

    class Remap:
        jerk = inner.jerk

        def __call__(self, step, ego4, k, ox, ov, oa):
            e = ego4.clone(); x = e[:, 0].contiguous()
            i = torch.bucketize(x, lx).clamp(1, len(lx) - 1)
            w = ((x - lx[i - 1]) / (lx[i] - lx[i - 1])).clamp(0, 1)
            e[:, 1] = torch.where(x >= 1.5, torch.full_like(x, -1.6), ly[i - 1] + w * (ly[i] - ly[i - 1]))
            return inner(step, e, k, ox, ov, oa)
    pol = Remap()
r = episodes.EpisodeRunner(n, 21, controller, pol, ctx, 16)
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
    hist.append(snap + ((r.last_rl.cpu().numpy().copy() if controller == "combined" else pol.jerk.cpu().numpy().copy()), r.status().copy()))
    if (hist[-1][-1] != 0).all():
        break
status = hist[-1][-1]
print("crashed %.3f merged %.3f timeout %.3f of %d" % ((status == 2).mean(), (status == 1).mean(), (status == 3).mean(), n))
crash_x = []
for e in np.nonzero(status == 2)[0]:
    t = next(i for i, h in enumerate(hist) if h[-1][e] == 2)
    crash_x.append(hist[t][0][e][0])
print("ego x at the last view before the crash: quantiles", np.round(np.quantile(crash_x, [0, .1, .25, .5, .75, .9, 1]), 1))
def nearest_dx(h, e):
    ego5, k, ox = h[0], h[1], h[2]
    d = ox[e, :k[e]] - ego5[e, 0]
    return d[np.argmin(np.abs(d))] if len(d) else np.nan
dxc = [nearest_dx(hist[next(i for i, h in enumerate(hist) if h[-1][e] == 2)], e) for e in np.nonzero(status == 2)[0]]
print("nearest vehicle dx (its front - ego front) at the last view before the crash, histogram -10..10 by 2:", np.histogram(dxc, bins=np.arange(-10, 11, 2))[0])
vc = [hist[next(i for i, h in enumerate(hist) if h[-1][e] == 2)][0][e][2] for e in np.nonzero(status == 2)[0]]
print("ego speed there: quantiles", np.round(np.quantile(vc, [0, .25, .5, .75, 1]), 1))
dxm = []
for e in np.nonzero(status == 1)[0]:
    t = next((i for i, h in enumerate(hist) if h[0][e][0] > -38.9), None)
    if t is not None: dxm.append(nearest_dx(hist[t], e))
print("merged episodes: nearest dx at the first view with x > -38.9, histogram -20..20 by 4:", np.histogram(dxm, bins=np.arange(-20, 21, 4))[0])
envs = np.nonzero(status == 2)[0]
e = envs[which]
t_end = next(i for i, h in enumerate(hist) if h[-1][e] == 2)
for t in range(max(0, t_end - 30), t_end + 1):
    ego5, k, ox, ov, oa, jerk, st = hist[t]
    kk = k[e]; dx = ox[e, :kk] - ego5[e, 0]
    order = np.argsort(np.abs(dx))[:4]; order = order[np.argsort(-dx[order])]
    print("t %3d ego x %7.2f y %5.2f v %5.2f a %5.2f | %s %5.2f | " % (t, ego5[e, 0], ego5[e, 1], ego5[e, 2], ego5[e, 3], "rl" if controller == "combined" else "jerk", jerk[e]) +
          " ".join("(dx %6.1f v %4.1f a %4.1f)" % (dx[i], ov[e, i], oa[e, i]) for i in order))
