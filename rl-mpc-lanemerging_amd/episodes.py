"""Batched SUMO-free merge episodes: the role of the reference's ``control.run_episode`` /
``control.evaluate_control`` (control.py:207-395) and of the per-episode columns of ``stats.StatsAggregator``
(stats.py:43-111), for N environments stepped in lock-step on one GPU.

The world restates the SUMO scenario the reference configures (Krauss vehicles of the "simple traffic distribution", the ego under
speed mode 22; see ``stmpc_sim_*`` in include/stmpc.h) but is not SUMO: results are comparable with the reference's reported numbers
(experiment_data/saved_data.csv) as DISTRIBUTIONS only.  Every tick is: planner view ->
controller (``st.do_st_control`` for all environments in one launch, or the combined controller) -> world step; nothing but a
periodic "all finished?" flag crosses to the host.
"""
import os

import numpy as np

from . import _capi, scenario, synth
from .config import Settings

# scenario constants of the reference's SUMO network and episode runner
SPAWN_X, DESPAWN_X = -250.0, 100.0          # highwayrear starts at x = -250, highwayahead ends at x = 100 (merge.net.xml:45-49)
EGO_START_ARC = 40.0                        # departPos = 40 on the ramp (control.py:42)
ARRIVE_X = 1.5 + 50.0                       # arrivalPos = 50 on highwayahead (control.py:42)
RAMP_START = (-250.47, 28.47)               # first point of lane ramp_0 (merge.net.xml:52)


def ego_start_position(route="lane"):
    """Point 40 m along the ramp from its start: on the lane's centre line (``route="lane"``, what SUMO does) or on the straight line from the
    ramp's start to the junction entry (``route=None``: the world of rounds 3-4, whose ego moves as the predictor assumes)."""
    if route == "lane":
        return scenario.point_at_arc(EGO_START_ARC)
    x0, y0 = RAMP_START
    d = np.hypot(-50.58 - x0, 1.71 - y0)
    ux, uy = (-50.58 - x0) / d, (1.71 - y0) / d
    return x0 + EGO_START_ARC * ux, y0 + EGO_START_ARC * uy


def sim_cfg(seed=0, max_episode_length=100.0, route="lane"):
    """``route``: "lane" (default) = the ego follows the centre line of its lanes in the reference's network (``scenario.py``; TraCI reports those
    positions), None = the straight lines the planner assumes (prediction.py:46-59; the world of rounds 3-4, kept for the mechanics test)."""
    S = Settings
    ex, ey = ego_start_position(route)
    g = lambda name, default: getattr(S, name, default)
    return _capi.SimCfg(tick_length=S.TICK_LENGTH, other_car_speed=g("OTHER_CAR_SPEED", 7.0), base_traffic_interval=g("BASE_TRAFFIC_INTERVAL", 1.2),
                        spawn_x=SPAWN_X, despawn_x=DESPAWN_X, ego_start_x=ex, ego_start_y=ey, arrive_x=ARRIVE_X, sensor_radius=g("SENSOR_RADIUS", 125.0),
                        start_speed=g("START_SPEED", 15.0), start_speed_std=g("START_SPEED_VARIANCE", 5.0), min_start_speed=g("MIN_START_SPEED", 5.0),
                        max_start_speed=g("MAX_START_SPEED", 25.0),
                        # SUMO vType "normal" of the simple traffic distribution (merge_impossible.rou.xml:3) + SUMO defaults (emergencyDecel, width)
                        veh_accel=4.5, veh_decel=6.0, veh_min_gap=1.0, veh_tau=0.5, veh_emergency_decel=9.0, veh_length=g("CAR_LENGTH", 5.0), veh_width=1.8, speed_dev=0.0,
                        vary_traffic_start_times=int(bool(g("VARY_TRAFFIC_START_TIMES", True))),
                        randomize_start_speed=int(bool(g("RANDOMIZE_START_SPEED", True))), max_ticks=int(max_episode_length / S.TICK_LENGTH),
                        yield_overlap=2,          # (the one junction rule, include/stmpc.h)
                        seed=int(seed), disruption_min_s=float(g("MERGE_POINT_X", -50.0))).set_route(np.stack(scenario.lane_polyline()[:2], axis=1) if route == "lane" else None)


class EpisodeRunner:
    """N merge episodes stepped in lock-step on the device, one ``tick()`` at a time (``run_episodes`` drives it to the end;
    ``bench.py --workload episodes`` times its ticks)."""

    def __init__(self, n, seed=0, controller="st", policy=None, ctx=None, kmax=32, max_episode_length=100.0):
        import torch
        self.torch = torch
        if controller not in ("st", "combined"):
            # (the policy alone -- TASK EVALUATE_DDPG -- is not on the solver's path: SURVEY section 2 marks ddpg.py out of scope; removed in round 6)
            raise ValueError("controller must be 'st' or 'combined', not %r" % (controller,))
        self.n, self.kmax, self.controller, self.policy = int(n), int(kmax), controller, policy
        self.ctx = ctx or _capi.default_context()
        self.params = _capi.Params.from_settings(Settings)
        self.cfg = sim_cfg(seed, max_episode_length)
        self.tick_length = Settings.TICK_LENGTH
        dev = torch.device("cuda", torch.cuda.current_device())
        H = _capi.num_t(self.params)
        z = lambda *shape, dtype=torch.float64: torch.zeros(shape, dtype=dtype, device=dev)
        self.d_ego5, self.d_k, self.d_ox, self.d_ov, self.d_oa = z(n, 5), z(n, dtype=torch.int32), z(n, kmax), z(n, kmax), z(n, kmax)
        self.d_path, self.d_bt, self.d_cost, self.d_speed = z(n, H, dtype=torch.int32), z(n, dtype=torch.int32), z(n), z(n)
        self.d_fine, self.d_fine_len = z(n, _capi.QP_NMAX), z(n, dtype=torch.int32)
        self.ccfg = _capi.CombinedCfg.from_settings(Settings, sparse_control=True) if controller == "combined" else None      # (a tick ends with a host-side status check anyway)
        self.takeovers, self.controlled = z(n), z(n)
        self.last_rl = torch.ones(n, dtype=torch.int32, device=dev)
        self.d_status = z(n, dtype=torch.int32)
        self.ticks_done = 0
        self.ctx.sim_init(self.cfg, n)

    def tick(self):
        """Planner view -> controller -> world step for every environment (finished environments idle)."""
        from . import combined
        torch, ctx, n, kmax = self.torch, self.ctx, self.n, self.kmax
        ctx.sim_view(self.cfg, n, kmax, self.d_ego5.data_ptr(), self.d_k.data_ptr(), self.d_ox.data_ptr(), self.d_ov.data_ptr(),
                     self.d_oa.data_ptr() if self.controller != "st" else 0)
        if self.controller == "st":
            ctx.st_control_batch_device(self.params, self.tick_length, n, kmax, self.d_ego5.data_ptr(), self.d_k.data_ptr(), self.d_ox.data_ptr(), self.d_ov.data_ptr(),
                                        self.d_path.data_ptr(), self.d_bt.data_ptr(), self.d_cost.data_ptr(), self.d_speed.data_ptr(), self.d_fine.data_ptr(),
                                        self.d_fine_len.data_ptr(), 0)
            cmd = self.d_speed
        else:
            d = combined.decide_batch_device(ctx, self.params, self.ccfg, self.d_ego5, self.d_k, self.d_ox, self.d_ov, self.policy, self.last_rl, d_oa=self.d_oa)
            cmd = d["speed"]
            # per-episode takeover share (the reference's stats count the ticks of the episode itself): finished environments keep
            # returning their final state from sim_view, their repeated decisions must not be counted
            ctx.sim_status_device(n, self.d_status.data_ptr())
            running = (self.d_status == 0).to(torch.float64)
            self.takeovers += d["takeover"].to(torch.float64) * running
            self.controlled += running
            self.last_rl = (d["takeover"] == 0).to(torch.int32)
        ctx.sim_step(self.params, self.cfg, n, cmd.data_ptr())
        self.ticks_done += 1

    def status(self):
        """Host copy of the status words (synchronises) after raising any latched device-side error."""
        status, _, _, _ = self.ctx.sim_read(self.n)
        self.ctx.check_error()          # (an asynchronous solver / QP error of the ticks since the last check)
        return status

    def result(self):
        status, ticks, acc, ego4 = self.ctx.sim_read(self.n)
        self.ctx.check_error()
        samples = np.maximum(acc[:, 4], 1.0)
        # mean |jerk| as the reference reports it: its jerk history holds a 0 for the first tick (control.py:284-287) and the mean is taken
        # over all ticks (stats.py:60) -- n samples, not n - 1 jerk terms
        out = {"crashed": (status == 2).astype(np.float64), "merged": (status == 1).astype(np.float64), "timed_out": (status == 3).astype(np.float64),
               "mean_speed": acc[:, 0] / samples, "max_speed": acc[:, 1], "mean_abs_jerk": acc[:, 2] / samples,
               "closest_distance": np.where(acc[:, 7] > 0, acc[:, 5], np.nan), "mean_closest_distance": np.where(acc[:, 7] > 0, acc[:, 6] / np.maximum(acc[:, 7], 1.0), np.nan),
               "time_taken": ticks * self.tick_length, "ticks": ticks, "status": status, "ego4": ego4}
        out["time_to_merge"] = np.where(status == 1, out["time_taken"], np.nan)
        # the reference's "disruption" columns (stats.py:64-68): deceleration of the nearest vehicle behind the ego, per controlled tick past MERGE_POINT_X
        have = acc[:, 10] > 0
        out["mean_disruption"] = np.where(have, acc[:, 8] / np.maximum(acc[:, 10], 1.0), np.nan)
        out["max_disruption"] = np.where(have, acc[:, 9], np.nan)
        out["total_disruption"] = np.where(have, acc[:, 8] * self.tick_length, np.nan)
        out["disruption_time"] = np.where(have, acc[:, 11] * self.tick_length, np.nan)
        if self.controller == "combined":
            out["percent_st"] = (self.takeovers / self.torch.clamp(self.controlled, min=1.0)).cpu().numpy()
        return out


def run_episodes(n, seed=0, controller="st", policy=None, ctx=None, kmax=32, max_episode_length=100.0, check_every=16, max_ticks=None):
    """Run ``n`` merge episodes to the end (or for ``max_ticks`` ticks); returns the per-episode columns of the reference's stats report
    (``crashed``, ``merged``, ``mean_speed``, ``max_speed``, ``mean_abs_jerk``, ``closest_distance``, ``mean_closest_distance``,
    ``time_taken``, ``time_to_merge`` (NaN unless merged)) plus ``ticks``, ``status`` (0 still running), ``ego4`` and ``percent_st``
    (combined controller only).

    controller: "st" = ``st.do_st_control`` every tick (TASK "ST"); "combined" = ``do_combined_control`` with ``policy``
    (see ``combined.decide_batch_device``)."""
    r = EpisodeRunner(n, seed, controller, policy, ctx, kmax, max_episode_length)
    limit = r.cfg.max_ticks + 1 if max_ticks is None else min(int(max_ticks), r.cfg.max_ticks + 1)
    for tick in range(limit):
        r.tick()
        if tick % check_every == check_every - 1 and (r.status() != 0).all():
            break
    return r.result()


def summary(stats):
    """Column means as the reference's report rows hold them (stats.py:145-158)."""
    return {k: float(np.nanmean(v)) for k, v in stats.items() if k not in ("ticks", "status", "ego4")}
