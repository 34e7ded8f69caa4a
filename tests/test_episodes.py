"""batched SUMO-free merge episodes.  STATISTICAL parity only: the world is the planner's own model, not SUMO
(its highway vehicles do not yield at the junction the way SUMO's IDM vehicles do), so the test checks the outcome pattern of the
reference's pure-ST evaluation in light traffic (experiment_data/saved_data.csv, row st_low: no crashes, every episode merges,
25.7 s to merge at 10.4 m/s mean speed) within wide bands, and only safety (no crashes) in dense traffic, where this world's
ego often waits at the merge point until the episode's time limit (measured: 57 % merged at 1.2 s headway; reference: 100 %)."""
import numpy as np
import pytest

# the reference's reported means for TASK "ST" (numbers copied as data: saved_data.csv rows 2, 13, 20)
REFERENCE_ST = {2.4: dict(crashed=0.0, merged=1.0, mean_speed=10.416, time_to_merge=25.659, mean_abs_jerk=1.074),
                1.8: dict(crashed=0.0, merged=1.0, mean_speed=9.297, time_to_merge=28.645, mean_abs_jerk=1.262),
                1.2: dict(crashed=0.0, merged=1.0, mean_speed=8.919, time_to_merge=29.838, mean_abs_jerk=1.105)}


@pytest.mark.gpu
@pytest.mark.parametrize("interval", [2.4, 1.2])
def test_st_episodes_match_the_reference_statistically(interval, gpu_ctx, restore_settings):
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import episodes
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.apply_overrides(dict(BASE_TRAFFIC_INTERVAL=interval, OTHER_CAR_SPEED=7.0))
    n = 256
    st = episodes.run_episodes(n, seed=7, controller="st", ctx=gpu_ctx)
    s = episodes.summary(st)
    ref = REFERENCE_ST[interval]
    assert (st["crashed"] + st["merged"] + st["timed_out"] == 1).all()
    assert s["crashed"] <= 0.02                                   # reference: 0.0
    assert 0.05 < s["mean_abs_jerk"] < 4.0 and 15.0 < s["max_speed"] <= 30.0      # reference: 1.07-1.26, 23.1-23.6
    if interval == 2.4:
        assert s["merged"] >= 0.95                                # reference: 1.0
        assert abs(s["time_to_merge"] - ref["time_to_merge"]) < 0.6 * ref["time_to_merge"]
        assert abs(s["mean_speed"] - ref["mean_speed"]) < 0.35 * ref["mean_speed"]
    else:
        assert s["merged"] >= 0.3
    # determinism: same seed, same episodes
    st2 = episodes.run_episodes(n, seed=7, controller="st", ctx=gpu_ctx)
    assert np.array_equal(st["ticks"], st2["ticks"]) and np.array_equal(st["mean_speed"], st2["mean_speed"])
    # denser traffic does not make merging faster
    assert np.isfinite(st["time_to_merge"][st["merged"] == 1]).all()
