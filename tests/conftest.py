import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def settings_from_golden(g):
    """Apply the Settings overrides a golden file was generated with; returns (Params, OrcParams)."""
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi
    from oracle import st_oracle as orc
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    pkg.Settings.CRASH_MIN_S = 20
    over = {str(k): float(v) for k, v in zip(g["override_keys"], g["override_vals"])}
    pkg.apply_overrides(over)
    p = _capi.Params.from_settings(pkg.Settings)
    return p, orc.OrcParams.from_dict(p.as_dict())


@pytest.fixture
def restore_settings():
    import rl_mpc_lanemerging_amd as pkg
    snap = pkg.Settings.snapshot()
    yield
    pkg.Settings.restore(snap)


@pytest.fixture(scope="session")
def gpu_ctx():
    from rl_mpc_lanemerging_amd import _capi
    return _capi.default_context()
