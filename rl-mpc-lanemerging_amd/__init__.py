"""rl-mpc-lanemerging_amd -- MI355X-native ST ("MPC") lattice solver.

A from-scratch gfx950 implementation of one hot path of jlubars/RL-MPC-LaneMerging: the
finite-horizon trajectory search of ``st.py`` / ``st_cy.pyx`` against the traffic predicted
by ``prediction.py``.  See DESIGN.md for scope and INTEGRATION.md for the drop-in boundary.

Import as ``rl_mpc_lanemerging_amd`` (the repo-root shim maps the hyphenated directory).
"""
from . import build, config, control, synth                       # noqa: F401
from .config import Settings, apply_overrides, REFERENCE_DEFAULT, SYNTHETIC_H40A21  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # the modules that need libstmpc.so are imported lazily so that `build` is usable before the first build
    if name in ("_capi", "st", "prediction", "sharding", "combined", "combined_bench", "episodes"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
