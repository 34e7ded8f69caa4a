"""CPU-only tests: the C-ABI library loads and exports every declared symbol, host helpers are exact,
the host-side mirror behaves like the reference's interface, and the product refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, load_golden


def _lib():
    import rl_mpc_lanemerging_amd as pkg
    if pkg.build.needs_build():
        pkg.build.build()
    from rl_mpc_lanemerging_amd import _capi
    return _capi


def test_library_exports_every_declared_symbol():
    capi = _lib()
    lib = capi.load()
    header = open(os.path.join(REPO, "include", "stmpc.h")).read()
    declared = set(re.findall(r"\b(stmpc_[a-z_0-9]+)\s*\(", header))
    declared -= {"stmpc_params", "stmpc_stats", "stmpc_ctx"}
    assert declared == set(capi.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_params_struct_layout_matches_header():
    capi = _lib()
    header = open(os.path.join(REPO, "include", "stmpc.h")).read()
    body = header[header.index("typedef struct stmpc_params {"):header.index("} stmpc_params;")]
    names = []
    for line in body.splitlines():
        line = line.split("/*")[0].strip()
        if line.startswith("double"):
            names += [n.strip() for n in line[len("double"):].rstrip(";").split(",")]
    assert names == [n for n, _ in capi.Params._fields_]
    assert ctypes.sizeof(capi.Params) == 8 * len(names)


def _header_struct_fields(name):
    """Field names and C types of `typedef struct <name> { ... } <name>;` in include/stmpc.h, in order (comments stripped)."""
    import re
    header = open(os.path.join(REPO, "include", "stmpc.h")).read()
    body = header[header.index("typedef struct %s {" % name) + len("typedef struct %s {" % name):header.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        if stmt.startswith("const "):
            stmt = stmt[6:]
        ctype, rest = stmt.split(" ", 1)
        for n in rest.split(","):
            n = n.strip()
            out.append((n.lstrip("*"), ctype + " *" if n.startswith("*") else ctype))
    return out


def test_every_other_struct_layout_matches_header():
    """stmpc_stats, stmpc_combined_cfg, stmpc_policy_features_cfg, stmpc_sim_cfg, stmpc_profile_totals: the ctypes mirrors have the header's
    fields in the header's order with the header's types (a field added on one side only would shift everything after it silently)."""
    capi = _lib()
    ctype_of = {"double": ctypes.c_double, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
                "double *": ctypes.POINTER(ctypes.c_double)}
    for name, cls in (("stmpc_stats", capi.Stats), ("stmpc_combined_cfg", capi.CombinedCfg), ("stmpc_policy_features_cfg", capi.FeaturesCfg),
                      ("stmpc_sim_cfg", capi.SimCfg), ("stmpc_profile_totals", capi.ProfileTotals)):
        want = _header_struct_fields(name)
        have = [(n, t) for n, t in cls._fields_]
        assert [n for n, _ in want] == [n for n, _ in have], name
        assert [ctype_of[t] for _, t in want] == [t for _, t in have], name
    lib = capi.load()
    f = capi.FeaturesCfg(cars_ahead=2, cars_behind=2, use_acceleration=1, time_feature=1)
    assert lib.stmpc_policy_features_len(ctypes.byref(f)) == 21          # (host-only helper: no GPU needed)
    f.use_acceleration, f.time_feature, f.cars_ahead = 0, 0, 3
    assert lib.stmpc_policy_features_len(ctypes.byref(f)) == 5 * 3 + 4


def test_host_helpers_match_reference_golden():
    capi = _lib()
    from rl_mpc_lanemerging_amd import control
    g = load_golden("golden_default.npz")
    for i in range(g["ego"].shape[0]):
        x, y = g["ego"][i, 0], g["ego"][i, 1]
        assert capi.ego_s(x, y) == g["ego"][i, 4]
        assert control.get_ego_s((float(x), float(y))) == g["ego"][i, 4]
    import rl_mpc_lanemerging_amd as pkg
    snap = pkg.Settings.snapshot()
    try:
        pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
        p = capi.Params.from_settings(pkg.Settings)
        assert capi.num_t(p) == g["t_values"].size == 18
        for i in range(g["ego"].shape[0]):
            assert capi.num_s(p, g["ego"][i, 4]) == g["num_s"][i]
            sv = __import__("rl_mpc_lanemerging_amd").st.s_values_for(g["ego"][i, 4], p)
            assert sv.size == g["num_s"][i]
        pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
        p = capi.Params.from_settings(pkg.Settings)
        assert capi.num_t(p) == 40 and capi.num_s(p, 0.0) == 7201
    finally:
        pkg.Settings.restore(snap)


def test_settings_load_from_reference_style_json(tmp_path):
    import json
    import rl_mpc_lanemerging_amd as pkg
    snap = pkg.Settings.snapshot()
    try:
        cfg = {"TASK": "ST", "S_DISCRETIZATION": 0.05, "T_DISCRETIZATION": 0.30, "FUTURE_S": 150.0, "FUTURE_T": 5.0,
               "V_WEIGHT": 0.5, "A_WEIGHT": 10.0, "J_WEIGHT": 10.0, "D_WEIGHT": 10.0, "MIN_ALLOWED_DISTANCE": 5,
               "CRASH_MIN_S": 20, "JERK_VALUES": {"0": -5, "1": 0}}
        f = tmp_path / "st_low.json"
        f.write_text(json.dumps(cfg))
        pkg.Settings.load_from_file(str(f))
        assert pkg.Settings.CRASH_MIN_S == 20 and pkg.Settings.TASK == "ST"
        assert pkg.Settings.JERK_VALUES == {0: -5, 1: 0}        # dict keys cast to int (config.py:167-168)
    finally:
        pkg.Settings.restore(snap)
        for k in ("TASK", "JERK_VALUES"):
            if hasattr(pkg.Settings, k):
                delattr(pkg.Settings, k)


def test_mean_abs_jerk_matches_oracle():
    capi = _lib()
    from rl_mpc_lanemerging_amd import st
    from oracle import st_oracle as orc
    g = load_golden("golden_default.npz")
    for i in range(0, 60):
        bt = int(g["best_t"][i])
        if bt < 2:
            continue
        seq = g["s_sequence"][i, :bt + 1]
        a = st.get_path_mean_abs_jerk(seq, g["ego"][i, 2], g["ego"][i, 3], 0.3)
        seqc = np.ascontiguousarray(seq)
        b = orc.lib().orc_path_mean_abs_jerk(seqc.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), seq.size,
                                             g["ego"][i, 2], g["ego"][i, 3], 0.3)
        c = capi.load().stmpc_path_mean_abs_jerk(seqc.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), seq.size,
                                                 g["ego"][i, 2], g["ego"][i, 3], 0.3)
        assert a == b == c


def test_mean_abs_jerk_matches_reference_golden():
    """st.get_path_mean_abs_jerk (library and its oracle twin) against the reference's own return values (golden_jerk.npz: 802
    paths of the state goldens, at the planning step and at the simulator tick)."""
    capi = _lib()
    from rl_mpc_lanemerging_amd import st
    from oracle import st_oracle as orc
    j = load_golden("golden_jerk.npz")
    src = [load_golden(str(f)) for f in j["files"]]
    dp = ctypes.POINTER(ctypes.c_double)
    for q in range(j["row"].size):
        g = src[int(j["file_index"][q])]
        seq = np.ascontiguousarray(g["s_sequence"][int(j["row"][q]), :int(j["length"][q])])
        want = j["mean_abs_jerk"][q]
        assert st.get_path_mean_abs_jerk(seq, j["v0"][q], j["a0"][q], j["dt"][q]) == want
        assert capi.load().stmpc_path_mean_abs_jerk(seq.ctypes.data_as(dp), seq.size, j["v0"][q], j["a0"][q], j["dt"][q]) == want
        if q % 8 == 0:
            assert orc.lib().orc_path_mean_abs_jerk(seq.ctypes.data_as(dp), seq.size, j["v0"][q], j["a0"][q], j["dt"][q]) == want


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product must raise, never silently compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    capi = _lib()
    with pytest.raises(capi.StmpcError) as e:
        capi.Context(0)
    assert e.value.code == capi.STMPC_ENODEV


def test_product_does_not_import_oracle():
    pkgdir = os.path.join(REPO, "rl-mpc-lanemerging_amd")
    for root, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("st_oracle.c header", ""), f
    assert "oracle" not in open(os.path.join(REPO, "include", "stmpc.h")).read()


def test_pretrained_actors_are_package_data():
    """The reference's pretrained actors (pretrained_models/*/policy.pt as tensors; configs/combined_medium_1.json:4 names one by MODEL_NAME) ship
    inside the package, and no module of the package refers to the test tree."""
    from rl_mpc_lanemerging_amd import actor
    pkgdir = os.path.realpath(os.path.join(REPO, "rl-mpc-lanemerging_amd"))
    for name in ("runs/ddpg_medium1_extended", "ddpg_low1", "default1", "moderate1", "fast1"):
        path = os.path.realpath(actor.weights_path(name))
        assert path.startswith(pkgdir + os.sep), path
        w = actor.load_weights(name)
        assert w["w0"].shape == (400, 21) and w["w1"].shape == (300, 400) and w["w2"].shape == (1, 300)
    for f in os.listdir(pkgdir):
        if f.endswith(".py"):
            assert "tests" not in open(os.path.join(pkgdir, f)).read(), f


def test_design_quotes_the_measured_numbers():
    """DESIGN.md section 5 does not retype measured numbers: its figures block is generated from profiles/r6/measured.json and the committed bench line
    (scripts/design_numbers.py), and must equal what those files say."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("design_numbers", os.path.join(REPO, "scripts", "design_numbers.py"))
    dn = importlib.util.module_from_spec(spec); spec.loader.exec_module(dn)
    text = open(os.path.join(REPO, "DESIGN.md")).read()
    assert dn.BEGIN in text and dn.END in text
    cur = text[text.index(dn.BEGIN):text.index(dn.END) + len(dn.END)]
    assert cur == dn.block("r6"), "run scripts/design_numbers.py r6"
    import json
    m = json.load(open(os.path.join(REPO, "profiles", "r6", "measured.json")))["h40a21"]
    assert ("%.3f ms" % m["dominant_kernel_avg_ms"]) in cur and ("%.2e" % m["hbm_roofline_frac"]) in cur


def test_synth_generator_is_seeded_and_ordered():
    from rl_mpc_lanemerging_amd import synth
    a = synth.generate_states(64, seed=3)
    b = synth.generate_states(64, seed=3)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    ego, k, ox, ov = a
    assert (np.diff(ox[:, :6], axis=1) < 0).all()            # front -> back
    assert (ego[:, 2] - ego[:, 3] * 0.3 >= -1e-12).all()


def test_library_carries_the_hash_of_its_sources():
    """The built library names the sources it was compiled from (build.source_hash -> -DSTMPC_SRC_HASH -> stmpc_backend_info); bench.py uses
    that to mark counters taken from another build as stale, and the committed counters of the newest round belong to these sources."""
    import json, os
    import rl_mpc_lanemerging_amd as pkg
    from rl_mpc_lanemerging_amd import _capi
    from conftest import REPO
    import pytest
    h = pkg.build.source_hash()
    if pkg.build.needs_build():
        pytest.skip("libstmpc.so is older than its sources (run __graft_entry__.build())")
    assert len(h) == 16 and _capi.library_source_hash() == h, "libstmpc.so was built from other sources: rebuild (python -c 'import __graft_entry__ as g; g.build()')"
    measured = json.load(open(os.path.join(REPO, "profiles", "r4", "measured.json")))
    assert all("csrc_hash" in v for v in measured.values())
