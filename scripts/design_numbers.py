#!/usr/bin/env python3
"""One source of truth for the round's headline figures: rewrites the block between the `numbers:begin` / `numbers:end` markers of DESIGN.md
(section 5) from profiles/<round>/measured.json (scripts/profile_summarize.py) and profiles/<round>/bench_h40a21.json (scripts/evidence.sh), so
that the text never retypes a measured number.  `tests/test_host_cpu.py::test_design_quotes_the_measured_numbers` fails when the block and
the files disagree.   usage: scripts/design_numbers.py [round, default r6] [--check]"""
import json, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- numbers:begin (scripts/design_numbers.py) -->", "<!-- numbers:end -->"


def block(rnd="r6"):
    P = os.path.join(REPO, "profiles", rnd)
    m = json.load(open(os.path.join(P, "measured.json")))["h40a21"]
    b = json.loads(open(os.path.join(P, "bench_h40a21.json")).read().strip().splitlines()[-1])
    rows = []
    add = lambda k, v, src: rows.append("| %s | %s | %s |" % (k, v, src))
    mj = "`profiles/%s/measured.json` h40a21" % rnd
    add("dominant kernel", "`%s`" % m["dominant_kernel"], mj + ".dominant_kernel")
    add("its average duration, `rocprofv3 --kernel-trace --stats`, %d launches" % m["dominant_kernel_calls"],
        "**%.3f ms** (min %.3f, max %.3f)" % (m["dominant_kernel_avg_ms"], m["dominant_kernel_min_ms"], m["dominant_kernel_max_ms"]), ".dominant_kernel_avg_ms")
    add("algorithmic bytes per launch (316 B x 4096, SURVEY 8d)", "%d" % m["algorithmic_bytes_per_launch"], ".algorithmic_bytes_per_launch")
    add("HBM roofline: achieved / 8 TB/s", "%.3f GB/s = **%.2e**" % (m["hbm_roofline_achieved_gbs"], m["hbm_roofline_frac"]), ".hbm_roofline_frac")
    add("measured HBM traffic per step (FETCH x 2 + WRITE, all kernels)", "%.1f MB" % (m["hbm_bytes_per_step"] / 1e6), ".hbm_bytes_per_step")
    v, s_, l = m["per_step_SQ_INSTS_VALU"], m["per_step_SQ_INSTS_SALU"], m["per_step_SQ_INSTS_LDS"]
    add("wave-instructions per step: VALU / SALU / LDS", "%.3f G / %.3f G / %.3f G" % (v / 1e9, s_ / 1e9, l / 1e9), ".per_step_SQ_INSTS_*")
    add("VALU issue over the dominant kernel's duration (4 cycles x VALU / (1024 SIMDs x 2.4 GHz x avg))",
        "%.0f %%" % (100.0 * v * 4.0 / (1024 * 2.4e9 * m["dominant_kernel_avg_ms"] * 1e-3)), "derived")
    add("waves parked: SQ_WAIT_ANY / SQ_WAVE_CYCLES", "%.0f %%" % (100.0 * m["per_step_SQ_WAIT_ANY"] / m["per_step_SQ_WAVE_CYCLES"]), ".per_step_SQ_WAIT_ANY")
    add("LDS: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE", "%.0f %%" % (100.0 * m["per_step_SQ_LDS_BANK_CONFLICT"] / m["per_step_SQ_LDS_IDX_ACTIVE"]), ".per_step_SQ_LDS_*")
    if "candidates_per_node" in m:
        add("candidates offered per expanded node / executed lane-slots per offered candidate", "%.2f / %.2f" % (m["candidates_per_node"], m["lane_slots_per_candidate"]), ".candidates_per_node")
    add("library sources the counters were taken from", "`%s`" % m["csrc_hash"], ".csrc_hash")
    bj = "`profiles/%s/bench_h40a21.json`" % rnd
    add("headline `bench.py` line (builder-run, same sources)", "**%.0f solves/s**, %.3f ms per step, seed-median %.0f" % (b["value"], b["ms_per_step"], b.get("value_seed_median") or 0), bj)
    add("... its HIP-event kernel window / the trace average above", "%.3f ms / %.3f ms" % (b["roofline"]["kernel_ms"], b["roofline"].get("rocprof_kernel_avg_ms") or 0), bj + " roofline")
    sec = b.get("secondary") or {}
    if sec:
        add("... `secondary`: reference lattice / `do_st_control` / combined tick", "%.2f M solves/s / %.2f M speeds/s / %.2f M ticks/s" % (
            sec["reference_lattice"]["value"] / 1e6, sec["st_control"]["value"] / 1e6, sec["combined_tick"]["value"] / 1e6), bj + " secondary")
    t = b["tiers"]
    add("... first window / second window / repeated passes / nodes per solve", "%d / %d / %d / %.0f" % (t["first_lds_window"], t["larger_lds_window"], t["bound_retries"], t["nodes_expanded_per_solve"]), bj + " tiers")
    cb = b.get("cpu_baseline")
    if cb:
        add("... CPU baseline (oracle, %d threads of %s)" % (cb["cores"], cb["cpu_model"]), "%.0f solves/s (heap Dijkstra, the reference's algorithm: %.0f)" % (cb["value"], cb["reference_algorithm_value"]), bj + " cpu_baseline")
    return "\n".join([BEGIN, "| figure | value | source |", "|---|---|---|"] + rows + [END])


def main():
    rnd = next((a for a in sys.argv[1:] if not a.startswith("--")), "r6")
    path = os.path.join(REPO, "DESIGN.md")
    s = open(path).read()
    new = block(rnd)
    if BEGIN not in s:
        raise SystemExit("DESIGN.md has no numbers block")
    cur = s[s.index(BEGIN):s.index(END) + len(END)]
    if "--check" in sys.argv:
        sys.exit(0 if cur == new else 1)
    open(path, "w").write(s.replace(cur, new))
    print(new)


if __name__ == "__main__":
    main()
