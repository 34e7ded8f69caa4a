#!/bin/bash
# Spill-placement evidence for the solve kernels: compiles csrc/stmpc.hip to gfx950 assembly and writes, for the two
# dominant instantiations, the per-depth census and the loop tree with the spill traffic of every loop's own blocks.
# usage: scripts/isa/spill_report.sh <out dir>        (needs hipcc; no GPU)
out=${1:-profiles/r4}; mkdir -p $out /tmp/isa
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude --cuda-device-only -S rl-mpc-lanemerging_amd/csrc/stmpc.hip -o /tmp/isa/stmpc.s 2>/dev/null
{
  echo "# scripts/isa/spill_report.sh: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math --cuda-device-only -S csrc/stmpc.hip, source hash $(python -c 'import rl_mpc_lanemerging_amd as p; print(p.build.source_hash())')"
  echo "# lane = v_readlane_b32 / v_writelane_b32 (SGPR spill moves), scratch = scratch_load / scratch_store (VGPR spills)"
  for k in "k_solveILb1ELb0ELb1ELi0ELi8ELb0ELi1ELi4E" "k_solveILb1ELb0ELb1ELi0ELi24ELb0ELi2ELi88E" "k_predictILi8E"; do
    echo; python scripts/isa/census.py /tmp/isa/stmpc.s $k
    echo; python scripts/isa/loops.py /tmp/isa/stmpc.s $k
  done
} > $out/spill_placement.txt
python scripts/resource_usage.py $out/resource_usage.txt > /dev/null
echo "wrote $out/spill_placement.txt $out/resource_usage.txt"
