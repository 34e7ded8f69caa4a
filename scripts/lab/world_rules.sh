#!/bin/bash
# Row f3's world against the reference's reported rows under each junction rule (STMPC_SIM_YIELD_OVERLAP 0-3) and both ego routes:
# scripts/lab/crash_probe.py (pure ST controller, 2048 episodes per headway) and scripts/lab/combined_episodes.py (combined controller and the
# actor alone, 1024 episodes per row).  usage (through gpurun): scripts/lab/world_rules.sh > gpurun_out/world_rules.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
fmt='
import sys,json
for line in sys.stdin:
    name,js=line.split(" ",1); d=json.loads(js)
    for k in ("here","reference"):
        print(name.ljust(24),k.ljust(10)," ".join("%s %.3f"%(a[:16],b) for a,b in d[k].items()))
'
for cfg in $(echo ${WORLD_CFGS:-lane:2,lane:3,lane:4,lane:1,lane:0,straight:0} | tr ',' ' '); do
  set -- ${cfg%%:*} ${cfg##*:}
  echo "== ego route $1, junction rule $2"
  STMPC_SIM_ROUTE=$1 STMPC_SIM_YIELD_OVERLAP=$2 python scripts/lab/crash_probe.py 2048 2>&1 | grep -A1 "^interval" | grep -v "^--" | cut -c1-560
  STMPC_SIM_ROUTE=$1 STMPC_SIM_YIELD_OVERLAP=$2 python scripts/lab/combined_episodes.py 1024 2>&1 | grep -v amdgpu.ids | python -c "$fmt"
done
