#!/usr/bin/env python3
"""Compile csrc/stmpc.hip with -Rpass-analysis=kernel-resource-usage and print one row per kernel
(registers, spills, scratch, occupancy, static LDS).  Usage: scripts/resource_usage.py [out.txt] [extra hipcc flags...]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import rl_mpc_lanemerging_amd as pkg

def main():
    out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else None
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    b = pkg.build
    cmd = [b.find_hipcc()] + b.HIPCC_FLAGS + extra + ["-I" + os.path.join(REPO, "include"), os.path.join(b.CSRC, "stmpc.hip"),
           "-o", "/tmp/libstmpc_ru.so", "-Rpass-analysis=kernel-resource-usage"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in txt.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    lines = ["%-88s %5s %5s %6s %6s %8s %4s %7s" % ("kernel", "SGPR", "VGPR", "sSpill", "vSpill", "scratchB", "occ", "LDS_B")]
    for r, nm in zip(rows, names):
        nm = re.sub(r"\(.*\)$", "", nm).replace("stmpc::", "").replace("void ", "")
        lines.append("%-88s %5s %5s %6s %6s %8s %4s %7s" % (nm[:88], r.get("TotalSGPRs", r.get("SGPRs")), r.get("VGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
                                                           r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
    s = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write("# hipcc " + " ".join(b.HIPCC_FLAGS + extra) + " -Rpass-analysis=kernel-resource-usage\n# k_solve<USE_LDS, GRID, FASTDIV, KT, FANMAX, S1GEN, RES>\n" + s)
    sys.stdout.write(s)

if __name__ == "__main__":
    main()
