#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases written by profiles/run_profile.sh into a text file."""
import glob, os, sqlite3, sys
out = sys.argv[1]
print("# rocprofv3 summary of", out)
tr = glob.glob(os.path.join(out, "trace", "*.db"))
if tr:
    cur = sqlite3.connect(tr[0]).cursor()
    print("# kernel-trace --stats (us)\nname,total_calls,total_duration_us,average_us,percentage")
    for r in cur.execute("select * from top_kernels"):
        print(",".join(str(x) for x in r))
for name in sorted(glob.glob(os.path.join(out, "pmc*"))):
    dbs = glob.glob(os.path.join(name, "*.db"))
    if not dbs:
        continue
    cur = sqlite3.connect(dbs[0]).cursor()
    print("# --pmc pass", os.path.basename(name), "(kernel,counter,dispatches,average per dispatch)")
    for r in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if "rocclr" in r[0]:
            continue
        print("%s,%s,%d,%.1f" % (r[0].split("(")[0], r[1], r[2], r[3]))
