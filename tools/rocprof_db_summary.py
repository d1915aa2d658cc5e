#!/usr/bin/env python3
"""Per-kernel summary (count, mean / min / max duration in us) of a rocprofv3 --kernel-trace SQLite output.
usage: python3 tools/rocprof_db_summary.py <results.db> [name-filter] [--csv]"""
import sqlite3
import sys

db = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, "
     f"max(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
if "--csv" in sys.argv:
    print("kernel,calls,avg_us,min_us,max_us")
for name, n, avg, mn, mx in c.execute(q):
    if flt and flt not in name:
        continue
    if "--csv" in sys.argv:
        print(f"\"{name}\",{n},{avg:.2f},{mn:.2f},{mx:.2f}")
    else:
        print(f"{avg:10.1f} us (min {mn:8.1f} max {mx:8.1f}) x{n:4d}  {name[:120]}")
