#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`).
usage: rocpd_summary.py results.db [steps] [top]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = list(db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot / 1e3:.2f} ms over {steps} steps = {tot / steps / 1e3:.3f} ms/step")
print(f"{'share':>6} {'calls/step':>10} {'avg us':>9} {'ms/step':>8}  kernel")
for name, cnt, total, avg in rows[:top]:
    short = re.sub(r"\(.*", "", name)[:80]
    print(f"{total / tot * 100:5.1f}% {cnt / steps:10.1f} {avg:9.1f} {total / steps / 1e3:8.3f}  {short}")
