"""Kernel launch order of one update step from a rocprofv3 kernel-trace database (between two clip_adam launches).
usage: python tools/kernel_sequence.py DB [which_step=3]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view') and name like '%kernels%'")]
t = "kernels" if "kernels" in tabs else tabs[0]
rows = list(db.execute(f"select name, start, end, grid_x from {t} order by start"))
marks = [i for i, r in enumerate(rows) if "clip_adam" in r[0]]
lo, hi = marks[which - 1] + 1, marks[which] + 1
prev_end = rows[lo - 1][2]
for name, start, end, gx in rows[lo:hi]:
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    short = short.split("(")[0][:90].replace(" ", "")
    print(f"{(start - prev_end) / 1000.0:7.1f} gap {(end - start) / 1000.0:8.1f} us  grid {gx:8d}  {short}")
    prev_end = end
