"""Summarise a rocprofv3 results database: per-kernel totals per step, optionally split by grid size.

    python tools/kstats.py gpurun_out/prof/x_results.db STEPS [name-substring ...]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1000.0 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total GPU busy per step: {tot / steps / 1000:.3f} ms")
for name, n, us in rows[:45]:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    short = short.split("(")[0][:70]
    print(f"{us / steps / 1000:8.3f} ms/step  {n / steps:7.1f} calls/step  {us / n:9.1f} us avg  {short}")
for pat in sys.argv[3:]:
    print("--", pat)
    q = ("select grid_x, grid_y, grid_z, count(*), avg(end-start)/1000.0, min(end-start)/1000.0 from kernels "
         "where name like ? group by grid_x, grid_y, grid_z order by 5 desc")
    for r in cur.execute(q, (f"%{pat}%",)):
        print("    grid(threads)", r[0:3], f"calls/step {r[3] / steps:.1f} avg {r[4]:.1f} us min {r[5]:.1f} us")
