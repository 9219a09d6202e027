"""The kernels of ONE update step in launch order, from a rocprofv3 kernel-trace database of an eager bench.py run:
name, duration, gap since the previous kernel's end.  The run must have been made with --steps S --warmup W; the last
step is printed.   python tools/step_trace.py DB STEPS_TOTAL"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2])
rows = list(db.execute("select name, start, end from kernels order by start"))
per = len(rows) // steps
last = rows[-per:]
print(f"{len(rows)} kernels, {per} per step; last step spans {(last[-1][2] - last[0][1]) / 1000:.1f} us, busy {sum(e - s for _, s, e in last) / 1000:.1f} us")
prev = None
for name, s, e in last:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
    gap = 0.0 if prev is None else (s - prev) / 1000
    print(f"{(e - s) / 1000:8.1f} us  gap {gap:6.1f}  {short}")
    prev = e
