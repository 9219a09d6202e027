"""The kernels of ONE update step in launch order, from a rocprofv3 kernel-trace database of a bench.py run:
name, duration, gap since the previous kernel's end.  Steps are cut after the optimizer's kernel (clip_adam_kernel,
the last launch of an update); step number WHICH (0-based, negative from the end) is printed.
   python tools/step_trace.py DB WHICH"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2])
rows = list(db.execute("select name, start, end from kernels order by start"))
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "clip_adam_kernel" in r[0]:
        steps.append(cur)
        cur = []
last = steps[which]
print(f"{len(rows)} kernels, {len(steps)} steps; step {which}: {len(last)} kernels, spans {(last[-1][2] - last[0][1]) / 1000:.1f} us, busy {sum(e - s for _, s, e in last) / 1000:.1f} us")
prev = None
for name, s, e in last:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
    gap = 0.0 if prev is None else (s - prev) / 1000
    print(f"{(e - s) / 1000:8.1f} us  gap {gap:6.1f}  {short}")
    prev = e
