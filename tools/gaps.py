"""Idle gaps between consecutive kernels in a rocprofv3 kernel trace (one stream): where the GPU waited for the host.

    python tools/gaps.py results.db STEPS [min_gap_us]
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
rows = list(db.cursor().execute("select name, start, end from kernels order by start"))
# keep whole update steps only: the span between the first and the last Adam kernel (set-up and tear-down excluded)
marks = [i for i, r in enumerate(rows) if "clip_adam_kernel" in r[0]]
if len(marks) >= 2:
    rows = rows[marks[0]:marks[-1] + 1]
    steps = float(len(marks) - 1)
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
gaps = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    g = (s1 - e0) / 1000.0
    if g > thr and g < 5000.0:          # longer pauses are between phases of the program, not launch gaps
        k = (short(n0), short(n1))
        gaps[k][0] += 1
        gaps[k][1] += g
        tot += g
print(f"idle in gaps > {thr} us: {tot / steps / 1000:.3f} ms/step")
for (a, b), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{g / steps:8.1f} us/step  {c / steps:5.1f}x  {g / c:7.1f} us  after {a:48s} before {b}")
