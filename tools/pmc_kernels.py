"""Per-kernel summary of one rocprofv3 --pmc results database: every collected counter averaged per launch, plus
VALU issue occupancy when SQ_INSTS_VALU and GRBM_GUI_ACTIVE are present.

    python tools/pmc_kernels.py results.db [name-substring]
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = f"%{sys.argv[2]}%" if len(sys.argv) > 2 else "%"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
dur = collections.defaultdict(float)
q = ("select dispatch_id, kernel_name, counter_name, sum(value), max(duration) from counters_collection where kernel_name like ? "
     "group by dispatch_id, counter_name")
for did, name, cn, val, d in db.cursor().execute(q, (pat,)):
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    acc[short][cn] += val
    if did not in cnt[short]:
        cnt[short].add(did)
        dur[short] += d
for k in sorted(acc, key=lambda k: -dur[k]):
    n = len(cnt[k])
    v = {c: x / n for c, x in acc[k].items()}
    line = [f"{k:48s} n={n:4d} {dur[k] / n / 1000:8.1f} us"]
    gui = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if gui and "SQ_INSTS_VALU" in v:
        line.append(f"valu-issue {v['SQ_INSTS_VALU'] * 4 / (gui * 1024):.2f}")
    if "SQ_WAVES" in v and v["SQ_WAVES"]:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
            if c in v:
                line.append(f"{c[9:].lower()}/wave {v[c] / v['SQ_WAVES']:.0f}")
    for c, x in sorted(v.items()):
        if c.startswith("SQ_INSTS") or c in ("SQ_WAVES",):
            continue
        line.append(f"{c} {x:.4g}")
    print("  ".join(line))
