"""Merge one rocprofv3 kernel trace and three --pmc passes (MFMA busy, FETCH_SIZE, WRITE_SIZE) of the same command
into a per-kernel table: time per step, matrix-pipe utilisation, HBM-side bytes and achieved GB/s.

    python tools/profile_table.py trace.db mfma.db fetch.db write.db TRACE_STEPS PMC_STEPS out.json

Counters are matched by kernel name and grid size (durations under counter collection are not used: the trace
run provides them).  FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes); both sizes are KiB.
"""
import collections
import json
import sqlite3
import sys

trace, mfma, fetch, write = sys.argv[1:5]
tsteps, psteps = float(sys.argv[5]), float(sys.argv[6])


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def counters(path, wanted):
    cur = sqlite3.connect(path).cursor()
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    q = "select dispatch_id, kernel_name, grid_size, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name"
    for did, name, grid, cn, val in cur.execute(q):
        if cn in wanted:
            k = (short(name), grid)
            acc[k][cn] += val
            n[k].add(did)
    return acc, {k: len(v) for k, v in n.items()}


cur = sqlite3.connect(trace).cursor()
times = collections.defaultdict(lambda: [0, 0.0])
for name, grid, cnt, tot in cur.execute("select name, grid_x * grid_y * grid_z, count(*), sum(end - start) from kernels group by name, grid_x * grid_y * grid_z"):
    k = (short(name), grid)
    times[k][0] += cnt
    times[k][1] += tot
m, mn = counters(mfma, {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
f, fn = counters(fetch, {"FETCH_SIZE"})
w, wn = counters(write, {"WRITE_SIZE"})

by_name = collections.defaultdict(lambda: {"calls_per_step": 0.0, "us_per_step": 0.0, "mfma_busy": 0.0, "simd_cycles": 0.0,
                                           "fetch_bytes_per_step": 0.0, "write_bytes_per_step": 0.0})
for k, (cnt, ns) in times.items():
    r = by_name[k[0]]
    r["calls_per_step"] += cnt / tsteps
    r["us_per_step"] += ns / 1000.0 / tsteps
    if k in m:
        r["mfma_busy"] += m[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / psteps
        r["simd_cycles"] += m[k]["GRBM_GUI_ACTIVE"] / 8.0 * 1024 / psteps      # GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
    if k in f:
        r["fetch_bytes_per_step"] += 2.0 * f[k]["FETCH_SIZE"] * 1024.0 / psteps
    if k in w:
        r["write_bytes_per_step"] += w[k]["WRITE_SIZE"] * 1024.0 / psteps
rows = []
for name, r in sorted(by_name.items(), key=lambda kv: -kv[1]["us_per_step"]):
    bytes_ = r["fetch_bytes_per_step"] + r["write_bytes_per_step"]
    rows.append({"kernel": name, "calls_per_step": round(r["calls_per_step"], 2), "us_per_step": round(r["us_per_step"], 1),
                 "mfma_util": round(r["mfma_busy"] / r["simd_cycles"], 3) if r["simd_cycles"] and r["mfma_busy"] else None,
                 "hbm_fetch_MB_per_step": round(r["fetch_bytes_per_step"] / 1e6, 1), "hbm_write_MB_per_step": round(r["write_bytes_per_step"] / 1e6, 1),
                 "hbm_GBps": round(bytes_ / (r["us_per_step"] * 1e-6) / 1e9, 0) if r["us_per_step"] > 0 else None})
tot = sum(r["us_per_step"] for r in rows)
print(f"GPU busy per step: {tot / 1000:.3f} ms")
print(f"{'kernel':58s} {'calls':>6s} {'us/step':>9s} {'mfma':>6s} {'fetch MB':>9s} {'write MB':>9s} {'GB/s':>6s}")
for r in rows[:40]:
    print(f"{r['kernel'][:58]:58s} {r['calls_per_step']:6.1f} {r['us_per_step']:9.1f} {str(r['mfma_util'] or ''):>6s} {r['hbm_fetch_MB_per_step']:9.1f} "
          f"{r['hbm_write_MB_per_step']:9.1f} {str(int(r['hbm_GBps'])) if r['hbm_GBps'] else '':>6s}")
json.dump({"gpu_busy_ms_per_step": tot / 1000, "kernels": rows}, open(sys.argv[7], "w"), indent=1)
