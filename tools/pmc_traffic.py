"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the other).

    python tools/pmc_traffic.py fetch_results.db write_results.db [out.json]

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 bytes, so fetched bytes are doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and
float atomics.  bn_apply_fwd_kernel (reads u and the residual, writes the activation: known byte counts) is
printed as the calibration row.
"""
import collections
import json
import sqlite3
import sys


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    q = ("select dispatch_id, kernel_name, grid_size, sum(value) from counters_collection where counter_name = ? "
         "group by dispatch_id")
    acc = collections.defaultdict(lambda: [0, 0.0])
    for _, name, grid, val in cur.execute(q, (counter,)):
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        k = (short, grid)
        acc[k][0] += 1
        acc[k][1] += val
    return acc


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1])):
    nf, f = fetch.get(k, [0, 0.0])
    nw, w = write.get(k, [0, 0.0])
    n = max(nf, nw)
    rows.append({"kernel": k[0], "grid_threads": k[1], "launches": n,
                 "fetch_bytes_per_launch": 2.0 * f * 1024.0 / max(nf, 1),      # x2: gfx950 FETCH_SIZE correction
                 "write_bytes_per_launch": w * 1024.0 / max(nw, 1)})
for r in rows[:40]:
    print(f"{r['kernel'][:60]:60s} grid {r['grid_threads']:>9d} x{r['launches']:<4d} fetch {r['fetch_bytes_per_launch'] / 1e6:9.1f} MB  "
          f"write {r['write_bytes_per_launch'] / 1e6:9.1f} MB")
if len(sys.argv) > 3:
    # bench.py quotes a figure from this file only when it was recorded with the kernel library and the bench script
    # that are running: both hashes are stored with the rows
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def sha16(path):
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]

    rec = {"lib_sha16": sha16(os.path.join(root, "ocean-online-multi-modal-root-cause-analysis_amd", "ocean", "libocean_hip.so")),
           "bench_sha16": sha16(os.path.join(root, "bench.py")),
           "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph",
           "kernels": rows}
    json.dump(rec, open(sys.argv[3], "w"), indent=1)
