"""Per-shape device durations of the small-tile GEMM launches in a rocprofv3 kernel-trace database of bench.py.
usage: python tools/small_gemm_step.py DB N_STEPS"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view') and name like '%kernels%'")]
t = "kernels" if "kernels" in tabs else tabs[0]
q = (f"select substr(name, instr(name, 'gemm_f32_kernel'), 52), grid_x, grid_y, grid_z, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 "
     f"from {t} where name like '%gemm_f32_kernel<64, 64%' or name like '%gemm_f32_kernel<128, 64%' group by 1, 2, 3, 4 order by 7 desc")
tot = 0.0
for r in db.execute(q):
    tot += r[6]
    print(f"{r[0]:52s} grid {r[1]:6d} {r[2]} {r[3]:3d}  calls/step {r[4] / steps:5.1f}  avg {r[5]:6.1f} us  us/step {r[6] / steps:7.1f}")
print(f"total {tot / steps:.1f} us/step")
