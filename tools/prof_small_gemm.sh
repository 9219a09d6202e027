#!/bin/bash
# Runs on the GPU box: device-side durations of the small (64x64-tile) GEMM launches of tools/bench_gemm.py for the
# environment given on the command line.  usage: tools/prof_small_gemm.sh OUTFILE [VAR=value ...]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/psg
timeout -k 10 100 rocprofv3 --kernel-trace --stats -d /tmp/psg -o s -- python $R/tools/bench_gemm.py 1024 128 > /tmp/psg.log 2>&1
python - <<PY >> $R/gpurun_out/$OUT
import sqlite3
db = sqlite3.connect("/tmp/psg/s_results.db")
print("== $*")
q = ("select substr(name, instr(name, 'gemm_f32_kernel'), 48), grid_x, grid_y, grid_z, count(*), avg(end-start)/1000.0, min(end-start)/1000.0 "
     "from kernels where name like '%gemm_f32_kernel<64, 64%' group by 1, 2, 3, 4 order by 1, 2, 4")
for r in db.cursor().execute(q):
    print(f"{r[0]:48s} grid {r[1]:6d} {r[2]} {r[3]:3d}  n={r[4]:3d}  avg {r[5]:6.1f} us  min {r[6]:6.1f} us")
PY
