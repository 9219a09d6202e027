#!/bin/bash
# Runs on the GPU box: kernel trace of the data-parallel update step with ONE RCCL rank (the N > 1 code path: phased
# BatchNorm statistics, gathered positives, gradient arena all-reduce), eager launches, next to the plain step.
# usage: tools/prof_dp1.sh OUTFILE BATCH
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; B=${2:-128}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd && mkdir -p /tmp/pd
export OCEAN_BENCH_DIST=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pd -o q -- python $R/bench.py --batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-graph > /tmp/pd/bench.json 2> /tmp/pd/err.log || { tail -20 /tmp/pd/err.log; exit 1; }
python $R/tools/kstats.py /tmp/pd/q_results.db 11 > $R/gpurun_out/$OUT
python - <<PY >> $R/gpurun_out/$OUT
import sqlite3
db = sqlite3.connect("/tmp/pd/q_results.db")
n, = db.execute("select count(*) from kernels").fetchone()
print("kernels per step:", n / 11.0)
PY
cat /tmp/pd/bench.json >> $R/gpurun_out/$OUT
