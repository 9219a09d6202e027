#!/bin/bash
# Runs on the GPU box: kernel trace of the update step at a small per-GPU batch (the strong-scaling shard), eager launches.
# usage: tools/prof_b128.sh OUTFILE BATCH
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; B=${2:-128}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq && mkdir -p /tmp/pq
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pq -o q -- python $R/bench.py --batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-graph > /tmp/pq/bench.json 2> /tmp/pq/err.log
python $R/tools/kstats.py /tmp/pq/q_results.db 11 "gemm_f32_kernel<64" "gemm_skinny" > $R/gpurun_out/$OUT
python - <<PY >> $R/gpurun_out/$OUT
import sqlite3
db = sqlite3.connect("/tmp/pq/q_results.db")
n, = db.execute("select count(*) from kernels").fetchone()
print("kernels per step:", n / 11.0)
PY
