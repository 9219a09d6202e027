#!/bin/bash
# Runs on the GPU box: kernel trace of a short eager bench; prints the per-kernel summary (raw database stays in /tmp).
# usage: tools/prof_quick.sh OUTFILE [name-substring ...]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq && mkdir -p /tmp/pq
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pq -o q -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > /tmp/pq/bench.json 2> /tmp/pq/err.log
python $R/tools/kstats.py /tmp/pq/q_results.db 11 "$@" > $R/gpurun_out/$OUT
