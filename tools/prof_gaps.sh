#!/bin/bash
# Runs on the GPU box: kernel trace of a short eager bench, then the idle-gap summary.  usage: tools/prof_gaps.sh OUTFILE
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq && mkdir -p /tmp/pq
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pq -o q -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > /tmp/pq/bench.json 2> /tmp/pq/err.log
python $R/tools/gaps.py /tmp/pq/q_results.db 8 4 > $R/gpurun_out/$1
