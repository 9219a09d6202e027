#!/bin/bash
# Runs on the GPU box: ordered kernel list of one eagerly launched update step.  usage: tools/step_trace.sh OUTFILE BATCH SEQ SERVICES
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; B=${2:-8}; T=${3:-64}; S=${4:-64}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st && mkdir -p /tmp/st
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/st -o q -- python $R/bench.py --batch $B --seq-len $T --services $S --steps 4 --warmup 2 --no-cpu-baseline --no-graph > /tmp/st/bench.json 2> /tmp/st/err.log || { tail -20 /tmp/st/err.log; exit 1; }
python $R/tools/step_trace.py /tmp/st/q_results.db 6 > $R/gpurun_out/$OUT
cat /tmp/st/bench.json >> $R/gpurun_out/$OUT
