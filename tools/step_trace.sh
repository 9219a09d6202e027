#!/bin/bash
# Runs on the GPU box: ordered kernel list of one update step.  usage: tools/step_trace.sh OUTFILE BATCH SEQ SERVICES [graph]
# Without "graph" the step is launched eagerly (gaps are host time); with it the step printed is a HIP-graph replay.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; B=${2:-8}; T=${3:-64}; S=${4:-64}; MODE=${5:-eager}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st && mkdir -p /tmp/st
if [ "$MODE" = graph ]; then FLAG=""; WHICH=5; else FLAG="--no-graph"; WHICH=-1; fi
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/st -o q -- python $R/bench.py --batch $B --seq-len $T --services $S --steps 4 --warmup 3 --no-cpu-baseline $FLAG > /tmp/st/bench.json 2> /tmp/st/err.log || { tail -20 /tmp/st/err.log; exit 1; }
python $R/tools/step_trace.py /tmp/st/q_results.db $WHICH > $R/gpurun_out/$OUT
cat /tmp/st/bench.json >> $R/gpurun_out/$OUT
