#!/bin/bash
# Runs on the GPU box: bench.py (graph replay, and eager for the small shapes) at the headline shape, batch 128 and the
# per-window shape of the streaming configuration.  usage: tools/ab_shapes.sh OUTFILE
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
: > $OUT
run() { echo "== $*" >> $OUT; timeout -k 10 200 python $R/bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['launch'])" >> $OUT; }
for rep in 1 2; do
  run --batch 1024 --steps 40 --warmup 5
  run --batch 128 --steps 60 --warmup 5
  run --batch 8 --seq-len 64 --services 64 --steps 200 --warmup 10
  run --batch 8 --seq-len 64 --services 64 --steps 200 --warmup 10 --no-graph
done
