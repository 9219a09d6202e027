#!/bin/bash
# Runs on the GPU box: bench.py (graph replay) at a given batch under each SET of environment settings given (a set is
# one argument, settings separated by commas), twice, interleaved.
# usage: tools/ab_env2.sh OUTFILE BATCH "A=1,B=2" "A=0" ...
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; B=$2; shift; shift
: > $OUT
for rep in 1 2; do
  for kv in "$@"; do
    echo "== $kv" >> $OUT
    env ${kv//,/ } timeout -k 10 200 python $R/bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline 2>>$OUT.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['all_gemm_ms_per_step'])" >> $OUT
  done
done
