#!/bin/bash
# Runs on the GPU box: tools/ab_shapes.sh's runs in this tree and in a second tree (an export of another commit with its
# library built), interleaved.  usage: tools/ab_trees.sh OUTFILE OTHER_TREE
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; O=$R/$2
: > $OUT
run() { t=$1; shift; echo "== $t $*" >> $OUT; (cd $t && timeout -k 10 200 python bench.py "$@" --no-cpu-baseline 2>/dev/null) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['launch'])" >> $OUT; }
for rep in 1 2; do
  for t in $O $R; do
    run $t --batch 1024 --steps 40 --warmup 5
    run $t --batch 128 --steps 60 --warmup 5
    run $t --batch 8 --seq-len 64 --services 64 --steps 200 --warmup 10
    run $t --batch 8 --seq-len 64 --services 64 --steps 200 --warmup 10 --no-graph
  done
done
