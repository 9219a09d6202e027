#!/bin/bash
# Runs on the GPU box: replayed step time at the headline shape, at one GPU's share of it on 8 GPUs, and at the streaming
# configuration's per-window update.  usage: tools/quick_bench.sh OUTFILE
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
: > $OUT
for cfg in "1024 128 256" "128 128 256" "8 64 64"; do
  set -- $cfg
  timeout -k 10 120 python $R/bench.py --batch $1 --seq-len $2 --services $3 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=%d T=%d N=%d: %.3f ms/step  %.0f samples/s  roofline %.3f (%s)' % ($1,$2,$3,d['ms_per_step'],d['value'],d['roofline']['frac'],d['config']['launch']))" >> $OUT
done
cat $OUT
