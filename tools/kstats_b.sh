#!/bin/bash
# Runs on the GPU box: per-kernel totals of the eagerly launched update step at a given batch.  usage: tools/kstats_b.sh OUTFILE BATCH
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; B=${2:-128}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kb && mkdir -p /tmp/kb
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/kb -o q -- python $R/bench.py --batch $B --steps 8 --warmup 2 --no-cpu-baseline --no-graph > /tmp/kb/bench.json 2> /tmp/kb/err.log || { tail -20 /tmp/kb/err.log; exit 1; }
python $R/tools/kstats.py /tmp/kb/q_results.db 10 gemm_f32 gemm_nt gemm_tn > $R/gpurun_out/$OUT
