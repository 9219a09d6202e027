#!/bin/bash
# Runs on the GPU box: HBM-side traffic per kernel of bench.py (eager launches): FETCH_SIZE and WRITE_SIZE in separate
# counter passes, merged into gpurun_out/$1.json (+ .txt).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptr && mkdir -p /tmp/ptr
timeout -s KILL 250 rocprofv3 --pmc FETCH_SIZE -d /tmp/ptr/fetch -o f -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > /tmp/ptr/fetch.log 2>&1
timeout -s KILL 250 rocprofv3 --pmc WRITE_SIZE -d /tmp/ptr/write -o w -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > /tmp/ptr/write.log 2>&1
python $R/tools/pmc_traffic.py /tmp/ptr/fetch/f_results.db /tmp/ptr/write/w_results.db $R/gpurun_out/$OUT.json > $R/gpurun_out/$OUT.txt
