#!/bin/bash
# Runs on the GPU box: HBM-side fetch / write bytes per GAT kernel launch of the GNN micro-benchmark (two counter passes).
# usage: tools/prof_gnn_traffic.sh OUTFILE
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pgf /tmp/pgw
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d /tmp/pgf -o f -- python $R/tools/bench_gnn.py 256 1024 3 > /tmp/pgf.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d /tmp/pgw -o w -- python $R/tools/bench_gnn.py 256 1024 3 > /tmp/pgw.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pgf/f_results.db /tmp/pgw/w_results.db /tmp/pg_traffic.json > $R/gpurun_out/$1
