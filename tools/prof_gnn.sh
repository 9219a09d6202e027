#!/bin/bash
# Runs on the GPU box: GNN micro-benchmark, then an instruction-mix counter pass over it (raw database stays in /tmp).
# usage: tools/prof_gnn.sh OUTFILE
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg
timeout -k 10 100 python $R/tools/bench_gnn.py > $R/gpurun_out/$1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pg -o g -- python $R/tools/bench_gnn.py 256 1024 3 > /tmp/pg.log 2>&1
python $R/tools/pmc_kernels.py /tmp/pg/g_results.db gat_ >> $R/gpurun_out/$1
