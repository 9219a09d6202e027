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
# L2 pass: hit rate of the XCD L2s and the read requests the CUs' vector caches send them (the gather kernels fetch
# close to their compulsory bytes from HBM; what they wait for is this L2 -> CU row traffic)
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d /tmp/pg2 -o g -- python $R/tools/bench_gnn.py 256 1024 3 > /tmp/pg2.log 2>&1
echo "-- L2 pass (per launch): TCC hit rate = HIT / (HIT + MISS); TCP_TCC_READ_REQ = vector-L1 -> L2 read requests" >> $R/gpurun_out/$1
python $R/tools/pmc_kernels.py /tmp/pg2/g_results.db gat_ >> $R/gpurun_out/$1
