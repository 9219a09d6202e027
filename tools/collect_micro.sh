#!/bin/bash
# Runs on the GPU box: headline bench line, GEMM micro-benchmark (+ matrix-pipe counters), secondary configurations.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python $R/bench.py > $OUT/bench_n1.json 2> /tmp/bench_n1.err
timeout -k 10 200 python $R/bench.py --no-graph --no-cpu-baseline > $OUT/bench_n1_eager.json 2> /tmp/bench_n1e.err
timeout -k 10 200 python -u $R/tools/bench_gemm.py 2>/dev/null > $OUT/gemm_microbench.txt
rm -rf /tmp/pmc_gemm
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc_gemm -o p -- python $R/tools/bench_gemm.py 1024 128 > /tmp/pmc_gemm.log 2>&1
python $R/tools/pmc_gemm.py /tmp/pmc_gemm/p_results.db > $OUT/gemm_pmc.txt
timeout -k 10 400 python -u $R/tools/bench_configs.py 2>/dev/null > $OUT/configs.jsonl
timeout -k 10 300 python -u $R/tools/bench_staging.py 2>/dev/null > $OUT/staging.jsonl
timeout -k 10 300 python -u $R/tools/bench_streaming.py 2>/dev/null > $OUT/streaming.jsonl
timeout -k 10 300 python -u $R/tools/bench_metrics.py 2>/dev/null > $OUT/metrics.jsonl
bash $R/tools/prof_gnn.sh final/gat_instruction_mix.txt
bash $R/tools/prof_gnn_traffic.sh final/gat_traffic.txt
echo collected
