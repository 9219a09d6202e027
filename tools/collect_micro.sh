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
# one GPU's share of the global batch on 8 GPUs, the plain step and the data-parallel code path with one RCCL rank
# (recorded with the collectives inside the graph, and eager), then the same at the full batch
timeout -k 10 200 python $R/bench.py --batch 128 --no-cpu-baseline > $OUT/bench_b128.json 2> /tmp/bench_b128.err
: > $OUT/bench_dp.jsonl
OCEAN_BENCH_DIST=1 timeout -k 10 200 python $R/bench.py --batch 128 --no-cpu-baseline >> $OUT/bench_dp.jsonl 2> $OUT/bench_dp1.err
OCEAN_BENCH_DIST=1 timeout -k 10 200 python $R/bench.py --batch 128 --no-cpu-baseline --no-graph >> $OUT/bench_dp.jsonl 2> $OUT/bench_dp2.err
OCEAN_BENCH_DIST=1 timeout -k 10 200 python $R/bench.py --no-cpu-baseline >> $OUT/bench_dp.jsonl 2> $OUT/bench_dp3.err
# two ranks on this one GPU with host-staged gloo collectives: the N > 1 control flow only, its rate means nothing
OCEAN_BENCH_REHEARSE=1 timeout -k 10 300 python $R/bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline >> $OUT/bench_dp.jsonl 2> $OUT/bench_dp4.err
timeout -k 10 200 python -u $R/tools/bench_gemm_ksweep.py 2>/dev/null > $OUT/gemm_ksweep.txt
bash $R/tools/prof_gnn.sh final/gat_instruction_mix.txt
bash $R/tools/prof_gnn_traffic.sh final/gat_traffic.txt
echo collected
