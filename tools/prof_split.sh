#!/bin/bash
# Runs on the GPU box: kernel trace of the headline step with the split-bf16 products (eager launches) and the labelled
# bench line itself.  usage: tools/prof_split.sh   (writes gpurun_out/final/split_*)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_split && mkdir -p /tmp/prof_split
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_split/trace -o t -- python $R/bench.py --split-bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $OUT/split_bench_under_rocprof_eager.json 2> /tmp/prof_split/trace.err
python $R/tools/kstats.py /tmp/prof_split/trace/t_results.db 16 > $OUT/split_kernel_stats_eager.txt
timeout -k 10 300 python $R/bench.py --split-bf16 --no-cpu-baseline > $OUT/split_bench_n1.json 2> /tmp/prof_split/bench.err
timeout -k 10 300 python $R/bench.py --split-bf16 --batch 128 --no-cpu-baseline > $OUT/split_bench_b128.json 2> /tmp/prof_split/bench128.err
OCEAN_GEMM_SPLIT_BF16=1 timeout -k 10 200 python -u $R/tools/bench_gemm.py 2>/dev/null > /tmp/prof_split/gemm.txt
sed -n 1,22p /tmp/prof_split/gemm.txt > $OUT/split_gemm_microbench.txt
echo collected
