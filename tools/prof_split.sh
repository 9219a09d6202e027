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
# the forward pass in fp32, the two gradient products split: the variant under which every parity test passes unchanged
timeout -k 10 300 python $R/bench.py --split-bf16 backward --no-cpu-baseline > $OUT/split_backward_bench_n1.json 2> /tmp/prof_split/benchb.err
(cd $R && OCEAN_GEMM_SPLIT_BF16=2 timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -p no:cacheprovider --deselect tests/test_gpu_split_bf16.py 2>&1 | grep -E "^FAILED|passed|failed" > $OUT/split_backward_gpu_tests.txt || true)
(cd $R && OCEAN_GEMM_SPLIT_BF16=1 timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -p no:cacheprovider --deselect tests/test_gpu_split_bf16.py 2>&1 | grep -E "^FAILED|passed|failed" > $OUT/split_all_gpu_tests.txt || true)
OCEAN_GEMM_SPLIT_BF16=1 timeout -k 10 200 python -u $R/tools/bench_gemm.py > /tmp/prof_split/gemm.txt 2> /tmp/prof_split/gemm.err || { tail -5 /tmp/prof_split/gemm.err; exit 1; }
sed -n 1,22p /tmp/prof_split/gemm.txt > $OUT/split_gemm_microbench.txt
(cd $R && timeout -k 10 300 python -u tools/split_training_check.py 200 2>/dev/null > $OUT/split_training.txt)
echo collected
