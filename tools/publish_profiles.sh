#!/bin/bash
# Copies the summaries under gpurun_out/final/ (written by collect_profiles.sh and collect_micro.sh on the GPU box)
# into profiles/ under the given file-name prefix.  usage: tools/publish_profiles.sh r01
set -e
P=${1:-r01}
F=gpurun_out/final
cp $F/bench_n1.json profiles/${P}_bench_n1.json
cp $F/bench_n1_eager.json profiles/${P}_bench_n1_eager.json
cp $F/kernel_table.txt profiles/${P}_kernel_table.txt
cp $F/kernel_table.json profiles/${P}_kernel_table.json
cp $F/pmc_traffic.json profiles/${P}_pmc_traffic.json
cp $F/kernel_stats.txt profiles/${P}_kernel_stats_eager.txt
cp $F/kernel_stats_graph.txt profiles/${P}_kernel_stats_graph.txt
cp $F/bench_under_rocprof_eager.json profiles/${P}_bench_under_rocprof_eager.json
cp $F/bench_under_rocprof_graph.json profiles/${P}_bench_under_rocprof_graph.json
cp $F/gemm_microbench.txt profiles/${P}_gemm_microbench.txt
cp $F/gemm_pmc.txt profiles/${P}_gemm_pmc.txt
cp $F/configs.jsonl profiles/${P}_configs.jsonl
cp $F/staging.jsonl profiles/${P}_staging.jsonl
cp $F/streaming.jsonl profiles/${P}_streaming.jsonl
cp $F/metrics.jsonl profiles/${P}_metrics.jsonl
cp $F/gat_instruction_mix.txt profiles/${P}_gat_instruction_mix.txt
cp $F/gat_traffic.txt profiles/${P}_gat_traffic.txt
for f in bench_b128.json bench_dp.jsonl gemm_ksweep.txt split_bench_n1.json split_bench_b128.json split_kernel_stats_eager.txt split_bench_under_rocprof_eager.json split_gemm_microbench.txt split_backward_bench_n1.json split_backward_gpu_tests.txt split_all_gpu_tests.txt split_training.txt; do
  if [ -f $F/$f ]; then cp $F/$f profiles/${P}_$f; fi
done
