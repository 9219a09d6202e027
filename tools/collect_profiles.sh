#!/bin/bash
# Runs on the GPU box: kernel trace + three counter passes of bench.py (eager launches, so every kernel is a separate
# dispatch), merged into small files under gpurun_out/final/ (the raw databases stay in /tmp).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_final && mkdir -p /tmp/prof_final
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_final/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $OUT/bench_under_rocprof_eager.json 2> /tmp/prof_final/trace.err
timeout -s KILL 250 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_final/mfma -o m -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > /tmp/prof_final/mfma.log 2>&1
timeout -s KILL 250 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_final/fetch -o f -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > /tmp/prof_final/fetch.log 2>&1
timeout -s KILL 250 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_final/write -o w -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > /tmp/prof_final/write.log 2>&1
# steps seen by the profiler: warmup + timed + the 3 eager steps bench.py adds afterwards to time the roofline kernel
python $R/tools/profile_table.py /tmp/prof_final/trace/t_results.db /tmp/prof_final/mfma/m_results.db /tmp/prof_final/fetch/f_results.db /tmp/prof_final/write/w_results.db 16 9 $OUT/kernel_table.json > $OUT/kernel_table.txt
python $R/tools/pmc_traffic.py /tmp/prof_final/fetch/f_results.db /tmp/prof_final/write/w_results.db $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
python $R/tools/kstats.py /tmp/prof_final/trace/t_results.db 16 gemm_f32_kernel > $OUT/kernel_stats.txt
# the same command with HIP-graph replay (the default): trace only
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_final/gtrace -o g -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof_graph.json 2> /tmp/prof_final/gtrace.err
python $R/tools/kstats.py /tmp/prof_final/gtrace/g_results.db 16 > $OUT/kernel_stats_graph.txt || true
echo collected
