#!/bin/bash
# Runs on the GPU box: tests, the weight-gradient micro-benchmark and the step with and without the lean row-reduction kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_gemm.py tests/test_gpu_dcnn.py tests/test_gpu_model.py tests/test_gpu_gnn.py -x -q --timeout 200 --timeout-method thread > gpurun_out/ab_tn_tests.txt 2>&1 || { tail -30 gpurun_out/ab_tn_tests.txt; exit 1; }
tail -1 gpurun_out/ab_tn_tests.txt
timeout -k 10 120 python -u tools/bench_gemm.py 2>/dev/null > gpurun_out/ab_tn_on.txt || exit 1
OCEAN_GEMM_NO_TN=1 timeout -k 10 120 python -u tools/bench_gemm.py 2>/dev/null > gpurun_out/ab_tn_off.txt || exit 1
paste gpurun_out/ab_tn_on.txt gpurun_out/ab_tn_off.txt | grep wgrad | cut -c1-130 | sed -n 1,7p
bash tools/ab_env.sh ab_tn.txt OCEAN_DUMMY=1 OCEAN_GEMM_NO_TN=1
cat gpurun_out/ab_tn.txt
