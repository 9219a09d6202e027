#!/bin/bash
# Runs on the GPU box: the convolution micro-benchmark and the step with and without the k-contiguous-operand kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_gemm.py tests/test_gpu_dcnn.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_graph_step.py -x -q --timeout 200 --timeout-method thread > gpurun_out/ab_nt_tests.txt 2>&1 || { tail -30 gpurun_out/ab_nt_tests.txt; exit 1; }
tail -2 gpurun_out/ab_nt_tests.txt
timeout -k 10 120 python -u tools/bench_gemm.py 2>/dev/null | head -22 > gpurun_out/ab_nt_on.txt || exit 1
OCEAN_GEMM_NO_NT=1 timeout -k 10 120 python -u tools/bench_gemm.py 2>/dev/null | head -22 > gpurun_out/ab_nt_off.txt || exit 1
paste gpurun_out/ab_nt_on.txt gpurun_out/ab_nt_off.txt | cut -c1-140
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_nt_bench_on.json 2>/dev/null || exit 1
OCEAN_GEMM_NO_NT=1 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/ab_nt_bench_off.json 2>/dev/null || exit 1
python - <<PY
import json
for f in ("on", "off"):
    d = json.loads(open("gpurun_out/ab_nt_bench_%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["all_gemm_ms_per_step"])
PY
