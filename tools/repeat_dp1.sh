#!/bin/bash
# Runs on the GPU box: the one-rank data-parallel bench (RCCL process group, collectives recorded into the step's HIP graph)
# N rounds of (batch 128 replayed, batch 128 eager, batch 1024 replayed), stopping at the first run that fails; its stderr
# is shown.  usage: tools/repeat_dp1.sh N
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1
for i in $(seq 1 $N); do
  for args in "--batch 128" "--batch 128 --no-graph" ""; do
    OCEAN_BENCH_DIST=1 timeout -k 10 200 python $R/bench.py --no-cpu-baseline $args > $R/gpurun_out/dp1_rep.json 2> $R/gpurun_out/dp1_rep.err
    rc=$?
    echo "round $i [$args] rc=$rc $(cut -c100-140 $R/gpurun_out/dp1_rep.json)"
    if [ $rc -ne 0 ]; then tail -n 60 $R/gpurun_out/dp1_rep.err | cut -c1-400; exit $rc; fi
  done
done
