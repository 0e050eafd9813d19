#!/bin/bash
# One GPU session: tests, smoke, bench (which runs its own PMC passes), the same bench under rocprofv3 kernel stats -> gpurun_out/<tag>_*
# usage: tools/gpu_round.sh <tag> [notests]
set -x
TAG="${1:-r03}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/${TAG}_pytest_gpu.txt
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a gpurun_out/${TAG}_pytest_gpu.txt
fi
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tee gpurun_out/${TAG}_bench.json
tail -5 gpurun_out/${TAG}_bench.err
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o kg -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras --no-batch1 --no-determinism > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof.err
cd $R
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
head -8 gpurun_out/${TAG}_bench_kernel_stats.csv
