#!/bin/bash
# Round-4 evidence run: full GPU suite, smoke, bench (C3 with every extra, C5, C5 g = 12), the same bench under rocprofv3 kernel stats,
# per-config table, latency, fuzz.   usage: tools/gpu_round4.sh <tag>
set -x
TAG="${1:-r04_final}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/${TAG}_pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/${TAG}_pytest_gpu.txt
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tee gpurun_out/${TAG}_bench.json | cut -c1-400
tail -4 gpurun_out/${TAG}_bench.err
timeout 900 python bench.py --config C5 --steps 6 --warmup 2 --no-cpu-baseline 2> gpurun_out/${TAG}_c5_bench.err | tee gpurun_out/${TAG}_c5_bench.json | cut -c1-300
timeout 900 python bench.py --config C5 --derivs 12 --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/${TAG}_c5g12_bench.err | tee gpurun_out/${TAG}_c5g12_bench.json | cut -c1-300
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o kg -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras --no-batch1 --no-determinism > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof.err
cd $R
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
head -6 gpurun_out/${TAG}_bench_kernel_stats.csv
rm -rf gpurun_out/prof
timeout 600 python tools/run_configs.py > gpurun_out/${TAG}_configs.json 2> gpurun_out/${TAG}_configs.err
tail -30 gpurun_out/${TAG}_configs.json
timeout 200 python tools/latency.py 2>&1 | tee gpurun_out/${TAG}_latency.txt
timeout 200 python tools/ei_loop.py 300 2>&1 | tee -a gpurun_out/${TAG}_latency.txt
timeout 300 python tools/dkg_sweep.py 2>&1 | tee gpurun_out/${TAG}_dkg_sweep.txt
timeout 200 python tools/chol_time.py 2>&1 | tee gpurun_out/${TAG}_chol_time.txt
timeout 200 python tools/ll_time.py 2>&1 | tee gpurun_out/${TAG}_ll_time.txt
