#!/bin/bash
# One GPU session: tests, smoke, bench, rocprof kernel stats + HBM-traffic PMC passes -> gpurun_out/
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 600 python bench.py 2> gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof $R/gpurun_out/hbm
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o kg -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-batch1 --no-traffic > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
# HBM traffic: FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md: TCC slots; FETCH_SIZE reads 1/2 on gfx950)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/hbm/fetch -o p -- python $R/tools/prof_kg.py C3 8 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/hbm/write -o p -- python $R/tools/prof_kg.py C3 8 2 > /dev/null 2>&1
cd $R
python tools/hbm_traffic.py gpurun_out/hbm | tee gpurun_out/hbm_traffic.json
find gpurun_out/prof -name "*stats*" | head
