cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python tools/run_configs.py > gpurun_out/r03_configs.json 2> gpurun_out/r03_configs.err
tail -3 gpurun_out/r03_configs.err
timeout 900 python bench.py --config C5 --steps 6 --warmup 2 2> gpurun_out/r03_c5_bench.err > gpurun_out/r03_c5_bench.json
tail -2 gpurun_out/r03_c5_bench.err
timeout 300 python tools/latency.py 2>&1 | tee gpurun_out/r03_latency.txt
timeout 300 python tools/small_n.py 2>&1 | tail -12 | tee gpurun_out/r03_small_n.txt
