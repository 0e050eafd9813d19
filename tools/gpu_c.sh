cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tools/bin/diagbench 2>&1 | grep "rep 4"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -q -x 2>&1 | tail -3 | tee gpurun_out/r03_k_pytest.txt
timeout 300 python tools/chol_time.py 3 2>&1 | grep two-level | tee gpurun_out/r03_k_chol_time.txt
timeout 300 python tools/ll_time.py 2>&1 | grep "N=8000" | tee gpurun_out/r03_k_ll_time.txt
