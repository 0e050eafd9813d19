cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -q -x 2>&1 | tail -3 | tee gpurun_out/r03_l_pytest.txt
timeout 300 python tools/chol_time.py 3 12 2>&1 | grep two-level | tee gpurun_out/r03_l_chol_time.txt
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_chol -o chol -- python $GRAFT_REPO_ROOT/tools/chol_prof.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
head -6 gpurun_out/prof_chol/chol_kernel_stats.csv | cut -c1-150
