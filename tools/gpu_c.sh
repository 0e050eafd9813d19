cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tools/bin/diagbench 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -q -x 2>&1 | tail -5 | tee gpurun_out/r03_c_pytest.txt
for cfg in "0" "1"; do
  echo "MOE_CHOL_FUSED_STEP=$cfg"
  MOE_CHOL_FUSED_STEP=$cfg timeout 300 python tools/chol_time.py 3 2>&1 | grep two-level
done | tee gpurun_out/r03_c_chol_time.txt
timeout 300 python tools/chol_time.py 12 2>&1 | grep two-level | tee -a gpurun_out/r03_c_chol_time.txt
timeout 300 python tools/ll_time.py 2>&1 | tail -8 | tee gpurun_out/r03_c_ll_time.txt
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_chol -o chol -- python $GRAFT_REPO_ROOT/tools/chol_prof.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
head -8 gpurun_out/prof_chol/chol_kernel_stats.csv | cut -c1-160 | tee gpurun_out/r03_c_chol_kernel_stats.csv
