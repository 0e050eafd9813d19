cd "${GRAFT_REPO_ROOT:-/root/repo}"
for fs in 0 1; do echo "MOE_CHOL_FUSED_STEP=$fs"; MOE_CHOL_FUSED_STEP=$fs timeout 300 python tools/ll_time.py 2>&1 | tail -8; done | tee gpurun_out/r03_d_ll_time.txt
