cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "1 0" "1 1" "1 0" "1 1"; do
  set -- $cfg
  echo "MOE_CHOL_FUSED_STEP=$1 MOE_CHOL_SYRK_OVERLAP=$2"
  MOE_CHOL_FUSED_STEP=$1 MOE_CHOL_SYRK_OVERLAP=$2 timeout 300 python tools/chol_time.py 3 2>&1 | grep two-level
done | tee gpurun_out/r03_c2_chol_time.txt
