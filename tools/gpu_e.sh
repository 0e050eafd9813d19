cd "${GRAFT_REPO_ROOT:-/root/repo}"
MOE_CHOL_TWO_LEVEL_MIN=128 timeout 300 python tools/gp_build_time.py C2 2>&1 | tail -12
