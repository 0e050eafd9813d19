#!/bin/bash
# the GPU suite with the round's new defaults switched OFF one at a time (the fall-back paths stay healthy)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_robust
mkdir -p $O
for v in "MOE_POOL=0" "MOE_CHOL_EARLY_INVERSE=0 MOE_KG_ZC_IN_FINISH=0 MOE_KG_GRAM_IN_STATE=0 MOE_KG_SMALL_LANE_MAX_SAMPLES=0"; do
  echo "== $v"
  env $v timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4
done > $O/pytest_switches_off.txt 2>&1
cat $O/pytest_switches_off.txt
