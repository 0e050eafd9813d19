#!/bin/bash
# on-the-fly-weights instantiation of the lane-parked MC kernel: bit identity + A/B at the headline shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_i
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_sweep.py -x -q -k "fly_kernel or lane_kernel" 2>&1 | tail -15) > $OUT/pytest.txt
cat $OUT/pytest.txt
for rep in 1 2; do
  for v in "0 16" "1 16" "1 12" "1 8"; do
    set -- $v
    echo "== MOE_KG_ONFLY=$1 MOE_KG_FLY_WAVES=$2"
    MOE_KG_ONFLY=$1 MOE_KG_FLY_WAVES=$2 timeout 300 python tools/prof_kg.py C3 64 4 2>&1 | grep -v "cov_build probe" | tail -2
  done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
