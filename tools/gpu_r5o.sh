#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_o
mkdir -p $OUT
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pool.py -x -q -k "early_inverse or pool" 2>&1 | tail -12) > $OUT/pytest.txt
cat $OUT/pytest.txt
for rep in 1 2; do
for v in "0 1" "1 1" "2 1" "1 0" "2 0"; do
  set -- $v
  echo "== MOE_CHOL_EARLY_INVERSE=$1 (0 off, 1 leading levels, 2 + top product)  MOE_CHOL_SIDE_PRIORITY=$2 (1 low)"
  MOE_CHOL_EARLY_INVERSE=$1 MOE_CHOL_SIDE_PRIORITY=$2 timeout 900 python tools/chol_time.py 3 2>&1 | grep -v "one-level"
  MOE_CHOL_EARLY_INVERSE=$1 MOE_CHOL_SIDE_PRIORITY=$2 timeout 600 python tools/build_sweep.py 4000 6000 12000 2>&1 | cut -c1-120 | tail -3
done
done > $OUT/chol_time.txt 2>&1
cat $OUT/chol_time.txt
