#!/bin/bash
# device-memory pool: GP build time with and without it, host-clock split of a build, then the whole GPU suite on the pool
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_j
mkdir -p $OUT
for pool in 0 1; do
  echo "== MOE_POOL=$pool"
  MOE_POOL=$pool MOE_BUILD_TRACE=1 timeout 600 python tools/chol_time.py 3 2>&1 | grep -v "one-level" | tail -8
done > $OUT/chol_time.txt 2>&1
cat $OUT/chol_time.txt
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $OUT/pytest.txt
cat $OUT/pytest.txt
