#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_k
mkdir -p $OUT
MOE_BUILD_TRACE=1 timeout 600 python tools/chol_time.py 3 2>&1 | grep -v "one-level" | head -5 > $OUT/chol_time.txt
cat $OUT/chol_time.txt
bash tools/build_timeline.sh 3 > $OUT/build_timeline.txt 2>&1
cat $OUT/build_timeline.txt
