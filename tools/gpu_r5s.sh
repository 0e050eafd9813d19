#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_s
mkdir -p $OUT
(timeout 2400 python -m pytest tests -x -q -m gpu -s 2>&1 | grep -a "DEBUG\|passed\|failed" | tail -20) > $OUT/e1.txt
cat $OUT/e1.txt
