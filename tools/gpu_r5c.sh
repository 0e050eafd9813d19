#!/bin/bash
# Round 5, GPU call C: the multi-rank optimiser tests + the batch-1 timeline with kernel names.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r05_c
mkdir -p $OUT
(timeout 1500 python -m pytest tests/test_gpu_multistart.py -x -q 2>&1 | tail -25) > $OUT/pytest_multistart.txt
bash tools/kg1_timeline.sh > $OUT/kg1_timeline.txt 2>&1
cat $OUT/pytest_multistart.txt $OUT/kg1_timeline.txt
