#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_f
mkdir -p $OUT
(timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -s -k "hundreds or hyperparameter_optimisers or golden" 2>&1 | tail -25) > $OUT/pytest.txt
cat $OUT/pytest.txt
