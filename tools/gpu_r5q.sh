#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_q
mkdir -p $OUT
timeout 600 python tools/selftest_debug.py 2>&1 | tail -4 > $OUT/selftest_debug.txt
cat $OUT/selftest_debug.txt
timeout 600 python -c "
from cornell_moe_amd import selftest
print('failures', selftest.run(verbose=True))" > $OUT/selftest.txt 2>&1
cat $OUT/selftest.txt
(timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q 2>&1 | tail -5) > $OUT/pytest.txt
cat $OUT/pytest.txt
