#!/bin/bash
# early-inverse schedule of the GP build: parity tests, build time with / without, timeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_n
mkdir -p $OUT
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pool.py -x -q -k "early_inverse or two_level or gemm128 or known_answers or pool" 2>&1 | tail -12) > $OUT/pytest.txt
cat $OUT/pytest.txt
for ei in 0 1; do
  echo "== MOE_CHOL_EARLY_INVERSE=$ei"
  MOE_CHOL_EARLY_INVERSE=$ei timeout 900 python tools/chol_time.py 3 12 2>&1 | grep -v "one-level"
  MOE_CHOL_EARLY_INVERSE=$ei timeout 600 python tools/build_sweep.py 2>&1 | tail -12
done > $OUT/chol_time.txt 2>&1
cat $OUT/chol_time.txt
bash tools/build_timeline.sh 3 > $OUT/build_timeline.txt 2>&1
grep -v "chol_colcopy\|syrk" $OUT/build_timeline.txt | head -60
