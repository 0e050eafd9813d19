#!/bin/bash
# (a) build time after the pinned hand-over of y - mean; (b) hardware-queue count against the ensemble steps of a suggestion;
# (c) the GPU suite on the pooled allocators
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_l
mkdir -p $OUT
MOE_BUILD_TRACE=1 timeout 600 python tools/chol_time.py 3 2>&1 | grep -v "one-level" | head -4 > $OUT/chol_time.txt
timeout 900 python tools/chol_time.py 12 2>&1 | tail -1 >> $OUT/chol_time.txt
cat $OUT/chol_time.txt
for q in default 8 16 32; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 600 python bench.py --config suggest --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('suggest  %.4f s' % d['value'])"
  timeout 600 python bench.py --config suggest_c3 --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('suggest_c3  %.4f s' % d['value'])"
done > $OUT/hwq.txt 2>&1
unset GPU_MAX_HW_QUEUES
cat $OUT/hwq.txt
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6) > $OUT/pytest.txt
cat $OUT/pytest.txt
