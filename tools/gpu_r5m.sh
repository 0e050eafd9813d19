#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_m
mkdir -p $OUT
for q in 1 2 3 default; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for t in 16 4; do
  MOE_MCMC_THREADS=$t timeout 600 python bench.py --config suggest --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('suggest threads $t  %.4f s' % d['value'])"
  done
done > $OUT/hwq.txt 2>&1
cat $OUT/hwq.txt
