#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_chk
mkdir -p $O
for v in default MOE_KG_COMPACT_GRID=0; do
  echo "== $v"
  env $( [ "$v" = default ] || echo $v ) timeout 300 python tools/latency.py 2>&1 | grep "C3 KG value+grad (1) \|last kernel"
  env $( [ "$v" = default ] || echo $v ) timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-determinism 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('bench', d['value'], d['kernel_ms_per_eval'], d['batch1']['ms_per_eval'], d['batch8']['value'])"
done > $O/compact_grid_check.txt 2>&1
cat $O/compact_grid_check.txt
