#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_z3
mkdir -p $O
for v in default MOE_KG_LANE=0 MOE_KG_MULTI_TRIAL=0; do
  echo "== $v"; env $( [ "$v" = default ] || echo $v ) timeout 600 python tools/golden_kg_debug.py 2>&1 | tail -20
done > $O/golden_kg_debug.txt 2>&1
cat $O/golden_kg_debug.txt
