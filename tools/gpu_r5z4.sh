#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_z4
mkdir -p $O
for v in default MOE_KG_MULTI_TRIAL=0; do
  echo "== $v"
  for rep in 1 2; do env $( [ "$v" = default ] || echo $v ) timeout 600 python bench.py --config suggest --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('suggest', d['value'], d['timeline']['ms_per_gradient_step']['median'])"; done
  for cfg in "C3:n=30,d=2,M=2000" "C3:n=100,d=4,M=2000" "C3:n=30,d=2,M=128"; do
    for R in 64 1; do
      printf "%-24s R=%-3s " "$cfg" $R
      env $( [ "$v" = default ] || echo $v ) timeout 300 python tools/prof_kg.py "$cfg" $R 4 2>&1 | grep "^rep 3" | sed 's/; passes.*//' | cut -c1-120
    done
  done
done > $O/multi_trial_small.txt 2>&1
cat $O/multi_trial_small.txt
