#!/bin/bash
# few samples per evaluation at the headline GP size (suggest_c3: 20 restarts x 128 samples per member call): the 16-wavefront
# on-the-fly-weights instantiation against the 8-wavefront slab kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_fly
mkdir -p $O
for v in default MOE_KG_ONFLY=1 "MOE_KG_ONFLY=1 MOE_KG_FLY_WAVES=12"; do
  echo "== $v"
  for rep in 1 2; do env $( [ "$v" = default ] || echo $v ) timeout 600 python bench.py --config suggest_c3 --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('suggest_c3', d['value'], d['timeline']['ms_per_gradient_step']['median'])"; done
  for R in 20 1; do printf "C3:M=128 R=%-3s " $R; env $( [ "$v" = default ] || echo $v ) timeout 300 python tools/prof_kg.py "C3:M=128" $R 4 2>&1 | grep "^rep 3" | sed 's/; passes.*//' | cut -c1-120; done
done > $O/fly_few_samples.txt 2>&1
cat $O/fly_few_samples.txt
