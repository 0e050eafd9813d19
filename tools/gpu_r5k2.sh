#!/bin/bash
# runtime knob: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) against the launch-bound paths
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_k2
mkdir -p $O
for v in default HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0; do
  echo "== $v"
  for rep in 1 2; do env $( [ "$v" = default ] || echo $v ) timeout 600 python bench.py --config suggest --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('suggest', d['value'], d['timeline']['ms_per_gradient_step']['median'])"; done
  env $( [ "$v" = default ] || echo $v ) timeout 300 python tools/latency.py 2>&1 | grep "C3 KG value+grad (1) \|C2 EI value+grad  \|last kernel"
done > $O/kernarg.txt 2>&1
cat $O/kernarg.txt
