#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_blk
mkdir -p $O
for v in default MOE_KG_BLOCKS=160; do
  echo "== $v"
  for cfg in "C3:M=128" "C3:n=30,d=2,M=128" "C3:M=200"; do
    printf "%-22s R=20 " "$cfg"; env $( [ "$v" = default ] || echo $v ) timeout 300 python tools/prof_kg.py "$cfg" 20 5 2>&1 | grep "^rep 4" | sed 's/; passes.*//' | cut -c1-120
  done
done > $O/blocks_single_stream.txt 2>&1
cat $O/blocks_single_stream.txt
