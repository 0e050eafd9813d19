#!/bin/bash
# one / two tiles at d <= 4: the 16-wavefront small-shape frame kernel (default) against the 8-wavefront lane-parked kernel (MOE_KG_SMALL_TILES=0),
# single stream, batches of 64 and of 1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_z
mkdir -p $O
for cfg in "C3:n=30,d=2,M=2000" "C3:n=60,d=2,M=2000" "C3:n=100,d=4,M=2000" "C3:n=120,d=3,q=2,M=2000" "C3:n=30,d=2,M=128"; do
  for R in 64 1; do
    for v in default MOE_KG_SMALL_TILES=0; do
      printf "%-28s R=%-3s %-22s " "$cfg" $R "$v"
      env $( [ "$v" = default ] || echo $v ) timeout 300 python tools/prof_kg.py "$cfg" $R 4 2>&1 | grep "^rep 3" | sed 's/; passes.*//' | cut -c1-120
    done
  done
done > $O/small_shape_single_stream.txt 2>&1
cat $O/small_shape_single_stream.txt
MOE_KG_SMALL_TILES=0 timeout 600 python tools/digest.py > $O/digest_small0.txt 2>&1
diff $O/digest_small0.txt profiles/r05_r_digest_prev.txt | head -6
