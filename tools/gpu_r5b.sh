#!/bin/bash
# Round 5, GPU call B: A/B of build variants under variants/ against the default library (C3, 64 per call), bit-identity test, batch-1 timeline.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r05_b
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_sweep.py -q -k "lane_kernel" 2>&1 | tail -5) > $OUT/pytest_lane.txt
echo "== default" > $OUT/ab.txt
timeout 120 python tools/prof_kg.py C3 64 4 2>&1 | grep -v probe >> $OUT/ab.txt
for v in variants/libmoe_hip_*.so; do
  [ -f "$v" ] || continue
  echo "== $v" >> $OUT/ab.txt
  MOE_LIB_PATH=$ROOT/$v timeout 120 python tools/prof_kg.py C3 64 4 2>&1 | grep -v probe >> $OUT/ab.txt
done
echo "== default again" >> $OUT/ab.txt
timeout 120 python tools/prof_kg.py C3 64 4 2>&1 | grep -v probe >> $OUT/ab.txt
bash tools/kg1_timeline.sh > $OUT/kg1_timeline.txt 2>&1
timeout 300 python tools/latency.py > $OUT/latency.txt 2>&1
cat $OUT/pytest_lane.txt $OUT/ab.txt $OUT/kg1_timeline.txt $OUT/latency.txt
