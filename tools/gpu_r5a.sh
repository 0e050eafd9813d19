#!/bin/bash
# Round 5, GPU call A: parity of the lane-parked MC kernel (full GPU suite), A/B timing against the round-4 kernel and the build
# variants under variants/, VALU instruction counts (one PMC pass each).   gpurun -- bash tools/gpu_r5a.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$ROOT/gpurun_out/r05_a
mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $OUT/pytest_gpu.txt
echo "== default (lane kernel)" > $OUT/ab.txt
timeout 120 python tools/prof_kg.py C3 64 4 >> $OUT/ab.txt 2>&1
echo "== MOE_KG_LANE=0 (round-4 kernel)" >> $OUT/ab.txt
MOE_KG_LANE=0 timeout 120 python tools/prof_kg.py C3 64 4 >> $OUT/ab.txt 2>&1
for v in variants/libmoe_hip_*.so; do
  [ -f "$v" ] || continue
  echo "== $v" >> $OUT/ab.txt
  MOE_LIB_PATH=$ROOT/$v timeout 120 python tools/prof_kg.py C3 64 4 >> $OUT/ab.txt 2>&1
done
echo "== default, batch of 8" >> $OUT/ab.txt
timeout 120 python tools/prof_kg.py C3 8 4 >> $OUT/ab.txt 2>&1
echo "== MOE_KG_LANE=0, batch of 8" >> $OUT/ab.txt
MOE_KG_LANE=0 timeout 120 python tools/prof_kg.py C3 8 4 >> $OUT/ab.txt 2>&1
cd /tmp
for mode in 1 0; do
  rm -rf /tmp/pmc_$mode
  MOE_KG_LANE=$mode timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_$mode -o p -- python $ROOT/tools/prof_kg.py C3 64 2 > /dev/null 2>&1
done
for mode in 1 0; do
  rm -rf /tmp/pmcb_$mode
  MOE_KG_LANE=$mode timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --output-format csv -d /tmp/pmcb_$mode -o p -- python $ROOT/tools/prof_kg.py C3 64 2 > /dev/null 2>&1
done
python - > $OUT/pmc.txt <<PY
import csv, glob, collections
for tag in ("pmc_1", "pmc_0", "pmcb_1", "pmcb_0"):
    acc = collections.defaultdict(list)
    for f in glob.glob("/tmp/%s/**/*counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(f)):
            if "kg_mc" in row["Kernel_Name"]:
                acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(tag, k, ["%.6g" % a for a in v])
PY
cat $OUT/pytest_gpu.txt $OUT/ab.txt $OUT/pmc.txt
