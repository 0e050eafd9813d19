#!/bin/bash
# Clock / busy counters of the KG MC kernel (separate --pmc passes, kernel trace only): tools/pmc_busy.sh <tag> [config] [restarts]
TAG="${1:-r04}"; CFG="${2:-C3}"; R="${3:-64}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
i=0
for CTRS in "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pb_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pb_$i -o p -- python $ROOT/tools/prof_kg.py $CFG $R 2 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
cols = None
for f in glob.glob("/tmp/pb_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if cols is None:
            cols = list(row.keys()); print("columns:", cols)
        if "kg_mc" in row["Kernel_Name"]:
            dur = None
            if "Start_Timestamp" in row and "End_Timestamp" in row:
                dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            acc[row["Counter_Name"]].append((float(row["Counter_Value"]), dur))
for k, v in sorted(acc.items()):
    print(k, ["%.6g (dur %s ns)" % (a, b) for a, b in v])
PY
