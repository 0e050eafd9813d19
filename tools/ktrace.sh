#!/bin/bash
# Kernel-level trace of a few KG evaluations: tools/ktrace.sh <tag> <config> <restarts> <reps>  -> gpurun_out/<tag>_kernel_stats.csv
TAG="$1"; CFG="${2:-C3}"; R="${3:-1}"; REPS="${4:-5}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $ROOT/gpurun_out/kt_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/kt_$TAG -o kt -- python $ROOT/tools/prof_kg.py $CFG $R $REPS > $ROOT/gpurun_out/${TAG}_run.txt 2> $ROOT/gpurun_out/${TAG}_err.txt
cd $ROOT
cp $(find gpurun_out/kt_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
cat gpurun_out/${TAG}_run.txt
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:40]:
    print("%-90s calls %6s  avg %10.1f ns  total %8.3f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
rm -rf gpurun_out/kt_$TAG
