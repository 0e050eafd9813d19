#!/bin/bash
# Kernel trace (+ memory copies) of an arbitrary python tool: tools/ktrace_cmd.sh <tag> <script> [args...]
TAG="$1"; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $ROOT/gpurun_out/kt_$TAG
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $ROOT/gpurun_out/kt_$TAG -o kt -- python $ROOT/"$@" > $ROOT/gpurun_out/${TAG}_run.txt 2> $ROOT/gpurun_out/${TAG}_err.txt
cd $ROOT
cat gpurun_out/${TAG}_run.txt
for f in $(find gpurun_out/kt_$TAG -name "*_stats.csv"); do
  echo "== $f"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print("%-80s calls %6s  avg %9.1f ns  total %8.3f ms  %5s%%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
cp $(find gpurun_out/kt_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/kt_$TAG
