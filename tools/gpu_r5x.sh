#!/bin/bash
# kernel census of one whole suggestion (bench.py --config suggest): launches and time per kernel name
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_x
mkdir -p $O
cd /tmp
rm -rf /tmp/sg
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sg -o sg -- python $R/bench.py --config suggest --steps 2 --no-cpu-baseline > $O/suggest_under_rocprof.json 2> /tmp/sg_err.txt
python - "$(find /tmp/sg -name '*kernel_stats.csv' | head -1)" > $O/suggest_kernel_census.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot_calls = sum(int(r["Calls"]) for r in rows); tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels launched: %d, summed duration %.1f ms" % (tot_calls, tot_ns / 1e6))
for r in rows[:40]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); name = re.sub(r"^void ", "", name).split("(")[0]
    print("%8s calls  %9.2f ms total  %8.1f us avg   %s" % (r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, name[-70:]))
PY
cat $O/suggest_kernel_census.txt
cut -c1-300 $O/suggest_under_rocprof.json
