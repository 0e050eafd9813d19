#!/bin/bash
# Per-dispatch durations of the kernels whose name contains <pattern>: tools/ktrace_list.sh <pattern> <script> [args...]
PAT="$1"; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ktl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktl -o kt -- python $ROOT/"$@" > /dev/null 2> /tmp/ktl_err.txt
python - "$PAT" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ktl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    if sys.argv[1] in r["Kernel_Name"]:
        print("%10.1f us  +%8.1f us  grid %-8s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                    r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?"), r["Kernel_Name"][:90]))
PY
