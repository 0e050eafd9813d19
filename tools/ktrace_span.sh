#!/bin/bash
# Span / busy time / gaps of the device work of <script>: tools/ktrace_span.sh <script> [args...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kts
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -o kt -- python $ROOT/"$@" > /dev/null 2> /tmp/kts_err.txt
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kts/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# split into bursts separated by > 2 ms of idle
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if int(r["Start_Timestamp"]) - int(cur[-1]["End_Timestamp"]) > 2_000_000:
        bursts.append(cur); cur = []
    cur.append(r)
bursts.append(cur)
for b in bursts:
    span = (int(b[-1]["End_Timestamp"]) - int(b[0]["Start_Timestamp"])) / 1e6
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in b) / 1e6
    gaps = [int(b[i + 1]["Start_Timestamp"]) - int(b[i]["End_Timestamp"]) for i in range(len(b) - 1)]
    big = sorted(gaps, reverse=True)[:5]
    print("burst: %4d kernels  span %8.3f ms  busy %8.3f ms  gaps %7.3f ms (median %.1f us, largest %s us)" % (
        len(b), span, busy, span - busy, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, [round(g / 1e3, 1) for g in big]))
PY
