#!/bin/bash
# Per-kernel timeline of the LAST GP build of tools/chol_prof.py (N = 8000): start offset, duration, gap, grid, name; then per-name sums.
#   tools/build_timeline.sh [g] [env assignments...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
G="${1:-3}"; shift
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/bt
env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -o kt -- python $ROOT/tools/chol_prof.py $G > /tmp/bt_out.txt 2> /tmp/bt_err.txt
python - <<'PY'
import csv, glob, re
from collections import defaultdict
rows = []
for f in glob.glob("/tmp/bt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[-64:], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
rows.sort()
# a build starts at a cov_build kernel with a large grid; take the last one
starts = [i for i, r in enumerate(rows) if "cov_build" in r[2]]
b = rows[starts[-1]:]
t0 = b[0][0]
span = (max(x[1] for x in b) - t0) / 1e3
busy = sum(e - s for s, e, *_ in b) / 1e3
print("last build: %d kernels, span %.1f us, sum of durations %.1f us" % (len(b), span, busy))
prev = None
agg = defaultdict(lambda: [0, 0.0])
for s, e, name, grid, wg in b:
    a = agg[name]; a[0] += 1; a[1] += (e - s) / 1e3
    if "chol_step" in name or "chol_diag" in name:
        prev = e
        continue  # (the 124 steps: summed below)
    print("%9.1f us  +%7.1f us  gap %6.1f  grid %-9s wg %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, grid, wg, name))
    prev = e
print("-- by kernel")
for name, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%5d x  %9.1f us  %s" % (cnt, us, name))
PY
