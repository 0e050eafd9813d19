#!/bin/bash
# Per-kernel timeline of ONE C3 q-KG value + gradient evaluation per call (the drop-in call pattern): name, start offset, duration, gap to
# the previous kernel's end, for the last call of tools/kg1_prof.py.   tools/kg1_timeline.sh [extra env assignments...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kg1t
env "$@" timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kg1t -o kt -- python $ROOT/tools/kg1_prof.py > /tmp/kg1t_out.txt 2> /tmp/kg1t_err.txt
cat /tmp/kg1t_out.txt
python - <<'PY'
import csv, glob, re
rows = []
for f in glob.glob("/tmp/kg1t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[-70:]))
for f in glob.glob("/tmp/kg1t/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))))
rows.sort()
# calls are separated by > 150 us of idle; take the last complete one
# a call = from one host->device copy to the next
starts = [i for i, r in enumerate(rows) if r[2].startswith("COPY") and "HOST_TO_DEVICE" in r[2]]
b = rows[starts[-1]:]
t0 = b[0][0]
prev = None
print("last call: %d device operations, span %.1f us" % (len(b), (max(x[1] for x in b) - t0) / 1e3))
for s, e, name in b:
    print("%9.1f us  +%7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, name))
    prev = e
PY
