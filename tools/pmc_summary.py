"""Summarise rocprofv3 --pmc CSVs: per kernel name, the mean of every counter over its dispatches (largest kernels first)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?")
            short = name.split("(")[0][-60:]
            agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
keys = sorted(agg, key=lambda k: -sum(agg[k].get("SQ_WAVE_CYCLES", agg[k].get("GRBM_GUI_ACTIVE", [0]))))
for k in keys[:8]:
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("    %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
