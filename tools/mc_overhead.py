"""Reduce tools/mc_overhead.sh: per-sample instruction counts of kg_mc_kernel in the four runs and the bilinear fit."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
rows = {}
for d in sorted(glob.glob(os.path.join(root, "C3_*"))):
    if not os.path.isdir(d):
        continue
    m = re.search(r"n_(\d+)_steps_(\d+)", d)
    n, steps = int(m.group(1)), int(m.group(2))
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "kg_mc_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    samples = 8 * 10000.0
    rows[(n, steps)] = {k: sum(v) / len(v) / samples for k, v in acc.items()}
    txt = open(d + ".txt").read().strip().splitlines()
    print("n=%d steps=%d: per sample %s   [%s]" % (n, steps, {k: round(v, 1) for k, v in sorted(rows[(n, steps)].items())},
                                                   txt[-2][:160] if len(txt) > 1 else ""))
for ctr in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
    try:
        t8, t16 = (508 + 4 + 63) // 64, (1020 + 4 + 63) // 64
        v = {k: rows[k][ctr] for k in rows}
        d_ = ((v[(1020, 6)] - v[(1020, 3)]) - (v[(508, 6)] - v[(508, 3)])) / (3.0 * (t16 - t8))
        c_ = ((v[(1020, 3)] - v[(508, 3)]) / (t16 - t8)) - 3 * d_
        b_ = (v[(508, 6)] - v[(508, 3)]) / 3.0 - d_ * t8
        a_ = v[(508, 3)] - 3 * b_ - (c_ + 3 * d_) * t8
        print("%s = %.0f + %.0f steps + (%.1f + %.1f steps) tiles  -> at 16 tiles, 6 steps: set-up %.0f, per-step outside the tile loops %.0f, "
              "tile loops %.0f, total %.0f" % (ctr, a_, b_, c_, d_, a_, 6 * b_, (c_ + 6 * d_) * t16, v[(1020, 6)]))
    except KeyError as e:
        print("missing", e)
