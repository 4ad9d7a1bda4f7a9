"""Summarise rocprofv3 --pmc csv outputs: per kernel (fa_* only) and counter, the mean over dispatches."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(list)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "fa::fa_" not in name:
                continue
            short = name[name.index("fa::") + 4:name.rindex("(")].replace("fa::", "")
            acc[(short, row["Counter_Name"])].append(float(row["Counter_Value"]))
lines = []
for k in sorted(acc):
    v = acc[k]
    # each dispatch may be reported per-dimension rows; report mean of rows and count
    lines.append(f"{k[0]:58s} {k[1]:34s} mean={sum(v)/len(v):.6g} n={len(v)} min={min(v):.6g} max={max(v):.6g}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "summary.txt"), "w").write(txt + "\n")
