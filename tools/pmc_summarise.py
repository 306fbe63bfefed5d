"""Per-kernel means of every counter in a rocprofv3 --pmc output directory.
Usage: python tools/pmc_summarise.py <dir> [substring of kernel name ...]"""
import csv
import glob
import json
import os
import sys

acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
                continue
            s = acc.setdefault(name, {}).setdefault(row["Counter_Name"], [0.0, 0])
            s[0] += float(row["Counter_Value"])
            s[1] += 1
print(json.dumps({k: {c: v[0] / v[1] for c, v in d.items()} for k, d in acc.items()}, indent=1))
