#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: sum of each counter over dispatches.
usage: pmc_summary.py <dir> [<dir> ...]"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40]
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name].add(row["Dispatch_Id"])
for name, ctrs in acc.items():
    print(f"== {name}  dispatches={len(calls[name])}")
    for c, v in sorted(ctrs.items()):
        print(f"   {c:28s} {v:.4g}")
