#!/usr/bin/env python3
"""Per-launch view of the newest `rocprofv3 --kernel-trace` run under a directory: the launches of the
LAST counting step (from the last partition kernel before the last leaf kernel), in order, with
duration and the gap to the previous launch.  usage: python tools/ktrace.py <dir> [max lines]"""
import csv
import glob
import re
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(msp2?_\w+|rx\w*_\w+|hash_\w+|result_\w+|fillBuffer\w*)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30], r["Grid_Size_X"]))
rows.sort()
last_p3 = max(i for i, r in enumerate(rows) if "p3_kernel" in r[2] or "q3" in r[2] or "p3" in r[2])
prev_p3 = max([i for i, r in enumerate(rows[:last_p3]) if "p3" in r[2]] or [-1])
step = rows[prev_p3 + 1:last_p3 + 1]
t0 = step[0][0]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 80
print(f"last step: {len(step)} launches, {(step[-1][1] - t0) / 1e6:.3f} ms from first start to last end")
for i, (s, e, name, grid) in enumerate(step[:limit]):
    gap = (s - step[i - 1][1]) / 1e3 if i else 0.0
    print(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e6:8.3f} ms  gap {gap:7.1f} us  {name}  grid {grid}")
