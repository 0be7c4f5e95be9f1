"""Print the counting kernels of the newest rocprofv3 --stats run under a directory: name, calls, avg ms."""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    m = re.search(r"(msp2?_\w+|rx\w*_\w+|hash_\w+|result_\w+|dense_\w+)", r["Name"])
    if m:
        print("%-28s calls=%-4s avg_ms=%.3f" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e6))
