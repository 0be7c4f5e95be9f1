import csv,sys,glob
f=sorted(glob.glob(sys.argv[1]+"/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "msp_p" in r["Name"]: print(r["Name"].split("(")[0][-16:], r["Calls"], float(r["AverageNs"])/1e6)
