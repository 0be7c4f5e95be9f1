#!/usr/bin/env python3
"""tools/traffic.sh summary -> the traffic_*.json bench.py reads (roofline.traffic).
usage: traffic_json.py <summary.txt> <reads> <read_len> <k> <canonical 0|1> [genome [git-sha]] > profiles/rNN/traffic_xx.json
(git-sha: the commit the measured library was built from -- bench.py echoes it in roofline.traffic_source, so a
constant that is older than the kernels it describes is visible in the bench line)
FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md ('HBM')
prescribes for wide coalesced reads on gfx950; WRITE_SIZE is taken as reported."""
import json
import re
import sys

txt = open(sys.argv[1]).read()
kern, disps = {}, {}
for m in re.finditer(r"^== (\S+?)(?:<.*?>)?\s+dispatches=(\d+)\n((?:   .*\n)+)", txt, re.M):
    name, disp, body = m.group(1), int(m.group(2)), m.group(3)
    if not re.match(r"(msp2?_p\d|rx\d|hash_|result_scan)", name):
        continue
    c = dict((a, float(b)) for a, b in re.findall(r"(\w+)\s+([0-9.e+]+)", body))
    kern[name.replace("_kernel", "")] = {"fetch_bytes": 2 * 1024 * c.get("FETCH_SIZE", 0.0),
                                         "write_bytes": 1024 * c.get("WRITE_SIZE", 0.0), "dispatches": disp}
    disps[name] = disp
# bytes PER COUNTING STEP: a step launches the leaf kernel once, the partition / second-level kernels once
# per chunk of the input (msp.hip) -- the counters above are sums over the run's dispatches
steps = min([d for n, d in disps.items() if not n.startswith("result_scan")] or [1])
for v in kern.values():
    v["fetch_bytes"] /= steps
    v["write_bytes"] /= steps
    v["dispatches_per_step"] = v.pop("dispatches") / steps
out = {"workload": {"reads": int(sys.argv[2]), "read_len": int(sys.argv[3]), "k": int(sys.argv[4]),
                    "canonical": bool(int(sys.argv[5]))},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/traffic.sh); counters "
                 "are in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md 'HBM' prescribes for wide coalesced reads "
                 "on gfx950; WRITE_SIZE taken as reported",
       "kernels": kern,
       "hbm_bytes_per_launch": sum(v["fetch_bytes"] + v["write_bytes"] for v in kern.values())}
if len(sys.argv) > 6:
    out["workload"]["genome"] = int(sys.argv[6])
if len(sys.argv) > 7:
    out["git"] = sys.argv[7]
print(json.dumps(out, indent=1))
