#!/usr/bin/env python3
"""Full-size parity, independent of the GPU: the ORACLE counts a whole BASELINE config on the host cores
(bounded memory: orc_synth_digest generates the reads block by block and counts the key space in slices)
and writes the digest that the `-m gpu` full-size tests then assert.

    python tools/oracle_full_digest.py c3        [--threads N] [--slices S] [--out FILE]
    python tools/oracle_full_digest.py c5shard
    python tools/oracle_full_digest.py c2

Test infrastructure (it loads oracle/liboracle.so); nothing here touches the GPU or the product library."""
import argparse
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import oracle_lib as orc  # noqa: E402

CONFIGS = {
    # name: (R, L, Glen, k)   -- SURVEY 8d; seeds 1 / 2 / 3, canonical
    "c2": (10_000_000, 150, 10_000_000, 15),
    "c3": (100_000_000, 150, 100_000_000, 31),
    "c3k63": (100_000_000, 150, 100_000_000, 63),
    "c5shard": (125_000_000, 250, 1_000_000_000, 63),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=sorted(CONFIGS))
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--slices", type=int, default=0, help="0: from the free host memory")
    ap.add_argument("--reads", type=int, default=0, help="count only the first N reads (rehearsal)")
    ap.add_argument("--out", default="")
    ap.add_argument("--git", default="", help="commit the oracle was built from (the GPU box has no .git)")
    a = ap.parse_args()
    R, L, G, k = CONFIGS[a.config]
    if a.reads:
        R = a.reads
    occ = R * (L - k + 1)
    per = 28 if k > 32 else 12          # key + run length (+ the sort's scratch is per partition)
    free = 0
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable"):
            free = int(ln.split()[1]) * 1024
    slices = a.slices or max(8 if occ > 10**9 else 1, -(-occ * per // max(1, int(free * 0.4))))
    t0 = time.time()
    dg = orc.synth_digest(R, L, G, k, orc.ORC_CANONICAL, a.threads, slices,
                          progress=lambda i, n: print(f"[oracle {a.config}] slice {i}/{n} done after {time.time() - t0:.1f} s",
                                                      file=sys.stderr, flush=True))
    dt = time.time() - t0
    sha = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                         cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    rec = {"config": a.config, "reads": R, "read_length": L, "genome": G, "k": k, "canonical": True,
           "seeds": [1, 2, 3], "digest_hex": [f"{x:016x}" for x in dg], "distinct": dg[0], "occurrences": dg[1],
           "expected_occurrences": occ, "threads": a.threads, "slices": slices, "seconds": round(dt, 2),
           "host_mem_available_bytes": free, "counter": "oracle/cfrk_oracle.c: orc_synth_digest",
           "command": "python " + " ".join(sys.argv), "git": a.git or sha or None}
    line = json.dumps(rec)
    print(line, flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(line + "\n")
    assert dg[1] == occ, "the generator makes no invalid bases: every window must be counted"


if __name__ == "__main__":
    main()
