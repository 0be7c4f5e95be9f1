#!/bin/bash
# Timing ablations of the partitioned path: one bench run per phase switched off (the counts of
# those runs are wrong by construction; only kernel_ms is read).  usage: tools/ablate.sh <outdir> [bench args]
out=$1; shift
mkdir -p gpurun_out/$out
for f in 0 0x100 0x200 0x300 0x400 0x800 0xF00 0x1000; do
  CFRK_DEBUG_FLAGS=$f python3 bench.py --steps 5 --warmup 1 --cpu-reads 0 "$@" > gpurun_out/$out/abl_$f.json 2> gpurun_out/$out/abl_$f.err
  python3 - "$f" gpurun_out/$out/abl_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
    print(f"flags {sys.argv[1]:>7}: chain {d['roofline']['kernel_ms']:.2f} ms")
except Exception as e:
    print(f"flags {sys.argv[1]:>7}: failed ({e})")
PY
done | tee gpurun_out/$out/summary.txt
