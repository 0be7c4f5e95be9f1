#!/bin/bash
# Same-box A/B of two builds of the library: cfrk_amd/libcfrk_hip.so (new) against
# tools/_bin/libcfrk_hip_prev.so (old; build it from the commit to compare with), alternating, profiled per kernel.
# usage (GPU box, repo root): tools/ab.sh <outdir> [bench args]
set -euo pipefail
out=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out/$out"
cp "$R/cfrk_amd/libcfrk_hip.so" /tmp/new.so
cp "$R/tools/_bin/libcfrk_hip_prev.so" /tmp/old.so
trap 'cp /tmp/new.so "$R/cfrk_amd/libcfrk_hip.so"' EXIT      # an interrupted run must not leave the OLD build installed
for round in 1 2; do
  for which in old new; do
    cp /tmp/$which.so "$R/cfrk_amd/libcfrk_hip.so"
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/$out/p_${which}_$round" -- python3 "$R/bench.py" --steps 3 --warmup 1 --cpu-reads 0 "$@" > "$R/gpurun_out/$out/${which}_$round.log" 2>&1
    cd "$R"
    echo "== $which (round $round)"; python3 tools/kstat.py "gpurun_out/$out/p_${which}_$round"
  done
done 2>&1 | tee "$R/gpurun_out/$out/summary.txt"
