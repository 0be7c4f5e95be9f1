#!/bin/bash
# Timing ablations of the counting kernels: one profiled bench run per phase switched off (the counts of those runs
# are wrong by construction; only kernel times are read).  The switches (cfrk_amd/csrc/msp.h: CFRK_ABL_*) exist in
# the ABLATION BUILD only -- build it first, in the container: `make -C cfrk_amd/csrc abl` ->
# tools/_bin/libcfrk_hip_abl.so (travels with gpurun); this script swaps it in for the runs and puts the product
# library back afterwards.   usage: tools/ablate.sh <outdir> "<flags...>" [bench args]
set -u
out=$1; flags=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out/$out"
[ -f "$R/tools/_bin/libcfrk_hip_abl.so" ] || { echo "tools/_bin/libcfrk_hip_abl.so is missing: make -C cfrk_amd/csrc abl"; exit 2; }
cp "$R/cfrk_amd/libcfrk_hip.so" /tmp/product.so
trap 'cp /tmp/product.so "$R/cfrk_amd/libcfrk_hip.so"' EXIT
cp "$R/tools/_bin/libcfrk_hip_abl.so" "$R/cfrk_amd/libcfrk_hip.so"
for f in $flags; do
  cd /tmp && export TMPDIR=/tmp
  CFRK_DEBUG_FLAGS=$f rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/$out/p_$f" -- python3 "$R/bench.py" --steps 3 --warmup 1 --cpu-reads 0 "$@" > "$R/gpurun_out/$out/abl_$f.log" 2>&1
  cd "$R"
  echo "== flags $f"; python3 tools/kstat.py "gpurun_out/$out/p_$f"
done 2>&1 | tee "$R/gpurun_out/$out/summary.txt"
