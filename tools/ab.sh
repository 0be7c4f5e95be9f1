#!/bin/bash
# Same-box A/B of two builds of the library: cfrk_amd/libcfrk_hip.so (new) against
# cfrk_amd/libcfrk_hip_prev.so.keep (old), alternating, profiled per kernel.
# usage (GPU box, repo root): tools/ab.sh <outdir> [bench args]
out=$1; shift
mkdir -p gpurun_out/$out
R=$GRAFT_REPO_ROOT
cp $R/cfrk_amd/libcfrk_hip.so /tmp/new.so
cp $R/cfrk_amd/libcfrk_hip_prev.so.keep /tmp/old.so
for round in 1 2; do
  for which in old new; do
    cp /tmp/$which.so $R/cfrk_amd/libcfrk_hip.so
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$out/p_${which}_$round -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-reads 0 "$@" > $R/gpurun_out/$out/${which}_$round.log 2>&1
    cd $R
    echo "== $which (round $round)"; python3 tools/kstat.py gpurun_out/$out/p_${which}_$round
  done
done 2>&1 | tee gpurun_out/$out/summary.txt
cp /tmp/new.so $R/cfrk_amd/libcfrk_hip.so
