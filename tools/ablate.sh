#!/bin/bash
# Timing ablations of the partitioned path: one profiled bench run per phase switched off (the counts
# of those runs are wrong by construction; only kernel times are read).
# usage: tools/ablate.sh <outdir> "<flags...>" [bench args]
out=$1; flags=$2; shift; shift
mkdir -p gpurun_out/$out
R=$GRAFT_REPO_ROOT
for f in $flags; do
  cd /tmp && export TMPDIR=/tmp
  CFRK_DEBUG_FLAGS=$f rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$out/p_$f -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-reads 0 "$@" > $R/gpurun_out/$out/abl_$f.log 2>&1
  cd $R
  echo "== flags $f"; python3 tools/kstat.py gpurun_out/$out/p_$f
done 2>&1 | tee gpurun_out/$out/summary.txt
