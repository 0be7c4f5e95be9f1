#!/bin/bash
# Round measurement set (GPU box, repo root): per-kernel times (rocprofv3 --kernel-trace --stats) and HBM
# traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) for the bench configs.  usage: tools/measure_all.sh <outdir> [configs...]
out=$1; shift
cfgs=${@:-c3 c2 k63}
mkdir -p gpurun_out/$out
for c in $cfgs; do
  tools/kprof.sh $out/k_$c --config $c > gpurun_out/$out/kprof_$c.txt 2>&1
  tools/traffic.sh $out/t_$c --config $c > gpurun_out/$out/traffic_$c.txt 2>&1
  echo "== $c"; cat gpurun_out/$out/k_$c/kstat.txt; tail -3 gpurun_out/$out/k_$c/bench.json | cut -c1-600
done
