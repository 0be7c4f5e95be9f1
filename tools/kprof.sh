#!/bin/bash
# per-kernel times of a bench run: rocprofv3 --kernel-trace --stats, summarised by tools/kstat.py
# usage (GPU box, repo root): tools/kprof.sh <outdir> [bench args]
out=$1; shift
mkdir -p gpurun_out/$out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$out/prof -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-reads 0 "$@" > $R/gpurun_out/$out/bench.log 2>&1
cd $R
python3 tools/kstat.py gpurun_out/$out/prof | tee gpurun_out/$out/kstat.txt
grep '^{' gpurun_out/$out/bench.log | tail -1 > gpurun_out/$out/bench.json
cp $(ls gpurun_out/$out/prof/*/*kernel_stats.csv | tail -1) gpurun_out/$out/kernel_stats.csv
