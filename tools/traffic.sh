#!/bin/bash
# HBM traffic of the counting kernels from TCC counters, separate passes as the microarch guide
# prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass).  usage: tools/traffic.sh <outdir> [bench args]
out=$1; shift
mkdir -p gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 1 --warmup 0 --cpu-reads 0 $@"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$out/fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/$out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$out/write -- python3 $R/bench.py $ARGS > $R/gpurun_out/$out/write.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/$out/fetch $R/gpurun_out/$out/write > $R/gpurun_out/$out/summary.txt 2>&1
cat $R/gpurun_out/$out/summary.txt
