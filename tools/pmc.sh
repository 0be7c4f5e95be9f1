#!/bin/bash
# PMC passes over a reduced bench run (counters only: no trace domains, as gpurun requires).
# usage (on the GPU box, from the repo root): tools/pmc.sh <outdir> [bench args]
out=$1; shift
mkdir -p gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--reads 20000000 --steps 1 --warmup 0 --cpu-reads 0 $@"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$out/p1 -- python3 $R/bench.py $ARGS > $R/gpurun_out/$out/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/$out/p2 -- python3 $R/bench.py $ARGS > $R/gpurun_out/$out/p2.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/$out/p1 $R/gpurun_out/$out/p2 > $R/gpurun_out/$out/summary.txt 2>&1
cat $R/gpurun_out/$out/summary.txt
