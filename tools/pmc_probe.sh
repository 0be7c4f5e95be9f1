#!/bin/bash
# Kernel times and SQ counters of the pipelined exchange's kernels (one rank's shard + an owner's groups, tools/dd_probe.py).
# usage (GPU box, repo root): tools/pmc_probe.sh <outdir> [dd_probe args: world reads k groups]
out=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$out/kt -- python3 $R/tools/dd_probe.py "$@" > $R/gpurun_out/$out/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$out/p1 -- python3 $R/tools/dd_probe.py "$@" > $R/gpurun_out/$out/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/$out/p2 -- python3 $R/tools/dd_probe.py "$@" > $R/gpurun_out/$out/p2.log 2>&1
cd $R
{ echo "== kernel times (rocprofv3 --kernel-trace --stats -- python3 tools/dd_probe.py $*)"; python3 tools/kstat.py gpurun_out/$out/kt; cat gpurun_out/$out/kt.log | tail -1; echo; echo "== SQ counters, summed over the run's dispatches of each kernel"; python3 tools/pmc_summary.py gpurun_out/$out/p1 gpurun_out/$out/p2; } > gpurun_out/$out/summary.txt 2>&1
cat gpurun_out/$out/summary.txt
