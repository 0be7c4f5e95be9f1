cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
bash tools/ksweep.sh 18 21 24 28 31 > gpurun_out/ks.log 2>&1; cat gpurun_out/ks.log
python bench.py --cpu-reads 0 > gpurun_out/c3.log 2>&1; cut -c1-200 gpurun_out/c3.log | tail -1
