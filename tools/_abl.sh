cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl0 -- python3 bench.py --cpu-reads 0 --steps 2 --warmup 1 > gpurun_out/abl0.log 2>&1
python3 tools/kstat.py gpurun_out/abl0 | grep msp; grep -o '"digest": [^]]*]' gpurun_out/abl0.log
