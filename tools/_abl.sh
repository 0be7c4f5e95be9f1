cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
CFRK_BENCH_INFO=1 timeout -k 10 300 python bench.py --cpu-reads 0 --steps 2 --warmup 1 --reads 125000000 --L 250 --k 63 > gpurun_out/big.log 2>&1; cut -c1-330 gpurun_out/big.log | tail -2
