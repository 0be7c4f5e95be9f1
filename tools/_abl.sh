cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl3 -- python3 bench.py --cpu-reads 0 --steps 1 --warmup 1 --reads 125000000 --L 250 --k 63 > gpurun_out/abl3.log 2>&1
python3 tools/kstat.py gpurun_out/abl3; grep -o '"ms_per_step": [^,]*' gpurun_out/abl3.log
