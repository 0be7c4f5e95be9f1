cd $GRAFT_REPO_ROOT
python bench.py --cpu-reads 0 --steps 2 --reads 1000000 > gpurun_out/b1.log 2>&1; cut -c1-300 gpurun_out/b1.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --reads 2000000 --same-gpu --dist-backend gloo --cpu-reads 0 > gpurun_out/b2.log 2>&1; grep -o '"step_breakdown_ms": {[^}]*}' gpurun_out/b2.log; grep -o '"sum_count_ok": [a-z]*' gpurun_out/b2.log; tail -2 gpurun_out/b2.log | cut -c1-300
