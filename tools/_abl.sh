cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
for k in 31 63; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 3 --steps 1 --warmup 1 --reads 1000000 --glen 2000000 --k $k --same-gpu --dist-backend gloo --cpu-reads 0 > gpurun_out/b2.log 2>&1; grep -o '"sum_count_ok": [a-z]*' gpurun_out/b2.log; grep -o '"digest": [^]]*]' gpurun_out/b2.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 3 --steps 1 --warmup 1 --reads 1000000 --glen 2000000 --k $k --same-gpu --dist-backend gloo --cpu-reads 0 --owner-hash > gpurun_out/b3.log 2>&1; grep -o '"digest": [^]]*]' gpurun_out/b3.log
python bench.py --steps 1 --warmup 0 --reads 3000000 --glen 2000000 --k $k --cpu-reads 0 2>/dev/null | grep -o '"digest": [^]]*]'
done
