cd $GRAFT_REPO_ROOT
timeout -k 5 200 python bench.py --steps 20 --warmup 3 --verify > gpurun_out/bench_n1.log 2>&1 && tail -1 gpurun_out/bench_n1.log
RTR_BENCH_SAME_DEVICE=1 timeout -k 5 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo > gpurun_out/bench_rehearse2.log 2>&1; echo rc=$?; tail -3 gpurun_out/bench_rehearse2.log
RTR_BENCH_SAME_DEVICE=1 timeout -k 5 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 3 --steps 5 --warmup 1 --backend gloo > gpurun_out/bench_rehearse3.log 2>&1; echo rc=$?; tail -2 gpurun_out/bench_rehearse3.log
