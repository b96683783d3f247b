#!/bin/bash
# Single-GPU rehearsals of the N>1 paths of bench.py (the real N-GPU runs are the driver's):
#  (0) profiles/shard_scaling.py : kernel times of shard 0 of N rendered one frame at a time (how each kernel's time divides)
#  (1) --emulate-rank-of N : what ONE rank does in an N-GPU run (shard 0 of N, F frames in flight on F streams), no gather
#  (2) --backend gloo with RTR_BENCH_SAME_DEVICE=1 : N processes on one GPU run the whole control flow incl. gather + verify
cd $GRAFT_REPO_ROOT
timeout -k 5 120 python profiles/shard_scaling.py 2>/dev/null
show='import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], "in flight", j["frames_in_flight"], j["value"], "Mrays/s", j["ms_per_step"], "ms/step; one at a time:", j["one_frame_at_a_time"])'
for f in 1 2 3 4; do
  timeout -k 5 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --frames-in-flight $f 2>/dev/null | python3 -c "$show" "N=1          "
done
for n in 2 4 8; do
  for f in 2 3 4; do
    timeout -k 5 120 python bench.py --steps 40 --warmup 6 --emulate-rank-of $n --frames-in-flight $f 2>/dev/null | python3 -c "$show" "one rank of $n"
  done
done
RTR_BENCH_SAME_DEVICE=1 timeout -k 5 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 7 --warmup 2 --backend gloo 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('gloo 2 ranks on one GPU: verify', j['verify'])"
RTR_BENCH_SAME_DEVICE=1 timeout -k 5 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 6 --warmup 1 --backend gloo 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('gloo 4 ranks on one GPU: verify', j['verify'])"
#  (3) RTR_BENCH_FORCE_DIST=1 : ONE rank through the real RCCL process group (init, async gather, de-interleave, verify)
RTR_BENCH_FORCE_DIST=1 timeout -k 5 200 python bench.py --steps 12 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('RCCL, 1 rank: verify', j['verify'], j['value'], 'Mrays/s incl. the self-gather of the whole frame')"
