#!/bin/bash
# One MI355X rendering shard 0 of N (no exchange) with the settings bench.py takes at that N: what a rank's rendering side costs per frame.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=gpurun_out/r04/bench_rank0_of_n_r04.log; : > $L
for n in 2 4 8; do
  python3 bench.py --emulate-rank-of $n --steps 96 --warmup 16 --no-cpu-baseline --present-frames 0 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('rank 0 of $n | ms/frame', j['ms_per_step'], '| frames per launch', j.get('frames_per_launch'), '| latency ms', j.get('frame_latency_ms'), '| frames in flight', j.get('frames_in_flight'), '| one at a time', (j.get('one_frame_at_a_time') or {}).get('ms_per_step'), '| kernels', j.get('kernels_ms'))" | tee -a $L
done
