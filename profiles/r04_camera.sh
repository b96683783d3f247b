#!/bin/bash
# Round 4, step 1: the bench with a frame time.  (a) the driver's command with the moving camera and the latency bound; (b) the same launches with a
# static and a moving camera, to split what frames-per-launch buys into launch amortisation and same-view cache sharing.
mkdir -p gpurun_out/r04
show() { python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=j.get('roofline') or {}
print('$1', '| ms/frame', j['ms_per_step'], '| Grays/s', round(j['value']/1e3,3), '| frames/launch', j['frames_per_launch'], '| latency ms', j['frame_latency_ms'], '| rays/frame', j['config']['rays_per_frame'],
      '| kernels (per frame)', j.get('kernels_ms_in_flight_event_brackets') or j['kernels_ms'], '| one at a time', (j.get('one_frame_at_a_time') or {}).get('ms_per_step'), '| iso kernels', j['kernels_ms'], '| probes', (j.get('latency') or {}).get('probes'))
"; }
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_cmd_1.log 2>gpurun_out/r04/bench_driver_cmd_1.err; echo "driver cmd rc $?"; show "driver cmd (path, 16.7 ms)" < gpurun_out/r04/bench_driver_cmd_1.log
for cam in static path; do
  for b in 1 8 20 32; do
    python3 bench.py --steps 64 --warmup 8 --batch $b --camera $cam --no-cpu-baseline --present-frames 0 2>/dev/null | show "camera=$cam batch=$b"
  done
done
