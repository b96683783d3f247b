"""Host cost of enqueueing one frame (rtr_render_async through ctypes) against the GPU time of that frame, for a 1/8 shard.
python profiles/host_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realtimeraytracer_amd import scenes, api, _abi as A
W, H = 1920, 1080
s = scenes.sponza_class(W, H)
ctx0 = api.Context(0)
scene = api.Scene(ctx0, s.desc)
for n in (1, 8):
    rows = api.shard_rows(H, 8, n)
    F = 4
    ctxs = [api.Context(0) for _ in range(F)]
    streams = [torch.cuda.Stream() for _ in range(F)]
    for c, st in zip(ctxs, streams):
        c.set_stream(st.cuda_stream)
    frames = [api.Frame(c, W, rows) for c in ctxs]
    p = api.make_params(W, H, shard_index=0, shard_count=n)
    infos = [s.scene_info(i) for i in range(8)]
    for i in range(8):
        api.render(scene, s.camera, infos[i], p, frames[i % F], asynchronous=True)
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for i in range(N):
        api.render(scene, s.camera, infos[i % 8], p, frames[i % F], asynchronous=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"shard 1/{n}: host enqueue {1e3 * (t1 - t0) / N:.3f} ms per frame; all {N} frames done after {1e3 * (t2 - t0) / N:.3f} ms per frame")
