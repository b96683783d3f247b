"""Times rtr_denoise_combine (8 a-trous dispatches + combine, application.cppm:391-445) on the bench scene at 1080p.
python profiles/time_denoise.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from realtimeraytracer_amd import scenes, api, _abi as A
ctx = api.Context(0)
W, H = 1920, 1080
s = scenes.sponza_class(W, H, ltc=scenes.synthetic_ltc())
scene = api.Scene(ctx, s.desc)
frame = api.Frame(ctx, W, H, 0xff)
p = api.make_params(W, H, images=A.IMAGES_RAYGEN5)
api.render(scene, s.camera, s.scene_info(0), p, frame)
for it in (4,):
    for _ in range(3):
        frame.denoise_combine(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        frame.denoise_combine(it)
    torch.cuda.synchronize()
    print(f"denoise x{it} (2 images each) + combine at {W}x{H}: {(time.perf_counter() - t0) * 1e3 / n:.3f} ms per call")
# the whole frame the reference presents: ray-gen with all five images (analytic LTC, shadowed, unshadowed, normal, position),
# four denoise steps, combine
for _ in range(3):
    api.render(scene, s.camera, s.scene_info(0), p, frame)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    api.render(scene, s.camera, s.scene_info(i), p, frame)
torch.cuda.synchronize()
t1 = time.perf_counter()
st = frame.stats()
print(f"ray-gen dispatch with 5 images: {(t1 - t0) * 1e3 / n:.3f} ms per frame (primary {st.primaryMs:.3f} gen {st.shadowGenMs:.3f} trace {st.shadowTraceMs:.3f} resolve {st.resolveMs:.3f})")
t0 = time.perf_counter()
for i in range(n):
    api.render(scene, s.camera, s.scene_info(i), p, frame)
    frame.denoise_combine(4)
torch.cuda.synchronize()
print(f"ray-gen (5 images) + denoise x4 + combine: {(time.perf_counter() - t0) * 1e3 / n:.3f} ms per presented frame")
