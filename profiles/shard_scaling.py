"""Per-rank kernel times when the bench frame is band-sharded N ways (one GPU renders shard 0 of N): how well each
kernel's time divides.  python profiles/shard_scaling.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtimeraytracer_amd import scenes, api, _abi as A
ctx = api.Context(0)
W, H = 1920, 1080
s = scenes.sponza_class(W, H)
scene = api.Scene(ctx, s.desc)
base = None
for n in (1, 2, 4, 8):
    rows = api.shard_rows(H, 8, n)
    frame = api.Frame(ctx, W, rows)
    p = api.make_params(W, H, shard_index=0, shard_count=n)
    acc = [0.0] * 5
    for i in range(12):
        api.render(scene, s.camera, s.scene_info(i), p, frame)
        st = frame.stats()
        if i >= 2:
            for k, v in enumerate((st.primaryMs, st.shadowGenMs, st.shadowTraceMs, st.resolveMs, st.totalMs)):
                acc[k] += v / 10
    if base is None:
        base = acc[4]
    print(f"N={n}: primary {acc[0]:.3f} gen {acc[1]:.3f} trace {acc[2]:.3f} resolve {acc[3]:.3f} total {acc[4]:.3f} ms  -> ideal {base / n:.3f}, kernel-only scaling {base / acc[4]:.2f}x")
