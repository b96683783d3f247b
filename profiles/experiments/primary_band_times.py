"""Where do the slow camera rays live?  Renders the bench frame one 8-row band at a time (shardCount = number of bands) and prints the
primary kernel's time per band next to the band's mean BVH2 visits per camera ray (counting form)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from realtimeraytracer_amd import scenes, api  # noqa: E402

W, H = 1920, 1080
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_class"
s = getattr(scenes, name)(W, H)
ctx = api.Context(0)
scene = api.Scene(ctx, s.desc)
nb = (H + 7) // 8
frame = api.Frame(ctx, W, 8)
rows = []
for b in range(nb):
    p = api.make_params(W, H, shard_index=b, shard_count=nb)
    api.render(scene, s.camera, s.scene_info(0), p, frame)
    api.render(scene, s.camera, s.scene_info(0), p, frame)
    t = frame.stats().primaryMs
    api.render(scene, s.camera, s.scene_info(0), api.make_params(W, H, shard_index=b, shard_count=nb, collect_stats=1), frame)
    st = frame.stats()
    rows.append((b, t, (st.numNodeVisits - st.numShadowNodeVisits) / max(st.numPrimaryRays, 1), st.primaryTailRays))
for b, t, v, tail in rows:
    print(f"band {b:3d} rows {8 * b:4d}-{8 * b + 7:4d}: primary {t * 1e3:7.1f} us  visits/ray {v:7.1f}  tail rays {tail}")
print("sum of band times %.3f ms, slowest band %.3f ms" % (sum(r[1] for r in rows), max(r[1] for r in rows)))
