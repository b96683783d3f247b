"""Work counters and scheduling of one frame of a bench workload, from the counting forms of the kernels
(profiles/print_stats.py [workload] [W] [H] [spp]).  The any-hit kernel's counting form is the timed kernel with counters:
same queue, same visits, same triangle tests."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtimeraytracer_amd import scenes, api, _abi as A  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sponza_class"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ctx = api.Context(0)
s = getattr(scenes, {"sponza_class": "sponza_class", "cornell": "cornell_box", "bunny_class": "bunny_class"}[name])(W, H)
scene = api.Scene(ctx, s.desc)
ss = scene.stats()
frame = api.Frame(ctx, W, H)
api.render(scene, s.camera, s.scene_info(0), api.make_params(W, H, spp=spp, collect_stats=0), frame)
t0 = frame.stats()
api.render(scene, s.camera, s.scene_info(0), api.make_params(W, H, spp=spp, collect_stats=1), frame)
st = frame.stats()
print(f"{name} {W}x{H} {spp}spp: {ss.numTriangles} triangles, {ss.numNodes} BVH2 nodes, {ss.numWideNodes} 4-wide records (layout {ss.wideLayoutVersion}), "
      f"wide centre ({ss.grid.wideCentreXY & 0xffff}, {ss.grid.wideCentreXY >> 16}, {ss.grid.wideCentreZ}) grid steps")
print("rays", st.numRays, "shadow", st.numShadowRays, "4-wide visits", st.numShadowNodeVisits, "shadow tri tests", st.numShadowTriTests, "tail rays", st.shadowTailRays)
print("per shadow ray: 4-wide visits %.2f  tri tests %.2f ; per primary ray: BVH2 visits %.2f  tri tests %.2f" % (
    st.numShadowNodeVisits / st.numShadowRays, st.numShadowTriTests / st.numShadowRays,
    (st.numNodeVisits - st.numShadowNodeVisits) / st.numPrimaryRays, (st.numTriTests - st.numShadowTriTests) / st.numPrimaryRays))
ii, il, ti, tl = st.shadowInnerIterations, st.shadowInnerActiveLanes, st.shadowTriIterations, st.shadowTriActiveLanes
print("any-hit kernel: node-phase trips %d, lanes/trip %.1f of 64 ; triangle-phase trips %d, lanes/trip %.1f of 64 ; refill passes %d" % (
    ii, il / max(ii, 1), ti, tl / max(ti, 1), st.shadowRefills))
print("timed form: primary %.3f gen %.3f trace %.3f resolve %.3f ms ; any-hit clock %.0f MHz ; counting form trace %.3f ms" % (
    t0.primaryMs, t0.shadowGenMs, t0.shadowTraceMs, t0.resolveMs, t0.shadowTraceClockMHz, st.shadowTraceMs))
