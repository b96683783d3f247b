"""How many (ray, triangle) pairs wait when a wave of the any-hit kernel starts a leaf phase (re-priced on the farthest-exit-first
walk).  Needs a library built with -DRTR_STATS_LEAF_PHASE=1 (profiles/build_flags_variant.sh "-DRTR_STATS_LEAF_PHASE=1" leafphase) copied over
librtr_hip.so on the GPU box; its counting form re-uses the per-trip counters: trips := leaf phases, lanes := waiting pairs, refills := phases with > 64 pairs +
(lanes at a leaf << 32).
    cp realtimeraytracer_amd/librtr_hip_leafphase.so realtimeraytracer_amd/librtr_hip.so && python profiles/experiments/leaf_phase_pairs.py [workload] [W] [H]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from realtimeraytracer_amd import scenes, api  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sponza_class"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
ctx = api.Context(0)
s = getattr(scenes, {"sponza_class": "sponza_class", "cornell": "cornell_box", "bunny_class": "bunny_class", "sponza_mixed": "sponza_mixed"}[name])(W, H)
scene = api.Scene(ctx, s.desc)
frame = api.Frame(ctx, W, H)
api.render(scene, s.camera, s.scene_info(0), api.make_params(W, H, spp=1, collect_stats=1), frame)
st = frame.stats()
phases, pairs = st.shadowTriIterations, st.shadowTriActiveLanes
over, lanes = st.shadowRefills & 0xffffffff, st.shadowRefills >> 32
print(f"{name} {W}x{H}: {st.numShadowRays} shadow rays, {st.numShadowTriTests} triangle tests ({st.numShadowTriTests / st.numShadowRays:.2f} per ray)")
print(f"leaf phases {phases}: lanes at a leaf {lanes / max(phases, 1):.1f}, waiting pairs {pairs / max(phases, 1):.1f} per phase "
      f"({pairs / max(lanes, 1):.2f} triangles per leaf met; {pairs / st.numShadowRays:.2f} per ray if all were tested), phases with more than 64 pairs {over} ({100.0 * over / max(phases, 1):.1f} %)")
print(f"node-phase trips {st.shadowInnerIterations}, lanes/trip {st.shadowInnerActiveLanes / max(st.shadowInnerIterations, 1):.1f}")
