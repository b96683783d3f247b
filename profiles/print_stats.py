import sys
sys.path.insert(0, '/root/repo')
from realtimeraytracer_amd import scenes, api, _abi as A
ctx = api.Context(0)
s = scenes.sponza_class(1920, 1080)
scene = api.Scene(ctx, s.desc)
frame = api.Frame(ctx, 1920, 1080)
p = api.make_params(1920, 1080, collect_stats=1)
api.render(scene, s.camera, s.scene_info(0), p, frame)
st = frame.stats()
print("rays", st.numRays, "shadow", st.numShadowRays, "nodes", st.numNodeVisits, "shadowNodes", st.numShadowNodeVisits, "tris", st.numTriTests, "shadowTris", st.numShadowTriTests)
print("per shadow ray: nodes4 %.2f tris %.2f ; per primary: nodes %.2f tris %.2f" % (st.numShadowNodeVisits/st.numShadowRays, st.numShadowTriTests/st.numShadowRays,
      (st.numNodeVisits-st.numShadowNodeVisits)/st.numPrimaryRays, (st.numTriTests-st.numShadowTriTests)/st.numPrimaryRays))
