"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same
inputs.  Bar: RGBA8 images bit-exact (integer framebuffer); HDR float accumulation bit-exact as well
(tolerance 0: both sides evaluate the same IEEE expression tree, include/rtr_math.h)."""
import os

import numpy as np
import pytest

from realtimeraytracer_amd import _abi as A
from realtimeraytracer_amd import api, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("queue_mode")]


def _gpu_render(ctx, setup, params, images=A.IMAGES_FRAMEBUFFER, frame=None, scene=None, frame_no=0):
    own = scene is None
    scene = scene or api.Scene(ctx, setup.desc)
    rows = api.shard_rows(params.height, params.bandRows or 8, params.shardCount or 1)
    frame = frame or api.Frame(ctx, params.width, rows, images)
    api.render(scene, setup.camera, setup.scene_info(frame_no), params, frame)
    return scene, frame


def _assert_same(gpu_img, cpu_img, what):
    diff = int((gpu_img != cpu_img).sum())
    if diff:
        ys, xs = np.nonzero(gpu_img != cpu_img)
        first = [(int(y), int(x), hex(int(gpu_img[y, x])), hex(int(cpu_img[y, x]))) for y, x in list(zip(ys, xs))[:5]]
        raise AssertionError(f"{what}: {diff} of {gpu_img.size} pixels differ; first (y,x,gpu,cpu): {first}")


@pytest.mark.parametrize("pipeline", [1, 2])
@pytest.mark.parametrize("spp", [1, 4])
def test_cornell_256_bit_exact(gpu_ctx, oracle, scene_cache, pipeline, spp):
    s = scenes.cornell_box(256, 256)
    p = api.make_params(256, 256, spp=spp, shadow_rays=3, collect_stats=1, pipeline=pipeline)
    scene, frame = _gpu_render(gpu_ctx, s, p)
    ref = oracle.render(s.desc, s.camera, s.scene_info(0), p, bvh=scene.export_bvh(), threads=8)
    _assert_same(frame.download(A.IMAGE_SHADOWED), ref.images[A.IMAGE_SHADOWED], f"cornell256 spp{spp} pipeline{pipeline}")
    g, c = frame.stats(), ref.stats
    for f in ("numRays", "numPrimaryRays", "numShadowRays", "numNodeVisits", "numTriTests", "numHits", "numLightFetches",
              "numLightTriFetches", "algorithmicBytes"):
        assert getattr(g, f) == getattr(c, f), f"counter {f}: gpu {getattr(g, f)} != oracle {getattr(c, f)}"


def test_cornell_256_vs_brute_force(gpu_ctx, oracle, scene_cache):
    """Independent check of packer + BVH builder + traversal: oracle without any BVH."""
    s = scenes.cornell_box(256, 256)
    p = api.make_params(256, 256, spp=2)
    scene, frame = _gpu_render(gpu_ctx, s, p)
    ref = oracle.render(s.desc, s.camera, s.scene_info(0), p, bvh=None, threads=8)
    _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], "cornell256 vs brute force")


def test_cornell_1080p_bit_exact(gpu_ctx, oracle, scene_cache):
    """BASELINE config 2: Cornell box, 1920x1080, 1 spp — pixel-match."""
    s = scenes.cornell_box(1920, 1080)
    p = api.make_params(1920, 1080, spp=1)
    scene, frame = _gpu_render(gpu_ctx, s, p)
    ref = oracle.render(s.desc, s.camera, s.scene_info(0), p, bvh=scene.export_bvh(), threads=16)
    img = frame.download()
    assert img.shape == (1080, 1920)
    _assert_same(img, ref.images[A.IMAGE_SHADOWED], "cornell 1080p")


# ---------------------------------------------------------------------------------------------------
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL5 = A.IMAGES_RAYGEN5 | A.IMG_BIT(A.IMAGE_HDR)
NAMES = {0: "analytic", 1: "shadowed", 2: "unshadowed", 6: "normal", 7: "position"}


@pytest.mark.parametrize("pipeline", [1, 2])
def test_cornell_golden_five_images_and_hdr(gpu_ctx, scene_cache, pipeline):
    """All five ray-gen images (k=5 mode, analytic through synthetic LTC tables) + HDR against the committed golden."""
    from realtimeraytracer_amd import scenes as S
    g = np.load(os.path.join(GOLD, "cornell_256_oracle.npz"))
    s = S.cornell_box(256, 256, ltc=S.synthetic_ltc())
    p = api.make_params(256, 256, spp=1, images=ALL5, collect_stats=1, pipeline=pipeline)
    scene, frame = _gpu_render(gpu_ctx, s, p, images=ALL5)
    for which, name in NAMES.items():
        _assert_same(frame.download(which), g[name], f"golden {name} pipeline{pipeline}")
    hdr = frame.download(A.IMAGE_HDR)
    assert np.array_equal(hdr.view(np.uint32), g["hdr"].view(np.uint32)), "HDR float buffer must match bit for bit (tolerance 0)"
    st = frame.stats()
    assert (st.numRays, st.numPrimaryRays, st.numShadowRays, st.numHits) == tuple(int(x) for x in g["counters"][:4])


def test_analytic_without_ltc_is_refused(gpu_ctx, scene_cache):
    s = scenes.cornell_box(64, 64)
    scene = api.Scene(gpu_ctx, s.desc)
    frame = api.Frame(gpu_ctx, 64, 64, A.IMAGES_RAYGEN5)
    with pytest.raises(api.RtrError) as e:
        api.render(scene, s.camera, s.scene_info(0), api.make_params(64, 64, images=A.IMAGES_RAYGEN5), frame)
    assert e.value.status == -4


def test_error_paths(gpu_ctx, scene_cache):
    s = scenes.cornell_box(64, 64)
    scene = api.Scene(gpu_ctx, s.desc)
    frame = api.Frame(gpu_ctx, 64, 64)
    with pytest.raises(api.RtrError):                              # frame extent mismatch
        api.render(scene, s.camera, s.scene_info(0), api.make_params(128, 64), frame)
    with pytest.raises(api.RtrError):                              # more lights than the scene has
        from realtimeraytracer_amd import host
        api.render(scene, s.camera, host.scene_info(0, 5, s.cam_pos), api.make_params(64, 64), frame)
    with pytest.raises(api.RtrError):                              # HDR accumulation without an HDR image
        api.render(scene, s.camera, s.scene_info(0), api.make_params(64, 64, accumulate=1), frame)
    with pytest.raises(api.RtrError):                              # bandRows must be a multiple of 8
        api.render(scene, s.camera, s.scene_info(0), api.make_params(64, 64, band_rows=4), frame)
    with pytest.raises(api.RtrError):
        frame.download(A.IMAGE_FINAL)


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_band_sharding_reassembles_bit_exact(gpu_ctx, scene_cache, shards):
    """Multi-GPU semantics on one device: N logical shards == the unsharded frame (pixels are independent)."""
    import torch
    from realtimeraytracer_amd import mgpu
    W, H = 320, 180                                                 # 22.5 bands: ragged last band + padding rows
    s = scenes.cornell_box(W, H)
    scene = api.Scene(gpu_ctx, s.desc)
    _, full_frame = _gpu_render(gpu_ctx, s, api.make_params(W, H, spp=2), scene=scene)
    full = full_frame.download()[:H]
    rows = api.shard_rows(H, 8, shards)
    gathered = np.zeros((shards, rows, W), np.uint32)
    for r in range(shards):
        p = api.make_params(W, H, spp=2, shard_index=r, shard_count=shards)
        _, fr = _gpu_render(gpu_ctx, s, p, scene=scene)
        gathered[r] = fr.download()
    _assert_same(mgpu.assemble_numpy(gathered, H, 8), full, f"{shards} shards (host assembly)")
    # the rank-0 de-interleave kernel, as bench.py uses it after the RCCL gather
    g = torch.from_numpy(gathered.view(np.int32)).cuda()
    dst = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()                                        # the upload ran on torch's stream, the kernel runs on the context's
    api.deinterleave_bands(gpu_ctx, g.data_ptr(), dst.data_ptr(), W, H, 8, shards)
    full_frame.wait()                                               # same ctx stream: joins the enqueued kernel
    torch.cuda.synchronize()
    _assert_same(dst.cpu().numpy().view(np.uint32), full, f"{shards} shards (device de-interleave)")


def test_hdr_accumulation_matches_oracle(gpu_ctx, oracle, scene_cache):
    """'N spp accumulated' (BASELINE config 5 semantics): N frames summed in the float HDR buffer, tonemapped once."""
    W, H, N = 160, 96, 4
    s = scenes.cornell_box(W, H)
    imgs = A.IMAGES_FRAMEBUFFER | A.IMG_BIT(A.IMAGE_HDR)
    scene = api.Scene(gpu_ctx, s.desc)
    frame = api.Frame(gpu_ctx, W, H, imgs)
    bvh = scene.export_bvh()
    hdr = np.zeros((H, W, 4), np.float32)
    ref = None
    for f in range(N):
        p = api.make_params(W, H, spp=1, images=imgs, accumulate=1, accumulated_frames=f)
        api.render(scene, s.camera, s.scene_info(f), p, frame)
        ref = oracle.render(s.desc, s.camera, s.scene_info(f), p, bvh=bvh, images=imgs, hdr=hdr, threads=8)
    g_hdr = frame.download(A.IMAGE_HDR)
    assert np.all(g_hdr[..., 3] == N)
    assert np.array_equal(g_hdr.view(np.uint32), hdr.view(np.uint32)), "accumulated HDR must match bit for bit"
    _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], "tonemapped accumulation")
    frame.clear()
    assert not frame.download(A.IMAGE_HDR).any()


@pytest.mark.parametrize("collect", [0, 1])
@pytest.mark.parametrize("pipeline", [1, 2])
def test_bunny_class_parity(gpu_ctx, oracle, scene_cache, pipeline, collect):
    """BASELINE config 3 geometry (81,920-triangle displaced icosphere, smooth normals), 4 spp, reduced extent.  collect = 0: the
    kernels the bench times (k_primary with its 16-entry LDS stack + redo tail, the spp > 1 branch of k_shadow_gen, the timed form
    of k_shadow_trace4), compared with the oracle directly; collect = 1: their counting forms, counters included."""
    W, H = 480, 272
    s = scenes.bunny_class(W, H)
    p = api.make_params(W, H, spp=4, collect_stats=collect, pipeline=pipeline)
    scene, frame = _gpu_render(gpu_ctx, s, p)
    assert scene.stats().numTriangles == 81920 + 512 + 2
    ref = oracle.render(s.desc, s.camera, s.scene_info(0), p, bvh=scene.export_bvh(), threads=16)
    _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], f"bunny-class pipeline{pipeline} collect{collect}")
    if collect:
        g = frame.stats()
        assert (g.numRays, g.numNodeVisits, g.numTriTests, g.numHits) == (ref.stats.numRays, ref.stats.numNodeVisits, ref.stats.numTriTests, ref.stats.numHits)


@pytest.mark.parametrize("collect", [0, 1])
@pytest.mark.parametrize("pipeline", [1, 2])
def test_sponza_class_parity(gpu_ctx, oracle, scene_cache, pipeline, collect):
    """BASELINE config 4 geometry (262 k triangles, 2 area lights), 1 spp, reduced extent, full oracle compare — the timed kernels
    (collect = 0) and their counting forms (collect = 1), each against the oracle directly."""
    W, H = 640, 360
    s = scenes.sponza_class(W, H)
    p = api.make_params(W, H, spp=1, collect_stats=collect, pipeline=pipeline)
    scene, frame = _gpu_render(gpu_ctx, s, p, frame_no=5)
    ref = oracle.render(s.desc, s.camera, s.scene_info(5), p, bvh=scene.export_bvh(), threads=16)
    _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], f"sponza-class pipeline{pipeline} collect{collect}")
    if collect:
        g = frame.stats()
        for f in ("numRays", "numShadowRays", "numNodeVisits", "numTriTests", "numShadowNodeVisits", "numShadowTriTests", "numHits", "shadowTraceBytes"):
            assert getattr(g, f) == getattr(ref.stats, f), f
        if pipeline == 2:
            # every lane counted in a trip of the node loop visited one 4-wide record there (tail rays walk the BVH2 elsewhere)
            assert g.shadowInnerIterations > 0 and (g.shadowInnerActiveLanes == g.numShadowNodeVisits or g.shadowTailRays > 0)
            assert g.shadowTriIterations > 0 and (g.shadowTriActiveLanes == g.numShadowTriTests or g.shadowTailRays > 0)
            assert g.shadowTraceClockMHz > 500.0


def test_sponza_1080p_properties_and_sampled_oracle(gpu_ctx, oracle, scene_cache):
    """BASELINE.json full size (config 4: 1920x1080, 1 spp).  Size-independent properties: megakernel == wavefront,
    idempotence, 8-shard reassembly == unsharded; plus the oracle on every 8th band (1/8 of the frame)."""
    from realtimeraytracer_amd import mgpu
    W, H = 1920, 1080
    s = scenes.sponza_class(W, H)
    scene = api.Scene(gpu_ctx, s.desc)
    gpu_ctx.set_tunable("primary_packet", 1)            # camera rays as 8x8 packets (k_primary_packet): one stack per tile, no ray is ever abandoned
    try:
        _, fw = _gpu_render(gpu_ctx, s, api.make_params(W, H, collect_stats=1, pipeline=2), scene=scene)
    finally:
        gpu_ctx.set_tunable("primary_packet", 0)
    wave = fw.download()
    stp = fw.stats()
    assert stp.primaryTailRays == 0
    _, fw = _gpu_render(gpu_ctx, s, api.make_params(W, H, collect_stats=1, pipeline=2), scene=scene, frame=fw)       # ... and one ray per lane (the default): the same frame
    _assert_same(fw.download(), wave, "camera rays one per lane vs as packets at 1080p")
    stw = fw.stats()
    _, fm = _gpu_render(gpu_ctx, s, api.make_params(W, H, collect_stats=1, pipeline=1), scene=scene)
    _assert_same(fm.download(), wave, "megakernel vs wavefront at 1080p")
    stm = fm.stats()
    # one ray per lane, camera rays walk the BVH2 in both pipelines (the staged one with a 16-entry stack + redo: the redone rays'
    # visits are counted twice); shadow rays walk the wide view in the staged pipeline only
    assert stw.numPrimaryRays == W * H and stw.numRays == stm.numRays == stp.numRays
    pw, pm = stw.numNodeVisits - stw.numShadowNodeVisits, stm.numNodeVisits - stm.numShadowNodeVisits
    assert pw >= pm and (pw == pm) == (stw.primaryTailRays == 0)
    assert stp.numShadowNodeVisits == stw.numShadowNodeVisits and stp.numHits == stw.numHits          # the walk of the camera rays changes their work only
    api.render(scene, s.camera, s.scene_info(0), api.make_params(W, H, pipeline=2), fw)
    _assert_same(fw.download(), wave, "idempotence")
    rows = api.shard_rows(H, 8, 8)
    gathered = np.zeros((8, rows, W), np.uint32)
    for r in range(8):
        _, fr = _gpu_render(gpu_ctx, s, api.make_params(W, H, shard_index=r, shard_count=8), scene=scene)
        gathered[r] = fr.download()
    _assert_same(mgpu.assemble_numpy(gathered, H, 8), wave, "8-shard reassembly at 1080p")
    p8 = api.make_params(W, H, shard_index=0, shard_count=8)
    ref = oracle.render(s.desc, s.camera, s.scene_info(0), p8, bvh=scene.export_bvh(), threads=16)
    _assert_same(gathered[0], ref.images[A.IMAGE_SHADOWED], "shard 0 of 8 vs oracle at 1080p")


def test_bunny_1080p_4spp_properties_and_sampled_oracle(gpu_ctx, oracle, scene_cache):
    """BASELINE.json config 3 at its own size (bunny-class, 1920x1080, 4 spp = the reference's NUM_PRIMARY_RAYS, raygen.rgen:8) on the
    timed kernels: megakernel == wavefront, idempotence, 8-shard reassembly == unsharded, and the oracle on every 8th band (shard
    0 of 8).  The fixture runs it with the plain and with the octant-binned queue."""
    from realtimeraytracer_amd import mgpu
    W, H = 1920, 1080
    s = scenes.bunny_class(W, H)
    scene = api.Scene(gpu_ctx, s.desc)
    _, fw = _gpu_render(gpu_ctx, s, api.make_params(W, H, spp=4, pipeline=2), scene=scene)
    wave = fw.download()
    _, fm = _gpu_render(gpu_ctx, s, api.make_params(W, H, spp=4, pipeline=1), scene=scene)
    _assert_same(fm.download(), wave, "bunny 1080p 4 spp: megakernel vs wavefront")
    api.render(scene, s.camera, s.scene_info(0), api.make_params(W, H, spp=4, pipeline=2), fw)
    _assert_same(fw.download(), wave, "bunny 1080p 4 spp: idempotence")
    rows = api.shard_rows(H, 8, 8)
    gathered = np.zeros((8, rows, W), np.uint32)
    for r in range(8):
        _, fr = _gpu_render(gpu_ctx, s, api.make_params(W, H, spp=4, shard_index=r, shard_count=8), scene=scene)
        gathered[r] = fr.download()
    _assert_same(mgpu.assemble_numpy(gathered, H, 8), wave, "bunny 1080p 4 spp: 8-shard reassembly")
    p8 = api.make_params(W, H, spp=4, shard_index=0, shard_count=8)
    ref = oracle.render(s.desc, s.camera, s.scene_info(0), p8, bvh=scene.export_bvh(), threads=16)
    _assert_same(gathered[0], ref.images[A.IMAGE_SHADOWED], "bunny 1080p 4 spp: shard 0 of 8 vs oracle")


def test_4k_accumulated_sample(gpu_ctx, oracle, scene_cache):
    """BASELINE config 5 as written — Sponza-class, 3840x2160, 16 frames accumulated in the float HDR buffer, 8 band-shards — on ONE of the
    eight shards (272 rows, 1.04 M pixels, 16 x 14 M rays) against the oracle: HDR bits and framebuffer bytes after the sixteenth frame."""
    W, H, N, SH = 3840, 2160, 16, 8
    s = scenes.sponza_class(W, H)
    imgs = A.IMAGES_FRAMEBUFFER | A.IMG_BIT(A.IMAGE_HDR)
    scene = api.Scene(gpu_ctx, s.desc)
    rows = api.shard_rows(H, 8, SH)
    frame = api.Frame(gpu_ctx, W, rows, imgs)
    bvh = scene.export_bvh()
    hdr = np.zeros((rows, W, 4), np.float32)
    for f in range(N):
        p = api.make_params(W, H, images=imgs, accumulate=1, accumulated_frames=f, shard_index=0, shard_count=SH)
        api.render(scene, s.camera, s.scene_info(f), p, frame)
        ref = oracle.render(s.desc, s.camera, s.scene_info(f), p, bvh=bvh, images=imgs, hdr=hdr, threads=16)
    assert np.array_equal(frame.download(A.IMAGE_HDR).view(np.uint32), hdr.view(np.uint32))
    _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], "4K accumulated shard")
    assert np.all(frame.download(A.IMAGE_HDR)[api.shard_rows(H, 8, SH) - 8 - 1, :, 3] == N)       # sixteen frames in every live pixel's sum
    # the whole 4K frame renders and every live pixel is opaque; its rows of shard 0 are the shard's first frame
    full = api.Frame(gpu_ctx, W, H, A.IMAGES_FRAMEBUFFER)
    api.render(scene, s.camera, s.scene_info(0), api.make_params(W, H), full)
    img = full.download()
    assert np.all((img >> 24) == 0xff)


def test_update_lights(gpu_ctx, oracle, scene_cache):
    import copy
    s = scenes.cornell_box(128, 128)
    scene = api.Scene(gpu_ctx, s.desc)
    frame = api.Frame(gpu_ctx, 128, 128)
    lights = s.host.lightInfos()
    new = A.RtrAreaLightInfo.from_buffer_copy(bytes(lights[0]))
    new.color[0], new.color[1], new.color[2], new.intensity, new.isTwoSided = 0.2, 0.9, 0.4, 35.0, 1
    scene.update_lights([new])
    p = api.make_params(128, 128)
    api.render(scene, s.camera, s.scene_info(0), p, frame)
    # oracle with the same edit applied to a copy of the descriptor's light array
    arr = (A.RtrAreaLightInfo * 1)(new)
    d = A.rtr_scene_desc.from_buffer_copy(bytes(s.desc))
    import ctypes as C
    d.lights = C.cast(arr, C.POINTER(A.RtrAreaLightInfo))
    ref = oracle.render(d, s.camera, s.scene_info(0), p, bvh=scene.export_bvh(), threads=8)
    _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], "after rtr_scene_update_lights")
    moved = A.RtrAreaLightInfo.from_buffer_copy(bytes(new))
    moved.transform[12] += 10.0
    with pytest.raises(api.RtrError):                              # geometry is baked into the BVH: refuse, do not ignore
        scene.update_lights([moved])


def test_async_render_into_external_torch_tensor(gpu_ctx, oracle, scene_cache):
    """What bench.py does for the RCCL gather: the framebuffer lives in a torch tensor, work is enqueued on a torch stream."""
    import torch
    W, H = 256, 128
    s = scenes.cornell_box(W, H)
    ctx = api.Context(0)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    scene = api.Scene(ctx, s.desc)
    frame = api.Frame(ctx, W, H)
    t = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    frame.bind_external(A.IMAGE_SHADOWED, t.data_ptr(), t.numel() * 4)
    p = api.make_params(W, H, spp=2)
    with torch.cuda.stream(stream):
        api.render(scene, s.camera, s.scene_info(2), p, frame, asynchronous=True)
    frame.wait()
    ref = oracle.render(s.desc, s.camera, s.scene_info(2), p, bvh=scene.export_bvh(), threads=8)
    _assert_same(t.cpu().numpy().view(np.uint32), ref.images[A.IMAGE_SHADOWED], "external tensor / async")
    assert frame.stats().totalMs > 0
    for o in (frame, scene, ctx):
        o.close()


def test_frames_in_flight_share_one_scene(gpu_ctx, oracle, scene_cache):
    """bench.py's pipelining: several contexts (one stream each) render different frames of ONE scene concurrently;
    each frame's work is ordered on its own context's stream and every image still equals the oracle's."""
    import torch
    W, H = 320, 200
    s = scenes.cornell_box(W, H)
    scene = api.Scene(gpu_ctx, s.desc)
    bvh = scene.export_bvh()
    p = api.make_params(W, H, spp=2)
    ctxs, streams, frames = [], [], []
    for _ in range(3):
        c = api.Context(0); st = torch.cuda.Stream(); c.set_stream(st.cuda_stream)
        ctxs.append(c); streams.append(st); frames.append(api.Frame(c, W, H))
    for rnd in range(2):                                           # buffers reused once, as the bench does
        for b in range(3):
            frames[b].wait()
            api.render(scene, s.camera, s.scene_info(3 * rnd + b), p, frames[b], asynchronous=True)
    for b in range(3):
        frames[b].wait()
        ref = oracle.render(s.desc, s.camera, s.scene_info(3 + b), p, bvh=bvh, threads=8)
        _assert_same(frames[b].download(), ref.images[A.IMAGE_SHADOWED], f"frame in flight {b}")
        assert frames[b].stats().shadowTraceMs > 0
    for o in frames + ctxs + [scene]:
        o.close()


def test_empty_scene_renders_sky(gpu_ctx, oracle):
    from realtimeraytracer_amd import host
    d = A.rtr_scene_desc()
    d.skyColor[0], d.skyColor[1], d.skyColor[2] = 0.5, 0.7, 1.0
    scene = api.Scene(gpu_ctx, d)
    frame = api.Frame(gpu_ctx, 40, 24)
    cam = host.Camera(60, (0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 24).getGPUData()
    p = api.make_params(40, 24, collect_stats=1)
    api.render(scene, cam, host.scene_info(0, 0, (0, 0, 5)), p, frame)
    ref = oracle.render(d, cam, host.scene_info(0, 0, (0, 0, 5)), p, bvh=scene.export_bvh(), threads=1)
    _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], "empty scene")
    assert frame.stats().numRays == 40 * 24 and frame.stats().numHits == 0


@pytest.mark.parametrize("size", [(256, 256), (333, 187)])
def test_denoise_combine_matches_oracle(gpu_ctx, oracle, scene_cache, size):
    """the 8 a-trous dispatches + combine of the reference frame loop, bit-exact on all five
    images they touch (sampled pair is overwritten by the ping-pong; quirks Q8-Q10 kept)."""
    W, H = size
    all8 = 0xff
    s = scenes.cornell_box(W, H, ltc=scenes.synthetic_ltc())
    scene = api.Scene(gpu_ctx, s.desc)
    frame = api.Frame(gpu_ctx, W, H, all8)
    p = api.make_params(W, H, spp=2, images=A.IMAGES_RAYGEN5)
    api.render(scene, s.camera, s.scene_info(0), p, frame)
    src = {k: frame.download(k) for k in (0, 1, 2, 6, 7)}
    frame.denoise_combine(4)
    ref = oracle.denoise_combine(src[0], src[1], src[2], src[6], src[7], iterations=4)
    for which in (A.IMAGE_SHADOWED, A.IMAGE_UNSHADOWED, A.IMAGE_DENOISED_SHADOWED, A.IMAGE_DENOISED_UNSHADOWED, A.IMAGE_FINAL):
        _assert_same(frame.download(which), ref[which], f"denoise/combine image {which} at {W}x{H}")
    fin = frame.download(A.IMAGE_FINAL)
    assert np.all((fin >> 24) == 0xff) and len(np.unique(fin)) > 100
    small = api.Frame(gpu_ctx, W, H, A.IMAGES_RAYGEN5)
    with pytest.raises(api.RtrError):
        small.denoise_combine(4)


@pytest.mark.parametrize("collect", [0, 1])
@pytest.mark.parametrize("pipeline", [1, 2])
def test_textured_room_parity(gpu_ctx, oracle, scene_cache, pipeline, collect):
    """texture maps (colour / specular / metallic), alpha-tested any-hit on both closest-hit and
    shadow rays, repeat addressing with tiled and negative uvs, equirect HDRI miss — all five images, bit-exact, from the timed
    kernels (collect = 0) and from their counting forms (collect = 1)."""
    W, H = 400, 248
    s = scenes.textured_room(W, H, ltc=scenes.synthetic_ltc())
    p = api.make_params(W, H, spp=2, images=A.IMAGES_RAYGEN5, collect_stats=collect, pipeline=pipeline)
    scene, frame = _gpu_render(gpu_ctx, s, p, images=A.IMAGES_RAYGEN5, frame_no=3)
    ref = oracle.render(s.desc, s.camera, s.scene_info(3), p, bvh=scene.export_bvh(), images=A.IMAGES_RAYGEN5, threads=16)
    for which, name in NAMES.items():
        _assert_same(frame.download(which), ref.images[which], f"textured room {name} pipeline{pipeline} collect{collect}")
    if collect:
        g = frame.stats()
        for f in ("numRays", "numNodeVisits", "numTriTests", "numHits", "numTexFetches", "numAlphaTests", "algorithmicBytes"):
            assert getattr(g, f) == getattr(ref.stats, f), f
        assert g.numAlphaTests > 10000 and g.numTexFetches > g.numHits


@pytest.mark.parametrize("collect", [0, 1])
def test_sponza_mixed_parity(gpu_ctx, oracle, scene_cache, queue_mode, collect):
    """The atrium with a real asset's triangle-size mix (scenes.sponza_mixed: large architecture triangles and column slivers beside
    fine cloth, an alpha-tested ivy layer; the host builder's insertion-based optimisation runs on it): the timed kernels and their
    counting forms against the oracle on the tree the GPU exported — image bytes and work counters."""
    W, H = 480, 272
    s = scenes.sponza_mixed(W, H)
    p = api.make_params(W, H, spp=1, collect_stats=collect, pipeline=2)
    scene, frame = _gpu_render(gpu_ctx, s, p, frame_no=4)
    ref = oracle.render(s.desc, s.camera, s.scene_info(4), p, bvh=scene.export_bvh(), threads=16)
    _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], f"sponza_mixed collect{collect}")
    if collect:
        g = frame.stats()
        for f in ("numRays", "numNodeVisits", "numTriTests", "numHits", "numShadowNodeVisits", "numShadowTriTests", "numAlphaTests", "shadowTailRays"):
            assert getattr(g, f) == getattr(ref.stats, f), f
        assert g.numAlphaTests > 1000


def test_missing_texture_is_refused_on_device_path(gpu_ctx, scene_cache):
    import ctypes as C
    s = scenes.cornell_box(32, 32)
    d = A.rtr_scene_desc.from_buffer_copy(bytes(s.desc))
    objs = (A.RtrObjectInfo * d.numObjects)(*[s.desc.objects[i] for i in range(d.numObjects)])
    objs[0].usesOpacityMap, objs[0].opacityIndex = 1, 2
    d.objects = C.cast(objs, C.POINTER(A.RtrObjectInfo))
    with pytest.raises(api.RtrError) as e:
        api.Scene(gpu_ctx, d)
    assert e.value.status == -1 and "opacity map" in str(e.value)


def _desc_from_arrays(verts, tris_idx, mesh_ranges, instances, objects, lights, sky=(0.5, 0.7, 1.0)):
    """Hand-built rtr_scene_desc (no OBJ round trip): verts (V,3), per-mesh (vertexOffset, indexOffset, vertexCount, indexCount),
    instances [(meshIndex, customIndex)], objects [RtrObjectInfo], lights [RtrAreaLightInfo].  Returns (desc, keepalive)."""
    import ctypes as C
    V = np.zeros((len(verts), 12), np.float32)
    V[:, :3] = verts
    idx = np.ascontiguousarray(tris_idx, np.uint32).reshape(-1)
    meshes = (A.RtrMesh * len(mesh_ranges))()
    for m, (vo, io, vc, ic) in zip(meshes, mesh_ranges):
        m.vertexOffset, m.indexOffset, m.vertexCount, m.indexCount, m.isOpaque = vo, io, vc, ic, 1
    inst = (A.RtrInstance * len(instances))()
    for i, (mi, ci) in zip(inst, instances):
        i.meshIndex, i.customIndex = mi, ci
        for k, v in enumerate((1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0)):
            i.transform[k] = float(v)
    objs = (A.RtrObjectInfo * max(len(objects), 1))(*objects)
    lts = (A.RtrAreaLightInfo * max(len(lights), 1))(*lights)
    d = A.rtr_scene_desc()
    d.vertices = V.ctypes.data_as(C.POINTER(A.RtrVertex)); d.numVertices = len(V)
    d.indices = idx.ctypes.data_as(C.POINTER(A.u32)); d.numIndices = len(idx)
    d.meshes, d.numMeshes = meshes, len(mesh_ranges)
    d.instances, d.numInstances = inst, len(instances)
    d.objects, d.numObjects = objs, len(objects)
    d.lights, d.numLights = lts, len(lights)
    d.skyColor[0], d.skyColor[1], d.skyColor[2] = sky
    return d, (V, idx, meshes, inst, objs, lts)


def test_deep_stack_rays_take_the_tail_kernel(gpu_ctx, oracle):
    """Rays that need more than the 16 LDS stack entries of k_shadow_trace: a 2^19-triangle row traversed end to end by
    skimming shadow rays (both children hit at every level).  The production kernel hands them to k_shadow_tail, the
    counting kernel spills to global memory; both must agree with the oracle (image, and counters for the latter)."""
    from realtimeraytracer_amd import host
    N = 1 << 19
    x = np.arange(N, dtype=np.float32)
    tri = np.stack([np.stack([x, np.full(N, -1.0, np.float32), np.full(N, -0.3, np.float32)], 1),
                    np.stack([x + 0.6, np.full(N, -1.0, np.float32), np.zeros(N, np.float32)], 1),
                    np.stack([x, np.full(N, -1.0, np.float32), np.full(N, 0.3, np.float32)], 1)], 1).reshape(-1, 3)
    recv = np.array([[-8, 0.3, -2], [-8, 0.3, 2], [-2, 0.3, 2], [-2, 0.3, -2]], np.float32)          # receiver, normal +y (geometric)
    lx = float(N + 50)
    lightq = np.array([[lx, -0.5, -0.5], [lx, 0.9, -0.5], [lx, 0.9, 0.5], [lx, -0.5, 0.5]], np.float32)
    verts = np.concatenate([lightq, recv, tri])
    idx = np.concatenate([np.array([0, 1, 2, 0, 2, 3], np.uint32), np.array([0, 2, 1, 0, 3, 2], np.uint32), np.arange(3 * N, dtype=np.uint32)])
    meshes = [(0, 0, 4, 6), (4, 6, 4, 6), (8, 12, 3 * N, 3 * N)]
    L = A.RtrAreaLightInfo()
    L.color[0], L.color[1], L.color[2], L.intensity = 1.0, 0.9, 0.8, 4.0e9
    L.vertexOffset, L.indexOffset, L.numTriangles, L.isTwoSided = 0, 0, 2, 1
    for k, v in enumerate(np.eye(4, dtype=np.float32).reshape(-1)):
        L.transform[k] = float(v)
    objs = []
    for vo, io in ((4, 6), (8, 12)):
        o = A.RtrObjectInfo()
        o.vertexOffset, o.indexOffset = vo, io
        o.color[0], o.color[1], o.color[2], o.specular, o.metallic = 0.8, 0.8, 0.8, 0.3, 0.0
        objs.append(o)
    d, keep = _desc_from_arrays(verts, idx, meshes, [(0, 0), (1, 1), (2, 2)], objs, [L])
    scene = api.Scene(gpu_ctx, d)
    st = scene.stats()
    assert st.numTriangles == N + 4 and st.maxDepth > 16, st.maxDepth
    W, H = 16, 8
    cam = host.Camera(30.0, (-5.0, 12.0, 0.0), (-5.0, 0.3, 0.01), (1.0, 0.0, 0.0), W, H).getGPUData()
    info = host.scene_info(0, 1, (-5.0, 12.0, 0.0))
    bvh = scene.export_bvh()
    imgs = {}
    for collect in (0, 1):                                   # timed form, then counting form, of the same kernel (+ tail kernel)
        p = api.make_params(W, H, spp=1, collect_stats=collect, pipeline=2)
        frame = api.Frame(gpu_ctx, W, H)
        api.render(scene, cam, info, p, frame)
        imgs[collect] = frame.download()
        if collect:
            ref = oracle.render(d, cam, info, p, bvh=bvh, threads=16)
            _assert_same(imgs[1], ref.images[A.IMAGE_SHADOWED], "deep-stack scene, counting kernel")
            g = frame.stats()
            assert g.numShadowRays == ref.stats.numShadowRays > 0
            # the rays that outgrow the 16-entry LDS stack are abandoned and redone over the BVH2 by k_shadow_tail; the counting
            # form counts both parts, and so does the oracle's restatement of the walk
            assert g.shadowTailRays > 0, "the skimming shadow rays must overflow the LDS stack"
            assert g.numNodeVisits == ref.stats.numNodeVisits and g.numTriTests == ref.stats.numTriTests
            assert ref.stats.numNodeVisits / ref.stats.numRays > 10000, "the skimming rays must really walk the whole row"
    _assert_same(imgs[0], imgs[1], "timed form vs counting form (LDS stack + overflow tail)")
    # the list of abandoned rays holds 1/16 of the queue (at least 2^20 entries); past that k_shadow_tail redoes the whole queue.
    # Forced here with a list of 4 entries: the same image
    # (RTR_TRACE_OVERFLOW_CAP exists only in librtr_hip_test.so, the product's sources built with -DRTR_TEST_HOOKS)
    os.environ["RTR_TRACE_OVERFLOW_CAP"] = "4"
    try:
        hctx = api.Context(0, test_hooks=True)
        hscene = api.Scene(hctx, d)
        fo = api.Frame(hctx, W, H)
        api.render(hscene, cam, info, api.make_params(W, H, spp=1, collect_stats=1, pipeline=2), fo)
        assert fo.stats().shadowTailRays > 4
        _assert_same(fo.download(), imgs[1], "abandoned-ray list overflowed: the redo kernel walks the whole queue")
        api.render(hscene, cam, info, api.make_params(W, H, spp=1, pipeline=2), fo)
        _assert_same(fo.download(), imgs[1], "abandoned-ray list overflowed, timed form")
        fo.close(); hscene.close(); hctx.close()
    finally:
        del os.environ["RTR_TRACE_OVERFLOW_CAP"]
    fm = api.Frame(gpu_ctx, W, H)
    api.render(scene, cam, info, api.make_params(W, H, spp=1, pipeline=1), fm)
    _assert_same(fm.download(), imgs[1], "megakernel on the deep-stack scene")
    # PRIMARY rays that need more than k_primary's 16 LDS entries: the same row squeezed to 0.01 units per triangle (primary
    # rays end at t = 10000) and a camera at its head looking down its length — ordered traversal keeps one pending far child
    # per level of the ~20-level tree -> k_primary_tail re-traces those pixel-samples.  (At this grazing angle Moeller-Trumbore's
    # |a| < EPSILON rule rejects every triangle, so the picture is sky: what is checked is that the abandoned pixel-samples get
    # their hit records from the tail kernel at all.)
    tri2 = tri.copy(); tri2[:, 0] *= np.float32(0.01)
    verts2 = np.concatenate([lightq * np.array([0.01, 1, 1], np.float32), recv, tri2])
    d2, keep2 = _desc_from_arrays(verts2, idx, meshes, [(0, 0), (1, 1), (2, 2)], objs, [L])
    scene2 = api.Scene(gpu_ctx, d2)
    assert scene2.stats().maxDepth > 16
    bvh2 = scene2.export_bvh()
    end = float(N) * 0.01
    cam2 = host.Camera(0.004, (-30.0, -0.995, 0.0), (0.8 * end, -1.0, 0.0), (0.0, 1.0, 0.0), W, H).getGPUData()    # inside the boxes' padding all the way
    info2 = host.scene_info(1, 1, (-30.0, -0.995, 0.0))
    out = {}
    # as 8x8 packets (one stack per tile, as deep as the tree: > 16 entries here), one ray per lane over the BVH2 (16 LDS entries + redo
    # tail) and one ray per lane over the 4-wide view (k_primary4: 16 entries, its abandoned rays walked again over the BVH2)
    for walk in ("packet", "solo", "wide"):
        gpu_ctx.set_tunable("primary_packet", int(walk == "packet"))
        gpu_ctx.set_tunable("primary_wide", int(walk == "wide"))
        try:
            for collect in (0, 1):
                p = api.make_params(W, H, spp=2, collect_stats=collect, pipeline=2)
                f2 = api.Frame(gpu_ctx, W, H)
                api.render(scene2, cam2, info2, p, f2)
                out[walk, collect] = f2.download()
                if collect:
                    ref2 = oracle.render(d2, cam2, info2, p, bvh=bvh2, threads=16, primary_packets=walk == "packet", primary_wide=walk == "wide")
                    _assert_same(out[walk, 1], ref2.images[A.IMAGE_SHADOWED], "skimming primary rays, counting kernels")
                    g2 = f2.stats()
                    assert (g2.numNodeVisits, g2.numTriTests, g2.numHits, g2.primaryTailRays) == (ref2.stats.numNodeVisits, ref2.stats.numTriTests, ref2.stats.numHits, ref2.stats.primaryTailRays), walk
                    assert (g2.numNodeVisits - g2.numShadowNodeVisits) / g2.numPrimaryRays > (1000 if walk != "wide" else 300), "primary rays must walk a long stretch of the row"
                    if walk != "wide":
                        assert (g2.primaryTailRays > 0) == (walk == "solo")
        finally:
            gpu_ctx.set_tunable("primary_packet", 0)
            gpu_ctx.set_tunable("primary_wide", 0)
        _assert_same(out[walk, 0], out[walk, 1], "timed camera-ray kernel vs its counting form")
    _assert_same(out["solo", 0], out["packet", 0], "camera rays one per lane (16 LDS entries + redo tail) vs as packets")
    _assert_same(out["solo", 0], out["wide", 0], "camera rays over the BVH2 vs over the 4-wide view")


def test_context_may_be_destroyed_before_its_children(gpu_ctx, scene_cache):
    """Garbage-collected bindings destroy objects in any order: a context outlives its last scene / frame internally."""
    s = scenes.cornell_box(64, 64)
    ctx = api.Context(0)
    scene = api.Scene(ctx, s.desc)
    frame = api.Frame(ctx, 64, 64)
    api.render(scene, s.camera, s.scene_info(0), api.make_params(64, 64), frame)
    before = frame.download()
    ctx.close()                                   # children still alive
    assert np.array_equal(frame.download(), before)
    frame.close()
    scene.close()                                 # the last child releases the context
    ctx2 = api.Context(0)
    ctx2.close()


def test_two_wide_any_hit_kernel_still_matches(scene_cache, tmp_path):
    """The production any-hit kernel walks the 4-wide view of the tree; RTR_TRACE_BVH4=0 selects the 2-wide kernel it grew out of
    (kept for A/B runs).  The switch is read once per process, so it is exercised in a child process: same picture as the
    oracle's."""
    import os
    import subprocess
    import sys
    code = """
import numpy as np
from realtimeraytracer_amd import _abi as A, api, scenes
from oracle import oracle_py as O
s = scenes.bunny_class(160, 96, subdiv=4)
ctx = api.Context(0); scene = api.Scene(ctx, s.desc); frame = api.Frame(ctx, 160, 96)
p = api.make_params(160, 96, spp=2)
api.render(scene, s.camera, s.scene_info(3), p, frame)
ref = O.render(s.desc, s.camera, s.scene_info(3), p, bvh=scene.export_bvh(), threads=8)
assert np.array_equal(frame.download(), ref.images[A.IMAGE_SHADOWED])
try:        # no counting form for the comparison kernel: refused, not answered with another kernel's counters
    api.render(scene, s.camera, s.scene_info(3), api.make_params(160, 96, spp=2, collect_stats=1), frame)
    raise SystemExit("collectStats was accepted with the 2-wide kernel")
except api.RtrError as e:
    assert "counting form" in str(e)
print("two-wide ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTR_TRACE_BVH4="0", PYTHONPATH=root, RTR_SCENE_CACHE=str(scene_cache))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "two-wide ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_persistent_camera_ray_kernel_matches(scene_cache, tmp_path):
    """RTR_PRIMARY_PERSIST=1 selects k_primary_persist (persistent waves with ballot refill over the camera rays; off by default: it
    loses with several frames in flight, profiles/r02/ab_primary_kernels.log).  Same walk per ray as k_primary, so: same five
    images as the oracle's from the timed form, same counters from the counting form — in a child process, the switch being read
    once per process."""
    import os
    import subprocess
    import sys
    code = """
import numpy as np
from realtimeraytracer_amd import _abi as A, api, scenes
from oracle import oracle_py as O
W, H = 333, 187
s = scenes.textured_room(W, H, ltc=scenes.synthetic_ltc())
ctx = api.Context(0); scene = api.Scene(ctx, s.desc); frame = api.Frame(ctx, W, H, A.IMAGES_RAYGEN5)
for collect in (0, 1):
    p = api.make_params(W, H, spp=3, images=A.IMAGES_RAYGEN5, collect_stats=collect)
    api.render(scene, s.camera, s.scene_info(5), p, frame)
    ref = O.render(s.desc, s.camera, s.scene_info(5), p, bvh=scene.export_bvh(), images=A.IMAGES_RAYGEN5, threads=8)
    for which in (0, 1, 2, 6, 7):
        assert np.array_equal(frame.download(which), ref.images[which]), (collect, which)
    if collect:
        g = frame.stats()
        for f in ("numRays", "numPrimaryRays", "numNodeVisits", "numTriTests", "numHits", "numAlphaTests", "primaryTailRays"):
            assert getattr(g, f) == getattr(ref.stats, f), f
print("persistent ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTR_PRIMARY_PERSIST="1", PYTHONPATH=root, RTR_SCENE_CACHE=str(scene_cache))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "persistent ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("batch", ["64", "512", "4096"])
def test_any_hit_batch_length_does_not_change_the_frame(scene_cache, batch):
    """RTR_TRACE_BATCH (rays a wave reserves per cursor atomic; by default 256, or 512 on the queue of a launch of several frames):
    which wave traces which rays changes, no ray's walk does — the frames of a four-frame launch equal the oracle's and the counting
    form's work counters equal the oracle's sum, in both queue modes.  In a child process: the knob is read once per process."""
    import os
    import subprocess
    import sys
    code = """
import numpy as np
from realtimeraytracer_amd import _abi as A, api, scenes
from oracle import oracle_py as O
W, H, N = 320, 184, 4
s = scenes.bunny_class(W, H)
ctx = api.Context(0); scene = api.Scene(ctx, s.desc); bvh = scene.export_bvh()
frames = [api.Frame(ctx, W, H) for _ in range(N)]
for collect in (0, 1):
    p = api.make_params(W, H, spp=2, collect_stats=collect, pipeline=2)
    api.render_batch(scene, [s.camera] * N, [s.scene_info(3 + b) for b in range(N)], p, frames)
    frames[0].wait()
    tot = {}
    for b in range(N):
        ref = O.render(s.desc, s.camera, s.scene_info(3 + b), p, bvh=bvh, threads=8)
        assert np.array_equal(frames[b].download(), ref.images[A.IMAGE_SHADOWED]), (collect, b)
        for f in ("numRays", "numShadowRays", "numShadowNodeVisits", "numShadowTriTests", "numNodeVisits", "numTriTests"):
            tot[f] = tot.get(f, 0) + getattr(ref.stats, f)
    if collect:
        g = frames[0].stats()
        for f, v in tot.items():
            assert getattr(g, f) == v, (f, getattr(g, f), v)
print("batch ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTR_TRACE_BATCH=batch, PYTHONPATH=root, RTR_SCENE_CACHE=str(scene_cache))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "batch ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_batch_limit_is_the_launch_that_still_fits(gpu_ctx, scene_cache):
    """rtr_render_batch_limit: RTR_MAX_BATCH where everything fits, the arithmetic of the 31-bit visibility slots where it does not
    (1080p at 4 spp with 13 queries per pixel-sample: 16 frames), 1 for the megakernel — and a launch of one frame more than the limit
    is refused before anything is allocated."""
    s = scenes.sponza_class(1920, 1080)
    scene = api.Scene(gpu_ctx, s.desc)
    nl = s.num_lights
    assert api.render_batch_limit(scene, api.make_params(1920, 1080, spp=1), nl) == A.MAX_BATCH
    assert api.render_batch_limit(scene, api.make_params(1920, 1080, spp=1, pipeline=1), nl) == 1
    queries = 1 + 3 * 2 * nl                                   # 3 shadow rays x 2 triangles per light quad + the directional light

    def expect(w, h, spp, shards=1):
        rows = api.shard_rows(h, 8, shards)
        blocks = (((rows + 7) // 8) * ((w + 7) // 8) * 64 + 255) // 256
        for n in range(A.MAX_BATCH, 1, -1):
            stride = 256
            while stride < blocks * 256 * spp * n:
                stride *= 2
            if stride * queries < 2 ** 31:
                return n
        return 1
    for (w, h, spp, shards) in ((1920, 1080, 4, 1), (3840, 2160, 1, 1), (3840, 2160, 4, 1), (3840, 2160, 16, 1), (1920, 1080, 4, 8), (7680, 4320, 8, 2)):
        p = api.make_params(w, h, spp=spp, shard_index=0, shard_count=shards)
        assert api.render_batch_limit(scene, p, nl) == expect(w, h, spp, shards), (w, h, spp, shards)
    assert expect(1920, 1080, 4) == 16
    p = api.make_params(1920, 1080, spp=4, pipeline=2)
    frames = [api.Frame(gpu_ctx, 1920, 1080) for _ in range(17)]
    with pytest.raises(api.RtrError):
        api.render_batch(scene, [s.camera] * 17, [s.scene_info(b) for b in range(17)], p, frames)
    for f in frames:
        f.close()


def test_d6_occluded_sample_with_overflowing_contribution(gpu_ctx, oracle, scene_cache):
    """Divergence D6.  The reference evaluates the BRDF of every light sample and multiplies by currShadow
    (raygen.rgen:244-270); the oracle does the same.  The product, when only the shadowed image is kept, does not evaluate the BRDF
    of an occluded sample.  Same bits whenever the contribution is finite (every other parity test); with a light whose
    colour x intensity overflows fp32 an OCCLUDED sample gives 0 * inf = NaN in the reference and poisons its pixel (tone-mapped
    to 0), while the product's pixel keeps its other, finite terms.  This test pins exactly that: every pixel that differs is
    black in the oracle and lit by the directional light in the product, such pixels exist, and with the unshadowed image
    requested too (the product then evaluates every sample) there is no difference at all."""
    import ctypes as C
    W, H = 256, 256
    s = scenes.cornell_box(W, H)
    lights = s.host.lightInfos()
    hot = A.RtrAreaLightInfo.from_buffer_copy(bytes(lights[0]))
    hot.color[0], hot.color[1], hot.color[2], hot.intensity = 3.0, 3.0, 3.0, 3.0e38        # colour * intensity = inf
    arr = (A.RtrAreaLightInfo * 1)(hot)
    d = A.rtr_scene_desc.from_buffer_copy(bytes(s.desc))
    d.lights = C.cast(arr, C.POINTER(A.RtrAreaLightInfo))
    scene = api.Scene(gpu_ctx, d)
    bvh = scene.export_bvh()
    p = api.make_params(W, H, spp=1)
    frame = api.Frame(gpu_ctx, W, H)
    api.render(scene, s.camera, s.scene_info(0), p, frame)
    got = frame.download()
    ref = oracle.render(d, s.camera, s.scene_info(0), p, bvh=bvh, threads=8).images[A.IMAGE_SHADOWED]
    diff = got != ref
    assert 0 < diff.sum() < W * H // 4, int(diff.sum())
    assert np.all(ref[diff] == 0xff000000), "a differing pixel must be one the reference poisons (NaN -> 0)"
    assert np.all(got[diff] != 0xff000000), "... and one the product still lights"
    both = A.IMAGES_FRAMEBUFFER | A.IMG_BIT(A.IMAGE_UNSHADOWED)
    p2 = api.make_params(W, H, spp=1, images=both)
    f2 = api.Frame(gpu_ctx, W, H, both)
    api.render(scene, s.camera, s.scene_info(0), p2, f2)
    ref2 = oracle.render(d, s.camera, s.scene_info(0), p2, bvh=bvh, images=both, threads=8)
    _assert_same(f2.download(A.IMAGE_SHADOWED), ref2.images[A.IMAGE_SHADOWED], "D6 scene, shadowed image with every sample evaluated")
    _assert_same(f2.download(A.IMAGE_UNSHADOWED), ref2.images[A.IMAGE_UNSHADOWED], "D6 scene, unshadowed image")


@pytest.mark.parametrize("fill", ["0", "1", "auto"])
def test_visibility_prefill_polarity_does_not_change_a_pixel(gpu_ctx, oracle, scene_cache, queue_mode, fill):
    """The visibility array is pre-filled with the commoner outcome of the frame object's last launch and the any-hit kernel stores
    only the other one (WRITE_SIZE 136 -> 34 MB on the bench frame): either pre-fill, and the automatic choice over three frames
    whose occlusion differs, must give the oracle's image."""
    if fill != "auto":
        os.environ["RTR_TRACE_VIS_FILL"] = fill
    try:
        # forcing the pre-fill is a switch of librtr_hip_test.so (the product's sources built with -DRTR_TEST_HOOKS); "auto" is the product
        ctx = api.Context(0, test_hooks=True) if fill != "auto" else gpu_ctx
        s = scenes.bunny_class(320, 184)
        scene = api.Scene(ctx, s.desc)
        frame = api.Frame(ctx, 320, 184)
        bvh = scene.export_bvh()
        for f in (0, 1, 2):
            for collect in (0, 1):
                p = api.make_params(320, 184, spp=1, collect_stats=collect, pipeline=2)
                api.render(scene, s.camera, s.scene_info(f), p, frame)
                ref = oracle.render(s.desc, s.camera, s.scene_info(f), p, bvh=bvh, threads=16)
                _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], f"pre-fill {fill}, frame {f}, collect {collect}")
    finally:
        os.environ.pop("RTR_TRACE_VIS_FILL", None)


@pytest.mark.parametrize("n", [2, 3, 4, 16])
def test_batched_frames_equal_single_renders(gpu_ctx, oracle, scene_cache, queue_mode, n):
    """rtr_render_batch_async: n frames (own camera, own seed, own images) in ONE launch of every kernel give the pixels n launches
    give — and the oracle's — sharded and unsharded; the counting form's counters are the sum of the single frames'; a frame that
    was a follower of a batch renders alone again afterwards."""
    from realtimeraytracer_amd import host
    W, H = 200, 120
    s = scenes.bunny_class(W, H)
    scene = api.Scene(gpu_ctx, s.desc)
    bvh = scene.export_bvh()
    cams = []
    for b in range(n):
        pos = (s.cam_pos[0] + 0.3 * b, s.cam_pos[1] + 0.1 * b, s.cam_pos[2])
        cams.append(host.Camera(55.0 + b, pos, (0.0, 0.6, 0.0), (0.0, 1.0, 0.0), W, H).getGPUData())
    infos = [s.scene_info(7 + 3 * b) for b in range(n)]
    for shard in ((0, 1), (1, 3)):
        rows = api.shard_rows(H, 8, shard[1])
        for collect in (0, 1):
            p = api.make_params(W, H, spp=2, collect_stats=collect, pipeline=2, shard_index=shard[0], shard_count=shard[1])
            singles, stats = [], []
            f1 = api.Frame(gpu_ctx, W, rows)
            for b in range(n):
                api.render(scene, cams[b], infos[b], p, f1)
                singles.append(f1.download()); stats.append(f1.stats())
            frames = [api.Frame(gpu_ctx, W, rows) for _ in range(n)]
            api.render_batch(scene, cams, infos, p, frames)
            frames[-1].wait()                                   # any frame of the batch joins the launch
            for b in range(n):
                _assert_same(frames[b].download(), singles[b], f"batch of {n}, frame {b}, shard {shard}, collect {collect}")
            frames[0].wait()
            if collect:
                g = frames[0].stats()
                for fld in ("numRays", "numPrimaryRays", "numShadowRays", "numNodeVisits", "numTriTests", "numShadowNodeVisits", "numShadowTriTests", "numHits"):
                    assert getattr(g, fld) == sum(getattr(t, fld) for t in stats), fld
            if shard == (0, 1) and collect == 0:
                ref = oracle.render(s.desc, cams[n - 1], infos[n - 1], p, bvh=bvh, threads=16)
                _assert_same(frames[n - 1].download(), ref.images[A.IMAGE_SHADOWED], f"batch of {n}, last frame vs oracle")
                api.render(scene, cams[0], infos[0], p, frames[n - 1])        # a follower leads its own launch afterwards
                _assert_same(frames[n - 1].download(), singles[0], "former follower rendered alone")
    # refused: too many frames, the same frame twice, the megakernel
    many = [api.Frame(gpu_ctx, W, H) for _ in range(A.MAX_BATCH + 1)]
    p = api.make_params(W, H, spp=1)
    with pytest.raises(api.RtrError):
        api.render_batch(scene, [cams[0]] * len(many), [infos[0]] * len(many), p, many)
    with pytest.raises(api.RtrError):
        api.render_batch(scene, [cams[0]] * 2, [infos[0]] * 2, p, [many[0], many[0]])
    with pytest.raises(api.RtrError):
        api.render_batch(scene, [cams[0]] * 2, [infos[0]] * 2, api.make_params(W, H, spp=1, pipeline=1), many[:2])


@pytest.mark.gpu
def test_batched_launch_ragged_extent_and_hdr_accumulation(gpu_ctx, oracle, scene_cache):
    """A full launch (RTR_MAX_BATCH frames) of a frame whose extent is no multiple of the 8x8 tile or the workgroup, at 3 spp, every
    frame accumulating into its OWN HDR image over two launches: bit-identical (HDR floats and tonemapped bytes) to the same frames
    rendered one per launch, and to the oracle for the first and the last frame of the batch."""
    W, H, N = 37, 19, A.MAX_BATCH
    s = scenes.cornell_box(W, H)
    imgs = A.IMAGES_FRAMEBUFFER | A.IMG_BIT(A.IMAGE_HDR)
    scene = api.Scene(gpu_ctx, s.desc)
    bvh = scene.export_bvh()
    batch = [api.Frame(gpu_ctx, W, H, imgs) for _ in range(N)]
    single = api.Frame(gpu_ctx, W, H, imgs)
    info = lambda b, f: s.scene_info(100 * f + 7 * b)
    for f in range(2):                                             # two accumulation steps
        p = api.make_params(W, H, spp=3, images=imgs, accumulate=1, accumulated_frames=f, pipeline=2)
        api.render_batch(scene, [s.camera] * N, [info(b, f) for b in range(N)], p, batch)
    batch[3].wait()
    for b in (0, 5, N - 1):
        single.clear()
        hdr = np.zeros((H, W, 4), np.float32)
        ref = None
        for f in range(2):
            p = api.make_params(W, H, spp=3, images=imgs, accumulate=1, accumulated_frames=f, pipeline=2)
            api.render(scene, s.camera, info(b, f), p, single)
            if b != 5:
                ref = oracle.render(s.desc, s.camera, info(b, f), p, bvh=bvh, images=imgs, hdr=hdr, threads=8)
        g = batch[b].download(A.IMAGE_HDR)
        assert np.all(g[..., 3] == 2)
        assert np.array_equal(g.view(np.uint32), single.download(A.IMAGE_HDR).view(np.uint32)), f"frame {b}: HDR of the batch vs one per launch"
        _assert_same(batch[b].download(), single.download(), f"frame {b}: batch vs one per launch")
        if ref is not None:
            assert np.array_equal(g.view(np.uint32), hdr.view(np.uint32)), f"frame {b}: accumulated HDR vs oracle"
            _assert_same(batch[b].download(), ref.images[A.IMAGE_SHADOWED], f"frame {b}: tonemapped accumulation vs oracle")
    for o in batch + [single, scene]:
        o.close()


@pytest.mark.gpu
def test_batch_orders_itself_against_the_frames_own_streams(gpu_ctx, scene_cache):
    """A launch of several frames runs on the LEADING frame's stream; the other frames live on their own contexts / streams.  Work
    enqueued for such a frame before the batch (a render of its own) and after it (another render of its own) is ordered around the
    launch without a host join in between: every download shows the last thing enqueued for that frame."""
    import torch
    from realtimeraytracer_amd import host
    W, H = 480, 270
    s = scenes.sponza_class(W, H)
    scene = api.Scene(gpu_ctx, s.desc)
    p = api.make_params(W, H, spp=1, pipeline=2)
    cams = [host.Camera(60.0 + 2 * b, (s.cam_pos[0], s.cam_pos[1] + 20.0 * b, s.cam_pos[2]), (300.0, 380.0, -20.0), (0.0, 1.0, 0.0), W, H).getGPUData() for b in range(4)]
    infos = [s.scene_info(11 + b) for b in range(4)]
    ref = api.Frame(gpu_ctx, W, H)
    singles = []
    for b in range(4):
        api.render(scene, cams[b], infos[b], p, ref)
        singles.append(ref.download())
    ctxs, streams, frames = [], [], []
    for _ in range(3):
        c = api.Context(0); st = torch.cuda.Stream(); c.set_stream(st.cuda_stream)
        ctxs.append(c); streams.append(st); frames.append(api.Frame(c, W, H))
    for rnd in range(3):
        # frame 1 renders view 3 on its own stream, the batch (led by frame 0) then overwrites it with view 1 ...
        api.render(scene, cams[3], infos[3], p, frames[1], asynchronous=True)
        api.render_batch(scene, cams[:3], infos[:3], p, frames)
        # ... and frame 2 is re-rendered alone (view 3) on its own stream straight after the batch wrote view 2 into it
        api.render(scene, cams[3], infos[3], p, frames[2], asynchronous=True)
        for b in range(3):
            frames[b].wait()
        _assert_same(frames[0].download(), singles[0], f"round {rnd}: leading frame")
        _assert_same(frames[1].download(), singles[1], f"round {rnd}: the batch comes after the frame's own earlier render")
        _assert_same(frames[2].download(), singles[3], f"round {rnd}: the frame's own later render comes after the batch")
    for o in frames + ctxs + [ref, scene]:
        o.close()


@pytest.mark.gpu
def test_batch_then_denoise_and_clear_on_a_following_frame(gpu_ctx, oracle, scene_cache):
    """rtr_render_batch_async's ordering contract covers everything enqueued for a frame, not only renders: rtr_denoise_combine and
    rtr_frame_clear on a frame that FOLLOWED in a batch (the launch ran on the leader's stream) come behind that launch without a host
    join, and a later batch comes behind them."""
    import torch
    W, H = 480, 270
    s = scenes.sponza_class(W, H, ltc=scenes.shipped_ltc())
    scene = api.Scene(gpu_ctx, s.desc)
    p = api.make_params(W, H, spp=1, images=A.IMAGES_RAYGEN5, pipeline=2)
    infos = [s.scene_info(40 + b) for b in range(3)]
    one = api.Frame(gpu_ctx, W, H, 0xff)
    want = []
    for b in range(3):
        api.render(scene, s.camera, infos[b], p, one)
        one.denoise_combine(4)
        want.append(one.download(A.IMAGE_FINAL))
    ctxs, streams, frames = [], [], []
    for _ in range(3):
        c = api.Context(0); st = torch.cuda.Stream(); c.set_stream(st.cuda_stream)
        ctxs.append(c); streams.append(st); frames.append(api.Frame(c, W, H, 0xff))
    for rnd in range(3):
        api.render_batch(scene, [s.camera] * 3, infos, p, frames)
        frames[2].denoise_combine(4)             # on frame 2's own stream, straight behind the launch on frame 0's
        _assert_same(frames[2].download(A.IMAGE_FINAL), want[2], f"round {rnd}: denoise + combine of a following frame sees the batch's images")
        frames[1].clear()                        # likewise
        assert not frames[1].download(A.IMAGE_SHADOWED).any(), f"round {rnd}: the clear comes after the batch"
        frames[0].wait()
    for o in frames + ctxs + [one, scene]:
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("parts", [2, 3, 5, 16])
def test_split_render_equals_the_unsplit_frame_and_the_oracle(gpu_ctx, oracle, scene_cache, queue_mode, parts):
    """rtr_render_split: ONE frame as `parts` band-shards on streams of their own, each writing its rows of the frame's images in
    place — the same bytes as rtr_render (RGBA8 and the accumulated HDR floats) and as the oracle, the counters of the counting form
    equal to the unsplit frame's, for interleaved 8-row bands and for taller ones; a plain render of the same frame object afterwards
    and a split render with more parts than bands behave."""
    W, H = 333, 201                                  # no multiple of a tile, of a band or of the parts
    s = scenes.bunny_class(W, H)
    imgs = A.IMAGES_FRAMEBUFFER | A.IMG_BIT(A.IMAGE_HDR)
    scene = api.Scene(gpu_ctx, s.desc)
    bvh = scene.export_bvh()
    whole, split = api.Frame(gpu_ctx, W, H, imgs), api.Frame(gpu_ctx, W, H, imgs)
    for band in (8, 40):
        whole.clear(); split.clear()
        hdr = np.zeros((H, W, 4), np.float32)
        for f in range(2):                           # two accumulation steps: the HDR image is read and written in place by every part
            for collect in (0, 1):
                if collect and f == 0:
                    continue
                p = api.make_params(W, H, spp=2, images=imgs, accumulate=1 if not collect else 0, accumulated_frames=f if not collect else 0, band_rows=band,
                                    collect_stats=collect, pipeline=2)
                if not collect:
                    api.render(scene, s.camera, s.scene_info(5 + f), p, whole)
                    api.render_split(scene, s.camera, s.scene_info(5 + f), p, split, parts)
                    ref = oracle.render(s.desc, s.camera, s.scene_info(5 + f), p, bvh=bvh, images=imgs, hdr=hdr, threads=16)
                    _assert_same(split.download(), whole.download(), f"{parts} parts, bands of {band}, step {f}: split vs unsplit")
                    _assert_same(split.download(), ref.images[A.IMAGE_SHADOWED], f"{parts} parts, bands of {band}, step {f}: split vs oracle")
                    assert np.array_equal(split.download(A.IMAGE_HDR).view(np.uint32), hdr.view(np.uint32)), "accumulated HDR of the split frame vs oracle"
                else:
                    cw, cs = api.Frame(gpu_ctx, W, H), api.Frame(gpu_ctx, W, H)
                    pc = api.make_params(W, H, spp=2, band_rows=band, collect_stats=1, pipeline=2)
                    api.render(scene, s.camera, s.scene_info(9), pc, cw)
                    api.render_split(scene, s.camera, s.scene_info(9), pc, cs, parts)
                    _assert_same(cs.download(), cw.download(), "counting forms, split vs unsplit")
                    a, b = cw.stats(), cs.stats()
                    for fld in ("numRays", "numPrimaryRays", "numShadowRays", "numNodeVisits", "numTriTests", "numShadowNodeVisits", "numShadowTriTests", "numHits"):
                        assert getattr(a, fld) == getattr(b, fld), fld
                    assert b.localPixels >= a.localPixels      # the parts are equal-size shards: padding rows included
                    assert b.totalMs > 0 and b.pipelineUsed == 2
                    cw.close(); cs.close()
    # the frame object renders unsplit again, and asynchronously split with the join left to rtr_frame_wait
    p = api.make_params(W, H, spp=1, pipeline=2)
    api.render(scene, s.camera, s.scene_info(1), p, whole)
    api.render(scene, s.camera, s.scene_info(1), p, split)
    _assert_same(split.download(), whole.download(), "a plain render after split renders")
    api.render_split(scene, s.camera, s.scene_info(2), p, split, parts, asynchronous=True)
    api.render(scene, s.camera, s.scene_info(2), p, whole)
    _assert_same(split.download(), whole.download(), "asynchronous split render, joined by the download")
    # refused: a shard of a sharded frame, too many parts, a frame that is not the whole frame
    with pytest.raises(api.RtrError):
        api.render_split(scene, s.camera, s.scene_info(0), api.make_params(W, H, shard_index=0, shard_count=2), split, parts)
    with pytest.raises(api.RtrError):
        api.render_split(scene, s.camera, s.scene_info(0), p, split, A.MAX_SPLIT + 1)
    small = api.Frame(gpu_ctx, W, api.shard_rows(H, 8, 2))
    with pytest.raises(api.RtrError):
        api.render_split(scene, s.camera, s.scene_info(0), p, small, 2)
    for o in (whole, split, small, scene):
        o.close()


@pytest.mark.gpu
def test_split_render_1080p_against_the_unsplit_frame(gpu_ctx, scene_cache):
    """BASELINE config 4 at full size through the latency path: the Sponza-class 1080p frame as 2 and 4 parts equals the unsplit
    frame byte for byte (which test_sponza_1080p_properties_and_sampled_oracle holds against the oracle)."""
    W, H = 1920, 1080
    s = scenes.sponza_class(W, H)
    scene = api.Scene(gpu_ctx, s.desc)
    p = api.make_params(W, H, spp=1)
    whole, split = api.Frame(gpu_ctx, W, H), api.Frame(gpu_ctx, W, H)
    api.render(scene, s.camera, s.scene_info(3), p, whole)
    for parts in (2, 4):
        split.clear()
        api.render_split(scene, s.camera, s.scene_info(3), p, split, parts)
        _assert_same(split.download(), whole.download(), f"1080p, {parts} parts")
    for o in (whole, split, scene):
        o.close()


@pytest.mark.parametrize("own_leaf", [1, 0])
@pytest.mark.parametrize("name", ["sponza_class", "sponza_mixed", "textured_room", "cornell_box"])
def test_own_leaf_start_of_the_shadow_walk_matches_the_oracle(gpu_ctx, oracle, scene_cache, name, own_leaf):
    """A shadow ray that leaves its surface point INTO the surface starts its walk at the leaf of the triangle it comes from
    (k_shadow_trace4's refill; tunable trace_own_leaf, default on): framebuffer bytes AND the any-hit work counters — record visits,
    triangle tests, trips and lanes of both loops — equal the oracle's with the rule on and with it off, over two frames (other seeds)."""
    W, H = 320, 184
    s = getattr(scenes, name)(W, H)
    scene = api.Scene(gpu_ctx, s.desc)
    bvh = scene.export_bvh()
    c = api.Context(0)
    c.set_tunable("trace_own_leaf", own_leaf)
    frame = api.Frame(c, W, H)
    try:
        for f in (0, 5):
            p = api.make_params(W, H, spp=2, shadow_rays=3, collect_stats=1, pipeline=2)
            api.render(scene, s.camera, s.scene_info(f), p, frame)
            ref = oracle.render(s.desc, s.camera, s.scene_info(f), p, bvh=bvh, threads=8, own_leaf=bool(own_leaf))
            _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], f"{name} own_leaf={own_leaf} frame {f}")
            g, o = frame.stats(), ref.stats
            for fld in ("numRays", "numShadowRays", "numShadowNodeVisits", "numShadowTriTests", "numNodeVisits", "numTriTests", "numAlphaTests", "numTexFetches", "shadowTailRays"):
                assert getattr(g, fld) == getattr(o, fld), f"{name} own_leaf={own_leaf}: counter {fld}: gpu {getattr(g, fld)} != oracle {getattr(o, fld)}"
            assert (ref.walk.ownLeafRays > 0) == bool(own_leaf)
            # the timed form writes the same bytes
            api.render(scene, s.camera, s.scene_info(f), api.make_params(W, H, spp=2, shadow_rays=3, pipeline=2), frame)
            _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], f"{name} own_leaf={own_leaf} frame {f}, timed form")
    finally:
        frame.close(); c.close(); scene.close()


@pytest.mark.parametrize("case", ["cornell_s1", "cornell_s3_spp3", "cornell_s20", "cornell_s40", "sponza_class", "sponza_mixed", "textured_room"])
def test_resolve_compact_and_per_pixel_forms_match_the_oracle(gpu_ctx, oracle, scene_cache, case):
    """k_resolve_compact (the framebuffer-only launch's default: the BRDF of a tile's VISIBLE samples dealt out densely over the wave's
    lanes, sums in the reference's order) and k_resolve (one lane per pixel walks its samples) against the oracle, framebuffer bytes and
    HDR bits, over two accumulated frames: 1 / 3 shadow rays per light triangle, several samples per pixel, light lists longer than a
    round's 32 steps (20 rays x 2 triangles + 1) and than its item list (40 rays: 81 steps), textures / alpha / HDRI sky."""
    W, H = 200, 120
    setup, spp, ns = {"cornell_s1": (scenes.cornell_box, 1, 1), "cornell_s3_spp3": (scenes.cornell_box, 3, 3), "cornell_s20": (scenes.cornell_box, 1, 20),
                      "cornell_s40": (scenes.cornell_box, 2, 40), "sponza_class": (scenes.sponza_class, 1, 3), "sponza_mixed": (scenes.sponza_mixed, 1, 3),
                      "textured_room": (scenes.textured_room, 2, 3)}[case]
    s = setup(W, H)
    imgs = A.IMAGES_FRAMEBUFFER | A.IMG_BIT(A.IMAGE_HDR)
    scene = api.Scene(gpu_ctx, s.desc)
    bvh = scene.export_bvh()
    c = api.Context(0)
    try:
        for compact in (1, 0):
            c.set_tunable("resolve_compact", compact)
            frame = api.Frame(c, W, H, imgs)
            hdr = np.zeros((H, W, 4), np.float32)
            ref = None
            for f in range(2):
                p = api.make_params(W, H, spp=spp, shadow_rays=ns, images=imgs, accumulate=1, accumulated_frames=f, pipeline=2)
                api.render(scene, s.camera, s.scene_info(f), p, frame)
                ref = oracle.render(s.desc, s.camera, s.scene_info(f), p, bvh=bvh, images=imgs, hdr=hdr, threads=8)
            assert np.array_equal(frame.download(A.IMAGE_HDR).view(np.uint32), hdr.view(np.uint32)), f"{case}: HDR bits, resolve_compact={compact}"
            _assert_same(frame.download(), ref.images[A.IMAGE_SHADOWED], f"{case}: framebuffer, resolve_compact={compact}")
            frame.close()
    finally:
        c.close()
        scene.close()


@pytest.mark.gpu
def test_tunables_belong_to_the_context_and_change_no_pixel(gpu_ctx, scene_cache):
    """rtr_ctx_set_tunable / rtr_ctx_get_tunable: scheduling knobs of the staged pipeline are read from the environment when a
    context is created and set per context afterwards; none of them changes a pixel; unknown names and values out of range are refused."""
    W, H = 256, 144
    s = scenes.bunny_class(W, H)
    scene = api.Scene(gpu_ctx, s.desc)
    p = api.make_params(W, H, spp=1, pipeline=2)
    base = api.Frame(gpu_ctx, W, H)
    api.render(scene, s.camera, s.scene_info(0), p, base)
    want = base.download()
    os.environ["RTR_TRACE_REFILL"] = "33"
    try:
        c = api.Context(0)
    finally:
        del os.environ["RTR_TRACE_REFILL"]
    assert c.get_tunable("trace_refill") == 33 and gpu_ctx.get_tunable("trace_refill") == 20
    fr = api.Frame(c, W, H)
    for name, value in (("trace_binned", 1), ("trace_batch", 64), ("trace_wgs_per_cu", 3), ("trace_inner_min", 5), ("trace_octant_forms", 0),
                        ("trace_top_nodes", 7), ("trace_own_leaf", 0), ("queue_nt", 3), ("resolve_row_waves", 1), ("resolve_compact", 0), ("primary_packet", 1), ("primary_persist", 1), ("trace_bvh4", 0)):
        c.set_tunable(name, value)
        assert c.get_tunable(name) == value
        api.render(scene, s.camera, s.scene_info(0), p, fr)
        _assert_same(fr.download(), want, f"tunable {name} = {value}")
    with pytest.raises(api.RtrError):
        c.set_tunable("no_such_knob", 1)
    with pytest.raises(api.RtrError):
        c.set_tunable("trace_refill", 0)
    with pytest.raises(api.RtrError):
        api.render(scene, s.camera, s.scene_info(0), api.make_params(W, H, collect_stats=1, pipeline=2), fr)     # trace_bvh4 = 0 has no counting form
    for o in (fr, c, base, scene):
        o.close()


def _run_staged_child(code, env, timeout):
    """A child process whose script announces every library call before making it (STAGE lines, flushed): when the child hangs, the
    timeout says where it stopped instead of nothing (round 2's 240-s hang left only RCCL's banner on stdout)."""
    import subprocess, sys
    try:
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        out = (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, (bytes, type(None))) else e.stdout
        stages = [ln for ln in (out or "").splitlines() if ln.startswith("STAGE")]
        raise AssertionError(f"child timed out after {timeout} s; last stage announced: {stages[-1] if stages else 'none'}\n" + (out or "")[-1500:])
    return r


def test_mgpu_library_single_process_through_rccl(gpu_ctx, oracle, scene_cache):
    """include/rtr_mgpu.h: the sharded frame behind the C ABI, ONE process driving every GPU of the box (1 on the test box) through a
    real RCCL communicator.  With one rank RTR_MGPU_SELF_EXCHANGE=1 makes the shard travel through grouped ncclSend / ncclRecv
    all the same, so communicator, communication stream, events, gather buffer and k_deinterleave are exercised; two frame slots
    are kept in flight.  Assembled frame == unsharded frame == oracle, 0 pixels.  The library's watchdog is set below the test's own
    timeout, so a stuck exchange comes back as an error that names the stage."""
    import textwrap
    # its own process: RCCL initialises its own state, and a failure (or a hang, hence the timeout) must not take the suite down
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, os.getcwd())
        def stage(s): print("STAGE", s, flush=True)
        stage("imports")
        import numpy as np, torch
        from realtimeraytracer_amd import _abi as A, api, mgpu, scenes
        from oracle import oracle_py as O
        n = torch.cuda.device_count()
        W, H = 640, 360
        s = scenes.cornell_box(W, H)
        stage("rtr_mgpu_create")
        m = mgpu.MultiGpu(devices=list(range(n)), frames_in_flight=2)
        assert m.info.nranks == n and m.info.nlocal == n and m.info.selfExchange == (1 if os.environ.get("RTR_MGPU_SELF_EXCHANGE") == "1" else 0)
        assert m.info.aborted == 0 and m.info.timeoutMs == 60000 and m.info.rcclVersion > 20000
        stage("rtr_mgpu_scene_create")
        m.scene_create(s.desc)
        p = api.make_params(W, H, spp=2)
        stage("rtr_mgpu_render_async slot 0")
        m.render_async(0, s.camera, s.scene_info(0), p)
        stage("rtr_mgpu_render_async slot 1")
        m.render_async(1, s.camera, s.scene_info(1), p)          # second slot in flight behind the first
        stage("rtr_mgpu_wait slot 0")
        m.wait(0)
        stage("rtr_mgpu_wait slot 1")
        m.wait(1)
        stage("rtr_mgpu_frame_download")
        got0, got1 = m.download(0), m.download(1)
        stage("rtr_mgpu_render_async slot 0 again")
        m.render_async(0, s.camera, s.scene_info(2), p)          # slot reuse: ordered behind its previous exchange by an event
        try:
            m.frame_device_ptr(0)
            raise SystemExit("rtr_mgpu_frame_device_ptr answered for a slot in flight")
        except RuntimeError as e:
            assert "in flight" in str(e)
        stage("rtr_mgpu_wait slot 0 again")
        m.wait(0)
        got2 = m.download(0)
        stage("another extent through the same slot")
        p2 = api.make_params(320, 184, spp=1)
        m.render_async(1, s.camera, s.scene_info(0), p2)         # the slot is re-created (phase 1 on every rank, joined) before anything is posted
        m.wait(1)
        small = m.download(1)
        stage("rtr_mgpu_render_batch_async slots 0, 1")
        m.render_batch_async([0, 1], [s.camera, s.camera], [s.scene_info(5), s.scene_info(6)], p)     # ONE launch of the pipeline per rank, each slot its own exchange
        stage("rtr_mgpu_wait slots 1, 0 of the batch")
        m.wait(1); m.wait(0)
        gb = {5: m.download(0), 6: m.download(1)}
        stage("reference renders")
        ctx = api.Context(0)
        scene = api.Scene(ctx, s.desc)
        frame = api.Frame(ctx, W, H)
        bvh = scene.export_bvh()
        for f, got in ((0, got0), (1, got1), (2, got2)):
            api.render(scene, s.camera, s.scene_info(f), p, frame)
            whole = frame.download()
            assert int((got != whole).sum()) == 0, ("assembled vs unsharded", f, int((got != whole).sum()))
            ref = O.render(s.desc, s.camera, s.scene_info(f), p, bvh=bvh, threads=8)
            assert int((got != ref.images[A.IMAGE_SHADOWED]).sum()) == 0, ("assembled vs oracle", f)
        for f, got in gb.items():
            api.render(scene, s.camera, s.scene_info(f), p, frame)
            assert int((got != frame.download()).sum()) == 0, ("batched slots, assembled vs unsharded", f)
        frame2 = api.Frame(ctx, 320, 184)
        api.render(scene, s.camera, s.scene_info(0), p2, frame2)
        assert int((small != frame2.download()).sum()) == 0, "assembled vs unsharded, second extent"
        shard = m.download_shard(0, 0)
        assert shard.shape == (api.shard_rows(H, 8, n), W)
        stage("rtr_mgpu_destroy")
        m.close()
        print("MGPU_OK", n)
    """)
    env = dict(os.environ, RTR_MGPU_SELF_EXCHANGE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", RTR_MGPU_TIMEOUT_MS="60000")
    r = _run_staged_child(code, env, 240)
    assert r.returncode == 0 and "MGPU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_mgpu_library_full_batches_overlapping_slot_groups():
    """librtr_mgpu.so with 32 frame slots: two full launches (RTR_MAX_BATCH slots each) in flight, then a third whose slots are taken
    from BOTH — another slot leads it on another stream, some of its frames were written last by the first launch's stream, others
    led or followed the second — without a host join in between; every assembled frame equals the frame rendered unsharded.  (What
    orders the launches is the events of rtr_render_batch_async and the slots' exchange events, not the caller.)"""
    import textwrap
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, os.getcwd())
        def stage(s): print("STAGE", s, flush=True)
        stage("imports")
        import numpy as np, torch
        from realtimeraytracer_amd import _abi as A, api, mgpu, scenes
        n = torch.cuda.device_count()
        B = A.MAX_BATCH
        W, H = 320, 184
        s = scenes.cornell_box(W, H)
        stage("rtr_mgpu_create")
        m = mgpu.MultiGpu(devices=list(range(n)), frames_in_flight=2 * B)
        stage("rtr_mgpu_scene_create")
        m.scene_create(s.desc)
        p = api.make_params(W, H, spp=1)
        want = {}
        def launch(slots, first):
            m.render_batch_async(slots, [s.camera] * len(slots), [s.scene_info(first + j) for j in range(len(slots))], p)
            for j, sl in enumerate(slots):
                want[sl] = first + j
        for rnd in range(2):
            stage(f"round {rnd}: launch over slots 0..{B - 1}")
            launch(list(range(B)), 1000 * rnd)
            stage(f"round {rnd}: launch over slots {B}..{2 * B - 1}")
            launch(list(range(B, 2 * B)), 1000 * rnd + 100)
            stage(f"round {rnd}: launch over slots {B // 2}..{B // 2 + B - 1} (both groups)")
            launch(list(range(B // 2, B // 2 + B)), 1000 * rnd + 200)
        stage("rtr_mgpu_wait, every slot")
        for sl in range(2 * B):
            m.wait(sl)
        got = {sl: m.download(sl) for sl in range(2 * B)}
        stage("reference renders")
        ctx = api.Context(0)
        scene = api.Scene(ctx, s.desc)
        frame = api.Frame(ctx, W, H)
        for sl, f in sorted(want.items()):
            api.render(scene, s.camera, s.scene_info(f), p, frame)
            bad = int((got[sl] != frame.download()).sum())
            assert bad == 0, ("slot", sl, "frame", f, "pixels differing", bad)
        stage("rtr_mgpu_destroy")
        m.close()
        print("MGPU_BATCHES_OK", n, len(want))
    """)
    env = dict(os.environ, RTR_MGPU_SELF_EXCHANGE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", RTR_MGPU_TIMEOUT_MS="60000")
    r = _run_staged_child(code, env, 240)
    assert r.returncode == 0 and "MGPU_BATCHES_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_mgpu_library_several_ranks_on_one_gpu_through_a_same_device_rccl_double(nranks):
    """librtr_mgpu.so's N > 1 code — enqueue() carrying out every rank's plan on its own streams from its own host thread: shards
    rendered into gather / local buffers, rank 0 receiving N - 1 shards at shardBytes * src while the others send, the de-interleave,
    slot reuse, full launches of RTR_MAX_BATCH slots, launches over slots of two earlier groups — with 2, 3 and 8 ranks on a box
    that has ONE GPU: the ranks share the device (the library's test hook RTR_MGPU_TEST_SHARED_DEVICE) and tests/fake_rccl/, preloaded
    in front of librccl.so, turns a send / receive pair into an event-ordered device-to-device copy.  Every assembled frame equals the
    same frame rendered unsharded.  What this cannot show is RCCL itself and the speed; what it does show is that the product's own
    ordering, offsets and host threads are right with more than one rank in them."""
    import textwrap
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        pytest.fail("tests/fake_rccl/libfake_rccl.so is missing: run __graft_entry__.build()")
    code = textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, os.getcwd())
        def stage(s): print("STAGE", s, flush=True)
        stage("imports")
        import numpy as np
        from realtimeraytracer_amd import _abi as A, api, mgpu, scenes
        n, B = {nranks}, A.MAX_BATCH
        W, H = 320, 184
        s = scenes.cornell_box(W, H)
        stage("rtr_mgpu_create")
        m = mgpu.MultiGpu(devices=[0] * n, frames_in_flight=2 * B)
        assert m.info.nranks == n and m.info.nlocal == n and m.info.rcclVersion == 99999, (m.info.nranks, m.info.rcclVersion)
        stage("rtr_mgpu_scene_create")
        m.scene_create(s.desc)
        p = api.make_params(W, H, spp=1)
        want = {{}}
        stage("single frames, two slots in flight, a slot reused")
        m.render_async(0, s.camera, s.scene_info(1), p); m.render_async(1, s.camera, s.scene_info(2), p)
        m.render_async(0, s.camera, s.scene_info(3), p)
        want[0], want[1] = 3, 2
        def launch(slots, first):
            m.render_batch_async(slots, [s.camera] * len(slots), [s.scene_info(first + j) for j in range(len(slots))], p)
            for j, sl in enumerate(slots):
                want[sl] = first + j
        stage("a launch over slots 2..5")
        launch([2, 3, 4, 5], 10)
        stage("rtr_mgpu_wait")
        for sl in range(6):
            m.wait(sl)
        got = {{sl: m.download(sl) for sl in range(6)}}
        shard = m.download_shard(0, n - 1)
        assert shard.shape == (api.shard_rows(H, 8, n), W)
        for rnd in range(2):
            stage(f"round {{rnd}}: two full launches and a third over slots of both")
            launch(list(range(B)), 1000 * rnd + 100)
            launch(list(range(B, 2 * B)), 1000 * rnd + 200)
            launch(list(range(B // 2, B // 2 + B)), 1000 * rnd + 300)
        for sl in range(2 * B):
            m.wait(sl)
        stage("reference renders")
        ctx = api.Context(0)
        scene = api.Scene(ctx, s.desc)
        frame = api.Frame(ctx, W, H)
        def same(img, f, what):
            api.render(scene, s.camera, s.scene_info(f), p, frame)
            bad = int((img != frame.download()).sum())
            assert bad == 0, (what, "frame", f, "pixels differing", bad)
        for sl, f in ((0, 3), (1, 2), (2, 10), (3, 11), (4, 12), (5, 13)):
            same(got[sl], f, ("first part, slot", sl))
        for sl, f in sorted(want.items()):
            same(m.download(sl), f, ("slot", sl))
        stage("another extent")
        p2 = api.make_params(200, 120, spp=2)
        m.render_async(1, s.camera, s.scene_info(7), p2); m.wait(1)
        small = m.download(1)
        f2 = api.Frame(ctx, 200, 120)
        api.render(scene, s.camera, s.scene_info(7), p2, f2)
        assert int((small != f2.download()).sum()) == 0
        stage("rtr_mgpu_destroy")
        m.close()
        print("MGPU_RANKS_OK", n, len(want))
    """)
    env = dict(os.environ, LD_PRELOAD=fake, RTR_MGPU_TEST_SHARED_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", RTR_MGPU_TIMEOUT_MS="60000")
    env.pop("RTR_MGPU_SELF_EXCHANGE", None)
    r = _run_staged_child(code, env, 300)
    assert r.returncode == 0 and "MGPU_RANKS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_scene_create_like_uploads_the_tree_without_building_it(gpu_ctx, scene_cache):
    """rtr_scene_create_like: the replica renders the same bytes, reports the same tree and did not spend a build (what
    rtr_mgpu_scene_create replicates the scene with); a description of another size is refused."""
    s = scenes.bunny_class(320, 184)
    a = api.Scene(gpu_ctx, s.desc)
    ctx2 = api.Context(0)
    b = api.Scene(ctx2, s.desc, like=a)
    sa, sb = a.stats(), b.stats()
    assert (sa.numTriangles, sa.numNodes, sa.maxDepth, sa.stackEntries, sa.numWideNodes) == (sb.numTriangles, sb.numNodes, sb.maxDepth, sb.stackEntries, sb.numWideNodes)
    assert sb.buildMs == 0.0 and sa.buildMs > 0.0
    assert bytes(a.export_bvh()[0]) == bytes(b.export_bvh()[0])
    p = api.make_params(320, 184, spp=1, collect_stats=1)
    fa, fb = api.Frame(gpu_ctx, 320, 184), api.Frame(ctx2, 320, 184)
    api.render(a, s.camera, s.scene_info(3), p, fa)
    api.render(b, s.camera, s.scene_info(3), p, fb)
    _assert_same(fa.download(), fb.download(), "scene made by rtr_scene_create_like")
    assert fa.stats().numShadowNodeVisits == fb.stats().numShadowNodeVisits
    other = scenes.cornell_box(64, 64)
    with pytest.raises(api.RtrError):
        api.Scene(ctx2, other.desc, like=a)
