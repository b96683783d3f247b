"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same
inputs.  Bar: RGBA8 images bit-exact (integer framebuffer); HDR float accumulation bit-exact as well
(tolerance 0: both sides evaluate the same IEEE expression tree, include/rtr_math.h)."""
import numpy as np
import pytest

from realtimeraytracer_amd import _abi as A
from realtimeraytracer_amd import api, scenes

pytestmark = pytest.mark.gpu


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
