"""N>1 path on CPU: world_size-2 (and 3) `gloo` process groups run the band sharding + the one gather step +
the de-interleave, with the CPU oracle standing in for the renderer; the assembled frame must be bit-identical
to the unsharded frame."""
import os
import socket
import sys

import numpy as np
import pytest

from realtimeraytracer_amd import mgpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_band_mapping_covers_every_row_once():
    for h, band, n in ((1080, 8, 8), (1080, 8, 3), (100, 8, 2), (7, 8, 2), (2160, 16, 5)):
        rows = mgpu.shard_rows(h, band, n)
        seen = np.zeros(h, int)
        for r in range(n):
            ys = mgpu.global_rows_of_shard(h, band, n, r)
            assert len(ys) == rows
            ok = ys[ys >= 0]
            seen[ok] += 1
            # band b belongs to rank b % n
            assert np.all((ok // band) % n == r)
        assert np.all(seen == 1)


def test_assemble_numpy_roundtrip():
    h, w, band, n = 50, 7, 8, 3
    full = np.arange(h * w, dtype=np.uint32).reshape(h, w)
    rows = mgpu.shard_rows(h, band, n)
    g = np.zeros((n, rows, w), np.uint32)
    for r in range(n):
        ys = mgpu.global_rows_of_shard(h, band, n, r)
        g[r][ys >= 0] = full[ys[ys >= 0]]
    assert np.array_equal(mgpu.assemble_numpy(g, h, band), full)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cache, out_path):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RTR_SCENE_CACHE": cache})
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from realtimeraytracer_amd import _abi as A
    from realtimeraytracer_amd import api, scenes
    from oracle import oracle_py as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 96, 52                                   # ragged: 6.5 bands -> padding rows on some ranks
    s = scenes.cornell_box(W, H)
    st, nodes, tris = api.host_build_bvh(s.desc)
    p = api.make_params(W, H, spp=2, shard_index=rank, shard_count=world)
    r = O.render(s.desc, s.camera, s.scene_info(1), p, bvh=(nodes, tris, st.grid), threads=2)
    local = torch.from_numpy(r.images[A.IMAGE_SHADOWED].view(np.int32).copy())
    gathered = mgpu.gather_to_root(dist, local, world, rank)
    if rank == 0:
        full = mgpu.assemble_numpy(gathered.numpy().view(np.uint32), H, 8)
        p1 = api.make_params(W, H, spp=2)
        ref = O.render(s.desc, s.camera, s.scene_info(1), p1, bvh=(nodes, tris, st.grid), threads=2).images[A.IMAGE_SHADOWED]
        np.save(out_path, np.array([int((full != ref[:H]).sum()), full.shape[0], full.shape[1]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_frame_gloo(world, tmp_path, scene_cache):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.npy")
    mp.spawn(_worker, args=(world, _free_port(), scene_cache, out), nprocs=world, join=True)
    diff, h, w = np.load(out)
    assert (h, w) == (52, 96)
    assert diff == 0, f"{diff} pixels differ between the gathered sharded frame and the single frame"
