"""N>1 path on CPU: world_size-2 (and 3) `gloo` process groups carry out librtr_mgpu.so's OWN exchange plan (rtr_mgpu_plan — the
operation list its enqueue() executes on the GPU: which rank sends what to whom, at which byte offset, grouped, behind which event)
over gloo point-to-point calls, with the CPU oracle standing in for the renderer and a numpy restatement of k_deinterleave; the
assembled frame must be bit-identical to the unsharded frame.  Two frames through the same slot, so the
slot-reuse edge is walked too."""
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
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from realtimeraytracer_amd import _abi as A
    from realtimeraytracer_amd import api, scenes
    from oracle import oracle_py as O
    import plan_exec as PE
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 96, 52                                   # ragged: 6.5 bands -> padding rows on some ranks
    s = scenes.cornell_box(W, H)
    st, nodes, tris = api.host_build_bvh(s.desc)
    frame_no = [1]

    def render_shard(index, count):                 # what RTR_MGPU_OP_RENDER stands for, by the oracle
        p = api.make_params(W, H, spp=2, shard_index=index, shard_count=count)
        return O.render(s.desc, s.camera, s.scene_info(frame_no[0]), p, bvh=(nodes, tris, st.grid), threads=2).images[A.IMAGE_SHADOWED]
    runner = PE.PlanRunner(rank, world, W, H, 8, render_shard, dist)
    ops = PE.plan(rank, world, W, H, 8)             # this rank's operations, from the library
    diffs = []
    for f in (1, 2):                                # the second frame re-uses the slot: its first WAIT now has an event to wait for
        frame_no[0] = f
        runner.run(ops)
        if rank == 0:
            ref = O.render(s.desc, s.camera, s.scene_info(f), api.make_params(W, H, spp=2), bvh=(nodes, tris, st.grid), threads=2).images[A.IMAGE_SHADOWED]
            diffs.append(int((runner.full() != ref[:H]).sum()))
        dist.barrier()
    # ... and a LAUNCH of two frames (rtr_mgpu_plan_batch: one render of both shards, ONE group holding both slots' transfers, a
    # de-interleave per slot), twice through the same two slots
    def render_slot(index, count, slot):
        p = api.make_params(W, H, spp=2, shard_index=index, shard_count=count)
        return O.render(s.desc, s.camera, s.scene_info(frame_no[0] + slot), p, bvh=(nodes, tris, st.grid), threads=2).images[A.IMAGE_SHADOWED]
    brunner = PE.PlanRunner(rank, world, W, H, 8, render_slot, dist, nslots=2)
    bops = PE.plan_batch(rank, world, W, H, 2, 8)
    for f in (5, 9):
        frame_no[0] = f
        brunner.run(bops)
        if rank == 0:
            for slot in (0, 1):
                ref = O.render(s.desc, s.camera, s.scene_info(f + slot), api.make_params(W, H, spp=2), bvh=(nodes, tris, st.grid), threads=2).images[A.IMAGE_SHADOWED]
                diffs.append(int((brunner.full(slot) != ref[:H]).sum()))
        dist.barrier()
    if rank == 0:
        np.save(out_path, np.array(diffs + [H, W]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_frame_gloo(world, tmp_path, scene_cache):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.npy")
    mp.spawn(_worker, args=(world, _free_port(), scene_cache, out), nprocs=world, join=True)
    *d, h, w = np.load(out)
    assert (h, w) == (52, 96) and len(d) == 6
    assert not any(d), f"{d} pixels differ between the frames assembled by the library's exchange plans (one-frame plan x 2, two-frame launch x 2 x 2 slots) and the single frames"
