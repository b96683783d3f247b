"""Tile-sharded multi-GPU frames — new functionality with no reference counterpart
(the reference picks ONE physical device, src/vulkan/context/device.cppm:118-140).

One process per GPU (torch.distributed; backend "nccl" IS RCCL on ROCm).  The scene is replicated;
the frame is cut into bands of `band_rows` rows and band b belongs to rank b % world_size
(interleaved so non-uniform scene cost balances).  Every rank renders its bands into a compact
(local_rows x width) RGBA8 buffer; ONE exchange step — a gather to rank 0 over xGMI — then a
de-interleave kernel on rank 0 (rtr_deinterleave_bands).  Pixels are independent and the PCG seeds
depend only on (x, y, sample, frame), so the assembled frame is bit-identical to the 1-GPU frame.

The gather/assemble logic is backend-agnostic so it is covered by world_size-2 `gloo` tests on CPU
(tests/test_distributed.py) with the oracle standing in for the renderer.
"""
import numpy as np


def shard_rows(height, band_rows, shard_count):
    if shard_count <= 1:
        return height
    bands = (height + band_rows - 1) // band_rows
    per = (bands + shard_count - 1) // shard_count
    return per * band_rows


def global_rows_of_shard(height, band_rows, shard_count, shard_index):
    """global y of every local row of a shard (-1 for padding rows)."""
    rows = shard_rows(height, band_rows, shard_count)
    lr = np.arange(rows)
    y = ((lr // band_rows) * shard_count + shard_index) * band_rows + (lr % band_rows)
    return np.where(y < height, y, -1)


def assemble_numpy(gathered, height, band_rows):
    """CPU restatement of rtr_deinterleave_bands: gathered[shard, local_row, x] -> full[y, x]."""
    n, rows, width = gathered.shape[:3]
    full = np.zeros((height,) + gathered.shape[2:], dtype=gathered.dtype)
    for r in range(n):
        ys = global_rows_of_shard(height, band_rows, n, r)
        ok = ys >= 0
        full[ys[ok]] = gathered[r][ok]
    return full


def gather_to_root(dist, local, world_size, rank, root=0):
    """The one exchange step: every rank's (local_rows x width [x C]) tensor -> rank `root`.
    Returns the [world, ...] stacked tensor on root, None elsewhere.  Works for gloo (CPU tensors)
    and nccl/RCCL (device tensors)."""
    import torch
    if world_size == 1:
        return local.unsqueeze(0)
    if rank == root:
        out = torch.empty((world_size,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, gather_list=[out[i] for i in range(world_size)], dst=root)
        return out
    dist.gather(local, gather_list=None, dst=root)
    return None
