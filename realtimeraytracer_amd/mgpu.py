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


class MultiGpu:
    """ctypes view of librtr_mgpu.so (include/rtr_mgpu.h): the sharded frame behind the C ABI — one rtr_ctx + one scene per
    rank, the RCCL communicator, one host thread per local rank, a communication stream per rank, frame slots.
    MultiGpu(devices=[0, 1, ...]) drives several GPUs from this process; MultiGpu.rank(device, rank, nranks, unique_id) is one
    rank of a one-process-per-GPU job (bench.py under torch.distributed.run)."""

    def __init__(self, devices=None, frames_in_flight=1, _handle=None):
        import ctypes as C
        from . import _abi as A
        self.lib = A.mgpu_lib()
        self.h = A.VP()
        if _handle is not None:
            self.h = _handle
        else:
            arr = (C.c_int * len(devices))(*devices)
            self._check(self.lib.rtr_mgpu_create(arr, len(devices), frames_in_flight, C.byref(self.h)), "rtr_mgpu_create")
        self.info = A.rtr_mgpu_info()
        self._check(self.lib.rtr_mgpu_get_info(self.h, C.byref(self.info)), "rtr_mgpu_get_info")
        self._extent = {}            # per slot: (height, width, band rows) of the last render

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _abi as A
        lib = A.mgpu_lib()
        buf = (C.c_ubyte * A.MGPU_ID_BYTES)()
        rc = lib.rtr_mgpu_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"rtr_mgpu_unique_id failed ({rc}): {lib.rtr_mgpu_last_error().decode()}")
        return bytes(buf)

    @classmethod
    def rank(cls, device, rank, nranks, unique_id, frames_in_flight=1):
        import ctypes as C
        from . import _abi as A
        lib = A.mgpu_lib()
        h = A.VP()
        buf = (C.c_ubyte * A.MGPU_ID_BYTES).from_buffer_copy(unique_id)
        rc = lib.rtr_mgpu_create_rank(device, rank, nranks, buf, frames_in_flight, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"rtr_mgpu_create_rank failed ({rc}): {lib.rtr_mgpu_last_error().decode()}")
        return cls(_handle=h)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.rtr_mgpu_last_error().decode()}")

    def scene_create(self, desc):
        import ctypes as C
        self._check(self.lib.rtr_mgpu_scene_create(self.h, C.byref(desc)), "rtr_mgpu_scene_create")

    def render_async(self, slot, camera, scene_info, params, exchange=True):
        import ctypes as C
        self._check(self.lib.rtr_mgpu_render_async(self.h, slot, C.byref(camera), C.byref(scene_info), C.byref(params), 0 if exchange else 1), "rtr_mgpu_render_async")
        self._extent[slot] = (params.height, params.width, params.bandRows or 8)

    def render_batch_async(self, slots, cameras, scene_infos, params, exchange=True):
        """rtr_mgpu_render_batch_async: len(slots) frames with one launch of the pipeline per rank, then every slot's exchange"""
        import ctypes as C
        from . import _abi as A
        n = len(slots)
        self._check(self.lib.rtr_mgpu_render_batch_async(self.h, (C.c_int * n)(*slots), n, (A.RtrCameraData * n)(*cameras), (A.RtrSceneInfo * n)(*scene_infos),
                                                         C.byref(params), 0 if exchange else 1), "rtr_mgpu_render_batch_async")
        for s1 in slots:
            self._extent[s1] = (params.height, params.width, params.bandRows or 8)

    def wait(self, slot):
        self._check(self.lib.rtr_mgpu_wait(self.h, slot), "rtr_mgpu_wait")

    def render(self, camera, scene_info, params):
        self.render_async(0, camera, scene_info, params)
        self.wait(0)

    def download(self, slot=0):
        H, W, _ = self._extent[slot]
        out = np.zeros((H, W), np.uint32)
        self._check(self.lib.rtr_mgpu_frame_download(self.h, slot, out.ctypes.data, out.nbytes), "rtr_mgpu_frame_download")
        return out

    def download_shard(self, slot=0, local_rank=0):
        H, W, band = self._extent[slot]
        out = np.zeros((shard_rows(H, band, self.info.nranks), W), np.uint32)
        self._check(self.lib.rtr_mgpu_shard_download(self.h, slot, local_rank, out.ctypes.data, out.nbytes), "rtr_mgpu_shard_download")
        return out

    def frame_stats(self, slot=0, local_rank=0):
        import ctypes as C
        from . import _abi as A
        st = A.rtr_frame_stats()
        self._check(self.lib.rtr_mgpu_frame_stats(self.h, slot, local_rank, C.byref(st)), "rtr_mgpu_frame_stats")
        return st

    def frame_device_ptr(self, slot=0):
        import ctypes as C
        from . import _abi as A
        p, n = A.VP(), C.c_size_t()
        self._check(self.lib.rtr_mgpu_frame_device_ptr(self.h, slot, C.byref(p), C.byref(n)), "rtr_mgpu_frame_device_ptr")
        return p.value, n.value

    def close(self):
        if self.h:
            self.lib.rtr_mgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
