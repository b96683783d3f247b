"""Test infrastructure: carries out a rank's exchange plan (rtr_mgpu_plan, include/rtr_mgpu.h — the list of operations
librtr_mgpu.so's enqueue() executes on the GPU) over another transport, so that the product's own offsets, lengths, peers and
ordering are exercised with more than one rank on a machine without GPUs.

  * `plan(rank, nranks, ...)`           the operations, straight from the library (no GPU needed to ask)
  * `check_plans(plans, ...)`           the invariants every set of per-rank plans must satisfy
  * `PlanRunner`                        executes one rank's plan: RENDER calls a caller-supplied shard renderer (the CPU oracle in
                                        the tests), SEND / RECV go through torch.distributed point-to-point calls between
                                        GROUP_START / GROUP_END, DEINTERLEAVE is the numpy restatement of k_deinterleave; stream
                                        and event edges are checked for having been recorded before they are waited for
Nothing here is on the product path."""
import ctypes as C

import numpy as np

from realtimeraytracer_amd import _abi as A
from realtimeraytracer_amd import mgpu


def plan(rank, nranks, width, height, band_rows=8, flags=0, self_exchange=0):
    lib = A.mgpu_lib()
    ops = (A.rtr_mgpu_op * A.MGPU_PLAN_MAX_OPS)()
    n = C.c_int(0)
    rc = lib.rtr_mgpu_plan(rank, nranks, width, height, band_rows, flags, self_exchange, ops, A.MGPU_PLAN_MAX_OPS, C.byref(n))
    if rc != 0:
        raise ValueError(f"rtr_mgpu_plan failed ({rc}): {lib.rtr_mgpu_last_error().decode()}")
    return [dict(kind=o.kind, stream=o.stream, peer=o.peer, buffer=o.buffer, event=o.event, slot=o.slot, offset=int(o.offset), bytes=int(o.bytes)) for o in ops[:n.value]]


def plan_batch(rank, nranks, width, height, nslots, band_rows=8, flags=0, self_exchange=0):
    """rtr_mgpu_plan_batch: the operations of ONE launch of nslots frames (what rtr_mgpu_render_batch_async carries out)"""
    lib = A.mgpu_lib()
    ops = (A.rtr_mgpu_op * A.MGPU_BATCH_PLAN_MAX_OPS)()
    n = C.c_int(0)
    rc = lib.rtr_mgpu_plan_batch(rank, nranks, width, height, band_rows, flags, self_exchange, nslots, ops, A.MGPU_BATCH_PLAN_MAX_OPS, C.byref(n))
    if rc != 0:
        raise ValueError(f"rtr_mgpu_plan_batch failed ({rc}): {lib.rtr_mgpu_last_error().decode()}")
    return [dict(kind=o.kind, stream=o.stream, peer=o.peer, buffer=o.buffer, event=o.event, slot=o.slot, offset=int(o.offset), bytes=int(o.bytes)) for o in ops[:n.value]]


def check_batch_plans(plans, width, height, nslots, band_rows=8, group_per_slot=False):
    """plans[r] = rank r's plan of one launch of nslots frames.  ONE exchange per launch: a single group holding every slot's
    transfers; between a pair of ranks the sends and the receives name the slots in the same order (RCCL matches in posting order);
    every slot's shards land once at shardBytes * src of THAT slot's gather buffer; one render, led by slot 0; a de-interleave and an
    exchange-done record per slot behind the group; every slot's previous exchange waited for before the render."""
    n = len(plans)
    shard = mgpu.shard_rows(height, band_rows, n) * width * 4
    sends, recvs = {}, {}
    for r, ops in enumerate(plans):
        kinds = [o["kind"] for o in ops]
        assert all(0 <= o["slot"] < nslots for o in ops), (r, "slot index")
        i_ren = kinds.index(A.MGPU_OP_RENDER)
        assert kinds.count(A.MGPU_OP_RENDER) == 1 and ops[i_ren]["slot"] == 0 and ops[i_ren]["peer"] == r and ops[i_ren]["bytes"] == shard, (r, "one render, led by slot 0")
        assert ops[i_ren]["buffer"] == (A.MGPU_BUF_GATHER if r == 0 else A.MGPU_BUF_LOCAL)
        guards = [o for o in ops[:i_ren]]
        assert sorted(o["slot"] for o in guards) == list(range(nslots)) and all(o["kind"] == A.MGPU_OP_WAIT and o["event"] == A.MGPU_EV_COMM_DONE and o["stream"] == A.MGPU_STREAM_RENDER for o in guards), (r, "every slot's previous exchange is waited for before the render")
        i_rec = next(i for i, o in enumerate(ops) if o["kind"] == A.MGPU_OP_RECORD and o["event"] == A.MGPU_EV_RENDER_DONE)
        i_wait = next(i for i, o in enumerate(ops) if o["kind"] == A.MGPU_OP_WAIT and o["event"] == A.MGPU_EV_RENDER_DONE)
        assert ops[i_rec]["stream"] == A.MGPU_STREAM_RENDER and ops[i_wait]["stream"] == A.MGPU_STREAM_COMM and i_ren < i_rec < i_wait and ops[i_rec]["slot"] == ops[i_wait]["slot"], (r, "render -> comm edge")
        xfer = [i for i, o in enumerate(ops) if o["kind"] in (A.MGPU_OP_SEND, A.MGPU_OP_RECV)]
        if n > 1 and group_per_slot:
            # the fallback (RTR_MGPU_GROUP_PER_SLOT): nslots groups one behind the other, group j holding exactly slot j's transfers
            starts = [i for i, k in enumerate(kinds) if k == A.MGPU_OP_GROUP_START]
            ends = [i for i, k in enumerate(kinds) if k == A.MGPU_OP_GROUP_END]
            assert len(starts) == len(ends) == nslots and i_wait < starts[0], (r, "a group per slot, after the wait")
            for j, (gs, ge) in enumerate(zip(starts, ends)):
                inside = ops[gs + 1:ge]
                assert gs < ge and (j == 0 or ends[j - 1] < gs), (r, j, "groups do not nest or overlap")
                assert inside and all(o["kind"] in (A.MGPU_OP_SEND, A.MGPU_OP_RECV) and o["slot"] == j for o in inside), (r, j, "group j holds slot j's transfers only")
            assert sum(e - s_ - 1 for s_, e in zip(starts, ends)) == len(xfer), (r, "no transfer outside a group")
        elif n > 1:
            assert kinds.count(A.MGPU_OP_GROUP_START) == 1 and kinds.count(A.MGPU_OP_GROUP_END) == 1, (r, "ONE group per launch")
            gs, ge = kinds.index(A.MGPU_OP_GROUP_START), kinds.index(A.MGPU_OP_GROUP_END)
            assert i_wait < gs < min(xfer) and max(xfer) < ge, (r, "every transfer of every slot inside the one group, after the wait")
        else:
            assert not xfer and A.MGPU_OP_GROUP_START not in kinds
        for i in xfer:
            o = ops[i]
            assert o["stream"] == A.MGPU_STREAM_COMM and o["bytes"] == shard
            if o["kind"] == A.MGPU_OP_SEND:
                assert r != 0 and o["peer"] == 0 and o["buffer"] == A.MGPU_BUF_LOCAL and o["offset"] == 0, (r, "send")
                sends.setdefault((r, 0), []).append(o["slot"])
            else:
                assert r == 0 and o["buffer"] == A.MGPU_BUF_GATHER and o["offset"] == shard * o["peer"], (r, "recv offset = shardBytes * src")
                recvs.setdefault((o["peer"], 0), []).append(o["slot"])
        tail = ops[(len(kinds) - kinds[::-1].index(A.MGPU_OP_GROUP_END)) if n > 1 else (i_wait + 1):]
        de = [o for o in tail if o["kind"] == A.MGPU_OP_DEINTERLEAVE]
        done = [o for o in tail if o["kind"] == A.MGPU_OP_RECORD and o["event"] == A.MGPU_EV_COMM_DONE]
        assert len(de) + len(done) == len(tail), (r, "behind the group: de-interleaves and exchange-done records only")
        assert sorted(o["slot"] for o in done) == list(range(nslots)) and all(o["stream"] == A.MGPU_STREAM_COMM for o in done), (r, "an exchange-done record per slot")
        if r == 0:
            assert sorted(o["slot"] for o in de) == list(range(nslots)) and all(o["bytes"] == width * height * 4 and o["buffer"] == A.MGPU_BUF_FULL for o in de), (r, "a de-interleave per slot")
            for j in range(nslots):
                assert tail.index(next(o for o in de if o["slot"] == j)) < tail.index(next(o for o in done if o["slot"] == j)), (r, j, "de-interleave before the slot's record")
        else:
            assert not de
    assert sends == recvs, ("per pair of ranks, sends and receives must name the slots in the same order", sends, recvs)
    if n > 1:
        assert sorted(sends) == [(src, 0) for src in range(1, n)] and all(v == list(range(nslots)) for v in sends.values()), "every other rank sends every slot's shard once, in slot order"


def check_plans(plans, width, height, band_rows=8):
    """plans[r] = plan of rank r of len(plans).  Raises AssertionError naming the broken invariant."""
    n = len(plans)
    shard = mgpu.shard_rows(height, band_rows, n) * width * 4
    sends, recvs = [], []
    for r, ops in enumerate(plans):
        kinds = [o["kind"] for o in ops]
        # slot reuse: the render waits for the slot's previous exchange before it overwrites the buffers
        assert ops[0]["kind"] == A.MGPU_OP_WAIT and ops[0]["stream"] == A.MGPU_STREAM_RENDER and ops[0]["event"] == A.MGPU_EV_COMM_DONE, (r, "first op")
        ren = ops[1]
        assert ren["kind"] == A.MGPU_OP_RENDER and ren["stream"] == A.MGPU_STREAM_RENDER and ren["peer"] == r and ren["bytes"] == shard and ren["offset"] == 0, (r, "render")
        # rank 0 renders straight into shard 0 of the gather buffer; the others into their own shard
        assert ren["buffer"] == (A.MGPU_BUF_GATHER if r == 0 else A.MGPU_BUF_LOCAL), (r, "render target")
        assert kinds.count(A.MGPU_OP_RENDER) == 1
        # the communication stream waits for the render before anything is sent
        i_rec = next(i for i, o in enumerate(ops) if o["kind"] == A.MGPU_OP_RECORD and o["event"] == A.MGPU_EV_RENDER_DONE)
        i_wait = next(i for i, o in enumerate(ops) if o["kind"] == A.MGPU_OP_WAIT and o["event"] == A.MGPU_EV_RENDER_DONE)
        assert ops[i_rec]["stream"] == A.MGPU_STREAM_RENDER and ops[i_wait]["stream"] == A.MGPU_STREAM_COMM and 1 < i_rec < i_wait, (r, "render -> comm edge")
        xfer = [i for i, o in enumerate(ops) if o["kind"] in (A.MGPU_OP_SEND, A.MGPU_OP_RECV)]
        if n > 1:
            gs, ge = kinds.index(A.MGPU_OP_GROUP_START), kinds.index(A.MGPU_OP_GROUP_END)
            assert kinds.count(A.MGPU_OP_GROUP_START) == 1 and kinds.count(A.MGPU_OP_GROUP_END) == 1
            assert i_wait < gs < min(xfer) and max(xfer) < ge, (r, "every transfer inside the one group, after the wait")
        else:
            assert not xfer and A.MGPU_OP_GROUP_START not in kinds
        for i in xfer:
            o = ops[i]
            assert o["stream"] == A.MGPU_STREAM_COMM and o["bytes"] == shard
            if o["kind"] == A.MGPU_OP_SEND:
                assert r != 0 and o["peer"] == 0 and o["buffer"] == A.MGPU_BUF_LOCAL and o["offset"] == 0, (r, "send")
                sends.append((r, o["peer"], o["bytes"]))
            else:
                assert r == 0 and o["buffer"] == A.MGPU_BUF_GATHER and o["offset"] == shard * o["peer"], (r, "recv offset = shardBytes * src")
                recvs.append((o["peer"], r, o["bytes"]))
        de = [i for i, k in enumerate(kinds) if k == A.MGPU_OP_DEINTERLEAVE]
        if r == 0:
            assert len(de) == 1 and ops[de[0]]["bytes"] == width * height * 4 and ops[de[0]]["buffer"] == A.MGPU_BUF_FULL and ops[de[0]]["stream"] == A.MGPU_STREAM_COMM
            assert de[0] > (max(xfer) + 1 if xfer else i_wait), (r, "de-interleave after the group")
        else:
            assert not de
        last = ops[-1]
        assert last["kind"] == A.MGPU_OP_RECORD and last["event"] == A.MGPU_EV_COMM_DONE and last["stream"] == A.MGPU_STREAM_COMM, (r, "last op records the exchange")
    # every send has its receive and the other way round; rank 0's buffer is covered exactly once
    assert sorted(sends) == sorted(recvs), ("sends and receives do not pair up", sends, recvs)
    assert sorted(src for src, _, _ in recvs) == list(range(1, n)), "one shard from every other rank"
    covered = sorted([0] + [shard * src for src, _, _ in recvs])
    assert covered == [shard * i for i in range(n)], "every shard lands once, at shardBytes * src"


class PlanRunner:
    """One rank's plan carried out on CPU.  `render_shard(shard_index, shard_count, slot) -> uint32 array (rows x width)` (slot: which
    frame of the launch).  Buffers exist per slot of the launch (nslots = 1 for rtr_mgpu_plan's one-frame plan)."""

    def __init__(self, rank, nranks, width, height, band_rows, render_shard, dist=None, nslots=1):
        self.rank, self.n, self.W, self.H, self.band, self.nslots = rank, nranks, width, height, band_rows, nslots
        self.render_shard, self.dist = render_shard, dist
        rows = mgpu.shard_rows(height, band_rows, nranks)
        self.shard_bytes = rows * width * 4
        self.bufs = []
        for _ in range(nslots):
            buf = {A.MGPU_BUF_LOCAL: np.zeros(self.shard_bytes, np.uint8)}
            if rank == 0:
                buf[A.MGPU_BUF_GATHER] = np.zeros(self.shard_bytes * nranks, np.uint8)
                buf[A.MGPU_BUF_FULL] = np.zeros(width * height * 4, np.uint8)
                buf[A.MGPU_BUF_LOCAL] = buf[A.MGPU_BUF_GATHER][:self.shard_bytes]     # rank 0's frame is bound to shard 0 of the gather buffer
            self.bufs.append(buf)
        self.buf = self.bufs[0]
        self.recorded = set()
        self.uses = 0

    def run(self, ops):
        import inspect
        import torch
        takes_slot = len(inspect.signature(self.render_shard).parameters) >= 3
        group, in_group = [], False
        for o in ops:
            k, sl = o["kind"], o.get("slot", 0)
            if k == A.MGPU_OP_WAIT:
                if o["event"] == A.MGPU_EV_COMM_DONE and self.uses == 0:
                    continue                                   # first use of the slot: nothing recorded yet
                assert (o["event"], sl) in self.recorded, ("waits for an event nobody recorded", o)
            elif k == A.MGPU_OP_RECORD:
                self.recorded.add((o["event"], sl))
            elif k == A.MGPU_OP_RENDER:                        # one launch: every slot's shard
                for j in range(self.nslots):
                    img = self.render_shard(o["peer"], self.n, j) if takes_slot else self.render_shard(o["peer"], self.n)
                    img = np.ascontiguousarray(img).view(np.uint8).reshape(-1)
                    assert img.size == o["bytes"], ("shard size", img.size, o["bytes"])
                    self.bufs[j][o["buffer"]][o["offset"]:o["offset"] + o["bytes"]] = img
            elif k == A.MGPU_OP_GROUP_START:
                assert not in_group
                in_group = True
            elif k in (A.MGPU_OP_SEND, A.MGPU_OP_RECV):
                assert in_group, "a transfer outside a group"
                view = torch.from_numpy(self.bufs[sl][o["buffer"]][o["offset"]:o["offset"] + o["bytes"]])
                group.append((k, view, o["peer"]))
            elif k == A.MGPU_OP_GROUP_END:
                in_group = False
                works = [(self.dist.isend(v, dst=p) if kk == A.MGPU_OP_SEND else self.dist.irecv(v, src=p)) for kk, v, p in group]
                for w in works:
                    w.wait()
                group = []
            elif k == A.MGPU_OP_DEINTERLEAVE:
                rows = self.shard_bytes // (self.W * 4)
                g = self.bufs[sl][A.MGPU_BUF_GATHER].view(np.uint32).reshape(self.n, rows, self.W)
                self.bufs[sl][A.MGPU_BUF_FULL][:] = mgpu.assemble_numpy(g, self.H, self.band).view(np.uint8).reshape(-1)
            else:
                raise AssertionError(("unknown operation", o))
        self.uses += 1

    def full(self, slot=0):
        return self.bufs[slot][A.MGPU_BUF_FULL].view(np.uint32).reshape(self.H, self.W)
