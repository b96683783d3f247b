"""The exchange of librtr_mgpu.so with more than one rank in it, on a CPU-only box: rtr_mgpu_plan() returns the list of operations
enqueue() executes for a rank (it IS that code path: csrc/mgpu/rtr_mgpu.cpp walks the same list), so a wrong offset, a missing
receive, an ungrouped transfer or a missing event edge in the library turns these red.  N = 1 ... 16, ragged extents, both flags."""
import ctypes as C

import numpy as np
import pytest

from realtimeraytracer_amd import _abi as A
from realtimeraytracer_amd import mgpu

import plan_exec as PE


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 16])
@pytest.mark.parametrize("extent", [(1920, 1080, 8), (96, 52, 8), (7, 7, 8), (3840, 2160, 16), (640, 360, 24)])
def test_plans_pair_up_and_cover_the_gather_buffer(n, extent):
    W, H, band = extent
    plans = [PE.plan(r, n, W, H, band) for r in range(n)]
    PE.check_plans(plans, W, H, band)
    assert all(len(p) <= A.MGPU_PLAN_MAX_OPS for p in plans)
    # the operation count of rank 0 is what RTR_MGPU_PLAN_MAX_OPS documents: 8 + (n - 1) with an exchange, 6 for one rank
    assert len(plans[0]) == (8 + (n - 1) if n > 1 else 6)


def test_checker_catches_a_wrong_offset_a_missing_receive_and_an_ungrouped_send():
    W, H, band, n = 96, 52, 8, 4
    good = [PE.plan(r, n, W, H, band) for r in range(n)]
    PE.check_plans(good, W, H, band)

    def mutated(fn):
        plans = [[dict(o) for o in p] for p in good]
        fn(plans)
        return plans
    recv = [i for i, o in enumerate(good[0]) if o["kind"] == A.MGPU_OP_RECV]

    def wrong_offset(p): p[0][recv[1]]["offset"] += 4
    def missing_recv(p): del p[0][recv[2]]
    def ungrouped(p): p[2][:] = [o for o in p[2] if o["kind"] not in (A.MGPU_OP_GROUP_START, A.MGPU_OP_GROUP_END)]
    def no_render_edge(p): p[1][:] = [o for o in p[1] if not (o["kind"] == A.MGPU_OP_WAIT and o["event"] == A.MGPU_EV_RENDER_DONE)]
    def no_slot_guard(p): del p[3][0]
    def swapped_peer(p): p[0][recv[0]]["peer"], p[0][recv[1]]["peer"] = p[0][recv[1]]["peer"], p[0][recv[0]]["peer"]
    for fn in (wrong_offset, missing_recv, ungrouped, no_render_edge, no_slot_guard, swapped_peer):
        with pytest.raises((AssertionError, ValueError, StopIteration)):
            PE.check_plans(mutated(fn), W, H, band)


def test_no_exchange_flag_renders_only_and_self_exchange_is_grouped():
    ops = PE.plan(2, 4, 640, 360, 8, flags=A.MGPU_NO_EXCHANGE)
    assert [o["kind"] for o in ops] == [A.MGPU_OP_WAIT, A.MGPU_OP_RENDER]
    ops = PE.plan(0, 1, 640, 360, 8, self_exchange=1)
    kinds = [o["kind"] for o in ops]
    gs, ge = kinds.index(A.MGPU_OP_GROUP_START), kinds.index(A.MGPU_OP_GROUP_END)
    inside = ops[gs + 1:ge]
    # a rank's send to itself and the receive that matches it sit in ONE group: one behind the other on a stream they never complete
    assert [o["kind"] for o in inside] == [A.MGPU_OP_SEND, A.MGPU_OP_RECV]
    assert inside[0]["buffer"] == A.MGPU_BUF_SELF_SRC and inside[1]["buffer"] == A.MGPU_BUF_GATHER and inside[0]["bytes"] == inside[1]["bytes"] == 360 * 640 * 4
    assert ops[1]["kind"] == A.MGPU_OP_RENDER and ops[1]["buffer"] == A.MGPU_BUF_SELF_SRC
    # without the hook a one-rank communicator exchanges nothing
    assert not [o for o in PE.plan(0, 1, 640, 360, 8) if o["kind"] in (A.MGPU_OP_SEND, A.MGPU_OP_RECV, A.MGPU_OP_GROUP_START)]


def test_plan_refuses_bad_arguments():
    lib = A.mgpu_lib()
    ops = (A.rtr_mgpu_op * A.MGPU_PLAN_MAX_OPS)()
    n = C.c_int(0)
    for args in ((2, 2, 8, 8), (-1, 2, 8, 8), (0, 0, 8, 8), (0, A.MGPU_MAX_RANKS + 1, 8, 8), (0, 2, 0, 8), (0, 2, 8, 0)):
        assert lib.rtr_mgpu_plan(args[0], args[1], args[2], args[3], 8, 0, 0, ops, A.MGPU_PLAN_MAX_OPS, C.byref(n)) == -1
    assert lib.rtr_mgpu_plan(0, 8, 64, 64, 8, 0, 0, ops, 3, C.byref(n)) == -1      # no room
    assert b"room for 3" in lib.rtr_mgpu_last_error()


@pytest.mark.parametrize("n", [2, 3, 5, 8])
def test_plans_executed_in_one_process_assemble_the_frame(n):
    """All ranks' plans run against an in-memory mailbox standing in for the transport: a synthetic image whose pixel value encodes
    (y, x) comes back assembled, twice (slot reuse waits for the previous exchange)."""
    W, H, band = 40, 52, 8
    truth = (np.arange(H, dtype=np.uint32)[:, None] << 16) | np.arange(W, dtype=np.uint32)[None, :]

    def shard_of(idx, cnt):
        ys = mgpu.global_rows_of_shard(H, band, cnt, idx)
        out = np.zeros((len(ys), W), np.uint32)
        out[ys >= 0] = truth[ys[ys >= 0]]
        return out

    class Mailbox:
        def __init__(self): self.box = {}
        class _W:
            def __init__(self, fn): self.fn = fn
            def wait(self): self.fn()
    mail = {}

    class Dist:
        def __init__(self, me): self.me = me
        def isend(self, v, dst):
            mail[(self.me, dst)] = v.clone()
            return Mailbox._W(lambda: None)
        def irecv(self, v, src):
            me = self.me
            return Mailbox._W(lambda: v.copy_(mail.pop((src, me))))
    runners = [PE.PlanRunner(r, n, W, H, band, shard_of, Dist(r)) for r in range(n)]
    for _ in range(2):
        for r in range(n - 1, -1, -1):          # senders first: the mailbox has no blocking
            runners[r].run(PE.plan(r, n, W, H, band))
        assert np.array_equal(runners[0].full(), truth)
        assert not mail


@pytest.mark.parametrize("n", [1, 2, 3, 4, 8, 16])
@pytest.mark.parametrize("nslots", [1, 2, 5, 16, 32])
def test_batch_plan_is_one_exchange_per_launch(n, nslots):
    """rtr_mgpu_plan_batch — what rtr_mgpu_render_batch_async executes: ONE ncclGroupStart ... ncclGroupEnd per launch holding every
    slot's transfers (sixteen grouped exchanges per sixteen-slot launch on the rank that also renders and de-interleaves
    became one), slots named in the same order on both ends of every pair of ranks."""
    W, H, band = 96, 52, 8
    plans = [PE.plan_batch(r, n, W, H, nslots, band) for r in range(n)]
    PE.check_batch_plans(plans, W, H, nslots, band)
    assert len(plans[0]) == (5 + nslots * (n + 2) if n > 1 else 3 + 3 * nslots) and len(plans[0]) <= A.MGPU_BATCH_PLAN_MAX_OPS
    if n > 1:
        assert len(plans[1]) == 5 + 3 * nslots
    # the one-frame plan is the batch plan of one slot
    assert PE.plan_batch(min(1, n - 1), n, W, H, 1, band) == PE.plan(min(1, n - 1), n, W, H, band)
    PE.check_plans([PE.plan_batch(r, n, W, H, 1, band) for r in range(n)], W, H, band)


@pytest.mark.parametrize("n", [2, 3, 8, 16])
@pytest.mark.parametrize("nslots", [1, 3, 32])
def test_group_per_slot_fallback_plan(n, nslots):
    """RTR_MGPU_GROUP_PER_SLOT (flag, or RTR_MGPU_GROUP_PER_SLOT=1 at creation): the same launch with one RCCL group per slot — the
    round-3 form, kept reachable should one group of (N - 1) x nslots receives ever be mis-ordered by a communicator.  Same transfers,
    same offsets, same slot order per pair of ranks; only the group boundaries differ."""
    W, H, band = 96, 52, 8
    plans = [PE.plan_batch(r, n, W, H, nslots, band, flags=A.MGPU_GROUP_PER_SLOT) for r in range(n)]
    PE.check_batch_plans(plans, W, H, nslots, band, group_per_slot=True)
    one = [PE.plan_batch(r, n, W, H, nslots, band) for r in range(n)]
    strip = lambda p: [o for o in p if o["kind"] not in (A.MGPU_OP_GROUP_START, A.MGPU_OP_GROUP_END)]
    assert [strip(p) for p in plans] == [strip(p) for p in one]
    assert max(len(p) for p in plans) <= A.MGPU_BATCH_PLAN_MAX_OPS
    if nslots > 1:
        with pytest.raises(AssertionError):
            PE.check_batch_plans(plans, W, H, nslots, band)          # and the one-group checker does tell the two apart


def test_batch_checker_catches_a_receive_outside_the_group_and_a_slot_order_mismatch():
    W, H, band, n, nslots = 96, 52, 8, 4, 3
    good = [PE.plan_batch(r, n, W, H, nslots, band) for r in range(n)]
    PE.check_batch_plans(good, W, H, nslots, band)

    def mutated(fn):
        plans = [[dict(o) for o in p] for p in good]
        fn(plans)
        return plans
    recv = [i for i, o in enumerate(good[0]) if o["kind"] == A.MGPU_OP_RECV]
    ge = next(i for i, o in enumerate(good[0]) if o["kind"] == A.MGPU_OP_GROUP_END)

    def recv_outside_group(p): p[0].insert(ge + 1, p[0].pop(recv[-1]))            # a slot's receive behind ncclGroupEnd
    def second_group(p):                                                          # the old form: a group per slot
        i = recv[n - 1]
        p[0].insert(i, dict(p[0][ge])); p[0].insert(i + 1, dict(good[0][recv[0] - 1]))
    def slot_order(p): p[2][:] = sorted(p[2], key=lambda o: -o["slot"] if o["kind"] == A.MGPU_OP_SEND else 0)      # rank 2 sends its slots backwards
    def missing_guard(p): del p[1][1]
    def missing_done(p): del p[3][-1]
    def wrong_slot_offset(p): p[0][recv[n]]["offset"] += 4
    for fn in (recv_outside_group, second_group, slot_order, missing_guard, missing_done, wrong_slot_offset):
        with pytest.raises((AssertionError, ValueError, StopIteration)):
            PE.check_batch_plans(mutated(fn), W, H, nslots, band)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_batch_plans_executed_in_one_process_assemble_every_slot(n):
    """All ranks' batch plans against an in-memory mailbox that matches transfers between a pair in POSTING ORDER (as RCCL does): three
    slots whose pixels encode (slot, y, x) come back assembled in their own slots, twice."""
    W, H, band, nslots = 40, 52, 8, 3
    truth = [((np.arange(H, dtype=np.uint32)[:, None] << 16) | np.arange(W, dtype=np.uint32)[None, :]) + (j << 28) for j in range(nslots)]

    def shard_of(idx, cnt, slot):
        ys = mgpu.global_rows_of_shard(H, band, cnt, idx)
        out = np.zeros((len(ys), W), np.uint32)
        out[ys >= 0] = truth[slot][ys[ys >= 0]]
        return out

    class _W:
        def __init__(self, fn): self.fn = fn
        def wait(self): self.fn()
    mail = {}

    class Dist:
        def __init__(self, me): self.me = me
        def isend(self, v, dst):
            mail.setdefault((self.me, dst), []).append(v.clone())
            return _W(lambda: None)
        def irecv(self, v, src):
            me = self.me
            return _W(lambda: v.copy_(mail[(src, me)].pop(0)))
    runners = [PE.PlanRunner(r, n, W, H, band, shard_of, Dist(r), nslots=nslots) for r in range(n)]
    for _ in range(2):
        for r in range(n - 1, -1, -1):          # senders first: the mailbox has no blocking
            runners[r].run(PE.plan_batch(r, n, W, H, nslots, band))
        for j in range(nslots):
            assert np.array_equal(runners[0].full(j), truth[j]), j
        assert not any(mail.values())
