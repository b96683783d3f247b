#!/usr/bin/env python3
"""bench.py — headline benchmark of the ray-tracing hot path (BASELINE.json metric:
Mrays/sec and ms/frame at 1920x1080, 1 spp, on 1/2/4/8 MI355X).

A "step" is one frame: the hot path over one batch of synthetic input (the procedural Sponza-class
atrium, ~262 k triangles, 2 area lights; BASELINE config 4), i.e. k_primary -> k_shadow_gen ->
k_shadow_trace4 -> k_resolve through the C ABI, scene resident in HBM.  With N > 1 the SAME frame is
band-sharded over the ranks (strong scaling), gathered to rank 0 with one RCCL gather over xGMI and
de-interleaved there; all of that is inside the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # N > 1, started plainly: ONE process drives the N devices through
                                                           # librtr_mgpu.so (rtr_mgpu_create: ncclCommInitAll, a host thread per rank)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # one process per GPU (rtr_mgpu_create_rank), as the driver launches it
    python bench.py --gpus N --launcher torchrun           # the same, with bench.py starting torch.distributed.run itself as a child
    python bench.py --gpus N --print-launch                # only say how this invocation would run (no GPU touched)

Rank 0 prints ONE JSON line.  `value` = all rays traced per second (primary + shadow, exact count
from the kernels' own counters in an untimed stats pass), whole job.  `roofline` is for the dominant
kernel (k_shadow_trace4, the any-hit traversal of the shadow-ray queue).  The kernel is bound by
vector-instruction ISSUE, not by HBM (its 17 MB tree is cache-resident: a few per cent of its algorithmic bytes
reach the fabric), so the block reports the ceiling that binds — SIMD issue cycles busy / SIMD cycles
of the launch, with the launch duration (HIP events on the launch stream) and the shader clock (s_memtime
over s_memrealtime stamps inside the launch) measured live, lane utilisation from the kernel's counting
form run in this process, and the counter totals of the committed rocprofv3 passes of the same command
(profiles/r03/pmc_roofline.json — used only when workload, kernel revision, triangle count and queue length
match this run) — and, beside it, the second unit that is nearly full (L1 tag look-ups per L1 per clock),
the HBM and L2 fractions and the algorithmic byte rate.  `roofline_secondary` is the one kernel of the frame
that IS HBM-bound (the queue build: bytes from the same counter passes over its live launch time) and
`frame_hbm` the whole frame's HBM bytes over the frame time.
`presented_frame` (N = 1) times the frame the reference presents: five images at 4 spp with the shipped LTC
tables, four a-trous rounds, combine.  `cpu_baseline` is the CPU oracle (a scalar C++ port, oracle/)
timed on the host cores on a bounded sample of the same workload — reported, not a target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0   # same guide: measured streaming copy (SURVEY 8d asks for the fraction of this as well)
L2_PEAK_GBS = 34500.0    # same guide: L2 aggregate, 8 XCDs


def default_frames_in_flight(n_gpus):
    """Frame objects in use per rank; default_batch() of them go into one launch of the pipeline.  A 1-spp frame — a 1/N shard of one
    even more so — is too little work per launch for the latency-bound kernels: many frames per launch, one launch at a time, render
    faster than any number of single-frame launches overlapped (profiles/r03/ab_frame_batch2.log, sweep_frames_per_launch.log,
    sweep_frames_per_launch_32.log: N = 1 2.25 ms per frame with four single-frame launches in flight, 2.13 with eight frames per launch,
    2.05 with sixteen, 2.02 with thirty-two).  Shards of 1/4 and 1/8 do as well with sixteen per launch and two launches in flight as
    with thirty-two per launch (one rank of 8: 0.275 / 0.278 ms per frame), and two launches leave the exchange of one to overlap the
    render of the other.  At N = 2, 3 (thirty-two half or third frames per launch) the same reason asks for two launches in flight:
    sixty-four frame objects (profiles/r03/sweep_two_launches_n2.log: no cost on the rendering side)."""
    return 64 if n_gpus in (2, 3) else 32


def default_batch(n_gpus):
    """frames per launch (rtr_render_batch_async / rtr_mgpu_render_batch_async, at most RTR_MAX_BATCH = 32); --batch 1 = one launch per frame"""
    return 32 if n_gpus < 4 else 16


def launch_sizes(count, batch):
    """`count` frames in as few launches of at most `batch` frames as possible, of equal size (+-1): 20 frames at 16 per launch are
    10 + 10, not 16 + 4 — a short launch is a slow one, and equal launches are what the per-launch figures of the line describe."""
    if count <= 0:
        return []
    nl = -(-count // max(batch, 1))
    base, extra = divmod(count, nl)
    return [base + 1] * extra + [base] * (nl - extra)


def launches_in_flight(n_gpus, groups):
    """Launches a rank keeps enqueued under a latency bound.  A frame's camera is fixed when its launch is enqueued, so a frame is
    (launches in flight) x (launch duration) old when its launch ends.  One GPU: one launch at a time (the next is enqueued when the last has
    been joined).  N > 1: two — while launch k runs, launch k + 1 is already enqueued, so that rank 0's exchange and de-interleave of k
    overlap the render of k + 1 — and the bound is applied to BOTH launches' duration."""
    return 1 if n_gpus <= 1 else max(1, min(2, groups))


def latency_fields(B, groups, steps, ms_per_step, latency):
    """frames_per_launch / frame_latency_ms of the JSON line, from the launches the timed region made (not from the probe):
    the longest timed launch x ms_per_step x the launches in flight"""
    timed = launch_sizes(steps, B)
    longest = max(timed) if timed else B
    latency = dict(latency, launches_in_flight=groups, frames_per_launch_limit=B,
                   frame_latency_is="launches_in_flight x the longest timed launch's frames x ms_per_step: the age of a launch's first frame when the launch ends "
                                    "(its camera was fixed when the launch was enqueued)")
    return {"frames_per_launch": longest, "frame_latency_ms": round(groups * longest * ms_per_step, 4), "latency": latency, "timed_launches": timed}


# sponza_mixed: the atrium with the triangle-size mix of a real asset (large walls / column slivers beside fine cloth, an alpha-tested layer);
# a second line for profiles/, never the headline (BASELINE config 4 is sponza_class)
WORKLOADS = {"sponza_class": "sponza_class", "cornell": "cornell_box", "bunny_class": "bunny_class", "sponza_mixed": "sponza_mixed"}
PROBE_MARGIN = 0.98   # a probed launch must fit the latency bound with 2 % to spare: the timed launches run a per cent or two off the probes
PATH_PERIOD = 192     # cameras of the scripted walk (scenes.SceneSetup.camera_path: 96 frames forward, 96 back); frame i uses camera i mod 192


class CameraSource:
    """The camera of frame i.  'path' (default): the reference's loop moves its camera every frame (Window::processInput, src/app/window.cppm:68-133,
    then Camera::updateGPUData) — here the scripted walk of scenes.SceneSetup.camera_path, so the frames of a launch are DIFFERENT
    views, as a real-time caller's are.  'static': every frame the set-up's camera (rounds 1-3; twenty copies of one visibility problem share the caches)."""

    def __init__(self, setup, mode):
        self.setup = setup
        self.path = setup.camera_path(PATH_PERIOD) if (mode == "path" and setup.cam_args and setup.walk_scale > 0) else None
        self.mode = "path" if self.path else "static"

    def index(self, i):
        return i % PATH_PERIOD if self.path else 0

    def camera(self, i):
        return self.path[i % PATH_PERIOD][0] if self.path else self.setup.camera

    def info(self, i, frame_no=None):
        """SceneInfo of frame i (frame_no: the seed when it is not i — the K frames accumulated into one step share step i's camera)"""
        pos = self.path[i % PATH_PERIOD][1] if self.path else None
        return self.setup.scene_info(i if frame_no is None else frame_no, cam_pos=pos)


def pick_frames_per_launch(limit_ms, cap, time_launch):
    """The largest launch (frames, <= cap) whose duration fits limit_ms.  time_launch(c) -> milliseconds of one launch of c frames, warmed.
    A few probes: an estimate from a small launch (ms per frame falls with the launch length, so the estimate is low), then upwards while
    it fits, or downwards until it does."""
    small = min(4, cap)
    c = max(1, min(cap, int(limit_ms / (time_launch(small) / small))))
    ms = time_launch(c) if c != small else time_launch(small)
    if ms <= limit_ms:
        while c < cap:
            nxt = min(cap, max(c + 1, int(c * limit_ms / ms)))
            ms2 = time_launch(nxt)
            if ms2 > limit_ms:
                if nxt == c + 1:
                    break
                cap = nxt - 1                       # the jump overshot: go on in single steps below it
                continue
            c, ms = nxt, ms2
    else:
        while c > 1:
            c = max(1, min(c - 1, int(c * limit_ms / ms)))
            ms = time_launch(c)
            if ms <= limit_ms:
                break
    return c


def launch_plan(args, env, argv):
    """How this invocation runs — decided before torch or any HIP library is imported, testable without a GPU (--print-launch).
    mode: 'single' (N = 1), 'rank' (this process is one rank of a torch.distributed.run job), 'inproc' (N > 1 started plainly: one
    process drives the N devices through librtr_mgpu.so), 'torchrun-child' (N > 1 started plainly with --launcher torchrun)."""
    n = args.gpus
    fif = args.frames_in_flight or default_frames_in_flight(max(n, getattr(args, 'emulate_rank_of', 0) or 0))
    # HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with 8 frames in flight every frame's stream gets its own
    env_defaults = {"GPU_MAX_HW_QUEUES": "8"} if fif >= 8 else {}
    base = {"n_gpus": n, "frames_in_flight": fif, "frames_per_launch": max(1, min(args.batch or default_batch(max(n, getattr(args, 'emulate_rank_of', 0) or 0)), fif)), "env_defaults": env_defaults}
    if n < 1:
        return dict(base, mode="error", why=f"--gpus {n}")
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != n:
            return dict(base, mode="error", why=f"--gpus {n} but WORLD_SIZE={ws}: under torch.distributed.run start one process per GPU (--nproc-per-node {n})")
        return dict(base, mode="rank" if n > 1 else "single", world=n, library_entry="rtr_mgpu_create_rank" if n > 1 else None)
    if n == 1 and env.get("RTR_BENCH_FORCE_INPROC") != "1":     # (forced: a one-GPU rehearsal of the in-process N > 1 path, one rank through real RCCL)
        return dict(base, mode="single", world=1, library_entry=None)
    if args.launcher == "torchrun" and n > 1:
        child = [a for a in argv if a not in ("--print-launch",)]
        for i, a in enumerate(child):               # the child must not start a grandchild
            if a == "--launcher":
                del child[i:i + 2]
                break
        child = [a for a in child if not a.startswith("--launcher=")]
        port = env.get("MASTER_PORT")
        if not port:                                # a free port for the rendezvous on 127.0.0.1
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = str(so.getsockname()[1])
        return dict(base, mode="torchrun-child", world=n, library_entry="rtr_mgpu_create_rank",
                    argv=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                          "--master-port", port, os.path.join(ROOT, "bench.py")] + child)
    # rehearsal on a one-GPU box: the N ranks share device 0 (the library's test hook) and tests/fake_rccl stands in for RCCL (LD_PRELOAD);
    # the line says so ("rehearsal") and its rate is that of ONE GPU time-sliced by N ranks — what the exchange machinery costs, not scaling
    shared = env.get("RTR_MGPU_TEST_SHARED_DEVICE") == "1"
    return dict(base, mode="inproc", world=n, library_entry="rtr_mgpu_create", devices=[0] * n if shared else list(range(n)), shared_device=shared)


def build_setup(args, scenes, np):
    """the workload: a procedural BASELINE scene, or a real asset (--obj)"""
    W, H = args.width, args.height
    if args.obj:
        vs = np.array([[float(t) for t in ln.split()[1:4]] for ln in open(args.obj, errors="replace") if ln.startswith("v ")], dtype=np.float64)
        if len(vs) == 0:
            raise SystemExit(f"bench.py: {args.obj} has no vertices")
        lo, hi = vs.min(0), vs.max(0)
        c, e = (lo + hi) / 2, np.maximum(hi - lo, 1e-6)
        view = np.array([float(t) for t in args.obj_view.split(",")])
        cam = tuple(float(x) for x in (c + view * e))
        side = float(0.25 * max(e[0], e[2]))
        light = (12.0, (1.0, 0.95, 0.9), (float(c[0]), float(hi[1] - 0.02 * e[1]), float(c[2])), (side, side, 1.0), (90.0, 0.0, 0.0))
        setup = scenes.custom_obj(args.obj, os.path.dirname(os.path.abspath(args.obj)), cam, tuple(float(x) for x in c), fov_y=60.0, width=W, height=H, lights=[light])
        args.workload = "obj:" + os.path.basename(args.obj)
        return setup
    return getattr(scenes, WORKLOADS[args.workload])(W, H)


FIRST_EXCHANGE_TIMEOUT_MS = 60000        # watchdog of the verified first exchange (RTR_MGPU_TIMEOUT_MS, if set, wins)


def start_line(what, **kw):
    """One line on stderr about how an N > 1 run starts (stdout carries the ONE JSON line): the first contact with real RCCL must not
    be a silent hang or an anonymous traceback in the driver's log."""
    print("bench.py: N>1 start: " + what + " | " + " ".join(f"{k}={v}" for k, v in kw.items()), file=sys.stderr, flush=True)


def start_failure(args, stage, why, rccl=None, code=3):
    """The run cannot start: the diagnostic on stderr AND as the JSON line on stdout (value null), then a non-zero exit."""
    start_line(f"FAILED at {stage}", why=why)
    print(json.dumps({"metric": "Mrays/sec", "value": None, "unit": "Mrays/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                      "error": {"stage": stage, "why": str(why)}, "rccl": rccl}), flush=True)
    raise SystemExit(code)


def verified_launches(mg, A, cams, p_run, K, B, unsharded):
    """The FIRST launches of an N > 1 run are verified ones, joined under a watchdog: (1) one frame — every rank's render, the grouped
    exchange, the de-interleave; (2) one launch of B frames in B slots, the shape the timed launches have (every slot's shards in the
    launch's exchange) — each assembled frame compared with rank 0's unsharded render of the same camera.  unsharded(i) -> the
    unsharded frame i as an array (rank 0) or None (the other ranks of a one-process-per-GPU job).
    Returns {"pixels_differing": n or None, "ms": first exchange, "batch_frames": B, "batch_ms": ...}; RuntimeError when the library
    reports a failure or the watchdog fires."""
    lib = A.mgpu_lib()
    own_watchdog = "RTR_MGPU_TIMEOUT_MS" not in os.environ
    if own_watchdog:
        lib.rtr_mgpu_set_timeout_ms(mg.h, FIRST_EXCHANGE_TIMEOUT_MS)
    try:
        t = time.perf_counter()
        for j, p in enumerate(p_run):
            mg.render_async(0, cams.camera(0), cams.info(0, j if K > 1 else None), p, exchange=(j == len(p_run) - 1))
        mg.wait(0)
        ms = (time.perf_counter() - t) * 1e3
        want = unsharded(0)
        bad = int((mg.download(0) != want).sum()) if want is not None else None
        batch_ms = None
        if B > 1 and K == 1:
            t = time.perf_counter()
            slots = list(range(B))
            mg.render_batch_async(slots, [cams.camera(i) for i in slots], [cams.info(i) for i in slots], p_run[0])
            for b in slots:
                mg.wait(b)
            batch_ms = (time.perf_counter() - t) * 1e3
            for b in slots:
                want = unsharded(b)
                if want is not None:
                    bad += int((mg.download(b) != want).sum())
    finally:
        if own_watchdog:
            lib.rtr_mgpu_set_timeout_ms(mg.h, 120000)
    return {"pixels_differing": bad, "ms": ms, "batch_frames": B if (B > 1 and K == 1) else 0, "batch_ms": batch_ms}


def bring_up(args, create, verify, agree=lambda ok: ok, root=True, how=""):
    """Makes the communicator and proves the exchange before anything is timed.  create() -> MultiGpu with its scene
    (RuntimeError on failure); verify(mg) -> verified_launches(...); agree(ok) -> whether EVERY rank says ok (a collective in a
    one-process-per-GPU job: a rank that failed must not leave the others inside one).
    Two attempts: the plan's ONE RCCL group per launch; if its verified launches fail, time out or come out wrong, the communicator is
    given up and everything is made again with one group PER SLOT (RTR_MGPU_GROUP_PER_SLOT=1, read at creation) — the fallback that
    was only ever reachable by hand.  A run whose second attempt fails too ends here: the reason on stderr and in the JSON line, rc != 0."""
    forced = os.environ.get("RTR_MGPU_GROUP_PER_SLOT") == "1"
    for per_slot in ([True] if forced else [False, True]):
        if per_slot:
            os.environ["RTR_MGPU_GROUP_PER_SLOT"] = "1"
        mg, rccl0, why, res = None, None, None, None
        try:
            mg = create()
            rccl0 = {"nranks": int(mg.info.nranks), "nlocal": int(mg.info.nlocal), "version": int(mg.info.rcclVersion), "group_per_slot": per_slot}
        except RuntimeError as e:
            why = e
        if not agree(why is None):                  # no communicator, no scene: not a question of grouping — no second attempt
            if root:
                start_failure(args, how + " / rtr_mgpu_scene_create", why or "another rank failed (its stderr says why)", rccl0)
            raise SystemExit(3)
        if root:
            start_line("communicator up", rccl_version=rccl0["version"], nranks=rccl0["nranks"], nlocal=rccl0["nlocal"], group_per_slot=int(per_slot))
        try:
            res = verify(mg)
        except RuntimeError as e:
            why = e
        ran = agree(why is None)
        good = ran and agree(not root or res["pixels_differing"] == 0)
        if root and ran:
            rccl0.update(first_exchange_verified=(res["pixels_differing"] == 0), first_exchange_pixels_differing=res["pixels_differing"], first_exchange_ms=round(res["ms"], 2),
                         first_batch_frames=res["batch_frames"], first_batch_ms=round(res["batch_ms"], 2) if res["batch_ms"] else None)
            start_line("first exchange " + ("verified" if good else "WRONG"), pixels_differing=res["pixels_differing"], ms=round(res["ms"], 1), batch_frames=res["batch_frames"],
                       grouping="one group per slot" if per_slot else "one group per launch")
        if good:
            return mg, rccl0
        if root and rccl0 is not None and not ran:
            rccl0["first_exchange_verified"] = False
        try:
            mg.close()
        except Exception:      # noqa: BLE001
            pass
        if per_slot:                                # nothing left to fall back to
            if root:
                if ran:
                    start_failure(args, "first exchange verification", f"the assembled frames differ from rank 0's unsharded renders in {res['pixels_differing']} pixels", rccl0, code=4)
                start_failure(args, "first grouped exchange (ncclGroupStart ... ncclGroupEnd, k_deinterleave)", why or "another rank failed (its stderr says why)", rccl0)
            raise SystemExit(4 if ran else 3)
        if root:
            start_line("the verified launches " + ("came out WRONG" if ran else f"FAILED ({why})") + " with one RCCL group per launch: giving the communicator up, again with one group per slot")


def run_inproc(args, K, plan):
    """N > 1 from ONE process: what a C++ application would do with include/rtr_mgpu.h.  rtr_mgpu_create makes the communicator
    (ncclCommInitAll) and a host thread per rank; every step enqueues the frame's N shards, the grouped send / recv to rank 0 on the
    communication streams and k_deinterleave, into one of the frame slots; the timed region is bracketed by a join of every slot of
    every rank and a device synchronise of every GPU, on one clock."""
    import numpy as np
    import torch

    from realtimeraytracer_amd import _abi as A
    from realtimeraytracer_amd import api, mgpu, scenes

    N, W, H, S = args.gpus, args.width, args.height, args.spp
    os.environ.setdefault("RTR_SCENE_CACHE", os.path.join("/tmp", "rtr_scene_cache_rank0"))
    nbuf = plan["frames_in_flight"]
    # the library says what is wrong when the devices are not there ("8 devices requested, 1 present"): no check of our own before it
    start_line("one process drives the ranks (rtr_mgpu_create: ncclCommInitAll, a host thread per rank)", asked=N, devices_present=torch.cuda.device_count(),
               devices=plan["devices"], shared_device=bool(plan.get("shared_device")))
    if not plan.get("shared_device") and torch.cuda.device_count() < N:      # the library's own message, before a scene is built for nothing
        try:
            mgpu.MultiGpu(devices=plan["devices"], frames_in_flight=1).close()
        except RuntimeError as e:
            start_failure(args, "rtr_mgpu_create (ncclCommInitAll)", e)
    setup = build_setup(args, scenes, np)
    cams = CameraSource(setup, args.camera)
    images = A.IMAGES_FRAMEBUFFER | (A.IMG_BIT(A.IMAGE_HDR) if K > 1 else 0)

    def params(collect=0, shard_index=0, shard_count=1, j=0):
        return api.make_params(W, H, spp=S, shadow_rays=args.shadow_rays, images=images, band_rows=args.band_rows, shard_index=shard_index,
                               shard_count=shard_count, accumulate=1 if j > 0 else 0, accumulated_frames=j, collect_stats=collect, pipeline=args.pipeline)

    # untimed: exact ray counts of the whole frame (they do not depend on how the frame is cut), on device 0 with a scene of its own
    ctx = api.Context(0)
    scene = api.Scene(ctx, setup.desc)
    sstats = scene.stats()
    whole = api.Frame(ctx, W, H, images)
    timed = range(args.warmup, args.warmup + args.steps)
    by_cam = {}
    for c in sorted({cams.index(i) for i in timed}):      # ray counts depend on the view, not on the seed: one counting render per distinct camera
        api.render(scene, cams.camera(c), cams.info(c), params(collect=1), whole)
        fs = whole.stats()
        by_cam[c] = (int(fs.numRays), int(fs.numPrimaryRays))
    rays_total = sum(by_cam[cams.index(i)][0] for i in timed) * K
    primary_total = sum(by_cam[cams.index(i)][1] for i in timed) * K
    rays_per_frame, primary_per_frame = rays_total // max(args.steps, 1), primary_total // max(args.steps, 1)

    p_run = [params(0, j=j) for j in range(K)]
    kern = {"primary": 0.0, "shadow_gen": 0.0, "shadow_trace": 0.0, "shadow_tail": 0.0, "resolve": 0.0, "n": 0}
    inflight = [False] * nbuf

    # frames per launch: what was asked for, capped by what one launch of a rank's shard can address (rtr_render_batch_limit)
    B = max(1, min(plan["frames_per_launch"], nbuf, api.render_batch_limit(scene, params(0, 0, N), setup.num_lights))) if K == 1 else 1

    # the communicator, and the first launches: verified — before anything is timed, probed or warmed up (bring_up)
    def create():
        m = mgpu.MultiGpu(devices=plan["devices"], frames_in_flight=nbuf)
        try:
            m.scene_create(setup.desc)
        except RuntimeError:
            m.close()
            raise
        return m

    def unsharded(i):
        for j in range(K):
            api.render(scene, cams.camera(i), cams.info(i, j if K > 1 else None), params(0, j=j), whole)
        return whole.download()
    mg, rccl0 = bring_up(args, create, lambda m: verified_launches(m, A, cams, p_run, K, B, unsharded), how="rtr_mgpu_create (ncclCommInitAll)")

    def collect(b):
        if not inflight[b]:
            return
        mg.wait(b)
        inflight[b] = False
        st = mg.frame_stats(b, 0)
        kern["primary"] += st.primaryMs; kern["shadow_gen"] += st.shadowGenMs
        kern["shadow_trace"] += st.shadowTraceMs; kern["shadow_tail"] += st.shadowTailMs; kern["resolve"] += st.resolveMs; kern["n"] += 1

    groups, launch_no, last_slot = max(nbuf // B, 1), [0], [0]

    def step(i):
        b = i % nbuf
        last_slot[0] = b
        collect(b)                                      # frame i - nbuf: done long ago unless the host runs ahead
        for j in range(K):
            mg.render_async(b, cams.camera(i), cams.info(i, j if K > 1 else None), p_run[j], exchange=(j == K - 1))
        inflight[b] = True

    def run_steps(first, count):
        """`count` frames from frame `first` on: B per launch of the pipeline on every rank (rtr_mgpu_render_batch_async), each with its own exchange"""
        if B == 1:
            for i in range(first, first + count):
                step(i)
            return
        i = first
        for c in launch_sizes(count, B):
            g0 = (launch_no[0] % groups) * B            # fixed groups of slots: the slot that leads a launch owns its scratch
            launch_no[0] += 1
            bufs = [g0 + j for j in range(c)]
            last_slot[0] = bufs[-1]
            for b in bufs:
                collect(b)
            mg.render_batch_async(bufs, [cams.camera(i + j) for j in range(c)], [cams.info(i + j) for j in range(c)], p_run[0])
            for b in bufs:
                inflight[b] = True
            i += c

    def drain():
        for b in range(nbuf):
            collect(b)

    def sync_all():
        for d in sorted(set(plan["devices"])):
            torch.cuda.synchronize(d)

    for b in range(nbuf):                               # set-up, not warm-up: every slot allocates on its first render
        step(b)
    drain()
    if B > 1:                                           # ... and the leading slots' scratch grows to the launch on the first batched one
        biggest = max(launch_sizes(args.steps, B) + launch_sizes(args.warmup, B) + [1])
        for _ in range(groups):
            run_steps(0, biggest)
        drain()
    # frames per launch under the latency bound (main(): "frames per launch under a latency bound"): single launches, probed
    latency = {"limit_ms": args.max_latency_ms or None, "probe_margin": PROBE_MARGIN, "probes": [], "chosen_by": "--batch" if args.batch else ("default" if not args.max_latency_ms else "probe")}
    if args.max_latency_ms > 0 and not args.batch and B > 1:
        cap = B

        def probe(c):
            nonlocal B, groups
            B, groups = c, 1
            best = None
            for k in range(4):
                sync_all()
                t = time.perf_counter()
                run_steps(0, c)
                drain()
                sync_all()
                dt = (time.perf_counter() - t) * 1e3
                if k:
                    best = dt if best is None else min(best, dt)
            latency["probes"].append({"frames": c, "launch_ms": round(best, 4)})
            return best
        lif = launches_in_flight(N, max(nbuf // 2, 1))      # N > 1: two launches enqueued, so a single launch may take half the bound
        c = pick_frames_per_launch(args.max_latency_ms * PROBE_MARGIN / lif, max(1, min(cap, nbuf // lif)), probe)
        B, groups = c, launches_in_flight(N, max(nbuf // c, 1))
        for _ in range(groups):                         # the slots that lead launches of this size grow their scratch now, not in the timed region
            run_steps(0, B)
        drain()
    run_steps(0, args.warmup)
    drain()
    kern.update({k: 0.0 for k in kern}); kern["n"] = 0
    sync_all()
    import ctypes as C
    info0 = A.rtr_mgpu_info()
    A.mgpu_lib().rtr_mgpu_get_info(mg.h, C.byref(info0))         # the host-time counters before the timed region (set-up allocates)
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    drain()
    sync_all()
    elapsed = time.perf_counter() - t0

    # the assembled frame of the last step against the same frame rendered unsharded on device 0
    last_i = args.warmup + args.steps - 1
    for j in range(K):
        api.render(scene, cams.camera(last_i), cams.info(last_i, j if K > 1 else None), params(0, j=j), whole)
    bad = int((mg.download(last_slot[0]) != whole.download()).sum())
    info = A.rtr_mgpu_info()
    A.mgpu_lib().rtr_mgpu_get_info(mg.h, C.byref(info))          # after the run: with the host time of the ranks' threads
    host_frames = max(int(info.enqueuedFrames) - int(info0.enqueuedFrames), 1)
    host_ms, host_rccl_ms = (info.enqueueHostMs - info0.enqueueHostMs) / host_frames, (info.enqueueRcclMs - info0.enqueueRcclMs) / host_frames
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    n = max(kern["n"], 1)
    out = {
        "metric": "Mrays/sec at 1920x1080 1spp (all rays: primary + shadow)" if (W, H, S, K) == (1920, 1080, 1, 1) else f"Mrays/sec at {W}x{H} {S}spp" + (f" x{K} frames accumulated in HDR" if K > 1 else ""),
        "value": round(rays_total / elapsed / 1e6, 2), "unit": "Mrays/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "ms_per_frame": round(ms_per_step / K, 4), "higher_is_better": True,
        **latency_fields(B, groups, args.steps, ms_per_step, latency),
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": ("file " + args.obj) if args.obj else "synthetic",
        "config": {"workload": f"{args.workload} {W}x{H} {S}spp{' x%d accumulated frames per step' % K if K > 1 else ''}, {sstats.numTriangles} triangles, {setup.num_lights} area lights, "
                               f"{args.shadow_rays} shadow rays/light-triangle, band-sharded x{N}",
                   "camera": ("a new view every frame: scripted walk (applyInput = Window::processInput), period %d frames" % PATH_PERIOD) if cams.mode == "path" else "static: every frame the same view",
                   "rays_per_frame": rays_per_frame, "primary_rays_per_frame": primary_per_frame, "rays_per_frame_is": "mean over the frames of the timed region",
                   "pipeline": "wavefront" if fs.pipelineUsed == 2 else "megakernel",
                   "bvh": {"nodes": int(sstats.numNodes), "max_depth": int(sstats.maxDepth), "lds_stack_entries": int(sstats.stackEntries), "build_ms": round(float(sstats.buildMs), 1)}},
        "primary_mrays_per_s": round(primary_total / elapsed / 1e6, 2),
        "frames_in_flight": nbuf,
        "kernels_ms_in_flight_event_brackets": {k: round(v / n, 4) for k, v in kern.items() if k != "n"},
        "kernels_scope": "rank 0's shard, per frame (a launch covers frames_per_launch frames); HIP-event brackets",
        # what RCCL saw: the size of the communicator, how many of its ranks this process drives, the library that is loaded
        "rccl": {"nranks": int(info.nranks), "nlocal": int(info.nlocal), "version": int(info.rcclVersion), "launch": "one process, rtr_mgpu_create (ncclCommInitAll, a host thread per rank)",
                 "first_exchange_verified": rccl0["first_exchange_verified"], "first_exchange_ms": rccl0["first_exchange_ms"],
                 "first_batch_frames": rccl0["first_batch_frames"], "first_batch_ms": rccl0["first_batch_ms"], "group_per_slot": rccl0["group_per_slot"],
                 "exchange": "grouped ncclSend / ncclRecv to rank 0 on a communication stream + k_deinterleave (librtr_mgpu.so, plan = rtr_mgpu_plan)",
                 "env": plan["env_defaults"],
                 # host time of the slowest rank's thread per frame it enqueued (stream waits, the launch, RCCL calls, event records): the
                 # ranks' threads run side by side, so this — not its sum over ranks — is what must stay below a shard's GPU time
                 "host_enqueue_ms_per_frame": round(host_ms, 4), "of_which_inside_rccl_calls": round(host_rccl_ms, 4)},
        "verify": {"assembled_vs_unsharded_pixels_differing": bad, "frame": last_i},
        "rehearsal": ("the %d ranks share ONE GPU (RTR_MGPU_TEST_SHARED_DEVICE=1; rcclVersion 99999 = tests/fake_rccl, a same-device double of RCCL): "
                      "the rate is one GPU's, time-sliced — not a scaling point" % N) if plan.get("shared_device") else None,
        "roofline": None, "roofline_note": "the roofline block is reported for the one-GPU run of the same workload (N = 1 line)",
        "cpu_baseline": None,
    }
    print(json.dumps(out), flush=True)
    mg.close()
    if bad:
        raise SystemExit(f"bench.py: {bad} pixels of the assembled frame differ from the unsharded frame")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="sponza_class", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--shadow-rays", type=int, default=3)
    ap.add_argument("--pipeline", type=int, default=0, help="0 default, 1 megakernel, 2 wavefront")
    ap.add_argument("--band-rows", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + RTR_BENCH_SAME_DEVICE=1 rehearses the N>1 control flow with several ranks on ONE GPU (shards staged through host memory)")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="N",
                    help="single-GPU rehearsal of what ONE rank does in an N-GPU run: renders shard 0 of N with the same two-stream "
                         "frame pipelining, without the gather (the printed value is this rank's rays/s, not a job total)")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="frame objects in use, each with its own stream (0 = default: 32; 1 = one frame at a time)")
    ap.add_argument("--batch", type=int, default=0, metavar="B",
                    help="frames per launch of the pipeline (rtr_render_batch_async; 0 = default: 32 below four GPUs, 16 and two launches in flight from four on; "
                         "1 = one launch per frame; capped by rtr_render_batch_limit).  The frames in flight are "
                         "rendered in groups of B: every kernel of the pipeline is launched once per group over B frames' work")
    ap.add_argument("--max-latency-ms", type=float, default=16.7, metavar="L",
                    help="frames are rendered in launches of several (throughput) — but a caller sees the FIRST frame of a launch when the launch ends: "
                         "pick the largest launch whose duration (frames_per_launch x ms_per_step = frame_latency_ms) fits L, measured before the timed region "
                         "(default 16.7 = one 60-Hz refresh; 0 = no bound).  An explicit --batch wins.")
    ap.add_argument("--camera", default="path", choices=["path", "static"],
                    help="'path' (default): the camera moves every frame along a scripted walk (applyInput, the reference's Window::processInput); "
                         "'static': every frame the same view (rounds 1-3)")
    ap.add_argument("--accumulate", type=int, default=1, metavar="K",
                    help="a step = K frames (frame = 0..K-1) summed in the float HDR buffer and tonemapped once (BASELINE config 5: "
                         "--width 3840 --height 2160 --accumulate 16)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="shortcut for a BASELINE.json config: 1 cornell 256x256 (the reference's CPU-runnable case), 2 cornell 1080p, 3 bunny_class 1080p 4 spp, 4 (default) sponza_class 1080p, "
                         "5 sponza_class 4K x16 accumulated")
    ap.add_argument("--isolated-frames", type=int, default=10,
                    help="after the timed region, render this many frames ONE AT A TIME to report per-kernel durations free of "
                         "cross-frame overlap (0 = skip; profiles/run_rocprof.sh skips it so rocprof's averages cover the timed launches only)")
    ap.add_argument("--present-frames", type=int, default=20,
                    help="after the timed region (N=1), time this many PRESENTED frames of the same scene as the reference's frame loop makes them "
                         "(application.cppm:391-457): ray-gen of all five images at 4 spp with the shipped LTC tables, four a-trous rounds on the "
                         "sampled pair, combine -> FINAL; reported as `presented_frame` (0 = skip)")
    ap.add_argument("--verify", action="store_true", help="after timing, check the assembled frame against the oracle on a row sample")
    ap.add_argument("--obj", default=None, metavar="PATH",
                    help="render a real asset instead of the procedural stand-in (SURVEY 8d: 'real bunny.obj / sponza.obj accepted if present'): "
                         "an OBJ file with its MTL / textures beside it; the camera looks at the centre of its bounds from --obj-view, one area light "
                         "hangs under the top of the bounds")
    ap.add_argument("--obj-view", default="-0.45,0.15,0.05", help="camera position as fractions of the bounds' extent from the centre (x,y,z)")
    ap.add_argument("--launcher", default="auto", choices=["auto", "inproc", "torchrun"],
                    help="how an N > 1 run that was started plainly (no WORLD_SIZE in the environment) gets its ranks: 'inproc' (= auto) "
                         "drives all N devices from this process through rtr_mgpu_create; 'torchrun' starts "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process, before this process has "
                         "touched the GPU, and relays its JSON line and return code")
    ap.add_argument("--print-launch", action="store_true", help="print, as one JSON line, how this invocation would be run, and exit")
    args = ap.parse_args()

    if args.config == 1:
        args.workload, args.width, args.height = "cornell", 256, 256
    elif args.config == 2:
        args.workload = "cornell"
    elif args.config == 3:
        args.workload, args.spp = "bunny_class", 4
    elif args.config == 5:
        args.width, args.height, args.accumulate = 3840, 2160, 16
    K = max(args.accumulate, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    plan = launch_plan(args, os.environ, sys.argv[1:])
    if args.print_launch:
        print(json.dumps(plan), flush=True)
        return
    if plan["mode"] == "error":
        raise SystemExit("bench.py: " + plan["why"])
    for k, v in plan["env_defaults"].items():          # before any HIP library is loaded
        os.environ.setdefault(k, v)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if plan["mode"] == "torchrun-child":
        # nothing in this process has touched the GPU (torch is not even imported yet): start the ranks as a child, relay
        import subprocess
        r = subprocess.run(plan["argv"], cwd=ROOT)
        raise SystemExit(r.returncode)
    if plan["mode"] == "inproc":
        return run_inproc(args, K, plan)
    world = plan["world"]
    os.environ.setdefault("RTR_SCENE_CACHE", os.path.join("/tmp", f"rtr_scene_cache_rank{rank}"))

    import numpy as np
    import torch
    import torch.distributed as dist

    from realtimeraytracer_amd import _abi as A
    from realtimeraytracer_amd import api, mgpu, scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the ray-tracing path has no CPU fallback")
    if os.environ.get("RTR_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # RTR_BENCH_FORCE_DIST=1 with one rank takes the N>1 code path (process group, gather, de-interleave, verify) through real
    # RCCL on a single GPU: a rehearsal of the collective calls, not a measurement
    dist_on = world > 1 or os.environ.get("RTR_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")

    W, H, S = args.width, args.height, args.spp
    setup = build_setup(args, scenes, np)
    cams = CameraSource(setup, args.camera)
    emu = args.emulate_rank_of if (world == 1 and args.emulate_rank_of > 1) else 0
    nshards = emu if emu else world
    # Several frames are kept in flight on separate streams (one context each, ONE shared scene): the tails of one
    # frame's latency-bound kernels, its small kernels and its RCCL gather overlap the next frames' kernels.  At N=1 that
    # is worth ~13 % (part of the 1.1 ms of primary / queue-build / resolve work hides under the traversal kernel); for a
    # 1/8-frame shard no single kernel can fill the GPU (259 k primary rays for 524 k lane slots), so it is where the
    # strong scaling comes from (one rank of 8: 0.82 -> 0.44 ms per frame).  Frames are independent, so results are
    # unchanged.  Per-kernel durations are taken from a one-frame-at-a-time pass after the timed region (see below).
    nbuf = plan["frames_in_flight"]
    ctxs = [api.Context(local_rank) for _ in range(nbuf)]
    streams = [torch.cuda.Stream(device=device) for _ in range(nbuf)]
    for c, st_ in zip(ctxs, streams):
        c.set_stream(st_.cuda_stream)
    ctx, stream = ctxs[0], streams[0]
    scene = api.Scene(ctx, setup.desc)
    sstats = scene.stats()
    rows = api.shard_rows(H, args.band_rows, nshards)
    images = A.IMAGES_FRAMEBUFFER | (A.IMG_BIT(A.IMAGE_HDR) if K > 1 else 0)
    frames = [api.Frame(ctxs[b], W, rows, images) for b in range(nbuf)]
    locals_ = [torch.zeros((rows, W), dtype=torch.int32, device=device) for _ in range(nbuf)]   # RGBA8 framebuffer of this shard
    for fr, lo in zip(frames, locals_):
        fr.bind_external(A.IMAGE_SHADOWED, lo.data_ptr(), lo.numel() * 4)
    frame, local = frames[0], locals_[0]
    gathered = [torch.zeros((world, rows, W), dtype=torch.int32, device=device) for _ in range(nbuf)] if (rank == 0 and dist_on) else None
    fulls = [torch.zeros((H, W), dtype=torch.int32, device=device) for _ in range(nbuf)] if rank == 0 else None
    full = fulls[0] if fulls else None

    def params(collect=0, shard_index=rank, shard_count=nshards, j=0):
        return api.make_params(W, H, spp=S, shadow_rays=args.shadow_rays, images=images, band_rows=args.band_rows, shard_index=shard_index,
                               shard_count=shard_count, accumulate=1 if j > 0 else 0, accumulated_frames=j,
                               collect_stats=collect, pipeline=args.pipeline)

    def render_step(fr, i, plist, asynchronous):
        """one step: K frames (frame = 0..K-1 when accumulating, else frame = i) into `fr`, stream-ordered"""
        for j in range(K):
            api.render(scene, cams.camera(i), cams.info(i, j if K > 1 else None), plist[j], fr, asynchronous=asynchronous)

    # ---- untimed stats pass: exact ray / node / triangle counts of the frames of the timed region -------------------------
    # Ray counts depend on the view, not on the frame number (every hit issues numLights x light-triangles x shadow-rays rays; the
    # seed only moves the sample positions): one counting render per distinct camera of the timed region, summed over the ranks.
    timed = range(args.warmup, args.warmup + args.steps)
    cam_ids = sorted({cams.index(i) for i in timed} | {cams.index(args.warmup + j) for j in range(max(args.isolated_frames, 0))})
    FIELDS = ("numRays", "numPrimaryRays", "numShadowRays", "algorithmicBytes", "shadowTraceBytes", "shadowInnerIterations", "shadowInnerActiveLanes",
              "shadowTriIterations", "shadowTriActiveLanes", "shadowRefills", "shadowTailRays", "numShadowNodeVisits", "numShadowTriTests")
    rows_ = []
    for c in cam_ids:
        api.render(scene, cams.camera(c), cams.info(c), params(collect=1), frame)
        fs = frame.stats()
        rows_.append([float(getattr(fs, f)) for f in FIELDS])
    counts = torch.tensor(rows_, dtype=torch.float64, device=device)
    local_counts = counts.clone()                     # this rank's own shard (rank 0: what its launches' roofline is made of)
    if dist_on:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    by_cam = {c: {f: int(counts[k][n].item()) for n, f in enumerate(FIELDS)} for k, c in enumerate(cam_ids)}
    by_cam_local = {c: {f: int(local_counts[k][n].item()) for n, f in enumerate(FIELDS)} for k, c in enumerate(cam_ids)}
    # a step of K accumulated frames issues K times the rays of one frame
    rays_total = sum(by_cam[cams.index(i)]["numRays"] for i in timed) * K
    primary_total = sum(by_cam[cams.index(i)]["numPrimaryRays"] for i in timed) * K
    rays_per_frame, primary_per_frame = rays_total // max(args.steps, 1), primary_total // max(args.steps, 1)      # means over the timed frames
    alg_bytes_mean = sum(by_cam[cams.index(i)]["algorithmicBytes"] for i in timed) / max(args.steps, 1)
    pipeline_used = fs.pipelineUsed

    def mean_local(f):
        return sum(by_cam_local[cams.index(i)][f] for i in timed) / max(args.steps, 1)
    # rank 0's any-hit launches, per frame, mean over the timed frames
    sched = {"node_loop_trips": mean_local("shadowInnerIterations"), "node_loop_lanes": mean_local("shadowInnerActiveLanes"),
             "triangle_loop_trips": mean_local("shadowTriIterations"), "triangle_loop_lanes": mean_local("shadowTriActiveLanes"),
             "refill_passes": mean_local("shadowRefills"), "tail_rays": mean_local("shadowTailRays"),
             "wide_visits": mean_local("numShadowNodeVisits"), "triangle_tests": mean_local("numShadowTriTests"), "shadow_rays": mean_local("numShadowRays")}
    sched = {k: int(round(v)) for k, v in sched.items()}
    trace_bytes_mean = mean_local("shadowTraceBytes")

    p_run = [params(0, j=j) for j in range(K)]
    kern = {"primary": 0.0, "shadow_gen": 0.0, "shadow_trace": 0.0, "shadow_tail": 0.0, "resolve": 0.0, "n": 0}
    clocks, clock_span = [], []

    works = [None] * nbuf

    class _Done:
        def wait(self):
            return True

    def gather_async(buf):
        """the one exchange step (RCCL over xGMI): every rank's shard -> rank 0, asynchronous w.r.t. the render stream"""
        if args.backend == "gloo":            # rehearsal only: stage through host memory
            streams[buf].synchronize()
            host = locals_[buf].cpu()
            if rank == 0:
                hosts = [torch.empty_like(host) for _ in range(world)]
                dist.gather(host, gather_list=hosts, dst=0)
                gathered[buf].copy_(torch.stack(hosts))
            else:
                dist.gather(host, gather_list=None, dst=0)
            return _Done()
        if rank == 0:
            return dist.gather(locals_[buf], gather_list=[gathered[buf][r] for r in range(world)], dst=0, async_op=True)
        return dist.gather(locals_[buf], gather_list=None, dst=0, async_op=True)

    def finish(buf):
        """stream-level wait for gather `buf`, then (rank 0) de-interleave it into that buffer's full frame"""
        if works[buf] is None:
            return
        works[buf].wait()
        works[buf] = None
        if rank == 0 and dist_on:
            api.deinterleave_bands(ctxs[buf], gathered[buf].data_ptr(), fulls[buf].data_ptr(), W, H, args.band_rows, world)

    last_buf = [0]

    inflight = [False] * nbuf

    # N > 1 over RCCL: the sharded frame comes from librtr_mgpu.so (include/rtr_mgpu.h) — the same C entry points a C++ caller of the
    # library would use: one rank per process (rtr_mgpu_create_rank; the communicator id travels through torch.distributed), nbuf
    # frame slots with a render stream each, a communication stream with grouped ncclSend / ncclRecv to rank 0, k_deinterleave
    use_lib = dist_on and args.backend == "nccl"
    mg = None
    rccl0 = None
    if use_lib:
        if rank == 0:
            start_line("one process per GPU (torch.distributed.run; rtr_mgpu_create_rank: ncclCommInitRank, the id broadcast through torch.distributed)",
                       asked=world, devices_present=torch.cuda.device_count(), local_rank=local_rank)

        def all_ranks_ok(ok):
            """every rank learns whether ALL ranks got through a stage (a rank that failed must not leave the others in a collective)"""
            t_ = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
            dist.all_reduce(t_, op=dist.ReduceOp.MIN)
            return bool(t_.item())

        def create():
            uid = torch.zeros(A.MGPU_ID_BYTES, dtype=torch.uint8, device=device)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(mgpu.MultiGpu.unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, src=0)
            m = mgpu.MultiGpu.rank(local_rank, rank, world, bytes(uid.cpu().numpy().tobytes()), frames_in_flight=nbuf)
            try:
                m.scene_create(setup.desc)
            except RuntimeError:
                m.close()
                raise
            return m
        first_whole = api.Frame(ctx, W, H, images) if rank == 0 else None

        def unsharded(i):
            if rank != 0:
                return None
            render_step(first_whole, i, [params(0, 0, 1, j=j) for j in range(K)], False)
            return first_whole.download()
        p_first = [params(0, j=j) for j in range(K)]
        B_first = max(1, min(plan["frames_per_launch"], nbuf, api.render_batch_limit(scene, p_first[0], setup.num_lights))) if K == 1 else 1
        mg, rccl0 = bring_up(args, create, lambda m: verified_launches(m, A, cams, p_first, K, B_first, unsharded), agree=all_ranks_ok, root=(rank == 0),
                             how="rtr_mgpu_create_rank (ncclCommInitRank)")
        if first_whole is not None:
            first_whole.close()

    def collect(buf):
        """host-side join of the frame that used `buf` (the other frames stay in flight) + its per-launch HIP-event times"""
        if not inflight[buf]:
            return
        frames[buf].wait()
        inflight[buf] = False
        st = frames[buf].stats()
        kern["primary"] += st.primaryMs; kern["shadow_gen"] += st.shadowGenMs
        kern["shadow_trace"] += st.shadowTraceMs; kern["shadow_tail"] += st.shadowTailMs; kern["resolve"] += st.resolveMs; kern["n"] += 1
        if st.shadowTraceClockMHz > 0:
            clocks.append(st.shadowTraceClockMHz); clock_span.append((st.shadowTraceClockMinMHz, st.shadowTraceClockMaxMHz))

    def lib_collect(b):
        if not inflight[b]:
            return
        mg.wait(b)
        inflight[b] = False
        st = mg.frame_stats(b, 0)
        kern["primary"] += st.primaryMs; kern["shadow_gen"] += st.shadowGenMs
        kern["shadow_trace"] += st.shadowTraceMs; kern["shadow_tail"] += st.shadowTailMs; kern["resolve"] += st.resolveMs; kern["n"] += 1
        if st.shadowTraceClockMHz > 0:
            clocks.append(st.shadowTraceClockMHz); clock_span.append((st.shadowTraceClockMinMHz, st.shadowTraceClockMaxMHz))

    # frames per launch: what was asked for, capped by what one launch of this rank's shard can address (rtr_render_batch_limit)
    B = max(1, min(plan["frames_per_launch"], nbuf, api.render_batch_limit(scene, p_run[0], setup.num_lights))) if (K == 1 and (use_lib or not dist_on)) else 1

    groups = max(nbuf // B, 1)
    launch_no = [0]
    marshalled = {}

    def prepare_launches(first, count, launch0):
        """marshal the argument arrays of the launches run_steps(first, count) will make, given the number of the first of them"""
        i, ln = first, launch0
        for c in launch_sizes(count, B):
            g0 = (ln % groups) * B
            key = (i, c, g0)
            if key not in marshalled:
                marshalled[key] = api.marshal_batch([cams.camera(i + j) for j in range(c)], [cams.info(i + j) for j in range(c)], [frames[g0 + j] for j in range(c)])
            i += c; ln += 1

    def step_batch(i0, count):
        """frames i0 .. i0+count-1 in ONE launch of every kernel (rtr_render_batch_async), on the stream of the first one's buffer.
        The frame objects are used in fixed groups of B, so the frame that leads a launch — and owns the launch's scratch — is always
        one of the same few, whatever --warmup and --steps are (a fresh leader would allocate inside the timed region)."""
        g0 = (launch_no[0] % groups) * B
        launch_no[0] += 1
        bufs = [g0 + j for j in range(count)]
        last_buf[0] = bufs[-1]
        if use_lib:                                     # one launch of the pipeline per rank for the batch, then every slot's exchange
            for b in bufs:
                lib_collect(b)
            mg.render_batch_async(bufs, [cams.camera(i0 + j) for j in range(count)], [cams.info(i0 + j) for j in range(count)], p_run[0])
            for b in bufs:
                inflight[b] = True
            return
        with torch.cuda.stream(streams[bufs[0]]):
            for b in bufs:
                collect(b)
            # the cameras / scene infos / frame handles of a launch are INPUTS: marshalled into ctypes arrays ahead of the timed region
            # (prepare_launches below), like the scene — building 20 structs in Python took 0.3 ms of a 41-ms timed region
            key = (i0, count, bufs[0])
            m = marshalled.get(key)
            if m is None:
                m = marshalled[key] = api.marshal_batch([cams.camera(i0 + j) for j in range(count)], [cams.info(i0 + j) for j in range(count)], [frames[b] for b in bufs])
            api.render_batch(scene, None, None, p_run[0], None, marshalled=m)
            for b in bufs:
                inflight[b] = True

    def run_steps(first, count):
        if B == 1:
            for i in range(first, first + count):
                step(i)
            return
        i = first
        for c in launch_sizes(count, B):
            step_batch(i, c)
            i += c

    def step(i):
        b = i % nbuf
        last_buf[0] = b
        if use_lib:
            lib_collect(b)                              # frame i-nbuf: done long ago unless the host runs ahead
            for j in range(K):
                mg.render_async(b, cams.camera(i), cams.info(i, j if K > 1 else None), p_run[j], exchange=(j == K - 1))
            inflight[b] = True
            return
        with torch.cuda.stream(streams[b]):
            collect(b)                                  # frame i-nbuf: done long ago unless the host runs ahead
            finish(b)                                   # its gather (stream-level wait) + de-interleave on rank 0
            render_step(frames[b], i, p_run, True)
            inflight[b] = True
            if dist_on:
                works[b] = gather_async(b)              # RCCL gather to rank 0; runs under the other streams' kernels

    def drain():
        if use_lib:
            for b in range(nbuf):
                lib_collect(b)
            return
        for b in range(nbuf):
            with torch.cuda.stream(streams[b]):
                collect(b)
                finish(b)

    def sync_all():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not warm-up: every frame object allocates its scratch (ray queue, hit records ...) on its first render; with
    # fewer warm-up steps than frames in flight that first render — a few synchronous hipMallocs — would land in the timed region
    for b in range(nbuf):
        render_step(frames[b], 0, p_run, False)
        if use_lib:
            step(b)                                     # the library's slots allocate on their first render too
    drain()
    if B > 1:                                           # the leading frames' scratch grows to the launch on its first batched render
        biggest = max(launch_sizes(args.steps, B) + launch_sizes(args.warmup, B) + [1])
        for _ in range(groups):
            run_steps(0, biggest)
        drain()
    # ---- frames per launch under a latency bound -----------------------------------------------------------------------------
    # A launch of B frames is ready when it ends: the first of its frames is B x ms_per_step old by then, and all B cameras had to be
    # known when it started.  The reference presents ONE frame per loop iteration (application.cppm:352-389,437); a caller that wants
    # the throughput of many frames per launch chooses how stale a frame may be.  Default: one 60-Hz refresh.  Probed here, before the
    # timed region, with single launches of the same frames (max over the ranks); an explicit --batch is taken as it is.
    latency = {"limit_ms": args.max_latency_ms or None, "probe_margin": PROBE_MARGIN, "probes": [], "chosen_by": "--batch" if args.batch else ("default" if not args.max_latency_ms else "probe")}
    if args.max_latency_ms > 0 and not args.batch and B > 1:
        cap = B

        def probe(c):
            nonlocal B, groups
            B, groups = c, 1                             # always the same leading frame: its scratch grows once, in the untimed first launch
            best = None
            for k in range(4):
                sync_all()
                t = time.perf_counter()
                run_steps(0, c)
                drain()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t) * 1e3
                if dist_on:
                    tt = torch.tensor([dt], dtype=torch.float64, device=device)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt = float(tt.item())
                if k:
                    best = dt if best is None else min(best, dt)
            latency["probes"].append({"frames": c, "launch_ms": round(best, 4)})
            return best
        lif = launches_in_flight(world if dist_on else 1, max(nbuf // 2, 1))      # N > 1: two launches enqueued, so a single launch may take half the bound
        c = pick_frames_per_launch(args.max_latency_ms * PROBE_MARGIN / lif, max(1, min(cap, nbuf // lif)), probe)
        B, groups = c, launches_in_flight(world if dist_on else 1, max(nbuf // c, 1))
        nbuf = B * groups                               # the frame objects of the launches in flight (one GPU: one launch at a time)
        for _ in range(groups):                         # the frames that lead launches of this size grow their scratch now, not in the timed region
            run_steps(0, B)
        drain()
    run_steps(0, args.warmup)
    drain()
    kern.update({"primary": 0.0, "shadow_gen": 0.0, "shadow_trace": 0.0, "shadow_tail": 0.0, "resolve": 0.0, "n": 0})
    del clocks[:], clock_span[:]
    if B > 1 and not use_lib:
        prepare_launches(args.warmup, args.steps, launch_no[0])
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    drain()
    torch.cuda.synchronize()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # After the timed region: the same frames ONE AT A TIME, so every kernel has the GPU to itself.  With several frames in
    # flight a launch's HIP-event bracket also contains the time it queued behind / shared CUs with the neighbouring frames'
    # kernels (k_shadow_trace4: 2.7 ms bracket, 2.1 ms dispatch begin->end in rocprof, 1.86 ms alone), so the per-kernel cost
    # and the roofline are taken from this pass; rocprofv3 of `--frames-in-flight 1` reproduces it (profiles/).
    kern_iso, iso_ms_per_frame, clocks_iso = None, None, []
    if args.isolated_frames > 0:
        kern_iso = {"primary": 0.0, "shadow_gen": 0.0, "shadow_trace": 0.0, "shadow_tail": 0.0, "resolve": 0.0}
        clocks_iso = []
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for j in range(args.isolated_frames):
            render_step(frames[0], args.warmup + j, p_run, False)
            st = frames[0].stats()
            kern_iso["primary"] += st.primaryMs / args.isolated_frames; kern_iso["shadow_gen"] += st.shadowGenMs / args.isolated_frames
            kern_iso["shadow_trace"] += st.shadowTraceMs / args.isolated_frames; kern_iso["resolve"] += st.resolveMs / args.isolated_frames
            kern_iso["shadow_tail"] += st.shadowTailMs / args.isolated_frames
            if st.shadowTraceClockMHz > 0:
                clocks_iso.append(st.shadowTraceClockMHz)
        iso_ms_per_frame = (time.perf_counter() - t1) * 1e3 / args.isolated_frames
        iso_rays = sum(by_cam_local[cams.index(args.warmup + j)]["numRays"] for j in range(args.isolated_frames))

    # The frame the reference presents, end to end, on the same scene (one at a time: rtr_denoise_combine is synchronous)
    presented = None
    if rank == 0 and not dist_on and not emu and not args.obj and args.present_frames > 0:
        psetup = getattr(scenes, WORKLOADS[args.workload])(W, H, ltc=scenes.shipped_ltc())
        pscene = api.Scene(ctx, psetup.desc)
        pframe = api.Frame(ctx, W, H, 0xff)
        pp = api.make_params(W, H, spp=4, shadow_rays=args.shadow_rays, images=A.IMAGES_RAYGEN5, pipeline=args.pipeline)
        acc, pk = {"kernels": 0.0}, {}
        for j in range(2 + args.present_frames):
            if j == 2:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            api.render(pscene, psetup.camera, psetup.scene_info(j), pp if j else api.make_params(W, H, spp=4, shadow_rays=args.shadow_rays, images=A.IMAGES_RAYGEN5,
                                                                                              pipeline=args.pipeline, collect_stats=1), pframe)
            if j == 0:
                prays = int(pframe.stats().numRays)
            if j >= 2:
                st = pframe.stats()
                acc["kernels"] += st.totalMs
                for k, v in (("primary", st.primaryMs), ("shadow_gen", st.shadowGenMs), ("shadow_trace", st.shadowTraceMs), ("resolve", st.resolveMs)):
                    pk[k] = pk.get(k, 0.0) + v / args.present_frames
            pframe.denoise_combine(4)
        pms = (time.perf_counter() - t1) * 1e3 / args.present_frames
        ray_ms = sum(acc.values()) / args.present_frames
        presented = {"ms_per_frame": round(pms, 4), "frames": args.present_frames, "spp": 4, "images": 5, "denoise_iterations": 4, "combine": True,
                     "ltc_tables": "shipped (realtimeraytracer_amd/data/ltc_tables.bin)", "ray_gen_kernels_ms": round(ray_ms, 4),
                     "kernels_ms": {k: round(v, 4) for k, v in pk.items()},
                     "denoise_combine_and_host_ms": round(pms - ray_ms, 4), "rays_per_frame": prays,
                     "scope": "one frame at a time, wall clock around rtr_render + rtr_denoise_combine; FINAL left in HBM"}
        pframe.close(); pscene.close()

    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    mrays = rays_total / elapsed / 1e6

    out = None
    if rank == 0:
        n = max(kern["n"], 1)
        bracket_ms = kern["shadow_trace"] / n                      # in-region HIP-event bracket, per frame
        # With B frames per launch and no more frame objects than one launch takes, ONE launch of every kernel is on the GPU at a time:
        # the HIP-event brackets of the TIMED REGION are the kernels' own durations, and the roofline is that of the launches the
        # headline number is made of — all of them, summed: busy cycles of the region's any-hit launches over their cycles.  Otherwise
        # launches of different frames overlap and the kernel's own duration comes from the one-frame-at-a-time pass after the timed region.
        sizes = launch_sizes(args.steps, B)
        own_launch = B > 1 and nbuf == B and not dist_on          # one launch at a time
        n_launches = len(sizes) if own_launch else 1
        launch_frames = (args.steps / n_launches) if own_launch else 1            # mean frames per launch of the launches described
        trace_ms_total = kern["shadow_trace"] if own_launch else (kern_iso["shadow_trace"] if kern_iso else bracket_ms)      # all the launches described, summed
        trace_ms = trace_ms_total / n_launches                                    # mean per launch
        trace_rays = sched["shadow_rays"] * (args.steps if own_launch else 1)     # shadow rays those launches traced (counting form, mean of the timed frames)
        trace_bytes = trace_bytes_mean * (args.steps if own_launch else 1)
        roofline, roofline2, frame_hbm = None, None, None
        if pipeline_used == 2 and trace_ms > 0:
            rev = A.hip_lib().rtr_kernel_revision().decode()
            pmc, pmc_note, pmc_key, pmc_file = None, None, None, None
            # The counter totals are NOT measured by this process (rocprofv3 has to wrap it): they are committed passes of this command
            # (profiles/pmc_r05.sh -> profiles/r05/pmc_roofline.json).  Used when workload, kernel revision and triangle count match this
            # run; among those, the passes whose launches are closest in length (same camera mode first), and the totals are SCALED by
            # rays traced (`pmc_scaled_by`: 1.0 = the committed passes are of exactly these launches) — counter totals of this kernel are
            # proportional to the rays it walks to within a per cent across launch lengths (profiles/r03/pmc_roofline.json: 33.9 - 34.4
            # vector instructions per ray from 1 to 32 frames per launch).
            base_key = f"{args.workload}_{W}x{H}_spp{S}_gpus{world}"
            for tpath in (os.path.join(ROOT, "profiles", "r05", "pmc_roofline.json"), os.path.join(ROOT, "profiles", "r04", "pmc_roofline.json")):
                try:
                    table = json.load(open(tpath))
                except Exception as e:      # noqa: BLE001
                    pmc_note = f"{tpath}: {e}"
                    continue
                cands = []
                for key, v in table.items():
                    if not isinstance(v, dict) or not (key == base_key or key.startswith(base_key + "_")):
                        continue
                    if v.get("kernel_revision") != rev or v.get("triangles") != int(sstats.numTriangles) or not v.get("rays_per_launch"):
                        continue
                    fpl = v.get("frames_per_launch") or (int(key.split("_batch")[1].split("_")[0]) if "_batch" in key else 1)
                    cands.append((0 if v.get("camera", "static") == cams.mode else 1, abs(fpl - launch_frames), key, v))
                if cands:
                    cands.sort(key=lambda t: t[:3])
                    pmc_key, pmc, pmc_file = cands[0][2], cands[0][3], os.path.relpath(tpath, ROOT)
                    break
                pmc_note = f"no committed counter passes of revision {rev} for {base_key} ({sstats.numTriangles} triangles)"
            scale = (trace_rays / n_launches) / pmc["rays_per_launch"] if pmc else None      # this run's mean launch over the committed one, in rays

            def per_launch(name):
                return pmc[name] * scale if pmc else None
            ck = clocks if own_launch else (clocks_iso if (kern_iso and clocks_iso) else clocks)
            clock_mhz = sorted(ck)[len(ck) // 2] if ck else None
            num_simds = 4 * torch.cuda.get_device_properties(device).multi_processor_count
            launch_cycles = trace_ms * 1e-3 * clock_mhz * 1e6 if clock_mhz else None
            busy = per_launch("SQ_ACTIVE_INST_VALU_quad") * 4 / num_simds if pmc else None
            traffic = per_launch("hbm_bytes_per_launch")
            frac = busy / launch_cycles if (busy and launch_cycles) else None
            lane_all = pmc["derived"]["valu_lane_utilisation"] if pmc else None
            # what the 4 cycles per instruction of the counter's convention are worth for THIS kernel's visit: a microbenchmark that issues
            # the visit's instruction mix as independent streams at eight waves per SIMD (profiles/microbench/visit_mix.hip, committed result)
            mix = None
            try:
                mix = json.load(open(os.path.join(ROOT, "profiles", "r04", "visit_mix_issue.json")))
            except Exception:      # noqa: BLE001
                pass
            roofline = {
                # the ceiling that binds: one SIMD issues one vector instruction at a time; SQ_ACTIVE_INST_VALU counts, in units of 4
                # cycles, the time SIMDs spent issuing them
                "bound": "valu_issue",
                "kernel": "k_shadow_trace4<16, true, false>: any-hit traversal of the shadow-ray queue, revision " + rev,
                "achieved": round(busy, 1) if busy else None, "peak": round(launch_cycles, 1) if launch_cycles else None,
                "unit": "SIMD cycles per launch (achieved: issuing vector instructions = SQ_ACTIVE_INST_VALU x 4 / SIMDs; peak: cycles of the launch); mean over the launches of the timed region",
                "frac": round(frac, 4) if frac else None,
                # the part of it that can still move: issue slots whose lanes did work (frac x the hardware's lane utilisation over all vector instructions)
                "frac_useful": round(frac * lane_all, 4) if (frac and lane_all) else None,
                "issue_cycles_per_inst_measured": mix,
                # the same instruction count priced at the visit mix's MEASURED issue cost instead of the counter's flat 4 cycles (the node
                # loop's mix; the triangle test holds more plain fma / mul, which issue at 2: an estimate, a little high)
                "frac_at_measured_issue_cost": round(per_launch("SQ_INSTS_VALU") * mix["cycles_per_valu_inst_per_simd_8_waves"] / num_simds / launch_cycles, 4)
                if (pmc and mix and launch_cycles and mix.get("cycles_per_valu_inst_per_simd_8_waves")) else None,
                "avg_launch_ms": round(trace_ms, 4), "launches": n_launches,
                "frames_per_launch": round(launch_frames, 2), "avg_ms_per_frame": round(trace_ms / launch_frames, 4),
                "avg_launch_ms_source": ("HIP events on the launch stream over the TIMED REGION: one launch of every kernel at a time, "
                                         f"launches of {sorted(set(sizes))} frames" if own_launch else
                                         (f"HIP events on the launch stream, {args.isolated_frames} frames rendered one at a time right after the timed region"
                                          if kern_iso else "HIP events on the launch stream over the timed region")),
                "one_frame_launch_ms": round(kern_iso["shadow_trace"], 4) if kern_iso else None,
                "in_flight_event_bracket_ms": round(bracket_ms, 4) if (kern_iso and not own_launch) else None,
                "clock_mhz": round(clock_mhz, 1) if clock_mhz else None,
                "clock_source": "s_memtime / s_memrealtime stamps around the launch's persistent loop (lane 0 of the first workgroup of each XCD; mean over the XCDs, median over the launches), same launches as avg_launch_ms",
                # the XCDs clock independently: slowest and fastest of them over the timed launches
                "clock_mhz_xcd_min_max": [round(min(a for a, _ in clock_span), 1), round(max(b for _, b in clock_span), 1)] if (clock_span and own_launch) else None,
                # SQ_ACTIVE_INST_VALU charges every vector instruction four cycles; simple ones issue faster, so on a part that clocks a few
                # per cent lower than the one the counters were collected on the ratio can come out above 1
                "frac_note": "achieved = committed counter passes (SQ_ACTIVE_INST_VALU x 4 cycles per instruction, an upper estimate for simple instructions), scaled by rays; peak = this run's launch time x this run's mean shader clock",
                "simds": num_simds,
                "valu_wave_insts_per_launch": int(per_launch("SQ_INSTS_VALU")) if pmc else None,
                "valu_wave_insts_per_ray": round(pmc["SQ_INSTS_VALU"] / pmc["rays_per_launch"], 2) if pmc else None,
                # dead lanes: what fraction of the lanes did work in the trips of the kernel's two loops (counting form of the same kernel,
                # run in this process), and what the hardware says for all vector instructions (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU))
                "lane_util": {"node_loop": round(sched["node_loop_lanes"] / max(64 * sched["node_loop_trips"], 1), 4),
                              "triangle_loop": round(sched["triangle_loop_lanes"] / max(64 * sched["triangle_loop_trips"], 1), 4),
                              "all_vector_instructions": round(lane_all, 4) if lane_all else None},
                "per_ray": {"wide_node_visits": round(sched["wide_visits"] / max(sched["shadow_rays"], 1), 3),
                            "triangle_tests": round(sched["triangle_tests"] / max(sched["shadow_rays"], 1), 3)},
                "schedule_per_frame": sched,
                # the memory side, for the record: HBM traffic from the FETCH_SIZE / WRITE_SIZE passes (gfx950 corrections in the json),
                # L1 -> L2 read requests x 64 B, L1 tag look-ups per L1 per clock
                "traffic": int(traffic) if traffic else None,
                "hbm_frac": round(traffic / (trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "l2_frac": round(per_launch("TCP_TCC_READ_REQ") * 64 / (trace_ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4) if pmc else None,
                # the second unit that is nearly full: one L1 (TCP) per CU, one tag look-up per clock
                "l1_tag_lookups_per_l1_clock": round(per_launch("TCP_TOTAL_CACHE_ACCESSES") / (num_simds / 4 * launch_cycles), 4) if (pmc and launch_cycles) else None,
                # `frac` says how busy the vector pipes are, not what the next instruction costs (measured, round 5)
                "margin_note": "7.8 % fewer vector instructions (the IEEE reciprocal sequences as v_rcp_f32 + one Newton step) leave this kernel's duration unchanged "
                               "(profiles/r05/pmc_rcp_instr.log); its time follows the record VISITS - dependent fetches through L1s at l1_tag_lookups_per_l1_clock - "
                               "with the vector pipes `frac` busy beside them (DESIGN.md 5)",
                # algorithmic bytes of the same kernel (its counting form): 64 B per 4-wide record visited + 48 B per triangle test + 37 B per
                # ray.  Served by LDS / L1 / L2 / Infinity Cache: this rate is NOT a fraction of any ceiling and is not the roofline
                "algorithmic_bytes_per_launch": int(trace_bytes / n_launches),
                "algorithmic_gbps": round(trace_bytes / (trace_ms_total * 1e-3) / 1e9, 1),
                "algorithmic_over_hbm_peak": round(trace_bytes / (trace_ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "pmc_source": pmc_file, "pmc_key": pmc_key, "pmc_scaled_by": round(scale, 4) if scale else None, "pmc_note": pmc_note if not pmc else None,
                "counters": ("achieved, traffic, l2_frac, l1_tag_lookups and valu_* are DERIVED FROM COMMITTED rocprofv3 counter passes of this command "
                             "(checked against this run: workload, kernel revision, triangles; scaled by rays traced, pmc_scaled_by); avg_launch_ms, clock_mhz, lane_util "
                             "node / triangle loop, per_ray, schedule and algorithmic_* are measured by this process") if pmc else None,
                "layout": {"bvh": int(sstats.bvhLayoutVersion), "wide": int(sstats.wideLayoutVersion)}}
            if pmc and pmc.get("kernels", {}).get("k_shadow_gen_oct") and (kern_iso or own_launch):
                # the one kernel of the frame that IS bound by HBM: it writes the ray queue (20 B per ray + 16 B per pixel-sample) as fast as the memory takes it
                g = pmc["kernels"]["k_shadow_gen_oct"]
                gen_ms = (kern["shadow_gen"] / n_launches) if own_launch else kern_iso["shadow_gen"]
                gbytes = (g["read_bytes"] + g["write_bytes"]) * scale
                # 20-B ray records + 16-B origins out, 20-B hit records in: what the kernel HAS to move (achieved = algorithmic bytes over
                # the launch's duration); `traffic` = what the FETCH_SIZE / WRITE_SIZE passes saw (more: object / vertex / light fetches that
                # missed the L2s, and — capped at 64 VGPRs for eight waves per SIMD — 13 spilled dwords per lane on their way through the memory side)
                galg = (20 * sched["shadow_rays"] + (20 + 16) * primary_per_frame / max(K, 1) / max(world, 1)) * launch_frames
                roofline2 = {"bound": "hbm", "kernel": "k_shadow_gen_oct: shadow-ray generation into the queue binned by direction octant",
                             "achieved": round(galg / (gen_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(galg / (gen_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "frac_of_measured_copy_rate": round(galg / (gen_ms * 1e-3) / 1e9 / HBM_MEASURED_COPY_GBS, 4),
                             "traffic": int(gbytes), "traffic_gbps": round(gbytes / (gen_ms * 1e-3) / 1e9, 1), "traffic_over_algorithmic": round(gbytes / galg, 3),
                             "read_bytes": int(g["read_bytes"] * scale), "write_bytes": int(g["write_bytes"] * scale),
                             "algorithmic_bytes_per_launch": int(galg),
                             "frames_per_launch": round(launch_frames, 2),
                             "avg_launch_ms": round(gen_ms, 4), "avg_launch_ms_source": "HIP events on the launch stream, " + ("timed region (one launch at a time)" if own_launch else "frames rendered one at a time (this run)"),
                             "counters": "traffic: bytes from the committed FETCH_SIZE / WRITE_SIZE passes (separate --pmc passes; FETCH_SIZE's streamed part doubled, gfx950), scaled by rays; achieved: this run's algorithmic bytes over this run's duration"}
                fb = pmc.get("frame_hbm_bytes")
                fb = fb * scale / launch_frames if fb else fb          # the counters are per launch
                frame_hbm = {"bytes_per_frame": int(fb) if fb else None, "per_kernel_per_launch": {kk: int((vv["read_bytes"] + vv["write_bytes"]) * scale) for kk, vv in pmc["kernels"].items()},
                             "gbps_at_ms_per_step": round(fb / (ms_per_step * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(fb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "scope": "sum of the frame's four kernels' HBM bytes (committed counter passes, scaled by rays) over this run's frame time"} if fb else None
        elif trace_ms == 0 and kern["primary"] > 0:
            mk_ms = kern["primary"] / n
            achieved = alg_bytes_mean / (mk_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "k_megakernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                        "algorithmic_bytes_per_launch": int(alg_bytes_mean), "avg_launch_ms": round(mk_ms, 4),
                        "bvh_layout_version": int(sstats.bvhLayoutVersion)}
        out = {
            "metric": "Mrays/sec at 1920x1080 1spp (all rays: primary + shadow)" if (W, H, S, K) == (1920, 1080, 1, 1) else f"Mrays/sec at {W}x{H} {S}spp" + (f" x{K} frames accumulated in HDR" if K > 1 else ""),
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "ms_per_frame": round(ms_per_step / K, 4), "higher_is_better": True,
            # a launch renders frames_per_launch frames and they are ready together when it ends: the age of the first of them, and how
            # far ahead the cameras had to be known (one_frame_at_a_time below: the same frames with one frame per launch)
            **latency_fields(B, groups, args.steps, ms_per_step, latency),
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": ("file " + args.obj) if args.obj else "synthetic",
            "config": {"workload": f"{args.workload} {W}x{H} {S}spp{' x%d accumulated frames per step' % K if K > 1 else ''}, {sstats.numTriangles} triangles, {setup.num_lights} area lights, "
                                   f"{args.shadow_rays} shadow rays/light-triangle, band-sharded x{world}",
                       "camera": ("a new view every frame: scripted walk (W / S held, cursor drifting; applyInput = Window::processInput), period %d frames" % PATH_PERIOD) if cams.mode == "path" else "static: every frame the same view",
                       "rays_per_frame": rays_per_frame, "primary_rays_per_frame": primary_per_frame, "rays_per_frame_is": "mean over the frames of the timed region (the counting form, one pass per view)",
                       "pipeline": "wavefront" if pipeline_used == 2 else "megakernel",
                       # how the shadow rays are walked and the frame resolved (tunables of the context; every setting renders the same bytes): a ray
                       # counts as a ray whether its walk starts at the root or at the leaf of the triangle it comes from
                       "shadow_walk": {"own_leaf_start": bool(ctx.get_tunable("trace_own_leaf")), "queue_binned_by_octant": int(ctx.get_tunable("trace_binned")),
                                       "child_entered_first": "nearest entry" if A.hip_lib().rtr_kernel_revision().decode().endswith("nearest-first") else "farthest exit",
                                       "resolve_compact": bool(ctx.get_tunable("resolve_compact"))},
                       "bvh": {"nodes": int(sstats.numNodes), "max_depth": int(sstats.maxDepth), "lds_stack_entries": int(sstats.stackEntries),
                               "build_ms": round(float(sstats.buildMs), 1)}},
            "primary_mrays_per_s": round(primary_total / elapsed / 1e6, 2),
            "frames_in_flight": nbuf,
            "kernels_ms": {k: round(v, 4) for k, v in kern_iso.items()} if kern_iso else {k: round(v / n, 4) for k, v in kern.items() if k != "n"},
            "kernels_ms_in_flight_event_brackets": {k: round(v / n, 4) for k, v in kern.items() if k != "n"} if kern_iso else None,
            "one_frame_at_a_time": {"ms_per_step": round(iso_ms_per_frame, 4), "mrays_per_s": round(iso_rays / args.isolated_frames * K / iso_ms_per_frame / 1e3, 2),
                                    "frames": args.isolated_frames, "frame_latency_ms": round(iso_ms_per_frame, 4),
                                    "scope": "rank 0's shard, no gather; one launch per frame, joined before the next (the reference's loop: application.cppm:352-389)"} if iso_ms_per_frame else None,
            "algorithmic_gbps_all_kernels": round(alg_bytes_mean * K / (ms_per_step * 1e-3) / 1e9, 2),
            "roofline": roofline,
            "roofline_secondary": roofline2,
            "frame_hbm": frame_hbm,
            "presented_frame": presented,
        }

    # ---- reported CPU baseline (rank 0, N=1 only): the oracle on a bounded sample --------------------------
    if rank == 0 and emu:
        out["emulated_rank_of"] = emu
        out["config"]["workload"] += f" [single-GPU rehearsal of ONE rank of {emu}: shard 0 only, no gather; value = this rank's rays/s]"
    if rank == 0 and not dist_on and not emu and not args.no_cpu_baseline:
        from oracle import oracle_py as O
        threads = min(os.cpu_count() or 1, 16)
        bvh = scene.export_bvh()
        # bounded sample: the whole frame, twice (best of two) — about 20-30 core-seconds of oracle work on the 16 host cores
        sample_shards = 1
        pc = api.make_params(W, H, spp=S, shadow_rays=args.shadow_rays, band_rows=args.band_rows, shard_index=0,
                             shard_count=sample_shards, collect_stats=1)
        best = None
        for _ in range(2):
            t1 = time.perf_counter()
            r = O.render(setup.desc, setup.camera, setup.scene_info(0), pc, bvh=bvh, threads=threads)
            dt = time.perf_counter() - t1
            best = dt if best is None else min(best, dt)
        dt = best
        # (i) of SURVEY 8d: one thread, on shard 0 of 16 of the same frame (every 16th band: ~1/16 of the rays)
        p1 = api.make_params(W, H, spp=S, shadow_rays=args.shadow_rays, band_rows=args.band_rows, shard_index=0, shard_count=16, collect_stats=1)
        t1 = time.perf_counter()
        r1 = O.render(setup.desc, setup.camera, setup.scene_info(0), p1, bvh=bvh, threads=1)
        dt1 = time.perf_counter() - t1
        single = {"value": round(r1.stats.numRays / dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                  "sample": f"shard 0 of 16 of the same frame ({r1.stats.numRays} rays, {dt1:.2f} s)"}
        out["cpu_baseline"] = {"value": round(r.stats.numRays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                               "sample": f"the same frame, all {r.stats.numRays} rays, best of 2 runs ({dt:.2f} s each), scalar C++ oracle (oracle/), "
                                         f"{threads} std::threads over rows, -O2 -ffp-contract=off",
                               "single_thread": single}
        if args.verify:
            api.render(scene, setup.camera, setup.scene_info(0), params(0), frames[0])     # frame 0 again, the one the oracle rendered
            gpu = locals_[0].cpu().numpy().view(np.uint32)
            bad = int((gpu != r.images[A.IMAGE_SHADOWED]).sum())
            out["verify"] = {"pixels_checked": int(gpu.size), "pixels_differing_vs_oracle": bad}
    elif rank == 0 and not dist_on:
        out["cpu_baseline"] = None
    elif rank == 0:
        out["cpu_baseline"] = None
        # N > 1: the assembled frame of the last step must equal the same frame rendered unsharded on this GPU
        last_i = args.warmup + args.steps - 1
        whole = api.Frame(ctx, W, H, images)
        render_step(whole, last_i, [params(0, 0, 1, j=j) for j in range(K)], False)
        torch.cuda.synchronize()
        assembled = mg.download(last_buf[0]) if use_lib else fulls[last_buf[0]].cpu().numpy().view(np.uint32)
        bad = int((assembled != whole.download()).sum())
        out["verify"] = {"assembled_vs_unsharded_pixels_differing": bad, "frame": last_i,
                         "gather": "librtr_mgpu.so: grouped ncclSend / ncclRecv to rank 0 on a communication stream + k_deinterleave" if use_lib else "torch.distributed gather (rehearsal backend)"}
        if use_lib:
            out["rccl"] = {"nranks": int(mg.info.nranks), "nlocal": int(mg.info.nlocal), "version": int(mg.info.rcclVersion),
                           "first_exchange_verified": rccl0["first_exchange_verified"], "first_exchange_ms": rccl0["first_exchange_ms"],
                           "first_batch_frames": rccl0["first_batch_frames"], "first_batch_ms": rccl0["first_batch_ms"], "group_per_slot": rccl0["group_per_slot"],
                           "launch": "one process per GPU under torch.distributed.run, rtr_mgpu_create_rank (ncclCommInitRank; the id travels through a broadcast)",
                           "exchange": "grouped ncclSend / ncclRecv to rank 0 on a communication stream + k_deinterleave (librtr_mgpu.so, plan = rtr_mgpu_plan)",
                           "env": plan["env_defaults"]}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if mg is not None:
        mg.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
