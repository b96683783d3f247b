"""bench.py is what the driver runs: its one JSON line must keep the contract (metric / value / roofline / cpu_baseline ...), the
presented-frame object, the real-asset option and the N > 1 code path (one rank through librtr_mgpu.so and real RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from realtimeraytracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, env=None, timeout=600):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def test_bench_line_keeps_the_contract(scene_cache):
    d = _bench(["--steps", "12", "--warmup", "2", "--width", "640", "--height", "360", "--present-frames", "2", "--verify"],
               env={"RTR_SCENE_CACHE": str(scene_cache)})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "presented_frame"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 100 and abs(d["ms_per_step"] * d["value"] * 1e3 / d["config"]["rays_per_frame"] - 1.0) < 0.02
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "clock_mhz", "lane_util", "per_ray", "kernel"):
        assert k in r, k
    assert r["bound"] == "valu_issue" and r["avg_launch_ms"] > 0 and 1500 < r["clock_mhz"] < 2600
    # up to thirty-two frames per launch of every kernel by default, a run cut into equal launches; the roofline is that of the launches of the timed region
    # the line describes the launches it TIMED: twelve steps are one launch of twelve frames, though thirty-two would have fitted the bound
    assert d["frames_per_launch"] == 12 and d["latency"]["frames_per_launch_limit"] == 32 and d["frames_in_flight"] == 32 and d["timed_launches"] == [12] and r["frames_per_launch"] == 12, (d["frames_per_launch"], d["latency"])
    assert d["latency"]["launches_in_flight"] == 1
    assert abs(r["avg_ms_per_frame"] * 12 - r["avg_launch_ms"]) < 1e-3 and r["one_frame_launch_ms"] > 0
    # a frame time beside the rate: the launch's duration is the age of its first frame; the launch size was chosen under the latency bound
    # (a 640x360 frame: thirty-two of them fit one 60-Hz refresh), every frame of the timed region a new view along the scripted walk
    # (both figures are rounded to four decimals in the line: the product of the rounded factor may be off by frames x 0.00005)
    assert abs(d["frame_latency_ms"] - d["frames_per_launch"] * d["ms_per_step"]) < d["frames_per_launch"] * 6e-5 + 1e-4
    assert d["latency"]["limit_ms"] == 16.7 and d["latency"]["chosen_by"] == "probe"
    assert all(p_["launch_ms"] > 0 for p_ in d["latency"]["probes"]) and d["latency"]["probes"][-1]["frames"] == d["latency"]["frames_per_launch_limit"]
    assert d["latency"]["probes"][-1]["launch_ms"] <= 16.7 and "scripted walk" in d["config"]["camera"]
    assert d["config"]["shadow_walk"] == {"own_leaf_start": True, "queue_binned_by_octant": 2, "child_entered_first": "farthest exit", "resolve_compact": True}
    assert d["one_frame_at_a_time"]["frame_latency_ms"] == d["one_frame_at_a_time"]["ms_per_step"] > 0
    assert "frac_useful" in r and "issue_cycles_per_inst_measured" in r
    assert r["frac"] is None or 0 < r["frac"] <= 1.05          # counters are committed for the default workload only (pmc_note says so otherwise)
    assert 0 < r["lane_util"]["node_loop"] <= 1 and 0 < r["lane_util"]["triangle_loop"] <= 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and c["sample"] and c["single_thread"]["cores"] == 1
    assert d["verify"]["pixels_differing_vs_oracle"] == 0
    p = d["presented_frame"]
    assert p["spp"] == 4 and p["images"] == 5 and p["denoise_iterations"] == 4 and p["ms_per_frame"] > 0


def test_bench_renders_a_real_asset(scene_cache):
    obj, _ = scenes.write_cornell(scene_cache)
    d = _bench(["--obj", obj, "--obj-view", "0.0,0.0,-1.9", "--steps", "6", "--warmup", "2", "--width", "320", "--height", "184", "--no-cpu-baseline", "--isolated-frames", "2"])
    assert d["data"].startswith("file ") and d["config"]["workload"].startswith("obj:") and d["value"] > 0
    assert d["config"]["primary_rays_per_frame"] == 320 * 184 and d["config"]["rays_per_frame"] > 320 * 184       # something was hit and lit


def test_bench_multi_rank_path_with_one_rank_through_rccl(scene_cache):
    d = _bench(["--steps", "10", "--warmup", "2", "--width", "640", "--height", "360", "--isolated-frames", "2"],
               env={"RTR_BENCH_FORCE_DIST": "1", "RTR_SCENE_CACHE": str(scene_cache), "MASTER_PORT": "29577"})
    assert d["verify"]["assembled_vs_unsharded_pixels_differing"] == 0 and "librtr_mgpu.so" in d["verify"]["gather"]
    assert d["cpu_baseline"] is None and d["presented_frame"] is None and d["value"] > 0
    assert d["rccl"]["nranks"] == 1 and d["rccl"]["nlocal"] == 1 and d["rccl"]["version"] > 20000 and "rtr_mgpu_create_rank" in d["rccl"]["launch"]
    # the one-process-per-GPU start goes through the same bring-up: communicator, then verified first launches, before anything is timed
    assert d["rccl"]["first_exchange_verified"] is True and d["rccl"]["group_per_slot"] is False and d["rccl"]["first_batch_frames"] > 1


def test_bench_in_process_multi_gpu_path_with_one_rank(scene_cache):
    """`python bench.py --gpus N` started plainly drives the N devices from one process through rtr_mgpu_create; here that code path
    with N = 1 (RTR_BENCH_FORCE_INPROC) and the shard sent to itself through RCCL."""
    d = _bench(["--steps", "10", "--warmup", "2", "--width", "640", "--height", "360"],
               env={"RTR_BENCH_FORCE_INPROC": "1", "RTR_MGPU_SELF_EXCHANGE": "1", "RTR_SCENE_CACHE": str(scene_cache)})
    assert d["verify"]["assembled_vs_unsharded_pixels_differing"] == 0 and d["n_gpus"] == 1 and d["value"] > 0
    assert d["rccl"]["nranks"] == 1 and d["rccl"]["nlocal"] == 1 and "rtr_mgpu_create " in d["rccl"]["launch"] + " "
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k


def test_bench_more_gpus_than_present_fails_from_the_library(scene_cache):
    """`python bench.py --gpus N` with N > the devices of the box: rc != 0 and the library's own message, not a launcher's."""
    import torch
    have = torch.cuda.device_count()
    want = have + 1
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RTR_SCENE_CACHE=str(scene_cache))
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "2", "--warmup", "1"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert f"{want} devices requested, {have} present" in r.stderr + r.stdout, r.stderr[-2000:]


def test_bench_gpus_4_started_plainly_with_the_ranks_sharing_one_gpu(scene_cache):
    """`python bench.py --gpus 4` the way the driver starts it — no launcher, one process, rtr_mgpu_create — carried out for real on a
    one-GPU box: the four ranks share device 0 (the library's test hook) and tests/fake_rccl stands in for RCCL.  The line is marked
    as a rehearsal (its rate is one GPU's, time-sliced), carries what the communicator looked like and the ranks' host time, and the
    assembled frame of the last step equals the unsharded one."""
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    assert os.path.exists(fake), "run __graft_entry__.build()"
    d = _bench(["--gpus", "4", "--steps", "24", "--warmup", "8", "--width", "640", "--height", "360"],
               env={"RTR_SCENE_CACHE": str(scene_cache), "LD_PRELOAD": fake, "RTR_MGPU_TEST_SHARED_DEVICE": "1"})
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["value"] > 100 and d["rehearsal"] and "share ONE GPU" in d["rehearsal"]
    r = d["rccl"]
    assert r["nranks"] == 4 and r["nlocal"] == 4 and r["version"] == 99999 and "rtr_mgpu_create" in r["launch"]
    assert 0 < r["host_enqueue_ms_per_frame"] < 5 and 0 <= r["of_which_inside_rccl_calls"] <= r["host_enqueue_ms_per_frame"]
    assert d["verify"]["assembled_vs_unsharded_pixels_differing"] == 0
    # N > 1 under the latency bound: two launches in flight, each allowed half the bound; a frame is two launches old when it lands
    # the first launch was a verified one (assembled frame against rank 0's unsharded render of the same camera), before anything was timed
    assert r["first_exchange_verified"] is True and r["first_exchange_ms"] > 0
    assert d["frames_in_flight"] == 32 and d["latency"]["frames_per_launch_limit"] == 16 and d["frames_per_launch"] == 12 and d["timed_launches"] == [12, 12]
    assert d["latency"]["launches_in_flight"] == 2 and abs(d["frame_latency_ms"] - 2 * 12 * d["ms_per_step"]) < 2e-3
    assert all(p_["launch_ms"] <= 16.7 / 2 for p_ in d["latency"]["probes"][-1:])
    assert "band-sharded x4" in d["config"]["workload"]



def test_bench_gpus_n_start_is_verified_and_a_wrong_exchange_stops_the_run(scene_cache):
    """first contact with RCCL at N > 1 must not be a silent failure.  Before any render bench.py --gpus N says
    (stderr) how many devices there are, which RCCL answered and how large the communicator is; its first launches — one frame, then one
    launch of the timed launches' shape — are VERIFIED ones.  The mutations (test build only): RTR_MGPU_TEST_WRONG_PLACE=2 puts every
    received shard in its neighbour's place while the launch's exchange is ONE group — the exchange completes, the frames are wrong — and
    the run must give the communicator up, come back with one group per slot and finish with that; =1 does it in both forms, and the
    run must stop: rc != 0, the reason on stderr and in the JSON line, no rate."""
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    base = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RTR_SCENE_CACHE=str(scene_cache), LD_PRELOAD=fake, RTR_MGPU_TEST_SHARED_DEVICE="1")
    base.pop("WORLD_SIZE", None); base.pop("RTR_MGPU_GROUP_PER_SLOT", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "6", "--warmup", "2", "--width", "320", "--height", "200"]

    def line(r):
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    good = subprocess.run(cmd, cwd=ROOT, env=base, capture_output=True, text=True, timeout=600)
    assert good.returncode == 0, good.stderr[-3000:]
    assert "N>1 start: one process drives the ranks" in good.stderr and "devices_present=" in good.stderr
    assert "communicator up | rccl_version=99999 nranks=3" in good.stderr and "first exchange verified | pixels_differing=0" in good.stderr and "grouping=one group per launch" in good.stderr
    d = line(good)
    assert d["rccl"]["first_exchange_verified"] is True and d["rccl"]["group_per_slot"] is False and d["rccl"]["first_batch_frames"] > 1 and d["value"] > 0
    # one group per launch misbehaves: the fallback takes over and the run completes
    fb = subprocess.run(cmd, cwd=ROOT, env=dict(base, RTR_MGPU_TEST_WRONG_PLACE="2"), capture_output=True, text=True, timeout=600)
    assert fb.returncode == 0, fb.stderr[-3000:]
    assert "first exchange WRONG" in fb.stderr and "again with one group per slot" in fb.stderr and "grouping=one group per slot" in fb.stderr
    d = line(fb)
    assert d["rccl"]["group_per_slot"] is True and d["rccl"]["first_exchange_verified"] is True and d["verify"]["assembled_vs_unsharded_pixels_differing"] == 0 and d["value"] > 0
    # wrong in both forms: nothing left to fall back to
    bad = subprocess.run(cmd, cwd=ROOT, env=dict(base, RTR_MGPU_TEST_WRONG_PLACE="1"), capture_output=True, text=True, timeout=600)
    assert bad.returncode == 4, (bad.returncode, bad.stderr[-2000:])
    assert bad.stderr.count("first exchange WRONG") == 2 and "FAILED at first exchange verification" in bad.stderr
    e = line(bad)
    assert e["value"] is None and e["error"]["stage"] == "first exchange verification" and e["rccl"]["first_exchange_verified"] is False and e["rccl"]["nranks"] == 3
    # the per-slot grouping asked for by hand: one attempt, with it
    per_slot = subprocess.run(cmd, cwd=ROOT, env=dict(base, RTR_MGPU_GROUP_PER_SLOT="1"), capture_output=True, text=True, timeout=600)
    assert per_slot.returncode == 0 and "group_per_slot=1" in per_slot.stderr and "grouping=one group per slot" in per_slot.stderr, per_slot.stderr[-2000:]
    assert line(per_slot)["verify"]["assembled_vs_unsharded_pixels_differing"] == 0
