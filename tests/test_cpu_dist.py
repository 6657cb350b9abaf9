"""CPU: the N>1 path (frame sharding + the one feature all-gather) with world_size 2 on gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from toc3d_amd import dist as tdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = list(tdist.frames_for_rank(n_frames, rank, world))
    # stand-in for this rank's neck output: value encodes (frame id, view)
    feat = torch.stack([torch.full((4, 3, 5), float(10 * mine[0] + v)) for v in range(6)])
    g = tdist.all_gather_features(feat, dtype=torch.float32)
    t = tdist.max_over_ranks(1.0 + rank, "cpu")
    q.put((rank, mine, g[:, :, 0, 0, 0].tolist(), t))
    dist.barrier()
    dist.destroy_process_group()


def test_frames_shard_like_the_reference_sampler():
    assert [list(tdist.frames_for_rank(8, r, 8)) for r in range(8)] == [[r] for r in range(8)]
    assert [list(tdist.frames_for_rank(8, r, 2)) for r in range(2)] == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert [list(tdist.frames_for_rank(5, r, 4)) for r in range(4)] == [[0, 1], [2, 3], [4], []]
    got = sorted(f for r in range(3) for f in tdist.frames_for_rank(7, r, 3))
    assert got == list(range(7))


def test_world2_gloo_allgather_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mine, g, t in res:
        assert mine == [rank]
        assert g == [[0.0, 1.0, 2.0, 3.0, 4.0, 5.0], [10.0, 11.0, 12.0, 13.0, 14.0, 15.0]]     # every rank sees both frames, in rank order
        assert t == 2.0                                                                        # max over ranks


def _bench_worker(rank, world, port, frames_total, q):
    """bench.py's N > 1 step / barrier / timing logic (toc3d_amd.dist.timed_steps + FeatureGather) with a stub model."""
    import time
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = list(tdist.frames_for_rank(frames_total, rank, world)) if frames_total else [rank]
    gather = tdist.FeatureGather((6, 4, 2, 3), "cpu", dtype=torch.float32, depth=2)
    log = []

    def step():
        for f in mine:
            time.sleep(0.01 * (1 + rank))                                  # stub backbone: rank 1 is the slow one
            feat = torch.full((6, 4, 2, 3), float(f))
            t = gather.submit(feat)                                        # exchange of frame f overlaps the next frame's "backbone"
            if t >= 1:
                log.append(gather.wait(t - 1)[:, 0, 0, 0, 0].tolist())     # consumer reads the previous frame's exchange

    elapsed = tdist.timed_steps(step, steps=3, warmup=1, device="cpu", finish=gather.drain)
    last = gather.out[(gather.n - 1) % 2][:, 0, 0, 0, 0].tolist()
    q.put((rank, mine, elapsed, log[-1], last, gather.n))
    dist.barrier()
    dist.destroy_process_group()


def _run_world2(target, *args):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_world2_timed_steps_weak_scaling_with_overlapped_gather():
    res = _run_world2(_bench_worker, 0)
    (r0, m0, t0, log0, last0, n0), (r1, m1, t1, log1, last1, n1) = res
    assert m0 == [0] and m1 == [1] and n0 == n1 == 4                       # 1 warm-up + 3 timed steps, one exchange each
    assert t0 == t1                                                        # max over ranks, identical on every rank
    assert t0 >= 3 * 0.02                                                  # ... and set by the slow rank (20 ms per step)
    assert log0 == log1 == [0.0, 1.0] and last0 == last1 == [0.0, 1.0]     # every rank sees both ranks' frames, in rank order


def test_world2_timed_steps_strong_scaling_chunks():
    res = _run_world2(_bench_worker, 4)                                    # 4 frames per step over 2 ranks: [0, 1] and [2, 3]
    (r0, m0, t0, log0, last0, n0), (r1, m1, t1, log1, last1, n1) = res
    assert m0 == [0, 1] and m1 == [2, 3] and n0 == n1 == 8
    assert t0 == t1 and t0 >= 3 * 2 * 0.02
    assert last0 == last1 == [1.0, 3.0]                                    # the step's last exchange: each rank's second frame
    assert log0 == log1 == [0.0, 2.0]                                      # the one before: each rank's first frame of the last step


def _census_worker(rank, world, port, corrupt, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    feat = torch.randn(6, 8, 4, 5, generator=torch.Generator().manual_seed(rank))
    if corrupt:
        # a broken exchange: rank 1's slice arrives damaged on every rank (the census must notice, on every rank)
        real = tdist.all_gather_features

        def damaged(f, out=None, dtype=torch.bfloat16):
            g = real(f, out, dtype)
            g[1, 0, 0, 0, 0] += 1.0
            return g
        tdist.all_gather_features = damaged
    c = tdist.exchange_census(feat, "cpu")
    q.put((rank, c["ranks_seen"], c["gather_bytes"], c["verified"]))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_exchange_census_proves_the_collective_saw_every_rank():
    for corrupt in (False, True):
        res = _run_world2(_census_worker, corrupt)
        for rank, seen, nbytes, ok in res:
            assert seen == [0, 1] and nbytes == 2 * 6 * 8 * 4 * 5 * 2
            assert ok is (not corrupt)
    one = tdist.exchange_census(torch.randn(6, 8, 4, 5), "cpu")            # no process group: a one-rank census
    assert one == {"ranks_seen": [0], "gather_bytes": 6 * 8 * 4 * 5 * 2, "verified": True}


def test_feature_gather_single_process_ring():
    g = tdist.FeatureGather((2, 3), "cpu", dtype=torch.float32, depth=2)
    t0 = g.submit(torch.full((2, 3), 1.0))
    t1 = g.submit(torch.full((2, 3), 2.0))
    assert g.wait(t0)[0, 0, 0].item() == 1.0 and g.wait(t1)[0, 0, 0].item() == 2.0
    t2 = g.submit(torch.full((2, 3), 3.0))                                 # reuses slot 0
    assert g.wait(t2)[0, 0, 0].item() == 3.0
    import pytest
    with pytest.raises(AssertionError):
        g.wait(t0)                                                         # older than the ring
    g.drain()


def test_pin_rank_to_cores_partitions_the_allowed_set():
    if not hasattr(os, "sched_getaffinity"):
        return
    before = os.sched_getaffinity(0)
    try:
        if len(before) >= 2:
            a = tdist.pin_rank_to_cores(0, 2)
            os.sched_setaffinity(0, before)
            b = tdist.pin_rank_to_cores(1, 2)
            assert a and b and not (set(a) & set(b)) and set(a) | set(b) <= set(before)
        assert tdist.pin_rank_to_cores(0, 1) is None
    finally:
        os.sched_setaffinity(0, before)


def test_single_process_allgather_is_a_copy():
    f = torch.randn(6, 8, 2, 3)
    g = tdist.all_gather_features(f, dtype=torch.float32)
    assert g.shape == (1, 6, 8, 2, 3) and torch.equal(g[0], f)


def test_bench_py_launcher_contract_world2_dry_run():
    """bench.py launched exactly as the driver launches it for N = 2 (python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 ... bench.py --gpus 2 --steps K --warmup W), in its dry-run mode (stand-in step on gloo, no kernels): rank env
    handling, frame sharding, the overlapped exchange, barriers, max over ranks and the ONE JSON line of rank 0 -- weak and strong scaling."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TOC3D_BENCH_DRY_RUN="1")
    for extra, scaling, fps, last, fpf in (([], "weak", 2, [0.0, 1.0], 1), (["--frames-total", "4"], "strong", 4, [0.0, 1.0, 2.0, 3.0], 2),
                                           (["--frames-total", "4", "--sequential-frames"], "strong", 4, None, 1)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"] + extra
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout                                   # rank 0 prints exactly one JSON line
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == scaling and d["higher_is_better"] is True
        assert d["config"]["frames_per_step"] == fps and d["value"] > 0 and "DRY RUN" in d["metric"]
        assert d["config"]["frames_per_forward"] == fpf                    # strong scaling: a rank's frames are ONE forward (B = frames per rank)
        assert abs(d["value"] - fps * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
        if last is not None:
            assert d["config"]["last_exchange"] == last                    # every rank's frame, in rank order
        # the self-proving part of an N > 1 line: rank census + per-rank checksums through the collective itself
        assert d["config"]["ranks_seen"] == [0, 1] and d["config"]["gather_verified"] is True
        assert d["config"]["gather_bytes"] == 2 * 6 * 4 * 2 * 5 * 2         # two ranks x bf16 features


def test_bench_py_launcher_contract_world8_dry_run():
    """The driver's N = 8 launch (python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8) in dry-run mode: weak scaling (one frame per rank)
    and strong scaling with --frames-total 64 (eight frames per rank as ONE forward of B = 8, samplers/distributed_sampler.py:41-44): every rank is seen, the
    exchange is verified through the collective itself, value = frames / max-over-ranks time.  (No GPU node is available to this build: SURVEY.md 8e, BASELINE
    config 5 -- this pins the control flow the driver's 8-GPU run will execute.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TOC3D_BENCH_DRY_RUN="1", OMP_NUM_THREADS="1")
    for extra, scaling, fps, fpf in (([], "weak", 8, 1), (["--frames-total", "64"], "strong", 64, 8)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"] + extra
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 8 and d["scaling"] == scaling and d["config"]["frames_per_step"] == fps and d["config"]["frames_per_forward"] == fpf
        assert d["config"]["ranks_seen"] == list(range(8)) and d["config"]["gather_verified"] is True
        assert d["config"]["gather_bytes"] == 8 * 6 * 4 * 2 * 5 * 2
        assert abs(d["value"] - fps * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
        if not extra:
            assert d["config"]["last_exchange"] == [float(r_) for r_ in range(8)]
