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


def test_single_process_allgather_is_a_copy():
    f = torch.randn(6, 8, 2, 3)
    g = tdist.all_gather_features(f, dtype=torch.float32)
    assert g.shape == (1, 6, 8, 2, 3) and torch.equal(g[0], f)
