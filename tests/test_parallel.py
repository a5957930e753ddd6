"""Multi-rank layout (CPU, gloo, world_size 2): object sharding and the end-of-run gather."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from graspqp_amd.parallel import gather_results, shard_objects


def test_shard_objects_partition():
    for n, w in ((64, 8), (10, 4), (3, 8), (256, 8), (1, 1)):
        blocks = [shard_objects(n, w, r) for r in range(w)]
        flat = [o for b in blocks for o in b]
        assert flat == list(range(n))  # contiguous, whole objects, nothing lost or duplicated
        assert max(len(b) for b in blocks) - min(len(b) for b in blocks) <= 1
    with pytest.raises(ValueError):
        shard_objects(4, 2, 2)


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    objs = shard_objects(3, world, rank)  # ragged: rank 0 owns 2 objects, rank 1 owns 1
    be, D, n = 4, 25, 12
    B = len(objs) * be
    # every rank is an independent reference-equivalent process over its objects: rows tagged by object id
    pose = torch.stack([torch.full((D,), float(o)) for o in objs for _ in range(be)])
    energy = torch.tensor([float(o) * 10 + i for o in objs for i in range(be)])
    idx = torch.tensor([[o] * n for o in objs for _ in range(be)])
    poses, energies, idxs = gather_results(pose, energy, idx, dst=0)
    if rank == 0:
        q.put((torch.cat(poses)[:, 0].tolist(), torch.cat(energies).tolist(), torch.cat(idxs)[:, 0].tolist()))
    else:
        assert poses is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_results_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    pose0, energy, idx0 = res
    assert pose0 == [0.0] * 4 + [1.0] * 4 + [2.0] * 4
    assert idx0 == [0] * 4 + [1] * 4 + [2] * 4
    assert energy == [o * 10.0 + i for o in range(3) for i in range(4)]


def _bench(*flags, env=None):
    import subprocess
    import sys

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags], env=e, capture_output=True, text=True,
                          timeout=300)


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` launches two ranks itself (the parent touches no GPU); their rendezvous, barrier-
    bracketed window, MAX over ranks and end-of-run gather run here over gloo (--selftest_ranks: no GPU work)."""
    import json

    r = _bench("--gpus", "2", "--dist_backend", "gloo", "--selftest_ranks", "--n_objects", "3", "--batch_size", "4")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["objects_per_rank"] == [3, 3] and out["rows_gathered"] == 2 * 3 * 4
    assert out["window_s_max_over_ranks"] >= 0.02  # the slower rank (rank 1 sleeps 20 ms) sets the window


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _bench("--gpus", "2", "--selftest_ranks", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
