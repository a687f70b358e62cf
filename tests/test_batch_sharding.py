"""The multi-GPU path: independent families sharded over ranks with no data-path collective.
World-size-2 gloo run on CPU: both ranks derive the same assignment without communicating, the
shards are disjoint and complete, and the bench's max-over-ranks timing reduction works."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from gaussdca.jl_amd.batch import batch_sizes, shard_families

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = batch_sizes(64)
    mine = shard_families(sizes, world)[rank]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((gathered, float(t.item())))
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0
    a, b = gathered
    assert sorted(a + b) == list(range(64)) and not (set(a) & set(b))


def test_lpt_balance_and_determinism():
    from gaussdca.jl_amd.batch import batch_sizes, family_cost, shard_families

    sizes = batch_sizes(256)
    assert all(100 <= n <= 600 and 5000 <= m <= 80000 for n, m in sizes)
    for world in (1, 2, 4, 8):
        sh = shard_families(sizes, world)
        assert sh == shard_families(sizes, world)
        assert sorted(sum(sh, [])) == list(range(256))
        loads = [sum(family_cost(*sizes[f]) for f in r) for r in sh]
        assert max(loads) / (sum(loads) / world) < 1.05
    with pytest.raises(ValueError):
        shard_families(sizes, 0)
