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


def test_lpt_model_against_measured_family_times():
    """the LPT cost model (gaussdca.jl_amd/batch.py) against MEASURED per-family device times -- all 256 families of
    BASELINE.json's batch configuration one after the other on one MI355X (profiles/r06_E_per_family.json, written by
    `bench.py --config E --dump-families`): the makespan the model predicts for its own sharding is within 5 % of the makespan those
    shards have by the measured times, at 2, 4 and 8 ranks; the measured makespan is within 3 % of the ideal (sum / world); and the
    model's per-family totals are within 25 % of the measured ones for 95 % of the families."""
    import json

    import numpy as np

    from gaussdca.jl_amd.batch import batch_sizes, family_cost, shard_families

    with open(os.path.join(ROOT, "profiles", "r06_E_per_family.json")) as f:
        fams = json.load(f)["families"]
    assert len(fams) == 256
    sizes = [(int(x["N"]), int(x["M"])) for x in fams]
    assert sorted(sizes) == sorted(batch_sizes(256))  # (the bench processes them in LPT order: the same families)
    tot = np.array([x["ms_total"] for x in fams]) * 1e-3
    model = np.array([family_cost(n, m) for n, m in sizes])
    ratio = model / tot
    assert 0.75 < np.percentile(ratio, 2.5) and np.percentile(ratio, 97.5) < 1.25, np.percentile(ratio, [2.5, 50, 97.5])
    assert abs(model.sum() / tot.sum() - 1.0) < 0.03
    for world in (2, 4, 8):
        sh = shard_families(sizes, world)
        pred = max(sum(model[f] for f in s) for s in sh)
        meas = max(sum(tot[f] for f in s) for s in sh)
        assert abs(pred / meas - 1.0) < 0.05, (world, pred, meas)
        assert meas / (tot.sum() / world) < 1.03, (world, meas, tot.sum() / world)


def _run_bench(args, timeout=300):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher around it) must itself start two ranks: the JSON's n_gpus is 2, the
    two ranks hold disjoint, complete shards of the batch (config E), the max-over-ranks reduction ran (gloo on CPU
    for this dry run; the real run uses RCCL for exactly the same two calls)."""
    from gaussdca.jl_amd.batch import batch_sizes, shard_families

    out = _run_bench(["--gpus", "2", "--dry-run", "--config", "E", "--families", "40"])
    assert out["dry_run"] and out["n_gpus"] == 2 and out["scaling"] == "strong"
    a, b = out["shards"]
    assert sorted(a + b) == list(range(40)) and not (set(a) & set(b))
    assert [a, b] == shard_families(batch_sizes(40), 2)
    assert out["max_over_ranks"] == 2.0
    # the headline config: weak scaling, one family per rank, different seeds (family id == rank)
    out = _run_bench(["--gpus", "2", "--dry-run"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["shards"] == [[0], [1]]
    out = _run_bench(["--gpus", "1", "--dry-run"])
    assert out["n_gpus"] == 1 and out["shards"] == [[0]]


def test_bench_under_an_external_launcher_is_one_rank():
    """The driver's form: torch.distributed.run around `bench.py --gpus N`: the ranks come from the launcher's
    environment, bench.py must not start further processes."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
           "--config", "E", "--families", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and sorted(sum(out["shards"], [])) == list(range(16))
