"""N > 1 host path on CPU: world_size-2 (and 3) gloo process groups run the package's sharding +
all-reduce + finalize logic.  The per-shard statistics come from the CPU oracle here (no GPU in this
container) — on the GPU box the same code path gets them from mcamd_price_paths (see bench.py)."""
import importlib
import math
import os
import socket
import sys

import pytest

from conftest import ROOT

pkg = importlib.import_module("monte-carlo-project-cuda_amd")
sharding = pkg.sharding


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 10_000_000, 1_000_000_007):
        for world in (1, 2, 3, 4, 8):
            parts = [sharding.shard_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (lo, c), (lo2, _) in zip(parts[:-1], parts[1:]):
                assert lo + c == lo2
            sizes = [c for _, c in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _worker(rank, world, port, n_total, n_steps, precision, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    pkg_ = importlib.import_module("monte-carlo-project-cuda_amd")
    from oracle import pyoracle as o
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = o.make_params(n_paths=n_total, n_steps=n_steps, seed=1234)

    def local_stats(lo, n_local):
        res = o.mc_paths(p, precision, lo, n_local)
        return res["sum"], res["sumsq"]

    res = pkg_.sharding.price_sharded(local_stats, n_total, world, rank,
                                      lambda s, s2, n: pkg_.capi.finalize(s, s2, n, 0.1, 1.0))
    q.put((rank, res.sum, res.sumsq, res.n, res.price, res.std_err))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_price_equals_single_process(world, oracle):
    import torch.multiprocessing as mp
    n_total, n_steps, precision = 20_001, 6, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, n_steps, precision, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = oracle.mc_paths(oracle.make_params(n_paths=n_total, n_steps=n_steps, seed=1234), precision, 0, n_total)
    fin = oracle.finalize(whole["sum"], whole["sumsq"], n_total, 0.1, 1.0)
    for rank, s, s2, n, price, se in got:
        assert n == n_total
        assert math.isclose(s, whole["sum"], rel_tol=1e-12) and math.isclose(s2, whole["sumsq"], rel_tol=1e-12)
        assert math.isclose(price, fin["price"], rel_tol=1e-12) and math.isclose(se, fin["std_err"], rel_tol=1e-9)
    # every rank holds the same reduced result
    assert len({(g[1], g[2], g[3]) for g in got}) == 1
