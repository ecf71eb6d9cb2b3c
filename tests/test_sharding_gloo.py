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


def _nmc_worker(rank, world, port, n_paths, n_steps, n_inner, q):
    """Nested MC sharded by outer path, as bench.py --workload nmc does it on N ranks: each rank prices the points of its
    outer paths (here: the oracle's brute-force point pricer with the SHIFTED point id) and ONE all-reduce of the
    statistics record {sum, sum of squares, points, work, live} gives every rank the whole job's diagnostic."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    pkg_ = importlib.import_module("monte-carlo-project-cuda_amd")
    from oracle import pyoracle as o
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, m = pkg_.sharding.shard_range(n_paths, world, rank)
    po = o.make_params(B=104.0, P1=1, P2=5, use_window=1, n_paths=n_paths, n_steps=n_steps, seed=1234)
    outer = o.mc_paths(po, 64, lo, m, want_traj=True, want_counts=True)
    pi = o.make_params(B=104.0, P1=1, P2=5, use_window=1, n_paths=n_paths, n_steps=n_steps, n_paths_inner=n_inner, seed=1235)
    prices = [o.nmc_point(pi, 64, (lo + q_) * n_steps + s_, s_, float(outer["traj"][s_, q_]), int(outer["counts"][s_, q_]))
              for q_ in range(m) for s_ in range(n_steps)]
    rec = [sum(prices), sum(p * p for p in prices), len(prices), 64.0 * len(prices), float(len(prices))]
    s, s2, n, work, live = pkg_.sharding.allreduce_vector(rec)
    fin = pkg_.capi.finalize_nmc_stats([s, s2, work / 64.0, live, 0.0, n])
    q.put((rank, fin.sum, fin.n, fin.price, fin.work_steps, fin.live_steps, prices))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_nested_mc_equals_single_process(oracle):
    import torch.multiprocessing as mp
    world, n_paths, n_steps, n_inner = 2, 5, 4, 30
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nmc_worker, args=(r, world, port, n_paths, n_steps, n_inner, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the whole job on one process: same streams (global path ids), so the shards' points are the whole job's points
    po = oracle.make_params(B=104.0, P1=1, P2=5, use_window=1, n_paths=n_paths, n_steps=n_steps, seed=1234)
    outer = oracle.mc_paths(po, 64, 0, n_paths, want_traj=True, want_counts=True)
    pi = oracle.make_params(B=104.0, P1=1, P2=5, use_window=1, n_paths=n_paths, n_steps=n_steps, n_paths_inner=n_inner, seed=1235)
    whole = [oracle.nmc_point(pi, 64, q_ * n_steps + s_, s_, float(outer["traj"][s_, q_]), int(outer["counts"][s_, q_]))
             for q_ in range(n_paths) for s_ in range(n_steps)]
    assert got[0][6] + got[1][6] == whole and max(whole) > 0
    for _, s, n, mean, work, live, _ in got:
        assert n == n_paths * n_steps and math.isclose(s, math.fsum(whole), rel_tol=1e-12)
        assert math.isclose(mean, math.fsum(whole) / n, rel_tol=1e-12) and work == 64.0 * n and live == n
