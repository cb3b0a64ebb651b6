"""CPU, world_size 2 over gloo: the N>1 path of bench.py (problem sharding + the one all-gather
used for the throughput report).  No GPU work: each rank solves its shard with the oracle."""
import os
import socket

import numpy as np
import pytest

from oxmpl_amd import scenarios, sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, per_gpu, q):
    import torch.distributed as dist
    from oracle import oracle_py as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.problem_range(rank, per_gpu)
    sc = scenarios.config2()
    chk = []
    iters = 0
    for pid in range(lo, hi):
        o = orc.OracleRRT(sc["dim"], sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"],
                          200, False, 42, pid)
        o.set_spheres(*sc["spheres"])
        o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
        o.solve(150)
        chk.append(o.checksum)
        iters += o.iterations
    dist.barrier()
    allst = sharding.gather_stats([1.0 + rank, float(iters), float(sum(c % 1000003 for c in chk))])
    agg = sharding.aggregate(allst)
    if rank == 0:
        q.put((allst.tolist(), agg, chk))
    else:
        q.put(("rank1", chk))
    dist.barrier()
    dist.destroy_process_group()


def test_problem_sharding_and_gather_world2():
    import torch.multiprocessing as mp
    from oracle import oracle_py as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, per_gpu = _free_port(), 3
    procs = [ctx.Process(target=_worker, args=(r, 2, port, per_gpu, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = [g for g in got if g[0] != "rank1"][0]
    r1 = [g for g in got if g[0] == "rank1"][0]
    allst, agg, chk0 = r0
    assert np.asarray(allst).shape == (2, 3)
    assert agg["t_max"] == 2.0 and agg["total_units"] == 2 * per_gpu * 150
    assert agg["value"] == agg["total_units"] / 2.0      # slowest rank's time
    # sharding invariance: the union of both shards equals one unsharded run over ids 0..5
    sc = scenarios.config2()
    want = []
    for pid in range(2 * per_gpu):
        o = orc.OracleRRT(sc["dim"], sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"],
                          200, False, 42, pid)
        o.set_spheres(*sc["spheres"])
        o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
        o.solve(150)
        want.append(o.checksum)
    assert chk0 + r1[1] == want
    assert sharding.problem_range(3, 1024) == (3072, 4096)


def test_gather_without_process_group():
    allst = sharding.gather_stats([0.5, 10.0])
    assert allst.shape == (1, 2)
    assert sharding.aggregate(allst)["value"] == 20.0
