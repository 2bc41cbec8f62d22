"""The N > 1 path on CPU: two processes over gloo shard an event-id range exactly as
bench.py does on N GPUs (contiguous ranges, no data-path collective, barrier + MAX/SUM of a
few scalars).  The per-rank "device" here is the CPU oracle; what is under test is the
sharding/reduction logic and that results do not depend on the number of ranks."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from attpc_engine_amd import sharding


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_partitions():
    for total, world in [(10, 2), (1_000_000, 8), (7, 3), (5, 8)]:
        ranges = [sharding.strong_shard(total, r, world, first_event=100) for r in range(world)]
        assert ranges[0][0] == 100 and sum(n for _, n in ranges) == total
        for (a, n), (b, _) in zip(ranges, ranges[1:]):
            assert a + n == b
        assert max(n for _, n in ranges) - min(n for _, n in ranges) <= 1
    assert sharding.weak_shard(1000, 3) == (3000, 1000)
    assert sharding.weak_shard(1000, 0, first_event=5) == (5, 1000)
    assert sharding.reduce_scalars(None, [1.0, 2.0], "max") == [1.0, 2.0]
    assert sharding.reduce_checksums(None, [(1 << 64) + 5]) == [5]


def _rank_main(rank: int, world_size: int, port: int, per_rank: int, queue) -> None:
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world_size),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from oracle import pyoracle as orc
    from tests.helpers import Inputs

    dist = sharding.init_process_group("gloo")
    assert sharding.world() == (rank, rank, world_size)
    inp = Inputs("be10dp")
    first, n = sharding.weak_shard(per_rank, rank)
    sharding.barrier(dist)
    res = orc.sim_batch(inp.kin, inp.det_raw, inp.layout, seed=2, first=first, n=n, threads=1)
    sharding.barrier(dist)
    (t_max,) = sharding.reduce_scalars(dist, [float(rank + 1)], "max")
    (points,) = sharding.reduce_scalars(dist, [float(res["stats"][0])], "sum")
    charge, keys = sharding.reduce_checksums(dist, [res["stats"][2], res["stats"][3]])
    if rank == 0:
        queue.put((t_max, points, charge, keys))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_match_one(tmp_path):
    from oracle import pyoracle as orc
    from tests.helpers import Inputs

    per_rank, world_size = 6, 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world_size, port, per_rank, queue)) for r in range(world_size)]
    for p in procs:
        p.start()
    t_max, points, charge, keys = queue.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    inp = Inputs("be10dp")
    whole = orc.sim_batch(inp.kin, inp.det_raw, inp.layout, seed=2, first=0, n=per_rank * world_size, threads=2)
    assert t_max == float(world_size)          # MAX over ranks
    assert points == float(whole["stats"][0])  # SUM over ranks == one process over the union range
    assert charge == whole["stats"][2] % (1 << 64)
    assert keys == whole["stats"][3] % (1 << 64)
