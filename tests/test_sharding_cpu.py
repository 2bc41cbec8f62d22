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


def _run_bench(args: list[str], env_extra: dict | None = None):
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    proc = subprocess.run([sys.executable, str(root / "bench.py"), "--stub-engine"] + args, env=env,
                          capture_output=True, text=True, timeout=240)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    return proc.returncode, [json.loads(ln) for ln in lines], proc.stderr


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no torchrun environment (the way the driver starts N = 1) must produce an
    N-rank number: the parent starts N fresh ranks, gloo sees them all, ONE JSON line comes out, with n_gpus = N, and
    the sums over ranks equal one rank over the union of the id ranges (stub engine: no GPU here)."""
    rc1, one, _ = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--events", "600"])
    rc2, two, err = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--events", "300"])
    assert rc1 == 0 and rc2 == 0, err
    assert len(one) == 1 and len(two) == 1           # exactly one line on stdout
    assert one[0]["n_gpus"] == 1 and two[0]["n_gpus"] == 2
    assert two[0]["config"]["ranks_reporting"] == 2 and two[0]["scaling"] == "weak"
    assert two[0]["config"]["global_events_per_step"] == 600 == one[0]["config"]["global_events_per_step"]
    for key in ("charge_checksum", "key_checksum", "points_per_event"):  # same global ids -> same totals
        assert two[0]["config"][key] == one[0]["config"][key]
    assert "stub" in two[0]["data"]

    # strong scaling (BASELINE configs[3] shape): a fixed total split into contiguous shares, uneven remainder included
    rc3, three, err = _run_bench(["--gpus", "3", "--steps", "2", "--warmup", "0", "--global-events", "601"])
    assert rc3 == 0, err
    assert three[0]["n_gpus"] == 3 and three[0]["scaling"] == "strong"
    assert three[0]["config"]["global_events_per_step"] == 601 and three[0]["config"]["ranks_reporting"] == 3


@pytest.mark.timeout(120)
def test_bench_refuses_a_world_that_is_not_what_was_asked_for():
    """--gpus N inside a torchrun environment of another size is an error, not an `n_gpus: 1` line."""
    rc, lines, err = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not lines and "WORLD_SIZE=4" in err
    rc, lines, err = _run_bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not lines


@pytest.mark.timeout(120)
def test_bench_fails_when_a_rank_fails():
    """A rank that dies makes the launcher exit non-zero (here: an unknown workload raises in every rank)."""
    rc, lines, _ = _run_bench(["--gpus", "2", "--workload", "no_such_workload"])
    assert rc != 0 and not lines
