"""N > 1 path on CPU: two gloo ranks shard requests round-robin, gather per-request
results to rank 0 and reduce the bench's timing scalars; no other collective exists."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

import helpers as H


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, H.ROOT)
    from dflash_amd import distributed as D
    D.init("gloo")
    n = 7
    mine = [{"request": i, "ids": [i, i * i], "rank": rank} for i in D.shard_indices(n)]
    got = D.gather(mine)
    t, u = D.reduce_timing(1.0 + rank, 10.0 * (rank + 1))
    if D.is_main():
        merged = D.merge_sharded(got, n)
        q.put((merged, t, u))
    D.destroy()


def test_two_rank_gloo_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, t, u = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [m["request"] for m in merged] == list(range(7))
    assert [m["rank"] for m in merged] == [0, 1, 0, 1, 0, 1, 0]      # benchmark.py:445 split
    assert all(m["ids"] == [m["request"], m["request"] ** 2] for m in merged)
    assert (t, u) == (2.0, 30.0)                                       # max time, summed units


def test_single_process_degrades():
    from dflash_amd import distributed as D
    assert D.size() == 1 and D.rank() == 0 and list(D.shard_indices(5)) == [0, 1, 2, 3, 4]
    assert D.gather("x") == ["x"]
    assert D.reduce_timing(1.5, 3.0) == (1.5, 3.0)
