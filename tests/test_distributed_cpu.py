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


def _run_bench(argv, env_extra=None):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), *argv], env=env, capture_output=True,
                       text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` (the driver's form, no RANK in the env) must itself start two
    ranks and report the world size the ranks actually saw — with gloo and no kernels here."""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "0", "--selftest-cpu"])
    assert rc == 0, err
    assert len(lines) == 1, lines          # rank 0 alone prints
    assert lines[0]["n_gpus"] == 2 and lines[0]["config"]["parallelism"] == "dp2"
    assert lines[0]["cycles_all_ranks"] == 6.0   # summed over both ranks
    assert lines[0]["value"] is None             # a selftest never reports a measurement


def test_bench_baseline_config2_shape_eight_ranks_four_requests_each():
    """BASELINE.json configs[2] in one command — `--gpus 8 --requests-per-gpu 4` — through the launcher with eight gloo
    ranks on the host (benchmark.py:445 shards 32 prompts over 8 ranks; run_benchmark.sh:120-133 starts them): the line
    reports the 8-rank world, 32 requests, dp8, a per-GPU key beside the all-rank sum."""
    rc, lines, err = _run_bench(["--gpus", "8", "--requests-per-gpu", "4", "--steps", "2", "--warmup", "0", "--selftest-cpu"],
                                {"OMP_NUM_THREADS": "1"})
    assert rc == 0, err
    assert len(lines) == 1, lines
    ln = lines[0]
    assert ln["n_gpus"] == 8 and ln["config"]["requests"] == 32 and ln["config"]["parallelism"] == "dp8"
    assert ln["cycles_all_ranks"] == 16.0
    assert "value_per_gpu" in ln and ln["value"] is None and ln["value_per_gpu"] is None


def test_bench_rejects_a_world_that_differs_from_gpus():
    rc, lines, err = _run_bench(["--gpus", "2", "--selftest-cpu"],
                                {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                 "MASTER_PORT": str(_free_port())})
    assert rc != 0 and not lines and "WORLD_SIZE=1" in err


def test_bench_launcher_forwards_a_failing_rank():
    # no GPU here: every rank exits non-zero before any rendezvous, and so must the launcher
    if torch.cuda.is_available():
        return
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "1"])
    assert rc != 0 and not lines
