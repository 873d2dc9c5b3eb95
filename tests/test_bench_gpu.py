"""bench.py on the GPU box: the one-line contract, and the N-rank control flow rehearsed on one GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=420):
    e = dict(os.environ, **(env or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=e,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line
    return json.loads(lines[0])


def test_bench_line_contract_small():
    """A short run on 8 target layers: every key of the contract, the roofline object from live HIP events around the
    lm_head launch, losslessness of the scripted run — and the `batch4` object (BASELINE configs[2]'s per-GPU leg: four
    requests per GPU as one ragged batch) timed behind the headline by the same command."""
    d = _run(["--steps", "6", "--warmup", "1", "--target-layers", "8", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "batch4"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["lossless_fraction"] == 1.0 and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]                          # the kernel with the largest share of the cycle: the gate/up GEMM
    lm = rf["also"][0]                          # beside it: the lm_head GEMM + fused argmax
    for r in (rf, lm):
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.4 < r["frac"] < 1.0   # (two event samples only)
    assert "EPI_SILU" in rf["kernel"] and rf["bytes_per_launch"] == 2 * 12288 * 4096 * 2
    assert 0.025 < rf["avg_ms"] < 0.06         # one gate/up launch (201 MB), not a whole layer (~0.09 ms)
    assert 0.15 < lm["avg_ms"] < 0.30          # the lm_head GEMM alone (1.245 GB), not the launch pair
    assert d["value_per_gpu"] == d["value"]
    assert "replay" in d["host_side"]["mode"]   # steady-state cycles come from the captured hipGraphs by default
    b4 = d["batch4"]
    assert b4["lossless_fraction"] == 1.0 and b4["requests"] == 4 and b4["steps"] == 6
    assert b4["value_per_gpu"] == b4["value"] > d["value"]          # four requests share the weight stream
    assert d["ms_per_step"] < b4["ms_per_step"] < 2.5 * d["ms_per_step"]
    r4 = b4["roofline"]
    assert r4["bound"] == "hbm" and r4["unit"] == "GB/s" and 0.3 < r4["frac"] < 1.0 and r4["bytes_per_launch"] == 151936 * 4096 * 2


def test_bench_eager_flag_and_batched_headline():
    """--eager: ctypes launches instead of graph replays (same ids); --requests-per-gpu 4: the ragged batch as the headline."""
    d = _run(["--steps", "4", "--warmup", "1", "--target-layers", "4", "--no-cpu-baseline", "--eager", "--no-batch4"])
    assert d["lossless_fraction"] == 1.0 and d["host_side"]["mode"] == "eager launches" and d["batch4"] is None
    d = _run(["--steps", "4", "--warmup", "1", "--target-layers", "4", "--no-cpu-baseline", "--requests-per-gpu", "4"])
    assert d["lossless_fraction"] == 1.0 and d["config"]["requests"] == 4 and "replay" in d["mode"]


def test_bench_other_workloads_run_lossless():
    """--workload llama31-8b (BASELINE configs[3]: T = 0.7, the sampling path) and qwen3-30b-a3b (configs[4]: sparse-MoE
    target under the EWMA block-size schedule) at reduced depth: the lines are well-formed and lossless."""
    d = _run(["--workload", "llama31-8b", "--steps", "6", "--warmup", "1", "--target-layers", "4", "--no-cpu-baseline"])
    assert d["lossless_fraction"] == 1.0 and "temp=0.7" in d["config"]["workload"] and d["batch4"] is None
    assert d["roofline"]["bytes_per_launch"] == 2 * 14336 * 4096 * 2 and d["host_side"]["mode"] == "eager launches"
    d = _run(["--workload", "qwen3-30b-a3b", "--steps", "8", "--warmup", "2", "--target-layers", "4", "--no-cpu-baseline"])
    assert d["lossless_fraction"] == 1.0 and sum(d["used_block_sizes"].values()) == 8
    assert set(d["used_block_sizes"]) <= {"8", "12", "16"}
    rf = d["roofline"]
    assert "k_moe_gate_up" in rf["kernel"] and 8 <= rf["active_experts_mean"] <= 128
    assert abs(rf["bytes_per_launch"] - rf["active_experts_mean"] * 2 * 768 * 2048 * 2) < 1


def test_two_rank_flow_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` end to end on hardware: the launcher starts two fresh ranks, they rendezvous at
    127.0.0.1, barrier around the timed region, reduce the timing scalars and rank 0 prints the line with n_gpus = 2.
    Rehearsal mode (DFL_BENCH_SHARE_GPU=1): both ranks on cuda:0 over gloo — RCCL refuses two ranks on one device;
    the 8-GPU run itself is the driver's."""
    d = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--target-layers", "8"], env={"DFL_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["requests"] == 2
    assert d["lossless_fraction"] == 1.0 and "rehearsal" in d
    assert d["cpu_baseline"] is None            # rank 0 at N = 1 only
