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
    lm_head launch, losslessness of the scripted run."""
    d = _run(["--steps", "6", "--warmup", "1", "--target-layers", "8", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["lossless_fraction"] == 1.0 and d["value"] > 0
    rf = d["roofline"]                          # the kernel with the largest share of the cycle: the gate/up GEMM
    lm = rf["also"][0]                          # beside it: the lm_head GEMM + fused argmax
    for r in (rf, lm):
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.4 < r["frac"] < 1.0   # (two event samples only)
    assert "EPI_SILU" in rf["kernel"] and rf["bytes_per_launch"] == 2 * 12288 * 4096 * 2
    assert 0.025 < rf["avg_ms"] < 0.06         # one gate/up launch (201 MB), not a whole layer (~0.09 ms)
    assert 0.15 < lm["avg_ms"] < 0.30          # the lm_head GEMM alone (1.245 GB), not the launch pair
    assert d["value_per_gpu"] == d["value"]


def test_two_rank_flow_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` end to end on hardware: the launcher starts two fresh ranks, they rendezvous at
    127.0.0.1, barrier around the timed region, reduce the timing scalars and rank 0 prints the line with n_gpus = 2.
    Rehearsal mode (DFL_BENCH_SHARE_GPU=1): both ranks on cuda:0 over gloo — RCCL refuses two ranks on one device;
    the 8-GPU run itself is the driver's."""
    d = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--target-layers", "8"], env={"DFL_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["requests"] == 2
    assert d["lossless_fraction"] == 1.0 and "rehearsal" in d
    assert d["cpu_baseline"] is None            # rank 0 at N = 1 only
