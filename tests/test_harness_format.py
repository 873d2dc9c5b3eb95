"""Harness wire format: numbers equal the reference's summarize_* (golden G7) and every
line run_block_sweep.sh greps for is present in the form its regexes expect."""
import json
import os
import re
from types import SimpleNamespace

import pytest

import helpers as H
from dflash_amd import harness

G = json.load(open(os.path.join(H.GOLDEN, "harness.json")))
NS = [SimpleNamespace(**s) for s in G["samples"]]

# run_block_sweep.sh:199-212
SWEEP_PATTERNS = [r"Decoding speedup: [0-9.]+$", r"Average Acceptance length: [0-9.]+$",
                  r"Speculative total_wall_s: [0-9.]+$", r"Speculative tokens_per_sec: [0-9.]+$",
                  r"Speculative TPOT: [0-9.]+$", r"Speculative TTFT: [0-9.]+$", r"^Hardware GPU:", r"^Hardware CUDA:",
                  r"^Hardware Torch:", r"Baseline total_wall_s: [0-9.]+$", r"Baseline tokens_per_sec: [0-9.]+$",
                  r"Baseline TPOT: [0-9.]+$", r"Baseline TTFT: [0-9.]+$", r"Acceptance length histogram:"]


def test_summaries_equal_reference():
    assert harness.summarize_mode(NS) == G["summarize_mode"]
    assert harness.summarize_profile(NS) == G["summarize_profile"]
    assert harness.summarize_profile([SimpleNamespace(profile_summary=None)]) is None


def test_stat_lines_match_the_sweep_regexes():
    responses = [{1: NS[i], 16: NS[(i + 1) % len(NS)]} for i in range(len(NS))]
    lines = harness.stat_lines(responses, 16, draft_steps=1, baseline=True, collect_profile=True,
                               gpu_name="AMD Instinct MI355X", runtime_version=None, torch_version="2.10.0+rocm7.0")
    for pat in SWEEP_PATTERNS:
        assert any(re.search(pat, ln) for ln in lines), pat
    sm = G["summarize_mode"]
    assert f"Baseline TPOT: {sm['avg_tpot_s']:.6f}" in lines          # both modes share the sample set here
    assert "Speculative profile total_profiled_cycles: 60" in lines
    hist = next(ln for ln in lines if ln.startswith("Acceptance length histogram:"))
    assert hist.count("%") == 17
    skipped = harness.stat_lines(responses, 16, baseline=False)
    assert "Decoding speedup: N/A (baseline skipped)" in skipped and not any("Baseline" in ln for ln in skipped)


def test_jsonl_rows(tmp_path):
    spec = SimpleNamespace(**{**G["samples"][0], "cycle_trace": [{"cycle_idx": 0, "tau": 3}]})
    row = harness.output_record(rank=0, dataset_row_idx=4, turn_index=0, dataset="synthetic", prompt_text="p",
                                input_text="i", block_size=16, draft_steps=1, speculative=spec, speculative_text="o")
    assert list(row) == ["rank", "dataset_row_idx", "turn_index", "dataset", "prompt_text", "input_text",
                         "block_size", "draft_steps", "baseline", "speculative"]
    assert list(row["speculative"]) == ["output_text", "num_input_tokens", "num_output_tokens", "wall_time_s",
                                        "ttft_s", "tpot_s", "acceptance_lengths", "profile_summary"]
    tr = list(harness.cycle_trace_records(spec, rank=0, dataset="synthetic", dataset_row_idx=4, turn_index=0,
                                          mode="speculative", block_size=16))
    assert tr == [{"rank": 0, "dataset": "synthetic", "dataset_row_idx": 4, "turn_index": 0, "mode": "speculative",
                   "block_size": 16, "cycle_idx": 0, "tau": 3}]
    harness.write_jsonl(tmp_path / "o" / "out.jsonl", [row])
    assert json.loads(open(tmp_path / "o" / "out.jsonl").read()) == row


@pytest.mark.gpu
def test_cli_stdout_feeds_the_sweep_script(tmp_path):
    """SURVEY.md §8f-3 end to end: `python -m dflash_amd.harness_cli` (benchmark.py's main on this
    package's loop: bs = 1 baseline + speculative run per prompt) on the GPU; run_block_sweep.sh's
    greps (:199-212) are applied to its REAL stdout with the script's own `grep -Eo ... | awk` forms,
    and the JSONL files hold the rows of benchmark.py:503-534 / :481-497."""
    import subprocess
    import sys
    out_p, tr_p = tmp_path / "o" / "out.jsonl", tmp_path / "o" / "trace.jsonl"
    r = subprocess.run([sys.executable, "-m", "dflash_amd.harness_cli", "--synthetic", "tiny", "--max-samples", "3",
                        "--max-new-tokens", "48", "--prompt-len", "40", "--scripted-tau", "5.0", "--collect-profile",
                        "--save-outputs-path", str(out_p), "--save-cycle-trace-path", str(tr_p)],
                       cwd=H.ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    for pat in SWEEP_PATTERNS:
        assert any(re.search(pat, ln) for ln in lines), (pat, r.stdout[-1500:])
    log = tmp_path / "bs16.log"
    log.write_text(r.stdout)

    def sweep_grep(expr, field):   # run_block_sweep.sh:199-204 verbatim form
        sh = f"(grep -Eo '{expr}' '{log}' || true) | tail -1 | awk '{{print ${field}}}'"
        return subprocess.run(["bash", "-c", sh], capture_output=True, text=True).stdout.strip()

    speedup = float(sweep_grep("Decoding speedup: [0-9.]+$", 3))
    tau = float(sweep_grep("Average Acceptance length: [0-9.]+$", 4))
    tps = float(sweep_grep("Speculative tokens_per_sec: [0-9.]+$", 3))
    base_tps = float(sweep_grep("Baseline tokens_per_sec: [0-9.]+$", 3))
    # (tokens_per_sec = tokens / wall time including TTFT, benchmark.py:255-260: for 48 tokens of a tiny model it is ruled by
    # one-off costs of the process, so only the TPOT-based speedup is compared here)
    assert speedup > 1.0 and 2.0 < tau < 9.0 and tps > 0 and base_tps > 0
    rows = [json.loads(ln) for ln in open(out_p)]
    assert len(rows) == 3 and [r_["dataset_row_idx"] for r_ in rows] == [0, 1, 2]
    for row in rows:
        assert list(row) == ["rank", "dataset_row_idx", "turn_index", "dataset", "prompt_text", "input_text",
                             "block_size", "draft_steps", "baseline", "speculative"]
        # lossless: the speculative text equals the bs = 1 baseline's (the target's own greedy continuation)
        assert row["speculative"]["output_text"] == row["baseline"]["output_text"]
        assert row["speculative"]["num_output_tokens"] == 48 and row["baseline"]["acceptance_lengths"] == [1] * 48
        assert row["speculative"]["profile_summary"]["profiled_cycles"] == len(row["speculative"]["acceptance_lengths"])
    tr = [json.loads(ln) for ln in open(tr_p)]
    spec = [t for t in tr if t["mode"] == "speculative"]
    assert {t["mode"] for t in tr} == {"baseline", "speculative"} and all(t["block_size"] in (1, 16) for t in tr)
    assert {"cycle_idx", "generated_tokens_before", "effective_block_size", "tau", "acceptance_ratio", "draft_s",
            "target_s", "cycle_s"} <= set(spec[0])
    assert sum(t["tau"] for t in spec if t["dataset_row_idx"] == 0) >= 48
