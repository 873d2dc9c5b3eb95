"""Harness wire format: the stat lines `benchmark.py:553-604` prints —
which `run_block_sweep.sh:199-212` greps into its CSV — the per-sample JSONL rows
(`benchmark.py:503-534`) and the per-cycle trace rows (`:481-497`), so that the existing
sweep scripts and analysis keep working on this package's output.

Host-only arithmetic; inputs are the namespaces `dflash_generate` returns (with the
harness-added `wall_time_s`)."""
from __future__ import annotations

import json
from itertools import chain
from typing import Iterable, Optional, Sequence

import numpy as np


def summarize_mode(samples: Sequence) -> dict:
    """benchmark.py:254-268: tokens_per_sec = total tokens / total wall."""
    walls = [float(s.wall_time_s) for s in samples]
    total_wall = float(np.sum(walls))
    total_tokens = int(np.sum([s.num_output_tokens for s in samples]))
    return {"total_wall_s": total_wall, "avg_wall_s": float(np.mean(walls)),
            "avg_ttft_s": float(np.mean([s.time_to_first_token for s in samples])),
            "avg_tpot_s": float(np.mean([s.time_per_output_token for s in samples])),
            "tokens_per_sec": float(total_tokens / max(total_wall, 1e-8)), "total_tokens": float(total_tokens)}


def summarize_profile(samples: Sequence) -> Optional[dict]:
    """benchmark.py:271-298."""
    ps = [p for p in (getattr(s, "profile_summary", None) for s in samples) if p is not None]
    if not ps:
        return None
    tot = {k: float(np.sum([p[k] for p in ps])) for k in
           ("target_prefill_s", "target_decode_s", "draft_decode_s", "cycle_decode_s_sum", "decode_wall_s")}
    cycles = int(np.sum([p["profiled_cycles"] for p in ps]))
    den = max(1e-12, tot["draft_decode_s"] + tot["target_decode_s"])
    n = len(ps)
    return {"total_target_prefill_s": tot["target_prefill_s"], "total_target_decode_s": tot["target_decode_s"],
            "total_draft_decode_s": tot["draft_decode_s"], "total_cycle_decode_s": tot["cycle_decode_s_sum"],
            "total_decode_wall_s": tot["decode_wall_s"], "total_profiled_cycles": float(cycles),
            "draft_share_decode": float(tot["draft_decode_s"] / den),
            "target_share_decode": float(tot["target_decode_s"] / den),
            "avg_target_prefill_s": float(tot["target_prefill_s"] / n),
            "avg_target_decode_s": float(tot["target_decode_s"] / n),
            "avg_draft_decode_s": float(tot["draft_decode_s"] / n),
            "avg_decode_wall_s": float(tot["decode_wall_s"] / n)}


def _profile_lines(prefix: str, prof: dict) -> list:
    return [f"{prefix} profile avg_target_prefill_s: {prof['avg_target_prefill_s']:.6f}",
            f"{prefix} profile avg_target_decode_s: {prof['avg_target_decode_s']:.6f}",
            f"{prefix} profile avg_draft_decode_s: {prof['avg_draft_decode_s']:.6f}",
            f"{prefix} profile target_share_decode: {prof['target_share_decode']:.4f}",
            f"{prefix} profile draft_share_decode: {prof['draft_share_decode']:.4f}",
            f"{prefix} profile total_profiled_cycles: {int(prof['total_profiled_cycles'])}"]


def stat_lines(responses: Sequence[dict], block_size: int, *, draft_steps: int = 1, baseline: bool = True,
               collect_profile: bool = False, gpu_name: str = "", runtime_version: Optional[str] = None,
               torch_version: str = "", world_size: int = 1) -> list:
    """The stdout block of benchmark.py:553-604.  `responses[i]` maps block size -> the
    namespace of that run (key 1 = baseline)."""
    spec = [r[block_size] for r in responses]
    sm = summarize_mode(spec)
    out = []
    bm = None
    if baseline:
        base = [r[1] for r in responses]
        bm = summarize_mode(base)
        out += [f"Baseline total_wall_s: {bm['total_wall_s']:.6f}", f"Baseline avg_wall_s: {bm['avg_wall_s']:.6f}",
                f"Baseline TTFT: {bm['avg_ttft_s']:.6f}", f"Baseline TPOT: {bm['avg_tpot_s']:.6f}",
                f"Baseline tokens_per_sec: {bm['tokens_per_sec']:.6f}"]
    out += [f"Speculative total_wall_s: {sm['total_wall_s']:.6f}", f"Speculative avg_wall_s: {sm['avg_wall_s']:.6f}",
            f"Speculative TTFT: {sm['avg_ttft_s']:.6f}", f"Speculative TPOT: {sm['avg_tpot_s']:.6f}",
            f"Speculative tokens_per_sec: {sm['tokens_per_sec']:.6f}"]
    out.append(f"Decoding speedup: {bm['avg_tpot_s'] / sm['avg_tpot_s']:.2f}" if baseline
               else "Decoding speedup: N/A (baseline skipped)")
    if collect_profile:
        sp = summarize_profile(spec)
        if sp is not None:
            out += _profile_lines("Speculative", sp)
        if baseline:
            bp = summarize_profile([r[1] for r in responses])
            if bp is not None:
                out += _profile_lines("Baseline", bp)
    tau = np.mean([np.mean(r[block_size].acceptance_lengths) for r in responses])
    out.append(f"Average Acceptance length: {tau:.2f}")
    acc = list(chain(*[r[block_size].acceptance_lengths for r in responses]))
    hist = [acc.count(b) / len(acc) for b in range(block_size + 1)]
    out.append(f"Acceptance length histogram: {[f'{x * 100:.1f}%' for x in hist]}")
    out += [f"Draft steps per cycle: {draft_steps}", f"Hardware GPU: {gpu_name}",
            f"Hardware CUDA: {runtime_version}",   # the reference prints torch.version.cuda here (None on ROCm)
            f"Hardware Torch: {torch_version}", f"Hardware World Size: {world_size}"]
    return out


def _mode_block(resp, output_text):
    return {"output_text": output_text, "num_input_tokens": resp.num_input_tokens,
            "num_output_tokens": resp.num_output_tokens, "wall_time_s": resp.wall_time_s,
            "ttft_s": resp.time_to_first_token, "tpot_s": resp.time_per_output_token,
            "acceptance_lengths": resp.acceptance_lengths, "profile_summary": resp.profile_summary}


def output_record(*, rank: int, dataset_row_idx: int, turn_index: int, dataset: str, prompt_text: str,
                  input_text: str, block_size: int, draft_steps: int, speculative, speculative_text: str,
                  baseline=None, baseline_text: Optional[str] = None) -> dict:
    """One row of --save-outputs-path (benchmark.py:503-534)."""
    return {"rank": rank, "dataset_row_idx": dataset_row_idx, "turn_index": turn_index, "dataset": dataset,
            "prompt_text": prompt_text, "input_text": input_text, "block_size": block_size,
            "draft_steps": draft_steps,
            "baseline": None if baseline is None else _mode_block(baseline, baseline_text),
            "speculative": _mode_block(speculative, speculative_text)}


def cycle_trace_records(resp, *, rank: int, dataset: str, dataset_row_idx: int, turn_index: int, mode: str,
                        block_size: int) -> Iterable[dict]:
    """Rows of --save-cycle-trace-path (benchmark.py:481-497)."""
    for row in getattr(resp, "cycle_trace", []):
        yield {"rank": rank, "dataset": dataset, "dataset_row_idx": dataset_row_idx, "turn_index": turn_index,
               "mode": mode, "block_size": int(block_size), **row}


def write_jsonl(path, rows) -> None:
    import os
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        for row in rows:
            f.write(json.dumps(row, ensure_ascii=False) + "\n")
