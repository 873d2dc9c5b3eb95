"""Host policy: `dflash_amd.scheduler.EWMAPerformanceScheduler` replays the decision
traces recorded from the reference (benchmark_dynamic_schedule.py:54-257)."""
import json
import os

import pytest

import helpers as H
from dflash_amd.scheduler import EWMAPerformanceScheduler

TRACES = json.load(open(os.path.join(H.GOLDEN, "scheduler.json")))


@pytest.mark.parametrize("idx", range(len(TRACES)))
def test_scheduler_trace(idx):
    tr = TRACES[idx]
    s = EWMAPerformanceScheduler(**tr["params"])
    for cyc, st in enumerate(tr["steps"]):
        assert s.select(cyc) == st["chosen"], cyc
        s.update(tau=st["tau"], cycle_s=st["cycle_s"], effective_bs=st["bs"], cycle_idx=cyc, l_gen=st["l_gen"])
        assert s.current == st["current"], cyc
        assert (s.cooldown_left, s.pending_target, s.pending_streak) == (
            st["cooldown_left"], st["pending_target"], st["pending_streak"]), cyc
        assert {str(k): v for k, v in s.tau_hat.items()} == st["tau_hat"], cyc
        assert {str(k): v for k, v in s.score_hat.items()} == st["score_hat"], cyc
        assert (s.adl_target_k, s.adl_target_bs, s.adl_lgen_hat, s.adl_lacc_hat) == (
            st["adl_target_k"], st["adl_target_bs"], st["adl_lgen_hat"], st["adl_lacc_hat"]), cyc


def test_scheduler_validation():
    base = dict(TRACES[0]["params"])
    for bad in (dict(scheduler_mode="x"), dict(ewma_alpha=0.0), dict(adl_rho=1.5), dict(adl_delta=-1.0),
                dict(adl_k_min=20, adl_k_max=10)):
        with pytest.raises(ValueError):
            EWMAPerformanceScheduler(**{**base, **bad})
