"""Host-side per-cycle block-size policy (stays Python, float64 host arithmetic).

Mirror of the reference's `EWMAPerformanceScheduler`
(benchmark_dynamic_schedule.py:54-257): same constructor keywords, same
`select(cycle_idx)` / `update(tau=, cycle_s=, effective_bs=, cycle_idx=, l_gen=)`
protocol and the same public attributes the harness logs per cycle
(`current`, `tau_hat`, `cycle_hat`, `score_hat`, `obs_count`, `adl_*`).
Decision traces are pinned against the reference in
tests/test_scheduler_golden.py.

The kernels behind `dflash_generate_policy` accept a different block size every
cycle, so nothing here touches the device.
"""
from __future__ import annotations

import math
from typing import Optional

_MODES = ("ewma", "adl_ewma")


def _blend(prev: Optional[float], obs: float, alpha: float) -> float:
    # first observation seeds the average (reference :124-127)
    return float(obs) if prev is None else float((1.0 - alpha) * prev + alpha * obs)


def _clip(x, lo, hi):
    return lo if x < lo else hi if x > hi else x


class EWMAPerformanceScheduler:
    def __init__(self, *, candidates, scheduler_mode, warmup_cycles, ewma_alpha, switch_margin,
                 required_streak, cooldown_cycles, probe_interval, low_accept_threshold,
                 low_accept_streak, adl_rho, adl_delta, adl_k_min, adl_k_max, adl_neighborhood):
        self.candidates = sorted(candidates)
        self.scheduler_mode = str(scheduler_mode)
        if self.scheduler_mode not in _MODES:
            raise ValueError("scheduler_mode must be one of {'ewma', 'adl_ewma'}.")
        self.ewma_alpha = float(ewma_alpha)
        if not 0.0 < self.ewma_alpha <= 1.0:
            raise ValueError("ewma_alpha must be in (0, 1].")
        self.adl_rho = float(adl_rho)
        if not 0.0 < self.adl_rho <= 1.0:
            raise ValueError("adl_rho must be in (0, 1].")
        self.adl_delta = float(adl_delta)
        if self.adl_delta < 0.0:
            raise ValueError("adl_delta must be >= 0.")
        self.adl_k_min, self.adl_k_max = int(adl_k_min), int(adl_k_max)
        if self.adl_k_min > self.adl_k_max:
            raise ValueError("adl_k_min must be <= adl_k_max.")

        # knobs, clamped exactly as the reference clamps them (:79-88,100)
        self.warmup_cycles = max(0, int(warmup_cycles))
        self.switch_margin = max(0.0, float(switch_margin))
        self.required_streak = max(1, int(required_streak))
        self.cooldown_cycles = max(0, int(cooldown_cycles))
        self.probe_interval = max(0, int(probe_interval))
        self.low_accept_threshold = float(low_accept_threshold)
        self.low_accept_streak = max(1, int(low_accept_streak))
        self.adl_neighborhood = max(0, int(adl_neighborhood))

        # state (:78,102-118): start on the largest candidate
        self.current = self.candidates[-1]
        self.cooldown_left = 0
        self.pending_target = self.current
        self.pending_streak = 0
        self.low_accept_count = 0
        self.last_probe_cycle = -1
        self.probe_cursor = 0
        self.tau_hat = dict.fromkeys(self.candidates)
        self.cycle_hat = dict.fromkeys(self.candidates)
        self.score_hat = dict.fromkeys(self.candidates)
        self.obs_count = dict.fromkeys(self.candidates, 0)
        self.adl_lgen_hat: Optional[float] = None
        self.adl_lacc_hat: Optional[float] = None
        self.adl_target_k = int(_clip(self.current, self.adl_k_min, self.adl_k_max))
        self.adl_target_bs = self._nearest_candidate(self.adl_target_k)

    # -- helpers ---------------------------------------------------------
    def _nearest_candidate(self, k: int) -> int:
        # ties go to the larger block (:141-142)
        return min(self.candidates, key=lambda b: (abs(b - k), -b))

    def _lower_neighbor(self, b: int) -> int:
        return self.candidates[max(0, self.candidates.index(b) - 1)]

    def _next_probe_candidate(self) -> int:
        n = len(self.candidates)
        for _ in range(n):
            b = self.candidates[self.probe_cursor % n]
            self.probe_cursor += 1
            if b != self.current:
                return b
        return self.current

    def _adl_candidate_pool(self):
        return {b for b in self.candidates if abs(b - self.adl_target_bs) <= self.adl_neighborhood}

    # -- protocol --------------------------------------------------------
    def select(self, cycle_idx: int) -> int:
        """Block size for this cycle (:148-160): round-robin during warm-up, a
        non-current candidate every `probe_interval` cycles after it, else `current`."""
        if cycle_idx < self.warmup_cycles:
            return self.candidates[cycle_idx % len(self.candidates)]
        if self.probe_interval > 0 and (cycle_idx - self.warmup_cycles) % self.probe_interval == 0:
            self.last_probe_cycle = cycle_idx
            return self._next_probe_candidate()
        return self.current

    def update(self, *, tau, cycle_s, effective_bs, cycle_idx, l_gen=None) -> None:
        """Feed back one cycle's acceptance and wall time (:162-257)."""
        tau, cycle_s, bs = float(tau), float(cycle_s), int(effective_bs)
        if bs not in self.tau_hat:  # clamped tail cycle: not a candidate, ignore (:174-177)
            return
        a = self.ewma_alpha
        self.tau_hat[bs] = _blend(self.tau_hat[bs], tau, a)
        self.cycle_hat[bs] = _blend(self.cycle_hat[bs], cycle_s, a)
        self.score_hat[bs] = float(self.tau_hat[bs] / max(1e-12, self.cycle_hat[bs]))
        self.obs_count[bs] += 1

        if self.scheduler_mode == "adl_ewma" and l_gen is not None:  # (:186-199)
            self.adl_lgen_hat = _blend(self.adl_lgen_hat, float(l_gen), self.adl_rho)
            self.adl_lacc_hat = _blend(self.adl_lacc_hat, tau, self.adl_rho)
            grow = self.adl_delta if self.adl_lacc_hat >= self.adl_lgen_hat else 0.0
            self.adl_target_k = int(_clip(math.ceil(self.adl_lgen_hat + grow), self.adl_k_min, self.adl_k_max))
            self.adl_target_bs = self._nearest_candidate(self.adl_target_k)

        # persistent low acceptance on the current size steps down at once (:201-215)
        if tau / max(1.0, float(bs)) < self.low_accept_threshold and bs == self.current:
            self.low_accept_count += 1
        else:
            self.low_accept_count = 0
        if self.low_accept_count >= self.low_accept_streak:
            lower = self._lower_neighbor(self.current)
            if lower != self.current:
                self.current = lower
                self.pending_target = lower
                self.pending_streak = 0
                self.cooldown_left = self.cooldown_cycles
            self.low_accept_count = 0

        if cycle_idx < self.warmup_cycles:
            return
        if self.cooldown_left > 0:
            self.cooldown_left -= 1
            return

        ranked = [(b, s) for b, s in self.score_hat.items() if s is not None]
        if not ranked:
            return
        if self.scheduler_mode == "adl_ewma":
            pool = self._adl_candidate_pool()
            near = [(b, s) for b, s in ranked if b in pool]
            ranked = near or ranked
        best_b, best_s = max(ranked, key=lambda bs_: bs_[1])  # first maximum wins, as in the reference
        cur_s = self.score_hat.get(self.current)
        if cur_s is None:
            cur_s = -math.inf
        gain = (best_s - cur_s) / max(1e-12, abs(cur_s))
        if best_b == self.current or not gain > self.switch_margin:
            self.pending_target, self.pending_streak = self.current, 0
            return
        if best_b == self.pending_target:
            self.pending_streak += 1
        else:
            self.pending_target, self.pending_streak = best_b, 1
        if self.pending_streak >= self.required_streak:
            self.current = best_b
            self.pending_streak = 0
            self.cooldown_left = self.cooldown_cycles
