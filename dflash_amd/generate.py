"""Decode loops over the HIP hot path, with the reference's call signatures and
return fields:

* `dflash_generate`        — benchmark.py:44-251 (fixed block, tail clamp,
                             block_size==1 baseline, draft_steps, per-cycle profile)
* `dflash_generate_policy` — benchmark_dynamic_schedule.py:260-434 (block size from
                             `EWMAPerformanceScheduler` every cycle)
* `DFlashDraftModel.spec_generate` (model.py) — model/dflash.py:192-277

All three drive a `DecodeSession` (prefill once, then `cycle()` per block), which
bench.py also steps directly.  Per cycle the host enqueues: draft forward + fused
lm_head/argmax (kernels), the caller's target forward (PyTorch-ROCm, outside the
path), the posterior argmax kernel and the accept/commit kernel; the only
device->host read is the 4-int accept result the loop needs to slice the
target's hidden states and roll its cache back.
"""
from __future__ import annotations

import os
import time
from types import SimpleNamespace
from typing import Callable, Optional

import torch

from . import ops
from .utils import extract_context_feature, sample

_TABLES = {}
MAX_BLOCK = 32  # rows per block the kernels take: two 16-row tiles


def cuda_time() -> float:
    torch.cuda.synchronize()
    return time.perf_counter()


def capture_graph(fn) -> "torch.cuda.CUDAGraph":
    """Capture fn()'s launches into a hipGraph WITHOUT `torch.cuda.graph()`'s preamble.  That context manager calls
    `torch.cuda.empty_cache()` before it begins a capture: every block in torch's caching allocator goes back to the
    driver — with a NativeTarget(keep_hf=False) that is the ~17 GB of the wrapped model's dropped weights — and the driver
    clears released VRAM in the background, on the GPU, for the next few hundred milliseconds.  Every kernel that runs
    meanwhile is 2 - 4 % slower (measured, round 4: DESIGN.md section 5, profiles/r4_graph_ab.txt; rounds 2 - 3 read this
    as "kernels are slower once a graph exists").  Nothing here needs the cache emptied: the launches allocate nothing."""
    g = torch.cuda.CUDAGraph()
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        g.capture_begin()
        try:
            fn()
        finally:
            g.capture_end()
    cur.wait_stream(side)
    return g


def _new_target_cache(target):
    if hasattr(target, "new_cache"):
        return target.new_cache()
    from transformers import DynamicCache  # the reference's choice (model/dflash.py:214)
    return DynamicCache()


def _taps(hidden_states, layer_ids):
    """extract_context_feature (model/utils.py:16-25); the kernels take bf16 rows, which
    is what a bf16 target (the reference's dtype, benchmark.py:404) already returns."""
    t = extract_context_feature(hidden_states, layer_ids)
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


def _bf16_table(w, dev):
    if w.dtype == torch.bfloat16 and w.device == dev and w.is_contiguous():
        return w.detach()
    key = (w.data_ptr(), tuple(w.shape), w._version)
    if key not in _TABLES:
        _TABLES.clear()
        _TABLES[key] = w.detach().to(device=dev, dtype=torch.bfloat16).contiguous()
    return _TABLES[key]


def _trim(output_ids, max_length, mask_token_id, stop_token_ids, n_in):
    """model/dflash.py:269-275."""
    output_ids = output_ids[:, :max_length]
    output_ids = output_ids[:, output_ids[0] != mask_token_id]
    if stop_token_ids is not None:
        st = torch.tensor(stop_token_ids, device=output_ids.device)
        idx = torch.isin(output_ids[0][n_in:], st).nonzero(as_tuple=True)[0]
        if idx.numel() > 0:
            output_ids = output_ids[:, : n_in + idx[0] + 1]
    return output_ids


class DecodeSession:
    """State of one request between cycles (what the reference keeps in local
    variables of its while-loop, model/dflash.py:206-233)."""

    def __init__(self, model, target, input_ids: torch.Tensor, *, mask_token_id: int, max_new_tokens: int,
                 max_block_size: int, stop_token_ids, temperature: float, draft_temperature: float = 0.0,
                 draft_token_hook: Optional[Callable] = None):
        dev = model.device
        if not input_ids.is_cuda:
            raise RuntimeError("dflash_amd: input_ids must be on the GPU")
        if input_ids.shape[0] != 1:
            raise NotImplementedError("batch = 1 per call, as in the reference; shard requests over ranks")
        if not 1 <= int(max_block_size) <= MAX_BLOCK:
            raise ValueError(f"block size {max_block_size}: the gfx950 kernels take blocks of 1..{MAX_BLOCK} rows "
                             f"(two 16-row tiles); checked here, before the target prefill runs")
        self.model, self.target, self.dev = model, target, dev
        self.input_ids = input_ids
        self.mask_token_id, self.temperature = mask_token_id, temperature
        self.draft_temperature, self.hook = draft_temperature, draft_token_hook
        self.stop_token_ids = stop_token_ids
        self.max_bs = max_block_size
        self.n_in = input_ids.shape[1]
        self.max_length = self.n_in + max_new_tokens
        self.output_ids = torch.full((1, self.max_length + self.max_bs), mask_token_id, dtype=torch.long, device=dev)
        self.position_ids = torch.arange(self.output_ids.shape[1], device=dev).unsqueeze(0)
        from .target import NativeTarget
        self.native = isinstance(target, NativeTarget)
        self.tcache = (target.new_cache(self.max_length + 2 * self.max_bs) if self.native
                       else _new_target_cache(target))
        self.use_draft = self.max_bs > 1
        self.dcache = model.new_cache(self.max_length + 2 * self.max_bs) if self.use_draft else None
        self.dyn = self.dcache.dyn if self.dcache is not None else torch.zeros(8, dtype=torch.int32, device=dev)
        self.embed_w = _bf16_table(target.model.embed_tokens.weight, dev)
        self.lm_wp = model.packed_lm_head(target.lm_head) if self.use_draft else None
        if self.native and self.lm_wp is not None:
            target.share_lm_head(self.lm_wp)  # one packed copy serves draft unmask and target posterior
        self.stop_t = torch.tensor(stop_token_ids, dtype=torch.long, device=dev) if stop_token_ids else None
        # the reference scans the whole buffer, mask slots included (model/dflash.py:265-268)
        self.stop_always = stop_token_ids is not None and mask_token_id in stop_token_ids
        # the accept result {acc, new start, stop, cycle}: pinned host memory the kernel writes with one 16-byte store and
        # this thread polls (no blit kernel, no interrupt-driven wake); DFL_HOST_RESULT=0: a device buffer + .tolist()
        self.poll_result = os.environ.get("DFL_HOST_RESULT", "1") != "0"
        self.result = (torch.zeros(4, dtype=torch.int32).pin_memory() if self.poll_result
                       else torch.zeros(4, dtype=torch.int32, device=dev))
        self._res_np = self.result.numpy() if self.poll_result else None
        self.block = torch.empty(1, self.max_bs, dtype=torch.long, device=dev)
        self._dyn_bs = None   # block size for which the draft cache's length record is already armed (see _draft)
        # run-ahead draft (DFL_RUN_AHEAD=0 turns it off): the NEXT cycle's draft forward is enqueued right behind the
        # accept kernel, before the host has read the acceptance length — every length it needs is in the device record
        # the accept kernel has just written — so the GPU does not idle through the host's turnaround (result poll,
        # bookkeeping, ~30 launches: 0.04-0.08 ms per cycle).  _ahead = block size of a draft already in flight.
        self.run_ahead = os.environ.get("DFL_RUN_AHEAD", "1") != "0"
        self._ahead = None
        self._ahead_ev = None
        self._armed = False   # True: the accept kernel has written the next cycle's block (bonus token + mask ids)
        self.start = self.n_in
        self.target_hidden = None
        self.hook_calls = 0
        self.stopped = False
        self.events = None  # set to a dict to have cycle() record (start, end) event pairs per phase
        self.record_events = True  # False: record nothing new, but still hand a run-ahead draft's pairs to self.events
        self.host_times = None  # a list: cycle() appends (seconds enqueueing launches, seconds polling for the result)
        self.draft_logits = None  # set to a bf16 [32, V] buffer to have the draft's logits materialised into it
        # the tapped rows live here from a verify to the next draft (own buffer: several
        # sessions may be interleaved on one NativeTarget)
        self.taps_buf = (torch.zeros(32, len(model.target_layer_ids) * model.config.hidden_size, dtype=torch.bfloat16,
                                     device=dev) if (self.native and self.use_draft) else None)

    @torch.inference_mode()
    def prefill(self) -> None:
        """model/dflash.py:218-229."""
        if self.native:
            out = self.target.prefill(self.input_ids, self.tcache, output_hidden_states=self.use_draft,
                                      tap_layers=self.model.target_layer_ids if self.use_draft else None)
        else:
            out = self.target(self.input_ids, position_ids=self.position_ids[:, :self.n_in],
                              past_key_values=self.tcache, use_cache=True, logits_to_keep=1,
                              output_hidden_states=self.use_draft)
        self.output_ids[:, :self.n_in] = self.input_ids
        self.output_ids[:, self.n_in:self.n_in + 1] = sample(out.logits, self.temperature)
        if self.use_draft:
            self.target_hidden = _taps(out.hidden_states, self.model.target_layer_ids)

    def _mark(self, key, which):
        if self.events is not None and self.record_events:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.events.setdefault(key, [None, None])[which] = e

    def _draft_ahead(self, bs) -> None:
        """The draft forward of the cycle AFTER the one whose accept kernel has just been enqueued (model/dflash.py:237-247
        of the next loop iteration): lengths from the device record, context rows = the verify's tap buffer (valid count
        = the record's tau), block = what the accept kernel re-armed.  Upper bounds for the host-side checks only."""
        m = self.model
        ev = None
        if self.events is not None and self.record_events:   # the caller times the phases: this draft's pairs belong to the NEXT cycle
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        hid = m.draft_block(self.dcache, th_rows=self.taps_buf[:16], tau=16, bs=bs, pos0=self.start + self.max_bs,
                            block_ids=self.block[0], embed=self.embed_w, dyn_ready=True, s_bound=self.start + self.max_bs)
        if ev:
            ev[2].record()
        _draft_ids(m, hid, self.lm_wp, bs, self.block[:, :bs], 0.0, self.draft_logits)
        if ev:
            ev[3].record()
            ev[1].record()
        self._ahead, self._ahead_ev = bs, ev

    def _draft(self, blk, bs, draft_steps):
        m = self.model
        ctx = self.target_hidden[0]
        if draft_steps == 1 and self._ahead == bs:   # enqueued behind the previous cycle's accept kernel already
            self._ahead = None
            self.dcache.length = self.start           # rows kept (S) + this cycle's context rows (tau) = the new start
            self._dyn_bs = None
            if self.events is not None and self._ahead_ev:
                e = self._ahead_ev
                self.events["draft"], self.events["lm_head"] = [e[0], e[1]], [e[2], e[3]]
            if self.hook is not None:
                self.hook(blk, self.start, self.hook_calls)
            self.hook_calls += 1
            return
        # (a run-ahead draft of another block size is simply abandoned: it touched scratch buffers, the draft tokens of
        # the block and the draft cache rows this forward writes again — the length record is the accept kernel's)
        self._ahead = None
        if draft_steps == 1:
            S = self.dcache.get_seq_length()
            head_rows = max(0, ctx.shape[0] - 16)
            if ctx.shape[0] > 16:  # cycle 0: the prompt's context rows, 16 at a time
                head = ctx.shape[0] - 16
                m.prefill_context(self.dcache, ctx[:head], S)
                ctx, S = ctx[head:], S + head
            # steady state: the accept kernel of the previous cycle wrote S, tau, pos0 and start of THIS forward into
            # the draft cache's record (dfl_accept_commit); only a changed block size or a context head needs new words
            ready = self._dyn_bs == bs and head_rows == 0
            hid = m.draft_block(self.dcache, th_rows=ctx, tau=ctx.shape[0], bs=bs, pos0=S, block_ids=blk[0],
                                embed=self.embed_w, dyn_ready=ready)
            self._dyn_bs = None   # until the accept of this cycle has re-armed the record
            self._mark("lm_head", 0)
            _draft_ids(m, hid, self.lm_wp, bs, blk, self.draft_temperature, self.draft_logits)
            self._mark("lm_head", 1)
            if self.hook is not None:
                self.hook(blk, self.start, self.hook_calls)
            self.hook_calls += 1
            return
        # benchmark.py:112-142: k full passes, each re-embedding the whole block, no draft
        # cache, positions rebuilt as [start-ctx_len, start+bs)
        ctx_len = ctx.shape[0]
        for _ in range(draft_steps):
            tmp = m.new_cache(ctx_len + bs)
            p0 = max(0, self.start - ctx_len)
            c2 = ctx
            if ctx_len > 16:
                m.prefill_context(tmp, ctx[:ctx_len - 16], p0)
                c2, p0 = ctx[ctx_len - 16:], p0 + ctx_len - 16
            hid = m.draft_block(tmp, th_rows=c2, tau=c2.shape[0], bs=bs, pos0=p0, block_ids=blk[0],
                                embed=self.embed_w, append=False)
            _draft_ids(m, hid, self.lm_wp, bs, blk, 0.0)
            if self.hook is not None:
                self.hook(blk, self.start, self.hook_calls)
            self.hook_calls += 1

    @torch.inference_mode()
    def cycle(self, bs: int, *, draft_steps: int = 1, want_hidden: Optional[bool] = None,
              after_draft: Optional[Callable] = None, ahead_ok: bool = False) -> SimpleNamespace:
        """One pass of model/dflash.py:235-268 with block size `bs` (>= 1).  ahead_ok: the caller promises that the next
        cycle uses the same block size whenever the tail allows it (fixed-size loops): its draft may be enqueued early."""
        t_call = time.perf_counter() if self.host_times is not None else 0.0
        start = self.start
        blk = self.block[:, :bs]
        if not self._armed:   # model/dflash.py:235; afterwards dfl_accept_commit_rearm leaves the same ids in self.block
            blk.copy_(self.output_ids[:, start:start + bs])
        if bs > 1:
            ahead = draft_steps == 1 and self._ahead == bs
            if not ahead:
                self._mark("draft", 0)
            self._draft(blk, bs, draft_steps)
            if not ahead:
                self._mark("draft", 1)
            if after_draft is not None:
                after_draft(blk)
        if want_hidden is None:
            want_hidden = self.use_draft
        # ---- target verify (outside the path; model/dflash.py:249-255)
        self._mark("target", 0)
        taps = None
        if self.native:
            posterior, taps = self.target.verify(
                blk[0], start, self.tcache, temperature=self.temperature,
                tap_layers=self.model.target_layer_ids if (want_hidden and self.use_draft) else (),
                taps_out=self.taps_buf)
            self._mark("target", 1)
        else:
            out = self.target(blk, position_ids=self.position_ids[:, start:start + bs],
                              past_key_values=self.tcache, use_cache=True, output_hidden_states=want_hidden)
            self._mark("target", 1)
            posterior = sample(out.logits, self.temperature)
        # ---- accept scan + commit + bookkeeping on the device (:258-268)
        if not (bs > 1 and draft_steps == 1):
            # (after a cached draft forward the record already holds start = pos0 + tau: either draft_block's
            # dfl_set_dyn2 wrote it — which also clears the stop / cycle words — or, on the dyn_ready steady state, the
            # previous cycle's accept kernel did.  STOP is sticky and CYCLE keeps counting across steady-state cycles:
            # a stopped session is terminal, run_decode leaves the loop on it)
            ops.set_dyn(self.dyn, 0, 0, bs, start)  # start word = pos0 + tau = start
        if self.poll_result:
            self._res_np[3] = -1   # the cycle counter (>= 1 once written) is the kernel's LAST store, behind a release
        ops.accept_commit(blk[0], posterior[0].contiguous(), bs, self.output_ids[0], self.dyn, self.stop_t,
                          self.result, rearm=(self.block[0], self.max_bs, self.mask_token_id),
                          dyn_t=self.tcache.dyn if (self.native and getattr(self, "_graph_bs", None)) else None)
        self._armed = True
        # the record now holds the next draft forward's S / tau / pos0 / start — if that cycle was a cached draft cycle
        # on this record (bs > 1, one draft step) and keeps the block size
        self._dyn_bs = bs if (bs > 1 and draft_steps == 1 and self.use_draft) else None
        # run-ahead: the next cycle's draft goes out now if it is certain to be a draft cycle of the same block size —
        # no tail clamp even if every token of this block is accepted, no stop tokens to end the request early, a
        # caller-driven block size (a scheduler picks the next size only after this cycle's result: `ahead_ok`)
        if (ahead_ok and self.run_ahead and self._dyn_bs == bs and self.native and want_hidden and self.stop_t is None
                and getattr(self.model, "attn_impl", "head") == "head"   # (the round-1 stage takes no device-driven lengths)
                and not self.stop_always and self.draft_temperature < 1e-5 and bs <= 16
                and start + 2 * bs <= self.max_length and start + self.max_bs + 16 + bs <= self.dcache.max_rows):
            self._draft_ahead(bs)
        t_enq = time.perf_counter() if self.host_times is not None else 0.0
        if self.poll_result:   # the cycle's one device->host hand-over: pinned memory, polled on the word written last
            t0 = time.perf_counter()
            while self._res_np[3] == -1:
                if time.perf_counter() - t0 > 0.05:   # a long verify: stop burning the core, block on the stream
                    torch.cuda.current_stream().synchronize()
                    if self._res_np[3] == -1:
                        raise RuntimeError("dfl_accept_commit: the result never arrived in host memory")
            res = self._res_np.tolist()   # words 0..2 were stored before the counter was released: read them now
        else:
            res = self.result.tolist()  # device->host copy (synchronises the stream)
        if self.host_times is not None:   # host share of the cycle: launches enqueued, then waiting for the GPU
            self.host_times.append((t_enq - t_call, time.perf_counter() - t_enq))
        tau = res[0] + 1
        self.start = start + tau
        self.tcache.crop(self.start)
        if want_hidden and self.use_draft:
            self.target_hidden = (taps[None, :tau] if self.native
                                  else _taps(out.hidden_states, self.model.target_layer_ids)[:, :tau, :])
        self.stopped = bool(self.stop_always or res[2])
        return SimpleNamespace(tau=tau, bs=bs, start=start, stop=self.stopped)

    # ------------------------------------------------------------------ hipGraph replay of the steady-state cycle
    def _graph_ok(self, bs: int) -> bool:
        """A cycle that can be replayed: fixed block size, T = 0, no stop ids, no tail clamp even if every token of this
        block and the next is accepted, and the previous cycle left this cycle's draft enqueued (run-ahead)."""
        if not (self.native and self.use_draft and bs == getattr(self, "_graph_bs", None) and self._ahead == bs
                and self.stop_t is None and not self.stop_always and self.temperature < 1e-5
                and (self.events is None or not self.record_events)   # (a dict that only receives a run-ahead draft's pairs is fine)
                and self.start + 2 * bs <= self.max_length and self.start + bs <= self._graph_bound):
            return False
        # The graphs hold RAW device pointers.  A shared draft model / NativeTarget replaces its RoPE table when another
        # caller needs more positions (`_rope_tab`), and several sessions may interleave on one target: a replaced table
        # means the captured launches would read freed (possibly reused) memory.  The session keeps the captured
        # tensors alive (`_graph_keep`), so the replay stays memory-safe in any case; a differing pointer only says the
        # owner has moved on to a larger table — its first rows are identical, the replay is still right, but the
        # session re-captures at the next opportunity instead of pinning the old table for ever.
        m, t = self.model, self.target
        if m._rope is None or t._rope is None or (m._rope[0].data_ptr(), t._rope[0].data_ptr()) != self._graph_rope:
            self._graph_bs = None      # run_decode captures again on the next replayable cycle; this one runs eagerly
            self._graphs, self._graph_keep = {}, None
            return False
        return True

    @torch.inference_mode()
    def capture(self, bs: int) -> None:
        """Capture the steady-state cycle of block size bs (2..16) into two hipGraphs: [target verify + accept] and
        [draft forward + lm_head of the NEXT cycle].  Every length the launches need is in the device records the
        accept kernel keeps (dfl_accept_commit_rearm_t: the draft cache's record and the target cache's block-form
        record), so one capture serves every cycle; the attention launches size their key splits for the caches'
        capacity.  Call after at least one cycle(bs, ahead_ok=True).  The host then spends two graph launches and one poll
        per cycle instead of ~215 ctypes calls."""
        if not (self.native and self.use_draft and 2 <= bs <= 16 and self.temperature < 1e-5 and self.stop_t is None
                and self.draft_temperature < 1e-5 and self.target.attn_impl == "head" and not self.target.fuse_oproj):
            raise ValueError("capture needs a NativeTarget ('head' attention), a draft, T = 0, no stop ids, 2 <= bs <= 16")
        if self._ahead != bs:
            raise RuntimeError("capture: run one cycle(bs, ahead_ok=True) first (the next draft must be in flight)")
        m, t = self.model, self.target
        # upper bounds of S / start for the replayed launches (they size key splits and RoPE tables, and must pass the
        # launchers' capacity checks): a replayed cycle has start <= max_length - 2 bs (_graph_ok)
        bound = min(self.max_length, self.dcache.max_rows - 16 - bs, self.tcache.max_rows - bs)
        if bound < self.start:
            raise RuntimeError("capture: the caches leave no room for a replayed cycle")
        self._graph_bound = bound
        m._rope_tab(bound + 64 + 64)
        t._rope_tab(bound + 64 + 64)
        # the target cache's record: block form for the cycle about to run (the accept kernel keeps it from here on)
        ops.set_dyn2(self.tcache.dyn, self.start, 0, bs, self.start)
        self.tcache._dyn_bs = bs
        torch.cuda.synchronize(self.dev)
        tl = self.model.target_layer_ids

        def verify_accept():
            post, _ = t.verify(self.block[0, :bs], bound, self.tcache, temperature=0.0, tap_layers=tl,
                               taps_out=self.taps_buf, dyn_lengths=True)
            ops.accept_commit(self.block[0, :bs], post[0].contiguous(), bs, self.output_ids[0], self.dyn, None, self.result,
                              rearm=(self.block[0], self.max_bs, self.mask_token_id), dyn_t=self.tcache.dyn)

        def draft_next():
            hid = m.draft_block(self.dcache, th_rows=self.taps_buf[:16], tau=16, bs=bs, pos0=bound, block_ids=self.block[0],
                                embed=self.embed_w, dyn_ready=True, s_bound=bound)
            _draft_ids(m, hid, self.lm_wp, bs, self.block[:, :bs], 0.0, self.draft_logits)

        self._graphs = {}
        for name, fn in (("verify", verify_accept), ("draft", draft_next)):
            self._graphs[name] = capture_graph(fn)
        self._graph_bs = bs
        # every tensor whose address sits in the captured launches and that this session does not own through another
        # attribute: the RoPE tables and the shared workspaces of the draft model and of the target
        self._graph_rope = (m._rope[0].data_ptr(), t._rope[0].data_ptr())
        self._graph_keep = (m._rope, t._rope, getattr(m, "ws", None), getattr(t, "ws", None), getattr(t, "_taps", None),
                            self.lm_wp, self.embed_w, getattr(t, "lm_wp", None))

    @torch.inference_mode()
    def cycle_graph(self, bs: int) -> SimpleNamespace:
        """One pass of model/dflash.py:235-268 by graph replay; falls back to cycle(bs, ahead_ok=True) where the cycle
        cannot be replayed (tail of the request, events being recorded)."""
        if not self._graph_ok(bs):
            return self.cycle(bs, ahead_ok=True)
        t_call = time.perf_counter() if self.host_times is not None else 0.0
        start = self.start
        # the draft of THIS cycle is in flight already (run-ahead): host bookkeeping of _draft()
        self._ahead = None
        self.dcache.length = start
        if self.events is not None and self._ahead_ev:   # its event pairs were recorded by the (eager) cycle that enqueued it
            e = self._ahead_ev
            self.events["draft"], self.events["lm_head"] = [e[0], e[1]], [e[2], e[3]]
        if self.hook is not None:
            self.hook(self.block[:, :bs], start, self.hook_calls)
        self.hook_calls += 1
        if self.poll_result:
            self._res_np[3] = -1
        self._graphs["verify"].replay()
        self._armed = True
        self._dyn_bs = bs
        self._graphs["draft"].replay()          # the NEXT cycle's draft, behind the accept kernel
        self._ahead, self._ahead_ev = bs, None
        t_enq = time.perf_counter() if self.host_times is not None else 0.0
        if self.poll_result:
            t0 = time.perf_counter()
            while self._res_np[3] == -1:
                if time.perf_counter() - t0 > 0.05:
                    torch.cuda.current_stream().synchronize()
                    if self._res_np[3] == -1:
                        raise RuntimeError("dfl_accept_commit: the result never arrived in host memory")
            res = self._res_np.tolist()
        else:
            res = self.result.tolist()
        if self.host_times is not None:
            self.host_times.append((t_enq - t_call, time.perf_counter() - t_enq))
        tau = res[0] + 1
        self.start = start + tau
        self.tcache.crop(self.start)
        self.target_hidden = self.taps_buf[None, :tau]
        self.stopped = bool(res[2])
        return SimpleNamespace(tau=tau, bs=bs, start=start, stop=self.stopped)

    # ------------------------------------------------------------------ hipGraph replay with a caller-chosen block size
    @torch.inference_mode()
    def capture_sizes(self, sizes) -> None:
        """One pair of hipGraphs per block size in `sizes` (2..16 each) for loops whose caller picks the size every cycle —
        the policy loop of benchmark_dynamic_schedule.py:319-379: [draft forward + lm_head] and [target verify + accept],
        every launch taking its lengths from the two device records.  cycle_sized() writes those records from the host's
        own bookkeeping (two 64-thread launches) and replays the pair; there is no run-ahead draft, since the next block's
        size is only known once the scheduler has seen this cycle's result.  Call after at least one draft cycle."""
        sizes = sorted({int(b) for b in sizes})
        if not (self.native and self.use_draft and sizes and 2 <= sizes[0] and sizes[-1] <= 16 and self.temperature < 1e-5
                and self.stop_t is None and self.draft_temperature < 1e-5 and self.target.attn_impl == "head"
                and not self.target.fuse_oproj and getattr(self.model, "attn_impl", "head") == "head"):
            raise ValueError("capture_sizes needs a NativeTarget ('head' attention), a draft, T = 0, no stop ids, sizes in 2..16")
        if not self._armed:
            raise RuntimeError("capture_sizes: run one cycle first")
        m, t = self.model, self.target
        bound = min(self.max_length, self.dcache.max_rows - 16 - sizes[-1], self.tcache.max_rows - sizes[-1])
        if bound < self.start:
            raise RuntimeError("capture_sizes: the caches leave no room for a replayed cycle")
        m._rope_tab(bound + 64 + 64)
        t._rope_tab(bound + 64 + 64)
        torch.cuda.synchronize(self.dev)
        tl = self.model.target_layer_ids
        graphs = {}
        for bs in sizes:
            def draft_now(bs=bs):
                hid = m.draft_block(self.dcache, th_rows=self.taps_buf[:16], tau=16, bs=bs, pos0=bound, block_ids=self.block[0],
                                    embed=self.embed_w, dyn_ready=True, s_bound=bound)
                _draft_ids(m, hid, self.lm_wp, bs, self.block[:, :bs], 0.0, self.draft_logits)

            def verify_accept(bs=bs):
                post, _ = t.verify(self.block[0, :bs], bound, self.tcache, temperature=0.0, tap_layers=tl,
                                   taps_out=self.taps_buf, dyn_lengths=True)
                ops.accept_commit(self.block[0, :bs], post[0].contiguous(), bs, self.output_ids[0], self.dyn, None, self.result,
                                  rearm=(self.block[0], self.max_bs, self.mask_token_id), dyn_t=self.tcache.dyn)

            graphs[bs] = (capture_graph(draft_now), capture_graph(verify_accept))
        self.tcache.crop(self.start)     # (verify's host-side bookkeeping ran with the bound)
        self._sgraphs, self._sgraph_bound = graphs, bound
        self._sgraph_rope = (m._rope[0].data_ptr(), t._rope[0].data_ptr())
        self._sgraph_keep = (m._rope, t._rope, getattr(m, "ws", None), getattr(t, "ws", None), getattr(t, "_taps", None),
                             self.lm_wp, self.embed_w, getattr(t, "lm_wp", None))

    def _sized_ok(self, bs: int) -> bool:
        g = getattr(self, "_sgraphs", None)
        if not (g and bs in g and self._armed and self.stop_t is None and not self.stop_always
                and (self.events is None or not self.record_events) and self.target_hidden is not None
                and self.target_hidden.shape[1] <= 16 and self.start + bs <= min(self.max_length, self._sgraph_bound)):
            return False
        m, t = self.model, self.target
        if m._rope is None or t._rope is None or (m._rope[0].data_ptr(), t._rope[0].data_ptr()) != self._sgraph_rope:
            self._sgraphs = None         # a replaced RoPE table: the captured launches hold the old one's address
            return False
        return True

    @torch.inference_mode()
    def cycle_sized(self, bs: int, *, after_draft: Optional[Callable] = None) -> SimpleNamespace:
        """One pass of model/dflash.py:235-268 with the caller's block size by graph replay (capture_sizes); falls back
        to cycle(bs) where the cycle cannot be replayed (a size that was not captured — the clamped tail —, events being
        recorded, cycle 0's prompt-length context)."""
        if not self._sized_ok(bs):
            return self.cycle(bs, after_draft=after_draft)
        t_call = time.perf_counter() if self.host_times is not None else 0.0
        start = self.start
        tau_c = int(self.target_hidden.shape[1])
        S = self.dcache.get_seq_length()
        # the two length records, exactly as the eager launches would write them (model.draft_block, NativeTarget.verify)
        ops.set_dyn2(self.dcache.dyn, S, tau_c, bs, S)
        ops.set_dyn2(self.tcache.dyn, start, 0, bs, start)
        self.tcache._dyn_bs = bs
        self._ahead = None
        dg, vg = self._sgraphs[bs]
        dg.replay()
        self.dcache.length = S + tau_c
        blk = self.block[:, :bs]
        if self.hook is not None:
            self.hook(blk, start, self.hook_calls)
        self.hook_calls += 1
        if after_draft is not None:
            after_draft(blk)
        if self.poll_result:
            self._res_np[3] = -1
        vg.replay()
        self._armed = True
        self._dyn_bs = bs
        t_enq = time.perf_counter() if self.host_times is not None else 0.0
        if self.poll_result:
            t0 = time.perf_counter()
            while self._res_np[3] == -1:
                if time.perf_counter() - t0 > 0.05:
                    torch.cuda.current_stream().synchronize()
                    if self._res_np[3] == -1:
                        raise RuntimeError("dfl_accept_commit: the result never arrived in host memory")
            res = self._res_np.tolist()
        else:
            res = self.result.tolist()
        if self.host_times is not None:
            self.host_times.append((t_enq - t_call, time.perf_counter() - t_enq))
        tau = res[0] + 1
        self.start = start + tau
        self.tcache.length = start + bs
        self.tcache.crop(self.start)
        self.target_hidden = self.taps_buf[None, :tau]
        self.stopped = bool(res[2])
        return SimpleNamespace(tau=tau, bs=bs, start=start, stop=self.stopped)

    def finish(self) -> torch.Tensor:
        return _trim(self.output_ids, self.max_length, self.mask_token_id, self.stop_token_ids, self.n_in)


@torch.inference_mode()
def run_decode(model, target, input_ids: torch.Tensor, *, mask_token_id: int, max_new_tokens: int,
               block_size: int, stop_token_ids, temperature: float, clamp_tail: bool,
               draft_steps: int = 1, collect_profile: bool = False, scheduler=None,
               draft_temperature: float = 0.0, draft_token_hook: Optional[Callable] = None,
               max_block_size: Optional[int] = None) -> SimpleNamespace:
    s = DecodeSession(model, target, input_ids, mask_token_id=mask_token_id, max_new_tokens=max_new_tokens,
                      max_block_size=max_block_size or block_size, stop_token_ids=stop_token_ids,
                      temperature=temperature, draft_temperature=draft_temperature,
                      draft_token_hook=draft_token_hook)
    t_prefill = cuda_time()
    s.prefill()
    time_to_first_token = cuda_time() - t_prefill

    decode_start = cuda_time()
    taus, used_bs, cycle_trace, lgens = [], [], [], []
    draft_prefill = True
    use_graphs = os.environ.get("DFL_GRAPH", "1") != "0" and s.native and getattr(target, "attn_impl", "") == "head" \
        and not getattr(target, "fuse_oproj", False)
    cyc = 0
    while s.start < s.max_length:
        cycle_t0 = cuda_time() if scheduler is not None else None
        if collect_profile:
            s.events = {}
            s._mark("cycle", 0)
        chosen = block_size if scheduler is None else scheduler.select(cyc)
        remaining = s.max_length - s.start
        bs = max(1, min(chosen, remaining)) if (clamp_tail or scheduler is not None) else chosen
        lg = [float(bs)]

        def after_draft(blk, lg=lg, bs=bs):
            nonlocal draft_prefill, decode_start
            # EOS-aware generated length of the policy loop (benchmark_dynamic_schedule.py:344-349)
            if scheduler is not None and s.stop_t is not None:
                pos = torch.isin(blk[0, 1:], s.stop_t).nonzero(as_tuple=True)[0]
                if pos.numel() > 0:
                    lg[0] = float(min(int(pos[0].item()) + 1, bs))
            if draft_prefill:
                # the TPOT clock restarts right after the FIRST draft call and before that cycle's target
                # forward (benchmark.py:145-147, benchmark_dynamic_schedule.py:352-354): it drops the
                # prompt-context projection, not cycle 0's verify
                draft_prefill = False
                decode_start = cuda_time()

        # hidden states: always in spec_generate / the policy loop, only for bs > 1 in the
        # harness form (benchmark.py:157)
        want_hidden = s.use_draft if (scheduler is not None or not clamp_tail) else bs > 1
        gen_before = s.start - s.n_in
        start_idx = s.start
        # fixed-size loops replay their steady-state cycles from two hipGraphs (DecodeSession.capture): the host's share
        # of a cycle drops from ~2.3 ms to ~0.2 ms at the same GPU time; DFL_GRAPH=0: eager launches
        if (use_graphs and scheduler is None and draft_steps == 1 and not collect_profile and bs == block_size
                and 2 <= bs <= 16 and want_hidden and not draft_prefill and s._ahead == bs and s.stop_t is None
                and temperature < 1e-5 and draft_temperature < 1e-5):
            if getattr(s, "_graph_bs", None) is None and not getattr(s, "_graph_off", False):
                try:
                    s.capture(bs)
                except (ValueError, RuntimeError):       # a session the graphs do not cover: eager cycles, as before
                    s._graph_off = True
            r = s.cycle_graph(bs)
        else:
            # (collect_profile: no run-ahead draft — its event pairs would be recorded during cycle N and handed to
            # cycle N + 1, outside that cycle's [cycle0, cycle1] marks; benchmark.py:149-160 times each phase inside its
            # own cycle)
            if (use_graphs and scheduler is not None and draft_steps == 1 and not collect_profile and want_hidden
                    and not draft_prefill and s.stop_t is None and temperature < 1e-5 and draft_temperature < 1e-5
                    and len(getattr(scheduler, "candidates", ())) > 0
                    and 2 <= min(scheduler.candidates) and max(scheduler.candidates) <= 16):
                # the policy loop by replay: one pair of graphs per candidate size (DecodeSession.capture_sizes)
                if getattr(s, "_sgraphs", None) is None and s._armed and not getattr(s, "_sgraphs_off", False):
                    try:
                        s.capture_sizes(scheduler.candidates)
                    except (ValueError, RuntimeError):   # a session the graphs do not cover: eager cycles, as before
                        s._sgraphs_off = True
                r = s.cycle_sized(bs, after_draft=after_draft)
            else:
                r = s.cycle(bs, draft_steps=draft_steps, want_hidden=want_hidden, after_draft=after_draft,
                            ahead_ok=scheduler is None and draft_steps == 1 and not collect_profile)
        taus.append(r.tau)
        used_bs.append(bs)
        lgens.append(lg[0])
        if scheduler is not None:
            cycle_s = cuda_time() - cycle_t0
            scheduler.update(tau=r.tau, cycle_s=cycle_s, effective_bs=bs, cycle_idx=cyc, l_gen=lg[0])
            cycle_trace.append({
                "cycle_idx": cyc, "start_idx": int(start_idx), "block_size": int(bs),
                "chosen_block_size": int(chosen), "tau": int(r.tau), "l_gen": float(lg[0]),
                "acceptance_ratio": float(r.tau / max(1, bs)), "cycle_s": float(cycle_s),
                "tau_hat": scheduler.tau_hat.get(bs), "cycle_hat": scheduler.cycle_hat.get(bs),
                "score_hat": scheduler.score_hat.get(bs), "current_block_size": int(scheduler.current),
                "adl_lgen_hat": scheduler.adl_lgen_hat, "adl_lacc_hat": scheduler.adl_lacc_hat,
                "adl_target_k": int(scheduler.adl_target_k), "adl_target_bs": int(scheduler.adl_target_bs)})
        elif collect_profile:
            s._mark("cycle", 1)
            cycle_trace.append({"cycle_idx": cyc, "generated_tokens_before": int(gen_before),
                                "effective_block_size": int(bs), "tau": int(r.tau),
                                "acceptance_ratio": float(r.tau / max(1, bs)), "_events": s.events})
        s.events = None
        cyc += 1
        if r.stop:
            break

    output_ids = s.finish()
    for m in (model, target):   # opt-in fused launches leave a flag instead of hanging when their wait runs out
        if getattr(m, "fuse_oproj", False):
            m.raise_if_failed()
    num_output_tokens = output_ids.shape[1] - s.n_in
    total_decode_time = cuda_time() - decode_start
    profile_summary = None
    if collect_profile and scheduler is None:
        profile_summary = _resolve_profile(cycle_trace, time_to_first_token, total_decode_time)
    return SimpleNamespace(output_ids=output_ids, num_input_tokens=s.n_in, num_output_tokens=num_output_tokens,
                           time_to_first_token=time_to_first_token,
                           time_per_output_token=total_decode_time / max(1, num_output_tokens),
                           acceptance_lengths=taus, used_block_sizes=used_bs, l_gen=lgens,
                           cycle_trace=cycle_trace, profile_summary=profile_summary)


def _draft_ids(model, hid_frag, lm_wp, bs, blk, draft_temperature, keep_logits=None):
    """blk[0, 1:bs] <- draft tokens.  Greedy (every loop but the policy one at T>0):
    fused lm_head GEMM + argmax.  benchmark_dynamic_schedule.py:342 samples the draft
    with the temperature: then the logits are materialised by the same GEMM and the
    reference's softmax + multinomial is applied to them.  keep_logits (bf16 [16 * tiles, V]): the caller's
    buffer for the materialised logits (the multi-candidate loop builds its candidates from them)."""
    if draft_temperature < 1e-5:
        model.draft_tokens(hid_frag, lm_wp, bs, blk[0], logits=keep_logits)
        return
    V = model.config.vocab_size
    logits = keep_logits
    if logits is None:
        logits = torch.empty(16 * ((bs + 15) // 16), V, dtype=torch.bfloat16, device=model.device)
    model.draft_tokens(hid_frag, lm_wp, bs, blk[0], logits=logits)
    blk[:, 1:bs] = sample(logits[1:bs].unsqueeze(0), draft_temperature)


def _resolve_profile(cycle_trace, ttft, decode_wall):
    """benchmark.py:208-240: turn the recorded event pairs into seconds."""
    torch.cuda.synchronize()
    tot = {"draft": 0.0, "target": 0.0, "cycle": 0.0}
    for row in cycle_trace:
        ev = row.pop("_events")
        for k in tot:
            pair = ev.get(k)
            s = pair[0].elapsed_time(pair[1]) / 1000.0 if pair and pair[0] is not None and pair[1] is not None else 0.0
            row[f"{k}_s"] = float(s)
            tot[k] += s
    den = max(1e-12, tot["draft"] + tot["target"])
    return {"target_prefill_s": float(ttft), "target_decode_s": float(tot["target"]),
            "draft_decode_s": float(tot["draft"]), "cycle_decode_s_sum": float(tot["cycle"]),
            "decode_wall_s": float(decode_wall), "profiled_cycles": int(len(cycle_trace)),
            "draft_share_decode": float(tot["draft"] / den), "target_share_decode": float(tot["target"] / den)}


def dflash_generate(model, target, input_ids: torch.Tensor, mask_token_id: int, max_new_tokens: int,
                    block_size: int, stop_token_ids, temperature: float = 0.0, collect_profile: bool = False,
                    draft_steps: int = 1, draft_token_hook=None) -> SimpleNamespace:
    """benchmark.py:44-55 signature; returns the namespace of :242-251."""
    if getattr(model, "wide_hidden", False):
        return _generate_wide_hidden(model, target, input_ids, mask_token_id, max_new_tokens, block_size, stop_token_ids,
                                     temperature, collect_profile, draft_steps, draft_token_hook)
    r = run_decode(model, target, input_ids, mask_token_id=mask_token_id, max_new_tokens=max_new_tokens,
                   block_size=block_size, stop_token_ids=stop_token_ids, temperature=temperature, clamp_tail=True,
                   draft_steps=draft_steps, collect_profile=collect_profile, draft_token_hook=draft_token_hook)
    return SimpleNamespace(output_ids=r.output_ids, num_input_tokens=r.num_input_tokens,
                           num_output_tokens=r.num_output_tokens, time_to_first_token=r.time_to_first_token,
                           time_per_output_token=r.time_per_output_token, acceptance_lengths=r.acceptance_lengths,
                           cycle_trace=r.cycle_trace, profile_summary=r.profile_summary)


def _generate_wide_hidden(model, target, input_ids, mask_token_id, max_new_tokens, block_size, stop_token_ids, temperature,
                          collect_profile=False, draft_steps=1, draft_token_hook=None) -> SimpleNamespace:
    """hidden_size > 4096: the request runs as a group of ONE through the ragged-batch kernels (their GEMMs cut K over
    workgroups; the single-request kernels keep a whole K = hidden row slice per workgroup).  Same loop semantics
    (benchmark.py:44-251, tail clamp included); no per-cycle profile, one draft step per cycle."""
    from .batch import dflash_generate_batch
    from .target import NativeTarget
    if not isinstance(target, NativeTarget):
        raise NotImplementedError("hidden_size > 4096: wrap the target in dflash_amd.NativeTarget (the draft forward and the "
                                  "verify share the ragged-batch launches)")
    if collect_profile or draft_steps != 1:
        raise NotImplementedError("hidden_size > 4096: no per-cycle profile and one draft step per cycle")
    hook = (lambda r, blk, start, call: draft_token_hook(blk, start, call)) if draft_token_hook else None
    return dflash_generate_batch(model, target, [input_ids], mask_token_id, max_new_tokens, block_size, stop_token_ids,
                                 temperature, draft_token_hook=hook, group_size=1, hook_block_view=True)[0]


def dflash_generate_policy(*, model, target, input_ids: torch.Tensor, mask_token_id: int, max_new_tokens: int,
                           stop_token_ids, temperature: float, fixed_block_size: Optional[int] = None,
                           scheduler=None, draft_token_hook=None) -> SimpleNamespace:
    """benchmark_dynamic_schedule.py:260-272 signature; returns the namespace of :425-434."""
    if fixed_block_size is None and scheduler is None:
        raise ValueError("Either fixed_block_size or scheduler must be provided.")
    max_bs = fixed_block_size if fixed_block_size is not None else max(scheduler.candidates)
    sched = scheduler if fixed_block_size is None else _Fixed(fixed_block_size)
    r = run_decode(model, target, input_ids, mask_token_id=mask_token_id, max_new_tokens=max_new_tokens,
                   block_size=max_bs, stop_token_ids=stop_token_ids, temperature=temperature, clamp_tail=True,
                   scheduler=sched, draft_temperature=temperature, max_block_size=max_bs,
                   draft_token_hook=draft_token_hook)
    if fixed_block_size is not None:
        for row in r.cycle_trace:
            for k in ("tau_hat", "cycle_hat", "score_hat", "current_block_size", "adl_lgen_hat", "adl_lacc_hat",
                      "adl_target_k", "adl_target_bs"):
                row[k] = None
    return SimpleNamespace(output_ids=r.output_ids, num_input_tokens=r.num_input_tokens,
                           num_output_tokens=r.num_output_tokens, time_to_first_token=r.time_to_first_token,
                           time_per_output_token=r.time_per_output_token, acceptance_lengths=r.acceptance_lengths,
                           used_block_sizes=r.used_block_sizes, cycle_trace=r.cycle_trace)


class _Fixed:
    """fixed_block_size path of dflash_generate_policy: a scheduler that never moves."""
    candidates = ()
    tau_hat = cycle_hat = score_hat = {}
    adl_lgen_hat = adl_lacc_hat = None

    def __init__(self, bs):
        self.bs = bs
        self.candidates = (bs,)
        self.current = self.adl_target_k = self.adl_target_bs = bs

    def select(self, cyc):
        return self.bs

    def update(self, **_):
        pass
