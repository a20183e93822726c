"""Continuous batching of utterances into one fixed-shape running batch.

No reference counterpart: the reference's `generate` (modeling_csm.py:591-702) runs one batch to completion and stops
when ALL rows emit an all-zero frame in the same step (:662).  Here a batch of `batch_size` rows keeps replaying the
captured frame-step graph; a row whose utterance has finished (all-zero frame, or its frame budget) is handed to the
next queued utterance through `csm_prefill_slot`: the new context is prefilled right-aligned against the batch's
current length (the layout of a left-padded row, which the reference positions the same way), the other rows never
notice.  Finished rows that find no successor are frozen by the per-row stop (they emit zeros).

A joining context LONGER than the batch's current length is admitted too: the resident rows are first moved up in the
cache by the difference (`csm_shift_context`: keys re-rotated, RoPE is relative, so their continuation changes by fp32
rounding only) and the batch's shared length grows to the new context's; the cache is re-homed into a larger engine
first when it lacks the room.  A row that finds no queued utterance stays idle (frozen by the per-row stop) and is
offered the queue again after every chunk.

Bounds (round 4).  Growth by re-homing is CAPPED: a queued context that would move the resident rows by more than
`max_shift` positions, push the batch's shared length beyond `max_total_len`, or be the `max_shifts_per_batch`-th move of
this batch stays at the head of the queue (FIFO: nothing overtakes it) and opens the NEXT batch once the running one has
drained -- instead of doubling the cache without limit.  (With a bf16 KV cache every move rounds the resident keys to
bf16 once more: `max_shifts_per_batch` bounds that accumulation too; the fp32 default cache re-rotates in fp32.)  Joins are
BUDGETED per chunk (`join_budget_rows` context frames prefilled between two chunks): a burst of joins cannot hold the
resident rows' next chunk back by more than one budget's prefill.  Real time is 12.5 frames/s per stream (80 ms per frame,
/root/reference/ARCHITECTURE.md:44): `latency` records per request the time to its first frame, every inter-chunk gap and
the chunks delivered later than chunk length x 80 ms after their predecessor.
"""
from __future__ import annotations

from collections import deque
from typing import Dict, Optional

import torch


class ContinuousBatcher:
    def __init__(self, model, batch_size: int, temperature: float = 1.0, topk: int = 50, max_new_frames: int = 100,
                 check_every: int = 8, seed: Optional[int] = None, initial_frames: Optional[int] = None, audio_decoder=None,
                 max_shift: int = 1024, max_total_len: int = 16384, max_shifts_per_batch: int = 8, join_budget_rows: int = 4096,
                 frame_seconds: float = 0.08, clock=None, clamp_audio_ids: bool = False):
        if batch_size < 1:
            raise ValueError("batch_size must be positive")
        self.model = model
        self.B = int(batch_size)
        self.temperature, self.topk = float(temperature), int(topk)
        self.default_budget = int(max_new_frames)
        self.check_every = max(1, int(check_every))
        self.seed = seed
        self.initial_frames = initial_frames   # frames of cache room reserved up front (default: every queued budget, <= 4096)
        self._queue = deque()
        self._next_id = 0
        # optional `MimiDecoder`: every chunk of frames is decoded to audio by ONE stream-group call for the whole batch
        # (csm_mimi_streams_*), a row's stream is restarted when a new utterance takes the row over; `self.audio` then holds
        # {request id: waveform [n * samples_per_frame]} (CPU) next to the frames `run()` returns
        self.audio_decoder = audio_decoder
        # a trained model never emits ids >= the codec's codebook size (2048 of the 2051 vocabulary entries); random-weight
        # benchmarks do: clamp_audio_ids=True folds them instead of raising (timing runs only)
        self.clamp_audio_ids = bool(clamp_audio_ids)
        self.audio: Dict[int, torch.Tensor] = {}
        self.joint_joins = True        # several joins of one chunk through one slot prefill (csm_prefill_slots); False: one by one
        self.joined_together = 0       # ... how many utterances joined that way (statistics)
        self.joined_mid_batch = 0      # utterances that took over a row of a running batch (statistics)
        self.shifted_for_long_context = 0   # joins whose context was longer than the running batch (resident rows moved up)
        self.max_shift, self.max_total_len, self.max_shifts_per_batch = int(max_shift), int(max_total_len), int(max_shifts_per_batch)
        self.join_budget_rows = int(join_budget_rows)
        self.deferred_to_next_batch = 0     # long contexts that were refused as a join and opened the next batch instead
        self.joins_deferred_by_budget = 0   # joins that waited one more chunk because the chunk's prefill budget was spent
        self.skip_ahead = 4                 # requests that may overtake a head deferred to the next batch (in total, per deferred head)
        self.overtakes = 0
        self._overtaken: Dict[int, int] = {}
        self._refused = None                # why the last _join refused the head of the queue: None | "growth" | "budget"
        self.frame_seconds = float(frame_seconds)
        import time as _time
        self._clock = clock or _time.perf_counter
        # per request: submit time, time to first frame, gaps between consecutive chunk deliveries, late chunks (gap > frames x 80 ms)
        self.latency: Dict[int, dict] = {}
        self.errors: Dict[int, str] = {}       # requests that failed on their own (the batch they ran in did not)

    def submit(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, max_new_frames: Optional[int] = None) -> int:
        """input_ids / attention_mask `[T, 33]` (or `[1, T, 33]`) of ONE utterance; returns its request id."""
        if input_ids.dim() == 3:
            input_ids, attention_mask = input_ids[0], attention_mask[0]
        if input_ids.dim() != 2 or input_ids.shape != attention_mask.shape:
            raise ValueError("one utterance: input_ids and attention_mask of shape [T, C+1]")
        if input_ids.shape[0] + self.check_every + 1 > self.max_total_len:
            # the cap on a batch's total length holds for the batch an utterance OPENS as well as for joins
            raise ValueError(f"context of {input_ids.shape[0]} frames cannot be served within max_total_len = {self.max_total_len}")
        rid = self._next_id
        self._next_id += 1
        self._queue.append((rid, input_ids.cpu(), attention_mask.cpu(), int(max_new_frames or self.default_budget)))
        self.latency[rid] = {"t_submit": self._clock(), "t_first_frame": None, "ttff_s": None, "t_last": None, "chunk_gaps_s": [],
                             "late_chunks": 0, "chunks": 0}
        return rid

    def _delivered(self, rid: int, frames: int, now: float):
        """`frames` (> 0) frames of request `rid` reached the caller at `now`"""
        L = self.latency[rid]
        if L["t_first_frame"] is None:
            L["t_first_frame"] = now
            L["ttff_s"] = now - L["t_submit"]
        else:
            gap = now - L["t_last"]
            L["chunk_gaps_s"].append(gap)
            if gap > frames * self.frame_seconds:
                L["late_chunks"] += 1
        L["t_last"] = now
        L["chunks"] += 1

    def latency_summary(self) -> dict:
        """time to first frame and inter-chunk gaps over all finished requests (seconds): p50 / p99 / max, late chunks"""
        import statistics
        tt = sorted(v["ttff_s"] for v in self.latency.values() if v["ttff_s"] is not None)
        gaps = sorted(g for v in self.latency.values() for g in v["chunk_gaps_s"])

        def q(xs, p):
            return xs[min(len(xs) - 1, int(p * len(xs)))] if xs else None
        return {"requests": len(tt), "ttff_s": {"p50": q(tt, 0.5), "p99": q(tt, 0.99), "max": tt[-1] if tt else None,
                                                "mean": statistics.fmean(tt) if tt else None},
                "inter_chunk_gap_s": {"p50": q(gaps, 0.5), "p99": q(gaps, 0.99), "max": gaps[-1] if gaps else None},
                "chunk_deadline_s": self.check_every * self.frame_seconds,
                "late_chunks": sum(v["late_chunks"] for v in self.latency.values()), "chunks": sum(v["chunks"] for v in self.latency.values())}

    # ------------------------------------------------------------------------------------------------------------
    def run(self) -> Dict[int, torch.Tensor]:
        """Generates every queued utterance; returns {request id: LongTensor [n, 32]} (frames before the all-zero
        frame, at most the request's budget) on the CPU."""
        results: Dict[int, torch.Tensor] = {}
        while self._queue:
            self._run_batch(results)
        return results

    def _run_batch(self, results):
        m, B = self.model, self.B
        C = m.config.audio_num_codebooks
        first = [self._queue.popleft() for _ in range(min(B, len(self._queue)))]
        for item in first:
            self._overtaken.pop(item[0], None)      # admitted (it opens this batch): its overtake count is history
        T0 = max(r[1].shape[0] for r in first)
        ids = torch.zeros(B, T0, C + 1, dtype=torch.long)
        mask = torch.zeros(B, T0, C + 1, dtype=first[0][2].dtype)
        rows = []                                   # per row: None (idle) or [rid, budget, list of frames]
        for b in range(B):
            rid, ri, rm, budget = first[b] if b < len(first) else first[0]
            T = ri.shape[0]
            ids[b, T0 - T:], mask[b, T0 - T:] = ri, rm                 # left padding
            rows.append([rid, budget, []] if b < len(first) else None)
        k = self.check_every
        # room for every queued budget up front where that is cheap; beyond it the engine is re-homed on the fly
        horizon = min(4096, sum(r[3] for r in first) + sum(r[3] for r in self._queue) + 2 * k)
        if self.initial_frames is not None:
            horizon = int(self.initial_frames)
        eng = m._ensure_engine(B, T0 + max(8 * k, horizon) + 1, max(4 * k, 32), B * T0)
        eng.reset()
        m._epoch += 1
        m._frame_pending = False
        eng.set_kv_start(m._kv_starts(mask, B, T0))
        eng.prefill(ids, mask, want_outputs=False)
        s = eng.sampling(temperature=self.temperature, topk=self.topk, seed=m._next_seed() if self.seed is None else int(self.seed),
                         row_offset=m.row_offset, per_row_stop=True)
        dec = self.audio_decoder
        if dec is not None:
            if dec.max_frames < B:
                raise ValueError(f"audio_decoder.max_frames ({dec.max_frames}) must be at least the batch size ({B})")
            dec.streams_open(B)
            spf = dec.cfg.samples_per_frame
            waves = [[] for _ in range(B)]
        self._shifts_this_batch = 0
        while any(r is not None for r in rows):
            if eng.frames + k > eng.max_frames:
                eng.rewind_frames()                 # every frame so far has been read out
            if eng.length + k + 1 > eng.max_len:
                eng = m._ensure_engine(B, eng.length + k + 1, max(4 * k, 32), 1, cont=True)   # re-homed, never restarted
            f0 = eng.frames
            eng.generate(s, k, m.use_graph)
            toks_dev = eng.read_frames(f0, k)
            toks = toks_dev.cpu()
            # pass 1 (host, before anything reaches the codec): how many of the chunk's frames every live row takes, and whether the
            # delivered ids fit the codec's codebook.  A request whose ids do not fails alone -- its tokens are returned,
            # its audio is not -- and the check comes BEFORE the decode: nothing out of range is clamped into the stateful codec on behalf
            # of a request that is about to be failed
            took_of, done_of = {}, {}
            for b, r in enumerate(rows):
                if r is None:
                    continue
                took, done = 0, False
                for i in range(k):
                    if bool((toks[b, i] == 0).all()) or len(r[2]) + took >= r[1]:
                        done = True
                        break
                    took += 1
                if dec is not None and took and not self.clamp_audio_ids and int(toks[b, :took].max()) >= dec.cfg.codebook_size:
                    self.errors[r[0]] = (f"generated token id {int(toks[b, :took].max())} is outside the codec's codebook "
                                         f"({dec.cfg.codebook_size}); no audio for this request")
                    done = True
                    waves[b] = []
                took_of[b], done_of[b] = took, done
            wav = None
            if dec is not None:
                # the chunk's frames of EVERY row through the codec, max_frames // B frames per stream-group call (frames of idle / finished /
                # failed rows are decoded too and dropped: their streams are reset when the row is taken over; the clamp only touches those)
                live = [b for b, r in enumerate(rows) if r is not None and r[0] not in self.errors]
                codes = toks_dev.clamp(max=dec.cfg.codebook_size - 1).permute(0, 2, 1).contiguous()      # [B, 32, k]
                step = max(1, dec.max_frames // B)
                wav_dev = torch.cat([dec.streams_decode(codes[:, :, a:a + step]) for a in range(0, k, step)], dim=-1)   # [B, 1, k * spf]
                # only the live rows' samples cross to the host (one copy), indexed back by row below
                wav_row = {b: i for i, b in enumerate(live)}
                wav = wav_dev[torch.tensor(live, device=wav_dev.device)].cpu() if live else None
            now = self._clock()
            for b, r in enumerate(rows):
                if r is None:
                    continue
                took, done = took_of[b], done_of[b]
                for i in range(took):
                    r[2].append(toks[b, i])
                if wav is not None and took and r[0] not in self.errors:
                    waves[b].append(wav[wav_row[b], 0, :took * spf])
                if took:
                    self._delivered(r[0], took, now)
                if done or len(r[2]) >= r[1]:
                    results[r[0]] = torch.stack(r[2]) if r[2] else torch.zeros(0, C, dtype=torch.long)
                    if dec is not None:
                        self.audio[r[0]] = torch.cat(waves[b]) if waves[b] else torch.zeros(0)
                        waves[b] = []
                    rows[b] = None
            # every idle row (just finished, or idle since an earlier chunk) is offered the queue.  Several joins in one chunk
            # go through ONE slot prefill where that fits (contexts no longer than the batch, max_prefill_rows in total): the
            # running batch waits for one short prefill instead of one per join
            idle = [b for b in range(B) if rows[b] is None]
            self._budget_left = self.join_budget_rows        # context frames that may be prefilled before the next chunk
            if self.joint_joins and len(idle) >= 2 and len(self._queue) >= 2:
                eng = self._join_many(eng, rows, idle, k, dec)
                idle = [b for b in range(B) if rows[b] is None]
            for b in idle:
                if not self._queue:
                    break
                others_live = any(r is not None for r in rows)
                st, eng = self._join(eng, b, k, others_live)
                if st is None and self._refused == "growth" and others_live:
                    # the head was refused IN THIS CALL for the growth / shift cap and waits for the NEXT batch: up to `skip_ahead` shorter
                    # requests behind it may take idle rows meanwhile -- a bounded overtake, so the deferred head cannot starve.  Not when it
                    # was refused for the chunk's prefill budget (it joins after the next chunk), and not when no row is live: then the
                    # batch ends and the head opens the next one -- an overtaker would keep this batch alive at occupancy 1
                    head = self._queue[0][0]
                    if self._overtaken.get(head, 0) >= self.skip_ahead:
                        break
                    for pos in range(1, min(len(self._queue), 1 + self.skip_ahead)):
                        if self._queue[pos][1].shape[0] <= eng.length and self._queue[pos][1].shape[0] <= self._budget_left:
                            self._overtaken[head] = self._overtaken.get(head, 0) + 1
                            self._queue.rotate(-pos)
                            item = self._queue.popleft()
                            self._queue.rotate(pos)
                            self._queue.appendleft(item)
                            self.overtakes += 1
                            st, eng = self._join(eng, b, k, True)
                            break
                if st is None:                      # the head of the queue waits (budget spent / growth cap)
                    break
                rows[b] = st
                if dec is not None:
                    dec.streams_reset(b)        # the row's audio stream starts from silence with the new utterance
        m._epoch += 1

    def _join_many(self, eng, rows, idle, k, dec):
        """As many of the queue's first utterances as there are idle rows join through ONE slot prefill (Engine.prefill_slots),
        FIFO order kept: the leading run of queued contexts that are no longer than the batch's current length and fit the
        prefill scratch together.  Whatever is left (a longer context at the head of the queue, a single join) goes through
        `_join`."""
        m = self.model
        if eng.length + k + 1 > eng.max_len:
            eng = m._ensure_engine(self.B, eng.length + k + 1, max(4 * k, 32), 1, cont=True)
        take = []
        smax = 0
        for item in list(self._queue)[:len(idle)]:
            S = item[1].shape[0]
            if S > eng.length or (len(take) + 1) * max(smax, S) > eng.max_prefill_rows:
                break
            if take and (len(take) + 1) * max(smax, S) > self._budget_left:      # (the first join of a chunk is always admitted)
                self.joins_deferred_by_budget += 1
                break
            take.append(item)
            smax = max(smax, S)
        if len(take) < 2:
            return eng
        use = idle[:len(take)]
        if not eng.prefill_slots(use, [t[1] for t in take], [t[2] for t in take]):
            return eng
        self._budget_left -= len(take) * smax
        for b, (rid, _, _, budget) in zip(use, take):
            self._queue.popleft()
            rows[b] = [rid, budget, []]
            self.joined_mid_batch += 1
            self.joined_together += 1
            if dec is not None:
                dec.streams_reset(b)
        return eng

    def _join(self, eng, row, k, others_live=True):
        """The first queued utterance takes over `row`.  Its context is placed right-aligned against the batch's current
        length; a longer one first moves the resident rows up (Engine.prefill_slot -> shift_context), after the engine has
        been re-homed into a larger one if the cache lacks the room.  Returns (row state, engine) -- or (None, engine) when
        the utterance has to WAIT: the chunk's prefill budget is spent (it joins after the next chunk), or admitting it would
        move the resident rows too far / too often or grow the batch beyond `max_total_len` (it stays at the head of the
        queue and opens the next batch; with no live row left the batch simply ends)."""
        m = self.model
        rid, ri, rm, budget = self._queue[0]
        S = ri.shape[0]
        if S > eng.length and (S - eng.length > self.max_shift or S + k + 1 > self.max_total_len or
                               self._shifts_this_batch >= self.max_shifts_per_batch):
            if getattr(self, "_deferred_rid", None) != rid:
                self.deferred_to_next_batch += 1
                self._deferred_rid = rid
            self._refused = "growth"
            return None, eng
        if S > self._budget_left and self._budget_left < self.join_budget_rows and others_live:
            self.joins_deferred_by_budget += 1      # not the first join of this chunk and it does not fit what is left
            self._refused = "budget"
            return None, eng
        self._refused = None
        self._overtaken.pop(rid, None)              # admitted: its overtake count is history
        self._queue.popleft()
        self._budget_left -= S
        need = max(eng.length, S) + k + 1
        if need > eng.max_len:
            eng = m._ensure_engine(self.B, need, max(4 * k, 32), 1, cont=True)
        if S > eng.length:
            self.shifted_for_long_context += 1
            self._shifts_this_batch += 1
        eng.prefill_slot(row, ri, rm)
        self.joined_mid_batch += 1
        return [rid, budget, []], eng
