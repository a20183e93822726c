"""CPU suite: host logic of the continuous batcher (csm_hf_amd/serving.py) on a stub engine -- who joins which row
when, budgets, the all-zero end-of-utterance frame, contexts longer than the running batch (the resident rows are moved
up), idle rows offered the queue again after every chunk, the frame ring wrapping."""
import types

import pytest

import torch

from csm_hf_amd.serving import ContinuousBatcher


class StubEngine:
    """Row r of the running batch emits frame t of its CURRENT utterance as key*100 + t + 1 in every codebook; an
    utterance with key % 5 == 0 ends with an all-zero frame after 3 frames (and keeps emitting zeros, like a frozen row)."""

    def __init__(self, B, max_len, max_frames):
        self.max_batch, self.max_len, self.max_frames = B, max_len, max_frames
        self.max_prefill_rows = 64
        self.length = self.frames = 0
        self.log = []

    def reset(self):
        self.length = self.frames = 0

    def set_kv_start(self, starts):
        self.starts = list(starts)

    def prefill(self, ids, mask, want_outputs=True):
        B, T = ids.shape[:2]
        self.length = T
        self.ring = torch.zeros(B, self.max_frames, 32, dtype=torch.long)
        self.key = [int(ids[b, -1, 0]) for b in range(B)]
        self.t = [0] * B
        self.log.append(("prefill", B, T))

    def sampling(self, **kw):
        return types.SimpleNamespace(**kw)

    def rewind_frames(self):
        self.frames = 0
        self.log.append(("rewind",))

    def generate(self, s, n, use_graph=True):
        assert self.frames + n <= self.max_frames and self.length + n + 1 <= self.max_len
        for i in range(n):
            for b, k in enumerate(self.key):
                v = 0 if (k % 5 == 0 and self.t[b] >= 3) else k * 100 + self.t[b] + 1
                self.ring[b, self.frames] = v
                self.t[b] += 1
            self.frames += 1
            self.length += 1

    def read_frames(self, first, n):
        return self.ring[:, first:first + n].clone()

    def prefill_slot(self, row, ids, mask):
        if ids.shape[0] > self.length:            # Engine.prefill_slot: resident rows move up, the shared length grows
            assert ids.shape[0] + 1 <= self.max_len, "the batcher must re-home the cache before a long context joins"
            self.log.append(("shift", ids.shape[0] - self.length))
            self.length = ids.shape[0]
        self.key[row] = int(ids[-1, 0])
        self.t[row] = 0
        self.log.append(("join", row, ids.shape[0], self.length))


    def prefill_slots(self, rows, ids_list, mask_list):
        # Engine.prefill_slots: several joins through one prefill; refuses (False) what does not fit
        S = max(t.shape[0] for t in ids_list)
        if len(rows) < 2 or S > self.length or len(rows) * S > self.max_prefill_rows:
            return False
        for row, ids in zip(rows, ids_list):
            self.key[row] = int(ids[-1, 0])
            self.t[row] = 0
            self.log.append(("join", row, ids.shape[0], self.length))
        self.log.append(("joint", tuple(rows)))
        return True


class StubModel:
    def __init__(self, max_len=4096, max_frames=64):
        self.config = types.SimpleNamespace(audio_num_codebooks=32)
        self._epoch, self._frame_pending, self.row_offset, self.use_graph = 0, False, 0, True
        self.max_len, self.max_frames = max_len, max_frames
        self.engines = []

    def _ensure_engine(self, B, need_len, frames, rows, cont=False, must_prefill_rows=0):
        if cont:
            e = self.engines[-1]
            e.max_len = max(need_len, 2 * e.max_len)
            e.log.append(("grow", e.max_len))
            return e
        e = StubEngine(B, max(need_len, self.max_len), self.max_frames)
        self.engines.append(e)
        return e

    def _kv_starts(self, mask, B, T):
        return [int((mask[b].sum(-1) == 0).sum()) for b in range(B)]

    def _next_seed(self):
        return 1


def utterance(key, T):
    ids = torch.zeros(T, 33, dtype=torch.long)
    ids[:, 0] = key
    mask = torch.ones(T, 33, dtype=torch.int32)
    return ids, mask


def test_rows_are_handed_over_and_results_are_per_utterance():
    m = StubModel()
    cb = ContinuousBatcher(m, batch_size=2, topk=1, check_every=4)
    specs = [(1, 6, 10), (2, 4, 3), (3, 5, 6), (10, 3, 9), (4, 7, 2)]          # (key, context frames, budget); key 10 ends early
    rid = [cb.submit(*utterance(k, T), max_new_frames=b) for k, T, b in specs]
    out = cb.run()
    assert sorted(out) == rid
    for r, (k, T, b) in zip(rid, specs):
        n = min(b, 3) if k % 5 == 0 else b                                      # the all-zero frame is not returned
        assert out[r].shape == (n, 32)
        assert torch.equal(out[r][:, 0], torch.arange(n) + k * 100 + 1)
    e = m.engines[0]
    assert e.log[0] == ("prefill", 2, 6) and e.starts == [0, 2]                 # shorter context left-padded
    joins = [x for x in e.log if x[0] == "join"]
    assert len(joins) == 3 == cb.joined_mid_batch and len(m.engines) == 1       # one batch served all five
    assert all(j[2] <= j[3] for j in joins)


def test_context_longer_than_the_batch_joins_by_moving_the_resident_rows_up_and_ring_wraps():
    m = StubModel(max_frames=8)
    cb = ContinuousBatcher(m, batch_size=2, topk=1, check_every=4)
    a = cb.submit(*utterance(1, 3), max_new_frames=2)
    b = cb.submit(*utterance(2, 3), max_new_frames=30)                          # 30 frames through an 8-frame ring
    c = cb.submit(*utterance(3, 500), max_new_frames=2)                         # longer than the running batch (length 7)
    d = cb.submit(*utterance(4, 3), max_new_frames=2)
    out = cb.run()
    assert [out[x].shape[0] for x in (a, b, c, d)] == [2, 30, 2, 2]
    assert torch.equal(out[b][:, 5], torch.arange(30) + 201)                    # the resident row never notices
    e = m.engines[0]
    assert len(m.engines) == 1 and ("rewind",) in e.log
    joins = [x for x in e.log if x[0] == "join"]
    assert joins[0][1:3] == (0, 500) and joins[1][1:3] == (0, 3)                # utterance 3, then 4, took over row 0
    assert ("shift", 493) in e.log and cb.shifted_for_long_context == 1         # 7 cached positions -> 500


def test_long_context_re_homes_the_cache_first_and_idle_rows_are_offered_the_queue_again():
    m = StubModel(max_len=0)
    cb = ContinuousBatcher(m, batch_size=2, topk=1, check_every=4, initial_frames=8)
    a = cb.submit(*utterance(1, 3), max_new_frames=2)
    b = cb.submit(*utterance(2, 3), max_new_frames=12)
    out_first = None
    c = cb.submit(*utterance(3, 300), max_new_frames=2)                         # needs a larger cache than the batch has
    out = cb.run()
    assert [out[x].shape[0] for x in (a, b, c)] == [2, 12, 2]
    e = m.engines[0]
    grow = [i for i, x in enumerate(e.log) if x[0] == "grow"]
    shift = [i for i, x in enumerate(e.log) if x[0] == "shift"]
    assert grow and shift and grow[0] < shift[0]                                # re-homed BEFORE the rows were moved up
    # an utterance submitted while a row sits idle is picked up at the next chunk boundary
    m2 = StubModel()
    cb2 = ContinuousBatcher(m2, batch_size=2, topk=1, check_every=2)
    x = cb2.submit(*utterance(1, 3), max_new_frames=2)
    y = cb2.submit(*utterance(2, 3), max_new_frames=20)
    orig = m2._ensure_engine

    def late_submit(*a_, **k_):
        eng = orig(*a_, **k_)
        gen = eng.generate

        def generate(s, n, use_graph=True):
            gen(s, n, use_graph)
            if eng.frames == 6 and not getattr(eng, "_late", False):            # row 0 has been idle for two chunks
                eng._late = True
                cb2.late = cb2.submit(*utterance(3, 3), max_new_frames=2)
        eng.generate = generate
        return eng
    m2._ensure_engine = late_submit
    out2 = cb2.run()
    assert out2[cb2.late].shape[0] == 2 and len(m2.engines) == 1                # joined the running batch, no new one
    assert [z for z in m2.engines[0].log if z[0] == "join"][0][1] == 0


def test_cache_growth_is_requested_as_a_continuation():
    m = StubModel(max_len=0)
    cb = ContinuousBatcher(m, batch_size=1, topk=1, check_every=4, initial_frames=4)
    r = cb.submit(*utterance(7, 5), max_new_frames=80)
    out = cb.run()
    assert out[r].shape[0] == 80 and any(x[0] == "grow" for x in m.engines[0].log)


def test_joins_of_one_chunk_go_through_one_slot_prefill():
    """several rows finishing in the same chunk: the queue's leading utterances join through ONE Engine.prefill_slots call (FIFO
    kept); a context longer than the running batch at the head of the queue falls back to the one-by-one path; switched off
    (`joint_joins = False`) nothing is joined together.  Results are per utterance either way."""
    specs = [(1, 6, 3), (2, 6, 3), (3, 6, 3), (4, 5, 4), (6, 4, 4), (7, 6, 2), (8, 40, 3), (9, 3, 2)]
    for joint in (True, False):
        m = StubModel()
        cb = ContinuousBatcher(m, batch_size=3, topk=1, check_every=4)
        cb.joint_joins = joint
        rid = [cb.submit(*utterance(k, T), max_new_frames=b) for k, T, b in specs]
        out = cb.run()
        assert sorted(out) == rid
        for r, (k, T, b) in zip(rid, specs):
            assert torch.equal(out[r][:, 0], torch.arange(b) + k * 100 + 1)
        e = m.engines[0]
        joint_calls = [x for x in e.log if x[0] == "joint"]
        if joint:
            assert joint_calls and joint_calls[0][1] == (0, 1, 2) and cb.joined_together >= 3      # keys 4, 6, 7 took rows 0-2 together
            assert any(x[0] == "shift" for x in e.log)                                             # the 40-frame context joined alone
        else:
            assert not joint_calls and cb.joined_together == 0
        assert cb.joined_mid_batch == 5


def test_growth_is_capped_a_too_long_context_opens_the_next_batch():
    """a join that would move the resident rows further than `max_shift` (or beyond `max_total_len`, or
    once too often) no longer re-homes the cache without bound: it stays at the head of the queue -- nothing overtakes it --
    and opens the NEXT batch when the running one has drained."""
    m = StubModel(max_frames=64)
    cb = ContinuousBatcher(m, batch_size=2, topk=1, check_every=4, max_shift=100)
    cb.skip_ahead = 0                                            # strict FIFO (round 5 default: a bounded overtake, tested below)
    a = cb.submit(*utterance(1, 3), max_new_frames=2)
    b = cb.submit(*utterance(2, 3), max_new_frames=14)
    c = cb.submit(*utterance(3, 900), max_new_frames=2)          # 900 - 7 > max_shift: may not join the running batch
    d = cb.submit(*utterance(4, 3), max_new_frames=2)            # queued behind it: waits too (FIFO)
    out = cb.run()
    assert [out[x].shape[0] for x in (a, b, c, d)] == [2, 14, 2, 2]
    assert len(m.engines) == 2 and cb.deferred_to_next_batch == 1 and cb.shifted_for_long_context == 0
    assert not any(x[0] == "shift" for x in m.engines[0].log) and not any(x[0] == "join" for x in m.engines[0].log)
    assert m.engines[1].log[0] == ("prefill", 2, 900)            # c and d opened the next batch together
    # round 5: with the default bounded overtake the short request behind the deferred head takes the idle row of the
    # RUNNING batch instead of waiting for it to drain; the head still opens the next batch
    m3 = StubModel(max_frames=64)
    cb3 = ContinuousBatcher(m3, batch_size=2, topk=1, check_every=4, max_shift=100)
    a3 = cb3.submit(*utterance(1, 3), max_new_frames=2)
    b3 = cb3.submit(*utterance(2, 3), max_new_frames=14)
    c3 = cb3.submit(*utterance(3, 900), max_new_frames=2)
    d3 = cb3.submit(*utterance(4, 3), max_new_frames=2)
    out3 = cb3.run()
    assert [out3[x].shape[0] for x in (a3, b3, c3, d3)] == [2, 14, 2, 2]
    assert cb3.overtakes == 1 and cb3.deferred_to_next_batch == 1 and len(m3.engines) == 2
    assert any(x[0] == "join" for x in m3.engines[0].log) and m3.engines[1].log[0][0] == "prefill" and m3.engines[1].log[0][2] == 900
    # within the cap the join happens as before; the number of moves per batch is bounded too
    m2 = StubModel(max_frames=64)
    cb2 = ContinuousBatcher(m2, batch_size=2, topk=1, check_every=2, max_shift=100, max_shifts_per_batch=1)
    cb2.submit(*utterance(1, 3), max_new_frames=2)
    cb2.submit(*utterance(2, 3), max_new_frames=30)
    cb2.submit(*utterance(3, 50), max_new_frames=2)              # first move: allowed
    cb2.submit(*utterance(4, 120), max_new_frames=2)             # second move of this batch: deferred
    out2 = cb2.run()
    assert len(out2) == 4 and cb2.shifted_for_long_context == 1 and cb2.deferred_to_next_batch == 1 and len(m2.engines) == 2


def test_join_budget_per_chunk_and_latency_record():
    """at most `join_budget_rows` context frames are prefilled between two chunks (the first join of a chunk
    is always admitted), and every request gets a time-to-first-frame, its inter-chunk gaps and its late chunks (a chunk of k
    frames is late when it follows its predecessor by more than k x 80 ms) on the batcher's clock."""
    t = [0.0]

    def clock():
        return t[0]
    m = StubModel(max_frames=64)
    cb = ContinuousBatcher(m, batch_size=3, topk=1, check_every=2, join_budget_rows=10, clock=clock)
    cb.joint_joins = False
    specs = [(1, 6, 2), (2, 6, 2), (3, 6, 12), (4, 6, 2), (6, 6, 2), (7, 6, 2)]
    rid = [cb.submit(*utterance(k, T), max_new_frames=bud) for k, T, bud in specs]
    orig = m._ensure_engine

    def timed(*a_, **k_):
        eng = orig(*a_, **k_)
        gen, pf = eng.generate, eng.prefill_slot

        def generate(s, n, use_graph=True):
            gen(s, n, use_graph)
            t[0] += 0.05 * n                      # 50 ms per frame: inside the 80 ms deadline

        def prefill_slot(row, ids, mask):
            pf(row, ids, mask)
            t[0] += 0.30                          # a slow join: the NEXT chunk of the resident rows arrives late
        eng.generate, eng.prefill_slot = generate, prefill_slot
        return eng
    m._ensure_engine = timed
    out = cb.run()
    assert sorted(out) == rid
    # rows 0 and 1 finish in the first chunk; two 6-frame contexts are queued but only 10 rows may be prefilled per chunk
    assert cb.joins_deferred_by_budget >= 1
    joins = [x for x in m.engines[0].log if x[0] == "join"]
    assert len(joins) == 3
    L = cb.latency
    assert abs(L[rid[0]]["ttff_s"] - 0.10) < 1e-9 and L[rid[0]]["late_chunks"] == 0
    long_one = L[rid[2]]                           # the 12-frame utterance lives through the joins
    assert long_one["chunks"] == 6 and len(long_one["chunk_gaps_s"]) == 5
    assert long_one["late_chunks"] >= 1 and max(long_one["chunk_gaps_s"]) >= 0.40 - 1e-9     # 0.10 generate + 0.30 join > 2 x 80 ms
    s = cb.latency_summary()
    assert s["requests"] == 6 and s["late_chunks"] >= 1 and s["chunk_deadline_s"] == 0.16 and s["ttff_s"]["max"] >= s["ttff_s"]["p50"]


def test_round5_serving_limits_and_param_signature():
    """a context that cannot fit `max_total_len` is refused at submit (the cap also holds for the batch an utterance
    opens); `errors` exists for requests that fail on their own; the parameter signature notices a REPLACED parameter of unchanged
    count."""
    import torch
    from csm_hf_amd import CSMConfig, CSMModel
    from csm_hf_amd.serving import ContinuousBatcher as BatchServer
    cfg = CSMConfig.tiny()
    m = CSMModel(cfg)
    srv = BatchServer(m, 2, max_total_len=64, check_every=4)
    assert srv.errors == {} and srv.skip_ahead >= 1
    ids = torch.zeros(80, cfg.audio_num_codebooks + 1, dtype=torch.long)
    with pytest.raises(ValueError, match="max_total_len"):
        srv.submit(ids, torch.ones_like(ids))
    srv.submit(ids[:20], torch.ones_like(ids[:20]))
    m2 = CSMModel(cfg)
    m2.load_state_dict({k: torch.zeros(v.shape) for k, v in m2.state_dict().items()})
    sig0 = m2._param_signature()
    m2.projection.weight = torch.nn.Parameter(torch.zeros_like(m2.projection.weight), requires_grad=False)
    assert m2._param_signature() != sig0


def test_overtake_only_for_a_growth_refusal_with_a_live_row_and_bookkeeping_is_pruned():
    """ADVICE r5 (medium).  (i) When the LAST live row of a batch finishes while the head of the queue is deferred for the growth cap, nothing
    overtakes: the batch ends and the head opens the next one (an overtaker used to keep the batch alive at occupancy 1 and make the head
    wait for whole generations).  (ii) A head refused only for the chunk's prefill BUDGET is not overtaken (it joins after the next
    chunk).  (iii) The overtake count of an admitted head is dropped.  (iv) A request whose delivered ids do not fit the codec fails
    alone, before anything of it reaches the codec."""
    # (i) both rows end in the same chunk; c (too long for this batch) is deferred; d is short and could have overtaken
    m = StubModel(max_frames=64)
    cb = ContinuousBatcher(m, batch_size=2, topk=1, check_every=4, max_shift=100)
    a = cb.submit(*utterance(1, 3), max_new_frames=4)
    b = cb.submit(*utterance(2, 3), max_new_frames=4)
    c = cb.submit(*utterance(3, 900), max_new_frames=2)
    d = cb.submit(*utterance(4, 3), max_new_frames=2)
    out = cb.run()
    assert [out[x].shape[0] for x in (a, b, c, d)] == [4, 4, 2, 2]
    assert cb.overtakes == 0 and len(m.engines) == 2
    assert not any(x[0] == "join" for x in m.engines[0].log), m.engines[0].log     # nobody joined the dying batch
    assert m.engines[1].log[0] == ("prefill", 2, 900)                              # c and d opened the next batch together
    assert cb._overtaken == {}
    # (ii) budget refusal: the head (60 frames) does not fit what is left of the chunk's budget after the first join; the short request
    # behind it must NOT slip past it
    m2 = StubModel(max_frames=64)
    cb2 = ContinuousBatcher(m2, batch_size=3, topk=1, check_every=2, join_budget_rows=64)
    r0 = cb2.submit(*utterance(1, 80), max_new_frames=2)
    r1 = cb2.submit(*utterance(2, 80), max_new_frames=2)
    r2 = cb2.submit(*utterance(6, 80), max_new_frames=30)
    r3 = cb2.submit(*utterance(3, 40), max_new_frames=2)        # first join of the chunk: admitted (40 of 64)
    r4 = cb2.submit(*utterance(4, 60), max_new_frames=2)        # 60 > 24 left: waits for the next chunk
    r5 = cb2.submit(*utterance(7, 5), max_new_frames=2)         # would fit the 24 left -- but FIFO holds for a budget refusal
    out2 = cb2.run()
    assert len(out2) == 6 and cb2.overtakes == 0 and cb2.joins_deferred_by_budget >= 1
    joins = [x for x in m2.engines[0].log if x[0] == "join"]
    assert [j[2] for j in joins][:3] == [40, 60, 5], joins      # context lengths in FIFO order
    # (iii) a deferred head that was overtaken and then admitted leaves no entry behind
    m3 = StubModel(max_frames=64)
    cb3 = ContinuousBatcher(m3, batch_size=2, topk=1, check_every=4, max_shift=100)
    cb3.submit(*utterance(1, 3), max_new_frames=2)
    cb3.submit(*utterance(2, 3), max_new_frames=14)
    cb3.submit(*utterance(3, 900), max_new_frames=2)
    cb3.submit(*utterance(4, 3), max_new_frames=2)
    cb3.run()
    assert cb3.overtakes == 1 and cb3._overtaken == {}


def test_out_of_range_ids_fail_the_request_before_the_codec_sees_them():
    class StubCodec:
        def __init__(self):
            self.cfg = types.SimpleNamespace(codebook_size=250, samples_per_frame=4)
            self.max_frames = 8
            self.seen = []

        def streams_open(self, n):
            self.n = n

        def streams_reset(self, b):
            pass

        def streams_decode(self, codes):
            self.seen.append(codes.clone())
            return torch.zeros(codes.shape[0], 1, codes.shape[2] * 4)

    m = StubModel(max_frames=64)
    dec = StubCodec()
    cb = ContinuousBatcher(m, batch_size=2, topk=1, check_every=2, audio_decoder=dec)
    ok = cb.submit(*utterance(1, 3), max_new_frames=4)          # ids 101.. : inside the codebook
    bad = cb.submit(*utterance(3, 3), max_new_frames=4)         # ids 301.. : outside
    out = cb.run()
    assert out[ok].shape[0] == 4 and bad in cb.errors and ok not in cb.errors
    assert cb.audio[ok].numel() == 4 * 4 and cb.audio[bad].numel() == 0
    assert out[bad].shape[0] == 2                               # the tokens of the chunk that failed are still returned
