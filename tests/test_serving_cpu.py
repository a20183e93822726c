"""CPU suite: host logic of the continuous batcher (csm_hf_amd/serving.py) on a stub engine -- who joins which row
when, budgets, the all-zero end-of-utterance frame, contexts that must wait, the frame ring wrapping."""
import types

import torch

from csm_hf_amd.serving import ContinuousBatcher


class StubEngine:
    """Row r of the running batch emits frame t of its CURRENT utterance as key*100 + t + 1 in every codebook; an
    utterance with key % 5 == 0 ends with an all-zero frame after 3 frames (and keeps emitting zeros, like a frozen row)."""

    def __init__(self, B, max_len, max_frames):
        self.max_batch, self.max_len, self.max_frames = B, max_len, max_frames
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
        assert ids.shape[0] <= self.length, "joining context longer than the batch"
        self.key[row] = int(ids[-1, 0])
        self.t[row] = 0
        self.log.append(("join", row, ids.shape[0], self.length))


class StubModel:
    def __init__(self, max_len=4096, max_frames=64):
        self.config = types.SimpleNamespace(audio_num_codebooks=32)
        self._epoch, self._frame_pending, self.row_offset, self.use_graph = 0, False, 0, True
        self.max_len, self.max_frames = max_len, max_frames
        self.engines = []

    def _ensure_engine(self, B, need_len, frames, rows, cont=False):
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


def test_context_longer_than_the_batch_waits_for_the_next_batch_and_ring_wraps():
    m = StubModel(max_frames=8)
    cb = ContinuousBatcher(m, batch_size=2, topk=1, check_every=4)
    a = cb.submit(*utterance(1, 3), max_new_frames=2)
    b = cb.submit(*utterance(2, 3), max_new_frames=30)                          # 30 frames through an 8-frame ring
    c = cb.submit(*utterance(3, 500), max_new_frames=2)                         # cannot join a batch of length ~10
    d = cb.submit(*utterance(4, 3), max_new_frames=2)                           # ... but the one behind it can
    out = cb.run()
    assert [out[x].shape[0] for x in (a, b, c, d)] == [2, 30, 2, 2]
    assert torch.equal(out[b][:, 5], torch.arange(30) + 201)
    assert len(m.engines) == 2 and ("rewind",) in m.engines[0].log
    assert [x for x in m.engines[0].log if x[0] == "join"][0][1:3] == (0, 3)     # utterance 4 took over row 0
    assert m.engines[1].log[0] == ("prefill", 2, 500)                           # utterance 3 started the next batch (row 1 idle)


def test_cache_growth_is_requested_as_a_continuation():
    m = StubModel(max_len=0)
    cb = ContinuousBatcher(m, batch_size=1, topk=1, check_every=4, initial_frames=4)
    r = cb.submit(*utterance(7, 5), max_new_frames=80)
    out = cb.run()
    assert out[r].shape[0] == 80 and any(x[0] == "grow" for x in m.engines[0].log)
