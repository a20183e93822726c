"""CPU suite: the N>1 path (batch split + gather) under gloo with world_size 2 and 3, uneven shards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from csm_hf_amd.sharded import shard_rows, gather_frames, generate_sharded


class StubModel:
    """Row-wise deterministic stand-in for CSMModel.generate (the engine needs a GPU)."""

    def __init__(self, zero_rows_at=None, zero_all_at=None):
        self.zero_rows_at, self.zero_all_at = zero_rows_at, zero_all_at
        self.row_offset = 0

    def generate(self, ids, mask, max_new_frames=3, stop_on_all_zeros=False, **_):
        # like the engine's Philox counter, the output depends on the GLOBAL row index (row_offset + local row): the
        # sharded result equals the unsharded one only if generate_sharded hands every shard its row offset
        key = ids.sum(dim=(1, 2)) + mask.sum(dim=(1, 2)) * 7 + (self.row_offset + torch.arange(ids.shape[0])) * 13
        f = torch.arange(max_new_frames)[None, :, None]
        c = torch.arange(32)[None, None, :]
        out = (key[:, None, None] * 31 + f * 5 + c) % 2051 + 1          # never 0 by itself
        if self.zero_rows_at is not None:                               # some rows end early ...
            out[ids[:, 0, 0] % 2 == 0, self.zero_rows_at] = 0
        if self.zero_all_at is not None and self.zero_all_at < max_new_frames:   # ... every row ends here
            out[:, self.zero_all_at] = 0
        if stop_on_all_zeros:                                           # the reference's rule over THIS call's rows
            z = (out == 0).all(dim=2).all(dim=0)
            if bool(z.any()):
                out = out[:, : int(z.nonzero()[0])]
        return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 2051, (B, 6, 33), generator=g)
    mask = torch.ones(B, 6, 33, dtype=torch.int32)
    out = generate_sharded(StubModel(), ids, mask, max_new_frames=4, stop_on_all_zeros=False)
    ref = StubModel().generate(ids, mask, max_new_frames=4)
    ok = torch.equal(out, ref)
    # global stop: rows with an even key go silent at frame 1, every row at frame 3 -> the unsharded reference rule
    # returns 3 frames; a shard holding only even-key rows must not stop at frame 1
    ids[:, 0, 0] = torch.arange(B) + 1          # row 0 odd (never early), row 1 even (early), ...
    m = StubModel(zero_rows_at=1, zero_all_at=3)
    want = m.generate(ids, mask, max_new_frames=6, stop_on_all_zeros=True)
    got = generate_sharded(m, ids, mask, max_new_frames=6, stop_on_all_zeros=True)
    ok = ok and want.shape[1] == 3 and torch.equal(got, want)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_rows_partition():
    for n in (1, 2, 5, 16, 128):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_rows(128, 3, 8) == (48, 64)          # BASELINE config 4: 16 rows per GPU


@pytest.mark.parametrize("world,B", [(2, 5), (3, 4), (2, 1)])
def test_generate_sharded_gloo(world, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == list(range(world)) and all(ok for _, ok in res)


def test_explicit_noise_is_sliced_per_engine_pass():
    """generate_sharded(noise=[n, B_total, 32, V]): every engine pass (MAX_ROWS_PER_PASS rows) receives the draws of ITS rows (global row
    index), not the full-batch tensor (which CSMModel._check_noise rejects on shape)."""
    seen = []

    class NoiseModel(StubModel):
        def generate(self, ids, mask, max_new_frames=3, noise=None, **kw):
            assert noise is not None and noise.shape[1] == ids.shape[0], (noise.shape, ids.shape)
            seen.append((self.row_offset, ids.shape[0], float(noise[0, 0, 0, 0]), float(noise[0, -1, 0, 0])))
            return super().generate(ids, mask, max_new_frames=max_new_frames, **kw)

    from csm_hf_amd.sharded import MAX_ROWS_PER_PASS as P
    B, n, V = P + 6, 2, 5
    ids = torch.ones(B, 2, 33, dtype=torch.long)
    mask = torch.ones(B, 2, 33, dtype=torch.int32)
    noise = torch.arange(B, dtype=torch.float32)[None, :, None, None].expand(n, B, 32, V).contiguous()   # value = global row
    out = generate_sharded(NoiseModel(), ids, mask, max_new_frames=n, stop_on_all_zeros=False, noise=noise)
    assert out.shape[0] == B
    assert seen == [(0, P, 0.0, float(P - 1)), (P, 6, float(P), float(P + 5))]
