"""Batch-sharded generation across the GPUs of one node.

Utterances are independent -- the only cross-row operation in the reference is the global all-zero stop
test (`modeling_csm.py:662`) -- so the path shards by contiguous blocks of batch rows with replicated
weights, one process per GPU, and NO collective on the per-frame path (the global stop is applied after the
gather, see `generate_sharded`).  RCCL (`backend="nccl"`) is used
only to gather the finished `[rows, n, 32]` frames; the same code runs under `gloo` on CPU for tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of rows owned by `rank` (first `n_rows % world` ranks take one extra row)."""
    base, extra = divmod(n_rows, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_frames(local: torch.Tensor, n_rows: int, group=None) -> torch.Tensor:
    """all_gather ragged row blocks `[rows_r, n, C]` into `[n_rows, n, C]` on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return local
    rank = dist.get_rank(group)
    sizes = [shard_rows(n_rows, r, world) for r in range(world)]
    cap = max(b - a for a, b in sizes)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    assert sizes[rank][1] - sizes[rank][0] == local.shape[0]
    return torch.cat([bufs[r][: b - a] for r, (a, b) in enumerate(sizes)], dim=0)


MAX_ROWS_PER_PASS = 128   # rows one engine pass keeps on the matrix-core activation-plane path (csrc/engine.hip: PL_GROUPS)


def _generate_rows(model, ids, mask, row0: int, kw) -> torch.Tensor:
    """One shard's rows in engine-sized passes; the sampler sees GLOBAL row indices (row0 + local row)."""
    outs = []
    prev = getattr(model, "row_offset", 0)
    if "seed" not in kw and hasattr(model, "_next_seed"):
        kw = dict(kw, seed=model._next_seed())    # ONE seed per sharded call (same on every rank: same torch seed, same
        # call count), so with global row indices the sampled result does not depend on the world size or pass split
    try:
        for a in range(0, ids.shape[0], MAX_ROWS_PER_PASS):
            # the sampler's Philox counter uses the GLOBAL row (shard start + local row): every rank derives the same
            # seed from torch's seed, so without the offset local row i of every GPU would draw the same noise
            model.row_offset = row0 + a
            kwa = kw
            if kw.get("noise") is not None:   # explicit draws [n, B_total, 32, V]: the rows of this pass (GLOBAL row index)
                kwa = dict(kw, noise=kw["noise"][:, row0 + a: row0 + a + min(MAX_ROWS_PER_PASS, ids.shape[0] - a)])
            outs.append(model.generate(ids[a:a + MAX_ROWS_PER_PASS], mask[a:a + MAX_ROWS_PER_PASS], **kwa))
    finally:
        model.row_offset = prev
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def generate_sharded(model, input_ids: torch.Tensor, attention_mask: torch.Tensor, group=None, **gen_kwargs) -> torch.Tensor:
    """Every rank passes the FULL `[B, T, 33]` batch; each generates its own rows; all ranks return the
    full `[B, n, 32]` result.

    `stop_on_all_zeros=True` keeps the reference's GLOBAL semantics (`modeling_csm.py:662`: stop at the first
    frame that is all-zero in every row of the batch) without a collective on the per-frame path: every shard
    generates `max_new_frames` frames un-stopped, the gathered result is cut at the first globally all-zero
    frame.  Rows are independent, so the frames before the cut are exactly what a per-frame global test
    would have produced; the price is the frames generated past the cut."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    stop = bool(gen_kwargs.get("stop_on_all_zeros", True))
    if world == 1 and input_ids.shape[0] <= MAX_ROWS_PER_PASS:
        return model.generate(input_ids, attention_mask, **gen_kwargs)
    kw = dict(gen_kwargs, stop_on_all_zeros=False)
    a, b = shard_rows(input_ids.shape[0], rank, world)
    if b > a:
        local = _generate_rows(model, input_ids[a:b], attention_mask[a:b], a, kw)
    else:  # more ranks than rows
        n = int(gen_kwargs.get("max_new_frames", 100))
        local = torch.zeros(0, n, input_ids.shape[2] - 1, dtype=torch.long, device=input_ids.device)
    full = gather_frames(local, input_ids.shape[0], group) if world > 1 else local
    if stop and full.shape[1] > 0:
        zero = (full == 0).all(dim=2).all(dim=0)          # [n]: frame f is all-zero in every row
        if bool(zero.any()):
            full = full[:, : int(zero.nonzero()[0])]
    return full
