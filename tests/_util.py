"""Shared helpers of the GPU suites."""
import contextlib

import torch

EXACT_KV = torch.float32     # fp32 KV cache: a bf16-weight model then reproduces the reference's fp32-arithmetic token stream bit for bit
DEFAULT_KV = "auto"          # what CSMModel ships with: the cache follows the checkpoint dtype (bf16 checkpoint -> bf16 K / V)


@contextlib.contextmanager
def kv_mode(model, dtype):
    """Run a block with `model.kv_dtype = dtype` (the engine is rebuilt on the first call inside; the old mode comes back afterwards).
    Round 6: no suite-wide pin any more -- a test that asserts bit-exactness against the reference's fp32-arithmetic
    fixtures says `EXACT_KV` itself, everything else runs in the shipped default."""
    old = model.kv_dtype
    model.reset_caches()
    model.kv_dtype = dtype
    try:
        yield model
    finally:
        model.reset_caches()
        model.kv_dtype = old
