"""GPU suite, round 5: query-tile coverage of the bf16-class context attention at shapes whose tile count does not divide
256, and the round's other additions (see the individual tests)."""
import os

import numpy as np
import pytest
import torch

from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
from oracle import csm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def csm1b_bf16():
    cfg = CSMConfig()
    sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=DEV, bf16_representable=True)
    m = CSMModel(cfg)
    m.load_state_dict(sd)
    del sd
    yield m.eval()
    m._drop_engine()


def _last_layer_kv(m, ids, mask, mode):
    m.prefill_precision = mode
    try:
        m.forward(ids.to(DEV), mask.to(DEV), use_cache=True)
        k, v = m._engine.export_kv()[-1]
        return k.cpu(), v.cpu(), m._engine.get_state()[0].cpu()
    finally:
        m.prefill_precision = "exact"


@pytest.mark.parametrize("S,B", [(1100, 1), (96, 16), (300, 4)])
@pytest.mark.parametrize("mode", ["bf16", "mxfp8"])
def test_context_attention_covers_every_query_tile(csm1b_bf16, S, B, mode):
    """`attn_prefill_bf16_kernel` walks the query tiles of every other wave of 256 workgroups in ascending
    order.  Round 4 chose the direction from the workgroup's own linear id, so a 256-boundary inside one (kv-head, sequence)
    row of the grid left tiles [0, min(c, gx - c)) uncomputed (stale rows in the attention output, hence in every later
    layer's K / V) whenever ceil(S / 32) did not divide 256 -- e.g. (S, B) = (1100, 1), (96, 16), (300, 4).  The last layer's
    K / V of EVERY position depend on every earlier layer's attention output at that position, so they are compared
    position by position with the exact mode on the same context, after a run on a DIFFERENT context has filled the
    buffers with foreign values.  Bar: per-position rel-L2 <= 0.15 for bf16 (the mode's class distance, measured <= 0.06),
    <= 0.8 for mxfp8 (class distance 0.36 overall); a stale tile shows as ~1.4 on its 32 positions."""
    m = csm1b_bf16
    ids, mask = synth_context(m.config, B, S // 4, S - S // 4, seed=11)
    other, omask = synth_context(m.config, B, S // 4, S - S // 4, seed=12)
    ke, ve, he = _last_layer_kv(m, ids, mask, "exact")
    _last_layer_kv(m, other, omask, mode)          # foreign values in every scratch buffer of the mode
    kb, vb, hb = _last_layer_kv(m, ids, mask, mode)
    assert torch.isfinite(kb).all() and torch.isfinite(vb).all()

    def per_pos(a, b):   # [B, n_kv, L, hd] -> rel-L2 per (row, position)
        num = (a.double() - b.double()).pow(2).sum(dim=(1, 3)).sqrt()
        den = b.double().pow(2).sum(dim=(1, 3)).sqrt()
        return (num / den)

    bar = 0.15 if mode == "bf16" else 0.8
    ek, ev = per_pos(kb, ke), per_pos(vb, ve)
    assert float(ek.max()) < bar and float(ev.max()) < bar, (float(ek.max()), float(ev.max()), int(ek.argmax()), int(ev.argmax()))
