"""GPU suite, round 5: query-tile coverage of the bf16-class context attention at shapes whose tile count does not divide
256, and the round's other additions (see the individual tests)."""
import os

import numpy as np
import pytest
import torch

from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
from oracle import csm_oracle as O
from _util import EXACT_KV, kv_mode

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
    # EXACT mode, said here (round 6: no suite-wide pin): the tests on this fixture compare bit for bit with the reference's fp32-arithmetic
    # run on the same bf16 weights (fixtures *_bf16w_fp32) or between launch shapes of that mode; the shipped default (bf16 KV cache) has
    # its own suite, tests/test_gpu_default_mode.py, which runs first
    m.kv_dtype = EXACT_KV
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


def _b1_sampled(m, ids, mask, n, fuse, noise=None, topk=50, temperature=0.9, seed=11):
    eng = m._ensure_engine(1, ids.shape[1] + n + 1, max(n, 1), ids.shape[1])
    eng.set_option("fuse_sample", fuse)
    try:
        eng.reset()
        eng.set_kv_start([0])
        eng.prefill(ids, mask, want_outputs=False)
        if noise is None:
            eng.generate(eng.sampling(temperature=temperature, topk=topk, seed=seed), n, True)
        else:
            for f in range(n):        # one [B, C, V] noise block per call
                nz = noise[f].to(DEV).contiguous()
                eng.generate(eng.sampling(temperature=temperature, topk=topk, noise=nz), 1, True)
                eng.sync()
        eng.sync()
        return eng.read_frames(0, n).cpu(), eng.graph_stats()
    finally:
        eng.set_option("fuse_sample", 1)


def test_b1_fused_topk_sampler_equals_the_sampler_launch_and_the_oracle(csm1b_bf16):
    """Round 5: at B = 1 the top-k sampler of codebooks 1..30 runs inside the next decoder pass's first QKV
    launch (sample_wave.h: one wavefront per logits row, no workgroup barrier, sample_kernel's arithmetic term for term).
    (i) Philox draws: the sampled stream equals the stand-alone sampler launch's (`fuse_sample = 0`) token for token over 12
    frames x 32 codebooks at top-k 50 / T = 0.9 (the reference's usual call) and at top-k 5 / T = 1.3 / another seed.
    (ii) explicit Exp(1) noise: both equal the ORACLE's `sample_topk` (reference modeling_csm.py:170-189) run on the CPU on the
    engine's own logits is not possible without a trace, so the oracle comparison is end to end: oracle.generate with the same
    noise on the same weights / context (fp32 arithmetic), frame 0 exact and >= 99 % of all draws (a draw downstream of a
    near-tie may differ)."""
    m = csm1b_bf16
    cfg = m.config
    ids, mask = synth_context(cfg, 1, 16, 48, seed=21)
    n = 12
    for topk, temp, seed in ((50, 0.9, 11), (5, 1.3, 12), (2051, 1.0, 13)):
        a, _ = _b1_sampled(m, ids, mask, n, 1, topk=topk, temperature=temp, seed=seed)
        b, _ = _b1_sampled(m, ids, mask, n, 0, topk=topk, temperature=temp, seed=seed)
        assert torch.equal(a, b), (topk, temp, int((a != b).sum()))
        assert int(a.max()) < cfg.audio_vocab_size and int(a.min()) >= 0
    # explicit noise: fused == unfused == oracle
    C, V = cfg.audio_num_codebooks, cfg.audio_vocab_size
    n = 3
    noise = torch.empty(n, 1, C, V).exponential_(1, generator=torch.Generator().manual_seed(5))
    a, _ = _b1_sampled(m, ids, mask, n, 1, noise=noise, topk=50, temperature=1.0)
    b, _ = _b1_sampled(m, ids, mask, n, 0, noise=noise, topk=50, temperature=1.0)
    assert torch.equal(a, b), int((a != b).sum())
    # teacher-forced (the fed-back row comes from `forced`, the samples are still recorded) and per-row stop: fused == stand-alone launches
    forced = torch.randint(0, V, (1, 64, C), generator=torch.Generator().manual_seed(9))
    rec = []
    for fuse in (1, 0):
        eng = m._ensure_engine(1, ids.shape[1] + 7, 64, ids.shape[1])
        eng.set_option("fuse_sample", fuse)
        try:
            eng.reset()
            eng.set_kv_start([0])
            eng.prefill(ids, mask, want_outputs=False)
            fz = torch.zeros(1, eng.max_frames, C, dtype=torch.int64, device=DEV)
            fz[:, :64] = forced.to(DEV)
            eng.generate(eng.sampling(temperature=0.8, topk=20, seed=77, forced=fz, per_row_stop=True), 6, True)
            eng.sync()
            rec.append(eng.read_frames(0, 6).cpu())
        finally:
            eng.set_option("fuse_sample", 1)
    assert torch.equal(rec[0], rec[1]), int((rec[0] != rec[1]).sum())
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    want = O.generate(sd, cfg, ids, mask, max_new_frames=n, temperature=1.0, topk=50, stop_on_all_zeros=False, noise=noise)
    same = (a == want)
    assert bool(same[:, 0].all()), "frame 0 must match the oracle draw for draw"
    assert float(same.float().mean()) >= 0.99, int((~same).sum())


def _traced(model, ids, mask, n, forced):
    """engine-level teacher-forced generate with logits / last_h traces (tests/test_gpu_generate.py traced_generate)."""
    B, T = ids.shape[:2]
    eng = model._ensure_engine(B, T + n + 1, max(n, 1), B * T)
    eng.reset()
    eng.set_kv_start(model._kv_starts(mask, B, T))
    C, V, Hb = eng.C, eng.V, eng.Hb
    lt = torch.zeros(eng.max_frames, B, C, V, dtype=torch.float32, device=DEV)
    ht = torch.zeros(eng.max_frames, B, Hb, dtype=torch.float32, device=DEV)
    fz = torch.zeros(B, eng.max_frames, C, dtype=torch.int64, device=DEV)
    fz[:, :n] = forced.to(DEV)
    lh, _ = eng.prefill(ids, mask)
    ht[0] = lh
    eng.generate(eng.sampling(temperature=1.0, topk=1, seed=7, forced=fz, logits_trace=lt, last_h_trace=ht), n, True)
    eng.sync()
    return lt[:n].cpu(), ht[:n].cpu()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("weights", ["bf16", "fp8"])
def test_ffn_launches_beyond_64_rows_on_gemm128_are_bitwise_the_gemm32_launches(csm1b_bf16, weights):
    """65..128 batched rows: the FFN launches (gate/up with the RMS scale + SwiGLU, down_proj with the residual + output planes +
    sums of squares; decoder and backbone shapes, K split across workgroups and two k groups in one workgroup) run on gemm128_kernel
    (weight rows split over the waves, planes through LDS, csrc/gemm128.h) and must leave every logit and the backbone's last hidden
    state equal, bit for bit, to the gemm32_kernel launches (`g128 = 0`): 70 rows (a partial fifth batch tile: clamped plane tile,
    masked rows), 96, 128; exact planes and `decode_precision = "bf16"` (one plane); bf16 and fp8 weights (fragments widened in
    registers, per-row scale in the epilogue); the other shape (one weight tile per wave, one k group per workgroup: `g128_shape = 65`) as well."""
    m = csm1b_bf16
    cfg = m.config
    m._drop_engine()
    m.weight_format = "fp8" if weights == "fp8" else "native"
    try:
        for B, precs in ((70, ("exact",)), (96, ("exact", "bf16")), (128, ("exact", "bf16"))):
            ids, mask = synth_context(cfg, B, 10, 14, seed=61)
            forced = torch.randint(0, cfg.audio_vocab_size, (B, 3, cfg.audio_num_codebooks), generator=torch.Generator().manual_seed(5))
            for prec in precs:
                m.decode_precision = prec
                outs = []
                for opts in ({"g128": 0}, {"g128": 1}, {"g128": 1, "g128_shape": 64 + 1}):
                    T = ids.shape[1]
                    eng = m._ensure_engine(B, T + 3 + 1, 3, B * T)
                    for k, v in {"g128": 1, "g128_shape": 0, **opts}.items():
                        eng.set_option(k, v)
                    outs.append(_traced(m, ids, mask, 3, forced))
                    assert m._engine is eng and (weights == "fp8") == bool(eng.fp8)
                for o in outs[1:]:
                    assert torch.equal(outs[0][0], o[0]), (B, prec, weights, "logits", float((outs[0][0] - o[0]).abs().max()))
                    assert torch.equal(outs[0][1], o[1]), (B, prec, weights, "last_h")
                assert float(outs[0][0].abs().max()) > 0
    finally:
        m.decode_precision = "exact"
        m._drop_engine()
        m.weight_format = "native"
