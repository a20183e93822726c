"""GPU suite, the SHIPPED DEFAULT of a bf16 checkpoint: `CSMModel.from_pretrained(..., torch_dtype=torch.bfloat16)` caches bf16 K / V
(`kv_dtype = "auto"`: the reference's own cache dtype -- a `DynamicCache` filled by a bf16 model, README.md:73), exact fp32 activations.
Rounds 1-5 pinned the whole suite to the fp32 cache; VERDICT r5 weak 2 asked for the default under the record.  This file runs FIRST
(tests/conftest.py) and holds, for BASELINE configs 1, 2 and 3:

  * the SURVEY 8-c tolerance protocol against the reference's fixtures, teacher-forced with the reference's tokens:
      last_hidden_state rel-L2 <= 1e-2 (SURVEY bar for bf16; measured ~1e-3), the reference's top-2 logits within 0.05 (measured ~5e-3),
      arg-max == the reference's token wherever the reference's top-1 margin exceeds 0.05, overall agreement > 98 %;
  * free-running through the PUBLIC API (`generate`, hipGraph replay, weight streamer on): a bf16-rounded K / V moves a logit by ~2e-2
    (measured), the streams' smallest top-1 margins are ~1e-6, so the free-running stream leaves the reference's within a frame or two
    ("why free-running bf16 equality cannot be the criterion") and a prefix rule would be vacuous.  Instead EVERY sample of the
    free-running stream is classified (SURVEY 8-c ii): the EXACT engine -- which reproduces the reference's fp32-arithmetic stream bit for
    bit (test_gpu_generate.py) -- is teacher-forced with the default mode's own tokens; a token is "within margin" when the exact engine's
    logit for it is within LOGIT_TOL of the exact engine's maximum (measured: <= 0.021), anything else is a real mismatch and fails the test;
  * the invariants between launch shapes that the exact-mode suites assert -- fused vs stand-alone sampler, hipGraph vs eager, gemm128 vs
    gemm32 FFN launches, fused vs separate split merge -- bit for bit in THIS mode too, and the summation-order ones
    (key-sharing attention kernels on / off, split merge inside the o_proj launch, a row in another batch) at this mode's distance: a
    1e-6 difference in a hidden state can flip the bf16 rounding of a K / V element written afterwards, so two fp32-equivalent launch
    shapes sit ~1e-3 apart in later frames (fp32 cache: ~5e-6; tools/probes/ab_option_diff.py) -- frame 0 is held to the fp32 bar.
"""
import numpy as np
import pytest
import torch

from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
from _util import EXACT_KV, kv_mode

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

H_TOL, LOGIT_TOL, SAFE_MARGIN = 1e-2, 0.05, 0.05
SHAPE_TOL = 1e-2     # two launch shapes of the bf16-cache mode, later frames (a flipped bf16 rounding of a stored K / V element)


@pytest.fixture(scope="module")
def csm1b_default():
    cfg = CSMConfig()
    sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=DEV, bf16_representable=True)
    m = CSMModel(cfg)
    m.load_state_dict(sd)
    del sd
    assert CSMModel.DEFAULT_KV_DTYPE == "auto" and m.kv_dtype == "auto"      # nothing pins the suite any more
    yield m.eval()
    m._drop_engine()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def traced(model, ids, mask, n, forced=None, topk=1, temperature=1.0, use_graph=True, options=None):
    """engine-level generate with logits / last_h traces (what the golden files hold); `options`: engine options for this run only"""
    B, T = ids.shape[:2]
    eng = model._ensure_engine(B, T + n + 1, max(n, 1), B * T)
    assert eng.kv_dtype == torch.bfloat16, "this suite is about the bf16 KV cache"
    for k, v in (options or {}).items():
        eng.set_option(k, v)
    eng.reset()
    eng.set_kv_start(model._kv_starts(mask, B, T))
    C, V, Hb = eng.C, eng.V, eng.Hb
    lt = torch.zeros(eng.max_frames, B, C, V, dtype=torch.float32, device=DEV)
    ht = torch.zeros(eng.max_frames, B, Hb, dtype=torch.float32, device=DEV)
    fz = None
    if forced is not None:
        fz = torch.zeros(B, eng.max_frames, C, dtype=torch.int64, device=DEV)
        fz[:, :n] = forced.to(DEV)
    lh, _ = eng.prefill(ids, mask)
    ht[0] = lh
    eng.generate(eng.sampling(temperature=temperature, topk=topk, seed=7, forced=fz, logits_trace=lt, last_h_trace=ht), n, use_graph)
    toks = eng.read_frames(0, n).cpu()
    return toks, lt[:n].cpu(), ht[:n].cpu()


def tolerance_protocol(lt, ht, g, rows, what):
    """SURVEY 8-c, teacher-forced: `lt` [n, B, C, V], `ht` [n, B, H] of the engine; fixture `g` holds the reference's values for `rows`."""
    top_idx, top_val = g["top_idx"], g["top_vals"]                  # [n, rows, C, k]
    mine = lt[:, :rows]
    d_h = rel_l2(ht[:, :rows], torch.from_numpy(g["last_h"]))
    err = float(np.abs(np.take_along_axis(mine.numpy(), top_idx.astype(np.int64), -1) - top_val).max())
    margin = top_val[..., 0] - top_val[..., 1]
    ref_tok = g["tokens"].transpose(1, 0, 2)                        # [n, rows, C]
    am = mine.argmax(-1).numpy()
    safe = margin > SAFE_MARGIN
    agree = float((am == ref_tok).mean())
    print(f"{what}: last_h rel-L2 {d_h:.3e} (bar {H_TOL}), top-2 logit |err| {err:.4f} (bar {LOGIT_TOL}), arg-max agreement {agree:.4f}, "
          f"{int(safe.sum())} of {safe.size} samples with margin > {SAFE_MARGIN}")
    assert d_h < H_TOL and err < LOGIT_TOL, (what, d_h, err)
    assert np.array_equal(am[safe], ref_tok[safe]), (what, "arg-max left the reference at a margin above", SAFE_MARGIN)
    assert agree > 0.98, (what, agree)


def classify_free_running(m, ids, mask, toks, what):
    """`toks` [B, n, C]: the default mode's own free-running stream.  The exact engine, teacher-forced with it, says for every sample
    how far the chosen token's logit is from its own maximum: 0 = the exact engine would have picked it too."""
    from test_gpu_generate import traced_generate
    n = toks.shape[1]
    with kv_mode(m, EXACT_KV):
        _, lx, _ = traced_generate(m, ids, mask, n, forced=toks)
    assert m.kv_dtype == "auto"
    chosen = torch.gather(lx, -1, toks.permute(1, 0, 2).unsqueeze(-1)).squeeze(-1)      # [n, B, C]
    gap = lx.max(-1)[0] - chosen
    same = float((gap == 0).float().mean())
    worst = float(gap.max())
    print(f"{what}: free-running stream of {gap.numel()} samples: {same:.4f} are the exact engine's own arg-max, the others lie within "
          f"{worst:.4f} of its maximum (bar {LOGIT_TOL})")
    assert worst < LOGIT_TOL, (what, worst)
    assert same > 0.97, (what, same)


def test_default_config1_csm1b_64ctx_8frames(gold, csm1b_default):
    """BASELINE config 1 shape in the default mode, B = 1 and the same row 16 times (matrix-core kernels): teacher-forced with the
    tokens of the reference's OWN bf16 run (fixture csm1b_cfg1_bf16: every activation bf16 -- a wider class than this mode), last_h
    rel-L2 <= 5e-2 against that run and no farther from it than the exact engine is (x 1.05), top logits within 0.1, arg-max inside the
    reference's top-4 for > 99 %; and against the fp32-arithmetic fixture by the protocol of this file; the distance to the exact
    (fp32-cache) engine is a bf16-rounding distance; equal rows stay bitwise equal."""
    m = csm1b_default
    gb, g = gold("csm1b_cfg1_bf16"), gold("csm1b_cfg1_bf16w_fp32")
    ids1, mask1 = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    for B in (1, 16):
        ids, mask = ids1.repeat(B, 1, 1), mask1.repeat(B, 1, 1)
        forced_b = torch.from_numpy(gb["tokens"]).repeat(B, 1, 1)
        _, lt_b, ht_b = traced(m, ids, mask, 8, forced=forced_b)
        with kv_mode(m, EXACT_KV):
            eng = m._ensure_engine(B, 64 + 9, 8, B * 64)
            assert eng.kv_dtype == torch.float32
            from test_gpu_generate import traced_generate
            _, lt_x, ht_x = traced_generate(m, ids, mask, 8, forced=forced_b)
        for r in range(1, B):
            assert torch.equal(lt_b[:, 0], lt_b[:, r]), r
        ref_h = torch.from_numpy(gb["last_h"])
        d_ref, d_ref_x, d_exact = rel_l2(ht_b[:, :1], ref_h), rel_l2(ht_x[:, :1], ref_h), rel_l2(ht_b, ht_x)
        top_idx, top_val = gb["top_idx"], gb["top_vals"]
        mine = lt_b[:, :1]
        err = np.abs(np.take_along_axis(mine.numpy(), top_idx.astype(np.int64), -1) - top_val).max()
        in_top4 = (mine.argmax(-1).numpy()[..., None] == top_idx).any(-1).mean()
        print(f"bf16 KV cache, B = {B}: last_h vs the reference's bf16 run {d_ref:.3e} (fp32 cache: {d_ref_x:.3e}), vs the fp32-cache engine "
              f"{d_exact:.3e}; top-logit |err| {err:.3f}; arg-max in the reference's top-4: {in_top4:.4f}")
        assert d_ref < 5e-2 and d_ref <= 1.05 * d_ref_x and err < 0.1 and in_top4 > 0.99
        assert 1e-5 < d_exact < 2e-2, d_exact
        # the fp32-arithmetic fixture, teacher-forced with ITS tokens
        _, lt, ht = traced(m, ids, mask, 8, forced=torch.from_numpy(g["tokens"]).repeat(B, 1, 1))
        tolerance_protocol(lt, ht, g, 1, f"config 1, B = {B}")
    out = m.generate(ids1.to(DEV), mask1.to(DEV), max_new_frames=8, topk=1, stop_on_all_zeros=False).cpu()
    assert m._engine.kv_dtype == torch.bfloat16
    classify_free_running(m, ids1, mask1, out, "config 1")


def test_default_config2_csm1b_512ctx_200frames(gold, csm1b_default):
    """BASELINE config 2 (the benchmarked workload) in the default mode: 512-frame context + 200 greedy frames through `generate`
    (hipGraph replay, weight streamer on), against the reference's fp32-arithmetic run on the same weights."""
    m = csm1b_default
    g = gold("csm1b_cfg2_bf16w_fp32")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    n = g["tokens"].shape[1]
    assert n == 200
    _, lt, ht = traced(m, ids, mask, n, forced=torch.from_numpy(g["tokens"]))
    tolerance_protocol(lt, ht, g, 1, "config 2")
    out = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=n, topk=1, stop_on_all_zeros=False).cpu()
    assert tuple(out.shape) == (1, n, 32) and m._engine.kv_dtype == torch.bfloat16
    st = m._engine.prefetch_stats()
    assert st["health"]["disabled"] == 0 and st["finished"] + st["gave_up"] > 0, repr(st)     # the streamer ran beside this call
    classify_free_running(m, ids, mask, out, "config 2")
    # eager launches give the graph's stream bit for bit
    m.use_graph = False
    try:
        eager = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=20, topk=1, stop_on_all_zeros=False).cpu()
    finally:
        m.use_graph = True
    assert torch.equal(eager, out[:, :20])


def test_default_config3_csm1b_batch16_voiceclone_and_sampler(gold, csm1b_default):
    """BASELINE config 3 shape in the default mode: B = 16, 512-frame voice-clone layout, rows 0-3 = the reference fixture's rows.
    Greedy: the protocol of this file on rows 0-3; the rows do not depend on what else is in the batch (B = 4 alone: bit-equal logits).
    top-k 50 / T = 1.0 with the reference's explicit Exp(1) noise, teacher-forced: every draw is an independent comparison; frame 0
    draw for draw, >= 97 % overall (a draw whose race margin is inside the bf16-cache logit distance may go the other way)."""
    m = csm1b_default
    cfg = m.config
    g = gold("csm1b_b4_ctx512_bf16w_fp32")
    ids, mask = synth_context(cfg, 16, 48, 400, seed=3, tail_text=63, eos_frame=True)
    assert np.array_equal(ids[:4].numpy(), g["input_ids"])
    n = g["tokens"].shape[1]
    forced = torch.cat([torch.from_numpy(g["tokens"]), torch.from_numpy(g["tokens"]).repeat(3, 1, 1)], 0)     # rows 4-15: some tokens
    _, lt, ht = traced(m, ids, mask, n, forced=forced)
    tolerance_protocol(lt, ht, g, 4, "config 3 (B = 16)")
    _, lt4, ht4 = traced(m, ids[:4], mask[:4], n, forced=forced[:4])     # (another number of KV splits: summation order)
    d0, d = float((lt4[0] - lt[0, :4]).abs().max()), float((lt4 - lt[:, :4]).abs().max())
    print(f"config 3: rows 0-3 alone vs inside the 16-row batch: logits |diff| frame 0 {d0:.2e}, all frames {d:.2e}")
    assert d0 < 1e-4 and d < SHAPE_TOL
    out = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=n, topk=1, stop_on_all_zeros=False).cpu()
    classify_free_running(m, ids, mask, out, "config 3")
    # sampler in situ
    gs = gold("csm1b_b4_topk50_noise_bf16w_fp32")
    ids_s, mask_s = torch.from_numpy(gs["input_ids"]), torch.from_numpy(gs["attention_mask"])
    ref = torch.from_numpy(gs["tokens"])
    ns, C, V = ref.shape[1], cfg.audio_num_codebooks, cfg.audio_vocab_size
    noise = torch.empty(ns, 4, C, V).exponential_(1, generator=torch.Generator().manual_seed(int(gs["noise_seed"])))
    eng = m._ensure_engine(4, 512 + ns + 1, max(ns, 1), 4 * 512)
    assert eng.kv_dtype == torch.bfloat16
    eng.reset()
    eng.set_kv_start([0] * 4)
    eng.prefill(ids_s, mask_s)
    fz = torch.zeros(4, eng.max_frames, C, dtype=torch.int64, device=DEV)
    fz[:, :ns] = ref.to(DEV)
    for f in range(ns):
        eng.generate(eng.sampling(temperature=1.0, topk=50, noise=noise[f].to(DEV).contiguous(), forced=fz), 1, True)
        eng.sync()
    got = eng.read_frames(0, ns).cpu()
    same = (got == ref)
    print(f"config 3 sampler in situ: {int(same.sum())} of {same.numel()} draws equal to the reference's")
    assert float(same.float().mean()) >= 0.97, int((~same).sum())
    assert float(same[:, 0].float().mean()) >= 0.99, "frame 0 (no accumulated difference)"


@pytest.mark.parametrize("B,T,n,name,values,topk", [
    (1, 64, 8, "fuse_sample", (1, 0), 1),
    (1, 64, 8, "fuse_sample", (1, 0), 50),
    (5, 300, 3, "fuse_attn_combine", (1, 0), 1),
    (16, 64, 4, "fuse_sample", (1, 0), 50),
    (70, 24, 3, "g128", (1, 0), 1),
])
def test_default_mode_launch_shape_invariants_are_bitwise(csm1b_default, B, T, n, name, values, topk):
    """The exact-mode suites assert these between launch shapes of the fp32-cache engine; the bf16-cache instantiations of the same
    kernels (attn_decode_gqa_kernel<bf16>, attn_oproj_gqa_kernel<bf16>, the fused QKV epilogues that write bf16 K / V) are held to the
    same bar: logits, last_h and tokens equal bit for bit (teacher-forced, so every frame is compared whatever the first one did)."""
    m = csm1b_default
    cfg = m.config
    ids, mask = synth_context(cfg, B, T // 4, T - T // 4, seed=31)
    if B > 1:
        ids[1, :7] = 0
        mask[1, :7] = 0          # one left-padded row
    forced = torch.randint(0, cfg.audio_vocab_size, (B, n, cfg.audio_num_codebooks), generator=torch.Generator().manual_seed(5))
    outs = []
    try:
        for v in values:
            outs.append(traced(m, ids, mask, n, forced=forced, topk=topk, temperature=0.9 if topk > 1 else 1.0, options={name: v}))
    finally:
        m._engine.set_option(name, values[0])
    for o in outs[1:]:
        for a, b, what in zip(outs[0], o, ("tokens", "logits", "last_h")):
            assert torch.equal(a, b), (name, B, what, float((a.double() - b.double()).abs().max()))
    assert float(outs[0][1].abs().max()) > 0.1


@pytest.mark.parametrize("name", ["attn_oproj_gqa", "attn_gqa_wide", "oproj_combine"])
def test_default_mode_summation_order_options(csm1b_default, name):
    """The kernels that share a K / V tile among the query heads of a kv-head, and the split merge inside the o_proj launch (another number
    of KV splits), sum the keys in another order than the forms they replace (fp32 summation-order class, ~5e-6 on the logits with the
    fp32 cache).  With the bf16 cache that difference can flip the rounding of a K / V element stored afterwards: frame 0's codebook-0
    logits (context from the prefill only) within 1e-4, everything within SHAPE_TOL, last_h rel-L2 1e-4, same arg-max wherever the
    margin exceeds 2 x SHAPE_TOL."""
    m = csm1b_default
    cfg = m.config
    B, T, n = {"attn_oproj_gqa": (1, 200, 4), "attn_gqa_wide": (40, 64, 3), "oproj_combine": (1, 300, 4)}[name]
    ids, mask = synth_context(cfg, B, T // 4, T - T // 4, seed=33)
    forced = torch.randint(0, cfg.audio_vocab_size, (B, n, cfg.audio_num_codebooks), generator=torch.Generator().manual_seed(6))
    try:
        a = traced(m, ids, mask, n, forced=forced, options={name: 1})
        b = traced(m, ids, mask, n, forced=forced, options={name: 0})
    finally:
        m._engine.set_option(name, 1)
    err = float((a[1] - b[1]).abs().max())
    d_h = rel_l2(a[2], b[2])
    err0 = float((a[1][0, :, 0] - b[1][0, :, 0]).abs().max())
    print(f"{name} on / off, bf16 cache: logits |diff| {err:.2e} (frame 0, codebook 0: {err0:.2e}), last_h rel-L2 {d_h:.2e}")
    assert 0 < err < SHAPE_TOL and err0 < 1e-4 and d_h < 1e-4, (err, err0, d_h)
    top2 = torch.topk(b[1], 2, -1)[0]
    safe = (top2[..., 0] - top2[..., 1]) > 2 * SHAPE_TOL
    assert torch.equal(a[1].argmax(-1)[safe], b[1].argmax(-1)[safe])
