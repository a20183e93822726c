"""GPU suite, end-to-end parity of the hot path through the C ABI: prefill + hipGraph-replayed frame
loop against (a) the committed golden vectors produced by the reference and (b) the oracle on fresh
seeded inputs.

Tolerances (stated, per BASELINE.json north_star):
  * fp32 model: token ids bit-exact; last_hidden_state / logits |err| <= 2e-4 (fp32 summation order).
  * bf16-weight model, fp32 activations+KV (default engine numerics): token ids bit-exact against the
    reference run in fp32 arithmetic on the same (bf16-representable) weights wherever the reference's
    top-1 margin exceeds 1e-4; hidden state rel-L2 <= 1e-4.
  * against the reference's own bf16 run (which rounds every activation to bf16): last_hidden_state
    rel-L2 <= 5e-2 -- that is the reference's own bf16-vs-fp32 discrepancy (2.8e-2 measured on the
    csm-1b config-1 fixture; frame-0 value asserted below) -- and the engine's argmax lies in the
    reference's near-tie set (margin-aware).
"""
import numpy as np
import pytest
import torch

from csm_hf_amd import CSMConfig, CSMModel
from csm_hf_amd.synth import synth_state_dict, synth_context
from oracle import csm_oracle as O
from _util import EXACT_KV, kv_mode

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_model(cfg, sd, dtype):
    m = CSMModel(cfg)
    m.load_state_dict({k: v.to(dtype) for k, v in sd.items()})
    return m.to(DEV).eval()


def traced_generate(model, ids, mask, n, forced=None, use_graph=True, topk=1, temperature=1.0, noise=None):
    """engine-level generate with logits / last_h traces (what the golden files hold)."""
    B, T = ids.shape[:2]
    eng = model._ensure_engine(B, T + n + 1, max(n, 1), B * T)
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
    nz = None if noise is None else noise.to(DEV).contiguous()
    s = eng.sampling(temperature=temperature, topk=topk, seed=7, forced=fz, logits_trace=lt, last_h_trace=ht, noise=nz)
    eng.generate(s, n, use_graph)
    toks = eng.read_frames(0, n).cpu()
    assert eng.device_counters() == (T + n, n)
    return toks, lt[:n].cpu(), ht[:n].cpu()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


# ---------------------------------------------------------------------------------------------------
# tiny config: reference golden vectors
# ---------------------------------------------------------------------------------------------------
def test_tiny_fp32_matches_reference_golden(gold):
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    g = gold("tiny_fp32")
    m = make_model(cfg, sd, torch.float32)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    for use_graph in (False, True):
        toks, lt, ht = traced_generate(m, ids, mask, 4, use_graph=use_graph)
        assert np.array_equal(toks.numpy(), g["tokens"]), f"graph={use_graph}"
        np.testing.assert_allclose(lt.numpy(), g["logits"], atol=2e-4, rtol=0)
        np.testing.assert_allclose(ht.numpy(), g["last_h"], atol=2e-4, rtol=0)
    # public API gives the same tokens
    out = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=4, temperature=1.0, topk=1, stop_on_all_zeros=False)
    assert out.dtype == torch.long and out.device.type == "cuda"
    assert np.array_equal(out.cpu().numpy(), g["tokens"])
    out0 = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=4, temperature=0.0, topk=50, stop_on_all_zeros=False)
    assert np.array_equal(out0.cpu().numpy(), g["tokens"])            # temperature 0 == argmax


def test_tiny_bf16_model_vs_reference(gold):
    """bf16 checkpoint: (i) bit-exact tokens vs the oracle in fp32 arithmetic on the same bf16 weights,
    (ii) teacher-forced tolerance vs the reference's own bf16 run."""
    cfg = CSMConfig.tiny()
    sd = {k: v.to(torch.bfloat16) for k, v in synth_state_dict(cfg, seed=0, std=0.05).items()}
    g = gold("tiny_bf16")
    m = make_model(cfg, sd, torch.bfloat16)
    m.kv_dtype = EXACT_KV      # this test asserts the exact mode (fp32 KV cache) against fp32-arithmetic values
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    tr = {}
    otoks = O.generate({k: v.float() for k, v in sd.items()}, cfg, ids, mask, max_new_frames=4, topk=1,
                       stop_on_all_zeros=False, trace=tr)
    toks, lt, ht = traced_generate(m, ids, mask, 4)
    tv = torch.topk(tr["logits"], 2, -1)[0]
    assert float((tv[..., 0] - tv[..., 1]).min()) > 1e-4
    assert torch.equal(toks, otoks)
    np.testing.assert_allclose(lt.numpy(), tr["logits"].numpy(), atol=2e-4, rtol=0)
    # teacher-forced against the reference's bf16 stream
    ref_tok = torch.from_numpy(g["tokens"])
    toks_f, lt_f, ht_f = traced_generate(m, ids, mask, 4, forced=ref_tok)
    ref_logits = torch.from_numpy(g["logits"])
    assert rel_l2(ht_f, torch.from_numpy(g["last_h"])) < 5e-2
    assert float((lt_f - ref_logits).abs().max()) < 0.15
    # margin-aware: our argmax must sit within the reference's near-tie set
    mine = lt_f.argmax(-1)
    ref_at_mine = ref_logits.gather(-1, mine[..., None])[..., 0]
    assert float((ref_logits.max(-1)[0] - ref_at_mine).max()) < 0.15


def test_tiny_padded_batch_rows_equal_solo_reference(gold):
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    g = gold("tiny_padded")
    m = make_model(cfg, sd, torch.float32)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    out = m.generate(ids.to(DEV), mask.float().to(DEV), max_new_frames=4, topk=1, stop_on_all_zeros=False)
    assert np.array_equal(out.cpu().numpy(), g["tokens"])


def test_tiny_api_generate_frame_forward_and_errors():
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    m = make_model(cfg, sd, torch.float32)
    ids, mask = synth_context(cfg, 2, 3, 5, seed=21)
    want = O.generate(sd, cfg, ids, mask, max_new_frames=3, topk=1, stop_on_all_zeros=False)
    # manual loop exactly like the reference's generate() body, through generate_frame + past_key_values
    m.setup_caches(2)
    pkv, cur, cm, got = None, ids.to(DEV), mask.to(DEV), []
    for _ in range(3):
        out = m.generate_frame(cur, cm, temperature=1.0, topk=1, past_key_values=pkv, return_dict=True)
        assert out.samples.shape == (2, 32) and out.samples.dtype == torch.long
        assert out.last_hidden_state.shape == (2, cfg.backbone_config.hidden_size) and out.logits.shape == (2, cfg.audio_vocab_size)
        got.append(out.samples)
        pkv = out.past_key_values
        cur = torch.cat([out.samples, torch.zeros(2, 1, dtype=torch.long, device=DEV)], 1).unsqueeze(1)
        cm = torch.zeros(2, 1, 33, dtype=mask.dtype, device=DEV)
        cm[:, :, :32] = 1
    assert torch.equal(torch.stack(got, 1).cpu(), want)
    # forward(): hidden/logits vs oracle; tuple return; stale cache handle; mis-shaped labels
    o = m.forward(ids.to(DEV), mask.to(DEV), use_cache=True)
    lh, lg, _ = O.forward(sd, cfg, ids, mask)
    torch.testing.assert_close(o.last_hidden_state.cpu(), lh, atol=2e-4, rtol=0)
    torch.testing.assert_close(o.logits.cpu(), lg, atol=2e-4, rtol=0)
    t = m.forward(ids.to(DEV), mask.to(DEV), use_cache=True, return_dict=False)
    assert isinstance(t, tuple) and len(t) == 3
    with pytest.raises(ValueError):
        m.forward(cur, cm, past_key_values=pkv)            # handle from an older engine state
    with pytest.raises(ValueError):
        m.forward(ids.to(DEV), mask.to(DEV), labels=ids[:, :2].to(DEV))        # labels must match input_ids
    with pytest.raises(RuntimeError):
        m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=1, topk=cfg.audio_vocab_size + 1)
    assert m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=0).shape == (2, 0, 32)


def test_tiny_stop_on_all_zeros():
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    sd["codebook0_head.weight"] = torch.zeros_like(sd["codebook0_head.weight"])
    sd["audio_head"] = torch.zeros_like(sd["audio_head"])
    m = make_model(cfg, sd, torch.float32)
    ids, mask = synth_context(cfg, 1, 2, 2, seed=5)
    out = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=5, topk=1, stop_on_all_zeros=True)
    assert out.shape == (1, 0, 32)                  # all-zero logits -> argmax 0 everywhere -> immediate stop
    out = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=5, topk=1, stop_on_all_zeros=False)
    assert out.shape == (1, 5, 32) and int(out.abs().sum()) == 0


def test_tiny_topk_sampling_matches_oracle_with_explicit_noise():
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    m = make_model(cfg, sd, torch.float32)
    ids, mask = synth_context(cfg, 2, 3, 4, seed=8)
    n, C, V = 3, 32, cfg.audio_vocab_size
    noise1 = torch.empty(2, C, V).exponential_(1, generator=torch.Generator().manual_seed(3))
    noise = noise1[None].repeat(n, 1, 1, 1)       # the engine indexes noise [b][cb][v] per frame
    want = O.generate(sd, cfg, ids, mask, max_new_frames=n, topk=10, temperature=0.8, stop_on_all_zeros=False, noise=noise)
    toks, _, _ = traced_generate(m, ids, mask, n, topk=10, temperature=0.8, noise=noise1)
    assert torch.equal(toks, want)
    # device RNG: reproducible for a fixed seed, different across seeds, tokens in range
    eng = m._engine
    outs = []
    for seed in (1, 1, 2):
        eng.reset()
        eng.prefill(ids, mask)
        eng.generate(eng.sampling(temperature=1.0, topk=50, seed=seed), n, True)
        outs.append(eng.read_frames(0, n).cpu())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    assert int(outs[2].min()) >= 0 and int(outs[2].max()) < V


def test_tiny_batch_rows_equal_solo_runs():
    """batch-sharding invariant: row b of a batch == that utterance alone."""
    cfg = CSMConfig.tiny()
    sd = {k: v.to(torch.bfloat16) for k, v in synth_state_dict(cfg, seed=0, std=0.05).items()}
    m = make_model(cfg, sd, torch.bfloat16)
    ids, mask = synth_context(cfg, 6, 4, 6, seed=31)
    full = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=3, topk=1, stop_on_all_zeros=False).cpu()
    for b in (0, 5):
        solo = m.generate(ids[b:b + 1].to(DEV), mask[b:b + 1].to(DEV), max_new_frames=3, topk=1, stop_on_all_zeros=False).cpu()
        assert torch.equal(solo[0], full[b])


# ---------------------------------------------------------------------------------------------------
# csm-1b: BASELINE configs 1 and 2 against the reference's golden vectors
# ---------------------------------------------------------------------------------------------------
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


def margin_check(toks, g, thresh=1e-4):
    """free-running tokens must equal the reference's up to (not including) the first sample whose
    reference top-1 margin is below `thresh`; returns the number of samples compared."""
    tv = g["top_vals"]
    margin = (tv[..., 0] - tv[..., 1])              # [n,B,C]
    n, B, C = margin.shape
    flat_m = margin.transpose(1, 0, 2).reshape(B, n * C)
    ref = g["tokens"].reshape(B, n * C)
    mine = toks.numpy().reshape(B, n * C)
    compared = 0
    for b in range(B):
        low = np.nonzero(flat_m[b] < thresh)[0]
        stop = int(low[0]) if len(low) else n * C
        assert np.array_equal(mine[b, :stop], ref[b, :stop]), f"row {b}: mismatch before the first low-margin sample {stop}"
        compared += stop
    return compared


def solo_margin_agree(model, ids_row, mask_row, batch_row_toks, n, thresh=1e-4):
    """A batch row against its SOLO run (single-sequence fp32-FMA kernels), by the same rule the reference-anchored tests
    use: the two greedy streams must be equal up to (not including) the first sample whose solo top-1 margin is below
    `thresh` -- two kernel families differ by fp32 summation order only, so nothing else may flip a token.  Returns
    (samples compared, samples in the stream)."""
    solo, lt, _ = traced_generate(model, ids_row, mask_row, n)
    tv = torch.topk(lt[:, 0], 2, -1)[0]                     # [n, C, 2]
    margin = (tv[..., 0] - tv[..., 1]).reshape(-1).numpy()
    low = np.nonzero(margin < thresh)[0]
    stop = int(low[0]) if len(low) else margin.size
    a, b = solo[0].reshape(-1).numpy(), batch_row_toks.reshape(-1).numpy()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a[:stop], b[:stop]), f"batch row differs from its solo run before the first low-margin sample {stop}"
    return stop, margin.size


def test_csm1b_config1_fp32_bit_exact(gold):
    """BASELINE config 1: csm-1b, 64-frame context, 8 greedy frames, fp32 -- bit-exact token ids."""
    cfg = CSMConfig()
    g = gold("csm1b_cfg1_fp32")
    sd = synth_state_dict(cfg, seed=0, dtype=torch.float32, device=DEV)
    m = CSMModel(cfg)
    m.load_state_dict(sd)
    del sd
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    ids2, _ = synth_context(cfg, 1, 16, 48, seed=1)
    assert torch.equal(ids, ids2)                   # the synthetic context is reproducible on this box
    toks, lt, ht = traced_generate(m, ids, mask, 8)
    m._drop_engine()
    assert float(g["min_margin"]) > 1e-4
    assert np.array_equal(toks.numpy(), g["tokens"])
    np.testing.assert_allclose(ht.numpy(), g["last_h"], atol=5e-4, rtol=0)
    tv = torch.topk(lt, 4, -1)[0].numpy()
    np.testing.assert_allclose(tv, g["top_vals"], atol=5e-4, rtol=0)


def test_csm1b_config1_bf16_weights(gold, csm1b_bf16):
    m = csm1b_bf16
    g = gold("csm1b_cfg1_bf16w_fp32")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    toks, lt, ht = traced_generate(m, ids, mask, 8)
    assert float(g["min_margin"]) > 1e-4
    assert np.array_equal(toks.numpy(), g["tokens"])          # bit-exact vs reference(fp32 arithmetic)
    assert rel_l2(ht, torch.from_numpy(g["last_h"])) < 1e-4
    # against the reference's own bf16 execution: tolerance + margin-aware, teacher-forced
    gb = gold("csm1b_cfg1_bf16")
    toks_f, lt_f, ht_f = traced_generate(m, ids, mask, 8, forced=torch.from_numpy(gb["tokens"]))
    assert rel_l2(ht_f, torch.from_numpy(gb["last_h"])) < 5e-2
    # frame 0 has identical inputs in all three runs: the engine is no farther from the bf16 reference
    # than the reference's own fp32-arithmetic run is
    d_ref = rel_l2(torch.from_numpy(g["last_h"][0]), torch.from_numpy(gb["last_h"][0]))
    assert rel_l2(ht_f[0], torch.from_numpy(gb["last_h"][0])) <= 1.02 * d_ref
    mine = lt_f.argmax(-1).numpy()                              # [n,B,C]
    top_idx, top_val = gb["top_idx"], gb["top_vals"]
    my_logit_top = np.take_along_axis(lt_f.numpy(), top_idx, -1)
    assert np.abs(my_logit_top - top_val).max() < 0.1          # bf16 logits: |err| <= 0.1 on |logit| <= 4
    in_top4 = (mine[..., None] == top_idx).any(-1)
    assert in_top4.mean() > 0.99


def test_csm1b_decode_precision_bf16_batched_vs_reference_bf16(gold, csm1b_bf16):
    """`decode_precision = "bf16"` (VERDICT r3 missing 2): batched decode with ONE nearest-even bf16 activation plane per
    hand-off -- the reference's own arithmetic class (README.md:73: the model runs in bf16, every nn.Linear of
    modeling_csm.py:156-167, 545-576 sees bf16 activations).  SURVEY 8-c protocol, teacher-forced with the tokens of the
    reference's OWN bf16 run (fixture csm1b_cfg1_bf16) on a batch of 3 equal rows (matrix-core kernels): last_h rel-L2 <= 5e-2,
    top logits within 0.1, arg-max inside the reference's top-4 for > 99 % of the samples; the rows of the batch agree
    bit for bit; and the mode sits between the exact engine and the bf16 reference: its distance to the EXACT engine
    (same tokens) is reported and must be a bf16-class distance (1e-4 .. 5e-2), i.e. the switch did something and not more
    than bf16 rounding."""
    m = csm1b_bf16
    gb = gold("csm1b_cfg1_bf16")
    g = gold("csm1b_cfg1_bf16w_fp32")
    ids1, mask1 = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    ids, mask = ids1.repeat(3, 1, 1), mask1.repeat(3, 1, 1)
    forced = torch.from_numpy(gb["tokens"]).repeat(3, 1, 1)
    try:
        _, lt_x, ht_x = traced_generate(m, ids, mask, 8, forced=forced)
        m.decode_precision = "bf16"
        _, lt_b, ht_b = traced_generate(m, ids, mask, 8, forced=forced)
    finally:
        m.decode_precision = "exact"
        m._drop_engine()
    assert torch.equal(lt_b[:, 0], lt_b[:, 1]) and torch.equal(lt_b[:, 0], lt_b[:, 2])     # equal rows stay equal
    ref_h = torch.from_numpy(gb["last_h"])                                             # [n, 1, H]
    d_ref = rel_l2(ht_b[:, :1], ref_h)
    d_exact = rel_l2(ht_b, ht_x)
    top_idx, top_val = gb["top_idx"], gb["top_vals"]                                   # [n, 1, C, 4]
    mine = lt_b[:, :1]
    err = np.abs(np.take_along_axis(mine.numpy(), top_idx, -1) - top_val).max()
    in_top4 = (mine.argmax(-1).numpy()[..., None] == top_idx).any(-1).mean()
    print(f"decode_precision=bf16, B = 3: last_h vs the reference's bf16 run {d_ref:.3e} (exact engine: {rel_l2(ht_x[:, :1], ref_h):.3e}), vs the exact "
          f"engine {d_exact:.3e}; top-logit |err| {err:.3f}; arg-max in the reference's top-4: {in_top4:.4f}")
    assert d_ref < 5e-2 and err < 0.1 and in_top4 > 0.99
    assert 1e-4 < d_exact < 5e-2, d_exact


def test_csm1b_prefill512_hidden_state(gold, csm1b_bf16):
    m = csm1b_bf16
    g = gold("csm1b_prefill512_bf16w_fp32")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    o = m.forward(ids.to(DEV), mask.to(DEV), use_cache=True)
    lh = o.last_hidden_state.float().cpu()
    # model dtype is bf16, so the API output is bf16-rounded; compare the engine's fp32 state too
    eng_lh, eng_lg = m._engine.get_state()
    assert rel_l2(eng_lh.cpu(), torch.from_numpy(g["last_h"][0])) < 1e-4
    assert rel_l2(lh, torch.from_numpy(g["last_h"][0])) < 1e-2
    tv = torch.topk(eng_lg.cpu(), 8, -1)
    np.testing.assert_allclose(tv[0].numpy(), g["top_vals"][0, :, 0], atol=1e-3, rtol=0)
    assert np.array_equal(tv[1].numpy()[:, 0], g["top_idx"][0, :, 0, 0])
    gb = gold("csm1b_prefill512_bf16")
    assert rel_l2(eng_lh.cpu(), torch.from_numpy(gb["last_h"][0])) < 5e-2


def test_csm1b_prefill_precision_bf16(gold, csm1b_bf16):
    """prefill_precision = "bf16" (VERDICT r1 item 6): the context GEMMs read activations rounded to bf16 -- what the
    reference's own bf16 execution does at every op -- on ONE bf16 MFMA pass instead of the exact three.  Stated
    tolerance (BASELINE north_star "backbone hidden states within a stated bf16 tolerance"): last_hidden_state rel-L2
    <= 5e-2 against the reference's bf16 run and <= 2.8e-2 (the reference's own bf16-vs-fp32 distance) against its
    fp32 run; the codebook-0 argmax lies in the fp32 reference's top-4.  The exact mode is untouched by the switch."""
    m = csm1b_bf16
    g = gold("csm1b_prefill512_bf16w_fp32")
    gb = gold("csm1b_prefill512_bf16")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    m.prefill_precision = "bf16"
    try:
        m.forward(ids.to(DEV), mask.to(DEV), use_cache=True)
        lh, lg = m._engine.get_state()
    finally:
        m.prefill_precision = "exact"
    lh, lg = lh.cpu(), lg.cpu()
    d32, d16 = rel_l2(lh, torch.from_numpy(g["last_h"][0])), rel_l2(lh, torch.from_numpy(gb["last_h"][0]))
    assert 1e-5 < d32 < 2.8e-2 and d16 < 5e-2, (d32, d16)
    assert int(lg.argmax(-1)[0]) in set(g["top_idx"][0, 0, 0, :4].tolist())
    m.forward(ids.to(DEV), mask.to(DEV), use_cache=True)
    assert rel_l2(m._engine.get_state()[0].cpu(), torch.from_numpy(g["last_h"][0])) < 1e-4
    # a generation after a bf16 prefill runs the exact decode kernels on the approximate context: tokens stay in range
    m.prefill_precision = "bf16"
    try:
        out = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=3, topk=1, stop_on_all_zeros=False)
    finally:
        m.prefill_precision = "exact"
    assert out.shape == (1, 3, 32) and int(out.min()) >= 0 and int(out.max()) < m.config.audio_vocab_size
    # the 128 x 256-tile GEMM (weight fragments straight from the fragment-order copy into the MFMA operand registers)
    # against the square-tile kernel on the same bf16 activations.  Without a K split the two are BITWISE equal (the
    # bf16 MFMA accumulates its products as one sequential fp32 chain in k order, whatever the instruction shape), which
    # pins every path of the kernel: 1024 frames route gate/up through it, 2000 (ragged last row block) down_proj too.
    # With the K split the wide launch prefers, the result moves by the distance two bf16-rounded trajectories have
    # (a last-bit change flips bf16 roundings downstream): bounded by the mode's stated tolerance against the exact mode.
    from csm_hf_amd.synth import synth_context
    for ctx in (1024, 2000):
        cids, cmask = synth_context(m.config, 1, ctx // 4, ctx - ctx // 4, seed=5)
        outs = {}
        for mode in ("exact", "bf16"):
            for wide, splitk in ((1, 0), (0, 0), (1, 1)) if mode == "bf16" else ((1, 1),):
                m.prefill_precision = mode
                try:
                    m.forward(cids.to(DEV), cmask.to(DEV), use_cache=True)          # engine exists from here on
                    m._engine.set_option("gemm_wide", wide)
                    m._engine.set_option("prefill_splitk", splitk)
                    m.forward(cids.to(DEV), cmask.to(DEV), use_cache=True)
                    outs[(mode, wide, splitk)] = m._engine.get_state()[0].cpu()    # fp32 hidden state
                finally:
                    m.prefill_precision = "exact"
                    m._engine.set_option("gemm_wide", 1)
                    m._engine.set_option("prefill_splitk", 1)
        assert torch.isfinite(outs[("bf16", 1, 0)]).all()
        assert torch.equal(outs[("bf16", 1, 0)], outs[("bf16", 0, 0)]), ctx
        assert rel_l2(outs[("bf16", 1, 1)], outs[("exact", 1, 1)]) < 2.8e-2, ctx


def test_csm1b_config2_200_frames(gold, csm1b_bf16):
    """BASELINE config 2 (the benchmarked workload): 512-frame context + 200 greedy frames, hipGraph
    replay, vs the reference's fp32-arithmetic run on the same weights."""
    m = csm1b_bf16
    g = gold("csm1b_cfg2_bf16w_fp32")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    n = g["tokens"].shape[1]
    toks, _, _ = traced_generate(m, ids, mask, n)
    compared = margin_check(toks, g, 1e-4)
    # the whole 6 400-token stream must match, not a prefix: the only licence to differ is AT a sample whose reference
    # top-1 margin is below 1e-4 (the stream has one at 1.5e-6; the engine agrees with the reference there too)
    margin = (g["top_vals"][..., 0] - g["top_vals"][..., 1])[:, 0].reshape(-1)
    low = np.nonzero(margin < 1e-4)[0]
    assert compared == (int(low[0]) if len(low) else 32 * n) and 32 * n == 6400
    mine, ref = toks.numpy().reshape(-1), g["tokens"].reshape(-1)
    diff = np.nonzero(mine != ref)[0]
    assert len(diff) == 0 or margin[diff[0]] < 1e-4, f"first mismatch at sample {diff[0]} with margin {margin[diff[0]]}"
    assert len(diff) == 0, "free-running stream left the reference at a near-tie (allowed by the margin rule, but new)"
    # teacher-forced over all 200 frames: argmax equals the reference wherever its margin > 1e-3
    toks_f, lt_f, _ = traced_generate(m, ids, mask, n, forced=torch.from_numpy(g["tokens"]))
    margin = g["top_vals"][..., 0] - g["top_vals"][..., 1]
    mine = lt_f.argmax(-1).numpy()
    ref = g["tokens"].transpose(1, 0, 2)
    safe = margin > 1e-3
    assert np.array_equal(mine[safe], ref[safe])
    assert (mine == ref).mean() > 0.999


# ---------------------------------------------------------------------------------------------------
# BASELINE config 3 / 4 / 5 shapes
# ---------------------------------------------------------------------------------------------------
def test_csm1b_batch16_rows_equal_solo_and_graph_equals_eager(csm1b_bf16):
    """config 4 per-GPU shape (16 utterances): the matrix-core batched kernels give every row the token stream
    of its solo (M = 1, fp32-FMA kernels) run up to the first sample whose solo margin is below 1e-4 (the rule of the
    reference-anchored tests); eager == hipGraph bit for bit."""
    m = csm1b_bf16
    cfg = m.config
    ids, mask = synth_context(cfg, 16, 16, 48, seed=41)
    full = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=6, topk=1, stop_on_all_zeros=False).cpu()
    m.use_graph = False
    eager = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=6, topk=1, stop_on_all_zeros=False).cpu()
    m.use_graph = True
    assert torch.equal(full, eager)
    compared = total = 0
    for b in (0, 7, 15):
        c, t = solo_margin_agree(m, ids[b:b + 1], mask[b:b + 1], full[b], 6)
        compared, total = compared + c, total + t
    assert compared >= total // 2, (compared, total)     # the rule is not vacuous: low-margin samples are rare


def test_csm1b_continuous_batching_on_the_matrix_core_path(csm1b_bf16):
    """csm_prefill_slot on csm-1b: a batch of 4 rows (matrix-core kernels on activation planes) serves 7 utterances; the
    ones that take over a row mid-batch give the token stream of their solo run wherever the margin allows -- the same
    rule as the batched-rows test above (two kernel families, fp32-class differences)."""
    from csm_hf_amd import ContinuousBatcher
    m = csm1b_bf16
    cfg = m.config
    reqs = []
    for i, (T, budget) in enumerate([(40, 4), (33, 9), (36, 3), (28, 6), (30, 5), (25, 4), (31, 3)]):
        ids, mask = synth_context(cfg, 1, T // 4, T - T // 4, seed=900 + i)
        reqs.append((ids[0], mask[0], budget))
    cb = ContinuousBatcher(m, batch_size=4, topk=1, check_every=3)
    rid = [cb.submit(a, b, max_new_frames=n) for a, b, n in reqs]
    out = cb.run()
    assert sorted(out) == rid and cb.joined_mid_batch >= 3
    compared = total = 0
    for r, (ids, mask, budget) in zip(rid, reqs):
        assert out[r].shape == (budget, cfg.audio_num_codebooks)
        c, t = solo_margin_agree(m, ids[None], mask[None], out[r], budget)
        compared, total = compared + c, total + t
    assert compared >= total // 2, (compared, total)


@pytest.mark.parametrize("B,opts", [(5, {}), (18, {}), (32, {}), (40, {}), (64, {}), (40, {"rows64": 0}), (70, {}), (16, {"use_planes": 0}), (16, {"tile_weights": 0})])
def test_csm1b_batched_rows_other_shapes_and_paths(csm1b_bf16, B, opts):
    """ragged batches (M < 16 on the matrix-core kernel; 18 and 32 rows = one launch of the 32-row kernel on planes,
    two 16-row groups where the input is fp32; 40 and 64 rows = ONE launch of the four-batch-tile form (round 3; `rows64 = 0`: 32 + 8);
    70 rows = 16-row groups without planes) and the A/B paths (no activation planes; row-major weights) against solo runs"""
    m = csm1b_bf16
    cfg = m.config
    ids, mask = synth_context(cfg, B, 12, 20, seed=43)
    try:
        m.setup_caches(B)
        eng = m._ensure_engine(B, 64, 8, B * 32)
        for k, v in opts.items():
            eng.set_option(k, v)
        full = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=4, topk=1, stop_on_all_zeros=False).cpu()
    finally:
        if m._engine is not None:
            for k in opts:
                m._engine.set_option(k, {"use_planes": 31, "tile_weights": 1, "rows64": 1}[k])
    compared = total = 0
    for b in sorted({0, min(17, B - 1), min(33, B - 1), B - 1}):
        c, t = solo_margin_agree(m, ids[b:b + 1], mask[b:b + 1], full[b], 4)
        compared, total = compared + c, total + t
    assert compared >= total // 2, (compared, total)


def test_csm1b_64_row_launch_is_bitwise_two_32_row_launches(csm1b_bf16):
    """17..128 batched rows in ONE matrix-core launch per linear (gemm32_kernel: several batch tiles per workgroup, further rows on
    blockIdx.z, shapes by launch kind -- gemm32.hip) against narrower launches: 32-row launches (`rows64 = 0`) and 16-row launches
    on gemm16_kernel (`rows64 = -1`).  Every accumulator sums its products in the same order in every form, so not only the
    generated frames but the LOGITS of every codebook and the backbone's last hidden state are equal bit for bit (round 4: the
    token comparison alone let a 1e-6 multiply-add contraction difference of one instantiation through) -- 24 rows (a partial second
    tile), 32, 48, 64, 100 (a partial seventh tile), 128.  Both arithmetic classes (exact planes / decode_precision bf16)."""
    m = csm1b_bf16
    cfg = m.config
    try:
        for B in (24, 32, 48, 64, 100, 128):
            ids, mask = synth_context(cfg, B, 12, 20, seed=47)
            for prec in (("exact", "bf16") if B in (24, 32, 64, 128) else ("exact",)):
                m.decode_precision = prec
                outs = []
                for r64 in ((1, -1) if B <= 32 else (1, 0, -1)):
                    T = ids.shape[1]
                    eng = m._ensure_engine(B, T + 3 + 1, 3, B * T)      # (traced_generate asks for the same engine)
                    eng.set_option("rows64", r64)
                    outs.append(traced_generate(m, ids, mask, 3))
                    assert m._engine is eng
                m._engine.set_option("rows64", 1)
                for o in outs[1:]:
                    assert torch.equal(outs[0][0], o[0]), (B, prec, "tokens")
                    assert torch.equal(outs[0][1], o[1]), (B, prec, "logits", float((outs[0][1] - o[1]).abs().max()))
                    assert torch.equal(outs[0][2], o[2]), (B, prec, "last_h")
    finally:
        m.decode_precision = "exact"
        if m._engine is not None:
            m._engine.set_option("rows64", 1)
        m._drop_engine()


def test_csm1b_config3_batch16_voiceclone_rows_vs_reference(gold, csm1b_bf16):
    """BASELINE config 3 shape at full size, against the REFERENCE (not the engine's own single-row kernels): B = 16,
    512-frame voice-clone layout (48 text + 400 audio + zero EOS frame + 63 text).  Rows 0-3 are the four distinct
    rows of the reference fixture `csm1b_b4_ctx512_bf16w_fp32`; rows 4-15 fill the batch.  The batched matrix-core
    kernels must give rows 0-3 the reference's greedy tokens (margin-aware, free running), its top-2 logits (5e-4) and
    its last_hidden_state (rel-L2 1e-4); the same rows must not depend on what else is in the batch."""
    m = csm1b_bf16
    cfg = m.config
    g = gold("csm1b_b4_ctx512_bf16w_fp32")
    ids, mask = synth_context(cfg, 16, 48, 400, seed=3, tail_text=63, eos_frame=True)
    assert ids.shape[1] == 512 and np.array_equal(ids[:4].numpy(), g["input_ids"]) and np.array_equal(mask[:4].numpy(), g["attention_mask"])
    n = g["tokens"].shape[1]
    toks, lt, ht = traced_generate(m, ids, mask, n)
    g4 = dict(g)
    compared = margin_check(toks[:4], g4, 1e-4)
    assert compared == 4 * n * 32, compared
    tv = torch.topk(lt[:, :4], 2, -1)[0].numpy()
    np.testing.assert_allclose(tv, g["top_vals"], atol=5e-4, rtol=0)
    assert rel_l2(ht[:, :4], torch.from_numpy(g["last_h"])) < 1e-4
    # B = 4 alone (16-row kernel with 4 live rows) gives the same four streams
    toks4, _, _ = traced_generate(m, ids[:4], mask[:4], n)
    assert torch.equal(toks4, toks[:4])


def test_csm1b_config3_topk50_sampler_in_situ_vs_reference(gold, csm1b_bf16):
    """The sampler inside the frame loop, all 32 codebooks, top-k = 50 / T = 1.0, against the REFERENCE run with the same
    explicit Exp(1) noise (fixture `csm1b_b4_topk50_noise_bf16w_fp32`, noise regenerated from its seed).  Teacher-forced
    with the reference's tokens so every one of the 2 x 4 x 32 draws is an independent comparison."""
    m = csm1b_bf16
    cfg = m.config
    g = gold("csm1b_b4_topk50_noise_bf16w_fp32")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    ref = torch.from_numpy(g["tokens"])                     # [4, n, 32]
    n, C, V = ref.shape[1], cfg.audio_num_codebooks, cfg.audio_vocab_size
    noise = torch.empty(n, 4, C, V).exponential_(1, generator=torch.Generator().manual_seed(int(g["noise_seed"])))
    eng = m._ensure_engine(4, 512 + n + 1, max(n, 1), 4 * 512)
    eng.reset()
    eng.set_kv_start([0] * 4)
    eng.prefill(ids, mask)
    fz = torch.zeros(4, eng.max_frames, C, dtype=torch.int64, device=DEV)
    fz[:, :n] = ref.to(DEV)
    for f in range(n):            # the engine takes one [B, C, V] noise block per call: one frame per call
        nz = noise[f].to(DEV).contiguous()
        eng.generate(eng.sampling(temperature=1.0, topk=50, noise=nz, forced=fz), 1, True)
        eng.sync()
    got = eng.read_frames(0, n).cpu()
    same = (got == ref)
    assert float(same.float().mean()) >= 0.99, f"{int((~same).sum())} of {same.numel()} draws differ"
    assert bool(same[:, 0].all()), "frame 0 (no accumulated difference) must match exactly"


def test_csm1b_config5_fp8_long_context_500_frames(csm1b_bf16):
    """BASELINE config 5 at full size: e4m3fn linear weights, 2048-frame prefill, 500 generated frames (positions to
    2547 > max_seq_len).  (i) the oracle (fp32 arithmetic, CPU) on the DEQUANTISED checkpoint: last_hidden_state and
    codebook-0 logits after the 2048-frame prefill, rel-L2 <= 1e-4 / |dlogit| <= 2e-3, same argmax; (ii) over all 500
    frames, teacher-forced, the engine's own fp32-weight path on the dequantised checkpoint (validated against the
    reference elsewhere in this file): argmax equal wherever that run's top-1 margin > 1e-3, hidden rel-L2 <= 2e-4."""
    cfg = csm1b_bf16.config
    sd = {k: v.detach() for k, v in csm1b_bf16.state_dict().items()}
    ids, mask = synth_context(cfg, 1, 256, 1792, seed=5)
    n = 500
    csm1b_bf16.weight_format = "fp8"
    try:
        csm1b_bf16.setup_caches(1, max_seq_len=2048 + n + 8, max_frames=n + 8)
        toks8, lt8, ht8 = traced_generate(csm1b_bf16, ids, mask, n)
        assert csm1b_bf16._engine.fp8 and csm1b_bf16._engine.device_counters() == (2048 + n, n)
    finally:
        csm1b_bf16.weight_format = "native"
        csm1b_bf16._drop_engine()
    sdq = _fp8_roundtrip_state_dict(cfg, sd)
    # (i) oracle prefill on the host
    with torch.inference_mode():
        lh, lg, _ = O.forward({k: v.cpu() for k, v in sdq.items()}, cfg, ids, mask)
    assert rel_l2(ht8[0], lh) < 1e-4
    assert float((lt8[0, :, 0] - lg).abs().max()) < 2e-3 and torch.equal(lt8[0, :, 0].argmax(-1), lg.argmax(-1))
    # (ii) fp32-weight engine on the dequantised checkpoint, teacher-forced with the fp8 run's tokens
    ref = CSMModel(cfg)
    ref.load_state_dict({k: v.to(DEV) for k, v in sdq.items()})
    del sdq
    ref.setup_caches(1, max_seq_len=2048 + n + 8, max_frames=n + 8)
    _, lt_r, ht_r = traced_generate(ref, ids, mask, n, forced=toks8)
    ref._drop_engine()
    assert rel_l2(ht8, ht_r) < 2e-4
    tv = torch.topk(lt_r, 2, -1)[0]
    safe = (tv[..., 0] - tv[..., 1]) > 1e-3
    a8, ar = lt8.argmax(-1), lt_r.argmax(-1)
    assert torch.equal(a8[safe], ar[safe]) and float((a8 == ar).float().mean()) > 0.998
    assert torch.equal(a8.permute(1, 0, 2), toks8)          # the recorded tokens are the argmax of the traced logits


def test_csm1b_config5_mxfp8_prefill_pinned_at_full_size(csm1b_bf16):
    """BASELINE configs[4] as it is benchmarked (`bench.py --weights fp8`): fp8 weights, the 2048-frame context prefilled with
    `prefill_precision = "mxfp8"` (backbone linears on v_mfma_scale_f32_16x16x128_f8f6f4, MX-fp8 weights AND activations), then
    500 generated frames.  Round 3 tied this arithmetic to the oracle only on the tiny model.

    What CAN be pinned at this size, measured first (tools/mx_pin_probe.py, profiles/r04_mx_pin_probe.txt): the 16-layer
    random-weight network with e4m3 activations is CHAOTIC -- the oracle with the OCP-MX rounding wrapped around its backbone
    linears (oracle/mx_sim.py) moves by 0.13-0.21 rel-L2 on last_hidden_state when 2e-7 RELATIVE noise (below one fp32 ulp)
    is put in front of every quantiser, 0.21-0.23 with 2e-5 (the matrix instruction's accumulation class): an element that
    crosses an e4m3 rounding step moves by 6 %, and the random network amplifies it to the size of the format's own distance
    from fp32 (0.32).  No implementation can sit closer to that oracle than the oracle sits to itself.  So:
    (i) per operation, at the config's own shapes (2048 rows x the four backbone linears): quantiser bytes + scales bit-exact
        vs mx_sim, GEMM within 5e-5 of sum |a||b| of the fp64 product of the dequantised operands (both tiles) -- this is the
        pin of the arithmetic;
    (ii) end to end: engine (exact attention) vs oracle+MX  <=  1.25 x the oracle's own distance to itself under 2e-5 noise
        (measured in this test) and <= 0.30 absolute (stated budget); the mode as shipped (bf16-pipe attention) <= 0.30;
    (iii) accuracy class: distance to the bf16-activation mode and to the exact mode on the same weights in 0.15 .. 0.45;
    (iv) the 500 frames generated from the MX context, teacher-forced through the exact-prefill engine: arg-max agreement
        where that run's top-1 margin exceeds 1.0 >= 90 % (measured 100 %) and >= 80 % over all 16 000 samples (measured 88 %:
        random-weight logits have small margins), and the recorded tokens are the arg-max of the traced logits (the
        decode path itself is the exact fp8-weight path validated in the test above)."""
    from oracle import mx_sim as MX
    cfg = csm1b_bf16.config
    sd = {k: v.detach() for k, v in csm1b_bf16.state_dict().items()}
    ids, mask = synth_context(cfg, 1, 256, 1792, seed=5)
    n = 500
    m = csm1b_bf16
    m.weight_format = "fp8"
    try:
        m.setup_caches(1, max_seq_len=2048 + n + 8, max_frames=n + 8)
        m.prefill_precision = "mxfp8"
        toks_mx, lt_mx, ht_mx = traced_generate(m, ids, mask, n)                      # (ii) + (iv): the mode as shipped
        eng = m._engine
        # (i) per operation at 2048 rows
        g = torch.Generator().manual_seed(11)
        lc = cfg.backbone_config
        H, F_, A_ = lc.hidden_size, lc.intermediate_size, (lc.num_attention_heads + 2 * lc.num_key_value_heads) * (lc.hidden_size // lc.num_attention_heads)
        for N, K in ((A_, H), (H, H), (2 * F_, H), (H, F_)):
            X = torch.randn(2048, K, generator=g) * torch.exp2(torch.randint(-4, 5, (2048, 1), generator=g).float())
            Wm = torch.randn(N, K, generator=g) * 0.02
            q, s_ = eng.k_mx_quantize(X)
            wq_, ws_ = MX.mx_quantize(X)
            assert torch.equal(q.cpu(), wq_) and torch.equal(s_.cpu(), ws_), (N, K)
            wq, ws = MX.mx_quantize(Wm)
            Ad, Wd = MX.mx_dequantize(wq_, ws_).to(DEV).double(), MX.mx_dequantize(wq, ws).to(DEV).double()
            want = Ad @ Wd.T
            bound = (Ad.abs() @ Wd.abs().T) * 5e-5 + 1e-30
            for big in (0, 256):                      # 128 x 128 tile, then the 256 x 256 tile where it covers the launch
                eng.set_option("gemm_256", big)
                got = eng.k_gemm_mx(wq, ws, wq_, ws_).double()
                assert bool(((got - want).abs() <= bound).all()), (N, K, big, float(((got - want).abs() / bound).max()))
            eng.set_option("gemm_256", 256)
            del Ad, Wd, want, bound, got
        eng.set_option("prefill_bf16_attn", 0)
        _, lt_strict, ht_strict = traced_generate(m, ids, mask, 1)                    # (ii): exact attention
        m._engine.set_option("prefill_bf16_attn", 1)
        m.prefill_precision = "bf16"
        _, _, ht_b = traced_generate(m, ids, mask, 1)
        m.prefill_precision = "exact"
        _, lt_x, ht_x = traced_generate(m, ids, mask, n, forced=toks_mx)              # exact context, teacher-forced
    finally:
        m.prefill_precision = "exact"
        m.weight_format = "native"
        m._drop_engine()
    # the oracle's checkpoint: heads / projection hold the fp8 values the engine multiplies with, the backbone linears the
    # checkpoint's own bf16 values (the MX copies are quantised from those, Engine.enable_mx)
    sdq = _fp8_roundtrip_state_dict(cfg, sd)
    for k, v in sd.items():
        if k.startswith("backbone.layers.") and k.endswith("_proj.weight"):
            sdq[k] = v.float()
    sdc = {k: v.cpu() for k, v in sdq.items()}
    lin = [v for k, v in sdc.items() if k.startswith("backbone.layers.") and k.endswith("_proj.weight")]

    class noisy(MX.mx_linears):          # the same simulation with 2e-5 relative noise in front of every activation quantiser
        def __enter__(self):
            self.orig = self.O.F.linear
            gen = torch.Generator().manual_seed(3)

            def lin_(x, w, b=None):
                if w.data_ptr() not in self.keys:
                    return self.orig(x, w, b)
                if w.data_ptr() not in self.cache:
                    self.cache[w.data_ptr()] = MX.mx_round(w)
                return self.orig(MX.mx_round(x * (1.0 + 2e-5 * torch.randn(x.shape, generator=gen))), self.cache[w.data_ptr()], b)
            self.O.F.linear = lin_
            return self

    with torch.inference_mode():
        with MX.mx_linears(O, lin):
            lh, lg, _ = O.forward(sdc, cfg, ids, mask)
        with noisy(O, lin):
            lh_n, _, _ = O.forward(sdc, cfg, ids, mask)
    self_dist = rel_l2(lh_n, lh)
    d_strict, d_ship = rel_l2(ht_strict[0], lh), rel_l2(ht_mx[0], lh)
    d_bf16, d_exact = rel_l2(ht_mx[0], ht_b[0]), rel_l2(ht_mx[0], ht_x[0])
    am = int(lt_strict[0, 0, 0].argmax(-1))
    rank = int((lg[0] > lg[0, am]).sum())
    tv = torch.topk(lt_x, 2, -1)[0]
    margin = tv[..., 0] - tv[..., 1]
    a_mx, a_x = lt_mx.argmax(-1), lt_x.argmax(-1)
    safe = margin > 1.0
    agree_safe = float((a_mx[safe] == a_x[safe]).float().mean()) if bool(safe.any()) else 1.0
    agree_all = float((a_mx == a_x).float().mean())
    by_margin = {t: (round(float((a_mx[margin > t] == a_x[margin > t]).float().mean()), 4), int((margin > t).sum())) for t in (0.1, 0.25, 0.5)}
    print("    arg-max agreement (fraction, samples) by exact-run margin:", by_margin)
    print(f"config 5 mxfp8 @ 2048 frames: last_h vs oracle+MX strict {d_strict:.3e}, shipped {d_ship:.3e}; oracle+MX vs itself under 2e-5 "
          f"noise {self_dist:.3e}; vs bf16 mode {d_bf16:.3e}, vs exact {d_exact:.3e}; the engine's c0 arg-max is rank {rank} of the oracle+MX "
          f"logits; 500 frames arg-max agreement with the exact context: {agree_all:.4f} overall, {agree_safe:.4f} where margin > 1.0 "
          f"({int(safe.sum())} of {safe.numel()} samples)")
    assert d_strict <= 1.25 * self_dist and d_strict < 0.30, (d_strict, self_dist)
    assert d_ship < 0.30, d_ship
    assert 0.15 < d_bf16 < 0.45 and 0.15 < d_exact < 0.45, (d_bf16, d_exact)
    assert agree_safe >= 0.90 and agree_all >= 0.80, (agree_safe, agree_all)     # measured 1.000 / 0.881
    assert torch.equal(a_mx.permute(1, 0, 2), toks_mx)


def test_csm1b_batch16_topk50_sampling_distribution(csm1b_bf16):
    """config 3 (B=16, topk=50, T=1.0, device Philox): codebook-0 samples follow softmax(top-50 logits).
    All 16 rows share one context, so the 16 x 8 seeds = 128 draws come from the same distribution."""
    m = csm1b_bf16
    cfg = m.config
    ids1, mask1 = synth_context(cfg, 1, 16, 48, seed=43)
    ids, mask = ids1.repeat(16, 1, 1), mask1.repeat(16, 1, 1)
    eng = m._ensure_engine(16, 64 + 4, 4, 16 * 64)
    draws = []
    for seed in range(8):
        eng.reset()
        eng.set_kv_start([0] * 16)
        lh, lg = eng.prefill(ids, mask)
        eng.generate(eng.sampling(temperature=1.0, topk=50, seed=1000 + seed), 1, True)
        draws.append(eng.read_frames(0, 1)[:, 0, 0].cpu())
    draws = torch.cat(draws)
    logits = lg[0].cpu()
    assert float((lg.cpu() - logits).abs().max()) < 1e-4            # identical rows -> identical logits
    top = torch.topk(logits, 50)
    p = torch.softmax(top[0], 0)
    assert bool(torch.isin(draws, top[1]).all())                      # never outside the top-50 set
    # chi-square against the expected multinomial (bins: top-8 tokens + rest)
    order = top[1][:8]
    obs = torch.tensor([float((draws == t).sum()) for t in order] + [float((~torch.isin(draws, order)).sum())])
    exp = torch.cat([p[:8], (1 - p[:8].sum())[None]]) * len(draws)
    chi2 = float(((obs - exp) ** 2 / exp.clamp_min(1e-9)).sum())
    assert chi2 < 27.9, (chi2, obs, exp)                               # chi^2_{8}, p = 0.0005


def test_csm1b_long_context_beyond_max_seq_len(csm1b_bf16):
    """config 5 shape (bf16 weights): a 2048-frame prefill followed by generation past max_seq_len = 2048.
    Positions are not clamped; split-KV attention with 64 splits == unsplit."""
    m = csm1b_bf16
    cfg = m.config
    ids, mask = synth_context(cfg, 1, 256, 1792, seed=5)
    m.setup_caches(1, max_seq_len=2048 + 24)
    out = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=12, topk=1, stop_on_all_zeros=False).cpu()
    assert out.shape == (1, 12, 32) and m._engine.device_counters() == (2048 + 12, 12)
    m._engine.set_option("nsplit_backbone", 1)
    ref = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=12, topk=1, stop_on_all_zeros=False).cpu()
    m._engine.set_option("nsplit_backbone", 64)
    out64 = m.generate(ids.to(DEV), mask.to(DEV), max_new_frames=12, topk=1, stop_on_all_zeros=False).cpu()
    m._engine.set_option("nsplit_backbone", 0)
    same = (out == ref).reshape(-1)
    assert bool(same[:64].all()) and float(same.float().mean()) > 0.8
    assert bool((out64 == ref).reshape(-1)[:64].all())
    # the long prefill agrees with a chunked prefill (4 x 512) of the same context
    eng = m._engine
    eng.reset()
    eng.set_kv_start([0])
    lh_a, lg_a = eng.prefill(ids, mask)
    eng.reset()
    eng.set_kv_start([0])
    for c in range(4):
        lh_b, lg_b = eng.prefill(ids[:, c * 512:(c + 1) * 512], mask[:, c * 512:(c + 1) * 512])
    assert rel_l2(lh_b.cpu(), lh_a.cpu()) < 1e-5


def _fp8_roundtrip_state_dict(cfg, sd):
    """checkpoint whose linear matrices hold exactly the values the fp8 engine computes with (q * s)."""
    from csm_hf_amd.engine import quantize_fp8_rows, dequantize_fp8_rows
    out = {}
    for k, v in sd.items():
        if k.endswith("proj.weight") or k in ("projection.weight", "codebook0_head.weight"):
            out[k] = dequantize_fp8_rows(*quantize_fp8_rows(v.float()))
        elif k == "audio_head":
            t = v.float().transpose(1, 2).contiguous()
            d = dequantize_fp8_rows(*quantize_fp8_rows(t.reshape(-1, t.shape[-1]))).view(t.shape)
            out[k] = d.transpose(1, 2).contiguous()
        else:
            out[k] = v.float()
    return out


def test_tiny_fp8_weights_vs_oracle():
    """fp8 engine == oracle run in fp32 on the dequantised checkpoint (tokens bit-exact, logits 2e-4)."""
    cfg = CSMConfig.tiny()
    sd = {k: v.to(torch.bfloat16) for k, v in synth_state_dict(cfg, seed=0, std=0.05).items()}
    sdq = _fp8_roundtrip_state_dict(cfg, sd)
    m = make_model(cfg, sd, torch.bfloat16)
    m.kv_dtype = EXACT_KV      # this test asserts the exact mode (fp32 KV cache) against fp32-arithmetic values
    m.weight_format = "fp8"
    ids, mask = synth_context(cfg, 2, 4, 6, seed=1)
    tr = {}
    want = O.generate(sdq, cfg, ids, mask, max_new_frames=4, topk=1, stop_on_all_zeros=False, trace=tr)
    toks, lt, ht = traced_generate(m, ids, mask, 4)
    tv = torch.topk(tr["logits"], 2, -1)[0]
    assert float((tv[..., 0] - tv[..., 1]).min()) > 1e-4
    assert torch.equal(toks, want)
    np.testing.assert_allclose(lt.numpy(), tr["logits"].numpy(), atol=2e-4, rtol=0)


def test_csm1b_fp8_weights(csm1b_bf16):
    """BASELINE config 5 weights: csm-1b with e4m3fn linears.  (i) against the SAME engine's validated fp32 path
    on the dequantised checkpoint: tokens equal (margin-aware), hidden rel-L2 <= 1e-4; (ii) against the bf16
    engine: stated tolerance last_h rel-L2 <= 0.3 -- per-row e4m3 noise (~3 % per weight) through 16 layers of
    RANDOM synthetic weights measures 0.21; this bounds gross errors only, (i) is the parity statement."""
    cfg = csm1b_bf16.config
    sd = csm1b_bf16.state_dict()
    ids, mask = synth_context(cfg, 1, 16, 48, seed=1)
    toks_bf16, _, ht_bf16 = traced_generate(csm1b_bf16, ids, mask, 4)
    csm1b_bf16.weight_format = "fp8"
    try:
        toks8, lt8, ht8 = traced_generate(csm1b_bf16, ids, mask, 4)
        assert csm1b_bf16._engine.fp8
    finally:
        csm1b_bf16.weight_format = "native"
        csm1b_bf16._drop_engine()
    assert rel_l2(ht8[0], ht_bf16[0]) < 0.3
    sdq = {k: v.to(DEV) for k, v in _fp8_roundtrip_state_dict(cfg, {k: v.detach() for k, v in sd.items()}).items()}
    ref = CSMModel(cfg)
    ref.load_state_dict(sdq)
    del sdq
    toks_r, lt_r, ht_r = traced_generate(ref, ids, mask, 4)
    ref._drop_engine()
    assert rel_l2(ht8, ht_r) < 1e-4
    tv = torch.topk(lt_r, 2, -1)[0]
    margin = (tv[..., 0] - tv[..., 1]).permute(1, 0, 2).reshape(-1)
    low = (margin < 1e-4).nonzero()
    stop = int(low[0]) if len(low) else margin.numel()
    assert stop >= 32 and torch.equal(toks8.reshape(-1)[:stop], toks_r.reshape(-1)[:stop])


def test_csm1b_fused_greedy_sampling_equals_sampler_kernel(gold, csm1b_bf16):
    """B=1 greedy fast path (argmax folded into the head launch + next QKV prologue) == the sampler-kernel path,
    free-running against the reference golden stream and under teacher forcing."""
    m = csm1b_bf16
    g = gold("csm1b_cfg1_bf16w_fp32")
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    eng = m._ensure_engine(1, 64 + 9, 8, 64)
    outs = {}
    for fuse in (1, 0):
        eng.set_option("fuse_sample", fuse)
        for forced in (None, torch.from_numpy(gold("csm1b_cfg1_bf16")["tokens"])):
            eng.reset()
            eng.set_kv_start([0])
            eng.prefill(ids, mask)
            fz = None
            if forced is not None:
                fz = torch.zeros(1, eng.max_frames, 32, dtype=torch.int64, device=DEV)
                fz[:, :8] = forced.to(DEV)
            eng.generate(eng.sampling(temperature=1.0, topk=1, forced=fz), 8, True)
            outs[(fuse, forced is not None)] = eng.read_frames(0, 8).cpu()
    eng.set_option("fuse_sample", 1)
    assert np.array_equal(outs[(1, False)].numpy(), g["tokens"])
    assert torch.equal(outs[(1, False)], outs[(0, False)])
    assert torch.equal(outs[(1, True)], outs[(0, True)])
