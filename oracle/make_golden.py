"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and `transformers`); the reference never
travels to the GPU box -- only the small .npz fixtures written here do.  Usage:

    python oracle/make_golden.py [--only tiny,sampler,processor,cfg1,prefill512,cfg2,b4,b4noise,loss,grad,grad1b,rng,rng1b] [--frames2 200]

What it does
  * imports /root/reference/modeling_csm.py unmodified;
  * installs the decode-mask shim of SURVEY.md Appendix B-1 (transformers 5.15 right-pads the `[B,1]`
    decode mask with zeros; the pinned 4.49 drops an all-ones mask) and SELF-CHECKS it: cached
    `generate` must equal the `use_cache=False` growing-context recompute bit-for-bit;
  * loads the deterministic synthetic checkpoint of `csm_hf_amd.synth` into the reference model with
    `load_state_dict(strict=True)` (this also fills `audio_head`, which the reference leaves
    uninitialised, modeling_csm.py:236-240);
  * records, by wrapping `modeling_csm.sample_topk` / `CSMModel.generate_frame`, every sampled
    logits row and every frame's `last_hidden_state`;
  * runs the oracle (oracle/csm_oracle.py) on the same inputs and stores whether it agreed
    (`oracle_tokens_equal`, `oracle_max_abs_logit_diff`) next to the vectors.

Greedy = (topk=1, temperature=1.0): the reference's temperature=0 divides by zero
(modeling_csm.py:181; SURVEY.md §0 finding 2).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import transformers.masking_utils as mu  # noqa: E402
import modeling_csm as REF  # noqa: E402  (the reference, unmodified)
from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding  # noqa: E402

from csm_hf_amd import CSMConfig  # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context  # noqa: E402
from oracle import csm_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _pad_left_valid(attention_mask, kv_length, kv_offset=0, **_):
    if attention_mask is not None and (n := kv_length + kv_offset - attention_mask.shape[-1]) > 0:
        return F.pad(attention_mask, (n, 0), value=True)
    return attention_mask


mu.prepare_padding_mask = _pad_left_valid


def ref_config(cfg: CSMConfig):
    from transformers import LlamaConfig

    def lc(c):
        d = c.to_dict()
        d.pop("model_type", None)
        return LlamaConfig(**d)
    return REF.CSMConfig(text_vocab_size=cfg.text_vocab_size, audio_vocab_size=cfg.audio_vocab_size,
                         audio_num_codebooks=cfg.audio_num_codebooks, max_seq_len=cfg.max_seq_len,
                         backbone_config=lc(cfg.backbone_config), decoder_config=lc(cfg.decoder_config))


def build_ref(cfg: CSMConfig, sd, dtype):
    rc = ref_config(cfg)
    with torch.device("meta"):
        model = REF.CSMModel(rc)
    model.load_state_dict({k: v.to(dtype) for k, v in sd.items()}, strict=True, assign=True)
    # non-persistent rotary buffers were created on meta: rebuild them on CPU
    model.backbone.rotary_emb = LlamaRotaryEmbedding(config=model.backbone.config)
    model.decoder.rotary_emb = LlamaRotaryEmbedding(config=model.decoder.config)
    model.eval()
    return model


class Recorder:
    """Wraps the reference's sampler + generate_frame to capture logits / hidden states."""

    def __init__(self, model, noise=None):
        self.model = model
        self.logits = []
        self.last_h = []
        self.noise = noise          # flat list of [B,V] tensors consumed in call order
        self._n = 0

    def __enter__(self):
        self._orig_sample = REF.sample_topk
        self._orig_multi = REF._multinomial_sample_one_no_sync
        self._orig_gf = self.model.generate_frame
        rec = self

        def sample(logits, topk, temperature):
            rec.logits.append(logits.detach().float().clone())
            return rec._orig_sample(logits, topk, temperature)

        def multi(probs):
            if rec.noise is None:
                return rec._orig_multi(probs)
            q = rec.noise[rec._n].to(probs.dtype)
            rec._n += 1
            return torch.argmax(probs / q, dim=-1, keepdim=True).to(dtype=torch.int)

        def gf(*a, **k):
            out = rec._orig_gf(*a, **k)
            rec.last_h.append(out.last_hidden_state.detach().float().clone())
            return out

        REF.sample_topk = sample
        REF._multinomial_sample_one_no_sync = multi
        self.model.generate_frame = gf
        return self

    def __exit__(self, *exc):
        REF.sample_topk = self._orig_sample
        REF._multinomial_sample_one_no_sync = self._orig_multi
        self.model.generate_frame = self._orig_gf


def topn(logits, n):
    v, i = torch.topk(logits, n, dim=-1)
    return v.numpy().astype(np.float32), i.numpy().astype(np.int32)


@torch.inference_mode()
def run_case(name, cfg, sd, dtype, ids, mask, frames, *, full_logits=False, topn_keep=4, extra=None,
             check_nocache=False, topk=1, temperature=1.0, noise=None):
    """`noise` [frames, B, C, V]: explicit Exp(1) draws substituted for the reference's `exponential_` call
    (modeling_csm.py:175) in sampling order (frame-major, codebook-minor), so a top-k run is reproducible bit for bit."""
    t0 = time.time()
    model = build_ref(cfg, sd, dtype)
    C, V = cfg.audio_num_codebooks, cfg.audio_vocab_size
    torch.manual_seed(1234)   # bf16 top-1 ties are broken by the global RNG (modeling_csm.py:175)
    flat = None if noise is None else [noise[f][:, c, :] for f in range(frames) for c in range(C)]
    with Recorder(model, noise=flat) as rec:
        toks = model.generate(ids, mask, max_new_frames=frames, temperature=temperature, topk=topk,
                              use_cache=True, stop_on_all_zeros=False)
    n = toks.shape[1]
    B = ids.shape[0]
    logits = torch.stack(rec.logits).view(n, C, B, V).permute(0, 2, 1, 3).contiguous()   # [n,B,C,V]
    last_h = torch.stack(rec.last_h)                                                      # [n,B,H]
    if check_nocache:
        # growing-context recompute with NO cache (reference `generate(use_cache=False)` would feed
        # only the last frame, modeling_csm.py:689-690, so drive generate_frame by hand)
        fi, fm, nc = ids, mask, []
        for _ in range(frames):
            o = model.generate_frame(fi, fm, temperature=1.0, topk=1, use_cache=False, return_dict=True)
            nc.append(o.samples)
            row = torch.cat([o.samples, torch.zeros(B, 1, dtype=torch.long)], 1).unsqueeze(1)
            m1 = torch.zeros(B, 1, C + 1, dtype=mask.dtype)
            m1[:, :, :C] = 1
            fi, fm = torch.cat([fi, row], 1), torch.cat([fm, m1], 1)
        assert torch.equal(toks, torch.stack(nc, 1)), "mask shim self-check failed: cached != no-cache"
    t_ref = time.time() - t0
    del model
    # --- run the oracle on the same inputs -------------------------------------------------------
    t0 = time.time()
    sdd = {k: v.to(dtype) for k, v in sd.items()}
    tr = {}
    torch.manual_seed(1234)
    otoks = O.generate(sdd, cfg, ids, mask, max_new_frames=frames, temperature=temperature, topk=topk,
                       stop_on_all_zeros=False, trace=tr, noise=noise)
    t_or = time.time() - t0
    eq = bool(torch.equal(otoks, toks))
    dl = float((tr["logits"] - logits).abs().max())
    dh = float((tr["last_h"] - last_h).abs().max())
    tv, ti = topn(logits, topn_keep)
    margin = tv[..., 0] - tv[..., 1]
    out = dict(input_ids=ids.numpy(), attention_mask=mask.numpy(), tokens=toks.numpy(),
               last_h=last_h.numpy(), top_vals=tv, top_idx=ti,
               min_margin=np.float32(margin.min()), n_exact_ties=np.int32((margin == 0).sum()),
               oracle_tokens_equal=np.int32(eq), oracle_max_abs_logit_diff=np.float32(dl),
               oracle_max_abs_lasth_diff=np.float32(dh), ref_seconds=np.float32(t_ref),
               oracle_seconds=np.float32(t_or), torch_threads=np.int32(torch.get_num_threads()),
               dtype=str(dtype))
    if full_logits:
        out["logits"] = logits.numpy()
    if extra:
        out.update(extra)
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"[golden] {name}: frames={n} min_margin={margin.min():.3e} ties={(margin == 0).sum()} "
          f"oracle_eq={eq} dlogit={dl:.2e} dh={dh:.2e} ref={t_ref:.1f}s oracle={t_or:.1f}s", flush=True)
    return toks


def gen_tiny():
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    ids, mask = synth_context(cfg, 2, 4, 6, seed=1)
    run_case("tiny_fp32", cfg, sd, torch.float32, ids, mask, 4, full_logits=True, check_nocache=True)
    run_case("tiny_bf16", cfg, sd, torch.bfloat16, ids, mask, 4, full_logits=True)
    # prefill hidden states per layer (kernel-level pins)
    with torch.inference_mode():
        model = build_ref(cfg, sd, torch.float32)
        out = model.backbone(inputs_embeds=model._embed_tokens(ids).mul(mask.unsqueeze(-1)).sum(2),
                             output_hidden_states=True, use_cache=False, return_dict=True)
        hs = torch.stack(out.hidden_states).numpy()   # [layers+1, B, S, H]; last entry is post-norm
        np.savez_compressed(os.path.join(GOLD, "tiny_fp32_hidden.npz"), hidden_states=hs,
                            last_hidden_state=out.last_hidden_state.numpy())
        # padded batch: expected = per-row SOLO reference runs (the oracle/product padding semantics)
        ids_b, mask_b = synth_context(cfg, 2, 4, 6, seed=3)
        pad = 3
        solo = []
        for b, cut in ((0, 0), (1, pad)):
            solo.append(model.generate(ids_b[b:b + 1, cut:], mask_b[b:b + 1, cut:], max_new_frames=4,
                                       temperature=1.0, topk=1, stop_on_all_zeros=False))
        ids_p, mask_p = ids_b.clone(), mask_b.clone()
        ids_p[1, :pad] = 0
        mask_p[1, :pad] = 0
        np.savez_compressed(os.path.join(GOLD, "tiny_padded.npz"), input_ids=ids_p.numpy(),
                            attention_mask=mask_p.numpy(), tokens=torch.cat(solo, 0).numpy(),
                            pad=np.int32(pad))
        print("[golden] tiny_padded, tiny_fp32_hidden written", flush=True)


def gen_rng():
    """Top-k sampling from torch's GLOBAL generator, exactly as the unmodified reference draws (modeling_csm.py:170-176):
    `run_case` seeds it (torch.manual_seed(1234)) right before `generate`; the fixture pins the sampled frames."""
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    ids, mask = synth_context(cfg, 2, 4, 6, seed=5)
    run_case("tiny_rng_topk5", cfg, sd, torch.float32, ids, mask, 6, full_logits=True, topk=5, temperature=0.9,
             extra=dict(torch_seed=np.int64(1234), topk=np.int32(5), temperature=np.float32(0.9)))


def gen_rng_1b():
    cfg = CSMConfig()
    sdb = synth_state_dict(cfg, seed=0, bf16_representable=True)
    ids, mask = synth_context(cfg, 1, 16, 48, seed=1)
    run_case("csm1b_rng_topk50_bf16w_fp32", cfg, sdb, torch.float32, ids, mask, 3, topn_keep=2, topk=50, temperature=0.9,
             extra=dict(torch_seed=np.int64(1234), topk=np.int32(50), temperature=np.float32(0.9)))


def gen_sampler():
    """K12 semantics: threshold keeps ties at the k-th value, double normalisation, exponential race."""
    g = torch.Generator().manual_seed(7)
    V = 2051
    logits = torch.randn(16, V, generator=g) * 0.65
    # rows 8..15: quantise so that exact ties (also at the k-th value) occur
    logits[8:] = (logits[8:] * 8).round() / 8
    noise = torch.empty(16, V).exponential_(1, generator=g)
    out = dict(logits=logits.numpy(), noise=noise.numpy())
    for topk in (1, 50):
        for T in (0.7, 1.0):
            REF_orig = REF._multinomial_sample_one_no_sync
            REF._multinomial_sample_one_no_sync = lambda p: torch.argmax(p / noise.to(p.dtype), -1, keepdim=True).to(torch.int)
            try:
                idx = REF.sample_topk(logits, topk, T)
            finally:
                REF._multinomial_sample_one_no_sync = REF_orig
            oidx = O.sample_topk(logits, topk, T, noise)
            assert torch.equal(idx, oidx)
            out[f"idx_k{topk}_T{T}"] = idx.squeeze(-1).numpy()
    np.savez_compressed(os.path.join(GOLD, "sampler.npz"), **out)
    print("[golden] sampler written", flush=True)


def gen_1b(which, frames2):
    cfg = CSMConfig()
    t0 = time.time()
    if "cfg1" in which:
        sd = synth_state_dict(cfg, seed=0)
        print(f"[golden] csm-1b fp32 weights {time.time() - t0:.1f}s", flush=True)
        ids, mask = synth_context(cfg, 1, 16, 48, seed=1)
        run_case("csm1b_cfg1_fp32", cfg, sd, torch.float32, ids, mask, 8)
        del sd
    sdb = synth_state_dict(cfg, seed=0, bf16_representable=True)
    if "cfg1" in which:
        ids, mask = synth_context(cfg, 1, 16, 48, seed=1)
        run_case("csm1b_cfg1_bf16w_fp32", cfg, sdb, torch.float32, ids, mask, 8)
        run_case("csm1b_cfg1_bf16", cfg, sdb, torch.bfloat16, ids, mask, 8)
    if "prefill512" in which:
        ids, mask = synth_context(cfg, 1, 128, 384, seed=2)
        run_case("csm1b_prefill512_bf16w_fp32", cfg, sdb, torch.float32, ids, mask, 1, topn_keep=8)
        run_case("csm1b_prefill512_bf16", cfg, sdb, torch.bfloat16, ids, mask, 1, topn_keep=8)
    if "cfg2" in which:
        ids, mask = synth_context(cfg, 1, 128, 384, seed=2)
        run_case("csm1b_cfg2_bf16w_fp32", cfg, sdb, torch.float32, ids, mask, frames2, topn_keep=2)
    if "b4" in which:
        # BASELINE configs[2] layout ("voice cloning"): 48 text + 400 audio + the all-zero EOS audio frame + 63 text
        # frames = 512, four DISTINCT rows, so the batched (matrix-core) decode kernels are pinned against the
        # reference itself rather than against the engine's own single-row kernels
        ids, mask = synth_context(cfg, 4, 48, 400, seed=3, tail_text=63, eos_frame=True)
        run_case("csm1b_b4_ctx512_bf16w_fp32", cfg, sdb, torch.float32, ids, mask, 4, topn_keep=2)
    if "b4noise" in which:
        # the sampler in situ: top-k = 50, T = 1.0 on all 32 codebooks with explicit Exp(1) noise [n,B,C,V]
        ids, mask = synth_context(cfg, 4, 48, 400, seed=3, tail_text=63, eos_frame=True)
        n = 2
        noise = torch.empty(n, 4, cfg.audio_num_codebooks, cfg.audio_vocab_size).exponential_(1, generator=torch.Generator().manual_seed(11))
        run_case("csm1b_b4_topk50_noise_bf16w_fp32", cfg, sdb, torch.float32, ids, mask, n, topn_keep=2, topk=50,
                 temperature=1.0, noise=noise, extra=dict(noise_seed=np.int32(11)))


def processor_cases():
    """Inputs shared by make_golden.py (reference processor) and tests/test_processor.py (ours)."""
    g = torch.Generator().manual_seed(0)
    wavs = [torch.rand(1920 * 5 + 100, generator=g), torch.rand(1920 * 3, generator=g), torch.rand(1920 * 9, generator=g)]
    convo1 = [{"role": "speaker_0", "content": [{"type": "text", "text": "Hello there"}, {"type": "audio"}]},
              {"role": "speaker_1", "content": [{"type": "text", "text": "Hi"}, {"type": "audio"}]},
              {"role": "speaker_0", "content": [{"type": "text", "text": "How are you today?"}]}]
    convo2 = [{"role": "speaker_3", "content": [{"type": "text", "text": "Short"}, {"type": "audio"}]}]
    return {
        "single": dict(messages=convo1, audios=wavs[:2]),
        "single_noamort": dict(messages=convo1, audios=wavs[:2], amortize_decoder_training=False, messages_training_mask=[1, 0, 1]),
        "trunc": dict(messages=convo1, audios=wavs[:2], max_length=20, amortize_decoder_training=False),
        "batch": dict(messages=[convo1, convo2], audios=[wavs[:2], [wavs[2]]], amortization_ratio=4),
    }


def gen_processor():
    """the reference CSMProcessor (processor.py) driven with stub tokenizers."""
    import random
    import processor as REFP  # the reference, unmodified
    from oracle.stub_tokenizers import StubTextTokenizer, StubAudioTokenizer
    ref = REFP.CSMProcessor(StubTextTokenizer(), StubAudioTokenizer())
    out = {}
    for name, kw in processor_cases().items():
        random.seed(5)
        r = ref(**kw)
        for k in ("input_ids", "attention_mask", "labels"):
            out[f"{name}.{k}"] = r[k].numpy()
    np.savez_compressed(os.path.join(GOLD, "processor.npz"), **out)
    print("[golden] processor written", flush=True)


def loss_inputs(cfg, batch, n_text, n_audio, seed):
    """Context + labels in the layout the reference's processor produces for training (processor.py:330-378): labels
    repeat the audio tokens of audio frames, text frames are -100; a few frames are dropped from the decoder loss (the
    1/16 amortisation leaves most frames unlabelled: here ~1/3 kept), and one frame has a single codebook masked."""
    ids, mask = synth_context(cfg, batch, n_text, n_audio, seed=seed)
    C = cfg.audio_num_codebooks
    labels = torch.full_like(ids, -100)
    labels[:, n_text:, :C] = ids[:, n_text:, :C]
    g = torch.Generator().manual_seed(seed)
    drop = torch.rand(batch, n_audio, generator=g) > 0.34
    drop[:, 0] = False                                   # the first audio frame stays (its predecessor is a text frame)
    for b in range(batch):
        for t in range(n_audio):
            if drop[b, t]:
                labels[b, n_text + t, 1:C] = -100        # codebook 0 stays labelled: it feeds the backbone loss
    labels[batch - 1, n_text + 1, 5] = -100              # a frame with ONE missing codebook is not a decoder frame
    return ids, mask, labels


def gen_loss():
    """Training forward (modeling_csm.py:367-465): loss / backbone_loss / decoder_loss of the reference itself."""
    for name, cfg, dtype, shape in (("tiny", CSMConfig.tiny(), torch.float32, (2, 4, 10)),
                                    ("csm1b", CSMConfig(), torch.bfloat16, (2, 6, 18))):
        t0 = time.time()
        if name == "tiny":
            sd = synth_state_dict(cfg, seed=0, std=0.05)
        else:
            sd = synth_state_dict(cfg, seed=0, dtype=dtype, bf16_representable=True)
        sd32 = {k: v.float() for k, v in sd.items()}
        model = build_ref(cfg, sd32, torch.float32)      # fp32 arithmetic on the (bf16-representable) weights
        ids, mask, labels = loss_inputs(cfg, *shape, seed=41)
        with torch.no_grad():
            out = model(input_ids=ids, attention_mask=mask, labels=labels, return_dict=True)
        o = O.forward_loss(sd32, cfg, ids, mask, labels)
        assert abs(float(o[0]) - float(out.loss)) < 2e-5 * abs(float(out.loss)), (o[0], out.loss)
        assert abs(float(o[1]) - float(out.backbone_loss)) < 2e-5 and abs(float(o[2]) - float(out.decoder_loss)) < 2e-5
        np.savez_compressed(os.path.join(GOLD, f"{name}_loss.npz"), input_ids=ids.numpy(), attention_mask=mask.numpy(),
                            labels=labels.numpy(), loss=np.float32(out.loss), backbone_loss=np.float32(out.backbone_loss),
                            decoder_loss=np.float32(out.decoder_loss), last_h=out.last_hidden_state.float().numpy(),
                            c0_logits=out.logits.float().numpy())
        print(f"[golden] {name}_loss written: loss {float(out.loss):.6f} = {float(out.backbone_loss):.6f} + "
              f"{float(out.decoder_loss):.6f}  ({time.time() - t0:.1f}s)", flush=True)


GRAD_KEYS = ["backbone.layers.0.self_attn.q_proj.weight", "backbone.layers.0.self_attn.k_proj.weight",
             "backbone.layers.1.self_attn.v_proj.weight", "backbone.layers.1.self_attn.o_proj.weight",
             "backbone.layers.0.mlp.gate_proj.weight", "backbone.layers.1.mlp.down_proj.weight",
             "backbone.layers.0.input_layernorm.weight", "backbone.layers.1.post_attention_layernorm.weight", "backbone.norm.weight",
             "decoder.layers.0.self_attn.q_proj.weight", "decoder.layers.1.self_attn.k_proj.weight", "decoder.layers.0.self_attn.o_proj.weight",
             "decoder.layers.1.mlp.up_proj.weight", "decoder.layers.1.input_layernorm.weight",
             "decoder.norm.weight", "projection.weight", "codebook0_head.weight"]


def gen_grad():
    """Training BACKWARD (consumer train.py:308-326): gradients the REFERENCE's `loss.backward()` leaves in `.grad`, tiny
    configuration, fp32, the committed `tiny_loss` inputs.  Stored: a selection of whole matrices (GRAD_KEYS), one slice of
    `audio_head`, the touched rows of the two embedding tables, and the gradient NORM of every parameter.  The oracle's own
    autograd (torch.autograd through oracle/csm_oracle.py: forward_loss) is checked against all of it here."""
    cfg = CSMConfig.tiny()
    sd = synth_state_dict(cfg, seed=0, std=0.05)
    model = build_ref(cfg, {k: v.float() for k, v in sd.items()}, torch.float32)
    for p_ in model.parameters():
        p_.requires_grad_(True)
    ids, mask, labels = loss_inputs(cfg, 2, 4, 10, seed=41)
    out = model(input_ids=ids, attention_mask=mask, labels=labels, return_dict=True)
    out.loss.backward()
    ref = {k: p_.grad.detach().clone() for k, p_ in model.named_parameters() if p_.grad is not None}
    osd = {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    with torch.enable_grad():
        O.forward_loss(osd, cfg, ids, mask, labels)[0].backward()
    worst = 0.0
    for k, gr in ref.items():
        err = float((osd[k].grad - gr).norm() / gr.norm().clamp_min(1e-20))
        worst = max(worst, err)
    assert set(ref) == set(osd) and worst < 1e-4, worst
    store = {"loss": np.float32(out.loss.detach())}
    for k in GRAD_KEYS:
        store["g." + k] = ref[k].numpy()
    store["g.audio_head.5"] = ref["audio_head"][5].numpy()
    for k in ("text_embeddings.weight", "audio_embeddings.weight"):
        rows = (ref[k].abs().sum(1) > 0).nonzero()[:, 0]
        store["rows." + k] = rows.numpy()
        store["g." + k] = ref[k][rows].numpy()
    names = sorted(ref)
    store["norm_names"] = np.array(names)
    store["norms"] = np.array([float(ref[k].norm()) for k in names], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, "tiny_grad.npz"), **store)
    print(f"[golden] tiny_grad written: {len(ref)} parameters, oracle autograd vs reference worst rel err {worst:.2e}", flush=True)


def gen_grad_1b():
    """csm-1b (bf16-representable weights, fp32 arithmetic), the committed `csm1b_loss` inputs: the gradient NORM of every
    parameter after the reference's `loss.backward()` and the leading 64 x 64 block of a few matrices (12 GB of gradients
    do not fit a fixture)."""
    cfg = CSMConfig()
    sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, bf16_representable=True)
    model = build_ref(cfg, {k: v.float() for k, v in sd.items()}, torch.float32)
    del sd
    for p_ in model.parameters():
        p_.requires_grad_(True)
    ids, mask, labels = loss_inputs(cfg, 2, 6, 18, seed=41)
    out = model(input_ids=ids, attention_mask=mask, labels=labels, return_dict=True)
    out.loss.backward()
    ref = {k: p_.grad for k, p_ in model.named_parameters() if p_.grad is not None}
    names = sorted(ref)
    store = {"loss": np.float32(out.loss.detach()), "norm_names": np.array(names),
             "norms": np.array([float(ref[k].double().norm()) for k in names], dtype=np.float64)}
    for k in ("backbone.layers.0.self_attn.q_proj.weight", "backbone.layers.15.mlp.down_proj.weight", "backbone.layers.7.mlp.gate_proj.weight",
              "decoder.layers.0.self_attn.v_proj.weight", "decoder.layers.3.mlp.up_proj.weight", "projection.weight", "codebook0_head.weight"):
        store["blk." + k] = ref[k][:64, :64].numpy().copy()
    store["blk.audio_head.7"] = ref["audio_head"][7, :64, :64].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "csm1b_grad.npz"), **store)
    print(f"[golden] csm1b_grad written: {len(ref)} parameters, loss {float(out.loss):.6f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="tiny,sampler,processor,cfg1,prefill512,cfg2,b4,b4noise,loss")
    ap.add_argument("--frames2", type=int, default=200)
    a = ap.parse_args()
    which = set(a.only.split(","))
    torch.manual_seed(0)
    if "tiny" in which:
        gen_tiny()
    if "sampler" in which:
        gen_sampler()
    if "processor" in which:
        gen_processor()
    if which & {"cfg1", "prefill512", "cfg2", "b4", "b4noise"}:
        gen_1b(which, a.frames2)
    if "loss" in which:
        gen_loss()
    if "grad" in which:
        gen_grad()
    if "grad1b" in which:
        gen_grad_1b()
    if "rng" in which:
        gen_rng()
    if "rng1b" in which:
        gen_rng_1b()


if __name__ == "__main__":
    main()
