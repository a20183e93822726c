"""CPU ORACLE for the CSM generation path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module,
and only as the checker / the timed CPU baseline.  The product path (`csm_hf_amd`) never imports it and
fails loudly when the HIP extension is missing.

What it is: a plain-torch (CPU) restatement of `CSMModel.generate / generate_frame / forward`
(`/root/reference/modeling_csm.py:170-189, 247-282, 321-365, 508-589, 631-702`) and of the
third-party `transformers.LlamaModel` arithmetic those functions call (the reference pins
`transformers==4.49.0`, `requirements.txt:4`; the transformer math is NOT under /root/reference --
SURVEY.md §8-c).  It imports neither the reference nor `transformers`.

Parity pin: the reference holds no tests, golden vectors or known-answer values for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself, run in the build
container by `oracle/make_golden.py` (which imports /root/reference with the decode-mask shim of
SURVEY.md Appendix B-1) and committed under `tests/golden/`.  `tests/test_oracle_golden.py` checks
this file against those vectors: fp32 token ids bit-exact, fp32 hidden states/logits to 1e-5.

Every op is issued in the order, dtype and shape the reference's eager execution issues it, so that in
fp32 the results are bit-identical to the reference on the same host/thread count and in bf16 every
rounding point of the reference ("->D") is reproduced.

Deliberate deviation (documented in DESIGN.md): left-padded batch rows mask their pad keys at EVERY
step (the reference forgets the pad mask on decode steps), so a padded row
equals its solo run.  Un-padded inputs are unaffected.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------
# A3  llama3 RoPE   (transformers modeling_rope_utils._compute_llama3_parameters; parameters from
#                    reference modeling_csm.py:78-85, 99-106)
# --------------------------------------------------------------------------------------------------
def llama3_inv_freq(head_dim: int, base: float, rope_scaling: Optional[dict]) -> torch.Tensor:
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(torch.float) / head_dim))
    if not rope_scaling or rope_scaling.get("type", rope_scaling.get("rope_type")) in (None, "default"):
        return inv_freq
    factor = rope_scaling["factor"]
    low = rope_scaling["low_freq_factor"]
    high = rope_scaling["high_freq_factor"]
    old_ctx = rope_scaling["original_max_position_embeddings"]
    low_wavelen = old_ctx / low
    high_wavelen = old_ctx / high
    wavelen = 2 * math.pi / inv_freq
    inv_llama = torch.where(wavelen > low_wavelen, inv_freq / factor, inv_freq)
    smooth = (old_ctx / wavelen - low) / (high - low)
    smoothed = (1 - smooth) * inv_llama / factor + smooth * inv_llama
    is_medium = ~(wavelen < high_wavelen) * ~(wavelen > low_wavelen)
    return torch.where(is_medium, smoothed, inv_llama)


def rope_cos_sin(inv_freq: torch.Tensor, position_ids: torch.Tensor, dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """LlamaRotaryEmbedding.forward (transformers modeling_llama.py:113-127): fp32 angle, cast to D."""
    inv = inv_freq[None, :, None].expand(position_ids.shape[0], -1, 1).to(torch.float)
    pos = position_ids[:, None, :].float()
    freqs = (inv @ pos).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):
    """modeling_llama.py:130-160 (half-split pairing; three roundings in eager)."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


# --------------------------------------------------------------------------------------------------
# A2  RMSNorm (modeling_llama.py:62-67)
# --------------------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    dt = x.dtype
    x32 = x.to(torch.float32)
    var = x32.pow(2).mean(-1, keepdim=True)
    x32 = x32 * torch.rsqrt(var + eps)
    return w * x32.to(dt)


def repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    b, h, s, d = x.shape
    if n_rep == 1:
        return x
    return x[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


@dataclass
class KVCache:
    """DynamicCache restated (transformers cache_utils.py:113-158): per-layer K/V grown by cat."""
    keys: List[Optional[torch.Tensor]] = field(default_factory=list)
    values: List[Optional[torch.Tensor]] = field(default_factory=list)
    key_valid: Optional[torch.Tensor] = None   # [B, L] bool; None = all valid (oracle's pad semantics)

    def length(self) -> int:
        return 0 if not self.keys or self.keys[0] is None else self.keys[0].shape[2]

    def update(self, layer: int, k: torch.Tensor, v: torch.Tensor):
        while len(self.keys) <= layer:
            self.keys.append(None)
            self.values.append(None)
        if self.keys[layer] is None:
            self.keys[layer], self.values[layer] = k, v
        else:
            self.keys[layer] = torch.cat([self.keys[layer], k], dim=-2)
            self.values[layer] = torch.cat([self.values[layer], v], dim=-2)
        return self.keys[layer], self.values[layer]


# --------------------------------------------------------------------------------------------------
# A4-A6  LlamaModel.forward with embed_tokens = Identity (modeling_llama.py:367-417, reference
#        call sites modeling_csm.py:345-354, 545-552, 568-576)
# --------------------------------------------------------------------------------------------------
def llama_forward(sd: Dict[str, torch.Tensor], prefix: str, lc, h: torch.Tensor,
                  position_ids: Optional[torch.Tensor], cache: Optional[KVCache],
                  new_valid: Optional[torch.Tensor] = None, hidden_trace: Optional[list] = None):
    """h [B,S,H] -> (final-normed hidden [B,S,H], cache).  `new_valid` [B,S] marks non-pad frames."""
    B, S, H = h.shape
    nq, nkv, hd = lc.num_attention_heads, lc.num_key_value_heads, lc.head_dim
    eps = lc.rms_norm_eps
    scaling = hd ** -0.5
    if cache is None:
        cache = KVCache()
    past = cache.length()
    if position_ids is None:  # modeling_llama.py:386-389 -- absolute cache index, NOT padding-aware
        position_ids = (torch.arange(S) + past).unsqueeze(0)
    inv_freq = llama3_inv_freq(hd, lc.rope_theta, lc.rope_scaling)
    cos, sin = rope_cos_sin(inv_freq, position_ids, h.dtype)

    # key-validity bookkeeping (oracle's padding semantics: pads masked at every step)
    if new_valid is not None and not bool(new_valid.all()):
        kv_new = new_valid.bool()
    else:
        kv_new = None
    if cache.key_valid is not None or kv_new is not None:
        old = cache.key_valid if cache.key_valid is not None else torch.ones(B, past, dtype=torch.bool)
        new = kv_new if kv_new is not None else torch.ones(B, S, dtype=torch.bool)
        cache.key_valid = torch.cat([old, new], dim=1)
    L = past + S
    attn_mask = None
    if cache.key_valid is not None:
        qpos = torch.arange(past, L)
        causal = qpos[:, None] >= torch.arange(L)[None, :]                   # [S, L]
        diag = qpos[:, None] == torch.arange(L)[None, :]                     # pad queries see themselves
        attn_mask = causal[None, None] & (cache.key_valid[:, None, None, :] | diag[None, None])

    for i in range(lc.num_hidden_layers):
        p = f"{prefix}.layers.{i}"
        r = h
        x = rmsnorm(h, sd[f"{p}.input_layernorm.weight"], eps)
        q = F.linear(x, sd[f"{p}.self_attn.q_proj.weight"]).view(B, S, -1, hd).transpose(1, 2)
        k = F.linear(x, sd[f"{p}.self_attn.k_proj.weight"]).view(B, S, -1, hd).transpose(1, 2)
        v = F.linear(x, sd[f"{p}.self_attn.v_proj.weight"]).view(B, S, -1, hd).transpose(1, 2)
        q, k = apply_rope(q, k, cos, sin)
        k, v = cache.update(i, k, v)
        if attn_mask is None:
            # sdpa_attention.py:97-163: no mask -> is_causal iff q_len > 1, native GQA
            a = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, scale=scaling,
                                               is_causal=S > 1, enable_gqa=True)
        else:
            a = F.scaled_dot_product_attention(q, repeat_kv(k, nq // nkv), repeat_kv(v, nq // nkv),
                                               attn_mask=attn_mask, dropout_p=0.0, scale=scaling,
                                               is_causal=False)
        a = a.transpose(1, 2).contiguous().reshape(B, S, -1).contiguous()
        h = r + F.linear(a, sd[f"{p}.self_attn.o_proj.weight"])
        r = h
        x = rmsnorm(h, sd[f"{p}.post_attention_layernorm.weight"], eps)
        g = F.linear(x, sd[f"{p}.mlp.gate_proj.weight"])
        u = F.linear(x, sd[f"{p}.mlp.up_proj.weight"])
        h = r + F.linear(F.silu(g) * u, sd[f"{p}.mlp.down_proj.weight"])
        if hidden_trace is not None:
            hidden_trace.append(h)
    return rmsnorm(h, sd[f"{prefix}.norm.weight"], eps), cache


# --------------------------------------------------------------------------------------------------
# A1  frame embedding (reference modeling_csm.py:247-282, 327-342)
# --------------------------------------------------------------------------------------------------
def embed_audio(sd, cfg, codebook: int, tokens: torch.Tensor) -> torch.Tensor:
    return F.embedding(tokens + codebook * cfg.audio_vocab_size, sd["audio_embeddings.weight"])


def embed_frames(sd, cfg, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor]):
    """[B,S,C+1] ids (+mask) -> (h [B,S,H], frame_valid [B,S] bool)."""
    B, S, _ = input_ids.shape
    C, V = cfg.audio_num_codebooks, cfg.audio_vocab_size
    text = F.embedding(input_ids[:, :, -1], sd["text_embeddings.weight"]).unsqueeze(-2)
    audio_tokens = input_ids[:, :, :-1] + V * torch.arange(C)
    audio = F.embedding(audio_tokens.view(-1), sd["audio_embeddings.weight"]).reshape(B, S, C, -1)
    embeds = torch.cat([audio, text], dim=-2)
    if attention_mask is not None:
        embeds = embeds * attention_mask.unsqueeze(-1)
        valid = attention_mask.sum(dim=-1) > 0
    else:
        valid = torch.ones(B, S, dtype=torch.bool)
    return embeds.sum(dim=2), valid


# --------------------------------------------------------------------------------------------------
# A8  sampler (reference modeling_csm.py:170-189)
# --------------------------------------------------------------------------------------------------
def sample_topk(logits: torch.Tensor, topk: int, temperature: float,
                noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Returns int32 [...,1].  `noise` (same shape as logits, ~Exp(1)) replaces the torch RNG draw."""
    logits = logits / temperature
    kth = torch.topk(logits, topk)[0][..., -1, None]
    scores = logits.masked_fill(logits < kth, -float("Inf"))
    scores = F.log_softmax(scores, dim=-1)
    probs = F.softmax(scores, dim=-1)
    q = torch.empty_like(probs).exponential_(1) if noise is None else noise.to(probs.dtype)
    return torch.argmax(probs / q, dim=-1, keepdim=True).to(dtype=torch.int)


@dataclass
class FrameOut:
    samples: torch.Tensor                  # [B, C] int64 (the model's own samples)
    last_hidden_state: torch.Tensor        # [B, Hb]
    logits: torch.Tensor                   # [B, V] codebook-0 logits
    cache: Optional[KVCache]
    all_logits: Optional[torch.Tensor] = None   # [B, C, V] logits of every codebook (trace)


# --------------------------------------------------------------------------------------------------
# reference CSMModel.forward, inference branch (modeling_csm.py:321-365)
# --------------------------------------------------------------------------------------------------
def forward(sd, cfg, input_ids, attention_mask, cache: Optional[KVCache] = None, use_cache: bool = True,
            hidden_trace: Optional[list] = None, position_ids: Optional[torch.Tensor] = None):
    """`position_ids` [B,S] | [1,S] | None: forwarded to the backbone as the reference does (modeling_csm.py:349)."""
    h, valid = embed_frames(sd, cfg, input_ids, attention_mask)
    hb, cache = llama_forward(sd, "backbone", cfg.backbone_config, h, position_ids, cache if use_cache else None,
                              new_valid=valid, hidden_trace=hidden_trace)
    c0_all = F.linear(hb, sd["codebook0_head.weight"])          # all S positions, like the reference
    return hb[:, -1, :], c0_all[:, -1, :], (cache if use_cache else None)


# --------------------------------------------------------------------------------------------------
# reference CSMModel.forward, labels branch (modeling_csm.py:367-465): codebook-0 cross-entropy over all positions
# (shifted by one) + decoder cross-entropy over the frames whose 32 audio labels are all present
# --------------------------------------------------------------------------------------------------
def forward_loss(sd, cfg, input_ids, attention_mask, labels):
    """Returns (loss, backbone_loss, decoder_loss, last_hidden [B,H], c0_logits [B,V]) like the reference's training
    forward does for `labels` [B,S,C+1] (-100 = ignored).  fp32 tensors, no autograd bookkeeping."""
    C, V = cfg.audio_num_codebooks, cfg.audio_vocab_size
    h, valid = embed_frames(sd, cfg, input_ids, attention_mask)
    hb, _ = llama_forward(sd, "backbone", cfg.backbone_config, h, None, None, new_valid=valid)   # :345-354
    c0_all = F.linear(hb, sd["codebook0_head.weight"])                                            # :361
    # :373-386  predict label[t+1] from hidden[t]
    backbone_loss = F.cross_entropy(c0_all[:, :-1, :].reshape(-1, V).float(), labels[:, 1:, 0].reshape(-1), ignore_index=-100)
    audio_tokens, audio_labels = input_ids[:, :, :C], labels[:, :, :C]                            # :389-392
    frames = (audio_labels != -100).all(dim=2).nonzero(as_tuple=False)                           # :395-396
    if frames.numel() > 0:
        b, t = frames[:, 0], frames[:, 1]
        frame_hidden = hb[b, t - 1]                       # :402-404 (t = 0 wraps to the last position, as indexing does)
        toks, labs = audio_tokens[b, t], audio_labels[b, t]
        proj = sd["projection.weight"]
        emb = F.embedding((toks + torch.arange(C) * V).view(-1), sd["audio_embeddings.weight"]).view(len(b), C, -1)   # :416-431
        dec_in = torch.cat([F.linear(frame_hidden, proj).unsqueeze(1), F.linear(emb, proj)], dim=1)                    # :434-440
        dh, _ = llama_forward(sd, "decoder", cfg.decoder_config, dec_in, None, None)              # :441-444, default causal
        logits = torch.einsum("fcd,cdv->fcv", dh[:, 1:C, :], sd["audio_head"])                    # :447-454
        decoder_loss = F.cross_entropy(logits.reshape(-1, V), labs[:, 1:].reshape(-1), ignore_index=-100)   # :458-463
    else:
        decoder_loss = torch.tensor(0.0)
    return backbone_loss + decoder_loss, backbone_loss, decoder_loss, hb[:, -1, :], c0_all[:, -1, :]


# --------------------------------------------------------------------------------------------------
# reference CSMModel.generate_frame (modeling_csm.py:508-589)
# --------------------------------------------------------------------------------------------------
def generate_frame(sd, cfg, input_ids, attention_mask, temperature=1.0, topk=50, cache=None, use_cache=True,
                   noise: Optional[torch.Tensor] = None, forced: Optional[torch.Tensor] = None,
                   trace_logits: bool = False) -> FrameOut:
    """`noise` [B,C,V] explicit Exp(1) draws; `forced` [B,C] teacher-forced tokens fed back instead of
    the model's own samples (the samples are still recorded)."""
    C = cfg.audio_num_codebooks
    last_h, c0_logits, cache = forward(sd, cfg, input_ids, attention_mask, cache, use_cache)
    B = last_h.size(0)
    tokens = torch.zeros(B, C, dtype=torch.long)
    logits_all = torch.zeros(B, C, cfg.audio_vocab_size, dtype=torch.float32) if trace_logits else None

    def nz(i):
        return None if noise is None else noise[:, i, :]

    c0 = sample_topk(c0_logits, topk, temperature, nz(0))
    tokens[:, 0] = c0.squeeze(-1)
    if trace_logits:
        logits_all[:, 0] = c0_logits.float()
    feed = c0 if forced is None else forced[:, 0:1].to(torch.int)
    c0_embed = embed_audio(sd, cfg, 0, feed)
    curr_h = torch.cat([last_h.unsqueeze(1), c0_embed], dim=1)
    curr_pos = torch.arange(0, curr_h.size(1)).unsqueeze(0).repeat(B, 1)
    proj = F.linear(curr_h, sd["projection.weight"])
    dh, dcache = llama_forward(sd, "decoder", cfg.decoder_config, proj, curr_pos, KVCache())
    for i in range(1, C):
        ci_logits = torch.matmul(dh[:, -1, :], sd["audio_head"][i - 1])
        ci = sample_topk(ci_logits, topk, temperature, nz(i))
        tokens[:, i] = ci.squeeze(-1)
        if trace_logits:
            logits_all[:, i] = ci_logits.float()
        if i < C - 1:
            feed = ci if forced is None else forced[:, i:i + 1].to(torch.int)
            e = embed_audio(sd, cfg, i, feed)
            p = F.linear(e, sd["projection.weight"])
            pos = torch.full((B, 1), i + 1)
            dh, dcache = llama_forward(sd, "decoder", cfg.decoder_config, p, pos, dcache)
    return FrameOut(tokens, last_h, c0_logits, cache, logits_all)


# --------------------------------------------------------------------------------------------------
# reference CSMModel.generate (modeling_csm.py:631-702)
# --------------------------------------------------------------------------------------------------
def generate(sd, cfg, input_ids, attention_mask, max_new_frames=100, temperature=1.0, topk=50,
             use_cache=True, stop_on_all_zeros=True, noise: Optional[torch.Tensor] = None,
             forced: Optional[torch.Tensor] = None, trace: Optional[dict] = None) -> torch.Tensor:
    """Returns LongTensor [B, n, C].  `noise` [n,B,C,V]; `forced` [B,n,C]; `trace` collects
    per-frame `last_h` [n,B,H] and `logits` [n,B,C,V] when a dict is passed."""
    B = input_ids.size(0)
    C = cfg.audio_num_codebooks
    frames = []
    cache = None
    ids, mask = input_ids, attention_mask
    full_ids, full_mask = input_ids, attention_mask
    for f in range(max_new_frames):
        out = generate_frame(sd, cfg, ids if use_cache else full_ids, mask if use_cache else full_mask,
                             temperature, topk, cache, use_cache,
                             None if noise is None else noise[f],
                             None if forced is None else forced[:, f],
                             trace_logits=trace is not None)
        new = out.samples
        cache = out.cache
        if trace is not None:
            trace.setdefault("last_h", []).append(out.last_hidden_state.float())
            trace.setdefault("logits", []).append(out.all_logits)
        if stop_on_all_zeros and bool(torch.all(new == 0)):
            break
        frames.append(new)
        fed = new if forced is None else forced[:, f]
        ids = torch.cat([fed, torch.zeros(B, 1, dtype=fed.dtype)], dim=1).unsqueeze(1)
        mask = torch.zeros(B, 1, C + 1, dtype=attention_mask.dtype)
        mask[:, :, :C] = 1
        if not use_cache:
            full_ids = torch.cat([full_ids, ids], dim=1)
            full_mask = torch.cat([full_mask, mask], dim=1)
    if trace is not None and trace.get("last_h"):
        trace["last_h"] = torch.stack(trace["last_h"])
        trace["logits"] = torch.stack(trace["logits"])
    if frames:
        return torch.stack(frames, dim=1)
    return torch.zeros(B, 0, C, dtype=torch.long)
