"""Deterministic synthetic weights and token contexts for CSM.

There are no pretrained csm-1b weights in this environment (SURVEY.md §0 finding 5), so every
parity test and benchmark runs on seeded synthetic weights of the real architecture.  The generator is
a counter-based integer hash evaluated with torch integer ops only, followed by exact float
arithmetic, so a tensor generated on the CPU (oracle side, golden fixtures) and on an MI355X
(product side) is bit-identical by construction -- unlike `torch.randn`, whose CPU and GPU streams
differ.

State-dict key layout = the reference checkpoint layout (reference
`modeling_csm.py:214-245` for the top-level tensors, `transformers.LlamaModel` naming below them).
The reference leaves `audio_head` uninitialised (`modeling_csm.py:236-240`); here it is filled like any
other matrix.
"""
from __future__ import annotations

import hashlib
from typing import Dict, Iterator, Tuple

import torch

_M32 = 0xFFFFFFFF


def _key32(name: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:4], "little")


def _hash32(x: torch.Tensor) -> torch.Tensor:
    """32-bit finalizer (lowbias32-style) emulated in int64; every intermediate < 2**63."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x21F0AAAD) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x735A2D97) & _M32
    x = x ^ (x >> 15)
    return x


def hash_uniform(name: str, numel: int, seed: int, device="cpu", chunk: int = 1 << 24) -> torch.Tensor:
    """fp32 tensor of `numel` values in (-1, 1), exactly reproducible on any device."""
    key = _key32(name, seed)
    out = torch.empty(numel, dtype=torch.float32, device=device)
    for s in range(0, numel, chunk):
        e = min(numel, s + chunk)
        idx = torch.arange(s, e, dtype=torch.int64, device=device)
        h = _hash32(idx + key)
        h = _hash32(h ^ ((key * 0x9E37) & _M32))
        # 24 random bits -> (k + 0.5 - 2^23) / 2^23, exact in fp32
        v = (h & 0xFFFFFF).to(torch.float32)
        out[s:e] = (v - 8388607.5) * (1.0 / 8388608.0)
    return out


def synth_tensor(name: str, shape, std: float, seed: int, device="cpu", mean: float = 0.0) -> torch.Tensor:
    """Uniform with the requested standard deviation (logits are CLT-Gaussian anyway)."""
    numel = 1
    for d in shape:
        numel *= int(d)
    u = hash_uniform(name, numel, seed, device)
    scale = torch.tensor(std * (3.0 ** 0.5), dtype=torch.float32, device=device)
    t = u * scale
    if mean != 0.0:
        t = t + torch.tensor(mean, dtype=torch.float32, device=device)
    return t.view(*shape)


def state_dict_spec(cfg) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """Yield (key, shape, kind) for the reference checkpoint layout; kind in {matrix, norm}."""
    bh = cfg.backbone_config.hidden_size
    dh = cfg.decoder_config.hidden_size
    V = cfg.audio_vocab_size
    C = cfg.audio_num_codebooks
    yield "text_embeddings.weight", (cfg.text_vocab_size, bh), "matrix"
    yield "audio_embeddings.weight", (V * C, bh), "matrix"
    yield "projection.weight", (dh, bh), "matrix"
    yield "codebook0_head.weight", (V, bh), "matrix"
    yield "audio_head", (C - 1, dh, V), "matrix"
    for prefix, lc in (("backbone", cfg.backbone_config), ("decoder", cfg.decoder_config)):
        H = lc.hidden_size
        hd = lc.head_dim
        nq, nkv, F = lc.num_attention_heads, lc.num_key_value_heads, lc.intermediate_size
        for i in range(lc.num_hidden_layers):
            p = f"{prefix}.layers.{i}"
            yield f"{p}.self_attn.q_proj.weight", (nq * hd, H), "matrix"
            yield f"{p}.self_attn.k_proj.weight", (nkv * hd, H), "matrix"
            yield f"{p}.self_attn.v_proj.weight", (nkv * hd, H), "matrix"
            yield f"{p}.self_attn.o_proj.weight", (H, nq * hd), "matrix"
            yield f"{p}.mlp.gate_proj.weight", (F, H), "matrix"
            yield f"{p}.mlp.up_proj.weight", (F, H), "matrix"
            yield f"{p}.mlp.down_proj.weight", (H, F), "matrix"
            yield f"{p}.input_layernorm.weight", (H,), "norm"
            yield f"{p}.post_attention_layernorm.weight", (H,), "norm"
        yield f"{prefix}.norm.weight", (lc.hidden_size,), "norm"


def synth_state_dict(cfg, seed: int = 0, dtype=torch.float32, device="cpu", std: float = 0.02,
                     bf16_representable: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded synthetic checkpoint.

    `bf16_representable=True` rounds every value to bf16 first and stores it in `dtype`, which lets an
    fp32 model and a bf16 model share numerically identical weights (used to pin the bf16-weight GPU
    path against the reference run in fp32 arithmetic).
    """
    sd = {}
    for key, shape, kind in state_dict_spec(cfg):
        if kind == "norm":
            t = synth_tensor(key, shape, 0.05, seed, device, mean=1.0)
        else:
            t = synth_tensor(key, shape, std, seed, device)
        if bf16_representable:
            t = t.to(torch.bfloat16)
        sd[key] = t.to(dtype)
    return sd


def synth_context(cfg, batch: int, n_text: int, n_audio: int, seed: int, tail_text: int = 0,
                  eos_frame: bool = False):
    """Synthetic `[B, S, C+1]` context and mask.

    Layout per row: `n_text` text frames (text column live), `n_audio` audio frames (codebook columns
    live), optionally one all-zero EOS audio frame and `tail_text` more text frames.  Live tokens avoid
    id 0 so no accidental all-zero frame appears.  Returns (input_ids int64, attention_mask int32).
    """
    C = cfg.audio_num_codebooks
    S = n_text + n_audio + (1 if eos_frame else 0) + tail_text
    ids = torch.zeros(batch, S, C + 1, dtype=torch.int64)
    mask = torch.zeros(batch, S, C + 1, dtype=torch.int32)
    for b in range(batch):
        name = f"ctx:{b}"
        u = hash_uniform(name, S * (C + 1), seed).view(S, C + 1).double() * 0.5 + 0.5  # (0,1)
        text = (u[:, C] * (cfg.text_vocab_size - 1)).long().clamp_(0, cfg.text_vocab_size - 2) + 1
        audio = (u[:, :C] * (cfg.audio_vocab_size - 1)).long().clamp_(0, cfg.audio_vocab_size - 2) + 1
        s = 0
        ids[b, s:s + n_text, C] = text[s:s + n_text]
        mask[b, s:s + n_text, C] = 1
        s += n_text
        ids[b, s:s + n_audio, :C] = audio[s:s + n_audio]
        mask[b, s:s + n_audio, :C] = 1
        s += n_audio
        if eos_frame:
            mask[b, s, :C] = 1  # all-zero audio frame, live
            s += 1
        if tail_text:
            ids[b, s:s + tail_text, C] = text[s:s + tail_text]
            mask[b, s:s + tail_text, C] = 1
    return ids, mask
