"""CPU restatement of Mimi's decode path -- TEST INFRASTRUCTURE ONLY (never imported by csm_hf_amd).

Algorithm: the Mimi codec of `moshi==0.2.2` (third-party, absent from /root/reference and from the image; the reference
calls it at README.md:58-60, 114-118 and train.py:363-365).  Restated from its published architecture as implemented by
transformers 5.15 `models/mimi/modeling_mimi.py` (cited per function), and PINNED against that implementation:
`oracle/make_golden_mimi.py` runs `transformers.MimiModel.decode` on seeded weights and stores its waveform next to this
oracle's (`tests/golden/mimi_*.npz`).  Plain torch functional ops, fp32.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def rvq_decode(sd, cfg, codes):
    """modeling_mimi.py:1070-1082, 1128-1139: codes [B, n_q, T] -> [B, hidden, T]"""
    out = 0.0
    k0 = 0
    for name, n in (("semantic", cfg.num_semantic_quantizers), ("acoustic", cfg.num_quantizers - cfg.num_semantic_quantizers)):
        p = f"quantizer.{name}_residual_vector_quantizer"
        q = 0.0
        for i in range(n):
            embed = sd[f"{p}.layers.{i}.codebook.embed_sum"] / sd[f"{p}.layers.{i}.codebook.cluster_usage"].clamp(min=1e-5)[:, None]   # :983-986
            q = q + F.embedding(codes[:, k0 + i], embed).permute(0, 2, 1)                                                        # :1004-1007, 1024-1027
        out = out + F.conv1d(q, sd[f"{p}.output_proj.weight"])                                                                   # :1080-1081
        k0 += n
    return out


def conv1d_causal(x, w, b, dilation=1):
    """MimiConv1d.forward with use_causal_conv, stride 1 (modeling_mimi.py:327-347): left pad (k-1)*dilation, no extra padding"""
    pad = (w.shape[-1] - 1) * dilation
    return F.conv1d(F.pad(x, (pad, 0)), w, b, dilation=dilation)


def convtr1d_causal(x, w, b, stride, groups=1):
    """MimiConvTranspose1d.forward, causal, trim_right_ratio 1 (modeling_mimi.py:399-405): trim k - stride samples on the right"""
    y = F.conv_transpose1d(x, w, b, stride=stride, groups=groups)
    return y[..., : y.shape[-1] - (w.shape[-1] - stride)]


def rope(x, pos, theta):
    """apply_rotary_pos_emb, default rope (modeling_mimi.py:524-548, rotate_half): x [B, h, L, hd]"""
    hd = x.shape[-1]
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    f = pos[:, None].float() * inv[None, :]
    cos, sin = torch.cat([f, f], -1).cos(), torch.cat([f, f], -1).sin()
    x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
    return x * cos + torch.cat([-x2, x1], -1) * sin


def transformer(sd, cfg, x):
    """MimiTransformerModel / MimiTransformerLayer (modeling_mimi.py:742-779, 870-925): x [B, L, H]"""
    B, L, H = x.shape
    nh, hd = cfg.num_attention_heads, cfg.head_dim
    pos = torch.arange(L)
    i, j = pos[:, None], pos[None, :]
    mask = (j <= i) & (j > i - cfg.sliding_window)                       # sliding-window causal mask
    for l in range(cfg.num_hidden_layers):
        p = f"decoder_transformer.layers.{l}"
        h = F.layer_norm(x, (H,), sd[f"{p}.input_layernorm.weight"], sd[f"{p}.input_layernorm.bias"], cfg.norm_eps)
        q = F.linear(h, sd[f"{p}.self_attn.q_proj.weight"]).view(B, L, nh, hd).transpose(1, 2)
        k = F.linear(h, sd[f"{p}.self_attn.k_proj.weight"]).view(B, L, nh, hd).transpose(1, 2)
        v = F.linear(h, sd[f"{p}.self_attn.v_proj.weight"]).view(B, L, nh, hd).transpose(1, 2)
        q, k = rope(q, pos, cfg.rope_theta), rope(k, pos, cfg.rope_theta)
        a = (q @ k.transpose(2, 3)) / math.sqrt(hd)
        a = a.masked_fill(~mask, float("-inf")).softmax(-1)
        o = (a @ v).transpose(1, 2).reshape(B, L, nh * hd)
        x = x + sd[f"{p}.self_attn_layer_scale.scale"] * F.linear(o, sd[f"{p}.self_attn.o_proj.weight"])
        h = F.layer_norm(x, (H,), sd[f"{p}.post_attention_layernorm.weight"], sd[f"{p}.post_attention_layernorm.bias"], cfg.norm_eps)
        h = F.linear(F.gelu(F.linear(h, sd[f"{p}.mlp.fc1.weight"])), sd[f"{p}.mlp.fc2.weight"])
        x = x + sd[f"{p}.mlp_layer_scale.scale"] * h
    return x


def seanet_decoder(sd, cfg, x):
    """MimiDecoder / MimiResnetBlock (modeling_mimi.py:931-961, 408-447): x [B, H, L] -> [B, 1, L * prod(ratios)]"""
    x = conv1d_causal(x, sd["decoder.layers.0.conv.weight"], sd["decoder.layers.0.conv.bias"])
    idx = 1
    for r in cfg.upsampling_ratios:
        x = convtr1d_causal(F.elu(x), sd[f"decoder.layers.{idx + 1}.conv.weight"], sd[f"decoder.layers.{idx + 1}.conv.bias"], r)
        p = f"decoder.layers.{idx + 2}.block"
        y = conv1d_causal(F.elu(x), sd[f"{p}.1.conv.weight"], sd[f"{p}.1.conv.bias"])
        y = conv1d_causal(F.elu(y), sd[f"{p}.3.conv.weight"], sd[f"{p}.3.conv.bias"])
        x = x + y
        idx += 3
    return conv1d_causal(F.elu(x), sd[f"decoder.layers.{idx + 1}.conv.weight"], sd[f"decoder.layers.{idx + 1}.conv.bias"])


def decode(sd, cfg, codes):
    """MimiModel._decode_frame / decode (modeling_mimi.py:1388-1455): codes [B, n_q, T] int64 -> waveform [B, 1, T * samples_per_frame]"""
    x = rvq_decode(sd, cfg, codes)
    x = convtr1d_causal(x, sd["upsample.conv.weight"], None, cfg.upsample_stride, groups=cfg.hidden_size)
    x = transformer(sd, cfg, x.transpose(1, 2)).transpose(1, 2)
    return seanet_decoder(sd, cfg, x)
