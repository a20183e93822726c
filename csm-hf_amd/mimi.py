"""Mimi decode on MI355X: the step right after the generation path.

The reference turns generated codes into audio with `audio_tokenizer.decode(gen_frames.permute(0, 2, 1))`
(/root/reference/README.md:114-118), where `audio_tokenizer` is the Mimi codec of the third-party package `moshi==0.2.2`
(/root/reference/requirements.txt:6, loaded at README.md:58-60 / train.py:363-365).  That package is not in the image;
its published architecture is also implemented by `transformers.models.mimi.modeling_mimi.MimiModel` (pinned here:
transformers 5.15), whose `decode()` is what this module mirrors (API, checkpoint key layout `kyutai/mimi`) and what the
parity fixtures are generated with (`oracle/make_golden_mimi.py`).

Decode path (modeling_mimi.py:1388-1406): split residual VQ decode (1 semantic + 31 acoustic codebooks, 256-d, 1x1 output
projections to 512) -> depthwise transposed conv upsample 12.5 -> 25 Hz -> 8-layer transformer (LayerNorm, RoPE, causal
sliding window 250, GELU MLP, layer scale) -> SEANet decoder (causal conv k=7, four [ELU, transposed conv x8 / x6 / x5 / x4,
residual block], ELU, conv k=3) -> 24 kHz waveform, 1920 samples per frame.

Everything runs in fp32 on the device through `libcsm_hip.so` (`csm_mimi_*`, include/csm_hip.h): every convolution and linear
is one launch of the exact-fp32 MFMA GEMM (activations are kept channels-last, so a causal convolution's k shifted rows are
one contiguous K = k * C_in operand row, and a stride-r transposed convolution is one GEMM with N = r * C_out); the small
element-wise steps have their own kernels (csrc/mimi.h).  torch is used to repack the weights once and to hold buffers.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch


@dataclass
class MimiDecodeConfig:
    """The fields of transformers.MimiConfig the decode path reads (defaults = kyutai/mimi)."""
    num_quantizers: int = 32
    num_semantic_quantizers: int = 1
    codebook_size: int = 2048
    codebook_dim: int = 256
    hidden_size: int = 512
    num_hidden_layers: int = 8
    num_attention_heads: int = 8
    head_dim: int = 64
    intermediate_size: int = 2048
    sliding_window: int = 250
    rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    upsampling_ratios: List[int] = field(default_factory=lambda: [8, 6, 5, 4])
    num_filters: int = 64
    kernel_size: int = 7
    last_kernel_size: int = 3
    residual_kernel_size: int = 3
    compress: int = 2
    upsample_stride: int = 2          # encodec frame rate 25 Hz / codec frame rate 12.5 Hz

    @property
    def samples_per_frame(self) -> int:
        return self.upsample_stride * math.prod(self.upsampling_ratios)

    @classmethod
    def tiny(cls) -> "MimiDecodeConfig":
        """Small shape for unit tests: same structure, GEMM-friendly sizes (multiples of 32 / 128)."""
        return cls(num_quantizers=4, codebook_size=64, codebook_dim=32, hidden_size=128, num_hidden_layers=2,
                   num_attention_heads=2, head_dim=64, intermediate_size=256, sliding_window=6,
                   upsampling_ratios=[4, 2], num_filters=64)


def decode_gemm_work(cfg: "MimiDecodeConfig", T: int):
    """Algorithmic work of one decode of T frames (one sequence), from the GEMM list of csrc/mimi.hip: decode_one --
    (flops of the matrix products incl. the windowed attention, bytes of the weight matrices read once).  Roofline inputs of
    tools/mimi_bench.py: one-shot decodes are matrix-pipe bound (fp32 MFMA), a one-frame streaming call is weight-stream bound."""
    H, D, A, F = cfg.hidden_size, cfg.codebook_dim, cfg.num_attention_heads * cfg.head_dim, cfg.intermediate_size
    gemms = [(T, H, 2 * D)]                                     # (rows, N, K): RVQ output projections
    L1 = T * cfg.upsample_stride
    for _ in range(cfg.num_hidden_layers):
        gemms += [(L1, 3 * A, H), (L1, H, A), (L1, F, H), (L1, H, F)]
    attn = 0
    for p in range(L1):                                         # QK^T and PV over the causal sliding window
        attn += 2 * 2 * A * min(p + 1, cfg.sliding_window)
    attn *= cfg.num_hidden_layers
    ch, L = cfg.num_filters << len(cfg.upsampling_ratios), L1
    gemms.append((L, ch, cfg.kernel_size * H))
    for r in cfg.upsampling_ratios:
        co, hid = ch // 2, ch // 2 // cfg.compress
        gemms.append((L, r * co, 2 * ch))
        L *= r
        gemms += [(L, hid, cfg.residual_kernel_size * co), (L, co, hid)]
        ch = co
    last = 2 * L * ch * cfg.last_kernel_size
    flops = sum(2 * r * n * k for r, n, k in gemms) + attn + last
    wbytes = sum(n * k * 4 for _, n, k in gemms) + ch * cfg.last_kernel_size * 4
    return flops, wbytes


def mimi_state_dict_spec(cfg: MimiDecodeConfig):
    """(key, shape, kind) of every tensor the decode path reads, in the `kyutai/mimi` (transformers) key layout."""
    H, D = cfg.hidden_size, cfg.codebook_dim
    for name, n in (("semantic", cfg.num_semantic_quantizers), ("acoustic", cfg.num_quantizers - cfg.num_semantic_quantizers)):
        p = f"quantizer.{name}_residual_vector_quantizer"
        for i in range(n):
            yield f"{p}.layers.{i}.codebook.embed_sum", (cfg.codebook_size, D), "embed"
            yield f"{p}.layers.{i}.codebook.cluster_usage", (cfg.codebook_size,), "usage"
        yield f"{p}.output_proj.weight", (H, D, 1), "conv"
    yield "upsample.conv.weight", (H, 1, 2 * cfg.upsample_stride), "conv"
    A = cfg.num_attention_heads * cfg.head_dim
    for l in range(cfg.num_hidden_layers):
        p = f"decoder_transformer.layers.{l}"
        for n_, shp in (("self_attn.q_proj", (A, H)), ("self_attn.k_proj", (A, H)), ("self_attn.v_proj", (A, H)),
                        ("self_attn.o_proj", (H, A)), ("mlp.fc1", (cfg.intermediate_size, H)), ("mlp.fc2", (H, cfg.intermediate_size))):
            yield f"{p}.{n_}.weight", shp, "linear"
        for n_ in ("input_layernorm", "post_attention_layernorm"):
            yield f"{p}.{n_}.weight", (H,), "norm"
            yield f"{p}.{n_}.bias", (H,), "bias"
        yield f"{p}.self_attn_layer_scale.scale", (H,), "scale"
        yield f"{p}.mlp_layer_scale.scale", (H,), "scale"
    ch = cfg.num_filters * 2 ** len(cfg.upsampling_ratios)
    yield "decoder.layers.0.conv.weight", (ch, H, cfg.kernel_size), "conv"
    yield "decoder.layers.0.conv.bias", (ch,), "bias"
    idx = 1
    for r in cfg.upsampling_ratios:
        yield f"decoder.layers.{idx + 1}.conv.weight", (ch, ch // 2, 2 * r), "convT"      # [C_in, C_out, k]
        yield f"decoder.layers.{idx + 1}.conv.bias", (ch // 2,), "bias"
        hid = ch // 2 // cfg.compress
        yield f"decoder.layers.{idx + 2}.block.1.conv.weight", (hid, ch // 2, cfg.residual_kernel_size), "conv"
        yield f"decoder.layers.{idx + 2}.block.1.conv.bias", (hid,), "bias"
        yield f"decoder.layers.{idx + 2}.block.3.conv.weight", (ch // 2, hid, 1), "conv"
        yield f"decoder.layers.{idx + 2}.block.3.conv.bias", (ch // 2,), "bias"
        ch //= 2
        idx += 3
    yield f"decoder.layers.{idx + 1}.conv.weight", (1, ch, cfg.last_kernel_size), "conv"
    yield f"decoder.layers.{idx + 1}.conv.bias", (1,), "bias"


def synth_mimi_state_dict(cfg: MimiDecodeConfig, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """Seeded synthetic decode-path checkpoint (hash-based like csm_hf_amd.synth: identical on every machine)."""
    from .synth import synth_tensor, hash_uniform
    sd = {}
    for key, shape, kind in mimi_state_dict_spec(cfg):
        if kind == "usage":
            sd[key] = hash_uniform("mimi:" + key, shape[0], seed, device) * 0.5 + 1.0          # in (0.5, 1.5)
        elif kind == "norm":
            sd[key] = synth_tensor("mimi:" + key, shape, 0.05, seed, device, mean=1.0)
        elif kind == "bias":
            sd[key] = synth_tensor("mimi:" + key, shape, 0.02, seed, device)
        elif kind == "scale":
            sd[key] = synth_tensor("mimi:" + key, shape, 0.05, seed, device, mean=0.3)
        elif kind == "embed":
            sd[key] = synth_tensor("mimi:" + key, shape, 0.5, seed, device)
        else:   # conv [C_out, C_in, k], convT [C_in, C_out, k], linear [N, K]: keep activations O(1) through the stack
            fan = shape[1] * (shape[2] if len(shape) == 3 else 1)
            if kind == "convT":
                fan = shape[0] * 2                                                          # two taps reach an output
            sd[key] = synth_tensor("mimi:" + key, shape, 1.0 / math.sqrt(max(fan, 1)), seed, device)
    return sd


# ---------------------------------------------------------------------------------------------------------------------
# the device path: ctypes binding of csm_mimi_* (include/csm_hip.h) and the one-time weight repacking
# ---------------------------------------------------------------------------------------------------------------------
_MAX_LAYERS, _MAX_RATIOS = 16, 8


class _MimiCfg(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_q", C.c_int32), ("n_sem", C.c_int32), ("codebook_size", C.c_int32),
                ("codebook_dim", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32),
                ("head_dim", C.c_int32), ("ffn", C.c_int32), ("window", C.c_int32), ("rope_theta", C.c_float),
                ("norm_eps", C.c_float), ("n_ratios", C.c_int32), ("ratios", C.c_int32 * _MAX_RATIOS),
                ("num_filters", C.c_int32), ("kernel_size", C.c_int32), ("last_kernel_size", C.c_int32),
                ("res_kernel_size", C.c_int32), ("compress", C.c_int32), ("up_stride", C.c_int32), ("max_frames", C.c_int32)]


_PL, _PR = C.c_void_p * _MAX_LAYERS, C.c_void_p * _MAX_RATIOS


class _MimiWeights(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("out_proj", C.c_void_p), ("upsample", C.c_void_p),
                ("ln1_w", _PL), ("ln1_b", _PL), ("wqkv", _PL), ("wo", _PL), ("ls1", _PL), ("ln2_w", _PL), ("ln2_b", _PL),
                ("w1", _PL), ("w2", _PL), ("ls2", _PL), ("conv0_w", C.c_void_p), ("conv0_b", C.c_void_p),
                ("up_w", _PR), ("up_b", _PR), ("res1_w", _PR), ("res1_b", _PR), ("res2_w", _PR), ("res2_b", _PR),
                ("last_w", C.c_void_p), ("last_b", C.c_void_p)]


def _pad_rows(w: torch.Tensor, mult: int = 128) -> torch.Tensor:
    n = (w.shape[0] + mult - 1) // mult * mult
    if n == w.shape[0]:
        return w.contiguous()
    out = torch.zeros(n, w.shape[1], dtype=w.dtype, device=w.device)
    out[: w.shape[0]] = w
    return out


def _conv_as_gemm(w: torch.Tensor) -> torch.Tensor:
    """conv1d weight [C_out, C_in, k] -> [pad128(C_out), k * C_in] with W'[co][j * C_in + ci] = w[co][ci][j]: the operand
    row of output position l is the k consecutive channels-last input rows l - (k - 1) .. l."""
    return _pad_rows(w.permute(0, 2, 1).reshape(w.shape[0], -1))


def _convtr_as_gemm(w: torch.Tensor, r: int) -> torch.Tensor:
    """conv_transpose1d weight [C_in, C_out, 2r], stride r, causal trim: [r * C_out, 2 * C_in], row s * C_out + co =
    [w[:, co, s + r] | w[:, co, s]] against the operand row [x[q - 1] ; x[q]] -> output position r q + s."""
    ci, co, k = w.shape
    assert k == 2 * r
    rows = [torch.cat([w[:, :, s + r].t(), w[:, :, s].t()], dim=1) for s in range(r)]        # each [C_out, 2 C_in]
    return torch.cat(rows, dim=0).contiguous()


def pack_mimi_weights(cfg: MimiDecodeConfig, sd: Dict[str, torch.Tensor], device) -> Dict[str, object]:
    """HF-layout (`kyutai/mimi`) decode-path tensors -> the fp32 device tensors csm_mimi_weights_t points at."""
    f = lambda k: sd[k].to(device=device, dtype=torch.float32)
    out: Dict[str, object] = {}
    emb, proj = [], []
    for name, n in (("semantic", cfg.num_semantic_quantizers), ("acoustic", cfg.num_quantizers - cfg.num_semantic_quantizers)):
        p = f"quantizer.{name}_residual_vector_quantizer"
        for i in range(n):
            emb.append(f(f"{p}.layers.{i}.codebook.embed_sum") / f(f"{p}.layers.{i}.codebook.cluster_usage").clamp(min=1e-5)[:, None])
        proj.append(f(f"{p}.output_proj.weight")[:, :, 0])
    out["embed"] = torch.stack(emb).contiguous()
    out["out_proj"] = torch.cat(proj, dim=1).contiguous()
    out["upsample"] = f("upsample.conv.weight")[:, 0, :].contiguous()
    for key in ("ln1_w", "ln1_b", "wqkv", "wo", "ls1", "ln2_w", "ln2_b", "w1", "w2", "ls2"):
        out[key] = []
    for l in range(cfg.num_hidden_layers):
        p = f"decoder_transformer.layers.{l}"
        out["ln1_w"].append(f(f"{p}.input_layernorm.weight").contiguous())
        out["ln1_b"].append(f(f"{p}.input_layernorm.bias").contiguous())
        out["wqkv"].append(torch.cat([f(f"{p}.self_attn.{n}_proj.weight") for n in "qkv"], dim=0).contiguous())
        out["wo"].append(f(f"{p}.self_attn.o_proj.weight").contiguous())
        out["ls1"].append(f(f"{p}.self_attn_layer_scale.scale").contiguous())
        out["ln2_w"].append(f(f"{p}.post_attention_layernorm.weight").contiguous())
        out["ln2_b"].append(f(f"{p}.post_attention_layernorm.bias").contiguous())
        out["w1"].append(f(f"{p}.mlp.fc1.weight").contiguous())
        out["w2"].append(f(f"{p}.mlp.fc2.weight").contiguous())
        out["ls2"].append(f(f"{p}.mlp_layer_scale.scale").contiguous())
    out["conv0_w"] = _conv_as_gemm(f("decoder.layers.0.conv.weight"))
    out["conv0_b"] = f("decoder.layers.0.conv.bias").contiguous()
    for key in ("up_w", "up_b", "res1_w", "res1_b", "res2_w", "res2_b"):
        out[key] = []
    idx = 1
    for r in cfg.upsampling_ratios:
        out["up_w"].append(_convtr_as_gemm(f(f"decoder.layers.{idx + 1}.conv.weight"), r))
        out["up_b"].append(f(f"decoder.layers.{idx + 1}.conv.bias").contiguous())
        out["res1_w"].append(_conv_as_gemm(f(f"decoder.layers.{idx + 2}.block.1.conv.weight")))
        out["res1_b"].append(f(f"decoder.layers.{idx + 2}.block.1.conv.bias").contiguous())
        out["res2_w"].append(_conv_as_gemm(f(f"decoder.layers.{idx + 2}.block.3.conv.weight")))
        out["res2_b"].append(f(f"decoder.layers.{idx + 2}.block.3.conv.bias").contiguous())
        idx += 3
    lw = f(f"decoder.layers.{idx + 1}.conv.weight")                      # [1, C, k]
    out["last_w"] = lw[0].t().reshape(-1).contiguous()                    # w'[j * C + c]
    out["last_b"] = f(f"decoder.layers.{idx + 1}.conv.bias").contiguous()
    return out


def load_mimi_checkpoint(path: str):
    """(MimiDecodeConfig, decode-path state dict) from a `kyutai/mimi`-layout directory (`config.json` +
    `model.safetensors`, what `transformers.MimiModel.save_pretrained` writes).  Weight-normalised convolutions stored as
    `...conv.parametrizations.weight.original0/1` (g, v) are folded to plain weights (w = g * v / ||v||, norm over all
    dimensions but the first, torch.nn.utils.parametrizations.weight_norm's default)."""
    import json
    import os
    from safetensors.torch import load_file
    hf = json.load(open(os.path.join(path, "config.json")))
    cfg = MimiDecodeConfig()
    for mine, theirs in (("num_quantizers", "num_quantizers"), ("num_semantic_quantizers", "num_semantic_quantizers"),
                         ("codebook_size", "codebook_size"), ("codebook_dim", "codebook_dim"), ("hidden_size", "hidden_size"),
                         ("num_hidden_layers", "num_hidden_layers"), ("num_attention_heads", "num_attention_heads"),
                         ("head_dim", "head_dim"), ("intermediate_size", "intermediate_size"), ("sliding_window", "sliding_window"),
                         ("norm_eps", "norm_eps"), ("upsampling_ratios", "upsampling_ratios"), ("num_filters", "num_filters"),
                         ("kernel_size", "kernel_size"), ("last_kernel_size", "last_kernel_size"),
                         ("residual_kernel_size", "residual_kernel_size"), ("compress", "compress")):
        if hf.get(theirs) is not None:
            setattr(cfg, mine, hf[theirs])
    rp = hf.get("rope_parameters") or {}
    cfg.rope_theta = float(rp.get("rope_theta", hf.get("rope_theta", cfg.rope_theta)))
    if hf.get("frame_rate") and hf.get("sampling_rate"):
        total = round(hf["sampling_rate"] / hf["frame_rate"])
        cfg.upsample_stride = total // math.prod(cfg.upsampling_ratios)
    for flag, want in (("use_causal_conv", True), ("use_conv_shortcut", False), ("num_residual_layers", 1), ("trim_right_ratio", 1.0)):
        if hf.get(flag, want) != want:
            raise ValueError(f"Mimi config {flag}={hf[flag]!r} is outside what the device path implements ({want!r})")
    if hf.get("num_key_value_heads", cfg.num_attention_heads) != cfg.num_attention_heads:
        raise ValueError("grouped-query attention in the Mimi transformer is not implemented")
    raw = load_file(os.path.join(path, "model.safetensors"))
    sd = {}
    for key, shape, _ in mimi_state_dict_spec(cfg):
        if key in raw:
            t = raw[key]
        elif key.endswith("conv.weight") and key[:-len("weight")] + "parametrizations.weight.original1" in raw:
            base = key[:-len("weight")] + "parametrizations.weight.original"
            g, v = raw[base + "0"].float(), raw[base + "1"].float()
            t = g * v / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
        else:
            raise KeyError(f"{key} missing from the checkpoint")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{key}: shape {tuple(t.shape)} != {tuple(shape)}")
        sd[key] = t.float()
    return cfg, sd


class MimiDecoder:
    """`decode(audio_codes [B, n_q, T]) -> waveform [B, 1, T * samples_per_frame]` like transformers' MimiModel.decode /
    the reference's `audio_tokenizer.decode` (README.md:114-118: `audio_tokenizer.decode(gen_frames.permute(0, 2, 1))`).
    `state_dict`: the decode-path tensors in the `kyutai/mimi` key layout (see mimi_state_dict_spec)."""

    def __init__(self, cfg: MimiDecodeConfig, state_dict: Optional[Dict[str, torch.Tensor]], device="cuda:0", max_frames: int = 512,
                 _packed: Optional[Dict[str, object]] = None):
        from .engine import load_library, ABI_VERSION, _ck
        self.cfg, self.device = cfg, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MimiDecoder runs on an AMD GPU only: csm_hf_amd has no CPU path")
        self.lib = load_library()
        self._ck = _ck
        self.lib.csm_mimi_create.argtypes = [C.POINTER(_MimiCfg), C.POINTER(C.c_void_p)]
        self.lib.csm_mimi_destroy.argtypes = [C.c_void_p]
        self.lib.csm_mimi_bind_weights.argtypes = [C.c_void_p, C.POINTER(_MimiWeights)]
        self.lib.csm_mimi_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.lib.csm_mimi_stream_reset.argtypes = [C.c_void_p]
        self.lib.csm_mimi_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        self.lib.csm_mimi_stream_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self.lib.csm_mimi_streams_open.argtypes = [C.c_void_p, C.c_int]
        self.lib.csm_mimi_streams_reset.argtypes = [C.c_void_p, C.c_int]
        self.lib.csm_mimi_streams_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self.n_streams = 0
        if cfg.num_hidden_layers > _MAX_LAYERS or len(cfg.upsampling_ratios) > _MAX_RATIOS:
            raise ValueError("too many transformer layers / upsampling ratios for csm_mimi_config_t")
        c = _MimiCfg(abi_version=ABI_VERSION, n_q=cfg.num_quantizers, n_sem=cfg.num_semantic_quantizers,
                     codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim, hidden=cfg.hidden_size,
                     layers=cfg.num_hidden_layers, heads=cfg.num_attention_heads, head_dim=cfg.head_dim, ffn=cfg.intermediate_size,
                     window=cfg.sliding_window, rope_theta=cfg.rope_theta, norm_eps=cfg.norm_eps, n_ratios=len(cfg.upsampling_ratios),
                     num_filters=cfg.num_filters, kernel_size=cfg.kernel_size, last_kernel_size=cfg.last_kernel_size,
                     res_kernel_size=cfg.residual_kernel_size, compress=cfg.compress, up_stride=cfg.upsample_stride,
                     max_frames=int(max_frames))
        for i, r in enumerate(cfg.upsampling_ratios):
            c.ratios[i] = int(r)
        self.max_frames = int(max_frames)
        with torch.cuda.device(self.device):
            h = C.c_void_p()
            _ck(self.lib, self.lib.csm_mimi_create(C.byref(c), C.byref(h)))
            self._h = h
            # the packed tensors stay alive with the object; `new_stream()` hands the SAME tensors to another handle
            self.packed = _packed if _packed is not None else pack_mimi_weights(cfg, state_dict, self.device)
            w = _MimiWeights()
            for name, _ in _MimiWeights._fields_:
                v = self.packed[name]
                if isinstance(v, list):
                    arr = getattr(w, name)
                    for i, t in enumerate(v):
                        arr[i] = t.data_ptr()
                else:
                    setattr(w, name, v.data_ptr())
            torch.cuda.synchronize(self.device)
            _ck(self.lib, self.lib.csm_mimi_bind_weights(self._h, C.byref(w)))

    def set_option(self, name: str, value: int):
        """`skinny_rows` (GEMMs of <= n rows on the weight-streaming skinny GEMM, default 16; 0 = none) / `splitk` (0 / 1)."""
        self._ck(self.lib, self.lib.csm_mimi_set_option(self._h, name.encode(), int(value)))

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", max_frames: int = 512) -> "MimiDecoder":
        cfg, sd = load_mimi_checkpoint(path)
        return cls(cfg, sd, device, max_frames)

    def decode(self, audio_codes: torch.Tensor) -> torch.Tensor:
        if audio_codes.dim() != 3 or audio_codes.shape[1] != self.cfg.num_quantizers:
            raise ValueError(f"audio_codes must be [B, {self.cfg.num_quantizers}, T], got {tuple(audio_codes.shape)}")
        B, _, T = audio_codes.shape
        if T < 1 or T > self.max_frames:
            raise ValueError(f"T = {T} outside 1..{self.max_frames} (max_frames)")
        if int(audio_codes.min()) < 0 or int(audio_codes.max()) >= self.cfg.codebook_size:
            raise ValueError("audio codes outside the codebook")
        codes = audio_codes.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty(B, 1, T * self.cfg.samples_per_frame, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib, self.lib.csm_mimi_decode(self._h, codes.data_ptr(), B, T, out.data_ptr()))
        return out

    # ---- streaming: one sequence, a few frames per call (e.g. every frame `generate_frame` returns) -----------------
    def new_stream(self, max_frames: Optional[int] = None) -> "MimiDecoder":
        """Another decoder on the SAME device weights (no second copy of the 79 M parameters): its own handle, scratch and
        stream state, so the rows of a generated batch can be streamed side by side -- one `stream_decode` per row and frame
        (0.8 ms each at the kyutai shape); the handles do not share anything mutable."""
        return MimiDecoder(self.cfg, None, self.device, self.max_frames if max_frames is None else max_frames, _packed=self.packed)

    def stream_reset(self):
        self._ck(self.lib, self.lib.csm_mimi_stream_reset(self._h))

    def stream_decode(self, audio_codes: torch.Tensor) -> torch.Tensor:
        """`audio_codes` [n_q, T] or [1, n_q, T]: the NEW frames of the stream; returns their samples `[1, 1, T * samples_per_frame]`.
        The chunks of a stream concatenate to `decode()` of the whole sequence (the handle keeps the transformer's K/V window
        and every convolution's left context, like transformers' decoder_past_key_values + padding cache)."""
        if audio_codes.dim() == 3:
            if audio_codes.shape[0] != 1:
                raise ValueError("a stream is one sequence")
            audio_codes = audio_codes[0]
        if audio_codes.dim() != 2 or audio_codes.shape[0] != self.cfg.num_quantizers:
            raise ValueError(f"audio_codes must be [{self.cfg.num_quantizers}, T]")
        T = audio_codes.shape[1]
        if T < 1 or T > self.max_frames:
            raise ValueError(f"T = {T} outside 1..{self.max_frames} (max_frames)")
        codes = audio_codes.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty(1, 1, T * self.cfg.samples_per_frame, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib, self.lib.csm_mimi_stream_decode(self._h, codes.data_ptr(), T, out.data_ptr()))
        return out

    # ---- stream groups: the rows of a generated batch streamed together, every launch covering all of them --------------
    def streams_open(self, n_streams: int):
        """Allocate the state of `n_streams` lockstep streams (replaces an earlier group); a call then takes up to
        `max_frames // n_streams` frames per stream."""
        with torch.cuda.device(self.device):
            self._ck(self.lib, self.lib.csm_mimi_streams_open(self._h, int(n_streams)))
        self.n_streams = int(n_streams)

    def streams_reset(self, stream: Optional[int] = None):
        """Restart one stream of the group (a batch row taken over by a new utterance), or all of them (`None`)."""
        self._ck(self.lib, self.lib.csm_mimi_streams_reset(self._h, -1 if stream is None else int(stream)))

    def streams_decode(self, audio_codes: torch.Tensor) -> torch.Tensor:
        """`audio_codes` [S, n_q, T]: the NEW frames of every stream of the group (e.g. `frames.permute(0, 2, 1)` of the frames one
        `generate_frame` / one chunk of `generate` produced for the batch); returns `[S, 1, T * samples_per_frame]`.  One pass over
        the codec for all S streams: a one-frame call for 16 / 64 streams costs about what one stream costs alone."""
        if self.n_streams < 1:
            raise RuntimeError("no stream group: call streams_open(n_streams) first")
        if audio_codes.dim() != 3 or audio_codes.shape[0] != self.n_streams or audio_codes.shape[1] != self.cfg.num_quantizers:
            raise ValueError(f"audio_codes must be [{self.n_streams}, {self.cfg.num_quantizers}, T], got {tuple(audio_codes.shape)}")
        S, _, T = audio_codes.shape
        if T < 1 or S * T > self.max_frames:
            raise ValueError(f"T = {T} outside 1..{self.max_frames // S} (max_frames // n_streams)")
        if int(audio_codes.min()) < 0 or int(audio_codes.max()) >= self.cfg.codebook_size:
            raise ValueError("audio codes outside the codebook")
        codes = audio_codes.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty(S, 1, T * self.cfg.samples_per_frame, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        with torch.cuda.device(self.device):
            self._ck(self.lib, self.lib.csm_mimi_streams_decode(self._h, codes.data_ptr(), T, out.data_ptr()))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.lib.csm_mimi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
