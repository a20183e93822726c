"""ctypes binding of libcsm_hip.so (include/csm_hip.h) and the host-side engine wrapper.

torch is used here for device memory (weights, outputs) only; every computation of the path is a HIP
kernel behind the C ABI.  There is NO CPU fallback: if the library cannot be loaded, or no GPU is
present, constructing an `Engine` raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional

import torch

from .configuration_csm import CSMConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcsm_hip.so")
ABI_VERSION = 7
DT_F32, DT_BF16, DT_FP8 = 0, 1, 2

EXPORTS = [
    "csm_engine_create", "csm_engine_destroy", "csm_bind_weights", "csm_build_proj_table", "csm_set_proj_table",
    "csm_reset", "csm_set_option", "csm_prefill", "csm_decode_frame", "csm_backbone_step", "csm_backbone_step_ids",
    "csm_get_state", "csm_generate", "csm_read_frames", "csm_frames_done", "csm_cur_len", "csm_set_kv_start",
    "csm_last_generate_ms", "csm_embed_sum", "csm_rmsnorm", "csm_gemv", "csm_gemm", "csm_sample_topk",
    "csm_attn_decode", "csm_rope_scatter", "csm_bench_gemv", "csm_sync", "csm_last_error", "csm_abi_version",
    "csm_rewind_frames", "csm_graph_stats", "csm_kv_copy", "csm_prefetch_stats", "csm_prefetch_health",
    "csm_set_debug_buffer", "csm_last_geoms", "csm_read_zero_counts",
    "csm_prefill_pos", "csm_kv_export", "csm_kv_import", "csm_set_length", "csm_forward_loss", "csm_prefill_slot", "csm_prefill_slots",
    "csm_mimi_create", "csm_mimi_destroy", "csm_mimi_bind_weights", "csm_mimi_decode", "csm_mimi_stream_reset",
    "csm_mimi_stream_decode", "csm_mimi_set_option", "csm_shift_context",
    "csm_mimi_streams_open", "csm_mimi_streams_reset", "csm_mimi_streams_decode",
    "csm_bind_mx_weights", "csm_mx_quantize", "csm_gemm_mx", "csm_forward_backward", "csm_gemm_bf16",
]


class LlamaCfg(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("ffn", C.c_int32), ("layers", C.c_int32), ("n_q", C.c_int32),
                ("n_kv", C.c_int32), ("head_dim", C.c_int32), ("rms_eps", C.c_float)]


class EngineCfg(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("text_vocab", C.c_int32), ("audio_vocab", C.c_int32),
                ("n_codebooks", C.c_int32), ("backbone", LlamaCfg), ("decoder", LlamaCfg),
                ("weight_dtype", C.c_int32), ("kv_dtype", C.c_int32), ("max_batch", C.c_int32),
                ("max_len", C.c_int32), ("max_frames", C.c_int32), ("max_prefill_rows", C.c_int32)]


class LayerW(C.Structure):
    _fields_ = [("wqkv", C.c_void_p), ("wo", C.c_void_p), ("wgu", C.c_void_p), ("wd", C.c_void_p),
                ("ln1", C.c_void_p), ("ln2", C.c_void_p),
                ("sqkv", C.c_void_p), ("so", C.c_void_p), ("sgu", C.c_void_p), ("sd", C.c_void_p)]


class StackW(C.Structure):
    _fields_ = [("layers", C.POINTER(LayerW)), ("final_norm", C.c_void_p), ("rope_cos", C.c_void_p),
                ("rope_sin", C.c_void_p), ("rope_positions", C.c_int32)]


class Weights(C.Structure):
    _fields_ = [("backbone", StackW), ("decoder", StackW), ("text_emb", C.c_void_p), ("audio_emb", C.c_void_p),
                ("proj_head0", C.c_void_p), ("audio_head_t", C.c_void_p), ("proj_table", C.c_void_p),
                ("s_proj_head0", C.c_void_p), ("s_audio_head", C.c_void_p)]


class LayerGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dwqkv", "dwo", "dwgu", "dwd", "dln1", "dln2")]


class StackGrads(C.Structure):
    _fields_ = [("layers", C.POINTER(LayerGrads)), ("final_norm", C.c_void_p)]


class Grads(C.Structure):
    _fields_ = [("backbone", StackGrads), ("decoder", StackGrads), ("text_emb", C.c_void_p), ("audio_emb", C.c_void_p),
                ("proj_head0", C.c_void_p), ("audio_head_t", C.c_void_p)]


class MxLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv", "qkv_s", "o", "o_s", "gu", "gu_s", "d", "d_s")]


class Sampling(C.Structure):
    _fields_ = [("temperature", C.c_float), ("topk", C.c_int32), ("seed", C.c_uint64), ("noise", C.c_void_p),
                ("forced", C.c_void_p), ("logits_trace", C.c_void_p), ("last_h_trace", C.c_void_p),
                ("row_offset", C.c_int32), ("per_row_stop", C.c_int32)]


_lib = None


def load_library(path: Optional[str] = None):
    """Load libcsm_hip.so and declare its prototypes.  Raises if missing -- there is no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("CSM_HIP_LIB") or LIB_PATH   # CSM_HIP_LIB: A/B builds of the same ABI
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build it with `python -m csm_hf_amd.build` (hipcc, gfx950); "
                           "csm_hf_amd has no CPU fallback")
    lib = C.CDLL(p)
    lib.csm_last_error.restype = C.c_char_p
    for name in EXPORTS:
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        if name != "csm_last_error":
            fn.restype = C.c_int
    if lib.csm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libcsm_hip.so ABI {lib.csm_abi_version()} != binding {ABI_VERSION}")
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    lib.csm_engine_create.argtypes = [C.POINTER(EngineCfg), i32, vp, C.POINTER(vp)]
    lib.csm_engine_destroy.argtypes = [vp]
    lib.csm_bind_weights.argtypes = [vp, C.POINTER(Weights)]
    lib.csm_build_proj_table.argtypes = [vp, vp]
    lib.csm_set_proj_table.argtypes = [vp, vp]
    lib.csm_reset.argtypes = [vp]
    lib.csm_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.csm_prefill.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.csm_decode_frame.argtypes = [vp, C.POINTER(Sampling)]
    lib.csm_backbone_step.argtypes = [vp, C.POINTER(Sampling)]
    lib.csm_backbone_step_ids.argtypes = [vp, vp, vp, i32, i32]
    lib.csm_get_state.argtypes = [vp, vp, vp]
    lib.csm_generate.argtypes = [vp, C.POINTER(Sampling), i32, i32]
    lib.csm_read_frames.argtypes = [vp, vp, i32, i32]
    lib.csm_frames_done.argtypes = [vp, C.POINTER(i32)]
    lib.csm_cur_len.argtypes = [vp, C.POINTER(i32)]
    lib.csm_set_kv_start.argtypes = [vp, C.POINTER(C.c_int32), i32]
    lib.csm_last_generate_ms.argtypes = [vp, C.POINTER(f32)]
    lib.csm_embed_sum.argtypes = [vp, vp, vp, i32, vp]
    lib.csm_rmsnorm.argtypes = [vp, vp, vp, i32, i32, f32, vp]
    lib.csm_gemv.argtypes = [vp, vp, i32, vp, i32, i32, vp, i32, vp, f32, vp]
    lib.csm_gemm.argtypes = [vp, vp, i32, vp, i32, i32, vp, i32, vp]
    lib.csm_sample_topk.argtypes = [vp, vp, i32, i32, f32, i32, C.c_uint64, vp, vp]
    lib.csm_attn_decode.argtypes = [vp, i32, i32, vp, vp, vp, i32, i32, vp]
    lib.csm_rope_scatter.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp]
    lib.csm_bench_gemv.argtypes = [vp, vp, C.c_size_t, i32, i32, i32, i32, vp, i32, vp, f32, vp, i32, i32, i32, i32,
                                   C.POINTER(f32), i32, i32, i32]
    lib.csm_sync.argtypes = [vp]
    lib.csm_rewind_frames.argtypes = [vp]
    lib.csm_graph_stats.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.csm_kv_copy.argtypes = [vp, vp]
    lib.csm_prefetch_stats.argtypes = [vp, C.POINTER(C.c_longlong)]
    lib.csm_prefetch_health.argtypes = [vp, C.POINTER(C.c_longlong)]
    lib.csm_read_zero_counts.argtypes = [vp, C.POINTER(C.c_int32), i32, i32]
    lib.csm_prefill_pos.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    lib.csm_forward_loss.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp]
    lib.csm_prefill_slot.argtypes = [vp, i32, vp, vp, i32]
    lib.csm_prefill_slots.argtypes = [vp, vp, vp, i32, vp, vp, i32]
    lib.csm_shift_context.argtypes = [vp, i32]
    lib.csm_bind_mx_weights.argtypes = [vp, C.POINTER(MxLayer), i32]
    lib.csm_mx_quantize.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.csm_gemm_mx.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, vp]
    lib.csm_gemm_bf16.argtypes = [vp, vp, i32, i32, vp, i32, vp, i32]
    lib.csm_forward_backward.argtypes = [vp, vp, vp, vp, i32, i32, vp, C.POINTER(Grads)]
    lib.csm_kv_export.argtypes = [vp, i32, vp, vp, i32]
    lib.csm_kv_import.argtypes = [vp, i32, vp, vp, i32, i32]
    lib.csm_set_length.argtypes = [vp, i32, i32]
    lib.csm_set_debug_buffer.argtypes = [vp, vp, i32]
    lib.csm_last_geoms.argtypes = [vp, C.POINTER(C.c_int32), i32, C.POINTER(i32)]
    if path is None:
        _lib = lib
    return lib


def _ck(lib, rc: int):
    if rc != 0:
        msg = lib.csm_last_error().decode(errors="replace")
        if rc == -3:
            raise ValueError(f"csm_hip capacity error: {msg}")
        raise RuntimeError(f"csm_hip error {rc}: {msg}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# ---- llama3 RoPE table (published algorithm: Llama-3.1 rope scaling; parameters from the reference
#      modeling_csm.py:78-85,99-106).  Computed on the CPU in fp32 exactly like the reference does
#      (angle = pos * inv_freq in fp32, then cos/sin), uploaded once. ------------------------------------
def llama3_inv_freq(head_dim: int, base: float, scaling: Optional[dict]) -> torch.Tensor:
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.int64).to(torch.float32) / head_dim))
    if not scaling or scaling.get("type", scaling.get("rope_type", "default")) in (None, "default"):
        return inv
    if scaling.get("type", scaling.get("rope_type")) != "llama3":
        raise ValueError(f"unsupported rope scaling {scaling}")
    factor, lo, hi = scaling["factor"], scaling["low_freq_factor"], scaling["high_freq_factor"]
    old = scaling["original_max_position_embeddings"]
    wavelen = 2 * math.pi / inv
    scaled = torch.where(wavelen > old / lo, inv / factor, inv)
    smooth = (old / wavelen - lo) / (hi - lo)
    mid = (1 - smooth) * scaled / factor + smooth * scaled
    is_mid = ~(wavelen < old / hi) * ~(wavelen > old / lo)
    return torch.where(is_mid, mid, scaled)


def rope_tables(lc, n_pos: int):
    inv = llama3_inv_freq(lc.head_dim, lc.rope_theta, lc.rope_scaling)
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return ang.cos().contiguous(), ang.sin().contiguous()


FP8_MAX = 448.0   # OCP e4m3fn


def quantize_fp8_rows(w: torch.Tensor):
    """Per-output-row symmetric e4m3fn quantisation: w[n,:] ~= q[n,:] * s[n].  Returns (uint8 bytes, fp32 scales)."""
    w32 = w.float()
    s = (w32.abs().amax(dim=-1, keepdim=True) / FP8_MAX).clamp_min(1e-12)
    q = (w32 / s).clamp_(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), s.squeeze(-1).contiguous()


def dequantize_fp8_rows(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    return q.view(torch.float8_e4m3fn).float() * s.unsqueeze(-1)


def quantize_mx_rows(w: torch.Tensor):
    """OCP MX-fp8 along the last dimension: blocks of 32, one shared E8M0 scale 2^(floor(log2(amax)) - 8) per block
    (8 = exponent of the largest e4m3 binade, 448 = 1.75 * 2^8), elements = w / scale saturated to +-448, rounded to
    nearest-even e4m3.  Returns (uint8 elements [..., K], uint8 scales [..., K/32], value 2^(byte - 127))."""
    K = w.shape[-1]
    if K % 32:
        raise ValueError("MX quantisation needs K % 32 == 0")
    x = w.float().reshape(*w.shape[:-1], K // 32, 32)
    amax = x.abs().amax(-1, keepdim=True)
    eb = ((amax.view(torch.int32) >> 23) & 0xff) - 8          # biased exponent of amax, minus emax(e4m3)
    eb = eb.clamp(0, 254)
    inv = ((254 - eb) << 23).view(torch.float32)              # 2^-(eb - 127), exact
    q = (x * inv).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(w.shape).contiguous(), eb.to(torch.uint8).reshape(*w.shape[:-1], K // 32).contiguous()


def dequantize_mx_rows(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    K = q.shape[-1]
    v = q.view(torch.float8_e4m3fn).float().reshape(*q.shape[:-1], K // 32, 32)
    sc = torch.exp2(s.float() - 127.0).unsqueeze(-1)
    return (v * sc).reshape(q.shape)


def pack_weights(cfg: CSMConfig, sd: Dict[str, torch.Tensor], device, wdtype: torch.dtype, max_len: int, fp8: bool = False):
    """Reference checkpoint layout -> engine layout (include/csm_hip.h).
    `fp8=True` (BASELINE config 5): the linear matrices are stored as e4m3fn + per-row fp32 scales; embedding
    tables and norm weights keep `wdtype` (bf16)."""
    def mat(t):
        return t.detach().to(device=device, dtype=wdtype).contiguous()

    def qmat(d, key, t):
        if fp8:
            d[key], d["s" + key[1:] if key.startswith("w") else "s_" + key] = quantize_fp8_rows(t)
        else:
            d[key] = t

    def vec(t):
        return t.detach().to(device=device, dtype=torch.float32).contiguous()

    packed = {"_keep": []}
    for prefix, lc, npos in (("backbone", cfg.backbone_config, max_len), ("decoder", cfg.decoder_config,
                                                                          cfg.audio_num_codebooks)):
        layers = []
        for i in range(lc.num_hidden_layers):
            p = f"{prefix}.layers.{i}"
            wqkv = torch.cat([mat(sd[f"{p}.self_attn.q_proj.weight"]), mat(sd[f"{p}.self_attn.k_proj.weight"]),
                              mat(sd[f"{p}.self_attn.v_proj.weight"])], dim=0).contiguous()
            g, u = mat(sd[f"{p}.mlp.gate_proj.weight"]), mat(sd[f"{p}.mlp.up_proj.weight"])
            wgu = torch.stack([g, u], dim=1).reshape(2 * g.shape[0], g.shape[1]).contiguous()
            del g, u
            L = dict(ln1=vec(sd[f"{p}.input_layernorm.weight"]), ln2=vec(sd[f"{p}.post_attention_layernorm.weight"]))
            qmat(L, "wqkv", wqkv)
            qmat(L, "wo", mat(sd[f"{p}.self_attn.o_proj.weight"]))
            qmat(L, "wgu", wgu)
            qmat(L, "wd", mat(sd[f"{p}.mlp.down_proj.weight"]))
            del wqkv, wgu
            layers.append(L)
        cos, sin = rope_tables(lc, npos)
        packed[prefix] = dict(layers=layers, final_norm=vec(sd[f"{prefix}.norm.weight"]), cos=cos.to(device),
                              sin=sin.to(device), npos=npos)
    packed["text_emb"] = mat(sd["text_embeddings.weight"])
    packed["audio_emb"] = mat(sd["audio_embeddings.weight"])
    qmat(packed, "proj_head0", torch.cat([mat(sd["projection.weight"]), mat(sd["codebook0_head.weight"])], 0).contiguous())
    aht = mat(sd["audio_head"]).transpose(1, 2).contiguous()
    if fp8:
        q, sc = quantize_fp8_rows(aht.reshape(-1, aht.shape[-1]))
        packed["audio_head_t"], packed["s_audio_head_t"] = q.view(aht.shape), sc
    else:
        packed["audio_head_t"] = aht
    packed["fp8"] = fp8
    return packed


_LIVE = 0


def live_engines() -> int:
    """engines of this process that hold a native handle (bench.py asserts ONE per process: one process per GPU)"""
    return _LIVE


class Engine:
    """One engine = one GPU, one stream, one resident batch."""

    def __init__(self, cfg: CSMConfig, state_dict: Dict[str, torch.Tensor], device, dtype: torch.dtype,
                 max_batch: int = 1, max_len: int = 2048, max_frames: int = 512, max_prefill_rows: int = 2048,
                 kv_dtype: torch.dtype = torch.float32, packed=None, weight_format: str = "native"):
        if not torch.cuda.is_available():
            raise RuntimeError("csm_hf_amd needs an AMD GPU (gfx950); no CPU fallback exists")
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"unsupported model dtype {dtype}: use torch.float32 or torch.bfloat16")
        if weight_format not in ("native", "fp8"):
            raise ValueError(f"unknown weight_format {weight_format!r}")
        self.fp8 = weight_format == "fp8"
        self.kv_dtype = torch.bfloat16 if kv_dtype == torch.bfloat16 else torch.float32
        if self.fp8 and dtype != torch.bfloat16:
            raise ValueError("weight_format='fp8' needs a bf16 model (embeddings and norms stay bf16)")
        self.lib = load_library()
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = dtype
        self.max_batch, self.max_len, self.max_frames = max_batch, max_len, max_frames
        self.max_prefill_rows = max(max_prefill_rows, 128)
        self.C = cfg.audio_num_codebooks
        self.V = cfg.audio_vocab_size
        self.Hb = cfg.backbone_config.hidden_size
        self.Hd = cfg.decoder_config.hidden_size
        ec = EngineCfg()
        ec.abi_version = ABI_VERSION
        ec.text_vocab, ec.audio_vocab, ec.n_codebooks = cfg.text_vocab_size, self.V, self.C
        for dst, lc in ((ec.backbone, cfg.backbone_config), (ec.decoder, cfg.decoder_config)):
            dst.hidden, dst.ffn, dst.layers = lc.hidden_size, lc.intermediate_size, lc.num_hidden_layers
            dst.n_q, dst.n_kv, dst.head_dim = lc.num_attention_heads, lc.num_key_value_heads, lc.head_dim
            dst.rms_eps = lc.rms_norm_eps
        ec.weight_dtype = DT_FP8 if self.fp8 else (DT_BF16 if dtype == torch.bfloat16 else DT_F32)
        ec.kv_dtype = DT_BF16 if kv_dtype == torch.bfloat16 else DT_F32
        ec.max_batch, ec.max_len, ec.max_frames = max_batch, max_len, max_frames
        ec.max_prefill_rows = self.max_prefill_rows
        self._h = C.c_void_p()
        torch.cuda.set_device(self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_engine_create(C.byref(ec), self.device.index or 0, None, C.byref(self._h)))
        global _LIVE
        _LIVE += 1
        self.packed = packed if packed is not None else pack_weights(cfg, state_dict, self.device, dtype, max_len, fp8=self.fp8)
        if bool(self.packed.get("fp8", False)) != self.fp8:
            raise ValueError("packed weights were built for a different weight_format")
        if self.packed["backbone"]["npos"] < max_len:
            cos, sin = rope_tables(cfg.backbone_config, max_len)
            self.packed["backbone"].update(cos=cos.to(self.device), sin=sin.to(self.device), npos=max_len)
        torch.cuda.current_stream().synchronize()
        self._bind()
        if "proj_table" not in self.packed:
            self.packed["proj_table"] = torch.empty(self.C * self.V, self.Hd, dtype=torch.float32, device=self.device)
            torch.cuda.current_stream().synchronize()
            _ck(self.lib, self.lib.csm_build_proj_table(self._h, _ptr(self.packed["proj_table"])))
        else:
            _ck(self.lib, self.lib.csm_set_proj_table(self._h, _ptr(self.packed["proj_table"])))
        self.batch = 0
        self.length = 0
        self.frames = 0
        self.has_mx = False
        if "mx" in self.packed:      # a re-homed engine keeps the MX copies of its predecessor
            self.enable_mx()

    def _bind(self):
        w = Weights()
        self._layer_arrays = []
        for name, dst in (("backbone", w.backbone), ("decoder", w.decoder)):
            st = self.packed[name]
            arr = (LayerW * len(st["layers"]))()
            for i, l in enumerate(st["layers"]):
                arr[i].wqkv, arr[i].wo, arr[i].wgu, arr[i].wd = (l["wqkv"].data_ptr(), l["wo"].data_ptr(),
                                                                 l["wgu"].data_ptr(), l["wd"].data_ptr())
                arr[i].ln1, arr[i].ln2 = l["ln1"].data_ptr(), l["ln2"].data_ptr()
                if self.fp8:
                    arr[i].sqkv, arr[i].so, arr[i].sgu, arr[i].sd = (l["sqkv"].data_ptr(), l["so"].data_ptr(),
                                                                     l["sgu"].data_ptr(), l["sd"].data_ptr())
            self._layer_arrays.append(arr)
            dst.layers = arr
            dst.final_norm = st["final_norm"].data_ptr()
            dst.rope_cos, dst.rope_sin = st["cos"].data_ptr(), st["sin"].data_ptr()
            dst.rope_positions = st["npos"]
        w.text_emb = self.packed["text_emb"].data_ptr()
        w.audio_emb = self.packed["audio_emb"].data_ptr()
        w.proj_head0 = self.packed["proj_head0"].data_ptr()
        w.audio_head_t = self.packed["audio_head_t"].data_ptr()
        w.proj_table = None
        if self.fp8:
            w.s_proj_head0 = self.packed["s_proj_head0"].data_ptr()
            w.s_audio_head = self.packed["s_audio_head_t"].data_ptr()
        _ck(self.lib, self.lib.csm_bind_weights(self._h, C.byref(w)))

    def enable_mx(self, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        """MX-fp8 copies of the backbone's packed linears (quantised from the checkpoint's own bf16 / fp32 values, not from
        the per-row fp8 copy) for `prefill_mx`: the context prefill on v_mfma_scale_f32_16x16x128_f8f6f4 (csrc/gemm_mx.h)."""
        if "mx" not in self.packed:
            if state_dict is None:
                raise ValueError("enable_mx needs the checkpoint (state_dict) the first time")
            lc = self.cfg.backbone_config
            layers = []
            for i in range(lc.num_hidden_layers):
                p = f"backbone.layers.{i}"
                t = lambda k: state_dict[f"{p}.{k}.weight"].detach().to(self.device, torch.float32)
                g, u = t("mlp.gate_proj"), t("mlp.up_proj")
                mats = dict(qkv=torch.cat([t("self_attn.q_proj"), t("self_attn.k_proj"), t("self_attn.v_proj")], 0),
                            o=t("self_attn.o_proj"), gu=torch.stack([g, u], dim=1).reshape(2 * g.shape[0], g.shape[1]),
                            d=t("mlp.down_proj"))
                del g, u
                L = {}
                for k, w in mats.items():
                    L[k], L[k + "_s"] = quantize_mx_rows(w)
                layers.append(L)
            self.packed["mx"] = layers
        arr = (MxLayer * len(self.packed["mx"]))()
        for i, L in enumerate(self.packed["mx"]):
            for k in ("qkv", "qkv_s", "o", "o_s", "gu", "gu_s", "d", "d_s"):
                setattr(arr[i], k, L[k].data_ptr())
        torch.cuda.synchronize(self.device)
        _ck(self.lib, self.lib.csm_bind_mx_weights(self._h, arr, len(self.packed["mx"])))
        self.has_mx = True

    def close(self):
        global _LIVE
        if getattr(self, "_h", None) and self._h.value:
            self.lib.csm_engine_destroy(self._h)
            self._h = C.c_void_p()
            _LIVE -= 1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ---------------------------------------------------------------------------------------
    def sync(self):
        _ck(self.lib, self.lib.csm_sync(self._h))

    def reset(self):
        _ck(self.lib, self.lib.csm_reset(self._h))
        self.batch = self.length = self.frames = 0

    def set_option(self, name: str, value: int):
        _ck(self.lib, self.lib.csm_set_option(self._h, name.encode(), int(value)))

    def set_kv_start(self, starts):
        arr = (C.c_int32 * len(starts))(*[int(s) for s in starts])
        _ck(self.lib, self.lib.csm_set_kv_start(self._h, arr, len(starts)))

    # ---- the path ---------------------------------------------------------------------------------------
    def _prep_ids(self, ids: torch.Tensor, mask: Optional[torch.Tensor]):
        ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
        m = None
        if mask is not None:
            m = (mask.to(self.device) != 0).to(torch.uint8).contiguous()
        torch.cuda.current_stream().synchronize()
        return ids, m

    def prefill(self, ids: torch.Tensor, mask: Optional[torch.Tensor], want_outputs: bool = True,
                position_ids: Optional[torch.Tensor] = None):
        """ids/mask [B,S,C+1].  Appends S positions; returns (last_h [B,Hb], c0_logits [B,V]) fp32.
        `position_ids` [B,S] (or [1,S]): caller-supplied RoPE positions (the reference forwards them to LlamaModel)."""
        B, S = ids.shape[0], ids.shape[1]
        ids, m = self._prep_ids(ids, mask)
        pos = None
        if position_ids is not None:
            pos = position_ids.to(self.device).expand(B, S).to(torch.int32).contiguous()
            if int(pos.min()) < 0 or int(pos.max()) >= self.packed["backbone"]["npos"]:
                raise ValueError(f"position_ids must lie in [0, {self.packed['backbone']['npos']}) (RoPE table of the engine)")
        lh = torch.empty(B, self.Hb, dtype=torch.float32, device=self.device) if want_outputs else None
        lg = torch.empty(B, self.V, dtype=torch.float32, device=self.device) if want_outputs else None
        torch.cuda.current_stream().synchronize()
        done = 0
        # chunk long contexts so that B*chunk fits the prefill scratch
        chunk = max(1, self.max_prefill_rows // B)
        while done < S:
            n = min(chunk, S - done)
            ci = ids[:, done:done + n].contiguous()
            cm = m[:, done:done + n].contiguous() if m is not None else None
            torch.cuda.current_stream().synchronize()
            if pos is not None:
                cp = pos[:, done:done + n].contiguous()
                torch.cuda.current_stream().synchronize()
                _ck(self.lib, self.lib.csm_prefill_pos(self._h, _ptr(ci), _ptr(cm), B, n, _ptr(cp), _ptr(lh), _ptr(lg)))
            elif n == 1 and self.length > 0:
                _ck(self.lib, self.lib.csm_backbone_step_ids(self._h, _ptr(ci), _ptr(cm), B, 0))
                _ck(self.lib, self.lib.csm_get_state(self._h, _ptr(lh), _ptr(lg)))
            else:
                _ck(self.lib, self.lib.csm_prefill(self._h, _ptr(ci), _ptr(cm), B, n, _ptr(lh), _ptr(lg)))
            self.sync()
            done += n
            self.length += n
        self.batch = B
        return lh, lg

    def prefill_slot(self, row: int, ids: torch.Tensor, mask: Optional[torch.Tensor]):
        """Continuous batching: a new utterance (ids/mask [S,C+1] or [1,S,C+1]) takes over batch row `row` of the running
        batch; its context is placed right-aligned against the current length.  A context LONGER than the batch's current
        length first moves the resident rows up by the difference (`shift_context`; the cache must have the room:
        S <= max_len).  Contexts longer than max_prefill_rows are prefilled in chunks inside the library."""
        if ids.dim() == 2:
            ids = ids.unsqueeze(0)
            mask = None if mask is None else mask.unsqueeze(0)
        S = ids.shape[1]
        if not 0 <= int(row) < max(self.batch, 1):
            raise ValueError(f"row {row} outside the running batch of {self.batch}")
        ids, m = self._prep_ids(ids, mask)          # validates the joining ids BEFORE the resident rows are touched
        if S > self.length:
            self.shift_context(S - self.length)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_prefill_slot(self._h, int(row), _ptr(ids), _ptr(m), S))
        self.sync()

    def prefill_slots(self, rows, ids_list, mask_list) -> bool:
        """Several new utterances take over several batch rows in ONE prefill (csm_prefill_slots): the contexts (each [S_i, C+1]) are
        left-padded to the longest and prefilled together -- the running batch waits for one short prefill instead of len(rows).
        Returns False (nothing done) when that does not fit: a context longer than the batch's current length, or more than
        max_prefill_rows rows in total; the caller then joins them one by one (prefill_slot)."""
        n = len(rows)
        if n == 0:
            return True
        S = max(int(t.shape[0]) for t in ids_list)
        if n < 2 or S > self.length or n * S > self.max_prefill_rows:
            return False
        C1 = ids_list[0].shape[-1]
        ids = torch.zeros(n, S, C1, dtype=torch.long)
        mask = torch.zeros(n, S, C1, dtype=torch.long)
        for i, (ri, rm) in enumerate(zip(ids_list, mask_list)):
            T = ri.shape[0]
            ids[i, S - T:] = ri
            mask[i, S - T:] = rm if rm is not None else 1
        ids_d, m_d = self._prep_ids(ids, mask)
        rows_a = (C.c_int32 * n)(*[int(r) for r in rows])
        lens_a = (C.c_int32 * n)(*[int(t.shape[0]) for t in ids_list])
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_prefill_slots(self._h, rows_a, lens_a, n, _ptr(ids_d), _ptr(m_d), S))
        self.sync()
        return True

    def shift_context(self, delta: int):
        """Move every resident row `delta` cache slots up (csm_shift_context): makes room for a joining context longer than
        the batch's current length.  The shared length grows by delta."""
        _ck(self.lib, self.lib.csm_shift_context(self._h, int(delta)))
        self.length += int(delta)

    def forward_loss(self, ids: torch.Tensor, mask: Optional[torch.Tensor], labels: torch.Tensor):
        """The reference's training forward (modeling_csm.py:367-465), forward only: returns (losses [3] fp32 on the
        device = loss, backbone_loss, decoder_loss; last_h [B,Hb]; c0_logits [B,V]).  Starts from an empty cache and
        leaves the context prefilled; B*S must fit the prefill scratch (max_prefill_rows)."""
        B, S = ids.shape[0], ids.shape[1]
        if B * S > self.max_prefill_rows:
            raise ValueError(f"B*S = {B * S} exceeds max_prefill_rows {self.max_prefill_rows}")
        ids, m = self._prep_ids(ids, mask)
        lab = labels.to(device=self.device, dtype=torch.int64).contiguous()
        if lab.shape != ids.shape:
            raise ValueError(f"labels {tuple(lab.shape)} must match input_ids {tuple(ids.shape)}")
        out = torch.empty(3, dtype=torch.float32, device=self.device)
        lh = torch.empty(B, self.Hb, dtype=torch.float32, device=self.device)
        lg = torch.empty(B, self.V, dtype=torch.float32, device=self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_forward_loss(self._h, _ptr(ids), _ptr(m), _ptr(lab), B, S, _ptr(out), _ptr(lh), _ptr(lg)))
        self.sync()
        self.length += S
        self.batch = B
        return out, lh, lg

    def forward_backward(self, ids: torch.Tensor, mask: Optional[torch.Tensor], labels: torch.Tensor):
        """The reference's training objective AND its gradients (csm_forward_backward; reference consumer train.py:308-326):
        returns (losses [3] fp32 = loss, backbone_loss, decoder_loss; {reference parameter name: fp32 gradient of `loss`}).
        The library accumulates into buffers in the engine's PACKED layout; they are unpacked here into the checkpoint's
        key layout (q/k/v rows of wqkv, de-interleaved gate/up rows of wgu, [projection; codebook0_head], audio_head
        transposed back to [C-1, Hd, V])."""
        B, S = ids.shape[0], ids.shape[1]
        ids, m = self._prep_ids(ids, mask)
        lab = labels.to(device=self.device, dtype=torch.int64).contiguous()
        if lab.shape != ids.shape:
            raise ValueError(f"labels {tuple(lab.shape)} must match input_ids {tuple(ids.shape)}")
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=self.device)
        g = Grads()
        keep, bufs = [], {}
        for name, dst, lc in (("backbone", g.backbone, self.cfg.backbone_config), ("decoder", g.decoder, self.cfg.decoder_config)):
            H, F, hd = lc.hidden_size, lc.intermediate_size, lc.head_dim
            nq, nkv = lc.num_attention_heads, lc.num_key_value_heads
            arr = (LayerGrads * lc.num_hidden_layers)()
            for i in range(lc.num_hidden_layers):
                L = dict(dwqkv=z((nq + 2 * nkv) * hd, H), dwo=z(H, nq * hd), dwgu=z(2 * F, H), dwd=z(H, F), dln1=z(H), dln2=z(H))
                for k, t in L.items():
                    setattr(arr[i], k, t.data_ptr())
                bufs[(name, i)] = L
            fn = z(H)
            bufs[(name, "norm")] = fn
            dst.layers, dst.final_norm = arr, fn.data_ptr()
            keep.append(arr)
        bc, dc = self.cfg.backbone_config, self.cfg.decoder_config
        te, ae = z(self.cfg.text_vocab_size, bc.hidden_size), z(self.C * self.V, bc.hidden_size)
        ph, ah = z(dc.hidden_size + self.V, bc.hidden_size), z(self.C - 1, self.V, dc.hidden_size)
        g.text_emb, g.audio_emb, g.proj_head0, g.audio_head_t = te.data_ptr(), ae.data_ptr(), ph.data_ptr(), ah.data_ptr()
        out = torch.empty(3, dtype=torch.float32, device=self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_forward_backward(self._h, _ptr(ids), _ptr(m), _ptr(lab), B, S, _ptr(out), C.byref(g)))
        self.sync()
        grads = {"text_embeddings.weight": te, "audio_embeddings.weight": ae, "projection.weight": ph[: dc.hidden_size],
                 "codebook0_head.weight": ph[dc.hidden_size:], "audio_head": ah.transpose(1, 2).contiguous()}
        for name, lc in (("backbone", bc), ("decoder", dc)):
            nqd, nkd = lc.num_attention_heads * lc.head_dim, lc.num_key_value_heads * lc.head_dim
            for i in range(lc.num_hidden_layers):
                L, p = bufs[(name, i)], f"{name}.layers.{i}"
                grads[f"{p}.self_attn.q_proj.weight"] = L["dwqkv"][:nqd]
                grads[f"{p}.self_attn.k_proj.weight"] = L["dwqkv"][nqd:nqd + nkd]
                grads[f"{p}.self_attn.v_proj.weight"] = L["dwqkv"][nqd + nkd:]
                grads[f"{p}.self_attn.o_proj.weight"] = L["dwo"]
                grads[f"{p}.mlp.gate_proj.weight"] = L["dwgu"][0::2]
                grads[f"{p}.mlp.up_proj.weight"] = L["dwgu"][1::2]
                grads[f"{p}.mlp.down_proj.weight"] = L["dwd"]
                grads[f"{p}.input_layernorm.weight"] = L["dln1"]
                grads[f"{p}.post_attention_layernorm.weight"] = L["dln2"]
            grads[f"{name}.norm.weight"] = bufs[(name, "norm")]
        return out, grads

    def step_ids(self, ids: torch.Tensor, mask: Optional[torch.Tensor], advance_frame: bool):
        B = ids.shape[0]
        ids, m = self._prep_ids(ids.reshape(B, -1), None if mask is None else mask.reshape(B, -1))
        _ck(self.lib, self.lib.csm_backbone_step_ids(self._h, _ptr(ids), _ptr(m), B, 1 if advance_frame else 0))
        lh = torch.empty(B, self.Hb, dtype=torch.float32, device=self.device)
        lg = torch.empty(B, self.V, dtype=torch.float32, device=self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_get_state(self._h, _ptr(lh), _ptr(lg)))
        self.sync()
        self.length += 1
        if advance_frame:
            self.frames += 1
        self.batch = B
        return lh, lg

    def sampling(self, temperature=1.0, topk=50, seed=0, noise=None, forced=None, logits_trace=None,
                 last_h_trace=None, row_offset=0, per_row_stop=False) -> Sampling:
        s = Sampling()
        s.temperature, s.topk, s.seed = float(temperature), int(topk), int(seed) & (2 ** 64 - 1)
        s.row_offset = int(row_offset)
        s.per_row_stop = 1 if per_row_stop else 0
        s.noise = None if noise is None else noise.data_ptr()
        s.forced = None if forced is None else forced.data_ptr()
        s.logits_trace = None if logits_trace is None else logits_trace.data_ptr()
        s.last_h_trace = None if last_h_trace is None else last_h_trace.data_ptr()
        self._keep = (noise, forced, logits_trace, last_h_trace)
        return s

    def decode_frame(self, s: Sampling):
        _ck(self.lib, self.lib.csm_decode_frame(self._h, C.byref(s)))

    def backbone_step(self, s: Sampling):
        _ck(self.lib, self.lib.csm_backbone_step(self._h, C.byref(s)))
        self.length += 1
        self.frames += 1

    def generate(self, s: Sampling, n_frames: int, use_graph: bool = True):
        _ck(self.lib, self.lib.csm_generate(self._h, C.byref(s), int(n_frames), 1 if use_graph else 0))
        self.length += n_frames
        self.frames += n_frames

    def last_generate_ms(self) -> float:
        ms = C.c_float()
        _ck(self.lib, self.lib.csm_last_generate_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def read_frames(self, first: int, n: int) -> torch.Tensor:
        out = torch.empty(self.batch, n, self.C, dtype=torch.int64, device=self.device)
        torch.cuda.current_stream().synchronize()
        if n:
            _ck(self.lib, self.lib.csm_read_frames(self._h, _ptr(out), first, n))
        self.sync()
        return out

    def get_state(self):
        lh = torch.empty(self.batch, self.Hb, dtype=torch.float32, device=self.device)
        lg = torch.empty(self.batch, self.V, dtype=torch.float32, device=self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_get_state(self._h, _ptr(lh), _ptr(lg)))
        self.sync()
        return lh, lg

    def export_kv(self):
        """Backbone KV cache of the resident batch in the HF layout: list over layers of (keys, values), each
        [B, n_kv, length, head_dim] fp32 on the device (what transformers' DynamicCache holds per layer)."""
        lc = self.cfg.backbone_config
        out = []
        for l in range(lc.num_hidden_layers):
            k = torch.empty(self.batch, lc.num_key_value_heads, self.length, lc.head_dim, dtype=torch.float32, device=self.device)
            v = torch.empty_like(k)
            torch.cuda.current_stream().synchronize()
            _ck(self.lib, self.lib.csm_kv_export(self._h, l, _ptr(k), _ptr(v), self.length))
            out.append((k, v))
        self.sync()
        return out

    def import_kv(self, layers):
        """Inverse of export_kv: the engine continues from the caller's per-layer (keys, values)."""
        lc = self.cfg.backbone_config
        if len(layers) != lc.num_hidden_layers:
            raise ValueError(f"expected {lc.num_hidden_layers} layers of (keys, values), got {len(layers)}")
        B, _, L, _ = layers[0][0].shape
        for l, (k, v) in enumerate(layers):
            if tuple(k.shape) != (B, lc.num_key_value_heads, L, lc.head_dim) or tuple(v.shape) != tuple(k.shape):
                raise ValueError(f"layer {l}: keys/values must be [B, {lc.num_key_value_heads}, L, {lc.head_dim}]")
            k = k.to(self.device, torch.float32).contiguous()
            v = v.to(self.device, torch.float32).contiguous()
            torch.cuda.current_stream().synchronize()
            _ck(self.lib, self.lib.csm_kv_import(self._h, l, _ptr(k), _ptr(v), B, L))
            self.sync()
        _ck(self.lib, self.lib.csm_set_length(self._h, B, L))
        self.sync()
        self.batch, self.length = B, L

    def zero_counts(self, first: int, n: int):
        """rows whose frame was all-zero, for frames first .. first+n-1 (one stream sync)."""
        a = (C.c_int32 * max(n, 1))()
        _ck(self.lib, self.lib.csm_read_zero_counts(self._h, a, int(first), int(n)))
        return [int(a[i]) for i in range(n)]

    def rewind_frames(self):
        """generate_frame streaming: frames already handed to the caller free their ring slots."""
        _ck(self.lib, self.lib.csm_rewind_frames(self._h))
        self.frames = 0

    def graph_stats(self):
        a, b = C.c_int(), C.c_int()
        _ck(self.lib, self.lib.csm_graph_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def prefetch_stats(self) -> dict:
        """Weight-streamer bookkeeping of the last generate() (csm_prefetch_stats)."""
        a = (C.c_longlong * 10)()
        _ck(self.lib, self.lib.csm_prefetch_stats(self._h, a))
        keys = ("gave_up", "finished", "skipped_late_sample", "xcd_rotation", "segments", "streamed_launches",
                "scheduled_bytes", "streamed_launch_bytes", "launches_counted", "frames")
        d = dict(zip(keys, [int(v) for v in a]))
        d["note"] = self.lib.csm_last_error().decode(errors="replace")
        d["health"] = self.prefetch_health()
        return d

    def prefetch_health(self) -> dict:
        """Run-time health of the weight streamer (csm_prefetch_health): whether it is still on, and why not."""
        a = (C.c_longlong * 8)()
        _ck(self.lib, self.lib.csm_prefetch_health(self._h, a))
        keys = ("disabled", "strikes", "gave_up_total", "finished_total", "streamer_launches", "budget_us", "probe_runs", "pending")
        d = dict(zip(keys, [int(v) for v in a]))
        d["reason"] = self.lib.csm_last_error().decode(errors="replace")
        return d

    def set_debug_buffer(self, buf: Optional[torch.Tensor], n_launches: int = 0):
        self._dbg_keep = buf
        _ck(self.lib, self.lib.csm_set_debug_buffer(self._h, _ptr(buf), int(n_launches)))

    def last_geoms(self, max_launches: int = 1024):
        a = (C.c_int32 * (5 * max_launches))()
        n = C.c_int()
        _ck(self.lib, self.lib.csm_last_geoms(self._h, a, max_launches, C.byref(n)))
        return [tuple(a[5 * i:5 * i + 5]) for i in range(n.value)]

    def adopt_state(self, other: "Engine"):
        """Move the live context (KV caches, counters, frame ring, pending logits) of `other` into this engine."""
        _ck(self.lib, self.lib.csm_kv_copy(self._h, other._h))
        self.batch, self.length, self.frames = other.batch, other.length, other.frames

    def device_counters(self):
        a, b = C.c_int(), C.c_int()
        _ck(self.lib, self.lib.csm_cur_len(self._h, C.byref(a)))
        _ck(self.lib, self.lib.csm_frames_done(self._h, C.byref(b)))
        return a.value, b.value

    # ---- per-kernel entry points (tests) ------------------------------------------------------------------
    def k_embed_sum(self, ids, mask):
        rows = ids.numel() // (self.C + 1)
        ids, m = self._prep_ids(ids.reshape(rows, self.C + 1), None if mask is None else mask.reshape(rows, self.C + 1))
        out = torch.empty(rows, self.Hb, dtype=torch.float32, device=self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_embed_sum(self._h, _ptr(ids), _ptr(m), rows, _ptr(out)))
        self.sync()
        return out

    def k_rmsnorm(self, x, w, eps):
        x = x.to(self.device, torch.float32).contiguous()
        w = w.to(self.device, torch.float32).contiguous()
        out = torch.empty_like(x)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_rmsnorm(self._h, _ptr(x), _ptr(w), x.shape[0], x.shape[1], eps, _ptr(out)))
        self.sync()
        return out

    def k_gemv(self, W, x, ln=None, eps=1e-5, scale=None):
        W = W.to(self.device).contiguous()
        sc = None if scale is None else scale.to(self.device, torch.float32).contiguous()
        x = x.to(self.device, torch.float32).contiguous()
        lnw = None if ln is None else ln.to(self.device, torch.float32).contiguous()
        y = torch.empty(x.shape[0], W.shape[0], dtype=torch.float32, device=self.device)
        torch.cuda.current_stream().synchronize()
        wd = {torch.bfloat16: DT_BF16, torch.uint8: DT_FP8}.get(W.dtype, DT_F32)
        _ck(self.lib, self.lib.csm_gemv(self._h, _ptr(W), wd, _ptr(sc), W.shape[0], W.shape[1], _ptr(x), x.shape[0],
                                        _ptr(lnw), eps, _ptr(y)))
        self.sync()
        return y

    def k_gemm(self, W, A, scale=None):
        W = W.to(self.device).contiguous()
        sc = None if scale is None else scale.to(self.device, torch.float32).contiguous()
        A = A.to(self.device, torch.float32).contiguous()
        out = torch.empty(A.shape[0], W.shape[0], dtype=torch.float32, device=self.device)
        torch.cuda.current_stream().synchronize()
        wd = {torch.bfloat16: DT_BF16, torch.uint8: DT_FP8}.get(W.dtype, DT_F32)
        _ck(self.lib, self.lib.csm_gemm(self._h, _ptr(W), wd, _ptr(sc), W.shape[0], W.shape[1], _ptr(A), A.shape[0], _ptr(out)))
        self.sync()
        return out

    def k_gemm_bf16(self, W, A, kernel: int):
        """C = A @ W^T with both operands bf16 (the producer rounded A), fp32 accumulate, on a pinned tile (csm_gemm_bf16)"""
        W = W.to(self.device, torch.bfloat16).contiguous()
        A = A.to(self.device, torch.bfloat16).contiguous()
        out = torch.empty(A.shape[0], W.shape[0], dtype=torch.float32, device=self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_gemm_bf16(self._h, _ptr(W), W.shape[0], W.shape[1], _ptr(A), A.shape[0], _ptr(out), int(kernel)))
        self.sync()
        return out

    def k_mx_quantize(self, x):
        x = x.to(self.device, torch.float32).contiguous()
        q = torch.empty(x.shape, dtype=torch.uint8, device=self.device)
        s = torch.empty(x.shape[0], x.shape[1] // 32, dtype=torch.uint8, device=self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_mx_quantize(self._h, _ptr(x), x.shape[0], x.shape[1], _ptr(q), _ptr(s)))
        self.sync()
        return q, s

    def k_gemm_mx(self, Wq, Ws, Aq, As):
        Wq, Ws, Aq, As = (t.to(self.device).contiguous() for t in (Wq, Ws, Aq, As))
        out = torch.empty(Aq.shape[0], Wq.shape[0], dtype=torch.float32, device=self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_gemm_mx(self._h, _ptr(Wq), _ptr(Ws), Wq.shape[0], Wq.shape[1], _ptr(Aq), _ptr(As), Aq.shape[0], _ptr(out)))
        self.sync()
        return out

    def k_sample(self, logits, topk, temperature, seed=0, noise=None):
        lg = logits.to(self.device, torch.float32).contiguous()
        nz = None if noise is None else noise.to(self.device, torch.float32).contiguous()
        out = torch.empty(lg.shape[0], dtype=torch.int32, device=self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_sample_topk(self._h, _ptr(lg), lg.shape[0], lg.shape[1], float(temperature), int(topk),
                                               int(seed), _ptr(nz), _ptr(out)))
        self.sync()
        return out

    def bench_gemv(self, N, K, M=1, norm=False, epi=0, nt=1, pool_mb=512, n_launch=200, reps=10, dtype=torch.bfloat16,
                   grid_cap=0, v2_tasks=0, force_generic=0):
        """us per launch of one GEMV shape inside a dependent hipGraph chain (weights cycle over a pool)."""
        esz = 2 if dtype == torch.bfloat16 else 4
        wbytes = N * K * esz
        n_w = max(1, (pool_mb << 20) // wbytes)
        W = (torch.randn(n_w * N * K // 4 + 16, device=self.device) * 0.02).to(dtype).repeat(4)[: n_w * N * K].contiguous()
        x = torch.randn(M, K + (force_generic >> 8), device=self.device)   # bits 8.. of force_generic = row-stride pad (floats)
        ln = torch.ones(K, device=self.device) if norm else None
        y = torch.zeros(M, N, device=self.device)
        us = C.c_float()
        torch.cuda.synchronize()
        _ck(self.lib, self.lib.csm_bench_gemv(self._h, _ptr(W), wbytes, n_w, DT_BF16 if dtype == torch.bfloat16 else DT_F32,
                                              N, K, _ptr(x), M, _ptr(ln), 1e-5, _ptr(y), epi, nt, n_launch, reps, C.byref(us),
                                              grid_cap, v2_tasks, force_generic))
        return float(us.value), wbytes

    def k_rope_scatter(self, which, layer, qkv, row_seq, row_pos):
        qkv = qkv.to(self.device, torch.float32).contiguous()
        rs = row_seq.to(self.device, torch.int32).contiguous()
        rp = row_pos.to(self.device, torch.int32).contiguous()
        lc = self.cfg.decoder_config if which else self.cfg.backbone_config
        q = torch.empty(qkv.shape[0], lc.num_attention_heads * lc.head_dim, dtype=torch.float32, device=self.device)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_rope_scatter(self._h, which, layer, _ptr(qkv), _ptr(rs), _ptr(rp), qkv.shape[0], _ptr(q)))
        self.sync()
        return q

    def k_attn(self, which, layer, q, row_seq, row_pos, nsplit=1):
        q = q.to(self.device, torch.float32).contiguous()
        rs = row_seq.to(self.device, torch.int32).contiguous()
        rp = row_pos.to(self.device, torch.int32).contiguous()
        out = torch.empty_like(q)
        torch.cuda.current_stream().synchronize()
        _ck(self.lib, self.lib.csm_attn_decode(self._h, which, layer, _ptr(q), _ptr(rs), _ptr(rp), q.shape[0], nsplit, _ptr(out)))
        self.sync()
        return out
