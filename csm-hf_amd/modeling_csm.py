"""Host-side mirror of the reference's model API for the generation path.

Same names, argument meaning and error behaviour as `/root/reference/modeling_csm.py`:
`CSMOutput` (:30-49), `sample_topk` (:179-189), `CSMModel.{forward, generate_frame, generate,
setup_caches, reset_caches}` (:284-702), HF-style `from_pretrained / save_pretrained / to / eval /
state_dict` on the reference checkpoint layout.  All arithmetic runs in
libcsm_hip.so (hand-written gfx950 kernels); torch only carries device memory.

Deviations from the reference, all deliberate (DESIGN.md "Deviations"):
  * `temperature == 0` means argmax (the reference divides by zero, :181).
  * greedy ties resolve to the lowest index (the reference draws among exact ties, :183-189).
  * left-padded rows mask their pads at every step, so a padded row equals its solo run (the reference
    forgets the pad mask on decode steps).
  * `past_key_values` is an opaque handle onto the engine-resident KV cache, not a `DynamicCache`.
  * the training branch (`labels=`, :367-465): `forward(labels=...)` returns the reference's loss, backbone_loss and
    decoder_loss; with gradients enabled and parameters that require them the loss carries an autograd node backed by the
    HIP backward pass (csm_forward_backward), so `loss.backward()` fills `.grad` like the reference's training loop.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, fields
from typing import Dict, Optional

import torch
import torch.nn as nn

from .configuration_csm import CSMConfig
from .engine import Engine, load_library


@dataclass
class CSMOutput:
    """reference modeling_csm.py:30-49 (a ModelOutput there; tuple/index/key access kept)."""
    last_hidden_state: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[object] = None
    samples: Optional[torch.Tensor] = None
    loss: Optional[torch.Tensor] = None
    backbone_loss: Optional[torch.Tensor] = None
    decoder_loss: Optional[torch.Tensor] = None

    def to_tuple(self):
        return tuple(getattr(self, f.name) for f in fields(self) if getattr(self, f.name) is not None)

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else self.to_tuple()[k]

    def keys(self):
        return [f.name for f in fields(self) if getattr(self, f.name) is not None]


class CSMKVCache:
    """Handle returned as `past_key_values`: the KV cache itself lives in the engine (pre-allocated, in
    place), this object only proves which engine state a later call continues from."""

    def __init__(self, model: "CSMModel", epoch: int, length: int, batch: int, frame_pending: bool):
        import weakref
        self._model_id = id(model)
        self._model_ref = weakref.ref(model)
        self.epoch, self.length, self.batch, self.frame_pending = epoch, length, batch, frame_pending

    def get_seq_length(self) -> int:
        return self.length

    def to_legacy_cache(self):
        """The cache in the HF layout the reference hands out (`DynamicCache`, modeling_csm.py:355-358): a tuple over
        backbone layers of (keys, values), each `[B, n_kv, length, head_dim]` fp32 on the device.  A copy: it can be
        kept, forked or edited and passed back as `past_key_values` later (the engine imports it)."""
        m = self._model_ref()
        if m is None or m._engine is None or self.epoch != m._epoch or self.length != m._engine.length:
            raise ValueError("stale cache handle: only the handle returned by the most recent call can be exported")
        return tuple(m._engine.export_kv())


def _hf_cache_layers(pkv):
    """(keys, values) per layer out of a legacy tuple / list, a transformers >= 4.56 DynamicCache (`.layers[i].keys`)
    or an older one (`.key_cache` / `.value_cache`); None if `pkv` is none of these."""
    if isinstance(pkv, (tuple, list)) and pkv and isinstance(pkv[0], (tuple, list)) and len(pkv[0]) >= 2 \
            and torch.is_tensor(pkv[0][0]):
        return [(l[0], l[1]) for l in pkv]
    if hasattr(pkv, "layers") and len(getattr(pkv, "layers")) and hasattr(pkv.layers[0], "keys"):
        return [(l.keys, l.values) for l in pkv.layers]
    if hasattr(pkv, "key_cache") and hasattr(pkv, "value_cache"):
        return list(zip(pkv.key_cache, pkv.value_cache))
    return None


def _hf_cache_is_empty(pkv) -> bool:
    """an HF cache object (or legacy tuple) that holds no positions yet"""
    if isinstance(pkv, (tuple, list)):
        return len(pkv) == 0
    if hasattr(pkv, "get_seq_length") and (hasattr(pkv, "layers") or hasattr(pkv, "key_cache")):
        try:
            return int(pkv.get_seq_length()) == 0
        except Exception:
            return False
    return False


_default_engine_for_sampling = {}


def sample_topk(logits: torch.Tensor, topk: int, temperature: float, seed: Optional[int] = None,
                noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reference `sample_topk` (modeling_csm.py:179-189): int32 `[..., 1]`, on the logits' device.
    Runs the K12 sampler kernel through the C ABI (stand-alone entry `csm_sample_topk`)."""
    import ctypes as C
    if logits.device.type != "cuda":
        raise RuntimeError("csm_hf_amd.sample_topk needs logits on the GPU (no CPU fallback)")
    lib = load_library()
    V = logits.shape[-1]
    if topk > V or topk < 1:
        raise RuntimeError("selected index k out of range")
    # (top-k keeps 3 x V floats in LDS: V <= 13 300; greedy has no limit -- the C entry reports it otherwise)
    lg = logits.reshape(-1, V).to(torch.float32).contiguous()
    nz = None if noise is None else noise.reshape(-1, V).to(logits.device, torch.float32).contiguous()
    out = torch.empty(lg.shape[0], dtype=torch.int32, device=logits.device)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    torch.cuda.current_stream(logits.device).synchronize()
    with torch.cuda.device(logits.device):   # the engine-less entry launches on the current device's null stream
        rc = lib.csm_sample_topk(None, C.c_void_p(lg.data_ptr()), lg.shape[0], V, float(temperature), int(topk), int(seed),
                                 None if nz is None else C.c_void_p(nz.data_ptr()), C.c_void_p(out.data_ptr()))
    if rc != 0:
        raise RuntimeError(lib.csm_last_error().decode())
    return out.reshape(*logits.shape[:-1], 1)


def _llama_modules(lc) -> nn.Module:
    """Parameter containers named exactly like transformers.LlamaModel (embed_tokens is Identity in the
    reference, modeling_csm.py:156-167, so it contributes no tensor)."""
    m = nn.Module()
    m.layers = nn.ModuleList()
    hd = lc.head_dim
    for _ in range(lc.num_hidden_layers):
        layer = nn.Module()
        att = nn.Module()
        att.q_proj = nn.Linear(lc.hidden_size, lc.num_attention_heads * hd, bias=False)
        att.k_proj = nn.Linear(lc.hidden_size, lc.num_key_value_heads * hd, bias=False)
        att.v_proj = nn.Linear(lc.hidden_size, lc.num_key_value_heads * hd, bias=False)
        att.o_proj = nn.Linear(lc.num_attention_heads * hd, lc.hidden_size, bias=False)
        mlp = nn.Module()
        mlp.gate_proj = nn.Linear(lc.hidden_size, lc.intermediate_size, bias=False)
        mlp.up_proj = nn.Linear(lc.hidden_size, lc.intermediate_size, bias=False)
        mlp.down_proj = nn.Linear(lc.intermediate_size, lc.hidden_size, bias=False)
        layer.self_attn, layer.mlp = att, mlp
        layer.input_layernorm = _Norm(lc.hidden_size)
        layer.post_attention_layernorm = _Norm(lc.hidden_size)
        m.layers.append(layer)
    m.norm = _Norm(lc.hidden_size)
    return m


class _CSMTrainLoss(torch.autograd.Function):
    """Autograd bridge of the HIP training pass: forward runs csm_forward_backward once (loss AND every gradient), backward
    hands the stored gradients out scaled by d(loss) -- what `loss.backward()` of the reference's HF-Trainer loop needs."""

    @staticmethod
    def forward(ctx, model, input_ids, attention_mask, labels, names, *params):
        out, grads = model.loss_and_grads(input_ids, attention_mask, labels)
        ctx.grads = [grads[n] for n in names]
        ctx.dtypes = [p.dtype for p in params]
        loss, bl, dl = out.loss.clone(), out.backbone_loss.clone(), out.decoder_loss.clone()
        ctx.mark_non_differentiable(bl, dl)      # the tensors that are RETURNED (marking the originals had no effect)
        return loss, bl, dl

    @staticmethod
    def backward(ctx, g_loss, g_bl, g_dl):
        grads, ctx.grads = ctx.grads, None      # one fp32 copy of every gradient (6 GB on csm-1b): released with this call
        if grads is None:
            raise RuntimeError("backward through CSMModel.forward(labels=...) a second time: the gradients were handed out once")
        gs = []
        for i, dt in enumerate(ctx.dtypes):
            gs.append((grads[i] * g_loss).to(dt))
            grads[i] = None
        return (None, None, None, None, None, *gs)


class _Norm(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))


class CSMModel(nn.Module):
    """reference `CSMModel` (modeling_csm.py:192-702), generation path only."""

    config_class = CSMConfig
    base_model_prefix = "csm"
    DEFAULT_KV_DTYPE = "auto"   # instances start with kv_dtype = this (tests that assert bit-exactness against the fp32-arithmetic fixtures set torch.float32 themselves)

    def __init__(self, config: CSMConfig):
        super().__init__()
        self.config = config
        with torch.device("meta"):
            self.backbone = _llama_modules(config.backbone_config)
            self.decoder = _llama_modules(config.decoder_config)
            bh, dh = config.backbone_config.hidden_size, config.decoder_config.hidden_size
            self.text_embeddings = nn.Embedding(config.text_vocab_size, bh)
            self.audio_embeddings = nn.Embedding(config.audio_vocab_size * config.audio_num_codebooks, bh)
            self.projection = nn.Linear(bh, dh, bias=False)
            self.codebook0_head = nn.Linear(bh, config.audio_vocab_size, bias=False)
            self.audio_head = nn.Parameter(torch.empty(config.audio_num_codebooks - 1, dh, config.audio_vocab_size))
        self.requires_grad_(False)
        self._using_kv_cache = False
        self._engine: Optional[Engine] = None
        self._engine_sig = None
        self._plist = None
        self._epoch = 0
        self._frame_pending = False
        self._caps = dict(max_batch=1, max_len=0, max_frames=0, max_prefill_rows=0)
        self.kv_dtype = type(self).DEFAULT_KV_DTYPE   # "auto" (default, round 5): the KV cache follows the model dtype -- a bf16 checkpoint caches bf16 K / V like the
        # reference's DynamicCache of a bf16 model (README.md:73); torch.float32 = the exact mode (a bf16-weight model then reproduces the
        # reference's fp32-arithmetic token stream bit for bit: what the parity tests and the headline bench pin); torch.bfloat16 forces bf16
        self.weight_format = "native"   # "fp8": linear weights as e4m3fn + per-row scales (BASELINE config 5)
        self.use_graph = True
        self.stop_check_interval = 8     # stop_on_all_zeros: frames replayed between two reads of the device-side stop counters
        self.last_row_lengths = None     # per-row frame counts of the last generate(per_row_stop=True)
        self.decode_precision = "exact"    # "bf16": BATCHED decode (B >= 2) hands activations between its matrix-core launches as one
        # nearest-even bf16 plane -- the reference's own arithmetic class (README.md:73 runs the model in bf16) -- instead of three
        # exact planes: a third of the matrix work and of the plane traffic; B = 1 (fp32 FMA kernels) is unaffected
        self.prefill_precision = "exact"   # "bf16": context GEMMs on bf16-rounded activations (one MFMA pass instead of three);
        # "mxfp8": context GEMMs on the block-scaled fp8 matrix instruction, weights AND activations in OCP MX-fp8 (e4m3 + one
        # E8M0 scale per 32 along K) -- 3 mantissa bits: its own accuracy class, opt-in
        self.seed = 0
        self.row_offset = 0             # global index of row 0 of this model's batch (batch-sharded generation)

    # ---- HF-style plumbing -------------------------------------------------------------------------------
    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = True):
        self._drop_engine()
        return super().load_state_dict(state_dict, strict=strict, assign=True)

    def _apply(self, fn, *a, **k):
        self._drop_engine()
        return super()._apply(fn, *a, **k)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype: Optional[torch.dtype] = None, device=None, **_):
        from safetensors.torch import load_file
        cfg = CSMConfig.from_pretrained(path)
        model = cls(cfg)
        f = os.path.join(path, "model.safetensors")
        if os.path.exists(f):
            sd = load_file(f)
        else:
            idx = json.load(open(os.path.join(path, "model.safetensors.index.json")))
            sd = {}
            for shard in sorted(set(idx["weight_map"].values())):
                sd.update(load_file(os.path.join(path, shard)))
        if torch_dtype is not None:
            sd = {k: v.to(torch_dtype) for k, v in sd.items()}
        model.load_state_dict(sd, strict=True)
        if device is not None:
            model.to(device)
        return model.eval()

    def save_pretrained(self, path: str):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        self.config.torch_dtype = self.dtype
        self.config.save_pretrained(path)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "model.safetensors"), metadata={"format": "pt"})

    # ---- engine management ---------------------------------------------------------------------------------
    def _param_signature(self):
        """(storage address, version counter) of every parameter: changes when a parameter is replaced or written in place."""
        # (the list is rebuilt on every call: a cached list missed `mod.weight = nn.Parameter(...)`, which keeps the parameter COUNT --
        #  ADVICE r4; ~200 parameters, microseconds)
        return tuple((id(q), q.data_ptr(), q._version) for q in self.parameters())

    def _drop_engine(self):
        self._plist = None
        if getattr(self, "_engine", None) is not None:
            self._engine.close()
            self._engine = None
            self._epoch += 1

    def setup_caches(self, max_batch_size: int, max_seq_len: Optional[int] = None, max_frames: Optional[int] = None):
        """reference :284-286 flips a flag; here it also sizes the engine-resident KV cache."""
        self._using_kv_cache = True
        self._caps["max_batch"] = max(self._caps["max_batch"], int(max_batch_size))
        if max_seq_len:
            self._caps["max_len"] = max(self._caps["max_len"], int(max_seq_len))
        if max_frames:
            self._caps["max_frames"] = max(self._caps["max_frames"], int(max_frames))

    def reset_caches(self):
        """reference :288-290 (`pass`): drop the cached context."""
        if self._engine is not None:
            self._engine.reset()
        self._epoch += 1
        self._frame_pending = False

    def _ensure_engine(self, batch: int, need_len: int, need_frames: int, prefill_rows: int, cont: bool = False,
                       must_prefill_rows: int = 0) -> Engine:
        """Create or re-size the engine.  `cont`: the call continues a live context (past_key_values): the engine may
        still grow (the reference's DynamicCache grows without bound), but then the resident KV cache, counters, frame
        ring and pending logits are MOVED into the larger engine (csm_kv_copy) -- a continuation never restarts from an
        empty cache.  `prefill_rows` is a sizing HINT for the prefill scratch (capped at 8192 rows: Engine.prefill chunks
        longer contexts); `must_prefill_rows` is a REQUIREMENT of callers that cannot chunk (the training forward needs
        B*S rows in one pass) and is a growth reason, uncapped."""
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError("CSMModel must be on an AMD GPU (model.to('cuda')): csm_hf_amd has no CPU path")
        # The engine multiplies PACKED copies of the parameters (q/k/v concatenated, gate/up interleaved, heads transposed,
        # transposed copies for the backward pass) next to aliases of the unpacked ones.  An in-place update of a parameter
        # (`optimizer.step()`, `p.data.copy_()`, `p.add_()`) bumps its version counter: the copies are then stale and the
        # engine is rebuilt from the live parameters before anything is computed (the next forward / backward
        # silently mixed updated o_proj / down_proj / embeddings with stale qkv / gate-up / heads).
        sig = self._param_signature()
        if self._engine is not None and sig != self._engine_sig:
            if cont:
                raise ValueError("parameters were modified in place while a KV cache is live: call reset_caches() first")
            self._drop_engine()
        c = self._caps
        kv_eff = self.kv_dtype if self.kv_dtype not in (None, "auto") else (torch.bfloat16 if p.dtype == torch.bfloat16 else torch.float32)
        if kv_eff not in (torch.float32, torch.bfloat16):
            raise ValueError(f"kv_dtype must be 'auto', torch.float32 or torch.bfloat16, got {self.kv_dtype!r}")
        if self._engine is not None and self._engine.kv_dtype != kv_eff:
            if cont:
                raise ValueError("kv_dtype changed while a KV cache is live: call reset_caches() first")
            self._drop_engine()
        if self._engine is not None and self._engine.fp8 != (self.weight_format == "fp8"):
            if cont:
                raise ValueError("weight_format changed while a KV cache is live")
            self._drop_engine()
        grow = (self._engine is None or batch > self._engine.max_batch or need_len > self._engine.max_len or
                need_frames > self._engine.max_frames or must_prefill_rows > self._engine.max_prefill_rows)
        if grow:
            c["max_batch"] = max(c["max_batch"], batch)
            # a live context that outgrows its cache doubles it (amortised like the reference's cat-grown cache)
            c["max_len"] = max(c["max_len"], need_len if not cont else max(need_len, 2 * self._engine.max_len),
                               self.config.max_seq_len)
            c["max_frames"] = max(c["max_frames"], need_frames, 256)
            c["max_prefill_rows"] = max(c["max_prefill_rows"], min(prefill_rows, 8192), int(must_prefill_rows))
            old = self._engine
            packed = old.packed if old is not None else None
            if old is not None and not cont:
                old.close()
                old = None
                self._epoch += 1
            eng = Engine(self.config, self.state_dict(), p.device, p.dtype, max_batch=c["max_batch"],
                         max_len=c["max_len"], max_frames=c["max_frames"],
                         max_prefill_rows=c["max_prefill_rows"], kv_dtype=kv_eff, packed=packed,
                         weight_format=self.weight_format)
            if old is not None:      # continuation: re-home the live state, then release the old engine
                eng.adopt_state(old)
                old.close()
            self._engine = eng
            self._engine_sig = sig
        if self.prefill_precision not in ("exact", "bf16", "mxfp8"):
            raise ValueError(f"prefill_precision must be 'exact', 'bf16' or 'mxfp8', got {self.prefill_precision!r}")
        want = 1 if self.prefill_precision in ("bf16", "mxfp8") else 0     # mxfp8: attention and the rest as in bf16 mode
        if getattr(self._engine, "_prefill_bf16", 0) != want:
            self._engine.set_option("prefill_bf16", want)
            self._engine._prefill_bf16 = want
        if self.decode_precision not in ("exact", "bf16"):
            raise ValueError(f"decode_precision must be 'exact' or 'bf16', got {self.decode_precision!r}")
        want_d = 1 if self.decode_precision == "bf16" else 0
        if getattr(self._engine, "_decode_bf16", 0) != want_d:
            self._engine.set_option("decode_bf16", want_d)
            self._engine._decode_bf16 = want_d
        want_mx = 1 if self.prefill_precision == "mxfp8" else 0
        if getattr(self._engine, "_prefill_mx", 0) != want_mx:
            if want_mx and not self._engine.has_mx:
                if p.dtype != torch.bfloat16:
                    raise ValueError("prefill_precision='mxfp8' needs a bf16 model")
                self._engine.enable_mx(self.state_dict())
            self._engine.set_option("prefill_mx", want_mx)
            self._engine._prefill_mx = want_mx
        return self._engine

    # ---- helpers -----------------------------------------------------------------------------------------------
    @staticmethod
    def _kv_starts(attention_mask: Optional[torch.Tensor], B: int, S: int):
        if attention_mask is None:
            return [0] * B
        valid = (attention_mask.sum(dim=-1) > 0).cpu()
        starts = []
        for b in range(B):
            v = valid[b]
            n_pad = int((~v).sum())
            if n_pad and not bool(v[n_pad:].all() and not v[:n_pad].any()):
                raise ValueError("only left padding is supported: masked frames must form a prefix of the row")
            starts.append(n_pad)
        return starts

    def _out_dtype(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(self.dtype)

    # ---- reference API ---------------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, temperature=1.0, topk=50,
                generate_frame=False, labels=None):
        """reference :292-482.  With `labels`, gradients enabled and parameters that require them (the reference's training
        call, train.py:308-326) the returned `loss` carries an autograd node: `out.loss.backward()` fills `.grad` of every
        parameter from the HIP backward pass (csm_forward_backward).  Everything else runs without autograd."""
        if labels is not None and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(input_ids, attention_mask, labels, position_ids, past_key_values, return_dict)
        return self._forward_nograd(input_ids, attention_mask, position_ids, past_key_values, use_cache, output_attentions,
                                    output_hidden_states, return_dict, temperature, topk, generate_frame, labels)

    @torch.no_grad()
    def loss_and_grads(self, input_ids, attention_mask, labels):
        """(CSMOutput with loss / backbone_loss / decoder_loss, {parameter name: fp32 gradient of `loss`}) -- the HIP training
        pass without the autograd bridge (reference objective modeling_csm.py:367-465 and its derivative)."""
        B, S = input_ids.shape[0], input_ids.shape[1]
        if tuple(labels.shape) != tuple(input_ids.shape):
            raise ValueError(f"labels {tuple(labels.shape)} must match input_ids {tuple(input_ids.shape)}")
        if self.weight_format == "fp8":
            raise ValueError("the training pass needs native (fp32 / bf16) weights")
        eng = self._ensure_engine(B, S + 1, 1, 32)
        eng.reset()
        self._epoch += 1
        self._frame_pending = False
        eng.set_kv_start(self._kv_starts(attention_mask, B, S))
        losses, grads = eng.forward_backward(input_ids, attention_mask, labels)
        return CSMOutput(loss=losses[0], backbone_loss=losses[1], decoder_loss=losses[2]), grads

    def _forward_train(self, input_ids, attention_mask, labels, position_ids, past_key_values, return_dict):
        if past_key_values is not None or position_ids is not None:
            raise NotImplementedError("forward(labels=...) takes a fresh context: no past_key_values / position_ids")
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        loss, bl, dl = _CSMTrainLoss.apply(self, input_ids, attention_mask, labels, names, *params)
        if not return_dict:
            return (loss,)
        return CSMOutput(loss=loss, backbone_loss=bl.detach(), decoder_loss=dl.detach())

    @torch.no_grad()
    def _forward_nograd(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, use_cache=None,
                        output_attentions=None, output_hidden_states=None, return_dict=None, temperature=1.0, topk=50,
                        generate_frame=False, labels=None):
        """reference :292-482, inference branch.  `temperature/topk/generate_frame` accepted and ignored
        as in the reference."""
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        use_cache = use_cache if use_cache is not None else self._using_kv_cache
        B, S = input_ids.shape[0], input_ids.shape[1]
        if labels is not None:
            return self._forward_loss(input_ids, attention_mask, labels, position_ids, past_key_values, use_cache, return_dict)
        if position_ids is not None and tuple(position_ids.shape) not in ((B, S), (1, S)):
            raise ValueError(f"position_ids must be [B, S] or [1, S], got {tuple(position_ids.shape)}")
        if past_key_values is not None and not isinstance(past_key_values, CSMKVCache) and _hf_cache_is_empty(past_key_values):
            past_key_values = None     # `DynamicCache()` / `()`: the usual HF way to start a context; the reference accepts it
        hf_layers = None if isinstance(past_key_values, CSMKVCache) else _hf_cache_layers(past_key_values)
        if hf_layers is not None:
            # a cache in the HF layout (exported earlier, forked, or built elsewhere): import it and continue from it.
            # The HF layout carries no pad information: every imported position is attended to (kv_start = 0), which is
            # what the reference does with such a cache on decode steps; export pad-free batches.
            L = hf_layers[0][0].shape[2]
            if hf_layers[0][0].shape[0] != B:
                raise ValueError("past_key_values batch does not match input_ids")
            eng = self._ensure_engine(B, L + S + 1, 1, B * S)
            eng.reset()
            self._epoch += 1
            self._frame_pending = False
            eng.set_kv_start([0] * B)
            eng.import_kv(hf_layers)
            past_key_values = CSMKVCache(self, self._epoch, L, B, False)
        cont = past_key_values is not None
        if cont:
            if not isinstance(past_key_values, CSMKVCache) or past_key_values.epoch != self._epoch or \
                    past_key_values._model_id != id(self) or self._engine is None or \
                    past_key_values.length != self._engine.length or past_key_values.batch != B:
                raise ValueError("stale or foreign past_key_values: the KV cache lives in the engine and only the "
                                 "handle returned by the most recent call can be continued")
        base = self._engine.length if cont else 0
        eng = self._ensure_engine(B, base + S + 1, 1, B * S, cont=cont)
        if not cont:
            eng.reset()
            self._epoch += 1
            self._frame_pending = False
            eng.set_kv_start(self._kv_starts(attention_mask, B, S))
        if cont and S == 1 and position_ids is None and eng.length > 0:
            last_h, c0 = eng.step_ids(input_ids, attention_mask, advance_frame=self._frame_pending)
        else:
            if cont and self._frame_pending:
                raise NotImplementedError("right after generate_frame only the reference's own continuation is supported: "
                                          "ONE frame, no position_ids (multi-frame / re-positioned steps are not)")
            last_h, c0 = eng.prefill(input_ids, attention_mask, position_ids=position_ids)
        self._frame_pending = False
        pkv = CSMKVCache(self, self._epoch, eng.length, B, False) if use_cache else None
        last_h, c0 = self._out_dtype(last_h), self._out_dtype(c0)
        if not return_dict:
            out = (last_h, c0)
            return out + (pkv,) if use_cache else out
        return CSMOutput(last_hidden_state=last_h, logits=c0, past_key_values=pkv)

    def _forward_loss(self, input_ids, attention_mask, labels, position_ids, past_key_values, use_cache, return_dict):
        """reference :367-465, forward only: loss = backbone_loss (codebook-0 cross-entropy, shifted by one position) +
        decoder_loss (codebooks 1..31 over the frames whose 32 audio labels are all present).  The losses come back as
        fp32 tensors WITHOUT autograd history -- this package has no backward pass; the call evaluates the reference's
        training objective.  Like the reference's training batches: a fresh context (no past_key_values / position_ids)."""
        if past_key_values is not None or position_ids is not None:
            raise NotImplementedError("forward(labels=...) takes a fresh context: no past_key_values / position_ids")
        B, S = input_ids.shape[0], input_ids.shape[1]
        if tuple(labels.shape) != tuple(input_ids.shape):
            raise ValueError(f"labels {tuple(labels.shape)} must match input_ids {tuple(input_ids.shape)}")
        eng = self._ensure_engine(B, S + 1, 1, max(B * S, 32), must_prefill_rows=B * S)   # one pass over all B*S rows
        eng.reset()
        self._epoch += 1
        self._frame_pending = False
        eng.set_kv_start(self._kv_starts(attention_mask, B, S))
        losses, last_h, c0 = eng.forward_loss(input_ids, attention_mask, labels)
        pkv = CSMKVCache(self, self._epoch, eng.length, B, False) if use_cache else None
        last_h, c0 = self._out_dtype(last_h), self._out_dtype(c0)
        loss, bl, dl = losses[0], losses[1], losses[2]
        if not return_dict:
            out = (loss, last_h, c0)
            return out + (pkv,) if use_cache else out
        return CSMOutput(last_hidden_state=last_h, logits=c0, past_key_values=pkv, loss=loss, backbone_loss=bl, decoder_loss=dl)

    @torch.no_grad()
    def generate_frame(self, input_ids, attention_mask, position_ids=None, temperature=1.0, topk=50,
                       past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                       return_dict=None, *, noise: Optional[torch.Tensor] = None, rng: Optional[str] = None):
        """reference :484-589.  `noise` (extension) `[B, 32, V]`: explicit Exp(1) draws that replace the device RNG -- the
        reference's `torch.empty_like(probs).exponential_(1)` (:175) made reproducible: with the reference's own draws the
        sampled tokens are the reference's.  `rng="torch"` (extension): the draws are taken from torch's GLOBAL generator
        exactly as the reference takes them (`_torch_rng_noise`), so `torch.manual_seed(s)` reproduces the reference's
        sampled frames."""
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        use_cache = use_cache if use_cache is not None else self._using_kv_cache
        out = self.forward(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                           past_key_values=past_key_values, use_cache=True, return_dict=True)
        eng = self._engine
        if eng.frames + 1 > eng.max_frames:
            # every frame of this path has already been handed to the caller (the reference keeps none either,
            # :578-589): restart the on-device ring instead of limiting a stream to max_frames frames
            eng.rewind_frames()
        if noise is None and rng is not None:
            noise = self._torch_rng_noise(rng, eng.batch)
        nz = None if noise is None else self._check_noise(noise, eng.batch).to(eng.device, torch.float32).contiguous()
        s = eng.sampling(temperature=temperature, topk=topk, seed=self._next_seed(), row_offset=self.row_offset, noise=nz)
        eng.decode_frame(s)
        tokens = eng.read_frames(eng.frames, 1)[:, 0, :]
        self._frame_pending = True
        pkv = CSMKVCache(self, self._epoch, eng.length, eng.batch, True) if use_cache else None
        if not use_cache:
            self._epoch += 1
        if not return_dict:
            return tokens
        return CSMOutput(last_hidden_state=out.last_hidden_state, logits=out.logits, past_key_values=pkv, samples=tokens)

    def _torch_rng_noise(self, rng: str, B: int) -> torch.Tensor:
        """One frame's Exp(1) draws `[B, 32, V]` consumed from torch's global generator the way the reference consumes it:
        32 calls of `torch.empty_like(probs).exponential_(1)` on a `[B, V]` tensor in codebook order
        (/root/reference/modeling_csm.py:170-176, called from :531 and :558-576).  `self.rng_device` ("cpu" by default: the
        generator a CPU run of the reference uses; set it to the model's device for the reference's GPU stream) and
        `self.rng_dtype` (the dtype of the reference's `probs`: its compute dtype) select the generator and the draw type."""
        if rng != "torch":
            raise ValueError("rng must be None (device Philox stream) or 'torch' (torch's global generator, as the reference)")
        C, V = self.config.audio_num_codebooks, self.config.audio_vocab_size
        dev = getattr(self, "rng_device", "cpu")
        dt = getattr(self, "rng_dtype", torch.float32)
        return torch.stack([torch.empty(B, V, dtype=dt, device=dev).exponential_(1) for _ in range(C)], 1).float()

    def _check_noise(self, noise: torch.Tensor, B: int) -> torch.Tensor:
        want = (B, self.config.audio_num_codebooks, self.config.audio_vocab_size)
        if tuple(noise.shape) != want:
            raise ValueError(f"noise must be {want} (one Exp(1) draw per row, codebook and vocabulary entry), got {tuple(noise.shape)}")
        return noise

    def _next_seed(self) -> int:
        self.seed += 1
        return (int(torch.initial_seed()) * 1000003 + self.seed) & (2 ** 63 - 1)

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, max_new_frames: int = 100,
                 temperature: float = 1.0, topk: int = 50, use_cache: bool = True, stop_on_all_zeros: bool = True,
                 *, seed: Optional[int] = None, per_row_stop: bool = False, noise: Optional[torch.Tensor] = None,
                 rng: Optional[str] = None):
        """reference :591-702.  Returns LongTensor `[B, n, 32]` on `input_ids.device`.  `seed` (extension, default: drawn
        from torch's seed and a call counter) keys the device Philox stream of the sampler.  `per_row_stop` (extension,
        SURVEY.md section 8 f-4): a row that has emitted an all-zero frame is frozen (emits zeros from then on), generation
        ends when the last row has finished, `self.last_row_lengths` holds every row's own frame count; the default keeps
        the reference's global rule (:662: stop when ALL rows emit an all-zero frame in the same step).  `noise`
        (extension) `[max_new_frames, B, 32, V]`: explicit Exp(1) draws instead of the device RNG (see generate_frame).
        `rng="torch"` (extension): every frame's draws come from torch's global generator in the reference's order and
        shapes, so the sampled frames are the reference's under the same `torch.manual_seed` (one launch per frame).

        One prefill, then per frame one replay of the captured hipGraph (31-step decoder loop + next
        backbone step).  With `stop_on_all_zeros` the host checks each frame (one sync per frame, like the
        reference's `torch.all(new_frame == 0)`, :662); without it no host sync happens until the end."""
        B, T = input_ids.shape[0], input_ids.shape[1]
        C = self.config.audio_num_codebooks
        if max_new_frames <= 0:
            return torch.zeros(B, 0, C, dtype=torch.long, device=input_ids.device)
        eng = self._ensure_engine(B, T + max_new_frames + 1, max_new_frames, B * T)
        eng.reset()
        self._epoch += 1
        self._frame_pending = False
        eng.set_kv_start(self._kv_starts(attention_mask, B, T))
        eng.prefill(input_ids, attention_mask, want_outputs=False)
        s = eng.sampling(temperature=temperature, topk=topk, seed=self._next_seed() if seed is None else int(seed),
                         row_offset=self.row_offset, per_row_stop=per_row_stop and stop_on_all_zeros)
        n = 0
        if noise is not None or rng is not None:
            # one frame per launch: the engine takes one [B, 32, V] block of draws per call
            if noise is not None and noise.shape[0] < max_new_frames:
                raise ValueError("noise holds fewer frames than max_new_frames")
            while n < max_new_frames:
                # rng: drawn frame by frame, so a run that stops early has consumed what the reference has consumed
                fr = noise[n] if noise is not None else self._torch_rng_noise(rng, B)
                nz = self._check_noise(fr, B).to(eng.device, torch.float32).contiguous()
                s.noise = nz.data_ptr()
                eng.generate(s, 1, self.use_graph)
                eng.sync()
                if stop_on_all_zeros and eng.zero_counts(n, 1)[0] >= B:
                    break
                n += 1
        elif stop_on_all_zeros:
            # The reference syncs once per frame (`torch.all(new_frame == 0)`, :662).  Here every backbone step counts the
            # all-zero rows of its frame on the device; k frames are replayed, k counters read with ONE sync, and the
            # result is cut at the first frame all B rows left empty -- exactly the frames the per-frame test returns
            # (rows are independent; the frames replayed past the cut are discarded).
            k = max(1, int(self.stop_check_interval))
            while n < max_new_frames:
                step = min(k, max_new_frames - n)
                eng.generate(s, step, self.use_graph)
                counts = eng.zero_counts(n, step)
                hit = next((i for i, c in enumerate(counts) if c >= B), None)
                if hit is not None:
                    n += hit
                    break
                n += step
        else:
            eng.generate(s, max_new_frames, self.use_graph)
            n = max_new_frames
        out = eng.read_frames(0, n) if n else torch.zeros(B, 0, C, dtype=torch.long, device=eng.device)
        if per_row_stop:
            first = torch.full((B,), n, dtype=torch.long, device=out.device)
            if n:
                zero = (out == 0).all(dim=2)                              # [B, n]
                first = torch.where(zero.any(dim=1), zero.float().argmax(dim=1), first)
            self.last_row_lengths = first.to(input_ids.device)
        self._epoch += 1  # generate() does not hand out a cache handle
        return out.to(input_ids.device)
