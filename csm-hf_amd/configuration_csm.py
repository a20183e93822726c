"""`CSMConfig` -- same fields, defaults and JSON layout as the reference (`modeling_csm.py:52-143`).

Standalone (no `transformers` import): the two nested Llama configs are plain attribute bags that
read and write the same keys HF's `LlamaConfig.to_dict()` produces, so a `config.json` written by the
reference loads here and vice versa (unknown keys are preserved verbatim).
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any, Dict

_ROPE_LLAMA3 = {
    "type": "llama3",
    "factor": 32.0,
    "low_freq_factor": 1.0,
    "high_freq_factor": 4.0,
    "original_max_position_embeddings": 8192,
}


class LlamaSubConfig:
    """The subset of `LlamaConfig` the generation path reads (reference `modeling_csm.py:68-109`)."""

    _defaults = dict(
        vocab_size=128256, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
        num_attention_heads=32, num_key_value_heads=8, max_position_embeddings=2048,
        rms_norm_eps=1e-5, attention_dropout=0.0, rope_theta=500000.0, rope_scaling=None,
        hidden_act="silu", attention_bias=False, mlp_bias=False, head_dim=None,
        architectures=["LlamaForCausalLM"], model_type="llama",
    )

    def __init__(self, **kw):
        d = copy.deepcopy(self._defaults)
        d.update(copy.deepcopy(kw))
        # transformers>=5 serialises rope settings as `rope_parameters`.
        rp = d.pop("rope_parameters", None)
        if rp and not d.get("rope_scaling"):
            rp = dict(rp)
            if "rope_theta" in rp:
                d["rope_theta"] = rp.pop("rope_theta")
            if "rope_type" in rp:
                rp["type"] = rp.pop("rope_type")
            d["rope_scaling"] = rp if rp.get("type", "default") != "default" else None
        for k, v in d.items():
            setattr(self, k, v)
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads
        if self.hidden_act != "silu":
            raise ValueError(f"only hidden_act='silu' is supported, got {self.hidden_act!r}")
        if self.attention_bias or self.mlp_bias:
            raise ValueError("attention_bias/mlp_bias are not supported (reference uses neither)")
        if self.num_attention_heads % self.num_key_value_heads:
            raise ValueError("num_attention_heads must be a multiple of num_key_value_heads")

    def to_dict(self) -> Dict[str, Any]:
        return copy.deepcopy(self.__dict__)


def _as_sub(cfg) -> LlamaSubConfig:
    if isinstance(cfg, LlamaSubConfig):
        return LlamaSubConfig(**cfg.to_dict())
    if isinstance(cfg, dict):
        return LlamaSubConfig(**cfg)
    if hasattr(cfg, "to_dict"):  # an HF LlamaConfig
        d = cfg.to_dict()
        keep = set(LlamaSubConfig._defaults) | {"rope_parameters"}
        return LlamaSubConfig(**{k: v for k, v in d.items() if k in keep})
    raise TypeError(f"cannot build a Llama sub-config from {type(cfg)}")


class CSMConfig:
    """Reference `CSMConfig` (`modeling_csm.py:52-143`): csm-1b defaults, nested backbone/decoder."""

    model_type = "csm"

    def __init__(self, text_vocab_size=128256, audio_vocab_size=2051, audio_num_codebooks=32,
                 max_seq_len=2048, backbone_config=None, decoder_config=None, **kwargs):
        self.text_vocab_size = text_vocab_size
        self.audio_vocab_size = audio_vocab_size
        self.audio_num_codebooks = audio_num_codebooks
        self.max_seq_len = max_seq_len
        if backbone_config is None:
            backbone_config = dict(hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
                                   num_attention_heads=32, num_key_value_heads=8,
                                   rope_scaling=dict(_ROPE_LLAMA3))
        if decoder_config is None:
            decoder_config = dict(hidden_size=1024, intermediate_size=8192, num_hidden_layers=4,
                                  num_attention_heads=8, num_key_value_heads=2,
                                  rope_scaling=dict(_ROPE_LLAMA3))
        self.backbone_config = _as_sub(backbone_config)
        self.decoder_config = _as_sub(decoder_config)
        # reference modeling_csm.py:128-129,140-141
        self.backbone_config.vocab_size = text_vocab_size
        self.backbone_config.max_position_embeddings = max_seq_len
        self.decoder_config.vocab_size = text_vocab_size
        self.decoder_config.max_position_embeddings = audio_num_codebooks
        self.use_return_dict = kwargs.pop("return_dict", True)
        self.torch_dtype = kwargs.pop("torch_dtype", kwargs.pop("dtype", None))
        self.extra = {k: v for k, v in kwargs.items() if k not in ("model_type", "architectures",
                                                                   "transformers_version")}

    # ---- HF-style (de)serialisation ------------------------------------------------------------
    def to_dict(self) -> Dict[str, Any]:
        d = dict(
            architectures=["CSMModel"], model_type=self.model_type,
            text_vocab_size=self.text_vocab_size, audio_vocab_size=self.audio_vocab_size,
            audio_num_codebooks=self.audio_num_codebooks, max_seq_len=self.max_seq_len,
            backbone_config=self.backbone_config.to_dict(),
            decoder_config=self.decoder_config.to_dict(),
        )
        if self.torch_dtype is not None:
            d["dtype"] = str(self.torch_dtype).replace("torch.", "")
        d.update(self.extra)
        return d

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "CSMConfig":
        return cls(**copy.deepcopy(d))

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    @classmethod
    def from_pretrained(cls, path: str) -> "CSMConfig":
        p = os.path.join(path, "config.json") if os.path.isdir(path) else path
        with open(p) as f:
            return cls.from_dict(json.load(f))

    @classmethod
    def tiny(cls, **over) -> "CSMConfig":
        """Small config for unit tests: same structure as csm-1b (head_dim 64 backbone / 128 decoder,
        GQA, llama3 RoPE, 32 codebooks), ~2.5 M parameters."""
        kw = dict(
            text_vocab_size=211, audio_vocab_size=51, audio_num_codebooks=32, max_seq_len=128,
            backbone_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                 num_attention_heads=4, num_key_value_heads=1,
                                 rope_scaling=dict(_ROPE_LLAMA3)),
            decoder_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                num_attention_heads=2, num_key_value_heads=1,
                                rope_scaling=dict(_ROPE_LLAMA3)),
        )
        kw.update(over)
        return cls(**kw)
