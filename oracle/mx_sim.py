"""CPU restatement of OCP MX-fp8 (microscaling) quantisation and of a context prefill whose linears multiply MX-quantised
activations by MX-quantised weights -- TEST INFRASTRUCTURE for `prefill_precision = "mxfp8"` (csrc/gemm_mx.h); imported by
tests/ only.  There is no reference counterpart (the reference runs bf16, README.md:73): the format is the published OCP
Microscaling Formats (MX) v1.0 specification -- blocks of 32 along the reduction dimension, one shared E8M0 scale
2^(floor(log2(amax)) - emax_elem) per block (emax_elem = 8 for e4m3), elements = x / scale converted to e4m3 with
saturation; a product of two MX vectors is the exact dot product of the dequantised values.  Parity of the engine's path is
therefore pinned against THIS restatement wrapped around the oracle's own linears (csm_oracle.py), which is itself pinned
against the reference."""
import torch
import torch.nn.functional as F

FP8_MAX = 448.0


def mx_quantize(t: torch.Tensor):
    """-> (e4m3 bytes [..., K] uint8, E8M0 bytes [..., K/32] uint8)"""
    K = t.shape[-1]
    x = t.float().reshape(*t.shape[:-1], K // 32, 32)
    amax = x.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(2.0 ** -126))).to(torch.int32) - 8
    eb = (e + 127).clamp(0, 254)
    scale = torch.exp2((eb - 127).float())
    q = (x / scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(t.shape), eb.to(torch.uint8).reshape(*t.shape[:-1], K // 32)


def mx_dequantize(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    K = q.shape[-1]
    v = q.view(torch.float8_e4m3fn).float().reshape(*q.shape[:-1], K // 32, 32)
    return (v * torch.exp2(s.float() - 127.0).unsqueeze(-1)).reshape(q.shape)


def mx_round(t: torch.Tensor) -> torch.Tensor:
    return mx_dequantize(*mx_quantize(t))


class mx_linears:
    """context manager: inside it, every `F.linear` of the oracle whose weight is one of `weights` (by data_ptr) multiplies
    MX-rounded activations by the MX-rounded weight; `act=False` keeps the activations (weights-only rounding)."""

    def __init__(self, oracle_module, weights, act=True):
        self.O, self.keys, self.act, self.cache = oracle_module, {w.data_ptr() for w in weights}, act, {}

    def __enter__(self):
        self.orig = self.O.F.linear

        def lin(x, w, b=None):
            if w.data_ptr() not in self.keys:
                return self.orig(x, w, b)
            k = w.data_ptr()
            if k not in self.cache:
                self.cache[k] = mx_round(w)
            return self.orig(mx_round(x) if self.act else x, self.cache[k], b)
        self.O.F.linear = lin
        return self

    def __exit__(self, *a):
        self.O.F.linear = self.orig
