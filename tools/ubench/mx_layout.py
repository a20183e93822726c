#!/usr/bin/env python
"""Which 32-element scale block does v_mfma_scale_f32_16x16x128_f8f6f4 apply to which byte of a lane's operand?  One-hot
operands at k0 through csm_gemm_mx with per-block activation scales 4^b: the product reveals the block.  Result on MI355X
(round 3), with a lane (row, g) fed 32 CONSECUTIVE k (chunks 2g, 2g + 1): k 0-15 -> block 0, 16-31 -> block 2, 32-47 -> 0,
48-63 -> 2, 64-79 -> 1, 80-95 -> 3, ...: the instruction's lane (row, g) holds k = 16 g .. 16 g + 15 in registers 0-3 and
k = 64 + 16 g .. in registers 4-7, and takes the scale of block b from lane row + 16 b.  gemm_mx_kernel therefore gives lane
g the chunks g and g + 4; with that this script prints block k0 // 32 for every k0."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from csm_hf_amd import CSMConfig
from csm_hf_amd.engine import Engine
from csm_hf_amd.synth import synth_state_dict

cfg = CSMConfig.tiny()
eng = Engine(cfg, synth_state_dict(cfg), "cuda:0", torch.float32, max_batch=1, max_len=64, max_frames=4, max_prefill_rows=128)
R = N = K = 128
s1 = torch.full((R, K // 32), 127, dtype=torch.uint8)
sA = s1.clone()
for b in range(4):
    sA[:, b] = 127 + 2 * b
res = []
for k0 in range(0, K, 4):
    w = torch.zeros(N, K, dtype=torch.uint8)
    a = torch.zeros(R, K, dtype=torch.uint8)
    w[:, k0] = a[:, k0] = 0x38                     # e4m3 1.0
    v = float(eng.k_gemm_mx(w, s1, a, sA)[3, 5])
    res.append((k0, int(round(math.log(v, 4))) if v > 0 else -1))
print(res)
print("scale block == k0 // 32 for every k0:", all(b == k0 // 32 for k0, b in res))
