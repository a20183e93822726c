// Flash-style causal GQA attention for prefill rows on the matrix cores (head_dim 64, up to 4 q-heads per
// kv-head), fp32 throughout (v_mfma_f32_32x32x2_f32 for Q K^T and for P V: bitwise fmaf chains), online
// softmax in fp32 like SDPA.  Replaces sdpa_attention_forward for q_len > 1 (transformers
// sdpa_attention.py:97-163: is_causal path, and the boolean-mask path for left-padded rows).
//
// One workgroup = (32 query positions, one kv-head, one sequence); wave g = query head j*G+g.  K and V tiles
// of 32 keys are staged ONCE per workgroup in LDS (shared by the 4 query heads); Q and P live in registers.
// Roofline: fp32 MFMA (157 TFLOP/s): 4*S^2*hd*n_q/2 flops per sequence per layer.
#pragma once
#include "common.h"

struct PrefillAttnArgs {
  const float* q;  // [B*S][n_q*64], pre-scaled by hd^-0.5, RoPE applied
  const void* kcache;
  const void* vcache;
  int n_q, n_kv, lmax;
  int S, past;          // row r = b*S + s sits at position past + s
  const int* kv_start;  // nullable: first valid key of each sequence (left padding)
  float* out;           // [B*S][n_q*64]
  bf16_t* oplanes;      // nullable: output as row-major planes [3][B*S][n_q*64] for the o_proj GEMM instead of `out`
  size_t plane_stride;
};

#ifndef CSM_ARGS_ONLY
// Both products are computed TRANSPOSED so that a lane owns one query row end to end:
//   S^T[key][row] = K Q^T   (A = K tile from LDS, B = Q from registers)  -> lane (row = lane&31, half) holds 16 keys
//   O^T[d][row]   = V^T P^T (A = V tile from LDS, B = P straight from the score accumulator registers)
// The contraction over keys in the second product runs in the order the score accumulator already holds them
// (key(t, half) = (t&3) + 8*(t>>2) + 4*half), so P never leaves its registers: no LDS transpose, and the softmax
// statistics are 16 in-lane max/add plus ONE exchange with the partner lane (lane ^ 32).
template <typename KT>
__global__ __launch_bounds__(256) void attn_prefill_kernel(PrefillAttnArgs a) {
  constexpr int HD = 64;
  __shared__ float Ks[32][HD + 1];
  __shared__ __attribute__((aligned(16))) float Vs[32][HD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = a.n_q / a.n_kv;
  const int qt = blockIdx.x, j = blockIdx.y, b = blockIdx.z;
  const int s0 = qt * 32;
  const int li = lane & 31, lh = lane >> 5;
  const bool head_live = wave < G;
  const int h = j * G + (head_live ? wave : 0);
  const int kv_lo = a.kv_start ? a.kv_start[b] : 0;
  const int s_last = min(a.S - 1, s0 + 31);
  const int kmax = a.past + s_last;  // last key any row of this tile may see
  const int s = s0 + li;             // this lane's query row
  const bool row_live = s < a.S && head_live;
  const int row_kmax = a.past + s;

  // Q in the B-operand layout of 32x32x2 (= the A layout): lane holds Q[row li][dim 2t + lh], t = 0..31
  float qreg[32];
  {
    const float* qrow = a.q + ((size_t)b * a.S + (s < a.S ? s : a.S - 1)) * a.n_q * HD + (size_t)h * HD + lh;
#pragma unroll
    for (int t = 0; t < 32; ++t) qreg[t] = row_live ? qrow[2 * t] : 0.f;
  }
  const KT* kc = reinterpret_cast<const KT*>(a.kcache) + ((size_t)b * a.n_kv + j) * (size_t)(HD / 4) * a.lmax * 4;
  const KT* vc = reinterpret_cast<const KT*>(a.vcache) + ((size_t)b * a.n_kv + j) * (size_t)a.lmax * HD;

  f32x16 o0 = (f32x16)(0.f), o1 = (f32x16)(0.f);   // O^T tiles: d = (reg&3) + 8*(reg>>2) + 4*lh (+32), row = li
  float m_run = -INFINITY, l_run = 0.f;

  for (int kt0 = kv_lo & ~31; kt0 <= kmax; kt0 += 32) {
    // ---- stage the K / V tile (keys kt0 .. kt0+31) once for the 4 query heads -------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int d4 = idx >> 5, t = idx & 31;
      const int key = min(kt0 + t, a.lmax - 1);
      const KT* src = kc + ((size_t)d4 * a.lmax + key) * 4;
      Ks[t][d4 * 4 + 0] = to_f32(src[0]);
      Ks[t][d4 * 4 + 1] = to_f32(src[1]);
      Ks[t][d4 * 4 + 2] = to_f32(src[2]);
      Ks[t][d4 * 4 + 3] = to_f32(src[3]);
      const int tv = idx >> 4, c4 = idx & 15;
      const KT* vsrc = vc + (size_t)min(kt0 + tv, a.lmax - 1) * HD + c4 * 4;
      f32x4 vv;
      vv[0] = to_f32(vsrc[0]); vv[1] = to_f32(vsrc[1]); vv[2] = to_f32(vsrc[2]); vv[3] = to_f32(vsrc[3]);
      *reinterpret_cast<f32x4*>(&Vs[tv][c4 * 4]) = vv;
    }
    __syncthreads();
    // ---- S^T[key][row] = sum_d K[key][d] Q[row][d] --------------------------------------------------------
    f32x16 sc = (f32x16)(0.f);
#pragma unroll
    for (int t = 0; t < 32; ++t) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[li][2 * t + lh], qreg[t], sc, 0, 0, 0);
    // accumulator layout: column = query row li, accumulator row r -> key (r&3) + 8*(r>>2) + 4*lh
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const bool valid = row_live && key >= kv_lo && key <= row_kmax;
      sc[r] = valid ? sc[r] : -INFINITY;
      mx = fmaxf(mx, sc[r]);
    }
    mx = xor32_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const bool any = m_new > -INFINITY;
    const float alpha = any ? __expf(m_run - m_new) : 1.f;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc[r] = any ? __expf(sc[r] - m_new) : 0.f;   // exp(-inf) = 0 for masked keys
      sum += sc[r];
    }
    sum = xor32_sum(sum);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    o0 *= alpha;
    o1 *= alpha;
    // ---- O^T[d][row] += sum_key V[key][d] P[row][key], keys in accumulator order -----------------------------
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int kk = (t & 3) + 8 * (t >> 2) + 4 * lh;
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kk][li], sc[t], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kk][32 + li], sc[t], o1, 0, 0, 0);
    }
    __syncthreads();   // the K/V tile is rewritten next iteration
  }
  if (!row_live) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;   // a fully masked (pad) row yields zeros
  float* dst = a.out + ((size_t)b * a.S + s) * a.n_q * HD + (size_t)h * HD;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {   // accumulator rows 4*r4 .. 4*r4+3 -> 4 consecutive d
    const int d = 8 * r4 + 4 * lh;
    f32x4 v0, v1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v0[i] = o0[4 * r4 + i] * inv; v1[i] = o1[4 * r4 + i] * inv; }
    if (a.oplanes) {
      bf16_t* pd = a.oplanes + ((size_t)b * a.S + s) * a.n_q * HD + (size_t)h * HD;
      store_rowplanes4(pd + d, a.plane_stride, v0);
      store_rowplanes4(pd + 32 + d, a.plane_stride, v1);
    } else {
      *reinterpret_cast<f32x4*>(dst + d) = v0;
      *reinterpret_cast<f32x4*>(dst + 32 + d) = v1;
    }
  }
}
#endif  // CSM_ARGS_ONLY

// returns -2 when the shape is not covered (head_dim != 64 or more than 4 q-heads per kv-head)
int launch_attn_prefill(hipStream_t st, int kvdtype, int B, int hd, const PrefillAttnArgs& a);
