// KV-cached GQA attention for single-query rows (decode; also used row-by-row for prefill v1).
//
// Replaces: transformers sdpa_attention_forward as reached from LlamaAttention.forward
// (sdpa_attention.py:97-163, modeling_llama.py:254-281) for q_len == 1 (all cached keys visible)
// and, with per-row positions, the causal prefill (key t visible to query at position p iff
// kv_start[b] <= t <= p).  Softmax in fp32 (as SDPA), GQA: q-head h reads kv-head h / (n_q/n_kv).
//
// Cache layout (written by the QKV epilogue, gemv.h / rope_scatter):
//   K [B][n_kv][hd/4][lmax][4]  -- position-major inside a 4-dim group, so that "lane = key position"
//                                  loads are 16-byte-per-lane coalesced (QK^T phase)
//   V [B][n_kv][lmax][hd]       -- "lane = output dim" loads are coalesced (PV phase)
// Roofline: HBM/L2 (KV streaming): algorithmic bytes = kv_len * n_kv * hd * 2 * sizeof(KT) per row.
// One workgroup = (row, kv-head, split); its 4 waves take the G = n_q/n_kv query heads that share the
// kv-head, so each K/V byte is fetched from HBM once per workgroup and re-served by L1/L2.
// Wavefront shuffles carry the softmax reductions; no LDS except the broadcast copy of q.
#pragma once
#include "common.h"

struct AttnArgs {
  const float* q;  // [rows][n_q*hd], pre-scaled by hd^-0.5, RoPE applied
  const void* kcache;
  const void* vcache;
  int n_q, n_kv, hd, lmax;
  const int* row_seq;   // nullable: row -> sequence slot (default row)
  const int* row_pos;   // nullable: per-row query position
  const int* pos_ptr;   // device scalar position (backbone decode) or
  int pos_const;        // constant
  const int* kv_start;  // nullable: per-sequence first valid key (left padding)
  int nsplit;
  float* out;   // [rows][n_q*hd]                  (nsplit == 1)
  float* part;  // [rows][n_q][nsplit][hd+2]       (nsplit > 1): acc[hd], m, l
};

#ifndef CSM_ARGS_ONLY
template <typename KT>
struct K4 {};
template <>
struct K4<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
};
template <>
struct K4<bf16_t> {
  static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    f32x4 r;
    r[0] = bf16_lo(u.x);
    r[1] = bf16_hi(u.x);
    r[2] = bf16_lo(u.y);
    r[3] = bf16_hi(u.y);
    return r;
  }
};

// HD = head_dim (64 or 128); each lane owns HD/64 output dims.
template <typename KT, int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a) {
  constexpr int DPL = HD / 64;
  __shared__ __attribute__((aligned(16))) float qs[16 * HD];  // up to 16 q-heads per kv-head
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = a.n_q / a.n_kv;
  int blk = blockIdx.x;
  const int sp = blk % a.nsplit;
  blk /= a.nsplit;
  const int j = blk % a.n_kv;
  const int row = blk / a.n_kv;
  const int b = a.row_seq ? a.row_seq[row] : row;
  const int pos = a.row_pos ? a.row_pos[row] : (a.pos_ptr ? *a.pos_ptr : a.pos_const);
  const int t_lo0 = a.kv_start ? a.kv_start[b] : 0;
  const int len = pos + 1 - t_lo0;
  int span = (len + a.nsplit - 1) / a.nsplit;
  span = (span + 15) & ~15;
  const int t_lo = t_lo0 + sp * span;
  int t_hi = t_lo + span;
  if (t_hi > pos + 1) t_hi = pos + 1;

  for (int i = tid; i < G * HD; i += 256) qs[i] = a.q[(size_t)row * a.n_q * HD + (size_t)j * G * HD + i];
  __syncthreads();

  const KT* kc = reinterpret_cast<const KT*>(a.kcache) + ((size_t)b * a.n_kv + j) * (size_t)(HD / 4) * a.lmax * 4;
  const KT* vc = reinterpret_cast<const KT*>(a.vcache) + ((size_t)b * a.n_kv + j) * (size_t)a.lmax * HD;

  for (int g = wave; g < G; g += 4) {
    const float* qg = qs + g * HD;
    float m_run = -INFINITY, l_run = 0.f;
    float acc[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] = 0.f;

    for (int t0 = t_lo; t0 < t_hi; t0 += 64) {
      const int t = t0 + lane;
      const bool valid = t < t_hi;
      const int tc = valid ? t : t_hi - 1;
      float s = 0.f;
#pragma unroll 8
      for (int d4 = 0; d4 < HD / 4; ++d4) {
        const f32x4 kv = K4<KT>::load(kc + ((size_t)d4 * a.lmax + tc) * 4);
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qg + d4 * 4);
        s = fmaf(qv[0], kv[0], s);
        s = fmaf(qv[1], kv[1], s);
        s = fmaf(qv[2], kv[2], s);
        s = fmaf(qv[3], kv[3], s);
      }
      if (!valid) s = -INFINITY;
      const float m_new = fmaxf(m_run, wave_max(s));
      const float p = valid ? __expf(s - m_new) : 0.f;
      const float alpha = __expf(m_run - m_new);  // first tile: exp(-inf) = 0
      l_run = l_run * alpha + wave_sum(p);
#pragma unroll
      for (int i = 0; i < DPL; ++i) acc[i] *= alpha;
      m_run = m_new;
      const int cnt = min(64, t_hi - t0);
      const KT* vrow = vc + (size_t)t0 * HD;
#pragma unroll 4
      for (int tt = 0; tt < cnt; ++tt) {
        const float pv = __shfl(p, tt, 64);
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[i] = fmaf(pv, to_f32(vrow[(size_t)tt * HD + lane + 64 * i]), acc[i]);
      }
    }
    const int h = j * G + g;
    if (a.nsplit == 1) {
      const float inv = 1.f / l_run;
#pragma unroll
      for (int i = 0; i < DPL; ++i) a.out[(size_t)row * a.n_q * HD + (size_t)h * HD + lane + 64 * i] = acc[i] * inv;
    } else {
      float* pp = a.part + (((size_t)row * a.n_q + h) * a.nsplit + sp) * (HD + 2);
#pragma unroll
      for (int i = 0; i < DPL; ++i) pp[lane + 64 * i] = acc[i];
      if (lane == 0) {
        pp[HD] = m_run;
        pp[HD + 1] = l_run;
      }
    }
  }
}

// merge the per-split partials: out[row][h][d] = sum_s acc_s e^{m_s-M} / sum_s l_s e^{m_s-M}
template <int HD>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* part, int n_q, int nsplit, float* out) {
  const int row = blockIdx.x;
  for (int i = threadIdx.x; i < n_q * HD; i += 256) {
    const int h = i / HD, d = i - h * HD;
    const float* pp = part + ((size_t)row * n_q + h) * nsplit * (HD + 2);
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, pp[s * (HD + 2) + HD]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < nsplit; ++s) {
      const float ms = pp[s * (HD + 2) + HD];
      const float w = (ms == -INFINITY) ? 0.f : __expf(ms - M);
      num = fmaf(pp[s * (HD + 2) + d], w, num);
      den = fmaf(pp[s * (HD + 2) + HD + 1], w, den);
    }
    out[(size_t)row * n_q * HD + i] = num / den;
  }
}

#endif  // CSM_ARGS_ONLY
int launch_attn(hipStream_t st, int kvdtype, int rows, const AttnArgs& a);
