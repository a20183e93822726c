// 32-key attention tile held entirely in registers: every K and V load of the tile is issued before any
// of them is consumed, so a tile costs ONE memory latency instead of a chain of them (the decode caches
// are tiny -- a few KB per head -- so these kernels are latency-, not bandwidth-bound).
//   QK^T : lane = (key t = lane&31, dim-half = lane>>5), HD/8 16-byte loads from the position-major K layout
//   PV   : lane = (dim group dg = lane % (HD/4), key phase tpar = lane / (HD/4)), HD/8 16-byte row loads
// Softmax statistics by wavefront shuffles; p is exchanged through a 32-float wave-private LDS strip.
#pragma once
#include "common.h"

template <typename KT>
__device__ __forceinline__ f32x4 ld_k4(const KT* p);
template <>
__device__ __forceinline__ f32x4 ld_k4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <>
__device__ __forceinline__ f32x4 ld_k4<bf16_t>(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  f32x4 r;
  r[0] = bf16_lo(u.x); r[1] = bf16_hi(u.x); r[2] = bf16_lo(u.y); r[3] = bf16_hi(u.y);
  return r;
}

template <typename KT, int HD>
struct AttnTile32 {
  static constexpr int NK = HD / 8;        // 16-byte loads per lane for K, and for V
  static constexpr int LPR = HD / 4;       // lanes per V row
  static constexpr int TP = 64 / LPR;      // key phases in the PV step
  f32x4 k[NK], v[NK];

  // keys t0 .. t0+cnt-1 (1 <= cnt <= 32) of one kv-head; kc/vc point at that head's cache
  __device__ __forceinline__ void load(const KT* kc, const KT* vc, int lmax, int t0, int cnt, int lane) {
    const int t = lane & 31, half = lane >> 5;
    const size_t tc = t0 + (t < cnt ? t : cnt - 1);
#pragma unroll
    for (int i = 0; i < NK; ++i) k[i] = ld_k4<KT>(kc + ((size_t)(half * NK + i) * lmax + tc) * 4);
    const int dg = lane % LPR, tpar = lane / LPR;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int tt = tpar + TP * i;
      const size_t tv = t0 + (tt < cnt ? tt : cnt - 1);
      v[i] = ld_k4<KT>(vc + tv * HD + 4 * dg);
    }
  }

  // online-softmax update for one query head; qh = that head's q in LDS, pbuf = 32-float wave-private strip
  __device__ __forceinline__ void accumulate(const float* qh, float* pbuf, int cnt, int lane, float& m_run,
                                             float& l_run, f32x4& acc) const {
    const int t = lane & 31, half = lane >> 5;
    // pair arithmetic, two independent accumulator pairs: with one wave per SIMD (decoder attention) nothing hides
    // the latency of a 64-deep dependent FMA chain; this one is 8 deep
    f32x2 sa = f32x2{0.f, 0.f}, sb = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(qh + (half * NK + i) * 4);
      if (i & 1) {
        sb = PKFMA((f32x2{qv[0], qv[1]}), (f32x2{k[i][0], k[i][1]}), sb);
        sb = PKFMA((f32x2{qv[2], qv[3]}), (f32x2{k[i][2], k[i][3]}), sb);
      } else {
        sa = PKFMA((f32x2{qv[0], qv[1]}), (f32x2{k[i][0], k[i][1]}), sa);
        sa = PKFMA((f32x2{qv[2], qv[3]}), (f32x2{k[i][2], k[i][3]}), sa);
      }
    }
    float s = (sa[0] + sa[1]) + (sb[0] + sb[1]);
    s = xor32_sum(s);
    const bool valid = t < cnt;
    if (!valid) s = -INFINITY;
    const float m_new = fmaxf(m_run, wave_max(s));
    const float p = valid ? __expf(s - m_new) : 0.f;
    const float alpha = __expf(m_run - m_new);  // first tile: exp(-inf) = 0
    l_run = l_run * alpha + wave_sum(half == 0 ? p : 0.f);
    m_run = m_new;
    __builtin_amdgcn_wave_barrier();
    if (half == 0) pbuf[t] = p;
    __builtin_amdgcn_wave_barrier();
    const int tpar = lane / LPR;
    acc *= alpha;
    f32x2 a01 = f32x2{acc[0], acc[1]}, a23 = f32x2{acc[2], acc[3]};
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const float pv = pbuf[tpar + TP * i];
      const f32x2 p2 = f32x2{pv, pv};
      a01 = PKFMA(p2, (f32x2{v[i][0], v[i][1]}), a01);
      a23 = PKFMA(p2, (f32x2{v[i][2], v[i][3]}), a23);
    }
    acc[0] = a01[0]; acc[1] = a01[1]; acc[2] = a23[0]; acc[3] = a23[1];
    __builtin_amdgcn_wave_barrier();
  }

  // sum the key phases; afterwards lanes < LPR hold dims 4*lane .. 4*lane+3
  static __device__ __forceinline__ f32x4 reduce(f32x4 acc) {
    if (LPR == 16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = xor16_sum(acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = xor32_sum(acc[i]);
    return acc;
  }
};
