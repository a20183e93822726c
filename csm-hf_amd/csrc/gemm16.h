// Skinny GEMM on the matrix cores for batched decode: y[M<=16, N] = f(x)[M,K] @ W[N,K]^T, bf16 or fp8 weights
// (fp8 e4m3 is widened to bf16 in registers -- exact -- and scaled per output row in the epilogue).
//
// Roofline: HBM (weights N*K*2 bytes read once per launch, independent of M).  Each 16x16x32 MFMA consumes
// a 16-row x 32-k weight fragment straight from registers (A operand) against the activations (B operand);
// no LDS round trip for the weights.  Activations stay fp32-exact: x is carried as three bf16 parts
// (x = hi + mid + lo by truncation, 3 x 8 = 24 mantissa bits, no rounding) and every weight fragment is multiplied
// by all three (bf16 x bf16 products are exact in fp32), so the result matches the M<=4 fp32-FMA kernels to fp32
// summation-order error; the matrix pipe has >8x headroom over the HBM stream even at 3 MFMAs per fragment.
// Three variants of the operand paths (template flags TL, XP, documented at the kernel):
//   row-major weights + fp32 x split in the kernel (k permuted so a lane reads 64 contiguous bytes of its row),
//   fragment-order weights (one coalesced 1 KiB load per fragment) + fp32 x split in the kernel,
//   fragment-order weights + activation planes written by the producing launch (no arithmetic ahead of the MFMAs).
// A workgroup owns a panel of PT 16-row tiles; its NW waves take one 128-wide k chunk each and meet in LDS.
//
// Same prologues/epilogues as gemv.h (plain | RMSNorm ; store | residual | SwiGLU | RoPE + KV append).
#pragma once
#include "gemv.h"

#ifndef CSM_ARGS_ONLY
typedef __attribute__((ext_vector_type(8))) short bf16x8;

// x = hi + mid + lo EXACTLY, each part a bf16: truncation splits fp32's 24 mantissa bits 8 + 8 + 8 and the
// two subtractions are exact, so no rounding happens anywhere (cheaper than three RNE conversions).
__device__ __forceinline__ void split3(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  uint32_t* ph = reinterpret_cast<uint32_t*>(&hi);
  uint32_t* pm = reinterpret_cast<uint32_t*>(&mid);
  uint32_t* pl = reinterpret_cast<uint32_t*>(&lo);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float v0 = p < 2 ? a[2 * p] : b[2 * p - 4], v1 = p < 2 ? a[2 * p + 1] : b[2 * p - 3];
    const uint32_t h0 = __float_as_uint(v0) & 0xffff0000u, h1 = __float_as_uint(v1) & 0xffff0000u;
    const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
    const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);
    ph[p] = (h0 >> 16) | h1;
    pm[p] = (m0 >> 16) | m1;
    pl[p] = (__float_as_uint(s0) >> 16) | (__float_as_uint(s1) & 0xffff0000u);
  }
}

// A-operand fragment (8 consecutive k of one weight row) as bf16x8, from bf16 or fp8 storage (exact widening)
template <typename WT>
struct AFrag;
template <>
struct AFrag<bf16_t> {
  u32x4 r;
  __device__ __forceinline__ void load(const bf16_t* p, int nt) {
    if (nt) r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    else r = *reinterpret_cast<const u32x4*>(p);
  }
  __device__ __forceinline__ bf16x8 get() const {
    bf16x8 f;
    *reinterpret_cast<u32x4*>(&f) = r;
    return f;
  }
};
template <>
struct AFrag<fp8_t> {
  uint2 r;
  __device__ __forceinline__ void load(const fp8_t* p, int nt) {
    if (nt) {
      const uint64_t u = __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(p));
      r.x = (uint32_t)u;
      r.y = (uint32_t)(u >> 32);
    } else {
      r = *reinterpret_cast<const uint2*>(p);
    }
  }
  __device__ __forceinline__ bf16x8 get() const {  // e4m3 -> fp32 (exact) -> bf16 by truncation (exact: 4 significant bits)
    bf16x8 f;
    uint32_t* pf = reinterpret_cast<uint32_t*>(&f);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int w = (int)(h < 2 ? r.x : r.y);   // word_sel must be a literal
      const f32x2 v = (h & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8(w, true) : __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
      pf[h] = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xffff0000u);
    }
    return f;
  }
};

// panel row of tile t, local row r (0..15).  QKV panels pair the two RoPE halves of a head (PT == 2).
template <int EPI, int PT>
__device__ __forceinline__ int g16_row(const GemvArgs& a, int panel, int t, int r) {
  if (EPI == EPI_QKV) {
    const int half = a.hd >> 1, spp = half / 16;  // sub-panels per head
    const int head = panel / spp, s = panel - head * spp;
    return head * a.hd + t * half + s * 16 + r;
  }
  return (panel * PT + t) * 16 + r;
}

// One 128-wide k chunk per wave: every load of the wave (its x slice, the norm weights, PT*4 weight
// fragments, epilogue inputs) is issued before anything is consumed, in the order of consumption.  grid = (row panels, KB); KB > 1 splits K across
// workgroups (K = 8192 down_proj): partial panels go to `slabs`, a per-panel ticket elects the last arriver,
// which sums them in fixed order (deterministic) and runs the epilogue.  PRO_NORM needs KB == 1.
// TL = weights read from the fragment-order copy a.Wt (launchers.hip: tile16_kernel) and k taken in natural
// order (lane group g, step j -> k = j*32 + g*8 + 0..7): a fragment load is then one contiguous 1 KiB (bf16)
// block per wavefront and an x load touches 16 half-lines; the row-major variant (TL = false, permuted k) is kept
// for unbound weights (csm_gemv / csm_gemm hooks) and as the A/B baseline.
// XP = activations arrive as planes (GemvArgs::xplanes): the B operands are loaded ready-made (12 coalesced 1 KiB
// loads per wave instead of 8 + 8 half-line loads of x and norm weights), there is no RMS exchange, no scaling and no
// split arithmetic ahead of the MFMAs; the RMS scale of row m is applied in the epilogue from the producer's per-tile
// sums of squares.  An ablation without the x / norm-weight loads and their arithmetic ran the B = 16 step in 4.57 ms
// instead of 5.75 ms -- that chain, not the weight stream, was the critical path of the small launches.  XP needs TL.
// NOTE on the 16-wave (1024-thread) variants: at 113-128 registers their four waves per SIMD take the whole 512-entry
// register file of every SIMD of a CU, so such a workgroup only starts on a CU with nothing else resident.  The weight
// streamer (prefetch.h) keeps a wave on every SIMD of every CU, so a frame-step that contains one of these launches is
// not streamed (launch_g16 reports `exclusive`); capping them at 96 registers (5 waves per SIMD) spills 24-58 registers.
// ONE (with XP) = decode_precision bf16: the planes hold ONE nearest-even bf16 value per activation (GemvArgs::pl1): a third of the
// plane loads, one MFMA per weight fragment, one-plane stores in the epilogue.  A template flag: as a run-time branch it cost the
// exact mode 3 % at B = 16 / 64 and moved last bits of the epilogue arithmetic.
// Kernel-argument preload (gemv.h GEMV_HOT_PARAMS has the why): the 14 leading dwords carry what a wave needs to request its activation
// planes and weight fragments -- weights (fragment-order copy if TL), activations (planes if XP), launch counter, position source
// (EPI_QKV: row_pos if there is one, else pos_ptr; other epilogues: out), norm weights, N, K, a packed word and pos_const.
//   packed: bit 0 nt, 1 prio, 2 kfast, 3 the position source is row_pos; bits 4-8 M, 9-13 KB, 14-18 hd / 8, 19-25 n_q (<= 127), 26-31 n_kv (<= 63)
#define G16_HOT_PARAMS const void* hW, const void* hx, unsigned* hprog, void* hp3, const float* hln, int hN, int hK, uint32_t hpk, int hi3
#define G16_HOT_ARGS(a, M_, KB_, TL_, XP_, EPI_)                                                                                              \
  ((TL_) ? (a).Wt : (a).W), ((XP_) ? (const void*)(a).xplanes : (const void*)(a).x), (a).prog,                                                 \
  ((EPI_) == EPI_QKV ? (void*)((a).row_pos ? (a).row_pos : (a).pos_ptr) : (void*)(a).out), (a).ln, (a).N, (a).K,                               \
  (uint32_t)(((a).nt ? 1u : 0u) | ((a).prio ? 2u : 0u) | ((a).kfast ? 4u : 0u) | ((a).row_pos ? 8u : 0u) | ((uint32_t)(M_) << 4) |             \
             ((uint32_t)(KB_) << 9) | ((((uint32_t)(a).hd >> 3) & 31u) << 14) | (((uint32_t)(a).n_q & 127u) << 19) | (((uint32_t)(a).n_kv & 63u) << 26)), \
  (a).pos_const
template <typename WT, typename KT, int PRO, int EPI, int NW, int PT, bool TL, bool XP, bool ONE = false>
__global__ __launch_bounds__(64 * NW) void gemm16_kernel(G16_HOT_PARAMS, GemvArgs a, float* slabs, int* tickets) {
  if (TL) a.Wt = hW; else a.W = hW;
  if (XP) a.xplanes = reinterpret_cast<const bf16_t*>(hx); else a.x = reinterpret_cast<const float*>(hx);
  a.prog = hprog; a.ln = hln; a.N = hN; a.K = hK;
  a.nt = (int)(hpk & 1u); a.prio = (int)((hpk >> 1) & 1u); a.kfast = (int)((hpk >> 2) & 1u);
  const int M = (int)((hpk >> 4) & 31u), KB = (int)((hpk >> 9) & 31u);
  if (EPI == EPI_QKV) {
    a.pos_const = hi3;
    if (hpk & 8u) { a.row_pos = reinterpret_cast<const int*>(hp3); a.pos_ptr = nullptr; }
    else { a.row_pos = nullptr; a.pos_ptr = reinterpret_cast<const int*>(hp3); }
    a.hd = (int)((hpk >> 14) & 31u) << 3; a.n_q = (int)((hpk >> 19) & 127u); a.n_kv = (int)(hpk >> 26);
  } else {
    a.out = reinterpret_cast<float*>(hp3);
  }
  if constexpr (!std::is_same<WT, fp8_t>::value) a.wscale = nullptr;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // red[NW][PT][256] | panel[PT][256] | flag[16] | stat[NW][16]
  float* red = lds;
  float* panel = lds + NW * PT * 256;
  int* flag = reinterpret_cast<int*>(panel + PT * 256);
  float* stat = panel + PT * 256 + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // K split across workgroups with a.kfast: the k split is the FASTEST grid index (launcher: dim3(KB, panels)), so that the workgroups that
  // read the same k slice of the activation planes run on one XCD (workgroup id mod 8) and share it in that L2 (round 5, gemm128.h)
  const bool kf = KB > 1 && a.kfast;
  const int bix = kf ? (int)blockIdx.y : (int)blockIdx.x, biy = kf ? (int)blockIdx.x : (int)blockIdx.y, gdx = kf ? (int)gridDim.y : (int)gridDim.x;
  (void)gdx;
  TL_BEGIN(a.dbg);
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  if (a.prog && bix == 0 && biy == 0 && tid == 0) atomicAdd(a.prog, 1u);   // weight streamer pacing
  const int K = a.K;
  const WT* W = reinterpret_cast<const WT*>(a.W);
  const int m = lane & 15, g = lane >> 4;
  const bool mlive = m < M;
  const int chunk = biy * NW + wave;          // 128-wide k chunk of this wave
  const int k0 = chunk * 128 + (TL ? g * 8 : g * 32);     // first k of this lane; step j adds (TL ? 32 : 8) * j
  constexpr int KJ = TL ? 32 : 8;

  // Load order matters: vmcnt retires in issue order, so whatever is consumed first must be requested first.
  // 1. the x slice (+ norm weights): L2 hits, consumed by the RMS statistic and the bf16 split while the weights
  //    are still in flight;  2. the weight fragments (HBM);  3. the residual values, consumed last.
  // Epilogue inputs of this thread's elements.  EPI_QKV: requested first of all (the cos/sin addresses depend on the
  // position loaded here).  EPI_RESID: requested LAST, behind the weights -- the residual values were written by
  // another XCD a few launches ago and come from HBM; requested first they would hold up the x slice behind them
  // (measured at M = 16: backbone down_proj 17.7 us first vs 12.3 us last).
  // EPI_RESID / EPI_SWIGLU run a quad epilogue: thread (tile t, C-layout lane l) owns the 4 consecutive output
  // columns of its accumulator registers -- 16-byte residual loads / stores, packed plane stores.
  constexpr bool QUAD = (EPI == EPI_RESID || EPI == EPI_SWIGLU);
  static_assert(!QUAD || PT * 64 <= 64 * NW, "one quad per thread");
  f32x4 rq = (f32x4)(0.f), lq = (f32x4)(1.f);
  constexpr int NE = (PT * 256 + 64 * NW - 1) / (64 * NW);
  float pre0[NE], pre1[NE];
  int ppos[NE];
  auto prefetch_epi = [&]() {
    if (QUAD) {
      if (EPI == EPI_RESID && tid < PT * 64) {
        const int t = tid >> 6, l = tid & 63, mm = l & 15;
        const int n0 = g16_row<EPI, PT>(a, bix, t, (l >> 4) * 4);
        if (mm < M && n0 < a.N) {
          rq = *reinterpret_cast<const f32x4*>(a.out + (size_t)mm * a.ldo + n0);
          if (a.oplanes && a.oln) lq = *reinterpret_cast<const f32x4*>(a.oln + n0);   // the consumer's norm weight
        }
      }
      return;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      pre0[e] = pre1[e] = 0.f;
      ppos[e] = 0;
      const int i = tid + e * 64 * NW;
      if (i < PT * 256) {
        const int t = i >> 8, l = (i >> 2) & 63, reg = i & 3;
        const int mm = l & 15, r = (l >> 4) * 4 + reg;
        const int n = g16_row<EPI, PT>(a, bix, t, r);
        if (mm < M && n < a.N) {
          if (EPI == EPI_QKV) {
            ppos[e] = row_position(a.row_pos, mm, a.pos_ptr, a.pos_const);
            const int half = a.hd >> 1, spp = half / 16;
            const int head = bix / spp, sidx = bix - head * spp;
            if (t == 0 && head < a.n_q + a.n_kv) {
              pre0[e] = a.cos_tab[(size_t)ppos[e] * half + sidx * 16 + r];
              pre1[e] = a.sin_tab[(size_t)ppos[e] * half + sidx * 16 + r];
            }
          }
        }
      }
    }
  };
  if (EPI == EPI_QKV) prefetch_epi();
  bf16x8 xh[4], xm[4], xl[4];
  f32x4 xa[4], xb[4], la[4], lb[4];
  // TIMING-ONLY knock-outs inside the launch (g16_slab bits 4-7, wrong results by construction; tools/g16_inkernel_probe.sh):
  // 16 = no activation-plane loads, 32 = no weight-fragment loads, 64 = no MFMAs, 128 = no plane stores in the epilogue
#ifdef CSM_G16_KO   // variant build only (python -c "from csm_hf_amd import build; build.build_library(defines=('CSM_G16_KO',), out='csm-hf_amd/libcsm_hip_g16ko.so')";
                    // CSM_HIP_LIB selects it): the runtime branches cost the default build 5 % of the B = 16 step
  const int ko = a.g16_slab >> 4;
#else
  constexpr int ko = 0;
#endif
  constexpr bool one = ONE && XP;
  if (XP) {
    const bf16_t* pp = a.xplanes + ((size_t)chunk * 256 + lane) * 8;
    const size_t ps = (size_t)K * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ko & 1) { xh[j] = xm[j] = xl[j] = (bf16x8)(short)(lane + j); continue; }
      xh[j] = *reinterpret_cast<const bf16x8*>(pp + j * 512);
      if (!one) {
        xm[j] = *reinterpret_cast<const bf16x8*>(pp + ps + j * 512);
        xl[j] = *reinterpret_cast<const bf16x8*>(pp + 2 * ps + j * 512);
      }
    }
  } else {
    const float* xrow = a.x + (size_t)(mlive ? m : 0) * a.ldx + k0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xa[j] = *reinterpret_cast<const f32x4*>(xrow + j * KJ);
      xb[j] = *reinterpret_cast<const f32x4*>(xrow + j * KJ + 4);
    }
    if (PRO == PRO_NORM) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        la[j] = *reinterpret_cast<const f32x4*>(a.ln + k0 + j * KJ);
        lb[j] = *reinterpret_cast<const f32x4*>(a.ln + k0 + j * KJ + 4);
      }
    }
  }
  AFrag<WT> wf[PT][4];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    if (TL) {
      // tile index of this panel tile (its first row is a multiple of 16); rows >= N are zero in the copy
      // (the last panel of a matrix whose tile count is not a multiple of PT -- the 2 051-row heads -- must not read past the copy: its
    //  extra tiles re-read the last real one and are dropped by the epilogue's n < N)
    const size_t tile = (size_t)min(g16_row<EPI, PT>(a, bix, t, 0) >> 4, ((a.N + 15) >> 4) - 1);
      const WT* wr = reinterpret_cast<const WT*>(a.Wt) + ((tile * (size_t)(K >> 7) + chunk) * 4) * 512 + lane * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ko & 2) { wf[t][j].load(reinterpret_cast<const WT*>(a.Wt) + lane * 8, 0); continue; }   // one hot KiB instead of the stream
        wf[t][j].load(wr + j * 512, a.nt);
      }
    } else {
      int n = g16_row<EPI, PT>(a, bix, t, lane & 15);
      n = n < a.N ? n : a.N - 1;  // clamp (partial last tile); results of clamped rows are never stored
      const WT* wr = W + (size_t)n * K + k0;
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[t][j].load(wr + j * 8, a.nt);
    }
  }
  if (EPI != EPI_QKV) prefetch_epi();
  if (XP && PRO == PRO_NORM && wave == 0) {
    // RMS scale of row m from the producer's per-tile sums of squares (fixed order); used only by the epilogue
    const float* sp = a.xss + (size_t)m * a.xss_ld;
    f32x4 pv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = g * 4 + 16 * i;
      pv[i] = (f32x4)(0.f);
      if (t < a.xss_n) pv[i] = *reinterpret_cast<const f32x4*>(sp + t);
    }
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) q += (pv[i][0] + pv[i][1]) + (pv[i][2] + pv[i][3]);
    q = xor32_sum(xor16_sum(q));
    if (lane < 16) stat[lane] = __builtin_amdgcn_rsqf(q * __builtin_amdgcn_rcpf((float)K) + a.eps);
  }
  if (!XP && PRO == PRO_NORM) {
    // RMS statistic of row m from the registers: lanes (m, g) of all NW waves cover the whole row.
    // The exchange uses a bare s_barrier (LDS counter only): __syncthreads() would also drain vmcnt, i.e. wait
    // for the weight fragments before any of this arithmetic could start.
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) ss += xa[j][i] * xa[j][i] + xb[j][i] * xb[j][i];
    ss = xor32_sum(xor16_sum(ss));
    if (lane < 16) stat[wave * 16 + lane] = ss;
    lds_barrier();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += stat[w * 16 + m];
    const float sc = rsqrtf(tot / (float)K + a.eps);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xa[j][i] = (xa[j][i] * sc) * la[j][i];
        xb[j][i] = (xb[j][i] * sc) * lb[j][i];
      }
  }
  if (!XP) {
    // exact 3-way bf16 split of the whole slice, still ahead of the first use of a weight fragment
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!mlive) { xa[j] = (f32x4)(0.f); xb[j] = (f32x4)(0.f); }
      if (one) {   // nearest-even bf16 like the producers' one-plane stores
        uint32_t* ph = reinterpret_cast<uint32_t*>(&xh[j]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float v0 = p < 2 ? xa[j][2 * p] : xb[j][2 * p - 4], v1 = p < 2 ? xa[j][2 * p + 1] : xb[j][2 * p - 3];
          ph[p] = (uint32_t)f32_to_bf16(v0) | ((uint32_t)f32_to_bf16(v1) << 16);
        }
      } else {
        split3(xa[j], xb[j], xh[j], xm[j], xl[j]);
      }
    }
  }
  f32x4 acc[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) acc[t] = (f32x4)(0.f);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const bf16x8 af = wf[t][j].get();
      if (ko & 4) { acc[t][0] += (float)af[0] + (float)xl[j][0] + (float)xm[j][1] + (float)xh[j][2]; continue; }   // operands stay live, no matrix work
      if (!one) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xl[j], acc[t], 0, 0, 0);  // small terms first
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xm[j], acc[t], 0, 0, 0);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xh[j], acc[t], 0, 0, 0);
    }
  }

  // ---- K reduction across the waves through LDS, fixed order (deterministic) ---------------------------
  // C/D layout of 16x16x32: col (= batch row m) = lane & 15, row (= weight row) = (lane >> 4) * 4 + reg
#pragma unroll
  for (int t = 0; t < PT; ++t) *reinterpret_cast<f32x4*>(red + ((wave * PT + t) * 64 + lane) * 4) = acc[t];
  __syncthreads();
  // quad q = 4 consecutive panel elements: summed over the waves, and (KB > 1) published, by ONE thread as 16 bytes
  constexpr int NQ = (PT * 64 + 64 * NW - 1) / (64 * NW);
  f32x4 mine[NQ];
#pragma unroll
  for (int e = 0; e < NQ; ++e) {
    const int q = tid + e * 64 * NW;
    mine[e] = (f32x4)(0.f);
    if (q < PT * 64) {
      // association: eight waves at a time, the groups of eight then in order -- what the 8-wave kernels with a K split across
      // workgroups compute (gemm32.h at K = 2048: two slabs of eight waves), so that a 16-wave launch of this kernel and a wider
      // launch of that one round a row alike, bit for bit
      f32x4 s = (f32x4)(0.f);
#pragma unroll
      for (int w0 = 0; w0 < NW; w0 += 8) {
        f32x4 sg = (f32x4)(0.f);
#pragma unroll
        for (int w = w0; w < w0 + 8 && w < NW; ++w) sg += *reinterpret_cast<const f32x4*>(red + w * PT * 256 + q * 4);
        s += sg;
      }
      mine[e] = s;
      *reinterpret_cast<f32x4*>(panel + q * 4) = s;
    }
  }
  if (KB > 1) {
    // ---- K reduction across workgroups: slab + ticket, last arriver combines (cdna guide, section 5 "in-launch split-K
    // reduction", the write-through form): 16-byte sc1 slab stores (one fabric write per quad; the 4-byte agent-scope atomic
    // stores used before are ~6x the time per byte), every wave drains its stores, one relaxed agent ticket; the reducer
    // reads the slabs with 16-byte sc1 loads (L1 bypassed, served coherently) -- valid because the producers stored sc1 --
    // instead of an agent acquire fence (buffer_inv sc1, ~1.7 us on its own) + plain loads.  Placement-independent.
    // A two-launch variant (partials, then a reduce kernel) measured the same or slower (dec 11.7 vs 11.1 us,
    // backbone 19.6 vs 16.9 us at M = 16). ---------------------
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, 0x7ffffff0, 0x00020000);
    const unsigned slab_off = (unsigned)(((size_t)bix * KB + biy) * (PT * 256) * sizeof(float));
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
      const int q = tid + e * 64 * NW;
      if (q < PT * 64) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mine[e]), rs, slab_off + q * 16, 0, /*sc1*/ 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int tk = __hip_atomic_fetch_add(tickets + bix, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = tk == KB - 1;
      if (last) __hip_atomic_store(tickets + bix, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = last;
    }
    __syncthreads();
    if (!*flag) { TL_END(0x70 + EPI + 8 * PRO); return; }
    const unsigned base_off = (unsigned)((size_t)bix * KB * (PT * 256) * sizeof(float));
    for (int q = tid; q < PT * 64; q += 64 * NW) {
      f32x4 v[16];  // all KB (<= 16) slab loads in flight at once, then a fixed-order sum
#pragma unroll
      for (int kb = 0; kb < 16; ++kb)
        v[kb] = kb < KB ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + (unsigned)(kb * PT * 1024 + q * 16), 0, /*sc1*/ 16))
                        : (f32x4)(0.f);
      f32x4 s = (f32x4)(0.f);
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) s += v[kb];
      *reinterpret_cast<f32x4*>(panel + q * 4) = s;
    }
  }
  __syncthreads();

  if (QUAD) {
    if (tid < PT * 64) {
      const int t = tid >> 6, l = tid & 63, mm = l & 15;
      const int n0 = g16_row<EPI, PT>(a, bix, t, (l >> 4) * 4);
      if (mm < M && n0 < a.N) {
        f32x4 pv = *reinterpret_cast<const f32x4*>(panel + t * 256 + l * 4);
        // XP + norm: the planes carry x * norm_weight, the RMS scale of the row multiplies the finished dot product
        const float rs = (XP && PRO == PRO_NORM) ? stat[mm] : 1.f;
        if (a.wscale) {
          const f32x4 ws = *reinterpret_cast<const f32x4*>(a.wscale + n0);
          pv[0] *= ws[0]; pv[1] *= ws[1]; pv[2] *= ws[2]; pv[3] *= ws[3];
        }
        pv[0] *= rs; pv[1] *= rs; pv[2] *= rs; pv[3] *= rs;
        if (EPI == EPI_RESID) {
          const f32x4 xn = rq + pv;
          *reinterpret_cast<f32x4*>(a.out + (size_t)mm * a.ldo + n0) = xn;
          if (a.oplanes && !(ko & 8)) {
            f32x4 xt;
            xt[0] = xn[0] * lq[0]; xt[1] = xn[1] * lq[1]; xt[2] = xn[2] * lq[2]; xt[3] = xn[3] * lq[3];
            store_planes4(a.oplanes, (size_t)a.N * 16, n0, mm, xt, one);
            if (a.oss) red[t * 64 + l] = (xn[0] * xn[0] + xn[1] * xn[1]) + (xn[2] * xn[2] + xn[3] * xn[3]);
          }
        } else {   // SwiGLU: (gate, up) pairs
          const float h0 = (pv[0] / (1.f + __expf(-pv[0]))) * pv[1];
          const float h1 = (pv[2] / (1.f + __expf(-pv[2]))) * pv[3];
          if (a.oplanes && !(ko & 8)) store_planes2(a.oplanes, (size_t)(a.N >> 1) * 16, n0 >> 1, mm, h0, h1, one);   // the consumer reads the planes only
          else if (a.oplanes) {}
          else *reinterpret_cast<f32x2*>(a.out + (size_t)mm * a.ldo + (n0 >> 1)) = f32x2{h0, h1};
        }
      }
    }
  }
  // ---- epilogue: element i = (t, l, reg): weight row = tile row (l>>4)*4+reg, batch row = l & 15 -------
#pragma unroll
  for (int e = 0; e < (QUAD ? 0 : NE); ++e) {
    const int i = tid + e * 64 * NW;
    if (i >= PT * 256) continue;
    const int t = i >> 8, l = (i >> 2) & 63, reg = i & 3;
    const int mm = l & 15, r = (l >> 4) * 4 + reg;
    if (mm >= M) continue;
    const int n = g16_row<EPI, PT>(a, bix, t, r);
    if (n >= a.N) continue;
    // XP + norm: the planes carry x * norm_weight, the RMS scale of the row multiplies the finished dot product
    const float rs = (XP && PRO == PRO_NORM) ? stat[mm] : 1.f;
    const float v = panel[i] * (a.wscale ? a.wscale[n] : 1.f) * rs;
    if (EPI == EPI_STORE) {
      a.out[(size_t)mm * a.ldo + n] = v;
    } else if (EPI == EPI_RESID) {
      const float xn = pre0[e] + v;
      a.out[(size_t)mm * a.ldo + n] = xn;
      if (a.oplanes) {
        store_planes(a.oplanes, (size_t)a.N * 16, n, mm, xn * pre1[e], one);
        if (a.oss) red[i] = xn * xn;   // red is free after the panel sum; tile sums are formed below
      }
    } else if (EPI == EPI_SWIGLU) {
      if (!(reg & 1)) {
        const float u = panel[i + 1] * (a.wscale ? a.wscale[n + 1] : 1.f) * rs;
        const float hv = (v / (1.f + __expf(-v))) * u;
        a.out[(size_t)mm * a.ldo + (n >> 1)] = hv;
        if (a.oplanes) store_planes(a.oplanes, (size_t)(a.N >> 1) * 16, n >> 1, mm, hv, one);
      }
    } else {  // EPI_QKV, PT == 2: tile 0 = first RoPE half, tile 1 = second half of the same head rows
      const int half = a.hd >> 1, spp = half / 16;
      const int head = bix / spp, s = bix - head * spp;
      const int hi = s * 16 + r;  // index inside the half
      const int b = a.row_seq ? a.row_seq[mm] : a.seq_base + mm;
      const int pos = ppos[e];
      KT* kc = reinterpret_cast<KT*>(a.kcache);
      KT* vc = reinterpret_cast<KT*>(a.vcache);
      if (head < a.n_q + a.n_kv) {
        if (t == 0) {
          const float v0 = v, v1 = panel[256 + (i & 255)] * (a.wscale ? a.wscale[n + half] : 1.f) * rs;
          const float c = pre0[e], sn = pre1[e];
          // explicit contraction, the form of rope_scatter_kernel / rope_epilogue_row (misc.h, gemm.h): left to the compiler, two
            // instantiations of this epilogue rounded differently (1e-6 in the logits between 16-row and wider launches, round 4)
            const float o0 = __fmaf_rn(v0, c, -__fmul_rn(v1, sn)), o1 = __fmaf_rn(v1, c, __fmul_rn(v0, sn));
          if (head < a.n_q) {
            float* q = a.qbuf + (size_t)mm * a.n_q * a.hd + head * a.hd;
            q[hi] = __fmul_rn(o0, a.qscale);
            q[hi + half] = __fmul_rn(o1, a.qscale);
          } else {
            const int j = head - a.n_q;
            store_kv(kc + k_index<KT>(b, j, hi, pos, a.n_kv, a.hd, a.lmax), o0);
            store_kv(kc + k_index<KT>(b, j, hi + half, pos, a.n_kv, a.hd, a.lmax), o1);
          }
        }
      } else {
        const int j = head - a.n_q - a.n_kv;
        store_kv(vc + v_index(b, j, pos, t * half + hi, a.n_kv, a.hd, a.lmax), v);
      }
    }
  }
  if (EPI == EPI_RESID && a.oplanes && a.oss) {
    // per-tile sums of x_new^2 for the consumer's RMS scale: thread (t, mm) adds its tile's 16 columns in fixed order
    __syncthreads();
    if (tid < PT * 16) {
      const int t = tid >> 4, mm = tid & 15;
      const int n0 = g16_row<EPI, PT>(a, bix, t, 0);
      if (mm < M && n0 < a.N) {
        const float q = (red[t * 64 + mm] + red[t * 64 + 16 + mm]) + (red[t * 64 + 32 + mm] + red[t * 64 + 48 + mm]);
        a.oss[(size_t)mm * a.oss_ld + (n0 >> 4)] = q;
      }
    }
  }
  if (a.bump_a && bix == 0 && tid == 0) {
    *a.bump_a += 1;
    if (a.bump_b) *a.bump_b += 1;
  }
  TL_END(0x70 + EPI + 8 * PRO);
}
#endif  // CSM_ARGS_ONLY

// returns -2 when the shape / dtype is not covered by the MFMA path
int launch_gemm16(hipStream_t st, int wdtype, int kvdtype, int M, int pro, int epi, const GemvArgs& a, float* slabs,
                  size_t slab_floats, int* tickets, int n_tickets);
