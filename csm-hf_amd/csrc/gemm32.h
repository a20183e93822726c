// Skinny GEMM on the matrix cores for 17..32 batched decode rows: the same launch as gemm16.h with TWO 16-row
// batch tiles per weight fragment, so the weights of a batch of 32 utterances are streamed once, not twice.
// Only the fast operand path exists here: fragment-order weights (tile16_kernel) and activations that arrive as
// planes written by the producing launch (GemvArgs::xplanes; rows 16..31 live in a second plane group right
// behind the first: group stride 3 * K * 16).  Launches without planes (layer-0 inputs) keep going through
// gemm16_kernel in two 16-row groups.  Prologues / epilogues, the split-K ticket reduction and the per-tile
// sums of squares are those of gemm16.h, indexed by (weight tile t, batch tile mt).
#pragma once
#include "gemm16.h"

#ifndef CSM_ARGS_ONLY
// MT = 16-row batch tiles per weight fragment: 2 (17..32 rows) or, round 3, 4 (33..64 rows: one launch and ONE pass over the
// weights for a 64-row batch instead of two 32-row launches -- B = 64 is launch-bound like every other batch size, so halving
// its launches is what counts).  With four tiles the B operands of tile mt+1 are requested while tile mt multiplies (two
// register sets) instead of all up front.  Round 4: up to 128 rows per launch (BASELINE configs[3]'s 128-row strong leg in ONE
// engine pass): the batch tiles beyond MT go to further workgroups of the panel (blockIdx.z), shapes in gemm32.hip.
// Kernel-argument preload (gemv.h GEMV_HOT_PARAMS has the why; gemm16.h G16_HOT_PARAMS is the 16-row form): fragment-order weights, planes,
// launch counter, position source (EPI_QKV: row_pos if there is one, else pos_ptr; other epilogues: out), the producer's sums of squares,
// N, K, a packed word and a second one.
//   packed : bit 0 nt, 1 prio, 2 kfast, 3 the position source is row_pos; bits 8-15 hd, 16-23 n_q, 24-31 n_kv
//   packed2: bits 0-7 pos_const (0 when a pointer gives the position), 8-15 M, 16-23 KB
#define G32_HOT_PARAMS const void* hWt, const void* hxp, unsigned* hprog, void* hp3, const float* hxss, int hN, int hK, uint32_t hpk, uint32_t hpk2
#define G32_HOT_ARGS(a, M_, KB_, EPI_)                                                                                                       \
  (a).Wt, (const void*)(a).xplanes, (a).prog, ((EPI_) == EPI_QKV ? (void*)((a).row_pos ? (a).row_pos : (a).pos_ptr) : (void*)(a).out), (a).xss, \
  (a).N, (a).K,                                                                                                                              \
  (uint32_t)(((a).nt ? 1u : 0u) | ((a).prio ? 2u : 0u) | ((a).kfast ? 4u : 0u) | ((a).row_pos ? 8u : 0u) | (((uint32_t)(a).hd & 255u) << 8) |   \
             (((uint32_t)(a).n_q & 255u) << 16) | (((uint32_t)(a).n_kv & 255u) << 24)),                                                       \
  (uint32_t)((((a).row_pos || (a).pos_ptr) ? 0u : ((uint32_t)(a).pos_const & 255u)) | ((uint32_t)(M_) << 8) | ((uint32_t)(KB_) << 16))
template <typename WT, typename KT, int PRO, int EPI, int NW, int PT, int MT = 2, bool ONE = false>
__global__ __launch_bounds__(64 * NW) void gemm32_kernel(G32_HOT_PARAMS, GemvArgs a, float* slabs, int* tickets) {
  a.Wt = hWt; a.xplanes = reinterpret_cast<const bf16_t*>(hxp); a.prog = hprog; a.xss = hxss; a.N = hN; a.K = hK;
  a.nt = (int)(hpk & 1u); a.prio = (int)((hpk >> 1) & 1u); a.kfast = (int)((hpk >> 2) & 1u);
  const int M = (int)((hpk2 >> 8) & 255u), KB = (int)((hpk2 >> 16) & 255u);
  if (EPI == EPI_QKV) {
    a.pos_const = (int)(hpk2 & 255u);
    if (hpk & 8u) { a.row_pos = reinterpret_cast<const int*>(hp3); a.pos_ptr = nullptr; }
    else { a.row_pos = nullptr; a.pos_ptr = reinterpret_cast<const int*>(hp3); }
    a.hd = (int)((hpk >> 8) & 255u); a.n_q = (int)((hpk >> 16) & 255u); a.n_kv = (int)(hpk >> 24);
  } else {
    a.out = reinterpret_cast<float*>(hp3);
  }
  if constexpr (!std::is_same<WT, fp8_t>::value) a.wscale = nullptr;
  constexpr int U = PT * MT;   // accumulator tiles per wave: u = t * MT + mt
  extern __shared__ __attribute__((aligned(16))) float lds[];  // red[NW][U][256] | panel[U][256] | flag[16] | stat[16 MT]
  float* red = lds;
  float* panel = lds + NW * U * 256;
  int* flag = reinterpret_cast<int*>(panel + U * 256);
  float* stat = panel + U * 256 + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // K split across workgroups with a.kfast: the k split is the FASTEST grid index (launcher: dim3(KB, panels)), so that the workgroups that
  // read the same k slice of the activation planes run on one XCD (workgroup id mod 8) and share it in that L2 (round 5, gemm128.h)
  const bool kf = KB > 1 && a.kfast;
  const int bix = kf ? (int)blockIdx.y : (int)blockIdx.x, biy = kf ? (int)blockIdx.x : (int)blockIdx.y, gdx = kf ? (int)gridDim.y : (int)gridDim.x;
  (void)gdx;
  const int K = a.K;
  const int m = lane & 15, g = lane >> 4;
  const int chunk = biy * NW + wave;
  // batch rows split across workgroups (blockIdx.z): this workgroup owns the batch tiles mt0 .. mt0 + MT - 1 of the launch.  The
  // small launches of a 128-row step (QKV: 48 panels) otherwise leave most of the chip idle while every wave walks 786 KB of planes
  const int mt0 = (int)blockIdx.z * MT;
  const int bx = (int)blockIdx.z * gdx + bix;   // slab / ticket slot of this (panel, row group)
  constexpr bool QUAD = (EPI == EPI_RESID || EPI == EPI_SWIGLU);
  constexpr int NQE = (U * 64 + 64 * NW - 1) / (64 * NW);   // epilogue quads per thread (2 with eight batch tiles x two weight tiles)
  constexpr int NE = (U * 256 + 64 * NW - 1) / (64 * NW);

  // ---- EPI_QKV epilogue inputs first (dependent chain: position -> cos/sin) -------------------------------
  float pre0[NE], pre1[NE];
  int ppos[NE];
  if (EPI == EPI_QKV) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      pre0[e] = pre1[e] = 0.f;
      ppos[e] = 0;
      const int i = tid + e * 64 * NW;
      if (i < U * 256) {
        const int u = i >> 8, t = u / MT, mt = u - t * MT, l = (i >> 2) & 63, reg = i & 3;
        const int mm = (mt0 + mt) * 16 + (l & 15), r = (l >> 4) * 4 + reg;
        const int n = g16_row<EPI, PT>(a, bix, t, r);
        if (mm < M && n < a.N) {
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
  // ---- B operands: ready-made planes (two batch tiles up front; with four: two register sets, tile mt+1 behind tile mt),
  // then the weight fragments --------------------------------------------------------------------------------
  constexpr int NB = MT < 2 ? MT : 2;
  bf16x8 xh[NB][4], xm[NB][4], xl[NB][4];
  const size_t ps = (size_t)K * 16;
  constexpr bool one = ONE;   // decode_precision = bf16 (GemvArgs::pl1): one nearest-even activation plane, one MFMA per weight fragment (gemm16.h)
  auto load_planes = [&](int mt, int buf) {
    const bf16_t* pp = a.xplanes + (size_t)(mt0 + mt) * 3 * ps + ((size_t)chunk * 256 + lane) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[buf][j] = *reinterpret_cast<const bf16x8*>(pp + j * 512);
      if (!one) {
        xm[buf][j] = *reinterpret_cast<const bf16x8*>(pp + ps + j * 512);
        xl[buf][j] = *reinterpret_cast<const bf16x8*>(pp + 2 * ps + j * 512);
      }
    }
  };
  load_planes(0, 0);
  if (MT == 2) load_planes(1, 1);
  AFrag<WT> wf[PT][4];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    // (the last panel of a matrix whose tile count is not a multiple of PT -- the 2 051-row heads -- must not read past the copy: its
    //  extra tiles re-read the last real one and are dropped by the epilogue's n < N)
    const size_t tile = (size_t)min(g16_row<EPI, PT>(a, bix, t, 0) >> 4, ((a.N + 15) >> 4) - 1);
    const WT* wr = reinterpret_cast<const WT*>(a.Wt) + ((tile * (size_t)(K >> 7) + chunk) * 4) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[t][j].load(wr + j * 512, a.nt);
  }
  // residual values (and the consumer's norm weight for the output planes): requested last, consumed last
  f32x4 rq[NQE], lq[NQE];
#pragma unroll
  for (int e = 0; e < NQE; ++e) {
    rq[e] = (f32x4)(0.f); lq[e] = (f32x4)(1.f);
    const int q = tid + e * 64 * NW;
    if (EPI == EPI_RESID && q < U * 64) {
      const int u = q >> 6, t = u / MT, mt = u - t * MT, l = q & 63, mm = (mt0 + mt) * 16 + (l & 15);
      const int n0 = g16_row<EPI, PT>(a, bix, t, (l >> 4) * 4);
      if (mm < M && n0 < a.N) {
        rq[e] = *reinterpret_cast<const f32x4*>(a.out + (size_t)mm * a.ldo + n0);
        if (a.oplanes && a.oln) lq[e] = *reinterpret_cast<const f32x4*>(a.oln + n0);
      }
    }
  }
  // RMS scale of the rows of batch tile mt = wave (waves 0 and 1) from the producer's per-tile sums of squares
  if (PRO == PRO_NORM && wave < MT) {
    const float* sp = a.xss + (size_t)((mt0 + wave) * 16 + m) * a.xss_ld;
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
    if (lane < 16) stat[wave * 16 + lane] = __builtin_amdgcn_rsqf(q * __builtin_amdgcn_rcpf((float)K) + a.eps);
  }
  f32x4 acc[PT][MT];
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4)(0.f);
  if (MT == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const bf16x8 af = wf[t][j].get();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (!one) {
            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xl[mt & 1][j], acc[t][mt], 0, 0, 0);  // small terms first
            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xm[mt & 1][j], acc[t][mt], 0, 0, 0);
          }
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xh[mt & 1][j], acc[t][mt], 0, 0, 0);
        }
      }
    }
  } else {
    // per accumulator the same k order as above (j ascending; lo, mid, hi inside a step): a row's result does not depend on
    // the batch size
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (mt + 1 < MT) load_planes(mt + 1, (mt + 1) & 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          const bf16x8 af = wf[t][j].get();
          if (!one) {
            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xl[mt & 1][j], acc[t][mt], 0, 0, 0);
            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xm[mt & 1][j], acc[t][mt], 0, 0, 0);
          }
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xh[mt & 1][j], acc[t][mt], 0, 0, 0);
        }
      }
    }
  }
  // ---- K reduction across the waves through LDS, fixed order ------------------------------------------------
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      *reinterpret_cast<f32x4*>(red + ((wave * U + t * MT + mt) * 64 + lane) * 4) = acc[t][mt];
  __syncthreads();
  constexpr int NQ = (U * 64 + 64 * NW - 1) / (64 * NW);   // quads (4 consecutive panel elements) per thread
  f32x4 mine[NQ];
#pragma unroll
  for (int e = 0; e < NQ; ++e) {
    const int q = tid + e * 64 * NW;
    mine[e] = (f32x4)(0.f);
    if (q < U * 64) {
      f32x4 s = (f32x4)(0.f);
#pragma unroll
      for (int w = 0; w < NW; ++w) s += *reinterpret_cast<const f32x4*>(red + w * U * 256 + q * 4);
      mine[e] = s;
      *reinterpret_cast<f32x4*>(panel + q * 4) = s;
    }
  }
  if (KB > 1) {   // split-K across workgroups: 16-byte sc1 slab stores + ticket, the last arriver combines with sc1 loads (gemm16.h)
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, 0x7ffffff0, 0x00020000);
    const unsigned slab_off = (unsigned)(((size_t)bx * KB + biy) * (U * 256) * sizeof(float));
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
      const int q = tid + e * 64 * NW;
      if (q < U * 64) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mine[e]), rs, slab_off + q * 16, 0, /*sc1*/ 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int tk = __hip_atomic_fetch_add(tickets + bx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = tk == KB - 1;
      if (last) __hip_atomic_store(tickets + bx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    const unsigned base_off = (unsigned)((size_t)bx * KB * (U * 256) * sizeof(float));
    for (int q = tid; q < U * 64; q += 64 * NW) {
      f32x4 v[16];
#pragma unroll
      for (int kb = 0; kb < 16; ++kb)
        v[kb] = kb < KB ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + (unsigned)(kb * U * 1024 + q * 16), 0, /*sc1*/ 16))
                        : (f32x4)(0.f);
      f32x4 s = (f32x4)(0.f);
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) s += v[kb];
      *reinterpret_cast<f32x4*>(panel + q * 4) = s;
    }
  }
  __syncthreads();

  // ---- epilogues ------------------------------------------------------------------------------------------
  if (QUAD) {
#pragma unroll
    for (int e = 0; e < NQE; ++e) {
      const int q = tid + e * 64 * NW;
      if (q >= U * 64) continue;
      const int u = q >> 6, t = u / MT, mt = u - t * MT, l = q & 63, mm = (mt0 + mt) * 16 + (l & 15);
      const int n0 = g16_row<EPI, PT>(a, bix, t, (l >> 4) * 4);
      if (mm < M && n0 < a.N) {
        f32x4 pv = *reinterpret_cast<const f32x4*>(panel + u * 256 + l * 4);
        const float rs = (PRO == PRO_NORM) ? stat[mt * 16 + (l & 15)] : 1.f;
        if (a.wscale) {
          const f32x4 ws = *reinterpret_cast<const f32x4*>(a.wscale + n0);
          pv[0] *= ws[0]; pv[1] *= ws[1]; pv[2] *= ws[2]; pv[3] *= ws[3];
        }
        pv[0] *= rs; pv[1] *= rs; pv[2] *= rs; pv[3] *= rs;
        if (EPI == EPI_RESID) {
          const f32x4 xn = rq[e] + pv;
          *reinterpret_cast<f32x4*>(a.out + (size_t)mm * a.ldo + n0) = xn;
          if (a.oplanes) {
            f32x4 xt;
            xt[0] = xn[0] * lq[e][0]; xt[1] = xn[1] * lq[e][1]; xt[2] = xn[2] * lq[e][2]; xt[3] = xn[3] * lq[e][3];
            const size_t ps = (size_t)a.N * 16;
            store_planes4(a.oplanes + (size_t)(mt0 + mt) * 3 * ps, ps, n0, l & 15, xt, one);
            if (a.oss) red[u * 64 + l] = (xn[0] * xn[0] + xn[1] * xn[1]) + (xn[2] * xn[2] + xn[3] * xn[3]);
          }
        } else {   // SwiGLU: (gate, up) pairs
          const float h0 = (pv[0] / (1.f + __expf(-pv[0]))) * pv[1];
          const float h1 = (pv[2] / (1.f + __expf(-pv[2]))) * pv[3];
          if (a.oplanes) {
            const size_t ps = (size_t)(a.N >> 1) * 16;
            store_planes2(a.oplanes + (size_t)(mt0 + mt) * 3 * ps, ps, n0 >> 1, l & 15, h0, h1, one);
          } else {
            *reinterpret_cast<f32x2*>(a.out + (size_t)mm * a.ldo + (n0 >> 1)) = f32x2{h0, h1};
          }
        }
      }
    }
    if (EPI == EPI_RESID && a.oplanes && a.oss) {   // per-tile sums of x_new^2 for the consumer's RMS scale
      __syncthreads();
      if (tid < U * 16) {
        const int u = tid >> 4, t = u / MT, mt = u - t * MT, mm = (mt0 + mt) * 16 + (tid & 15);
        const int n0 = g16_row<EPI, PT>(a, bix, t, 0);
        if (mm < M && n0 < a.N) {
          const int c = tid & 15;
          a.oss[(size_t)mm * a.oss_ld + (n0 >> 4)] = (red[u * 64 + c] + red[u * 64 + 16 + c]) + (red[u * 64 + 32 + c] + red[u * 64 + 48 + c]);
        }
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int i = tid + e * 64 * NW;
      if (i >= U * 256) continue;
      const int u = i >> 8, t = u / MT, mt = u - t * MT, l = (i >> 2) & 63, reg = i & 3;
      const int mm = (mt0 + mt) * 16 + (l & 15), r = (l >> 4) * 4 + reg;
      if (mm >= M) continue;
      const int n = g16_row<EPI, PT>(a, bix, t, r);
      if (n >= a.N) continue;
      const float rs = (PRO == PRO_NORM) ? stat[mt * 16 + (l & 15)] : 1.f;
      const float v = panel[i] * (a.wscale ? a.wscale[n] : 1.f) * rs;
      if (EPI == EPI_STORE) {
        a.out[(size_t)mm * a.ldo + n] = v;
      } else {  // EPI_QKV, PT == 2: tile 0 = first RoPE half, tile 1 = second half of the same head rows
        const int half = a.hd >> 1, spp = half / 16;
        const int head = bix / spp, s = bix - head * spp;
        const int hi = s * 16 + r;
        const int b = a.row_seq ? a.row_seq[mm] : a.seq_base + mm;
        const int pos = ppos[e];
        KT* kc = reinterpret_cast<KT*>(a.kcache);
        KT* vc = reinterpret_cast<KT*>(a.vcache);
        if (head < a.n_q + a.n_kv) {
          if (t == 0) {
            const float v0 = v, v1 = panel[(MT + mt) * 256 + (i & 255)] * (a.wscale ? a.wscale[n + half] : 1.f) * rs;
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
  }
  if (a.bump_a && bix == 0 && blockIdx.z == 0 && tid == 0) {
    *a.bump_a += 1;
    if (a.bump_b) *a.bump_b += 1;
  }
}
#endif  // CSM_ARGS_ONLY

// 17..64 rows (two or four batch tiles per weight fragment); needs a.xplanes and a.Wt; returns -2 when the shape is not covered
int gemm32_configure_all();   // dynamic-LDS attributes of the 64-row instantiations (call once per engine, outside capture)
int launch_gemm32(hipStream_t st, int wdtype, int kvdtype, int M, int pro, int epi, const GemvArgs& a, float* slabs,
                  size_t slab_floats, int* tickets, int n_tickets);
