// Prefill GEMM on the matrix cores: C[R,N] (+)= A[R,K] @ W[N,K]^T with fp32 activations.
//
// Two kernels: gemm_bf16x3_kernel (bf16 / fp8 weights, below) and, for fp32 weights, gemm_f32mfma_kernel which uses the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TFLOP/s peak) so
// that prefill hidden states carry fp32-class error for BOTH weight dtypes (bf16 weights are widened
// while staging to LDS).  Roofline: MFMA (f32 rate).  Algorithmic flops = 2*R*N*K.
// Tile 128x128x32 per 256-thread workgroup, 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32;
// LDS rows padded to 33 floats => conflict-free ds_read_b32 operand fetches.
//
// Replaces the q/k/v/o/gate/up/down nn.Linear calls of transformers.LlamaModel when q_len > 1
// (reference call site modeling_csm.py:345-354).
#pragma once
#include "common.h"

enum { GEPI_STORE = 0, GEPI_RESID = 1, GEPI_SWIGLU = 2, GEPI_PARTIAL = 3, GEPI_ROPE = 4 };

// GEPI_ROPE (the QKV GEMM of a prefill whose head_dim is 64, LDS-DMA tiles only): the epilogue of the GEMM is the whole of
// rope_scatter_kernel (misc.h) -- rotate q / k by the row's position (HF rotate_half pairs d, d + 32), scale q by hd^-0.5 into
// qbuf, append k / v to the caches in their engine layouts -- so the [R][n_q + 2 n_kv][64] fp32 QKV matrix never goes to memory
// (written + re-read: 2 x 25 MB per layer at 2 048 rows) and one launch per layer disappears.  A wave tile of these kernels is
// 64 weight rows = ONE head, and a lane holds columns 16 ni + 4 g .. + 3 for ni = 0..3: both halves of a rotation pair sit in
// the same lane (ni and ni + 2).  Replaces transformers' apply_rotary_pos_emb + DynamicCache.update behind q_proj / k_proj /
// v_proj at q_len > 1 (modeling_llama.py:130-176, 267-281; reference call site modeling_csm.py:345-354).
struct RopeEpi {
  const float* cos_tab;   // [pos][32]
  const float* sin_tab;
  const int* row_seq;     // [R] cache slot of the row's sequence
  const int* row_pos;     // [R] cache position
  const int* rope_pos;    // nullable: rotation position (position_ids) when it differs from the cache position
  float* qbuf;            // [R][n_q * 64]
  void* kcache;           // [B][n_kv][16][lmax][4]
  void* vcache;           // [B][n_kv][lmax][64]
  int n_q, n_kv, lmax;
  int kv_bf16;            // cache element type: 0 fp32, 1 bf16
  float qscale;
};

struct GemmArgs {
  const float* A;  // [R][lda]
  int lda;
  const void* W;  // [N][K]
  const float* wscale;  // per-row scale (fp8 weights), nullable
  int R, N, K;
  float* C;  // STORE/RESID: [R][ldc]; SWIGLU: [R][ldc] with N/2 columns
  int ldc;
  int f32_mfma;  // force the fp32-MFMA kernel also for bf16 / fp8 weights (A/B measurements)
  // bf16x3 kernel only: activations already split by the producer -- row-major planes [3][rows][K] (A is ignored) --
  // and, for the SwiGLU epilogue, the output written as planes [3][rows][N/2] for the down_proj GEMM (C is ignored).
  // A plane stride of 0 means ONE plane (activations rounded to bf16 by the producer: prefill_precision = bf16).
  const bf16_t* Aplanes;
  size_t a_plane_stride;
  bf16_t* Cplanes;
  size_t c_plane_stride;
  // split-K (GEPI_PARTIAL, grid.y = ksplit): split s accumulates k in [s K/ksplit, (s+1) K/ksplit) and stores its
  // partial product to Cpart + s * part_stride, row-major [R][N]; the consumer (rmsnorm_kernel) adds the partials to the
  // residual stream in fixed order.  A one-utterance prefill has only R x N / 64^2 = 256 output tiles for the N = 2048
  // projections -- one 4-wave workgroup per CU, nothing to hide a k-step's load latency behind.
  int ksplit;
  float* Cpart;
  size_t part_stride;
  // (split 0 adding straight into the residual stream -- one partial array less written and re-read -- was bitwise the
  //  all-partials form and SLOWER at every length: its read-modify-write made those workgroups the launch's tail; removed in round 4)
  // fragment-order copy of W (launchers.hip tile16_kernel; nullable): enables gemm_wide_kernel for one-plane activations
  const void* Wt;
  // gemm_wide_kernel switches (per engine, csm_set_option): wide = 0 keeps the square tile; wide_depth = weight-fragment
  // sets in registers (1: two workgroups per CU, 4: one); wide_exact = 1 also routes three-plane (exact) launches to it
  // (measured slower: off)
  int wide, wide_depth, wide_exact;
  // gemm_dma_bf16_kernel (gemm_mx.h: both operands staged by LDS-DMA from row-major memory, 128 x 128 x 64 tile) for one-plane
  // bf16 launches.  Bits: 1 = one-plane launches of at most dma_max_rows rows, 2 = one-plane always (A/B), 4 = three-plane
  // (exact) launches of 256 ... dma_max_rows rows, 8 = three-plane always (A/B), 32 = three-plane launches on 32-wide k-steps
  // (gemm_dma3_k32_kernel: two workgroups per CU).  Measured (csm-1b, whole prefill,
  // profiles/r03_prefill.txt): one plane 64 / 512 / 2048 rows 1.68 / 2.78 / 6.80 -> 1.53 / 2.29 / 6.31 ms, 8192 rows 19.6 ->
  // 22.3 ms (gemm_wide_kernel's 128 x 256 tile wins there); three planes 512 / 2048 rows 5.03 / 17.2 -> 4.52 / 16.1 ms
  int dma, dma_max_rows;
  int dma_min_wgs;   // three-plane launches with fewer 128 x 128 tiles x K splits than this stay on the 64 x 64 square tile (A/B; 0 = no limit)
  // gemm256_kernel (gemm256.h: the same staging, 256 x 256 tile, 8 waves): one-plane bf16 launches whose 256 x 256 tiles (x K
  // splits) number at least big256 (0 = never)
  int big256;
  int dma_skinny;   // 1: gemm_dma_bf16_kernel with 64 (32) activation rows per workgroup for the PARTIAL / SWIGLU launches of prefills of up to
                    // 256 rows with one plane, 768 rows with three (there also instead of the 64 x 64 square tile); <= 32 rows: 32.  A/B bits: 2 = three planes up to 4096 rows
                    // (2 048 rows: 16.2 -> 21.7 ms), 4 = keep 128 rows for launches of >= 512 workgroups (512 / 768 rows exact 4.38 / 6.59 -> 4.53 / 6.75 ms)
  RopeEpi rope;   // GEPI_ROPE only
};

#ifndef CSM_ARGS_ONLY
// One activation row of one head (wave tile): v[ni] = columns 16 ni + 4 g .. + 3 of the head.  Same arithmetic, in the same
// order, as rope_scatter_kernel.
__device__ __forceinline__ void rope_epilogue_row(const RopeEpi& p, int r, int head, int g, const f32x4 (&v)[4]) {
  const int b = p.row_seq[r], pos = p.row_pos[r];
  if (head >= p.n_q + p.n_kv) {   // v head: plain append
    const int j = head - p.n_q - p.n_kv;
    const size_t at = (((size_t)b * p.n_kv + j) * p.lmax + pos) * 64 + 4 * g;
    if (p.kv_bf16) {
      bf16_t* vr = reinterpret_cast<bf16_t*>(p.vcache) + at;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        *reinterpret_cast<uint2*>(vr + 16 * ni) = make_uint2((uint32_t)f32_to_bf16(v[ni][0]) | ((uint32_t)f32_to_bf16(v[ni][1]) << 16),
                                                              (uint32_t)f32_to_bf16(v[ni][2]) | ((uint32_t)f32_to_bf16(v[ni][3]) << 16));
    } else {
      float* vr = reinterpret_cast<float*>(p.vcache) + at;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4*>(vr + 16 * ni) = v[ni];
    }
    return;
  }
  const int rpos = p.rope_pos ? p.rope_pos[r] : pos;
  f32x4 o[4];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {   // dims 16 h2 + 4 g .. + 3 (first half) pair with the same + 32
    const f32x4 c = *reinterpret_cast<const f32x4*>(p.cos_tab + (size_t)rpos * 32 + 16 * h2 + 4 * g);
    const f32x4 s = *reinterpret_cast<const f32x4*>(p.sin_tab + (size_t)rpos * 32 + 16 * h2 + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v0 = v[h2][e], v1 = v[h2 + 2][e];
      o[h2][e] = __fmaf_rn(v0, c[e], -__fmul_rn(v1, s[e]));       // the contraction rope_scatter_kernel is written in (misc.h)
      o[h2 + 2][e] = __fmaf_rn(v1, c[e], __fmul_rn(v0, s[e]));
    }
  }
  if (head < p.n_q) {
    float* q = p.qbuf + (size_t)r * p.n_q * 64 + head * 64 + 4 * g;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = __fmul_rn(o[ni][e], p.qscale);
      *reinterpret_cast<f32x4*>(q + 16 * ni) = t;
    }
  } else {   // k head: [hd/4 = 16][lmax][4], dims 16 ni + 4 g .. + 3 are group 4 ni + g
    const int j = head - p.n_q;
    const size_t base = (((size_t)b * p.n_kv + j) * 16) * p.lmax;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const size_t at = (base + (size_t)(4 * ni + g) * p.lmax + pos) * 4;
      if (p.kv_bf16)
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.kcache) + at) =
            make_uint2((uint32_t)f32_to_bf16(o[ni][0]) | ((uint32_t)f32_to_bf16(o[ni][1]) << 16),
                       (uint32_t)f32_to_bf16(o[ni][2]) | ((uint32_t)f32_to_bf16(o[ni][3]) << 16));
      else
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.kcache) + at) = o[ni];
    }
  }
}
#endif

// K splits of a residual-epilogue prefill GEMM (o_proj, down_proj): enough workgroups for ~4 per CU, k-steps of 64
static inline int prefill_ksplit(int R, int N, int K, int cap = 4) {
  const long tiles64 = (long)((R + 63) / 64) * (N / 64);
  if (((long)((R + 127) / 128) * (N / 128)) >= 256 || tiles64 >= 768) return 1;
  int s = (int)((1024 + tiles64 - 1) / tiles64);
  if (s > cap) s = cap;
  while (s > 1 && (K % (64 * s))) --s;
  return s;
}

#ifndef CSM_ARGS_ONLY
template <typename WT, int EPI>
__global__ __launch_bounds__(256) void gemm_f32mfma_kernel(GemmArgs a) {
  constexpr int BM = 128, BN = 128, BK = 32, LD = BK + 1;
  __shared__ float As[BM * LD];
  __shared__ float Ws[BN * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int nbn = a.N / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
  const int r0 = bm * BM, n0 = bn * BN;
  const WT* W = reinterpret_cast<const WT*>(a.W);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);

  // GEPI_PARTIAL (grid.y = ksplit, K % (32 ksplit) == 0): this workgroup's K range only, result to Cpart (see GemmArgs)
  const int kspan = EPI == GEPI_PARTIAL ? a.K / a.ksplit : a.K;
  const int kbeg = EPI == GEPI_PARTIAL ? (int)blockIdx.y * kspan : 0, kend = kbeg + kspan;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    // ---- stage A tile (fp32) -------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 3, c4 = idx & 7;
      f32x4 v = (f32x4)(0.f);
      if (r0 + row < a.R) v = *reinterpret_cast<const f32x4*>(a.A + (size_t)(r0 + row) * a.lda + k0 + c4 * 4);
      float* d = As + row * LD + c4 * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    // ---- stage W tile (widen to fp32) ----------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 2, c8 = idx & 3;
      W8<WT> w;
      w.load(W + (size_t)(n0 + row) * a.K + k0 + c8 * 8);
      float* d = Ws + row * LD + c8 * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = w.get(e);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int kc = kk + (lane >> 5);
      const float a0 = As[(wr * 64 + (lane & 31)) * LD + kc];
      const float a1 = As[(wr * 64 + 32 + (lane & 31)) * LD + kc];
      const float b0 = Ws[(wc * 64 + (lane & 31)) * LD + kc];
      const float b1 = Ws[(wc * 64 + 32 + (lane & 31)) * LD + kc];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) --------------
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = r0 + wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int n = n0 + wc * 64 + ni * 32 + (lane & 31);
        const float v = acc[mi][ni][reg] * (a.wscale ? a.wscale[n] : 1.f);
        if (EPI == GEPI_SWIGLU) {
          const float o = __shfl_xor(v, 1, 64);  // even lane: gate (own), up (partner)
          if (!(lane & 1) && r < a.R) {
            const float hv = (v / (1.f + __expf(-v))) * o;
            if (a.Cplanes) store_rowplane1(a.Cplanes + (size_t)r * (a.N >> 1) + (n >> 1), a.c_plane_stride, hv);
            else a.C[(size_t)r * a.ldc + (n >> 1)] = hv;
          }
        } else if (r < a.R) {
          if (EPI == GEPI_PARTIAL) {
            a.Cpart[(size_t)blockIdx.y * a.part_stride + (size_t)r * a.N + n] = v;
          }
          else if (EPI == GEPI_RESID) a.C[(size_t)r * a.ldc + n] += v;
          else a.C[(size_t)r * a.ldc + n] = v;
        }
      }
}

// ---------------------------------------------------------------------------------------------------
// bf16 / fp8 weights: the same GEMM on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 2.5 PFLOP/s dense).
// fp32 activations are split EXACTLY into three bf16 planes (8 + 8 + 8 mantissa bits, truncation, no rounding)
// while they are staged into LDS; every weight fragment is multiplied by the three planes, small terms first,
// so the result carries fp32 summation-order error only (bf16 x bf16 products are exact in the fp32
// accumulator) at 1/3 of the bf16 MFMA rate = 5x the fp32-MFMA rate.  LDS rows are padded to 80 bytes:
// the four 16-lane groups of a ds_read_b128 then cover all 64 banks exactly once.
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short g_bf16x8;

template <typename WT>
__device__ __forceinline__ u32x4 load_w8_as_bf16(const WT* p);
template <>
__device__ __forceinline__ u32x4 load_w8_as_bf16<bf16_t>(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
template <>
__device__ __forceinline__ u32x4 load_w8_as_bf16<fp8_t>(const fp8_t* p) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  u32x4 o;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int w = (int)(h < 2 ? r.x : r.y);
    const f32x2 v = (h & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8(w, true) : __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
    o[h] = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xffff0000u);
  }
  return o;
}

// BT = square block tile (128 or 64); 4 waves as 2x2, each wave (BT/2)x(BT/2) = (BT/64)^2 MFMA tiles of 32x32.
// The 64x64 tile is used when the 128x128 grid would leave most of the chip idle (prefill of one utterance
// through the N = 2048 projections: 4 x 16 tiles).
// BKT = k-step (64 when K % 64 == 0: half the barriers per weight byte, 144-byte LDS rows; else 32)
// NPL = activation planes multiplied per weight fragment: 3 (exact fp32 activations) or 1 (bf16 activations, AP only)
template <typename WT, int EPI, int BT, int BKT, bool AP, int NPL = 3>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(GemmArgs a) {
  static_assert(NPL == 3 || (NPL == 1 && AP), "the one-plane form reads producer-written planes");
  constexpr int BM = BT, BN = BT, BK = BKT, LDK = BK + 8;   // bf16 elements per LDS row (80 / 144 bytes: conflict-free b128 reads)
  constexpr int A4 = BK / 4, W8N = BK / 8;                        // f32x4 pieces per A row, 8-weight pieces per W row
  constexpr int NA = BM * A4 / 256, NWL = BN * W8N / 256;         // pieces per thread per k-step
  constexpr int NP = BM * W8N / 256;                              // AP: 8-element pieces per plane per thread
  constexpr int TI = BT / 64;                              // MFMA tiles per wave per dimension
  __shared__ __attribute__((aligned(16))) bf16_t Ap[NPL][BM * LDK];
  __shared__ __attribute__((aligned(16))) bf16_t Ws[BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int nbn = a.N / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
  const int r0 = bm * BM, n0 = bn * BN;
  const WT* W = reinterpret_cast<const WT*>(a.W);

  f32x16 acc[TI][TI];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j) acc[i][j] = (f32x16)(0.f);

  // register prefetch: the global loads of k-step s+1 are issued right after the staging barrier of step s and fly
  // under its LDS reads and MFMAs (at R = 512 the grid is one workgroup per CU, so nothing else hides them)
  f32x4 pa[AP ? 1 : NA];
  u32x4 pp[AP ? NPL * NP : 1];
  u32x4 pw[NWL];
  const int kspan = EPI == GEPI_PARTIAL ? a.K / a.ksplit : a.K;
  const int kbeg = EPI == GEPI_PARTIAL ? (int)blockIdx.y * kspan : 0, kend = kbeg + kspan;
  auto fetch = [&](int k0) {
    if (AP) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / W8N, c8 = idx % W8N;
        const bf16_t* src = a.Aplanes + (size_t)(r0 + row) * a.K + k0 + c8 * 8;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          pp[NPL * i + p] = (u32x4)(0u);
          if (r0 + row < a.R) pp[NPL * i + p] = *reinterpret_cast<const u32x4*>(src + p * a.a_plane_stride);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / A4, c4 = idx % A4;
        pa[i] = (f32x4)(0.f);
        if (r0 + row < a.R) pa[i] = *reinterpret_cast<const f32x4*>(a.A + (size_t)(r0 + row) * a.lda + k0 + c4 * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / W8N, c8 = idx % W8N;
      pw[i] = load_w8_as_bf16<WT>(W + (size_t)(n0 + row) * a.K + k0 + c8 * 8);
    }
  };
  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    // ---- stage A: ready-made planes, or fp32 -> three bf16 planes --------------------------------
    if (AP) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / W8N, c8 = idx % W8N;
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(&Ap[p][row * LDK + c8 * 8]) = pp[NPL * i + p];
      }
    }
#pragma unroll
    for (int i = 0; i < (AP ? 0 : NA); ++i) {
      const int idx = tid + i * 256;
      const int row = idx / A4, c4 = idx % A4;
      const f32x4 v = pa[i];
      uint32_t h[2], m[2], l[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float v0 = v[2 * p], v1 = v[2 * p + 1];
        const uint32_t h0 = __float_as_uint(v0) & 0xffff0000u, h1 = __float_as_uint(v1) & 0xffff0000u;
        const float q0 = v0 - __uint_as_float(h0), q1 = v1 - __uint_as_float(h1);
        const uint32_t m0 = __float_as_uint(q0) & 0xffff0000u, m1 = __float_as_uint(q1) & 0xffff0000u;
        const float s0 = q0 - __uint_as_float(m0), s1 = q1 - __uint_as_float(m1);
        h[p] = (h0 >> 16) | h1;
        m[p] = (m0 >> 16) | m1;
        l[p] = (__float_as_uint(s0) >> 16) | (__float_as_uint(s1) & 0xffff0000u);
      }
      const int off = row * LDK + c4 * 4;
      *reinterpret_cast<uint2*>(&Ap[0][off]) = make_uint2(h[0], h[1]);
      if (NPL == 3) {
        *reinterpret_cast<uint2*>(&Ap[NPL - 2][off]) = make_uint2(m[0], m[1]);
        *reinterpret_cast<uint2*>(&Ap[NPL - 1][off]) = make_uint2(l[0], l[1]);
      }
    }
    // ---- stage W (bf16 as is, fp8 widened exactly) -------------------------------------------------
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / W8N, c8 = idx % W8N;
      *reinterpret_cast<u32x4*>(&Ws[row * LDK + c8 * 8]) = pw[i];
    }
    lds_barrier();
    if (k0 + BK < kend) fetch(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      const int ko = kk + (lane >> 5) * 8;
      g_bf16x8 af[TI][NPL], bf[TI];
#pragma unroll
      for (int mi = 0; mi < TI; ++mi)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          *reinterpret_cast<u32x4*>(&af[mi][p]) =
              *reinterpret_cast<const u32x4*>(&Ap[p][(wr * (BM / 2) + mi * 32 + (lane & 31)) * LDK + ko]);
#pragma unroll
      for (int ni = 0; ni < TI; ++ni)
        *reinterpret_cast<u32x4*>(&bf[ni]) = *reinterpret_cast<const u32x4*>(&Ws[(wc * (BN / 2) + ni * 32 + (lane & 31)) * LDK + ko]);
#pragma unroll
      for (int mi = 0; mi < TI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TI; ++ni) {
          if (NPL == 3) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][NPL - 1], bf[ni], acc[mi][ni], 0, 0, 0);  // lo
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][NPL - 2], bf[ni], acc[mi][ni], 0, 0, 0);  // mid
          }
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][0], bf[ni], acc[mi][ni], 0, 0, 0);  // hi
        }
    }
    lds_barrier();   // LDS-only: the prefetched loads stay in flight
  }
  // ---- epilogue (same C/D layout as the fp32 kernel) ------------------------------------------------
#pragma unroll
  for (int mi = 0; mi < TI; ++mi)
#pragma unroll
    for (int ni = 0; ni < TI; ++ni)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = r0 + wr * (BM / 2) + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int n = n0 + wc * (BN / 2) + ni * 32 + (lane & 31);
        const float v = acc[mi][ni][reg] * (a.wscale ? a.wscale[n] : 1.f);
        if (EPI == GEPI_SWIGLU) {
          const float o = __shfl_xor(v, 1, 64);
          if (!(lane & 1) && r < a.R) {
            const float hv = (v / (1.f + __expf(-v))) * o;
            if (a.Cplanes) store_rowplane1(a.Cplanes + (size_t)r * (a.N >> 1) + (n >> 1), a.c_plane_stride, hv);
            else a.C[(size_t)r * a.ldc + (n >> 1)] = hv;
          }
        } else if (r < a.R) {
          if (EPI == GEPI_PARTIAL) {
            a.Cpart[(size_t)blockIdx.y * a.part_stride + (size_t)r * a.N + n] = v;
          }
          else if (EPI == GEPI_RESID) a.C[(size_t)r * a.ldc + n] += v;
          else a.C[(size_t)r * a.ldc + n] = v;
        }
      }
}


// ---- prefill_precision = bf16, launches with enough output tiles: 128 x 256 x 64 tile ------------------------------------
// Four waves side by side along N (wave tile 128 x 64 = 8 x 4 MFMA tiles of 16 x 16, v_mfma_f32_16x16x32_bf16).  The
// ACTIVATION tile goes through LDS (all four waves read all of it; double-buffered: one barrier per k-step); each wave
// loads ITS OWN weight fragments straight from the fragment-order copy of the matrix (GemmArgs::Wt, the copy the decode
// kernels stream: one fragment = one contiguous 1 KiB block per wavefront) into the MFMA operand registers -- nobody
// else needs them, so they never touch LDS.  A fragment register is reloaded for the next k-step right after the MFMAs
// that consumed it: every weight load has one whole k-step of matrix work in front of it.
// Why: per k-step the square-tile kernel moves 2 KB through LDS per 32x32x16 MFMA (64 KB read + 32 KB written against
// 512 matrix clocks per wave: 768 clocks of the 128 B/clk LDS port -> LDS-bound at 2/3 of the matrix rate before any
// barrier); this one moves 16 KB written + 64 KB read against 1024 matrix clocks (640 port clocks).
// (A first version read the fragments from the ROW-MAJOR matrix, lane = one weight row: 32-byte pieces of 32 different
// lines per load instruction -- the texture path, not LDS, became the limit and it ran 1.4x SLOWER than the square tile.)
// Roofline: bf16 MFMA, 2 R N K flops.
typedef __attribute__((ext_vector_type(8))) short gw_bf16x8;
// NPL = activation planes: 1 (prefill_precision = bf16) or 3 (exact: every weight fragment is multiplied by the lo, mid and
// hi plane, small terms first; 120 KB of LDS and ~300 registers: one workgroup per CU, three times the matrix work per
// weight byte; measured slower than the square tile and off by default: GemmArgs::wide_exact)
template <typename WT, int EPI, int DEPTH, int NPL = 1>
__global__ __launch_bounds__(256, (DEPTH == 1 && NPL == 1) ? 2 : 1) void gemm_wide_kernel(GemmArgs a) {
  // LDS rows of 160 bytes: with the lane groups ds_read_b128 / ds_write_b128 are serviced in ({0-3,12-15,20-27},
  // {4-11,16-19,28-31}, +32) the operand reads (lane = row & 15, k group = lane >> 4) touch 16 distinct 16-byte bank
  // slots per group at this stride (144 bytes, right for the 32x32x16 operand layout, gave 40 % conflict cycles here)
  constexpr int BM = 128, BN = 256, BK = 64, LDK = BK + 16;
  constexpr int MI = BM / 16, NI = BN / 4 / 16, KS = BK / 32;
  extern __shared__ __attribute__((aligned(16))) bf16_t As_all[];   // [2 buffers][NPL planes][BM * LDK]
  auto As = [&](int buf, int p) { return As_all + (size_t)(buf * NPL + p) * (BM * LDK); };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, g = lane >> 4;
  const int nbm = (a.R + BM - 1) / BM, nbn = a.N / BN;
  // tile order: workgroup b runs on XCD b % 8 (plus a per-stream rotation); the workgroups of one XCD walk the row
  // blocks of ONE weight panel before moving to the next, so a panel is fetched from HBM once per XCD L2, not per row block
  int bm, bn;
  if (nbn % 8 == 0) {
    const int q = (int)blockIdx.x >> 3;
    bm = q % nbm;
    bn = (q / nbm) * 8 + ((int)blockIdx.x & 7);
  } else {
    bm = (int)blockIdx.x % nbm;
    bn = (int)blockIdx.x / nbm;
  }
  const int r0 = bm * BM, n0 = bn * BN + wave * (BN / 4);
  const int kspan = EPI == GEPI_PARTIAL ? a.K / a.ksplit : a.K;
  const int kbeg = EPI == GEPI_PARTIAL ? (int)blockIdx.y * kspan : 0, kend = kbeg + kspan;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4)(0.f);

  // this thread's four 16-byte pieces of the activation tile: row = arow + 32 i, k piece ac8.  Each write lane group
  // holds rows {r, r + 4} of its wave's 8 rows (slots 10 r + c and 10 (r + 4) + c = +8 mod 16: conflict-free)
  const int l5 = lane & 31;
  const bool wg2 = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;
  const int widx = wg2 ? (l5 < 12 ? l5 - 4 : l5 < 20 ? l5 - 8 : l5 - 16) : (l5 < 4 ? l5 : l5 < 16 ? l5 - 8 : l5 - 12);
  const int arow = wave * 8 + (lane >> 5) * 2 + (wg2 ? 1 : 0) + 4 * (widx >> 3), ac8 = widx & 7;
  const bf16_t* asrc = a.Aplanes + (size_t)(r0 + arow) * a.K + ac8 * 8;
  u32x4 pa[NPL][4];
  auto fetch_a = [&](int k0) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pa[p][i] = (u32x4)(0u);
        if (r0 + arow + 32 * i < a.R) pa[p][i] = *reinterpret_cast<const u32x4*>(asrc + p * a.a_plane_stride + (size_t)(32 * i) * a.K + k0);
      }
  };
  auto stage_a = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(As(buf, p) + (arow + 32 * i) * LDK + ac8 * 8) = pa[p][i];
  };
  // fragment (16-row tile t, 32-wide k block kb) of the copy starts at (t * K/32 + kb) * 512 elements
  const WT* wsrc = reinterpret_cast<const WT*>(a.Wt) + (size_t)(n0 >> 4) * a.K * 16 + lane * 8;
  const size_t tstride = (size_t)a.K * 16;
  // DEPTH k-steps of weight fragments in registers (one wave per SIMD: the register file is this wave's alone): the set
  // of step s is reloaded for step s + DEPTH right after its MFMAs, i.e. every weight load has DEPTH k-steps of matrix
  // work (DEPTH x 0.43 us at the full matrix rate) to cover an HBM round trip under load (~2 us)
  u32x4 wf[DEPTH][NI][KS];
  auto wload = [&](int set, int ni, int ks, int k0) {
    wf[set][ni][ks] = load_w8_as_bf16<WT>(wsrc + ni * tstride + (size_t)((k0 >> 5) + ks) * 512);
  };
  // k is walked from a per-workgroup starting step, wrapping around: the workgroups of a launch would otherwise all sit
  // at the same k at the same time, and with tiles 16 K elements (a power of two) apart every fragment request of that
  // moment lands on the same few L2 / HBM channels (measured: ~1/4 of the L2 bandwidth, 24 % matrix-pipe busy)
  const int nstep = kspan / BK;
  const int s0 = 0;   // (a per-workgroup rotated start of the k walk changed the summation order and did not pay: removed in round 4)
  auto kof = [&](int i) {   // k of logical step i (0 <= i < nstep + DEPTH)
    int t = s0 + i;
    t = t >= nstep ? t - nstep : t;
    t = t >= nstep ? t - nstep : t;
    return kbeg + t * BK;
  };
  fetch_a(kof(0));
#pragma unroll
  for (int set = 0; set < DEPTH; ++set)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (set < nstep) wload(set, ni, ks, kof(set));
  stage_a(0);
  lds_barrier();
  auto xload = [&](gw_bf16x8* xf, int buf, int p, int ks) {   // B operands: lane (row m16, k group g) of the 8 row tiles
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      *reinterpret_cast<u32x4*>(&xf[mi]) = *reinterpret_cast<const u32x4*>(As(buf, p) + (mi * 16 + m16) * LDK + ks * 32 + g * 8);
  };
  auto step = [&](int set, int buf, int i) {
    const bool more = i + 1 < nstep, mored = i + DEPTH < nstep;
    if (more) fetch_a(kof(i + 1));
    const int kd = kof(i + DEPTH);
    constexpr bool XDB = DEPTH > 1 && NPL == 1;   // room for both halves' operands only in the one-plane one-wave-per-SIMD form
    gw_bf16x8 xf[XDB ? KS : NPL][MI];
    if (XDB) xload(xf[0], buf, 0, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (!XDB) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) xload(xf[p], buf, p, ks);
      } else if (ks + 1 < KS) {
        xload(xf[ks + 1], buf, 0, ks + 1);   // the next half's operands fly under this half's MFMAs
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        gw_bf16x8 wfr;
        *reinterpret_cast<u32x4*>(&wfr) = wf[set][ni][ks];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          if (NPL == 3) {   // planes are stored hi, mid, lo: small terms first
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, xf[2][mi], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, xf[1][mi], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, xf[0][mi], acc[mi][ni], 0, 0, 0);
          } else {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, xf[XDB ? ks : 0][mi], acc[mi][ni], 0, 0, 0);
          }
        }
        if (mored) wload(set, ni, ks, kd);
      }
    }
    if (more) stage_a(buf ^ 1);
    lds_barrier();   // LDS only: the weight loads of the next steps stay in flight
  };
  for (int i = 0; i < nstep; i += DEPTH) {   // the launcher guarantees a k span that is a multiple of DEPTH * BK
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) step(d, (i + d) & 1, i + d);
  }
  // ---- epilogue: C/D layout of 16x16x32 with the weights as the A operand: lane (m16, g) holds
  //      C[row tile mi, row m16][col tile ni, columns 4g .. 4g+3] ----------------------------------------------------------
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int r = r0 + mi * 16 + m16;
    if (r >= a.R) continue;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + ni * 16 + 4 * g;
      f32x4 v = acc[mi][ni];
      if (a.wscale) {
        const f32x4 ws = *reinterpret_cast<const f32x4*>(a.wscale + n);
        v[0] *= ws[0]; v[1] *= ws[1]; v[2] *= ws[2]; v[3] *= ws[3];
      }
      if (EPI == GEPI_SWIGLU) {   // (gate, up) pairs: columns n/2, n/2 + 1 of the output
        const float h0 = (v[0] / (1.f + __expf(-v[0]))) * v[1], h1 = (v[2] / (1.f + __expf(-v[2]))) * v[3];
        if (a.Cplanes && a.c_plane_stride == 0) {
          *reinterpret_cast<uint32_t*>(a.Cplanes + (size_t)r * (a.N >> 1) + (n >> 1)) = (uint32_t)f32_to_bf16(h0) | ((uint32_t)f32_to_bf16(h1) << 16);
        } else if (a.Cplanes) {
          store_rowplane1(a.Cplanes + (size_t)r * (a.N >> 1) + (n >> 1), a.c_plane_stride, h0);
          store_rowplane1(a.Cplanes + (size_t)r * (a.N >> 1) + (n >> 1) + 1, a.c_plane_stride, h1);
        } else {
          float* c = a.C + (size_t)r * a.ldc + (n >> 1);
          c[0] = h0; c[1] = h1;
        }
      } else if (EPI == GEPI_PARTIAL) {
        *reinterpret_cast<f32x4*>(a.Cpart + (size_t)blockIdx.y * a.part_stride + (size_t)r * a.N + n) = v;
      } else if (EPI == GEPI_RESID) {
        f32x4* c = reinterpret_cast<f32x4*>(a.C + (size_t)r * a.ldc + n);
        const f32x4 o = *c;
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        *c = v;
      } else {
        *reinterpret_cast<f32x4*>(a.C + (size_t)r * a.ldc + n) = v;
      }
    }
  }
}

#endif  // CSM_ARGS_ONLY
int launch_gemm(hipStream_t st, int wdtype, int epi, const GemmArgs& a);
