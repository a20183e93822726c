// Prefill GEMM on the CDNA4 block-scaled fp8 matrix instruction (BASELINE configs[4]: "fp8 weights (CDNA4 fp8 MFMA)"):
//   C[R,N] (+)= A[R,K] @ W[N,K]^T,  A and W in OCP MX-fp8: e4m3 elements + one E8M0 (power-of-two) scale per 32 elements
//   along K, multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 (dequantisation fused into the instruction, fp32 accumulate;
//   2x the bf16 matrix rate -- the non-scaled fp8 MFMA of gfx950 issues at the bf16 rate, cdna guide section 3).
//
// Structure (the guide's "128^2 tile + global_load_lds width 16" rung with the counted-vmcnt overlap): 256 threads = 4 waves
// as 2 x 2, wave tile 64 x 64 = 4 x 4 MFMA tiles, k-step 128; BOTH operand tiles and their scales are staged by LDS-DMA
// (`global_load_lds`: no VGPR round trip, no ds_write pass), two LDS stages, the DMA of step s+1 in flight under the
// MFMAs of step s (raw s_barrier + counted s_waitcnt vmcnt, never a __syncthreads()).  LDS-DMA writes a wave's 64 x 16 B
// lane-linearly, so the bank swizzle is applied to the per-lane GLOBAL source address: 16-byte chunk c of row r is stored
// at chunk c ^ f(r), f(r) = bit1(r) << 1 | bit3(r) << 2 -- conflict-free for the operand reads (lane = row & 15, chunks
// g and g + 4 for g = lane >> 4) under the lane grouping in which this chip services ds_read_b128 (gemm.h: gemm_wide_kernel).
// The WEIGHTS are the A operand of the instruction and the activations the B operand, so a lane of the accumulator holds
// four consecutive output COLUMNS of one activation row: 16-byte epilogue accesses, SwiGLU (gate, up) pairs in-lane.
//
// Numerics: every product e4m3 x e4m3 x 2^(sa + sb) is exact in fp32; the instruction accumulates in fp32.  What is NOT
// exact is the activation format (3 mantissa bits): measured on the synthetic csm-1b checkpoint, 128-frame context, last
// hidden state against the same MX-dequantised weights with fp32 activations: rel-L2 0.23 (bf16 activations: 0.012) --
// a different accuracy class, which is why this path is opt-in (`prefill_precision = "mxfp8"`).
//
// Replaces the q/k/v/o/gate/up/down nn.Linear calls of transformers.LlamaModel at q_len > 1 (reference call site
// modeling_csm.py:345-354) when the engine holds MX-quantised backbone weights.  Roofline: MFMA, 2 R N K flops against the
// dense MX-fp8 peak (~4.6 PFLOP/s measured ceiling, 5 PFLOP/s nominal).
#pragma once
#include "common.h"
#include "gemm.h"

struct GemmMxArgs {
  const uint8_t* Aq;   // [R][K] e4m3
  const uint8_t* As;   // [R][K/32] E8M0 (value 2^(byte - 127))
  const uint8_t* Wq;   // [N][K] e4m3
  const uint8_t* Ws;   // [N][K/32]
  int R, N, K;         // N % 128 == 0, K % 128 == 0
  float* C;            // STORE / RESID: [R][ldc]; SWIGLU: [R][ldc], N/2 columns
  int ldc;
  int ksplit;          // GEPI_PARTIAL: grid.y splits of K (K % (128 ksplit) == 0), partial products to Cpart + s * part_stride
  float* Cpart;
  size_t part_stride;
  // GEPI_SWIGLU, nullable: the SwiGLU output leaves the launch ALREADY in MX-fp8 ([R][N/2] e4m3 + [R][N/64] scales), the A
  // operand of the down_proj GEMM -- a wave tile's 64 weight rows are exactly one 32-column block of an activation row, so
  // the block maximum is two cross-lane steps away; saves the fp32 round trip (4 bytes written + read per element) and the
  // quantiser launch.  C is then not written.
  uint8_t* Cq;
  uint8_t* Cs;
  int big;             // host-side: workgroup count from which the 256 x 256 tile (gemm256.h) takes the launch (0 = never)
  int skinny;          // host-side: > 0 = PARTIAL / SWIGLU launches of at most this many rows run 64 (<= 32 rows: 32) activation rows per workgroup
  RopeEpi rope;        // GEPI_ROPE only (gemm.h)
};

// fp32 rows -> MX-fp8: q [rows][K] e4m3 + s [rows][K/32] E8M0, the OCP MX recipe (shared scale 2^(floor(log2(amax)) - 8),
// elements saturated to +-448, round to nearest even)
struct MxQuantArgs {
  const float* x;   // [rows][ldx]
  int ldx;
  int rows, K;      // K % 32 == 0
  uint8_t* q;
  uint8_t* s;
};

#ifndef CSM_ARGS_ONLY
typedef __attribute__((ext_vector_type(8))) int mx_v8i;

__global__ __launch_bounds__(256) void mx_quant_rows_kernel(MxQuantArgs a) {
  // 4 lanes per 32-element block (8 elements each); a 256-thread workgroup covers 64 blocks
  const size_t blk = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
  const int sub = threadIdx.x & 3;
  const int bpr = a.K >> 5;
  const size_t nblk = (size_t)a.rows * bpr;
  const bool live = blk < nblk;
  const size_t row = live ? blk / bpr : 0;
  const int kb = live ? (int)(blk - row * bpr) : 0;
  const float* src = a.x + row * a.ldx + kb * 32 + sub * 8;
  f32x4 v0 = (f32x4)(0.f), v1 = (f32x4)(0.f);
  if (live) { v0 = *reinterpret_cast<const f32x4*>(src); v1 = *reinterpret_cast<const f32x4*>(src + 4); }
  float m = fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))),
                  fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  // biased exponent of amax (floor(log2) for normal numbers; amax == 0 or subnormal -> smallest scale), minus emax(e4m3) = 8
  int eb = (int)((__float_as_uint(m) >> 23) & 0xff) - 8;
  eb = eb < 0 ? 0 : (eb > 254 ? 254 : eb);
  const float inv = __uint_as_float((uint32_t)(254 - eb) << 23);   // 2^-(eb - 127): exact (eb in 0..254 -> exponent field 254..0;
  // field 0 would be zero/subnormal: only for eb = 254, i.e. amax >= 2^135, which fp32 data of this model never reaches)
  float w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float t = (i < 4 ? v0[i] : v1[i - 4]) * inv;
    w[i] = fminf(fmaxf(t, -448.f), 448.f);
  }
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(w[0], w[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(w[2], w[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(w[4], w[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(w[6], w[7], hi, true);
  if (live) {
    *reinterpret_cast<uint2*>(a.q + row * a.K + kb * 32 + sub * 8) = make_uint2((uint32_t)lo, (uint32_t)hi);
    if (sub == 0) a.s[row * bpr + kb] = (uint8_t)eb;
  }
}

// BM = activation rows per workgroup (128; 64 / 32 for the split-K / SwiGLU launches of short prefills, as gemm_dma_bf16_kernel below)
template <int EPI, int BM = 128>
__global__ __launch_bounds__(256, 2) void gemm_mx_kernel(GemmMxArgs a) {
  constexpr int BN = 128, BK = 128;                    // weight rows, k (= bytes) per step
  constexpr int RT = BM / 32, AI = BM / 32;            // 16-row MFMA tiles per wave along the activation rows; activation-tile DMA instructions per wave
  constexpr int ATILE = BM * BK, TILE = BN * BK;       // activation tile; 16 KiB weight tile
  constexpr int STAGE = ATILE + TILE + 2 * 128 * 4;    // + the two scale tiles [128 rows][4 k blocks] (the activation one is used up to row BM)
  extern __shared__ __attribute__((aligned(16))) uint8_t mx_lds[];   // [2 stages][A tile | W tile | A scales | W scales]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int j16 = lane & 15, kb = lane >> 4;
  const int nbm = (a.R + BM - 1) / BM, nbn = a.N / BN;
  // tile order (as gemm_wide_kernel): the workgroups of one XCD (blockIdx % 8) walk every row block of ONE weight panel
  // before the next panel, and an XCD only ever touches the weight panels n = 8 q + xcd: W is fetched once per chip
  int bm, bn;
  if (nbn % 8 == 0) {
    const int q = (int)blockIdx.x >> 3;
    bm = q % nbm;
    bn = (q / nbm) * 8 + ((int)blockIdx.x & 7);
  } else {
    bm = (int)blockIdx.x % nbm;
    bn = (int)blockIdx.x / nbm;
  }
  const int r0 = bm * BM, n0 = bn * BN;
  const int kspan = EPI == GEPI_PARTIAL ? a.K / a.ksplit : a.K;
  const int kbeg = EPI == GEPI_PARTIAL ? (int)blockIdx.y * kspan : 0;
  const int nk = kspan / BK;
  const int K32 = a.K >> 5;

  // ---- LDS-DMA sources of this lane.  Operand tiles: wave w, instruction i covers tile rows 32 w + 8 i .. + 8; lane l
  // lands at row + (l >> 3), chunk position l & 7, and must therefore FETCH chunk (l & 7) ^ f(row).
  const uint8_t* asrc[AI];
  const uint8_t* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2));
    wsrc[i] = a.Wq + (size_t)(n0 + row) * a.K + kbeg + c * 16;
  }
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = wave * (BM / 4) + i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2));
    const int ra = min(r0 + row, a.R - 1);                        // rows past R: a valid address, the result is never stored
    asrc[i] = a.Aq + (size_t)ra * a.K + kbeg + c * 16;
  }
  // scale tiles: waves 0 / 1 fetch the activation rows 0-63 / 64-127, waves 2 / 3 the weight rows (4 bytes = 4 k blocks each)
  const uint8_t* ssrc;
  {
    const int row = (wave & 1) * 64 + lane;
    ssrc = wave < 2 ? a.As + (size_t)min(r0 + row, a.R - 1) * K32 + (kbeg >> 5) : a.Ws + (size_t)(n0 + row) * K32 + (kbeg >> 5);
  }
  auto issue = [&](int ks, int st) {
    uint8_t* base = mx_lds + st * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (size_t)ks * BK),
                                       (__attribute__((address_space(3))) void*)(base + (wave * AI + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (size_t)ks * BK),
                                       (__attribute__((address_space(3))) void*)(base + ATILE + (wave * 4 + i) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ssrc + (size_t)ks * 4),
                                     (__attribute__((address_space(3))) void*)(base + ATILE + TILE + wave * 256), 4, 0, 0);
  };

  f32x4 acc[RT][4];   // [activation-row tile][weight-row tile]
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);

  // operand read offsets of this lane inside a tile: row (tile t) = half * 64 + 16 t + j16.  The instruction's K layout
  // (measured, tools/ubench/mx_layout.py): lane (row, g) holds k = 16 g .. 16 g + 15 in its first four registers and
  // k = 64 + 16 g .. in the last four, while the SCALE of 32-block b is taken from lane row + 16 b -- so lane g reads the
  // 16-byte chunks g and g + 4 (stored at g ^ f, (g + 4) ^ f; f depends on row & 15 = j16 only: tiles are 16 rows apart)
  const int fsw = (((j16 >> 1) & 1) << 1) | (((j16 >> 3) & 1) << 2);
  const int c0 = (kb ^ fsw) * 16, c1 = ((kb + 4) ^ fsw) * 16;
  const int arow0 = (wr * (BM / 2) + j16) * BK, wrow0 = (wc * 64 + j16) * BK;
  const int asc0 = (wr * (BM / 2) + j16) * 4 + kb, wsc0 = (wc * 64 + j16) * 4 + kb;

  issue(0, 0);
  for (int ks = 0; ks < nk; ++ks) {
    const int st = ks & 1;
    if (ks + 1 < nk) {
      issue(ks + 1, st ^ 1);
      // this wave's AI + 5 DMA instructions of step ks have landed; step ks+1 stays in flight
      if (AI == 4) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else if (AI == 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                          // ... and so have every other wave's
    const uint8_t* At = mx_lds + st * STAGE;
    const uint8_t* Wt = At + ATILE;
    const uint8_t* Asc = At + ATILE + TILE;
    const uint8_t* Wsc = Asc + 128 * 4;
    mx_v8i af[RT], wf[4];
    int sa[RT], sw[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < RT) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(At + arow0 + t * 16 * BK + c0);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(At + arow0 + t * 16 * BK + c1);
        af[t < RT ? t : 0] = mx_v8i{(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
        sa[t < RT ? t : 0] = (int)Asc[asc0 + t * 64];
      }
      const u32x4 w0 = *reinterpret_cast<const u32x4*>(Wt + wrow0 + t * 16 * BK + c0);
      const u32x4 w1 = *reinterpret_cast<const u32x4*>(Wt + wrow0 + t * 16 * BK + c1);
      wf[t] = mx_v8i{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
      sw[t] = (int)Wsc[wsc0 + t * 64];
    }
#pragma unroll
    for (int ri = 0; ri < RT; ++ri)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ri][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], af[ri], acc[ri][ni], 0, 0, 0, sw[ni], 0, sa[ri]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // every wave has read stage st before step ks+2 overwrites it
  }

  // ---- epilogue: lane (j16, g = kb) holds C[activation row 16 ri + j16][weight rows 16 ni + 4 g .. + 3] -----------------
  if (EPI == GEPI_SWIGLU && a.Cq) {
    const int F2 = a.N >> 1;                       // SwiGLU output columns
    const int cb0 = (n0 + wc * 64) >> 1;           // first output column of this wave tile = one 32-column MX block
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
      const int r = r0 + wr * (BM / 2) + ri * 16 + j16;
      float h[4][2];
      float m = 0.f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const f32x4 v = acc[ri][ni];
        h[ni][0] = (v[0] / (1.f + __expf(-v[0]))) * v[1];
        h[ni][1] = (v[2] / (1.f + __expf(-v[2]))) * v[3];
        m = fmaxf(m, fmaxf(fabsf(h[ni][0]), fabsf(h[ni][1])));
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64));         // the four lanes (j16, g = 0..3) hold the row's 32 columns
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      int eb = (int)((__float_as_uint(m) >> 23) & 0xff) - 8;
      eb = eb < 0 ? 0 : (eb > 254 ? 254 : eb);
      const float inv = __uint_as_float((uint32_t)(254 - eb) << 23);
      if (r < a.R) {
        uint8_t* dst = a.Cq + (size_t)r * F2 + cb0 + kb * 2;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const float q0 = fminf(fmaxf(h[ni][0] * inv, -448.f), 448.f), q1 = fminf(fmaxf(h[ni][1] * inv, -448.f), 448.f);
          const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q0, q1, 0, false);
          *reinterpret_cast<uint16_t*>(dst + ni * 8) = (uint16_t)pk;
        }
        if (kb == 0) a.Cs[(size_t)r * (F2 >> 5) + (cb0 >> 5)] = (uint8_t)eb;
      }
    }
    return;
  }
  if (EPI == GEPI_ROPE) {   // this wave tile is one head of the QKV projection (gemm.h: RopeEpi)
    const int head = (n0 + wc * 64) >> 6;
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
      const int r = r0 + wr * (BM / 2) + ri * 16 + j16;
      if (r < a.R) rope_epilogue_row(a.rope, r, head, kb, acc[ri]);
    }
    return;
  }
#pragma unroll
  for (int ri = 0; ri < RT; ++ri) {
    const int r = r0 + wr * (BM / 2) + ri * 16 + j16;
    if (r >= a.R) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + 4 * kb;
      f32x4 v = acc[ri][ni];
      if (EPI == GEPI_SWIGLU) {   // (gate, up) pairs: output columns n/2, n/2 + 1
        const float h0 = (v[0] / (1.f + __expf(-v[0]))) * v[1], h1 = (v[2] / (1.f + __expf(-v[2]))) * v[3];
        *reinterpret_cast<f32x2*>(a.C + (size_t)r * a.ldc + (n >> 1)) = f32x2{h0, h1};
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

// ---- the same structure for prefill_precision = bf16 (one bf16 activation plane x bf16 weights, v_mfma_f32_16x16x32_bf16) ----
// A k-step of 64 bf16 elements is the same 128-byte LDS row as the fp8 kernel's 128 elements, and a lane (row, g) of the
// 16x16x32 operand holds k = 32 s + 8 g .. + 7 of sub-step s = chunk g (s = 0) and chunk g + 4 (s = 1) of the row: the tile
// image, the swizzle and the two 16-byte reads per lane are IDENTICAL -- only the multiply differs (two MFMAs, no scales).
// The matrix instruction accumulates its products as one fp32 chain in ascending k, so without a K split the result is
// bitwise that of gemm_bf16x3_kernel<NPL = 1> and gemm_wide_kernel (test_csm1b_prefill_precision_bf16).
// Both operands arrive by LDS-DMA from ROW-MAJOR memory: no fragment-order weight copy is needed (gemm_wide_kernel's +1.9 GB).
typedef __attribute__((ext_vector_type(8))) short dma_bf16x8;
// NPL = activation planes: 1 (prefill_precision = bf16) or 3 (exact mode: planes hi | mid | lo a_plane_stride apart, every
// weight fragment multiplied by lo, mid, hi in that order -- small terms first, like gemm_bf16x3_kernel).  With three planes a
// stage is 64 KiB (one workgroup per CU), a k-step is 96 MFMAs per wave against 32 KB of operand reads: matrix-pipe bound,
// where the one-plane form (32 MFMAs, 16 KB) sits at the LDS port's limit.
// BM = activation rows per workgroup: 128, or 64 / 32 for SHORT prefills (R <= 64 / 32: the 128-row tile multiplies 64 / 96 dead rows --
// 13.7 us of matrix work per workgroup on the gate/up projection whatever R is --; with BM rows the wave tile is BM/2 x 64, the
// activation tile BM x 128 B and a stage 24 / 20 KiB, so two or three workgroups share a CU and the launch is bound by its weight DMA).
template <int EPI, int NPL, int BM = 128>
__global__ __launch_bounds__(256, NPL == 1 ? 2 : 1) void gemm_dma_bf16_kernel(GemmArgs a) {
  constexpr int BN = 128, BKB = 128, BKE = 64;   // k-step: 128 bytes = 64 elements
  constexpr int RT = BM / 32;                    // 16-row MFMA tiles per wave along the activation rows
  constexpr int AI = BM / 32;                    // LDS-DMA instructions per wave for one activation plane tile (8 rows each)
  constexpr int ATILE = BM * BKB, WTILE = BN * BKB, STAGE = NPL * ATILE + WTILE;
  extern __shared__ __attribute__((aligned(16))) uint8_t mx_lds[];   // [2 stages][A planes | W tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int j16 = lane & 15, g = lane >> 4;
  const int nbm = (a.R + BM - 1) / BM, nbn = a.N / BN;
  int bm, bn;
  if (nbn % 8 == 0) {
    const int q = (int)blockIdx.x >> 3;
    bm = q % nbm;
    bn = (q / nbm) * 8 + ((int)blockIdx.x & 7);
  } else {
    bm = (int)blockIdx.x % nbm;
    bn = (int)blockIdx.x / nbm;
  }
  const int r0 = bm * BM, n0 = bn * BN;
  const int kspan = EPI == GEPI_PARTIAL ? a.K / a.ksplit : a.K;
  const int kbeg = EPI == GEPI_PARTIAL ? (int)blockIdx.y * kspan : 0;
  const int nk = kspan / BKE;
  const uint8_t* Ab = reinterpret_cast<const uint8_t*>(a.Aplanes);
  const uint8_t* Wb = reinterpret_cast<const uint8_t*>(a.W);
  const size_t rowb = (size_t)a.K * 2, psb = a.a_plane_stride * 2;
  const uint8_t* asrc[AI];
  const uint8_t* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2));
    wsrc[i] = Wb + (size_t)(n0 + row) * rowb + (size_t)kbeg * 2 + c * 16;
  }
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = wave * (BM / 4) + i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2));
    const int ra = min(r0 + row, a.R - 1);
    asrc[i] = Ab + (size_t)ra * rowb + (size_t)kbeg * 2 + c * 16;
  }
  auto issue = [&](int ks, int st) {
    uint8_t* base = mx_lds + st * STAGE;
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int i = 0; i < AI; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + p * psb + (size_t)ks * BKB),
                                         (__attribute__((address_space(3))) void*)(base + p * ATILE + (wave * AI + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (size_t)ks * BKB),
                                       (__attribute__((address_space(3))) void*)(base + NPL * ATILE + (wave * 4 + i) * 1024), 16, 0, 0);
  };
  f32x4 acc[RT][4];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);
  const int fsw = (((j16 >> 1) & 1) << 1) | (((j16 >> 3) & 1) << 2);
  const int c0 = (g ^ fsw) * 16, c1 = ((g + 4) ^ fsw) * 16;
  const int arow0 = (wr * (BM / 2) + j16) * BKB, wrow0 = (wc * 64 + j16) * BKB;

  issue(0, 0);
  for (int ks = 0; ks < nk; ++ks) {
    const int st = ks & 1;
    if (ks + 1 < nk) {
      issue(ks + 1, st ^ 1);
      // the NPL AI + 4 DMA instructions per wave of the next stage stay in flight
      if (NPL * AI + 4 == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (NPL * AI + 4 == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (NPL * AI + 4 == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (NPL * AI + 4 == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (NPL * AI + 4 == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (NPL * AI + 4 == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const uint8_t* At = mx_lds + st * STAGE;
    const uint8_t* Wt = At + NPL * ATILE;
    dma_bf16x8 wf[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      *reinterpret_cast<u32x4*>(&wf[t][0]) = *reinterpret_cast<const u32x4*>(Wt + wrow0 + t * 16 * BKB + c0);
      *reinterpret_cast<u32x4*>(&wf[t][1]) = *reinterpret_cast<const u32x4*>(Wt + wrow0 + t * 16 * BKB + c1);
    }
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
      dma_bf16x8 af[NPL][2];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        *reinterpret_cast<u32x4*>(&af[p][0]) = *reinterpret_cast<const u32x4*>(At + p * ATILE + arow0 + ri * 16 * BKB + c0);
        *reinterpret_cast<u32x4*>(&af[p][1]) = *reinterpret_cast<const u32x4*>(At + p * ATILE + arow0 + ri * 16 * BKB + c1);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int p = NPL - 1; p >= 0; --p)   // lo, mid, hi: small terms first
            acc[ri][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][s2], af[p][s2], acc[ri][ni], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (EPI == GEPI_ROPE) {   // this wave tile is one head of the QKV projection (gemm.h: RopeEpi)
    const int head = (n0 + wc * 64) >> 6;
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
      const int r = r0 + wr * (BM / 2) + ri * 16 + j16;
      if (r < a.R) rope_epilogue_row(a.rope, r, head, g, acc[ri]);
    }
    return;
  }
#pragma unroll
  for (int ri = 0; ri < RT; ++ri) {
    const int r = r0 + wr * (BM / 2) + ri * 16 + j16;
    if (r >= a.R) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + 4 * g;
      f32x4 v = acc[ri][ni];
      if (EPI == GEPI_SWIGLU) {
        const float h0 = (v[0] / (1.f + __expf(-v[0]))) * v[1], h1 = (v[2] / (1.f + __expf(-v[2]))) * v[3];
        if (a.Cplanes && NPL == 1) {   // one bf16 plane
          *reinterpret_cast<uint32_t*>(a.Cplanes + (size_t)r * (a.N >> 1) + (n >> 1)) = (uint32_t)f32_to_bf16(h0) | ((uint32_t)f32_to_bf16(h1) << 16);
        } else if (a.Cplanes) {        // three exact planes for the down_proj GEMM
          store_rowplane1(a.Cplanes + (size_t)r * (a.N >> 1) + (n >> 1), a.c_plane_stride, h0);
          store_rowplane1(a.Cplanes + (size_t)r * (a.N >> 1) + (n >> 1) + 1, a.c_plane_stride, h1);
        } else {
          *reinterpret_cast<f32x2*>(a.C + (size_t)r * a.ldc + (n >> 1)) = f32x2{h0, h1};
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

// (A three-plane form on 32-wide k-steps with two workgroups per CU -- gemm_dma3_k32_kernel, round 3 -- was bitwise this
// kernel and not faster on whole prefills (profiles/r03_prefill.txt); removed in round 4.)
#endif  // CSM_ARGS_ONLY

// -2 = shape / operands not covered (the caller falls back to the other prefill GEMM kernels)
int launch_gemm_dma_bf16(hipStream_t st, int epi, const GemmArgs& a);
// host-side launchers (gemm_mx.hip); -2 = shape not covered
int launch_gemm_mx(hipStream_t st, int epi, const GemmMxArgs& a);
int launch_mx_quant(hipStream_t st, const MxQuantArgs& a);
