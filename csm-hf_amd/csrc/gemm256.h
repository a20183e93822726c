// 256 x 256 tile of the LDS-DMA prefill GEMM (gemm_mx.h), for launches that fill the chip with such tiles (long
// contexts, batched prefills):   C[R,N] (+)= A[R,K] @ W[N,K]^T
//   MX = false: one bf16 activation plane x bf16 weights on v_mfma_f32_16x16x32_bf16  (prefill_precision = "bf16")
//   MX = true : OCP MX-fp8 operands on v_mfma_scale_f32_16x16x128_f8f6f4              (prefill_precision = "mxfp8")
//
// Why a second tile: gemm_dma_bf16_kernel / gemm_mx_kernel (128 x 128, four waves of 64 x 64) read 16 operand fragments from
// LDS per 32 MFMAs and sit at the LDS port's limit (matrix pipe 14-28 % busy).  Here 512 threads = 8
// waves as 2 (activation-row halves) x 4 (weight-row quarters), wave tile 128 x 64 = 8 x 4 MFMA tiles: 24 fragment reads per
// 64 MFMAs, every staged byte is used by twice as many MFMAs, one workgroup per CU (2 x 64 KiB of LDS).
//
// Schedule of one k-step (128 bytes of K per row; stage st = k-step & 1), one s_barrier per k-step:
//   top : the four weight fragments pairs of the step (8 reads)
//   P0-3: phase q multiplies activation tiles 2q, 2q+1 by all four weight tiles (16 MFMAs) while the fragments of phase q+1
//         are on their way from LDS into the OTHER activation register set (phase 3 fetches tiles 0, 1 of the NEXT step)
//   between P2 and P3: s_waitcnt lgkmcnt(0) (this wave has read all of stage st) + vmcnt(0) (its share of the DMA of step
//         ks+1, issued one whole k-step earlier, has landed) -> s_barrier -> stage st^1 is visible to every wave and stage st
//         is free -> the DMA of step ks+2 into stage st is issued and stays in flight for a whole k-step.
// LDS-DMA data is ordered for a ds_read only by the issuing waves' vmcnt wait followed by a barrier the reader has passed
// (cdna guide: "read a staged buffer after the wait that retires it + a barrier"): every read of stage st^1 sits behind
// that barrier.  Operand image, swizzle and fragment reads are those of gemm_mx.h (measured layout, tools/ubench/mx_layout.py).
// The matrix instruction accumulates in ascending k as one fp32 chain, so without a K split the bf16 form is BITWISE the
// 128 x 128 kernels' result (tests/test_gpu_round3.py).
//
// Replaces the same nn.Linear calls as gemm_mx.h (reference call site modeling_csm.py:345-354).  Roofline: MFMA.
#pragma once
#include "gemm_mx.h"

#ifndef CSM_ARGS_ONLY
template <int EPI, bool MX, int VAR, typename ARGS>
__global__ __launch_bounds__(512) void gemm256_kernel(ARGS a) {
  constexpr int BM = 256, BN = 256, BKB = 128;        // activation rows, weight rows, bytes of K per row and step
  constexpr int ESZ = MX ? 1 : 2;
  constexpr int TILE = BM * BKB;                      // 32 KiB per operand tile
  constexpr int SC = MX ? BM * 4 : 0;                 // scale tile per operand: [256 rows][4 blocks of 32]
  constexpr int STAGE = 2 * TILE + 2 * SC;
  extern __shared__ __attribute__((aligned(16))) uint8_t g256_lds[];   // [2 stages][A tile | W tile | A scales | W scales]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int j16 = lane & 15, g = lane >> 4;
  const int nbm = (a.R + BM - 1) / BM, nbn = a.N / BN;
  // tile order of gemm_mx_kernel: the workgroups of one XCD (blockIdx % 8) walk every row block of ONE weight panel before
  // the next panel, and an XCD only touches the panels n = 8 q + xcd
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
  const int kspan = EPI == GEPI_PARTIAL ? a.K / a.ksplit : a.K;          // elements
  const int kbeg = EPI == GEPI_PARTIAL ? (int)blockIdx.y * kspan : 0;
  const int nk = kspan * ESZ / BKB;
  const size_t rowb = (size_t)a.K * ESZ;
  const uint8_t* Ab;
  const uint8_t* Wb;
  if constexpr (MX) { Ab = a.Aq; Wb = a.Wq; }
  else { Ab = reinterpret_cast<const uint8_t*>(a.Aplanes); Wb = reinterpret_cast<const uint8_t*>(a.W); }

  // ---- LDS-DMA sources: wave w, instruction i covers tile rows 32 w + 8 i .. + 8 of each operand; lane l lands at row
  // + (l >> 3), chunk position l & 7 and must FETCH chunk (l & 7) ^ f(row), f(r) = bit1(r) << 1 | bit3(r) << 2.  Every address
  // is (wave-uniform 64-bit base) + (32-bit lane offset): the global_load_lds SADDR form, two VGPRs for all sixteen streams
  // instead of a 64-bit pointer per stream -- bit 3 of the row is i & 1, bit 1 is bit 1 of l >> 3, so only even / odd i differ
  // (R % 256 == 0 is required by the launcher: no row clamp).
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lrow = (unsigned)(lane >> 3);
  const unsigned cl = (unsigned)(lane & 7) ^ (((lrow >> 1) & 1u) << 1);
  const unsigned loff[2] = {lrow * (unsigned)rowb + cl * 16u, lrow * (unsigned)rowb + (cl ^ 4u) * 16u};
  const uint8_t* abase = Ab + (size_t)(r0 + wv * 32) * rowb + (size_t)kbeg * ESZ;
  const uint8_t* wbase = Wb + (size_t)(n0 + wv * 32) * rowb + (size_t)kbeg * ESZ;
  const uint8_t* sbase = nullptr;   // MX: waves 0-3 fetch the activation scales of rows 64 w .. + 63, waves 4-7 the weight scales
  unsigned soff = 0;
  if constexpr (MX) {
    const int K32 = a.K >> 5;
    sbase = (wv < 4 ? a.As + (size_t)(r0 + (wv & 3) * 64) * K32 : a.Ws + (size_t)(n0 + (wv & 3) * 64) * K32) + (kbeg >> 5);
    soff = (unsigned)lane * (unsigned)K32;
  }
  auto issue = [&](int ks, int st) {
    uint8_t* base = g256_lds + st * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase + (size_t)i * 8 * rowb + (size_t)ks * BKB + loff[i & 1]),
                                       (__attribute__((address_space(3))) void*)(base + (wv * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + (size_t)i * 8 * rowb + (size_t)ks * BKB + loff[i & 1]),
                                       (__attribute__((address_space(3))) void*)(base + TILE + (wv * 4 + i) * 1024), 16, 0, 0);
    if constexpr (MX)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + (size_t)ks * 4 + soff),
                                       (__attribute__((address_space(3))) void*)(base + 2 * TILE + wv * 256), 4, 0, 0);
  };

  // VAR 2: the same pieces through the BUFFER form (wave-uniform descriptor + scalar offset + one 32-bit lane offset: no
  // vector arithmetic per piece), issued ONE OR TWO AT A TIME between MFMA groups instead of nine in a row behind the barrier
  // -- a piece costs 60-185 issue cycles (cdna guide) and all eight waves leave the barrier together, so the burst left the
  // matrix pipes empty for ~1k cycles per k-step.
  const auto arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7ffffff0, 0x00020000);
  const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7ffffff0, 0x00020000);
  const auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(MX ? sbase : abase), 0, 0x7ffffff0, 0x00020000);
  const int rowb8 = (int)(8 * rowb);
  auto piece = [&](int kt, int st, int pc) {     // pc 0-3: activation rows 8 pc .. + 8 of this wave's 32, 4-7: weight rows, 8: scales (MX)
    uint8_t* base = g256_lds + st * STAGE;
    if (pc < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (__attribute__((address_space(3))) void*)(base + (wv * 4 + pc) * 1024), 16,
                                               (int)loff[pc & 1], pc * rowb8 + kt * BKB, 0, 0);
    else if (pc < 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(base + TILE + (wv * 4 + pc - 4) * 1024), 16,
                                               (int)loff[pc & 1], (pc - 4) * rowb8 + kt * BKB, 0, 0);
    else if constexpr (MX)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (__attribute__((address_space(3))) void*)(base + 2 * TILE + wv * 256), 4, (int)soff, kt * 4, 0, 0);
  };

  f32x4 acc[8][4];   // [activation-row tile][weight-row tile]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);

  // fragment read offsets of this lane inside a stage (gemm_mx.h): 16-byte chunks g and g + 4 of row (tile, j16), swizzled
  const int fsw = (((j16 >> 1) & 1) << 1) | (((j16 >> 3) & 1) << 2);
  const int c0 = (g ^ fsw) * 16, c1 = ((g + 4) ^ fsw) * 16;
  const int aoff = (wr * 128 + j16) * BKB, woff = TILE + (wc * 64 + j16) * BKB;
  const int asc = 2 * TILE + (wr * 128 + j16) * 4 + g, wsc = 2 * TILE + SC + (wc * 64 + j16) * 4 + g;

  // a fragment = the two 16-byte chunks of a row as ONE 8-register tuple (the operand shape of the MX instruction; the bf16
  // instruction takes its aligned halves)
  mx_v8i wf[4];          // weight fragments of the step
  mx_v8i af[2][2];       // [register set][tile of the pair]
  int sw[4], sa[2][2];
  auto ld_frag = [&](const uint8_t* p) {
    if constexpr (VAR == 4) return mx_v8i{0, 0, 0, 0, 0, 0, 0, 0};
    const u32x4 lo = *reinterpret_cast<const u32x4*>(p + c0), hi = *reinterpret_cast<const u32x4*>(p + c1);
    if constexpr (VAR == 3) { asm volatile("" ::"v"(lo), "v"(hi)); }
    return mx_v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  auto read_w = [&](const uint8_t* sb) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf[t] = ld_frag(sb + woff + t * 16 * BKB);
      if constexpr (MX) sw[t] = (int)sb[wsc + t * 64];
    }
  };
  auto read_a = [&](const uint8_t* sb, int pair, int set) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      af[set][t] = ld_frag(sb + aoff + (pair * 2 + t) * 16 * BKB);
      if constexpr (MX) sa[set][t] = (int)sb[asc + (pair * 2 + t) * 64];
    }
  };
  typedef __attribute__((ext_vector_type(4))) int g256_v4i;
  auto read_wh = [&](const uint8_t* sb, int h) {          // weight tiles 2h, 2h+1
#pragma unroll
    for (int t = 2 * h; t < 2 * h + 2; ++t) {
      wf[t] = ld_frag(sb + woff + t * 16 * BKB);
      if constexpr (MX) sw[t] = (int)sb[wsc + t * 64];
    }
  };
  // activation tiles 2 pair, 2 pair + 1 (register set `set`) x weight tiles n0 .. n1 - 1
  auto mma = [&](int pair, int set, int n0 = 0, int n1 = 4) {
    if constexpr (VAR == 3) return;
    __builtin_amdgcn_s_setprio(1);
    if constexpr (MX) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ni = n0; ni < n1; ++ni)
          acc[pair * 2 + t][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], af[set][t], acc[pair * 2 + t][ni], 0, 0, 0, sw[ni], 0, sa[set][t]);
    } else {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int ni = n0; ni < n1; ++ni) {
            const g256_v4i wv = s2 ? __builtin_shufflevector(wf[ni], wf[ni], 4, 5, 6, 7) : __builtin_shufflevector(wf[ni], wf[ni], 0, 1, 2, 3);
            const g256_v4i av = s2 ? __builtin_shufflevector(af[set][t], af[set][t], 4, 5, 6, 7) : __builtin_shufflevector(af[set][t], af[set][t], 0, 1, 2, 3);
            acc[pair * 2 + t][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dma_bf16x8, wv), __builtin_bit_cast(dma_bf16x8, av),
                                                                              acc[pair * 2 + t][ni], 0, 0, 0);
          }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  issue(0, 0);
  if (nk > 1) {
    issue(1, 1);
    if constexpr (MX) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();                       // stage 0 has landed for every wave
  read_a(g256_lds, 0, 0);
  if constexpr (VAR == 0) {
  // The loop body is ONE basic block (no branch between the phases: with a conditional DMA issue the compiler sank the
  // MFMAs of phases 0-2 below the barrier and spilled): every step but the last runs the barrier and issues a DMA -- past the
  // end it re-fetches the last k-step into the stage nobody reads any more -- and the last step is peeled.
  for (int ks = 0; ks + 1 < nk; ++ks) {
    const uint8_t* sb = g256_lds + (ks & 1) * STAGE;
    const uint8_t* sn = g256_lds + ((ks & 1) ^ 1) * STAGE;
    read_w(sb);
    __builtin_amdgcn_sched_barrier(0);
    read_a(sb, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(sb, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_a(sb, 3, 1);                                  // the last read of stage st
    __builtin_amdgcn_sched_barrier(0);
    mma(2, 0);
    __builtin_amdgcn_sched_barrier(0);
    // this wave is done reading stage st, and its share of the DMA of step ks+1 (the only DMA in flight) has landed ...
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // ... and so for every wave: stage st^1 is visible, stage st is free
    issue(min(ks + 2, nk - 1), ks & 1);
    read_a(sn, 0, 0);                                  // tiles 0, 1 of the next step, consumed after the next weight reads
    __builtin_amdgcn_sched_barrier(0);
    mma(3, 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    const uint8_t* sb = g256_lds + ((nk - 1) & 1) * STAGE;
    read_w(sb);
    __builtin_amdgcn_sched_barrier(0);
    read_a(sb, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(sb, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_a(sb, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(3, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant last DMA must not outlive the workgroup's LDS allocation
  }

  } else if constexpr (VAR == 1 || VAR >= 3) {
    // VAR 3 / 4 / 5 (tools/bench_gemm_mx.py only: WRONG results): VAR 1 without its MFMAs / without its fragment reads /
    // without the DMA inside the loop -- which of the three streams bounds a k-step
    // VAR 1: the weight fragments live in two halves (tiles 0-1 | 2-3) and every phase multiplies by one half at a time, so
    // that the NEXT step's halves can be fetched as soon as the last phase has issued its MFMAs on the old ones: no fragment
    // read is waited for with an empty matrix pipe at the top of a step.
#define SB __builtin_amdgcn_sched_barrier(0)
    read_wh(g256_lds, 0);
    read_wh(g256_lds, 1);
    for (int ks = 0; ks + 1 < nk; ++ks) {
      const uint8_t* sb = g256_lds + (ks & 1) * STAGE;
      const uint8_t* sn = g256_lds + ((ks & 1) ^ 1) * STAGE;
      read_a(sb, 1, 1); SB;
      mma(0, 0, 0, 2); SB;
      mma(0, 0, 2, 4); SB;
      read_a(sb, 2, 0); SB;
      mma(1, 1, 0, 2); SB;
      mma(1, 1, 2, 4); SB;
      read_a(sb, 3, 1); SB;                              // the last read of stage st
      mma(2, 0, 0, 2); SB;
      mma(2, 0, 2, 4); SB;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // stage st^1 visible to every wave, stage st free
      if constexpr (VAR != 5) issue(min(ks + 2, nk - 1), ks & 1);
      read_a(sn, 0, 0); SB;
      mma(3, 1, 0, 2); SB;
      read_wh(sn, 0); SB;
      mma(3, 1, 2, 4); SB;
      read_wh(sn, 1); SB;
    }
    {
      const uint8_t* sb = g256_lds + ((nk - 1) & 1) * STAGE;
      read_a(sb, 1, 1); SB;
      mma(0, 0, 0, 2); SB;
      mma(0, 0, 2, 4); SB;
      read_a(sb, 2, 0); SB;
      mma(1, 1, 0, 2); SB;
      mma(1, 1, 2, 4); SB;
      read_a(sb, 3, 1); SB;
      mma(2, 0, 0, 2); SB;
      mma(2, 0, 2, 4); SB;
      mma(3, 1, 0, 2); SB;
      mma(3, 1, 2, 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef SB
  } else {
    // VAR 2 = VAR 1 with the DMA of a k-step spread over the phases: pieces 0-3 of step ks+2 behind the barrier of step ks
    // (one per MFMA group), pieces 4-8 in the first phases of step ks+1; the barrier of step ks+1 waits for all of them.
#define SB __builtin_amdgcn_sched_barrier(0)
    // (the prologue above issued steps 0 and 1 in full: the first iteration's top-of-step pieces re-fetch pieces 4-8 of step 1
    // into the same place -- harmless, and the loop body stays one basic block)
    read_wh(g256_lds, 0);
    read_wh(g256_lds, 1);
    for (int ks = 0; ks + 1 < nk; ++ks) {
      const uint8_t* sb = g256_lds + (ks & 1) * STAGE;
      const uint8_t* sn = g256_lds + ((ks & 1) ^ 1) * STAGE;
      const int k1 = ks + 1, k2 = min(ks + 2, nk - 1);
      read_a(sb, 1, 1); SB;
      mma(0, 0, 0, 2); SB;
      piece(k1, k1 & 1, 4); piece(k1, k1 & 1, 5); SB;
      mma(0, 0, 2, 4); SB;
      piece(k1, k1 & 1, 6); piece(k1, k1 & 1, 7); SB;
      read_a(sb, 2, 0); SB;
      mma(1, 1, 0, 2); SB;
      piece(k1, k1 & 1, 8); SB;
      mma(1, 1, 2, 4); SB;
      read_a(sb, 3, 1); SB;                              // the last read of stage st
      mma(2, 0, 0, 2); SB;
      mma(2, 0, 2, 4); SB;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // stage st^1 visible to every wave, stage st free
      read_a(sn, 0, 0); SB;
      piece(k2, ks & 1, 0); piece(k2, ks & 1, 1); SB;
      mma(3, 1, 0, 2); SB;
      read_wh(sn, 0); SB;
      piece(k2, ks & 1, 2); piece(k2, ks & 1, 3); SB;
      mma(3, 1, 2, 4); SB;
      read_wh(sn, 1); SB;
    }
    {
      const uint8_t* sb = g256_lds + ((nk - 1) & 1) * STAGE;
      read_a(sb, 1, 1); SB;
      mma(0, 0, 0, 2); SB;
      mma(0, 0, 2, 4); SB;
      read_a(sb, 2, 0); SB;
      mma(1, 1, 0, 2); SB;
      mma(1, 1, 2, 4); SB;
      read_a(sb, 3, 1); SB;
      mma(2, 0, 0, 2); SB;
      mma(2, 0, 2, 4); SB;
      mma(3, 1, 0, 2); SB;
      mma(3, 1, 2, 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef SB
  }

  // ---- epilogue: lane (j16, g) holds C[activation row 16 ri + j16][weight rows 16 ni + 4 g .. + 3] ------------------
  if constexpr (MX) {
    if (EPI == GEPI_SWIGLU && a.Cq) {
      const int F2 = a.N >> 1;
      const int cb0 = (n0 + wc * 64) >> 1;             // this wave tile's 64 weight rows = one 32-column MX block of the output
#pragma unroll
      for (int ri = 0; ri < 8; ++ri) {
        const int r = r0 + wr * 128 + ri * 16 + j16;
        float h[4][2];
        float m = 0.f;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const f32x4 v = acc[ri][ni];
          h[ni][0] = (v[0] / (1.f + __expf(-v[0]))) * v[1];
          h[ni][1] = (v[2] / (1.f + __expf(-v[2]))) * v[3];
          m = fmaxf(m, fmaxf(fabsf(h[ni][0]), fabsf(h[ni][1])));
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        int eb = (int)((__float_as_uint(m) >> 23) & 0xff) - 8;
        eb = eb < 0 ? 0 : (eb > 254 ? 254 : eb);
        const float inv = __uint_as_float((uint32_t)(254 - eb) << 23);
        if (r < a.R) {
          uint8_t* dst = a.Cq + (size_t)r * F2 + cb0 + g * 2;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const float q0 = fminf(fmaxf(h[ni][0] * inv, -448.f), 448.f), q1 = fminf(fmaxf(h[ni][1] * inv, -448.f), 448.f);
            const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q0, q1, 0, false);
            *reinterpret_cast<uint16_t*>(dst + ni * 8) = (uint16_t)pk;
          }
          if (g == 0) a.Cs[(size_t)r * (F2 >> 5) + (cb0 >> 5)] = (uint8_t)eb;
        }
      }
      return;
    }
  }
  if (EPI == GEPI_ROPE) {   // this wave tile is one head of the QKV projection (gemm.h: RopeEpi)
    const int head = (n0 + wc * 64) >> 6;
#pragma unroll
    for (int ri = 0; ri < 8; ++ri) {
      const int r = r0 + wr * 128 + ri * 16 + j16;
      if (r < a.R) rope_epilogue_row(a.rope, r, head, g, acc[ri]);
    }
    return;
  }
#pragma unroll
  for (int ri = 0; ri < 8; ++ri) {
    const int r = r0 + wr * 128 + ri * 16 + j16;
    if (r >= a.R) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + 4 * g;
      f32x4 v = acc[ri][ni];
      if (EPI == GEPI_SWIGLU) {   // (gate, up) pairs: output columns n/2, n/2 + 1
        const float h0 = (v[0] / (1.f + __expf(-v[0]))) * v[1], h1 = (v[2] / (1.f + __expf(-v[2]))) * v[3];
        bool done = false;
        if constexpr (!MX) {
          if (a.Cplanes) {        // one bf16 plane for the down_proj GEMM
            *reinterpret_cast<uint32_t*>(a.Cplanes + (size_t)r * (a.N >> 1) + (n >> 1)) = (uint32_t)f32_to_bf16(h0) | ((uint32_t)f32_to_bf16(h1) << 16);
            done = true;
          }
        }
        if (!done) *reinterpret_cast<f32x2*>(a.C + (size_t)r * a.ldc + (n >> 1)) = f32x2{h0, h1};
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

// -2 = shape not covered / the 256 x 256 tiles would not fill the chip (the caller keeps the 128 x 128 kernels)
int launch_gemm256_bf16(hipStream_t st, int epi, const GemmArgs& a, int min_wgs);
int launch_gemm256_mx(hipStream_t st, int epi, const GemmMxArgs& a, int min_wgs);
