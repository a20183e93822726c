// Host launchers for attention / GEMM / misc kernels.
#define CSM_PREFETCH_KERNELS 1
#define CSM_ATTN_OPROJ_KERNEL 1
#include "attn.h"
#include "attn_prefill.h"
#include "attn_oproj.h"
#include "gemm.h"
#include "misc.h"
#include "prefetch.h"

int launch_attn(hipStream_t st, int kvdtype, int rows, const AttnArgs& a) {
  if (a.n_q % a.n_kv != 0 || a.n_q / a.n_kv > 16) return -1;
  if (a.nsplit < 1 || a.nsplit > 64) return -1;
  if (a.lmax >= (1 << 24) || a.n_q > 255 || a.n_kv > 63 || (!a.row_pos && !a.pos_ptr && (a.pos_const < 0 || a.pos_const > 127))) return -1;   // packed into the preloaded word (attn.h ATTN_HOT_ARGS)
  if (a.oplanes && rows > 128) return -1;
  const int G = a.n_q / a.n_kv;
  if (a.gqa && a.hd == 64 && G == 4 && !a.tickets && (!a.oplanes || rows <= 128)) {
    const int g2 = rows * a.n_kv * a.nsplit;
    if (kvdtype == 1) hipLaunchKernelGGL((attn_decode_gqa_kernel<bf16_t>), dim3(g2), dim3(256), 0, st, ATTN_HOT_ARGS(a, rows), a);
    else hipLaunchKernelGGL((attn_decode_gqa_kernel<float>), dim3(g2), dim3(256), 0, st, ATTN_HOT_ARGS(a, rows), a);
    int e = (int)hipGetLastError();
    if (e || a.nsplit == 1 || a.no_combine) return e;
    hipLaunchKernelGGL((attn_combine_kernel<64>), dim3(rows * a.n_q), dim3(64), 0, st, a.part, a.n_q, a.nsplit, a.out, a.oplanes, a.pl1, a.dbg ? a.dbg + 4096 : nullptr);
    return (int)hipGetLastError();
  }
  const int grid = rows * a.n_kv * a.nsplit * (a.one_wave ? G : 1);
  const int bd = a.one_wave ? 64 : 256;
  if (a.hd == 64) {
    if (a.tile_prefetch) {
      if (kvdtype == 1) hipLaunchKernelGGL((attn_decode_kernel<bf16_t, 64, true>), dim3(grid), dim3(bd), 0, st, ATTN_HOT_ARGS(a, rows), a);
      else hipLaunchKernelGGL((attn_decode_kernel<float, 64, true>), dim3(grid), dim3(bd), 0, st, ATTN_HOT_ARGS(a, rows), a);
    } else if (kvdtype == 1) hipLaunchKernelGGL((attn_decode_kernel<bf16_t, 64>), dim3(grid), dim3(bd), 0, st, ATTN_HOT_ARGS(a, rows), a);
    else hipLaunchKernelGGL((attn_decode_kernel<float, 64>), dim3(grid), dim3(bd), 0, st, ATTN_HOT_ARGS(a, rows), a);
  } else if (a.hd == 128) {
    if (kvdtype == 1) hipLaunchKernelGGL((attn_decode_kernel<bf16_t, 128>), dim3(grid), dim3(bd), 0, st, ATTN_HOT_ARGS(a, rows), a);
    else hipLaunchKernelGGL((attn_decode_kernel<float, 128>), dim3(grid), dim3(bd), 0, st, ATTN_HOT_ARGS(a, rows), a);
  } else {
    return -1;
  }
  int e = (int)hipGetLastError();
  if (e) return e;
  if (a.nsplit > 1 && !a.tickets && !a.no_combine) {
    if (a.nsplit > 64) return -1;
    if (a.hd == 64) hipLaunchKernelGGL((attn_combine_kernel<64>), dim3(rows * a.n_q), dim3(64), 0, st, a.part, a.n_q, a.nsplit, a.out, a.oplanes, a.pl1, a.dbg ? a.dbg + 4096 : nullptr);
    else hipLaunchKernelGGL((attn_combine_kernel<128>), dim3(rows * a.n_q), dim3(64), 0, st, a.part, a.n_q, a.nsplit, a.out, a.oplanes, a.pl1, a.dbg ? a.dbg + 4096 : nullptr);
    e = (int)hipGetLastError();
  }
  return e;
}

template <typename WT, int BT, int BK>
static int launch_gemm_x3_bk(hipStream_t st, int epi, const GemmArgs& a) {
  const int grid = ((a.R + BT - 1) / BT) * (a.N / BT);
  if (epi == GEPI_PARTIAL) {   // split-K partial products (64 x 64 tiles, producer-written planes)
    if (!a.Aplanes || a.ksplit < 1 || a.K % (BK * a.ksplit) || !a.Cpart) return -1;
    const dim3 g2(grid, a.ksplit);
    if (a.a_plane_stride == 0) hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_PARTIAL, BT, BK, true, 1>), g2, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_PARTIAL, BT, BK, true, 3>), g2, dim3(256), 0, st, a);
    return (int)hipGetLastError();
  }
  if (a.Aplanes && a.a_plane_stride == 0) {   // one plane: activations rounded to bf16 by the producer (prefill_precision = bf16)
    if (a.K % 8) return -1;
    switch (epi) {
      case GEPI_STORE: hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_STORE, BT, BK, true, 1>), dim3(grid), dim3(256), 0, st, a); break;
      case GEPI_RESID: hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_RESID, BT, BK, true, 1>), dim3(grid), dim3(256), 0, st, a); break;
      case GEPI_SWIGLU: hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_SWIGLU, BT, BK, true, 1>), dim3(grid), dim3(256), 0, st, a); break;
      default: return -1;
    }
    return (int)hipGetLastError();
  }
  if (a.Aplanes) {
    if (a.K % 8) return -1;
    switch (epi) {
      case GEPI_STORE: hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_STORE, BT, BK, true>), dim3(grid), dim3(256), 0, st, a); break;
      case GEPI_RESID: hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_RESID, BT, BK, true>), dim3(grid), dim3(256), 0, st, a); break;
      case GEPI_SWIGLU: hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_SWIGLU, BT, BK, true>), dim3(grid), dim3(256), 0, st, a); break;
      default: return -1;
    }
    return (int)hipGetLastError();
  }
  switch (epi) {
    case GEPI_STORE: hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_STORE, BT, BK, false>), dim3(grid), dim3(256), 0, st, a); break;
    case GEPI_RESID: hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_RESID, BT, BK, false>), dim3(grid), dim3(256), 0, st, a); break;
    case GEPI_SWIGLU: hipLaunchKernelGGL((gemm_bf16x3_kernel<WT, GEPI_SWIGLU, BT, BK, false>), dim3(grid), dim3(256), 0, st, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
template <typename WT, int BT>
static int launch_gemm_x3_bt(hipStream_t st, int epi, const GemmArgs& a) {
  if (a.K % 64 == 0) return launch_gemm_x3_bk<WT, BT, 64>(st, epi, a);
  return launch_gemm_x3_bk<WT, BT, 32>(st, epi, a);
}

// 128 x 256 tiles (gemm_wide_kernel) for one-plane activations when they fill the chip (GemmArgs::wide & co.)
template <typename WT, int EPI, int DEPTH, int NPL>
static int launch_gemm_wide_k(hipStream_t st, const dim3& grid, const GemmArgs& a) {
  constexpr size_t lds = (size_t)2 * NPL * 128 * 80 * sizeof(bf16_t);   // 40 KB (one plane) / 120 KB (three)
  auto fn = gemm_wide_kernel<WT, EPI, DEPTH, NPL>;
  // hipFuncSetAttribute applies to the CURRENT device: remember per device (a process may hold engines on several GPUs)
  static unsigned long long configured = 0ull;
  if (lds > 64 * 1024) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(configured & bit)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
      configured |= bit;
    }
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, a);
  return (int)hipGetLastError();
}
template <typename WT, int DEPTH, int NPL>
static int launch_gemm_wide_d(hipStream_t st, int epi, const GemmArgs& a) {
  const int tiles = ((a.R + 127) / 128) * (a.N / 256);
  const dim3 grid(tiles, epi == GEPI_PARTIAL ? a.ksplit : 1);
  switch (epi) {
    case GEPI_STORE: return launch_gemm_wide_k<WT, GEPI_STORE, DEPTH, NPL>(st, grid, a);
    case GEPI_RESID: return launch_gemm_wide_k<WT, GEPI_RESID, DEPTH, NPL>(st, grid, a);
    case GEPI_SWIGLU: return launch_gemm_wide_k<WT, GEPI_SWIGLU, DEPTH, NPL>(st, grid, a);
    case GEPI_PARTIAL: return launch_gemm_wide_k<WT, GEPI_PARTIAL, DEPTH, NPL>(st, grid, a);
    default: return -1;
  }
}
template <typename WT>
static int launch_gemm_wide(hipStream_t st, int epi, const GemmArgs& a) {
  if (a.a_plane_stride != 0) return launch_gemm_wide_d<WT, 1, 3>(st, epi, a);   // exact: three planes
  return a.wide_depth == 4 ? launch_gemm_wide_d<WT, 4, 1>(st, epi, a) : launch_gemm_wide_d<WT, 1, 1>(st, epi, a);
}
// The three-plane (exact) form of the wide tile measured SLOWER than the square tile (512 / 2048 frames: 5.50 / 20.4 vs
// 5.33 / 19.5 ms; 16 x 512: 70 vs 58 ms): with 120 KB of LDS and 436 registers one wave per SIMD has to cover its own LDS and
// load latencies, the square tile runs two to three workgroups per CU.  GemmArgs::wide_exact keeps it reachable for A/B.
static bool gemm_wide_ok(int epi, const GemmArgs& a) {
  if (!a.wide || !a.Wt || !a.Aplanes || a.N % 256 || a.K % 256 || a.ldc % 4) return false;
  const bool exact = a.a_plane_stride != 0;
  if (exact && !a.wide_exact) return false;
  if (epi == GEPI_SWIGLU && a.Cplanes && (a.c_plane_stride != 0) != exact) return false;
  const int ks = epi == GEPI_PARTIAL ? a.ksplit : 1;
  if (epi == GEPI_PARTIAL && (ks < 1 || a.K % (256 * ks) || !a.Cpart)) return false;
  const long wgs = (long)((a.R + 127) / 128) * (a.N / 256) * ks;
  // measured (csm-1b shapes, tools/prefill_bench.py), one plane: at exactly one workgroup per CU the wide tile only ties
  // the square one (gate/up at 512 rows: 59 vs 57 us; QKV at 2048 rows, 192 tiles: 52 vs 48 us); with two per CU it wins
  // (gate/up at 2048 rows: 156 vs 187 us); the split-K partials win from one per CU on (67-84 vs 99 us).
  // Three planes: one workgroup per CU by construction (120 KB of LDS)
  if (exact) return wgs >= 192;
  return wgs >= (epi == GEPI_PARTIAL ? 256 : 320);
}

int launch_gemm_dma_bf16(hipStream_t st, int epi, const GemmArgs& a);   // gemm_mx.hip
int launch_gemm256_bf16(hipStream_t st, int epi, const GemmArgs& a, int min_wgs);   // gemm_mx.hip (gemm256.h)
template <typename WT>
static int launch_gemm_x3(hipStream_t st, int epi, const GemmArgs& a) {
  if (a.big256 > 0 && sizeof(WT) == 2 && a.Aplanes && a.a_plane_stride == 0) {   // enough 256 x 256 tiles to fill the chip
    const int r = launch_gemm256_bf16(st, epi, a, a.big256);
    if (r != -2) return r;
  }
  if (a.dma && sizeof(WT) == 2) {   // bf16 weights, one activation plane: the LDS-DMA tile
    const bool exact = a.a_plane_stride != 0;   // three planes: dma bit 2 (value 4) enables, bit 3 (8) forces
    // (three planes: one 4-wave workgroup per CU -- below ~256 rows the 64 x 64 square tile's many small workgroups win:
    //  64 rows 1.96 vs 2.32 ms per prefill; 512 / 2048 rows 5.03 / 17.2 -> 4.52 / 16.1 ms; 16 x 512 rows 56 vs 71 ms)
    const long dma_wgs = (long)((a.R + 127) / 128) * (a.N / 128) * (epi == GEPI_PARTIAL ? a.ksplit : 1);
    const bool skinny3 = exact && (a.dma & 4) && a.dma_skinny && a.R <= ((a.dma_skinny & 2) ? 4096 : 768) && (epi == GEPI_PARTIAL || epi == GEPI_SWIGLU);   // prefills of up to 768 rows, three planes: the BM = 64 / 32 LDS-DMA tile instead of the 64 x 64 square tile
    if (exact ? ((a.dma & 8) || skinny3 || ((a.dma & 4) && a.R >= 256 && a.R <= a.dma_max_rows && dma_wgs >= a.dma_min_wgs)) : ((a.dma & 2) || ((a.dma & 1) && a.R <= a.dma_max_rows))) {
      const int r = launch_gemm_dma_bf16(st, epi, a);
      if (r != -2) return r;
    }
  }
  if (epi == GEPI_ROPE) return -2;   // only the LDS-DMA tiles carry the RoPE / cache-append epilogue: the caller runs STORE + rope_scatter
  if (gemm_wide_ok(epi, a)) return launch_gemm_wide<WT>(st, epi, a);
  // 128x128 tiles unless they would occupy fewer than 256 workgroups
  if (epi == GEPI_PARTIAL) return a.K % 64 ? -1 : launch_gemm_x3_bk<WT, 64, 64>(st, epi, a);
  if (((a.R + 127) / 128) * (a.N / 128) < 256) return launch_gemm_x3_bt<WT, 64>(st, epi, a);
  return launch_gemm_x3_bt<WT, 128>(st, epi, a);
}

template <typename WT>
static int launch_gemm_t(hipStream_t st, int epi, const GemmArgs& a) {
  const int grid = ((a.R + 127) / 128) * (a.N / 128);
  switch (epi) {
    case GEPI_STORE: hipLaunchKernelGGL((gemm_f32mfma_kernel<WT, GEPI_STORE>), dim3(grid), dim3(256), 0, st, a); break;
    case GEPI_RESID: hipLaunchKernelGGL((gemm_f32mfma_kernel<WT, GEPI_RESID>), dim3(grid), dim3(256), 0, st, a); break;
    case GEPI_SWIGLU: hipLaunchKernelGGL((gemm_f32mfma_kernel<WT, GEPI_SWIGLU>), dim3(grid), dim3(256), 0, st, a); break;
    case GEPI_PARTIAL:
      if (a.ksplit < 1 || a.K % (32 * a.ksplit) || !a.Cpart) return -1;
      hipLaunchKernelGGL((gemm_f32mfma_kernel<WT, GEPI_PARTIAL>), dim3(grid, a.ksplit), dim3(256), 0, st, a);
      break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

int launch_gemm(hipStream_t st, int wdtype, int epi, const GemmArgs& a) {
  if (a.N % 128 != 0 || a.K % 32 != 0 || a.R < 1) return -1;
  if (epi == GEPI_ROPE && (wdtype != 1 || a.f32_mfma || !a.Aplanes || !a.rope.qbuf)) return -2;
  if (wdtype == 2) return a.f32_mfma ? launch_gemm_t<fp8_t>(st, epi, a) : launch_gemm_x3<fp8_t>(st, epi, a);
  if (wdtype == 1) return a.f32_mfma ? launch_gemm_t<bf16_t>(st, epi, a) : launch_gemm_x3<bf16_t>(st, epi, a);
  return launch_gemm_t<float>(st, epi, a);
}

template <typename KT, typename WT>
static int launch_attn_oproj_t(hipStream_t st, const AttnOprojArgs& a) {
  const int K = a.n_q * a.hd, tpr = K / (K >= 1024 ? 16 : 8), rows = 64 * a.n_q / tpr;
  const dim3 grid(a.N / rows), block(64 * a.n_q);
  const size_t lds = ((size_t)2 * a.n_q * a.hd + (size_t)a.n_q * 32) * sizeof(float);
  const bool gqa = a.gqa && a.hd == 128 && a.n_q == 8 && a.n_kv == 2 && a.N % 8 == 0;
  const void* fn = gqa ? (const void*)attn_oproj_gqa_kernel<KT, WT>
                       : (a.hd == 64 ? (const void*)attn_oproj_kernel<KT, WT, 64> : (const void*)attn_oproj_kernel<KT, WT, 128>);
  if (a.beside_streamer) {
    // the weight streamer keeps one 72-register wave on every SIMD: a workgroup of this launch (n_q / 4 waves per SIMD)
    // must fit beside it, or the chain would stall until the streamer gives up (registers are allocated in blocks of 8)
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return -2;
    const int alloc = (fa.numRegs + 7) & ~7, per_simd = (a.n_q + 3) / 4;
    if (per_simd * alloc + ((PF_STREAMER_VGPRS + 7) & ~7) > 512) return -2;
  }
  if (gqa) {
    const size_t lds2 = ((size_t)8 * 4 * 128 * 2 + 8 * 128 + 8 * 4 * 2) * sizeof(float);
    hipLaunchKernelGGL((attn_oproj_gqa_kernel<KT, WT>), dim3(a.N / 8), dim3(512), lds2, st, AO_HOT_ARGS(a), a);
    return (int)hipGetLastError();
  }
  if (a.hd == 64) hipLaunchKernelGGL((attn_oproj_kernel<KT, WT, 64>), grid, block, lds, st, AO_HOT_ARGS(a), a);
  else hipLaunchKernelGGL((attn_oproj_kernel<KT, WT, 128>), grid, block, lds, st, AO_HOT_ARGS(a), a);
  return (int)hipGetLastError();
}
int launch_attn_oproj(hipStream_t st, int wdtype, int kvdtype, const AttnOprojArgs& a) {
  if ((a.hd != 64 && a.hd != 128) || a.lmax > 32 || a.n_q % a.n_kv) return -2;
  if (!a.pos_ptr && (a.pos_const < 0 || a.pos_const >= (1 << 23))) return -2;   // packed beside the priority bit in the preloaded word (attn_oproj.h AO_HOT_ARGS)
  if (a.n_q != 2 && a.n_q != 4 && a.n_q != 8) return -2;
  {
    const int K = a.n_q * a.hd, tpr = K / (K >= 1024 ? 16 : 8);   // lanes per output row: half a wave or a wave
    if ((tpr != 32 && tpr != 64) || a.N % (64 * a.n_q / tpr)) return -2;
  }
  if (kvdtype == 1) {
    if (wdtype == 2) return launch_attn_oproj_t<bf16_t, fp8_t>(st, a);
    if (wdtype == 1) return launch_attn_oproj_t<bf16_t, bf16_t>(st, a);
    return launch_attn_oproj_t<bf16_t, float>(st, a);
  }
  if (wdtype == 2) return launch_attn_oproj_t<float, fp8_t>(st, a);
  if (wdtype == 1) return launch_attn_oproj_t<float, bf16_t>(st, a);
  return launch_attn_oproj_t<float, float>(st, a);
}

int launch_ce_rows(hipStream_t st, const CeArgs& a) {
  if (a.rows < 1) return 0;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(a.rows), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}
int launch_ce_reduce(hipStream_t st, const float* row_loss, const int* labels, int V, int rows, double* acc) {
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, labels, V, rows, acc);
  return (int)hipGetLastError();
}
int launch_loss_finalize(hipStream_t st, const double* acc, int frames, float* out3) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, st, acc, frames, out3);
  return (int)hipGetLastError();
}
int launch_dec_input(hipStream_t st, int frames, const DecInArgs& a) {
  if (a.Hd % 4) return -1;
  hipLaunchKernelGGL(dec_input_kernel, dim3(frames, a.P), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

int launch_embed(hipStream_t st, int wdtype, int rows, const EmbedArgs& a) {
  if (a.H % 8 != 0) return -1;
  if (a.C + 1 > 64) return -1;
  const dim3 grid(rows, (a.H + 511) / 512);
  if (wdtype == 1) hipLaunchKernelGGL((embed_sum_kernel<bf16_t>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((embed_sum_kernel<float>), grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

int launch_rmsnorm(hipStream_t st, const float* x, int ldx, const float* w, int rows, int H, float eps, float* out,
                   int ldo, const int* frame_ptr, size_t frame_stride, int frame_add, bf16_t* planes, size_t plane_stride,
                   const float* part, int nsplit, size_t part_stride, int ldp, uint8_t* mxq, uint8_t* mxs) {
  if (H % 4 != 0 || (mxq && (H % 32 != 0 || !mxs))) return -1;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(256), 0, st, x, ldx, w, H, eps, out, ldo, frame_ptr,
                     frame_stride, frame_add, planes, plane_stride, part, nsplit, part_stride, ldp, mxq, mxs);
  return (int)hipGetLastError();
}

int launch_swiglu_reduce(hipStream_t st, const float* part, int nsplit, size_t part_stride, int rows, int F, float* out, int ldo,
                         bf16_t* planes, size_t plane_stride, uint8_t* mxq, uint8_t* mxs) {
  if (rows < 1 || F % 4 || nsplit < 1 || (!planes && !mxq && (!out || ldo % 4)) || (mxq && (F % 32 || !mxs))) return -1;
  const size_t quads = (size_t)rows * (F / 4);
  hipLaunchKernelGGL(swiglu_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, part, nsplit, part_stride, rows, F, out, ldo,
                     planes, plane_stride, mxq, mxs);
  return (int)hipGetLastError();
}

int launch_rope_scatter(hipStream_t st, int kvdtype, int rows, const RopeArgs& a) {
  const dim3 grid(rows, (a.nsplit > 1 && rows < 1024) ? 4 : 1);
  if (kvdtype == 1) hipLaunchKernelGGL((rope_scatter_kernel<bf16_t>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((rope_scatter_kernel<float>), grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

int configure_sample() {   // once per engine (= per device: the attribute is per device), outside any capture: the top-k path may need more than 64 KiB of LDS
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(160 * 1024 - 4096));
}
int launch_kv_convert(hipStream_t st, int kvdtype, const KvConvArgs& a) {
  const size_t n = (size_t)a.B * a.n_kv * a.len * a.hd;
  if (n == 0) return 0;
  const int grid = (int)((n + 255) / 256);
  if (kvdtype == 1) hipLaunchKernelGGL((kv_convert_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((kv_convert_kernel<float>), dim3(grid), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

int launch_kv_shift(hipStream_t st, int kvdtype, const KvShiftArgs& a) {
  const size_t n = (size_t)a.B * a.n_kv * a.len * (a.hd / 2);
  if (n == 0) return 0;
  const int grid = (int)((n + 255) / 256);
  if (kvdtype == 1) hipLaunchKernelGGL((kv_shift_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((kv_shift_kernel<float>), dim3(grid), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}
int launch_add_ints(hipStream_t st, int* p, int n, int delta) {
  hipLaunchKernelGGL(add_ints_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, n, delta);
  return (int)hipGetLastError();
}

int launch_sample(hipStream_t st, int rows, const SampleArgs& a) {
  const bool greedy = a.topk <= 1 || a.temperature == 0.f;
  // top-k: scaled logits | survivor values | survivor indices in LDS; greedy streams the row from memory
  const size_t lds = greedy ? 0 : (size_t)3 * a.V * sizeof(float);
  if (lds > 160 * 1024 - 4096) return -3;   // V > 13 300: the reference's sampler has no such limit (documented)
  hipLaunchKernelGGL(sample_kernel, dim3(rows), dim3(256), lds, st, SMP_HOT_ARGS(a), a);
  return (int)hipGetLastError();
}
__global__ void set_rng_kernel(uint64_t* p, uint64_t seed, uint64_t row_offset) { p[0] = seed; p[1] = row_offset; }
int launch_set_rng(hipStream_t st, uint64_t* p, uint64_t seed, uint64_t row_offset) {
  hipLaunchKernelGGL(set_rng_kernel, dim3(1), dim3(1), 0, st, p, seed, row_offset);
  return (int)hipGetLastError();
}

// ---- tiny utility kernels -----------------------------------------------------------------------
__global__ void rows_iota_kernel(int* row_seq, int* row_pos, int R, int S, int past) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < R) {
    row_seq[r] = r / S;
    row_pos[r] = past + r % S;
  }
}
__global__ void set_int_kernel(int* p, int v) { *p = v; }
// rows of several sequences that live in arbitrary cache slots (csm_prefill_slots): row r belongs to sequence r / S = slot slots[r / S]
__global__ void rows_slots_kernel(int* row_seq, int* row_pos, int R, int S, int past, const int* slots) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < R) {
    row_seq[r] = slots[r / S];
    row_pos[r] = past + r % S;
  }
}

template <typename WT>
__global__ void widen_rows_kernel(const WT* src, float* dst, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i < n) {
    W8<WT> w;
    w.load(src + i);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[i + e] = w.get(e);
  }
}

// Fragment-order copy of a row-major [N][K] weight matrix for the MFMA skinny GEMM (gemm16.h): 16-row tiles,
// 128-wide k chunks, and inside a chunk four 1 KiB (bf16) blocks -- block j holds, for lane l, the 8 weights
// W[tile*16 + (l & 15)][chunk*128 + j*32 + (l >> 4)*8 ..+8], i.e. exactly the A operand of one 16x16x32 MFMA, so a
// wavefront's fragment load is one fully coalesced stream.  Rows >= N are zero.  ES = element bytes (2 | 1).
template <typename V>
__global__ __launch_bounds__(256) void tile16_kernel(const V* __restrict__ W, V* __restrict__ Wt, int N, int K) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // one 8-weight fragment piece
  const int lane = (int)(i & 63);
  const size_t blk = i >> 6;            // (tile * K/128 + chunk) * 4 + j
  const int j = (int)(blk & 3);
  const size_t tc = blk >> 2;
  const int nchunk = K >> 7;
  const int chunk = (int)(tc % nchunk);
  const size_t tile = tc / nchunk;
  if (tile * 16 >= (size_t)((N + 15) & ~15)) return;
  const size_t n = tile * 16 + (lane & 15);
  const int k = chunk * 128 + j * 32 + (lane >> 4) * 8;
  V v = V{};
  if (n < (size_t)N) v = W[(n * K + k) / 8];
  Wt[i] = v;
}

// W, Wt: device pointers; esz = bytes per weight (2 = bf16, 1 = fp8); K % 128 == 0
int launch_tile16(hipStream_t st, const void* W, void* Wt, int N, int K, int esz) {
  if (K % 128) return -1;
  const size_t pieces = (size_t)((N + 15) / 16) * 16 * (K / 8);
  const int grid = (int)((pieces + 255) / 256);
  if (esz == 2) hipLaunchKernelGGL((tile16_kernel<uint4>), dim3(grid), dim3(256), 0, st, (const uint4*)W, (uint4*)Wt, N, K);
  else if (esz == 1) hipLaunchKernelGGL((tile16_kernel<uint2>), dim3(grid), dim3(256), 0, st, (const uint2*)W, (uint2*)Wt, N, K);
  else return -1;
  return (int)hipGetLastError();
}

int launch_rows_iota(hipStream_t st, int* row_seq, int* row_pos, int R, int S, int past) {
  hipLaunchKernelGGL(rows_iota_kernel, dim3((R + 255) / 256), dim3(256), 0, st, row_seq, row_pos, R, S, past);
  return (int)hipGetLastError();
}
int launch_rows_slots(hipStream_t st, int* row_seq, int* row_pos, int R, int S, int past, const int* slots) {
  hipLaunchKernelGGL(rows_slots_kernel, dim3((R + 255) / 256), dim3(256), 0, st, row_seq, row_pos, R, S, past, slots);
  return (int)hipGetLastError();
}
int launch_set_int(hipStream_t st, int* p, int v) {
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, st, p, v);
  return (int)hipGetLastError();
}
int launch_widen(hipStream_t st, int wdtype, const void* src, float* dst, size_t n) {
  if (n % 8) return -1;
  const int grid = (int)((n / 8 + 255) / 256);
  if (wdtype == 1) hipLaunchKernelGGL((widen_rows_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, dst, n);
  else hipLaunchKernelGGL((widen_rows_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)src, dst, n);
  return (int)hipGetLastError();
}

// ---- weight streamer (prefetch.h) ------------------------------------------------------------------------
int launch_pf_where(hipStream_t st, unsigned* out8) {
  hipLaunchKernelGGL(pf_where_kernel, dim3(8), dim3(256), 0, st, out8);
  return (int)hipGetLastError();
}
int launch_pf_concurrency_probe(hipStream_t waiter_stream, hipStream_t setter_stream, unsigned* flag, unsigned* out) {
  hipLaunchKernelGGL(pf_wait_kernel, dim3(1), dim3(1), 0, waiter_stream, flag, out, (long long)300000);   // 3 ms
  hipLaunchKernelGGL(pf_set_kernel, dim3(1), dim3(1), 0, setter_stream, flag);
  return (int)hipGetLastError();
}
int launch_pf_rate_probe(hipStream_t waiter_stream, hipStream_t chain_stream, unsigned* flag, unsigned* out, int n, hipEvent_t ev0, hipEvent_t ev1) {
  // the chain is a hipGraph of dependent launches (dispatched by the command processor at its own rate, like the frame-step: eager
  // launches are host-bound at ~2 us each and hide what happens between two packets)
  static thread_local hipGraphExec_t ge = nullptr;
  static thread_local int ge_n = 0, ge_dev = -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (!ge || ge_n != n || ge_dev != dev) {   // (one cached probe graph per thread; rebuilt for another device)
    if (ge) { hipGraphExecDestroy(ge); ge = nullptr; }
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(chain_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return -1;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(pf_nop_kernel, dim3(256), dim3(256), 0, chain_stream, (unsigned*)nullptr);
    if (hipStreamEndCapture(chain_stream, &g) != hipSuccess || !g) return -1;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { hipGraphDestroy(g); return -1; }
    hipGraphDestroy(g);
    ge_n = n;
    ge_dev = dev;
    hipGraphLaunch(ge, chain_stream);            // first launch: uploads
    hipStreamSynchronize(chain_stream);
  }
  if (waiter_stream) hipLaunchKernelGGL(pf_wait_kernel, dim3(256), dim3(256), 0, waiter_stream, flag, out, (long long)500000);   // 5 ms; one workgroup per CU like the streamer
  if (hipEventRecord(ev0, chain_stream) != hipSuccess) return -1;
  if (hipGraphLaunch(ge, chain_stream) != hipSuccess) return -1;
  if (hipEventRecord(ev1, chain_stream) != hipSuccess) return -1;
  hipLaunchKernelGGL(pf_set_kernel, dim3(1), dim3(1), 0, chain_stream, flag);
  return (int)hipGetLastError();
}
int launch_weight_prefetch(hipStream_t st, int grid, const PfArgs& a) {
  hipLaunchKernelGGL(weight_prefetch_kernel, dim3(grid), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}
