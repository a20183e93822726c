// Instantiations + launchers of the MX-fp8 prefill GEMM and the row quantiser (gemm_mx.h).
#include <hip/hip_runtime.h>

#include "gemm256.h"

template <int EPI, int BM = 128>
static int launch_mx_epi(hipStream_t st, const dim3& grid, const GemmMxArgs& a) {
  constexpr size_t lds = 2 * (BM * 128 + 128 * 128 + 2 * 128 * 4);   // two stages of (A tile | W tile | scales) = 66 KiB at BM = 128
  auto fn = gemm_mx_kernel<EPI, BM>;
  // more than 64 KiB of dynamic LDS: raise the limit once per DEVICE (the attribute is per device)
  static unsigned long long configured = 0ull;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(configured & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    configured |= bit;
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, a);
  return (int)hipGetLastError();
}

int launch_gemm_mx(hipStream_t st, int epi, const GemmMxArgs& a) {
  if (a.R < 1 || a.N % 128 || a.K % 128 || !a.Aq || !a.As || !a.Wq || !a.Ws) return -2;
  if (a.big > 0) {
    const int r = launch_gemm256_mx(st, epi, a, a.big);
    if (r != -2) return r;
  }
  const int ks = epi == GEPI_PARTIAL ? a.ksplit : 1;
  if (ks < 1 || a.K % (128 * ks)) return -2;
  // short prefills (GemmMxArgs::skinny = row bound): 64 / 32 activation rows per workgroup for the split-K / SwiGLU launches
  const int bm = (a.skinny > 0 && (epi == GEPI_PARTIAL || epi == GEPI_SWIGLU)) ? (a.R <= 32 ? 32 : (a.R <= a.skinny ? 64 : 128)) : 128;
  const dim3 grid(((a.R + bm - 1) / bm) * (a.N / 128), ks);
  if (bm == 64) return epi == GEPI_PARTIAL ? launch_mx_epi<GEPI_PARTIAL, 64>(st, grid, a) : launch_mx_epi<GEPI_SWIGLU, 64>(st, grid, a);
  if (bm == 32) return epi == GEPI_PARTIAL ? launch_mx_epi<GEPI_PARTIAL, 32>(st, grid, a) : launch_mx_epi<GEPI_SWIGLU, 32>(st, grid, a);
  switch (epi) {
    case GEPI_STORE: return launch_mx_epi<GEPI_STORE>(st, grid, a);
    case GEPI_RESID: return launch_mx_epi<GEPI_RESID>(st, grid, a);
    case GEPI_SWIGLU: return launch_mx_epi<GEPI_SWIGLU>(st, grid, a);
    case GEPI_PARTIAL: return launch_mx_epi<GEPI_PARTIAL>(st, grid, a);
    case GEPI_ROPE: return launch_mx_epi<GEPI_ROPE>(st, grid, a);
    default: return -1;
  }
}

int launch_mx_quant(hipStream_t st, const MxQuantArgs& a) {
  if (a.rows < 1 || a.K % 32 || a.ldx % 4) return -2;
  const size_t nblk = (size_t)a.rows * (a.K / 32);
  hipLaunchKernelGGL(mx_quant_rows_kernel, dim3((unsigned)((nblk + 63) / 64)), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

template <int EPI, int NPL, int BM = 128>
static int launch_dma_epi(hipStream_t st, const dim3& grid, const GemmArgs& a) {
  constexpr size_t lds = 2 * (NPL * BM * 128 + 128 * 128);   // two stages of (A planes | W tile): 64 KiB (one plane) / 128 KiB (three) at BM = 128
  auto fn = gemm_dma_bf16_kernel<EPI, NPL, BM>;
  if (lds > 64 * 1024) {   // raise the dynamic-LDS limit once per DEVICE
    static unsigned long long configured = 0ull;
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
int launch_gemm_dma_bf16(hipStream_t st, int epi, const GemmArgs& a) {
  if (!a.Aplanes || !a.W || a.wscale || a.R < 1 || a.N % 128 || a.K % 64 || a.ldc % 4) return -2;
  const bool exact = a.a_plane_stride != 0;
  if (epi == GEPI_SWIGLU && a.Cplanes && (a.c_plane_stride != 0) != exact) return -2;
  const int ks = epi == GEPI_PARTIAL ? a.ksplit : 1;
  if (ks < 1 || a.K % (64 * ks) || (epi == GEPI_PARTIAL && !a.Cpart)) return -2;
  // short prefills (GemmArgs::dma_skinny): 64 / 32 activation rows per workgroup instead of 128 -- no matrix work on dead rows
  // Measured (whole csm-1b prefills, profiles/r03_prefill_skinny_tiles.txt): one plane: 64 rows per workgroup win up to 256 rows
  // (192 rows 1.86 -> 1.77 ms) and lose from 384 on (512 rows 2.34 -> 2.61); three planes: they win up to 768 rows (384 / 768 rows 4.36 / 7.33 -> 3.67 / 6.71), tie at 1024 and lose at 2048 (16.2 -> 21.7)
  const long wgs128 = (long)((a.R + 127) / 128) * (a.N / 128) * ks;   // dma_skinny bit 2 (value 4, A/B): 64-row workgroups only for launches short of 512 workgroups
  const int bm = (a.dma_skinny && (epi == GEPI_PARTIAL || epi == GEPI_SWIGLU) && !((a.dma_skinny & 4) && wgs128 >= 512 && a.R > 64))
                     ? (a.R <= 32 ? 32 : (a.R <= (exact ? ((a.dma_skinny & 2) ? 4096 : 768) : 256) ? 64 : 128)) : 128;
  const dim3 grid(((a.R + bm - 1) / bm) * (a.N / 128), ks);
  if (bm != 128) {
#define DMA_BM(E) return bm == 64 ? (exact ? launch_dma_epi<E, 3, 64>(st, grid, a) : launch_dma_epi<E, 1, 64>(st, grid, a)) \
                                  : (exact ? launch_dma_epi<E, 3, 32>(st, grid, a) : launch_dma_epi<E, 1, 32>(st, grid, a))
    if (epi == GEPI_PARTIAL) DMA_BM(GEPI_PARTIAL);
    DMA_BM(GEPI_SWIGLU);
#undef DMA_BM
  }
#define DMA_EPI(E) return exact ? launch_dma_epi<E, 3>(st, grid, a) : launch_dma_epi<E, 1>(st, grid, a)
  switch (epi) {
    case GEPI_STORE: DMA_EPI(GEPI_STORE);
    case GEPI_RESID: DMA_EPI(GEPI_RESID);
    case GEPI_SWIGLU: DMA_EPI(GEPI_SWIGLU);
    case GEPI_PARTIAL: DMA_EPI(GEPI_PARTIAL);
    case GEPI_ROPE: DMA_EPI(GEPI_ROPE);
    default: return -1;
  }
#undef DMA_EPI
}

// ---- 256 x 256 tile (gemm256.h) ----------------------------------------------------------------------------------------
template <int EPI, bool MX, int VAR, typename ARGS>
static int launch_256_var(hipStream_t st, const dim3& grid, const ARGS& a) {
  constexpr size_t lds = 2 * (2 * 256 * 128 + (MX ? 2 * 256 * 4 : 0));   // two stages: 128 KiB (+ 4 KiB of scales)
  auto fn = gemm256_kernel<EPI, MX, VAR, ARGS>;
  static unsigned long long configured = 0ull;   // the dynamic-LDS limit is a per-DEVICE attribute of the function
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(configured & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    configured |= bit;
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), lds, st, a);
  return (int)hipGetLastError();
}

// min_wgs: low 24 bits = workgroups (tiles x K splits) a launch must have, bits 24-26 = schedule variant (gemm256.h VAR, A/B).
// Measured (profiles/r03_gemm256.txt): VAR 1 (weight fragments in halves) is 1-3 % ahead of VAR 0; VAR 2 (DMA pieces spread over
// the phases, buffer form) wins 5-10 % on single 2048-row GEMMs and LOSES 3-6 % on whole prefills and at 8192 rows.
static inline int g256_variant(int min_wgs) {   // 0 = the default schedule (VAR 1), 1 = VAR 0, 2 = VAR 2, 3-5 = knock-outs
  const int sel = (min_wgs >> 24) & 7;
  return sel == 0 ? 1 : (sel == 1 ? 0 : sel);
}
template <int EPI, bool MX, typename ARGS>
static int launch_256_epi(hipStream_t st, const dim3& grid, const ARGS& a, int var) {
  if (var == 0) return launch_256_var<EPI, MX, 0, ARGS>(st, grid, a);
  if (var == 1) return launch_256_var<EPI, MX, 1, ARGS>(st, grid, a);
  if constexpr (EPI == GEPI_STORE && MX) {   // knock-out variants of the microbenchmark (wrong results by construction)
    if (var == 3) return launch_256_var<EPI, MX, 3, ARGS>(st, grid, a);
    if (var == 4) return launch_256_var<EPI, MX, 4, ARGS>(st, grid, a);
    if (var == 5) return launch_256_var<EPI, MX, 5, ARGS>(st, grid, a);
  }
  return launch_256_var<EPI, MX, 2, ARGS>(st, grid, a);
}

int launch_gemm256_bf16(hipStream_t st, int epi, const GemmArgs& a, int min_wgs) {
  const int var = g256_variant(min_wgs);
  min_wgs &= 0xffffff;
  if (!a.Aplanes || a.a_plane_stride != 0 || !a.W || a.wscale || a.R < 256 || a.R % 256 || a.N % 256 || a.K % 64 || a.ldc % 4) return -2;
  if (epi == GEPI_SWIGLU && a.Cplanes && a.c_plane_stride != 0) return -2;
  const int ks = epi == GEPI_PARTIAL ? a.ksplit : 1;
  if (ks < 1 || a.K % (64 * ks) || (epi == GEPI_PARTIAL && !a.Cpart)) return -2;
  const long tiles = (long)((a.R + 255) / 256) * (a.N / 256);
  if (tiles * ks < min_wgs) return -2;
  const dim3 grid((unsigned)tiles, ks);
  switch (epi) {
    case GEPI_STORE: return launch_256_epi<GEPI_STORE, false, GemmArgs>(st, grid, a, var);
    case GEPI_RESID: return launch_256_epi<GEPI_RESID, false, GemmArgs>(st, grid, a, var);
    case GEPI_SWIGLU: return launch_256_epi<GEPI_SWIGLU, false, GemmArgs>(st, grid, a, var);
    case GEPI_PARTIAL: return launch_256_epi<GEPI_PARTIAL, false, GemmArgs>(st, grid, a, var);
    case GEPI_ROPE: return launch_256_var<GEPI_ROPE, false, 1, GemmArgs>(st, grid, a);   // default schedule only
    default: return -1;
  }
}

int launch_gemm256_mx(hipStream_t st, int epi, const GemmMxArgs& a, int min_wgs) {
  const int var = g256_variant(min_wgs);
  min_wgs &= 0xffffff;
  if (a.R < 256 || a.R % 256 || a.N % 256 || a.K % 128 || !a.Aq || !a.As || !a.Wq || !a.Ws) return -2;
  const int ks = epi == GEPI_PARTIAL ? a.ksplit : 1;
  if (ks < 1 || a.K % (128 * ks)) return -2;
  const long tiles = (long)((a.R + 255) / 256) * (a.N / 256);
  if (tiles * ks < min_wgs) return -2;
  const dim3 grid((unsigned)tiles, ks);
  switch (epi) {
    case GEPI_STORE: return launch_256_epi<GEPI_STORE, true, GemmMxArgs>(st, grid, a, var);
    case GEPI_RESID: return launch_256_epi<GEPI_RESID, true, GemmMxArgs>(st, grid, a, var);
    case GEPI_SWIGLU: return launch_256_epi<GEPI_SWIGLU, true, GemmMxArgs>(st, grid, a, var);
    case GEPI_PARTIAL: return launch_256_epi<GEPI_PARTIAL, true, GemmMxArgs>(st, grid, a, var);
    case GEPI_ROPE: return launch_256_var<GEPI_ROPE, true, 1, GemmMxArgs>(st, grid, a);   // default schedule only
    default: return -1;
  }
}
