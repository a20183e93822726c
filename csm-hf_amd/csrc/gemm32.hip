// Instantiations + launcher of the 32-row MFMA skinny GEMM (gemm32.h).
#include "gemm32.h"

template <typename WT, typename KT, int PRO, int EPI, int NW, int PT, int MT, bool ONE = false>
static int launch_g32(hipStream_t st, int M, int KB, const GemvArgs& a, float* slabs, size_t slab_floats, int* tickets,
                      int n_tickets) {
  if (!ONE && a.pl1) return launch_g32<WT, KT, PRO, EPI, NW, PT, MT, true>(st, M, KB, a, slabs, slab_floats, tickets, n_tickets);
  const int Z = ((M + 15) / 16 + MT - 1) / MT;   // row groups of MT batch tiles, one workgroup each per panel (blockIdx.z)
  constexpr int U = PT * MT;
  int gx;
  if (EPI == EPI_QKV) gx = (a.n_q + 2 * a.n_kv) * ((a.hd >> 1) / 16);
  else gx = ((a.N + 15) / 16 + PT - 1) / PT;
  if (KB > 1 && ((size_t)gx * Z * KB * U * 256 > slab_floats || gx * Z > n_tickets)) return -2;
  const size_t lds = ((size_t)NW * U * 256 + U * 256 + 16 + 16 * MT) * sizeof(float);
  auto fn = gemm32_kernel<WT, KT, PRO, EPI, NW, PT, MT, ONE>;
  if (lds > 64 * 1024) {   // four batch tiles x two weight tiles: 72 KiB -- raise the dynamic-LDS limit once per DEVICE
    // (gemm32_configure_all() does this at engine creation, outside any stream capture; this is the safety net)
    static unsigned long long configured = 0ull;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(configured & bit)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
      configured |= bit;
    }
  }
  if (a.configure_only) return 0;
  if (M > 255 || KB > 255 || (EPI == EPI_QKV && (a.hd > 255 || a.n_q > 255 || a.n_kv > 255 || (!a.row_pos && !a.pos_ptr && (a.pos_const < 0 || a.pos_const > 255))))) return -2;   // packed into the preloaded words (gemm32.h G32_HOT_ARGS)
  hipLaunchKernelGGL(fn, (KB > 1 && a.kfast) ? dim3(KB, gx, Z) : dim3(gx, KB, Z), dim3(64 * NW), lds, st, G32_HOT_ARGS(a, M, KB, EPI), a, slabs, tickets);
  return (int)hipGetLastError();
}

// (weight tiles per panel PT, batch tiles per workgroup MT) -> instantiation.  U = PT * MT accumulator tiles <= 16 (LDS: 9 KiB each).
// Rows beyond MT tiles go to further workgroups of the same panel (blockIdx.z): they re-read the panel's weights from the caches.
// With many rows the activation planes, not the weights, are the larger operand (6 bytes per row and k against 2 per weight
// row and k), so tall panels (PT = 4) with few batch tiles move the fewest bytes.
template <typename WT, typename KT, int PRO, int EPI>
static int launch_sel(hipStream_t st, int M, int pt, int mt, int KB, const GemvArgs& a, float* slabs, size_t sf, int* tk, int nt) {
#define G32_CASE(P, T) if (pt == P && mt == T) return launch_g32<WT, KT, PRO, EPI, 8, P, T>(st, M, KB, a, slabs, sf, tk, nt)
  G32_CASE(2, 2);
  if constexpr (EPI == EPI_SWIGLU) { G32_CASE(4, 4); G32_CASE(4, 2); }
  if constexpr (EPI == EPI_RESID) { G32_CASE(1, 1); G32_CASE(4, 1); G32_CASE(2, 4); G32_CASE(4, 4); }
  if constexpr (EPI == EPI_STORE) { G32_CASE(1, 2); }
#undef G32_CASE
  return -2;
}

template <typename WT>
static int launch_gemm32_t(hipStream_t st, int kvdtype, int M, int pro, int epi, const GemvArgs& a, float* slabs,
                           size_t slab_floats, int* tickets, int n_tickets) {
  if (pro != PRO_PLAIN && pro != PRO_NORM) return -2;
  if (a.K % 1024 != 0) return -2;                          // 8 waves, one 128-wide chunk each
  if ((epi == EPI_RESID || epi == EPI_SWIGLU) && (a.N % 16 != 0 || a.ldo % 4 != 0)) return -2;
  // 8 waves per workgroup always (two batch tiles need ~170 VGPRs: a 16-wave workgroup would spill), one 128-wide
  // chunk per wave, the rest of K split across workgroups -- allowed for the normed launches too, because on planes the
  // RMS scale is applied by the (last-arriver) epilogue
  const int nchunks = a.K / 128;
  const int KB = nchunks / 8;
  if (nchunks % 8 || KB > 16) return -2;
  // shapes (weight tiles per panel, batch tiles per workgroup; rows beyond go to blockIdx.z), measured at 24 / 32 / 64 / 128 rows
  // (profiles/r04_g32_shapes.txt); a row's arithmetic does not depend on the shape (tests: logits bitwise against 16-row launches):
  //   QKV (2, 2) always -- a 128-row QKV launch has 48 panels: one workgroup per panel left 200 CUs idle while each wave walked
  //       786 KB of planes.  ((2, 1) is NOT used: its RoPE epilogue compiles to a different multiply-add contraction, 1e-6 off)
  //   o_proj: (1, 1) up to 32 rows, (2, 2) beyond;  heads: (1, 2) up to 32 rows, (2, 2) beyond
  //   gate/up: (4, 2) up to 32 rows, (4, 4) beyond;  K-split down_proj: (4, 1) up to 32 rows, (2, 4) up to 64, (4, 4) beyond
  int pt = 2, mt = 2;
  if (M <= 32) {
    if (epi == EPI_RESID) { pt = a.K > 2048 ? 4 : 1; mt = 1; }
    if (epi == EPI_STORE) pt = 1;
    if (epi == EPI_SWIGLU) pt = 4;
  } else if (epi == EPI_SWIGLU) {
    pt = 4; mt = 4;
  } else if (epi == EPI_RESID && a.K > 2048) {
    pt = M > 64 ? 4 : 2; mt = 4;
  }
  if (epi == EPI_QKV) {
    if (pro != PRO_NORM || (a.hd != 64 && a.hd != 128)) return -2;
    if (kvdtype == 1) return launch_sel<WT, bf16_t, PRO_NORM, EPI_QKV>(st, M, 2, 2, KB, a, slabs, slab_floats, tickets, n_tickets);
    return launch_sel<WT, float, PRO_NORM, EPI_QKV>(st, M, 2, 2, KB, a, slabs, slab_floats, tickets, n_tickets);
  }
  if (pro == PRO_NORM && epi == EPI_SWIGLU) return launch_sel<WT, float, PRO_NORM, EPI_SWIGLU>(st, M, pt, mt, KB, a, slabs, slab_floats, tickets, n_tickets);
  if (pro == PRO_PLAIN && epi == EPI_RESID) return launch_sel<WT, float, PRO_PLAIN, EPI_RESID>(st, M, pt, mt, KB, a, slabs, slab_floats, tickets, n_tickets);
  if (pro == PRO_NORM && epi == EPI_STORE) return launch_sel<WT, float, PRO_NORM, EPI_STORE>(st, M, pt, mt, KB, a, slabs, slab_floats, tickets, n_tickets);
  return -2;
}

int launch_gemm32(hipStream_t st, int wdtype, int kvdtype, int M, int pro, int epi, const GemvArgs& a, float* slabs,
                  size_t slab_floats, int* tickets, int n_tickets) {
  if ((wdtype != 1 && wdtype != 2) || M < 17 || M > 128 || !a.xplanes || !a.Wt) return -2;
  if (wdtype == 2) return launch_gemm32_t<fp8_t>(st, kvdtype, M, pro, epi, a, slabs, slab_floats, tickets, n_tickets);
  return launch_gemm32_t<bf16_t>(st, kvdtype, M, pro, epi, a, slabs, slab_floats, tickets, n_tickets);
}

// the dynamic-LDS limit of every instantiation that needs more than 64 KiB (four batch tiles x two weight tiles), once per
// engine = per device, outside any stream capture (the launches themselves run inside captured frame-steps)
int gemm32_configure_all() {
  static bf16_t dummy_planes[8];
  GemvArgs a{};
  a.configure_only = 1;
  a.xplanes = dummy_planes; a.Wt = dummy_planes;
  a.K = 1024; a.N = 2048; a.ldo = 2048; a.hd = 128; a.n_q = 8; a.n_kv = 2;
  static const int combos[4][2] = {{PRO_NORM, EPI_QKV}, {PRO_NORM, EPI_SWIGLU}, {PRO_PLAIN, EPI_RESID}, {PRO_NORM, EPI_STORE}};
  for (int wd = 1; wd <= 2; ++wd)
    for (int kd = 0; kd < 2; ++kd)
      for (auto& c : combos)
        for (int K : {1024, 8192}) {   // KB == 1 and the K-split panel choice of the residual launches
          if (K == 8192 && c[0] == PRO_NORM) continue;
          a.K = K;
          for (int one = 0; one < 2; ++one)
            for (int rows : {32, 64, 128}) {
              a.pl1 = one;
              const int r = launch_gemm32(nullptr, wd, kd, rows, c[0], c[1], a, nullptr, (size_t)1 << 30, nullptr, 1 << 20);
              if (r != 0 && r != -2) return r;
            }
        }
  return 0;
}
