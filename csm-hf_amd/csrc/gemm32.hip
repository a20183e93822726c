// Instantiations + launcher of the 32-row MFMA skinny GEMM (gemm32.h).
#include "gemm32.h"

template <typename WT, typename KT, int PRO, int EPI, int NW, int PT, int MT, bool ONE = false>
static int launch_g32(hipStream_t st, int M, int KB, const GemvArgs& a, float* slabs, size_t slab_floats, int* tickets,
                      int n_tickets) {
  if (!ONE && a.pl1) return launch_g32<WT, KT, PRO, EPI, NW, PT, MT, true>(st, M, KB, a, slabs, slab_floats, tickets, n_tickets);
  constexpr int U = PT * MT;
  int gx;
  if (EPI == EPI_QKV) gx = (a.n_q + 2 * a.n_kv) * ((a.hd >> 1) / 16);
  else gx = ((a.N + 15) / 16 + PT - 1) / PT;
  if (KB > 1 && ((size_t)gx * KB * U * 256 > slab_floats || gx > n_tickets)) return -2;
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
  hipLaunchKernelGGL(fn, dim3(gx, KB), dim3(64 * NW), lds, st, a, M, KB, slabs, tickets);
  return (int)hipGetLastError();
}

template <typename WT, typename KT, int PRO, int EPI, int PT>
static int launch_nw32(hipStream_t st, int M, int nw, int KB, const GemvArgs& a, float* slabs, size_t sf, int* tk, int nt) {
  if (nw != 8) return -2;
  if (M > 64) {   // 65..128 rows (round 4): eight batch tiles per weight fragment -- a 128-row batch streams the weights ONCE.  One quad per
    // thread: one weight tile per panel, except the QKV panels that pair the two RoPE halves of a head (element epilogue)
    // Two weight tiles per panel wherever the launch asked for at least two (the activation planes of a batch tile are loaded once
    // per PANEL: one-tile panels moved twice the plane bytes of two 64-row passes, 17.1 ms per 128-row step); 144 KiB of LDS
    return launch_g32<WT, KT, PRO, EPI, 8, (PT >= 2 ? 2 : 1), 8>(st, M, KB, a, slabs, sf, tk, nt);
  }
  if (M > 32) {   // 33..64 rows: four batch tiles per weight fragment (panels of at most two weight tiles: LDS, one quad per thread)
    if (PT > 2) return -2;
    return launch_g32<WT, KT, PRO, EPI, 8, (PT > 2 ? 2 : PT), 4>(st, M, KB, a, slabs, sf, tk, nt);
  }
  return launch_g32<WT, KT, PRO, EPI, 8, PT, 2>(st, M, KB, a, slabs, sf, tk, nt);
}

template <typename WT>
static int launch_gemm32_t(hipStream_t st, int kvdtype, int M, int pro, int epi, const GemvArgs& a, float* slabs,
                           size_t slab_floats, int* tickets, int n_tickets) {
  if (pro != PRO_PLAIN && pro != PRO_NORM) return -2;
  if (a.K % 1024 != 0) return -2;                          // 8 or 16 waves, one 128-wide chunk each
  if ((epi == EPI_RESID || epi == EPI_SWIGLU) && (a.N % 16 != 0 || a.ldo % 4 != 0)) return -2;
  // 8 waves per workgroup always (two batch tiles need ~170 VGPRs: a 16-wave workgroup would spill), one 128-wide
  // chunk per wave, the rest of K split across workgroups -- allowed for the normed launches too, because on planes the
  // RMS scale is applied by the (last-arriver) epilogue
  const int nchunks = a.K / 128;
  const int nw = 8, KB = nchunks / 8;
  if (nchunks % 8 || KB > 16) return -2;
  if (epi == EPI_QKV) {
    if (pro != PRO_NORM || (a.hd != 64 && a.hd != 128)) return -2;
    if (kvdtype == 1) return launch_nw32<WT, bf16_t, PRO_NORM, EPI_QKV, 2>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);
    return launch_nw32<WT, float, PRO_NORM, EPI_QKV, 2>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);
  }
  // panels: gate/up 32 rows (as on planes at M <= 16), the residual launches 16 rows (64 when that leaves >= 128
  // workgroups per K split), the heads 16 rows
  const int ntiles = (a.N + 15) / 16;
  if (pro == PRO_NORM && epi == EPI_SWIGLU) return launch_nw32<WT, float, PRO_NORM, EPI_SWIGLU, 2>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);
  if (pro == PRO_PLAIN && epi == EPI_RESID) {
    if (KB > 1 && ntiles >= 128 && M <= 32) return launch_nw32<WT, float, PRO_PLAIN, EPI_RESID, 4>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);
    if (KB > 1 && ntiles >= 128) return launch_nw32<WT, float, PRO_PLAIN, EPI_RESID, 2>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);
    return launch_nw32<WT, float, PRO_PLAIN, EPI_RESID, 1>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);
  }
  if (pro == PRO_NORM && epi == EPI_STORE) return launch_nw32<WT, float, PRO_NORM, EPI_STORE, 1>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);
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
            for (int rows : {64, 128}) {
              a.pl1 = one;
              const int r = launch_gemm32(nullptr, wd, kd, rows, c[0], c[1], a, nullptr, (size_t)1 << 30, nullptr, 1 << 20);
              if (r != 0 && r != -2) return r;
            }
        }
  return 0;
}
