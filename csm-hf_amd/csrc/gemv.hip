// Instantiations + host launcher of the decode GEMV family (see gemv.h).
#include "gemv.h"


template <typename WT, typename KT, int M, int PRO, int EPI>
static int launch_one(hipStream_t st, const GemvArgs& a) {
  auto fn = gemv_kernel<WT, KT, M, PRO, EPI>;
  if (a.configure_only) {  // done once at engine creation, outside any stream capture
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    return (int)e;
  }
  const size_t lds = ((size_t)M * a.K + (size_t)M * 4) * sizeof(float);
  if (lds > 160 * 1024) return -1;
  const int ntask = (EPI == EPI_QKV) ? (a.N >> 1) : ((a.N + 1) >> 1);
  int grid = (ntask + 3) / 4;
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, a);
  return (int)hipGetLastError();
}

template <typename WT, typename KT, int M>
static int launch_m(hipStream_t st, int pro, int epi, const GemvArgs& a) {
  if (pro == PRO_PLAIN && epi == EPI_STORE) return launch_one<WT, KT, M, PRO_PLAIN, EPI_STORE>(st, a);
  if (pro == PRO_NORM && epi == EPI_STORE) return launch_one<WT, KT, M, PRO_NORM, EPI_STORE>(st, a);
  if (pro == PRO_PLAIN && epi == EPI_RESID) return launch_one<WT, KT, M, PRO_PLAIN, EPI_RESID>(st, a);
  if (pro == PRO_NORM && epi == EPI_SWIGLU) return launch_one<WT, KT, M, PRO_NORM, EPI_SWIGLU>(st, a);
  if (pro == PRO_NORM && epi == EPI_QKV) return launch_one<WT, KT, M, PRO_NORM, EPI_QKV>(st, a);
  return -1;
}

template <typename WT, typename KT>
static int launch_wk(hipStream_t st, int M, int pro, int epi, const GemvArgs& a) {
  switch (M) {
    case 1: return launch_m<WT, KT, 1>(st, pro, epi, a);
    case 2: return launch_m<WT, KT, 2>(st, pro, epi, a);
    case 3: return launch_m<WT, KT, 3>(st, pro, epi, a);
    case 4: return launch_m<WT, KT, 4>(st, pro, epi, a);
  }
  return -1;
}

// wdtype/kvdtype: 0 = fp32, 1 = bf16.  KT only matters for EPI_QKV; other epilogues use the fp32
// instantiation to keep the kernel count down.
int launch_gemv(hipStream_t st, int wdtype, int kvdtype, int M, int pro, int epi, const GemvArgs& a) {
  if (!a.configure_only && (a.K % 8 != 0 || a.K < 8)) return -1;
  const bool kvb = (epi == EPI_QKV) && kvdtype == 1;
  if (wdtype == 1) {
    return kvb ? launch_wk<bf16_t, bf16_t>(st, M, pro, epi, a) : launch_wk<bf16_t, float>(st, M, pro, epi, a);
  }
  return kvb ? launch_wk<float, bf16_t>(st, M, pro, epi, a) : launch_wk<float, float>(st, M, pro, epi, a);
}

// set the dynamic-LDS limit on every instantiation (call once per process before any capture)
int gemv_configure_all() {
  GemvArgs a{};
  a.configure_only = 1;
  static const int combos[5][2] = {{PRO_PLAIN, EPI_STORE}, {PRO_NORM, EPI_STORE}, {PRO_PLAIN, EPI_RESID},
                                   {PRO_NORM, EPI_SWIGLU}, {PRO_NORM, EPI_QKV}};
  for (int wd = 0; wd < 2; ++wd)
    for (int kd = 0; kd < 2; ++kd)
      for (int M = 1; M <= 4; ++M)
        for (auto& c : combos) {
          if (c[1] != EPI_QKV && kd == 1) continue;
          int r = launch_gemv(nullptr, wd, kd, M, c[0], c[1], a);
          if (r) return r;
        }
  return 0;
}
