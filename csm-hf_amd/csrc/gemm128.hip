// Instantiations + launcher of the 65..128-row FFN GEMM (gemm128.h).
#include "gemm128.h"

template <typename WT, int PRO, int EPI, int PT, int H = 1, bool ONE = false>
static int launch_g128(hipStream_t st, int M, int KB, const GemvArgs& a, float* slabs, size_t slab_floats, int* tickets, int n_tickets) {
  if (!ONE && a.pl1) return launch_g128<WT, PRO, EPI, PT, H, true>(st, M, KB, a, slabs, slab_floats, tickets, n_tickets);
  constexpr int NP = ONE ? 1 : 3;
  const int Z = ((M + 15) / 16 + 3) / 4;
  const int gx = ((a.N + 15) / 16 + 4 * PT - 1) / (4 * PT);
  constexpr int U = 4 * PT * 4;
  if (KB > 1 && ((size_t)gx * Z * KB * U * 256 > slab_floats || gx * Z > n_tickets)) return -2;
  const size_t lds = (size_t)3 * 4 * NP * 4 * 1024 + 64 * sizeof(float);
  auto fn = gemm128_kernel<WT, PRO, EPI, PT, ONE, H>;
  if (lds > 64 * 1024) {   // raise the dynamic-LDS limit once per DEVICE (gemm128_configure_all() at engine creation; this is the safety net)
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
  G128Args g{};
  g.Wt = a.Wt; g.xplanes = a.xplanes; g.xss = a.xss; g.oplanes = a.oplanes; g.oln = a.oln; g.oss = a.oss; g.out = a.out; g.wscale = a.wscale;
  g.slabs = slabs; g.tickets = tickets; g.bump_a = a.bump_a; g.bump_b = a.bump_b; g.dbg = nullptr;
  g.xss_n = a.xss_n; g.xss_ld = a.xss_ld; g.oss_ld = a.oss_ld; g.ldo = a.ldo; g.K = a.K; g.N = a.N; g.M = M; g.KB = KB; g.eps = a.eps; g.xcdmap = a.xcdmap;
  hipLaunchKernelGGL(fn, KB > 1 ? dim3(KB, gx, Z) : dim3(gx, 1, Z), dim3(256 * H), lds, st, g);   // K split: the k group fastest (gemm128.h)
  return (int)hipGetLastError();
}

template <typename WT>
static int launch_gemm128_t(hipStream_t st, int M, int pro, int epi, const GemvArgs& a, float* slabs, size_t sf, int* tk, int nt) {
  if (a.K % 1024 != 0 || a.N % 16 != 0 || a.ldo % 4 != 0) return -2;
  int groups = a.K / 1024;   // below: the K splits (grid.y) of the launch
  if (groups > 8) return -2;   // the last arriver holds up to eight slabs per accumulator in registers
  // K split across workgroups = one 1 024-wide group each (the association of the narrower kernels); tall matrices (gate/up) walk all
  // of K in one workgroup.  Shapes (weight tiles per wave; profiles/r05_g128_shapes.txt): 2 for the tall matrices, 1 for the K-split ones
  const int shape = a.g128_shape & 0xff;   // A/B override: low nibble = weight tiles per wave (1 / 2), bit 6 = one k group per workgroup for gate/up
  const int pt = shape & 15;
  if (pro == PRO_NORM && epi == EPI_SWIGLU) {
    // gate/up: all of K in one workgroup (K = 2 048: two groups in sequence, 37 us against 49 with two workgroups + slab exchange)
    if (!(shape & 64)) groups = 1;
    if (pt == 1) return launch_g128<WT, PRO_NORM, EPI_SWIGLU, 1>(st, M, groups, a, slabs, sf, tk, nt);
    return launch_g128<WT, PRO_NORM, EPI_SWIGLU, 2>(st, M, groups, a, slabs, sf, tk, nt);
  }
  if (pro == PRO_PLAIN && epi == EPI_RESID && a.K > 2048) {   // down_proj; the o_proj launches (K = 1 024 / 2 048: 8-16 workgroups here) stay on gemm32.h
    // one weight tile per wave while that fills the chip in one round (decoder: 16 panels x 8 k groups x 2 row groups = 256 workgroups);
    // two for the backbone's 2 048 rows (the same 256 instead of 512 in two rounds)
    const bool two = pt ? pt == 2 : (size_t)(a.N / 64) * groups * ((M + 63) / 64) > 256;
    if (two) return launch_g128<WT, PRO_PLAIN, EPI_RESID, 2>(st, M, groups, a, slabs, sf, tk, nt);
    return launch_g128<WT, PRO_PLAIN, EPI_RESID, 1>(st, M, groups, a, slabs, sf, tk, nt);
  }
  return -2;
}

int launch_gemm128(hipStream_t st, int wdtype, int M, int pro, int epi, const GemvArgs& a, float* slabs, size_t slab_floats,
                   int* tickets, int n_tickets) {
  if ((wdtype != 1 && wdtype != 2) || M < 33 || M > 128 || !a.xplanes || !a.Wt) return -2;
  if (wdtype == 2) return launch_gemm128_t<fp8_t>(st, M, pro, epi, a, slabs, slab_floats, tickets, n_tickets);
  return launch_gemm128_t<bf16_t>(st, M, pro, epi, a, slabs, slab_floats, tickets, n_tickets);
}

int gemm128_configure_all() {
  static bf16_t dummy_planes[8];
  GemvArgs a{};
  a.configure_only = 1;
  a.xplanes = dummy_planes; a.Wt = dummy_planes;
  a.N = 2048; a.ldo = 2048;
  static const int combos[2][2] = {{PRO_NORM, EPI_SWIGLU}, {PRO_PLAIN, EPI_RESID}};
  for (int wd = 1; wd <= 2; ++wd)
    for (auto& c : combos)
      for (int K : {1024, 8192})
        for (int one = 0; one < 2; ++one)
          for (int shape : {0, 1, 2}) {
            a.K = K; a.pl1 = one; a.g128_shape = shape;
            const int r = launch_gemm128(nullptr, wd, 128, c[0], c[1], a, nullptr, (size_t)1 << 30, nullptr, 1 << 20);
            if (r != 0 && r != -2) return r;
          }
  return 0;
}
