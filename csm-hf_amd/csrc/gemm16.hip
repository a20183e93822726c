// Instantiations + launcher of the MFMA skinny GEMM (gemm16.h).
#include "gemm16.h"

template <typename WT, typename KT, int PRO, int EPI, int NW, int PT>
static int launch_g16(hipStream_t st, int M, int KB, const GemvArgs& a, float* slabs, size_t slab_floats, int* tickets,
                      int n_tickets) {
  int gx;
  if (EPI == EPI_QKV) gx = (a.n_q + 2 * a.n_kv) * ((a.hd >> 1) / 16);
  else gx = ((a.N + 15) / 16 + PT - 1) / PT;
  if (KB > 1 && ((size_t)gx * KB * PT * 256 > slab_floats || gx > n_tickets)) return -2;
  const size_t lds = ((size_t)NW * PT * 256 + PT * 256 + 16 + NW * 16) * sizeof(float);
  if (a.xplanes && !a.Wt) return -2;   // planes are laid out for the fragment-order k order
  if (KB > 31 || M > 31 || (EPI == EPI_QKV && (a.hd > 248 || (a.hd & 7) || a.n_q > 127 || a.n_kv > 63))) return -2;   // packed into the preloaded word (gemm16.h G16_HOT_ARGS)
  if (a.geom_out && a.Wt) {            // weight streamer (prefetch.h, kind 2): workgroup (bx, by) -> fragment blocks
    PfGeom& g = *a.geom_out;
    g.W = a.Wt; g.N = a.N; g.K = a.K; g.esz = (int)sizeof(WT); g.kind = 2;
    g.grid = gx * KB; g.tpb = PT; g.iters = NW; g.stride = gx; g.ntask = a.K / 128;
    g.hd = a.hd; g.n_rope_heads = (EPI == EPI_QKV) ? a.n_q + a.n_kv : 0;
    // does a workgroup of this launch leave room for one streamer wave per SIMD?  (registers are allocated in blocks of 8)
    hipFuncAttributes fa{};
    const void* fn = a.xplanes ? (const void*)gemm16_kernel<WT, KT, PRO, EPI, NW, PT, true, true>
                               : (const void*)gemm16_kernel<WT, KT, PRO, EPI, NW, PT, true, false>;
    if (hipFuncGetAttributes(&fa, fn) == hipSuccess) {
      const int alloc = (fa.numRegs + 7) & ~7, per_simd = (NW + 3) / 4;
      g.exclusive = per_simd * alloc + ((PF_STREAMER_VGPRS + 7) & ~7) > 512;
    } else {
      g.exclusive = NW >= 16;
    }
  }
  if (a.pl1 && !a.xplanes && a.oplanes) return -1;   // one-plane mode exists on the planes path only (every producer of planes is itself fed planes)
  if (a.xplanes && a.pl1) hipLaunchKernelGGL((gemm16_kernel<WT, KT, PRO, EPI, NW, PT, true, true, true>), (KB > 1 && a.kfast) ? dim3(KB, gx) : dim3(gx, KB), dim3(64 * NW), lds, st, G16_HOT_ARGS(a, M, KB, true, true, EPI), a, slabs, tickets);
  else if (a.xplanes) hipLaunchKernelGGL((gemm16_kernel<WT, KT, PRO, EPI, NW, PT, true, true>), (KB > 1 && a.kfast) ? dim3(KB, gx) : dim3(gx, KB), dim3(64 * NW), lds, st, G16_HOT_ARGS(a, M, KB, true, true, EPI), a, slabs, tickets);
  else if (a.Wt) hipLaunchKernelGGL((gemm16_kernel<WT, KT, PRO, EPI, NW, PT, true, false>), (KB > 1 && a.kfast) ? dim3(KB, gx) : dim3(gx, KB), dim3(64 * NW), lds, st, G16_HOT_ARGS(a, M, KB, true, false, EPI), a, slabs, tickets);
  else hipLaunchKernelGGL((gemm16_kernel<WT, KT, PRO, EPI, NW, PT, false, false>), (KB > 1 && a.kfast) ? dim3(KB, gx) : dim3(gx, KB), dim3(64 * NW), lds, st, G16_HOT_ARGS(a, M, KB, false, false, EPI), a, slabs, tickets);
  return (int)hipGetLastError();
}

template <typename WT, typename KT, int PRO, int EPI, int PT>
static int launch_nw(hipStream_t st, int M, int nw, int KB, const GemvArgs& a, float* slabs, size_t sf, int* tk, int nt) {
  if (nw == 16) return launch_g16<WT, KT, PRO, EPI, 16, PT>(st, M, KB, a, slabs, sf, tk, nt);
  if (nw == 8) return launch_g16<WT, KT, PRO, EPI, 8, PT>(st, M, KB, a, slabs, sf, tk, nt);
  return launch_g16<WT, KT, PRO, EPI, 4, PT>(st, M, KB, a, slabs, sf, tk, nt);
}

template <typename WT>
static int launch_gemm16_t(hipStream_t st, int kvdtype, int M, int pro, int epi, const GemvArgs& a, float* slabs,
                           size_t slab_floats, int* tickets, int n_tickets) {
  if (pro != PRO_PLAIN && pro != PRO_NORM) return -2;
  if (a.K % 512 != 0 || a.ldx % 4 != 0) return -2;
  if ((epi == EPI_RESID || epi == EPI_SWIGLU) && (a.N % 16 != 0 || a.ldo % 4 != 0)) return -2;   // quad epilogue: 16-byte rows
  // waves per workgroup x workgroup-level K splits: every wave owns exactly one 128-wide chunk
  const int nchunks = a.K / 128;
  int nw = nchunks >= 16 ? 16 : (nchunks >= 8 ? 8 : 4), KB = 1;
  if (nchunks > 16) {  // K = 8192: 8 waves x 8 workgroup-level splits (measured best of {4x16, 8x8, 16x4})
    if (pro == PRO_NORM) return -2;
    nw = 8;
    KB = nchunks / 8;
    if (KB > 16 || nchunks % 8) return -2;
  }
  // (a K split of a normed launch needs the planes form: there the RMS scale multiplies the finished sum in the epilogue)
  if (a.g16_nw > 0 && a.g16_kb > 0 && a.g16_nw * a.g16_kb == nchunks && (pro != PRO_NORM || a.g16_kb == 1 || a.xplanes)) {
    nw = a.g16_nw;
    KB = a.g16_kb;
  }
  if (nchunks % nw) return -2;
  if (epi == EPI_QKV) {
    if (pro != PRO_NORM || (a.hd != 64 && a.hd != 128) || (KB != 1 && !a.xplanes)) return -2;
    if (kvdtype == 1) return launch_nw<WT, bf16_t, PRO_NORM, EPI_QKV, 2>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);
    return launch_nw<WT, float, PRO_NORM, EPI_QKV, 2>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);
  }
  // 64-row panels (x slice reused by 4 tiles) when that still leaves >= 128 workgroups, else 16-row panels; gate/up
  // on activation planes: 32-row panels (two workgroups per CU, one's tail under the other's stream: B=16 5.20 -> 5.04 ms);
  // round 3: the K-split down_proj of the decoder (64 tiles x 8 splits) on planes takes 32-row panels too -- 256 workgroups,
  // half the plane traffic through the CUs' L1s (swept nw x kb x pt in {8x8, 16x4, 4x16} x {1, 2, 4}: 5.05 ... 5.43 ms per
  // step, 8 x 8 x 2 best, 5.15 -> 5.05; profiles/r03_b16_g16_down_sweep.txt)
  const int ntiles = (a.N + 15) / 16;
  const bool big = a.g16_pt ? a.g16_pt == 4 : (KB > 1 ? ntiles >= 128 : ntiles / 4 >= 128);
#define G16(P, E)                                                                                              \
  if (pro == P && epi == E) {                                                                                  \
    if (a.g16_pt == 2 || (!a.g16_pt && a.xplanes && (E == EPI_SWIGLU || (KB > 1 && !big))))                         \
      return launch_nw<WT, float, P, E, 2>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);             \
    if (big) return launch_nw<WT, float, P, E, 4>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);       \
    return launch_nw<WT, float, P, E, 1>(st, M, nw, KB, a, slabs, slab_floats, tickets, n_tickets);                \
  }
  G16(PRO_PLAIN, EPI_STORE) G16(PRO_NORM, EPI_STORE) G16(PRO_PLAIN, EPI_RESID) G16(PRO_NORM, EPI_SWIGLU)
#undef G16
  return -2;
}

int launch_gemm16(hipStream_t st, int wdtype, int kvdtype, int M, int pro, int epi, const GemvArgs& a, float* slabs,
                  size_t slab_floats, int* tickets, int n_tickets) {
  if ((wdtype != 1 && wdtype != 2) || M < 1 || M > 16 || a.configure_only) return -2;
  if (wdtype == 2) return launch_gemm16_t<fp8_t>(st, kvdtype, M, pro, epi, a, slabs, slab_floats, tickets, n_tickets);
  return launch_gemm16_t<bf16_t>(st, kvdtype, M, pro, epi, a, slabs, slab_floats, tickets, n_tickets);
}
