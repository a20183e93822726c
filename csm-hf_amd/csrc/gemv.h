// Weight-streaming skinny GEMM ("GEMV") for decode: y[M,N] = f(x)[M,K] @ W[N,K]^T, M <= 4 per launch.
//
// Roofline: HBM.  Algorithmic bytes = N*K*sizeof(WT) (each weight read exactly once per launch); x is
// M*K*4 bytes re-read per block from L2.  One wave owns a PAIR of output rows per task so that the
// fused epilogues that need two outputs (RoPE pair i / i+hd/2, SwiGLU gate/up) stay wave-local.
// 64 lanes x 16 B = 1 KiB per load instruction, 2*U loads in flight per wave; no LDS round trip for
// the weights (cdna_hip_programming.md "GEMV / M<=16 decode weights" row).
//
// Replaces (reference call sites): q/k/v/o_proj, gate/up/down_proj inside transformers.LlamaModel as
// called from modeling_csm.py:345-354,545-552,568-576; codebook0_head (:361), projection (:542),
// audio_head matmul (:557); RMSNorm (transformers modeling_llama.py:62-67) as prologue; RoPE
// (modeling_llama.py:130-160) + DynamicCache append (cache_utils.py:144-145) as the QKV epilogue.
#pragma once
#include "common.h"

enum { PRO_PLAIN = 0, PRO_NORM = 1 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_QKV = 3 };

struct GemvArgs {
  const void* W;
  int N, K;
  const float* x;  // [M][ldx]
  int ldx;
  const float* ln;  // PRO_NORM weight [K]
  float eps;
  float* out;  // EPI_STORE/RESID: [M][ldo] indexed by output row; EPI_SWIGLU: [M][ldo] indexed by pair
  int ldo;
  // EPI_QKV
  int n_q, n_kv, hd;
  float qscale;
  const float* cos_tab;  // [pos][hd/2]
  const float* sin_tab;
  const int* pos_ptr;    // device scalar (backbone decode: current length), or
  int pos_const;         // constant position (decoder pass), used when pos_ptr == nullptr
  const int* row_pos;    // optional per-row positions (overrides both)
  const int* row_seq;    // optional row -> sequence slot (default: seq_base + row index)
  int seq_base;
  float* qbuf;           // [M][n_q*hd]
  void* kcache;          // [B][n_kv][hd/4][lmax][4]
  void* vcache;          // [B][n_kv][lmax][hd]
  int lmax;
  // end-of-kernel counter bumps (thread 0 of block 0), used by the head kernel to advance the
  // device-side frame index / backbone length inside a graph
  int* bump_a;
  int* bump_b;
  int nt;  // non-temporal weight loads
  int configure_only;  // host-side: only set the kernel's dynamic-LDS attribute, do not launch
};

#ifndef CSM_ARGS_ONLY
template <typename KT>
__device__ __forceinline__ size_t k_index(int b, int j, int d, int t, int n_kv, int hd, int lmax) {
  return ((((size_t)b * n_kv + j) * (hd >> 2) + (d >> 2)) * lmax + t) * 4 + (d & 3);
}
__device__ __forceinline__ size_t v_index(int b, int j, int t, int d, int n_kv, int hd, int lmax) {
  return (((size_t)b * n_kv + j) * lmax + t) * hd + d;
}

template <typename WT, typename KT, int M, int PRO, int EPI>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [M][K] then red[M][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  float* red = xs + (size_t)M * K;

  // ---- prologue: stage (optionally RMS-normalised) x into LDS --------------------------------------
  {
    float ss[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      ss[m] = 0.f;
      const float* xr = a.x + (size_t)m * a.ldx;
      for (int k = tid * 4; k < K; k += 1024) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
        *reinterpret_cast<f32x4*>(xs + (size_t)m * K + k) = v;
        if (PRO == PRO_NORM) ss[m] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      }
    }
    if (PRO == PRO_NORM) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float s = wave_sum(ss[m]);
        if (lane == 0) red[m * 4 + wave] = s;
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float s = red[m * 4 + 0] + red[m * 4 + 1] + red[m * 4 + 2] + red[m * 4 + 3];
        float sc = rsqrtf(s / (float)K + a.eps);
        for (int k = tid * 4; k < K; k += 1024) {
          f32x4 v = *reinterpret_cast<f32x4*>(xs + (size_t)m * K + k);
          f32x4 w = *reinterpret_cast<const f32x4*>(a.ln + k);
          v[0] = (v[0] * sc) * w[0];
          v[1] = (v[1] * sc) * w[1];
          v[2] = (v[2] * sc) * w[2];
          v[3] = (v[3] * sc) * w[3];
          *reinterpret_cast<f32x4*>(xs + (size_t)m * K + k) = v;
        }
      }
    }
    __syncthreads();
  }

  const WT* W = reinterpret_cast<const WT*>(a.W);
  const int nch = K >> 3;
  const int half = a.hd >> 1;
  const int ntask = (EPI == EPI_QKV) ? (a.N >> 1) : ((a.N + 1) >> 1);
  constexpr int U = 4;

  for (int task = blockIdx.x * 4 + wave; task < ntask; task += gridDim.x * 4) {
    int r0, r1, head = 0, hi = 0;
    if (EPI == EPI_QKV) {
      head = task / half;
      hi = task - head * half;
      if (head < a.n_q + a.n_kv) {
        r0 = head * a.hd + hi;
        r1 = r0 + half;
      } else {
        r0 = head * a.hd + 2 * hi;
        r1 = r0 + 1;
      }
    } else {
      r0 = 2 * task;
      r1 = r0 + 1;
    }
    const bool has1 = r1 < a.N;
    const WT* w0p = W + (size_t)r0 * K;
    const WT* w1p = W + (size_t)(has1 ? r1 : r0) * K;

    float acc0[M], acc1[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc0[m] = acc1[m] = 0.f;

    for (int cb = 0; cb < nch; cb += 64 * U) {
      W8<WT> w0[U], w1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = cb + u * 64 + lane;
        if (c < nch) {
          if (a.nt) {
            w0[u].load_nt(w0p + (size_t)c * 8);
            w1[u].load_nt(w1p + (size_t)c * 8);
          } else {
            w0[u].load(w0p + (size_t)c * 8);
            w1[u].load(w1p + (size_t)c * 8);
          }
        } else {
          w0[u].zero();
          w1[u].zero();
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int c = cb + u * 64 + lane;
        c = c < nch ? c : nch - 1;
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + (size_t)m * K + c * 8);
          const f32x4 xb = *reinterpret_cast<const f32x4*>(xs + (size_t)m * K + c * 8 + 4);
          float s0 = acc0[m], s1 = acc1[m];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s0 = fmaf(w0[u].get(i), xa[i], s0);
            s1 = fmaf(w1[u].get(i), xa[i], s1);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s0 = fmaf(w0[u].get(4 + i), xb[i], s0);
            s1 = fmaf(w1[u].get(4 + i), xb[i], s1);
          }
          acc0[m] = s0;
          acc1[m] = s1;
        }
      }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      acc0[m] = wave_sum(acc0[m]);
      acc1[m] = wave_sum(acc1[m]);
    }
    if (lane == 0) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float v0 = acc0[m], v1 = acc1[m];
        if (EPI == EPI_STORE) {
          a.out[(size_t)m * a.ldo + r0] = v0;
          if (has1) a.out[(size_t)m * a.ldo + r1] = v1;
        } else if (EPI == EPI_RESID) {
          a.out[(size_t)m * a.ldo + r0] += v0;
          if (has1) a.out[(size_t)m * a.ldo + r1] += v1;
        } else if (EPI == EPI_SWIGLU) {
          a.out[(size_t)m * a.ldo + task] = (v0 / (1.f + __expf(-v0))) * v1;
        } else {  // EPI_QKV
          const int b = a.row_seq ? a.row_seq[m] : a.seq_base + m;
          const int pos = a.row_pos ? a.row_pos[m] : (a.pos_ptr ? *a.pos_ptr : a.pos_const);
          KT* kc = reinterpret_cast<KT*>(a.kcache);
          KT* vc = reinterpret_cast<KT*>(a.vcache);
          if (head < a.n_q + a.n_kv) {
            const float c = a.cos_tab[(size_t)pos * half + hi];
            const float s = a.sin_tab[(size_t)pos * half + hi];
            const float o0 = v0 * c - v1 * s;
            const float o1 = v1 * c + v0 * s;
            if (head < a.n_q) {
              float* q = a.qbuf + (size_t)m * a.n_q * a.hd + head * a.hd;
              q[hi] = o0 * a.qscale;
              q[hi + half] = o1 * a.qscale;
            } else {
              const int j = head - a.n_q;
              store_kv(kc + k_index<KT>(b, j, hi, pos, a.n_kv, a.hd, a.lmax), o0);
              store_kv(kc + k_index<KT>(b, j, hi + half, pos, a.n_kv, a.hd, a.lmax), o1);
            }
          } else {
            const int j = head - a.n_q - a.n_kv;
            store_kv(vc + v_index(b, j, pos, 2 * hi, a.n_kv, a.hd, a.lmax), v0);
            store_kv(vc + v_index(b, j, pos, 2 * hi + 1, a.n_kv, a.hd, a.lmax), v1);
          }
        }
      }
    }
  }
  if (a.bump_a && blockIdx.x == 0 && tid == 0) {
    *a.bump_a += 1;
    if (a.bump_b) *a.bump_b += 1;
  }
}

#endif  // CSM_ARGS_ONLY
// host-side launcher (defined in gemv.hip)
int launch_gemv(hipStream_t st, int wdtype, int kvdtype, int M, int pro, int epi, const GemvArgs& a);
