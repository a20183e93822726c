// Prefill GEMM on the matrix cores: C[R,N] (+)= A[R,K] @ W[N,K]^T with fp32 activations.
//
// v1 uses the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TFLOP/s peak) so
// that prefill hidden states carry fp32-class error for BOTH weight dtypes (bf16 weights are widened
// while staging to LDS).  Roofline: MFMA (f32 rate).  Algorithmic flops = 2*R*N*K.
// Tile 128x128x32 per 256-thread workgroup, 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32;
// LDS rows padded to 33 floats => conflict-free ds_read_b32 operand fetches.
//
// Replaces the q/k/v/o/gate/up/down nn.Linear calls of transformers.LlamaModel when q_len > 1
// (reference call site modeling_csm.py:345-354).
#pragma once
#include "common.h"

enum { GEPI_STORE = 0, GEPI_RESID = 1, GEPI_SWIGLU = 2 };

struct GemmArgs {
  const float* A;  // [R][lda]
  int lda;
  const void* W;  // [N][K]
  const float* wscale;  // per-row scale (fp8 weights), nullable
  int R, N, K;
  float* C;  // STORE/RESID: [R][ldc]; SWIGLU: [R][ldc] with N/2 columns
  int ldc;
};

#ifndef CSM_ARGS_ONLY
template <typename WT, int EPI>
__global__ __launch_bounds__(256) void gemm_f32mfma_kernel(GemmArgs a) {
  constexpr int BM = 128, BN = 128, BK = 32, LD = BK + 1;
  __shared__ float As[BM * LD];
  __shared__ float Ws[BN * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int nbn = a.N / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
  const int r0 = bm * BM, n0 = bn * BN;
  const WT* W = reinterpret_cast<const WT*>(a.W);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);

  for (int k0 = 0; k0 < a.K; k0 += BK) {
    // ---- stage A tile (fp32) -------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 3, c4 = idx & 7;
      f32x4 v = (f32x4)(0.f);
      if (r0 + row < a.R) v = *reinterpret_cast<const f32x4*>(a.A + (size_t)(r0 + row) * a.lda + k0 + c4 * 4);
      float* d = As + row * LD + c4 * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    // ---- stage W tile (widen to fp32) ----------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 2, c8 = idx & 3;
      W8<WT> w;
      w.load(W + (size_t)(n0 + row) * a.K + k0 + c8 * 8);
      float* d = Ws + row * LD + c8 * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = w.get(e);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int kc = kk + (lane >> 5);
      const float a0 = As[(wr * 64 + (lane & 31)) * LD + kc];
      const float a1 = As[(wr * 64 + 32 + (lane & 31)) * LD + kc];
      const float b0 = Ws[(wc * 64 + (lane & 31)) * LD + kc];
      const float b1 = Ws[(wc * 64 + 32 + (lane & 31)) * LD + kc];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) --------------
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = r0 + wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int n = n0 + wc * 64 + ni * 32 + (lane & 31);
        const float v = acc[mi][ni][reg] * (a.wscale ? a.wscale[n] : 1.f);
        if (EPI == GEPI_SWIGLU) {
          const float o = __shfl_xor(v, 1, 64);  // even lane: gate (own), up (partner)
          if (!(lane & 1) && r < a.R) a.C[(size_t)r * a.ldc + (n >> 1)] = (v / (1.f + __expf(-v))) * o;
        } else if (r < a.R) {
          if (EPI == GEPI_RESID) a.C[(size_t)r * a.ldc + n] += v;
          else a.C[(size_t)r * a.ldc + n] = v;
        }
      }
}

#endif  // CSM_ARGS_ONLY
int launch_gemm(hipStream_t st, int wdtype, int epi, const GemmArgs& a);
