// Micro-benchmark (MI355X): where does a k-step of the square-tile prefill GEMM go at one or two workgroups per CU?
// The 128 x 128 x 64 one-plane body of csrc/gemm.h (gemm_bf16x3_kernel<..., 128, 64, true, 1>: bf16 activation plane and
// bf16 weights staged through LDS with a one-step register prefetch, 4 waves as 2 x 2, 32x32x16 MFMA) re-stated with
// KNOCK-OUT switches instead of timestamps (no perturbation): bit 0 = no global loads inside the k loop, bit 1 = no LDS
// staging and no barriers, bit 2 = no LDS operand reads (operands stay in registers), bit 3 = no MFMAs.
// Shapes: the gate/up GEMM of a 512-frame prefill (R 512, N 16384, K 2048: 512 workgroups) and QKV (N 3072: 96).
// build: hipcc --offload-arch=gfx950 -O3 gemmphase.hip -o gemmphase ; run: ./gemmphase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Args { const bf16_t* A; const bf16_t* W; float* C; int R, N, K; };

template <int MODE>
__global__ __launch_bounds__(256) void k_gemm(Args a) {
  constexpr int BM = 128, BN = 128, BK = 64, LDK = BK + 8, W8N = BK / 8, NP = BM * W8N / 256;
  constexpr bool NOLOAD = MODE & 1, NOSTAGE = MODE & 2, NOLDSRD = MODE & 4, NOMFMA = MODE & 8;
  __shared__ __attribute__((aligned(16))) bf16_t Ap[BM * LDK];
  __shared__ __attribute__((aligned(16))) bf16_t Ws[BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int nbn = a.N / BN, bm = blockIdx.x / nbn, bn = blockIdx.x % nbn, r0 = bm * BM, n0 = bn * BN;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);
  u32x4 pp[NP], pw[NP];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + i * 256, row = idx / W8N, c8 = idx % W8N;
      pp[i] = *reinterpret_cast<const u32x4*>(a.A + (size_t)(r0 + row) * a.K + k0 + c8 * 8);
      pw[i] = *reinterpret_cast<const u32x4*>(a.W + (size_t)(n0 + row) * a.K + k0 + c8 * 8);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + i * 256, row = idx / W8N, c8 = idx % W8N;
      *reinterpret_cast<u32x4*>(&Ap[row * LDK + c8 * 8]) = pp[i];
      *reinterpret_cast<u32x4*>(&Ws[row * LDK + c8 * 8]) = pw[i];
    }
  };
  fetch(0);
  if (NOSTAGE) { stage(); lds_barrier(); }
  bf16x8 af[2], bf[2];
  if (NOLDSRD) {
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(&af[i]) = pp[i];
      *reinterpret_cast<u32x4*>(&bf[i]) = pw[i];
    }
  }
  for (int k0 = 0; k0 < a.K; k0 += BK) {
    if (!NOSTAGE) { stage(); lds_barrier(); }
    if (!NOLOAD && k0 + BK < a.K) fetch(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      const int ko = kk + (lane >> 5) * 8;
      if (!NOLDSRD) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) *reinterpret_cast<u32x4*>(&af[mi]) = *reinterpret_cast<const u32x4*>(&Ap[(wr * 64 + mi * 32 + (lane & 31)) * LDK + ko]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) *reinterpret_cast<u32x4*>(&bf[ni]) = *reinterpret_cast<const u32x4*>(&Ws[(wc * 64 + ni * 32 + (lane & 31)) * LDK + ko]);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if (!NOMFMA) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
          else acc[mi][ni][kk >> 4] += (float)(((const unsigned*)&af[mi])[0] ^ ((const unsigned*)&bf[ni])[0]);   // keeps the operands alive
        }
    }
    if (!NOSTAGE) lds_barrier();
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = r0 + wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int n = n0 + wc * 64 + ni * 32 + (lane & 31);
        a.C[(size_t)r * a.N + n] = acc[mi][ni][reg];
      }
}

// The same body with the LDS tiles double-buffered: ONE barrier per k-step, and the next step's tile is written (from the
// registers its global loads landed in) BEFORE this step's operand reads and MFMAs, so the ds_writes run under the matrix
// work of the same wave instead of between two barriers.  Same accumulation order: bitwise the same result.
__global__ __launch_bounds__(256) void k_gemm_db(Args a) {
  constexpr int BM = 128, BN = 128, BK = 64, LDK = BK + 8, W8N = BK / 8, NP = BM * W8N / 256;
  __shared__ __attribute__((aligned(16))) bf16_t Ap[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) bf16_t Ws[2][BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int nbn = a.N / BN, bm = blockIdx.x / nbn, bn = blockIdx.x % nbn, r0 = bm * BM, n0 = bn * BN;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16)(0.f);
  u32x4 pp[NP], pw[NP];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + i * 256, row = idx / W8N, c8 = idx % W8N;
      pp[i] = *reinterpret_cast<const u32x4*>(a.A + (size_t)(r0 + row) * a.K + k0 + c8 * 8);
      pw[i] = *reinterpret_cast<const u32x4*>(a.W + (size_t)(n0 + row) * a.K + k0 + c8 * 8);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + i * 256, row = idx / W8N, c8 = idx % W8N;
      *reinterpret_cast<u32x4*>(&Ap[buf][row * LDK + c8 * 8]) = pp[i];
      *reinterpret_cast<u32x4*>(&Ws[buf][row * LDK + c8 * 8]) = pw[i];
    }
  };
  fetch(0);
  stage(0);
  if (BK < a.K) fetch(BK);
  lds_barrier();
  int buf = 0;
  for (int k0 = 0; k0 < a.K; k0 += BK, buf ^= 1) {
    if (k0 + BK < a.K) stage(buf ^ 1);            // every wave passed the barrier after its reads of buf ^ 1
    if (k0 + 2 * BK < a.K) fetch(k0 + 2 * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      const int ko = kk + (lane >> 5) * 8;
      bf16x8 af[2], bf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) *reinterpret_cast<u32x4*>(&af[mi]) = *reinterpret_cast<const u32x4*>(&Ap[buf][(wr * 64 + mi * 32 + (lane & 31)) * LDK + ko]);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) *reinterpret_cast<u32x4*>(&bf[ni]) = *reinterpret_cast<const u32x4*>(&Ws[buf][(wc * 64 + ni * 32 + (lane & 31)) * LDK + ko]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
    lds_barrier();
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = r0 + wr * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int n = n0 + wc * 64 + ni * 32 + (lane & 31);
        a.C[(size_t)r * a.N + n] = acc[mi][ni][reg];
      }
}

static float run_db(const Args& a, int reps, float* dC2, size_t cbytes, bool* same) {
  const int grid = (a.R / 128) * (a.N / 128);
  // bitwise check against the single-buffered body
  hipLaunchKernelGGL(k_gemm<0>, dim3(grid), dim3(256), 0, 0, a);
  Args b = a; b.C = dC2;
  hipLaunchKernelGGL(k_gemm_db, dim3(grid), dim3(256), 0, 0, b);
  CK(hipDeviceSynchronize());
  std::vector<float> h1(cbytes / 4), h2(cbytes / 4);
  CK(hipMemcpy(h1.data(), a.C, cbytes, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), dC2, cbytes, hipMemcpyDeviceToHost));
  *same = memcmp(h1.data(), h2.data(), cbytes) == 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gemm_db, dim3(grid), dim3(256), 0, 0, b);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

template <int MODE>
static float run(const Args& a, int reps) {
  const int grid = (a.R / 128) * (a.N / 128);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_gemm<MODE>, dim3(grid), dim3(256), 0, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gemm<MODE>, dim3(grid), dim3(256), 0, 0, a);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main() {
  const int R = 512, K = 2048;
  for (int N : {16384, 3072}) {
    std::vector<bf16_t> hA((size_t)R * K), hW((size_t)N * K);
    unsigned s = 12345u;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = (bf16_t)(0x3c00u + ((s >> 16) & 0x3ffu) + ((s >> 31) << 15)); }
    for (auto& v : hW) { s = s * 1664525u + 1013904223u; v = (bf16_t)(0x3a00u + ((s >> 16) & 0x3ffu) + ((s >> 31) << 15)); }
    Args a{};
    bf16_t *dA, *dW; float* dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, (size_t)R * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    a.A = dA; a.W = dW; a.C = dC; a.R = R; a.N = N; a.K = K;
    const int reps = 50;
    const double gf = 2.0 * R * N * K * 1e-9;
    printf("R %d N %d K %d: %d workgroups of 128 x 128, %d k-steps of 64, %.1f GFLOP\n", R, N, K, (R / 128) * (N / 128), K / 64, gf);
    auto line = [&](const char* what, float us) { printf("  %-58s %7.1f us  (%6.0f TFLOP/s equivalent, %.2f us per k-step)\n", what, us, gf / us * 1e3, us / (K / 64)); };
    line("full body", run<0>(a, reps));
    line("no global loads in the loop", run<1>(a, reps));
    line("no LDS staging, no barriers (loads still issued)", run<2>(a, reps));
    line("no global loads, no staging, no barriers", run<3>(a, reps));
    line("MFMA only (operands in registers)", run<7>(a, reps));
    line("no MFMA (loads, staging, barriers, LDS reads)", run<8>(a, reps));
    line("no MFMA, no global loads (staging, barriers, LDS reads)", run<9>(a, reps));
    line("LDS reads only (no loads, staging, barriers, MFMA)", run<11>(a, reps));
    float* dC2; bool same = false;
    CK(hipMalloc(&dC2, (size_t)R * N * 4));
    const float t_db = run_db(a, reps, dC2, (size_t)R * N * 4, &same);
    line("double-buffered LDS, one barrier per k-step (full work)", t_db);
    printf("  double-buffered result bitwise equal to the full body: %s\n", same ? "yes" : "NO");
    CK(hipFree(dC2));
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC));
  }
  return 0;
}
