// Stand-alone timing of gemm128_kernel (csm-hf_amd/csrc/gemm128.h) on the decoder FFN shapes of a 128-row step, with TIMING-ONLY
// knock-outs selected at compile time (-DCSM_G128_VARIANT=bits; the product library is built without it):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm --amdgpu-mfma-vgpr-form [-DCSM_G128_VARIANT=n] tools/ubench/g128_bench.hip -o g128_bench_n
// usage: g128_bench [gateup|down] [reps] [rows = 128]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../csm-hf_amd/csrc/gemm128.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const bool down = argc > 1 && !strcmp(argv[1], "down");
  const int reps = argc > 2 ? atoi(argv[2]) : 200;
  const int M = argc > 3 ? atoi(argv[3]) : 128, K = down ? 8192 : 1024, N = down ? 1024 : 16384;
  const int KB = K / 1024;
  size_t wbytes = (size_t)N * K * 2, pbytes = (size_t)(M / 16) * 3 * K * 16 * 2;
  bf16_t *W, *P, *OP; float *xss, *out, *oln, *slabs; int* tickets; uint32_t* dbg;
  CK(hipMalloc(&W, wbytes)); CK(hipMalloc(&P, pbytes)); CK(hipMalloc(&OP, (size_t)(M / 16) * 3 * (down ? N : N / 2) * 16 * 2));
  CK(hipMalloc(&xss, (size_t)M * 128 * 4)); CK(hipMalloc(&out, (size_t)M * N * 4)); CK(hipMalloc(&oln, N * 4));
  CK(hipMalloc(&slabs, (size_t)64 << 20)); CK(hipMalloc(&tickets, 4096 * 4)); CK(hipMalloc(&dbg, (size_t)4096 * 64 * 4));
  std::vector<uint16_t> h(wbytes / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (uint16_t)((i * 2654435761u) >> 22 & 0xff);   // small positive bf16 values
  CK(hipMemcpy(W, h.data(), wbytes, hipMemcpyHostToDevice));
  h.resize(pbytes / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3b00 + (uint16_t)((i * 40503u) >> 8 & 0xff);
  CK(hipMemcpy(P, h.data(), pbytes, hipMemcpyHostToDevice));
  CK(hipMemset(xss, 0, (size_t)M * 128 * 4)); CK(hipMemset(out, 0, (size_t)M * N * 4)); CK(hipMemset(oln, 0, N * 4));
  CK(hipMemset(tickets, 0, 4096 * 4)); CK(hipMemset(dbg, 0, (size_t)4096 * 64 * 4));
  G128Args a{};
  a.Wt = W; a.xplanes = P; a.K = K; a.N = N; a.ldo = down ? N : N / 2; a.out = out; a.oplanes = OP; a.oln = oln;
  a.xss = xss; a.xss_n = 64; a.xss_ld = 128; a.eps = 1e-5f; a.dbg = dbg;
  float* oss; CK(hipMalloc(&oss, (size_t)M * 128 * 4)); a.oss = down ? oss : nullptr; a.oss_ld = 128;
  a.M = M; a.KB = KB; a.slabs = slabs; a.tickets = tickets;
  const int PT = down ? 1 : 2;
  const int gx = (N / 16) / (4 * PT), Z = (M + 63) / 64;
  const size_t lds = (size_t)3 * 4 * 3 * 4 * 1024 + 64 * 4;
#ifndef CSM_G128_H
#define CSM_G128_H 1
#endif
  auto fu = gemm128_kernel<bf16_t, PRO_NORM, EPI_SWIGLU, 2, false, CSM_G128_H>;
  auto fd = gemm128_kernel<bf16_t, PRO_PLAIN, EPI_RESID, 1, false, CSM_G128_H>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fu), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (down) hipLaunchKernelGGL(fd, dim3(KB, gx, Z), dim3(256 * CSM_G128_H), lds, st, a);
    else hipLaunchKernelGGL(fu, dim3(gx, KB, Z), dim3(256 * CSM_G128_H), lds, st, a);
  };
  for (int i = 0; i < 20; ++i) launch();
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
#ifndef CSM_G128_VARIANT
#define CSM_G128_VARIANT 0
#endif
  printf("%s H %d variant %d: grid %d x %d x %d  %.2f us per launch (back to back, %d launches)\n", down ? "down  " : "gateup", CSM_G128_H, CSM_G128_VARIANT, gx, KB, Z, ms * 1e3 / reps, reps);
  if (CSM_G128_VARIANT & 64) {   // chunk time stamps of the last launch: per workgroup, wave 0: entry, chunk 0..7 (after the barrier), end of loop, exit -- 10 ns ticks
    std::vector<uint32_t> d((size_t)gx * KB * Z * 64);
    CK(hipMemcpy(d.data(), dbg, d.size() * 4, hipMemcpyDeviceToHost));
    uint32_t t0 = 0xffffffffu;
    for (size_t w = 0; w < (size_t)gx * KB * Z; ++w) if (d[w * 64 + 10] < t0) t0 = d[w * 64 + 10];
    for (size_t w = 0; w < (size_t)gx * KB * Z; w += (size_t)gx * KB * Z / 16) {
      printf("wg %4zu: entry %5u, epilogue inputs requested %5u, prologue issued %5u |", w, d[w * 64 + 10] - t0, d[w * 64 + 15] - t0, d[w * 64 + 14] - t0);
      for (int c = 0; c < 9; ++c) printf(" %5u", d[w * 64 + c] - t0);
      printf(" | epilogue at %5u exit %5u | shader clocks entry -> exit %u = %.0f MHz\n", d[w * 64 + 11] - t0, d[w * 64 + 9] - t0, d[w * 64 + 13] - d[w * 64 + 12],
             (double)(d[w * 64 + 13] - d[w * 64 + 12]) / ((d[w * 64 + 9] - d[w * 64 + 10]) * 0.01));
    }
  }
  return 0;
}
