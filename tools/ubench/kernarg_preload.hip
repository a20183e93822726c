// Does kernarg preloading (SGPRs initialised by the dispatcher instead of an s_load at wave start) shorten a dependent launch
// of the frame-step chain?  A hipGraph of 200 dependent GEMV-like launches (each workgroup: 16 KB of "weights" from a pointer in the
// kernel arguments + the previous launch's 4 KB output vector; 256 / 1024 workgroups), arguments passed
//   S: as one struct by value (what libcsm_hip.so does: the first instruction of every wave is an s_load of the kernarg segment),
//   F: the hot fields as leading scalar parameters (preloaded when built with -mllvm -amdgpu-kernarg-preload-count=N).
// Build twice:  hipcc --offload-arch=gfx950 -O3 kernarg_preload.hip -o kp_off
//               hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=14 kernarg_preload.hip -o kp_on
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args { const float* W; const float* x; float* out; int K; int rows_per_wg; int pad[40]; const float* other; };

template <bool WT = false>
__device__ __forceinline__ void body(const float* W, const float* x, float* out, int K, int rpw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every wave: rpw rows of K floats; 16-byte loads, all issued before use
  const int row0 = (blockIdx.x * 4 + wave) * rpw;
  float acc = 0.f;
  for (int r = 0; r < rpw; ++r) {
    const float4* wp = reinterpret_cast<const float4*>(W + (size_t)(row0 + r) * K);
    const float4* xp = reinterpret_cast<const float4*>(x);
    float s = 0.f;
    for (int k = lane; k < K / 4; k += 64) { const float4 w = wp[k], v = xp[k]; s += w.x * v.x + w.y * v.y + w.z * v.z + w.w * v.w; }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
      float* p = out + (row0 + r) % K;
      const float v = s * 1e-3f;
      if (WT) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
      else *p = v;
    }
  }
}
__global__ __launch_bounds__(256) void k_struct(Args a) { body(a.W, a.x, a.out, a.K, a.rows_per_wg); }
__global__ __launch_bounds__(256) void k_flat_wt(const float* W, const float* x, float* out, int K, int rpw, Args rest) {
  body<true>(W, x, out, K, rpw);
  if (rest.pad[7] == 12345) out[0] = rest.other[0];
}
__global__ __launch_bounds__(256) void k_flat(const float* W, const float* x, float* out, int K, int rpw, Args rest) {
  body(W, x, out, K, rpw);
  if (rest.pad[7] == 12345) out[0] = rest.other[0];     // keeps the struct alive without touching it on the hot path
}

int main(int argc, char** argv) {
  const int K = 1024; const int NL = argc > 1 ? atoi(argv[1]) : 200; const int REPL = argc > 2 ? atoi(argv[2]) : 10;
  for (int grid : {256, 1024}) {
    const int rpw = 1;
    const size_t wfloats = (size_t)grid * 4 * rpw * K;
    float *W, *xa, *xb;
    CK(hipMalloc(&W, wfloats * sizeof(float) * 8)); CK(hipMalloc(&xa, K * 4)); CK(hipMalloc(&xb, K * 4));
    CK(hipMemset(W, 0, wfloats * sizeof(float) * 8)); CK(hipMemset(xa, 0, K * 4)); CK(hipMemset(xb, 0, K * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int mode = 0; mode < 3; ++mode) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < NL; ++i) {
        Args a{}; a.W = W + (size_t)(i % 8) * wfloats; a.x = (i & 1) ? xb : xa; a.out = (i & 1) ? xa : xb; a.K = K; a.rows_per_wg = rpw; a.other = xa;
        if (mode == 0) hipLaunchKernelGGL(k_struct, dim3(grid), dim3(256), 0, st, a);
        else if (mode == 1) hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, st, a.W, a.x, a.out, a.K, a.rows_per_wg, a);
        else hipLaunchKernelGGL(k_flat_wt, dim3(grid), dim3(256), 0, st, a.W, a.x, a.out, a.K, a.rows_per_wg, a);
      }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int w = 0; w < REPL; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("grid %4d  %s : %.3f us per launch (%d dependent launches per graph, several replays, %zu KB of weights per launch)\n", grid,
             mode == 0 ? "struct by value      " : (mode == 1 ? "flat leading scalars " : "flat + sc0 sc1 stores"), best * 1e3f / (REPL * NL), NL, wfloats * 4 / 1024);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    CK(hipFree(W)); CK(hipFree(xa)); CK(hipFree(xb));
  }
  return 0;
}
