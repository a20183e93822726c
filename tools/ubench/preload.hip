// Does kernarg preloading (SGPR preload of the first kernel arguments) shorten a dependent launch chain?
#include <hip/hip_runtime.h>
#include <cstdio>
struct BigArgs { const float* in; float* out; int n; int pad[60]; };
__global__ void k_struct(BigArgs a) { float v = a.in[threadIdx.x]; if (threadIdx.x == 0) a.out[blockIdx.x] = v + 1.f; }
__global__ void k_flat(const float* in, float* out, int n) { float v = in[threadIdx.x]; if (threadIdx.x == 0) out[blockIdx.x] = v + 1.f; }
__global__ void k_mixed(const float* in, float* out, BigArgs rest) { float v = in[threadIdx.x]; if (threadIdx.x == 0) out[blockIdx.x] = v + (float)rest.n; }
template <typename F>
static float time_graph(hipStream_t st, int n, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) launch(i);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st); hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / (reps * n);
}
int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  float *a, *b; hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20); hipMemset(a, 0, 1 << 20); hipMemset(b, 0, 1 << 20);
  for (int grid : {256, 1024}) {
    BigArgs x{a, b, 0, {}}, y{b, a, 0, {}};
    printf("grid %4d: struct %.3f", grid, time_graph(st, 600, 20, [&](int i) { hipLaunchKernelGGL(k_struct, dim3(grid), dim3(256), 0, st, (i & 1) ? x : y); }));
    printf("  flat %.3f", time_graph(st, 600, 20, [&](int i) { hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, st, (i & 1) ? a : b, (i & 1) ? b : a, 0); }));
    printf("  mixed %.3f us/kernel\n", time_graph(st, 600, 20, [&](int i) { hipLaunchKernelGGL(k_mixed, dim3(grid), dim3(256), 0, st, (i & 1) ? a : b, (i & 1) ? b : a, x); }));
  }
  return 0;
}
