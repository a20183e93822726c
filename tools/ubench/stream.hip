// What limits a 16.8 MB "down_proj"-shaped stream?  Variants of one kernel, dependent chain in a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
typedef __attribute__((ext_vector_type(4))) float f4;
__device__ __forceinline__ float wsum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

// each wave: NL 16-byte loads per lane, all issued up front.  XL: also load a x slice of NL*8 floats per lane.
// RED: 0 none, 1 wave_sum, 2 wave_sum + LDS + barrier across the 4 waves
template <int NL, int XL, int RED>
__global__ __launch_bounds__(256) void k(const u4* W, const float* x, float* out) {
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u4* p = W + ((size_t)blockIdx.x * 4 + wave) * NL * 64 + lane;
  u4 w[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) w[i] = __builtin_nontemporal_load(p + i * 64);
  float acc = 0.f;
  if (XL) {
    f4 xa[NL], xb[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { xa[i] = *(const f4*)(x + wave * NL * 512 + i * 512 + lane * 8); xb[i] = *(const f4*)(x + wave * NL * 512 + i * 512 + lane * 8 + 4); }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      acc += __uint_as_float(w[i].x << 16) * xa[i].x + __uint_as_float(w[i].x & 0xffff0000u) * xa[i].y + __uint_as_float(w[i].y << 16) * xa[i].z + __uint_as_float(w[i].y & 0xffff0000u) * xa[i].w;
      acc += __uint_as_float(w[i].z << 16) * xb[i].x + __uint_as_float(w[i].z & 0xffff0000u) * xb[i].y + __uint_as_float(w[i].w << 16) * xb[i].z + __uint_as_float(w[i].w & 0xffff0000u) * xb[i].w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NL; ++i) acc += __uint_as_float(w[i].x) + __uint_as_float(w[i].y) + __uint_as_float(w[i].z) + __uint_as_float(w[i].w);
  }
  if (RED >= 1) acc = wsum(acc);
  if (RED == 2) {
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    acc = part[0] + part[1] + part[2] + part[3];
  }
  if (threadIdx.x == 0 || (RED == 0 && acc == 1.2345f)) out[blockIdx.x] = acc;
}

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
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * n);
}

int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const size_t pool = (size_t)1 << 30;
  u4* W; hipMalloc(&W, pool); hipMemset(W, 0x3c, pool);
  float *x, *out; hipMalloc(&x, 1 << 20); hipMalloc(&out, 1 << 20); hipMemset(x, 0, 1 << 20);
  for (size_t mb : {16, 32}) {
    const size_t bytes = mb << 20, nslots = pool / bytes;
#define RUN(NL, XL, RED)                                                                                   \
    {                                                                                                      \
      const int grid = (int)(bytes / (4 * NL * 1024));                                                     \
      float us = time_graph(st, 200, 10, [&](int i) {                                                      \
        hipLaunchKernelGGL((k<NL, XL, RED>), dim3(grid), dim3(256), 0, st, W + (size_t)(i % nslots) * (bytes / 16), x, out); }); \
      printf("%2zu MiB  loads/lane %2d  grid %5d  x-slice %d  reduce %d : %.2f us  %.2f TB/s\n", mb, NL, grid, XL, RED, us, bytes / us / 1e6); \
    }
    RUN(2, 0, 0) RUN(4, 0, 0) RUN(8, 0, 0) RUN(16, 0, 0)
    RUN(4, 0, 1) RUN(8, 0, 1) RUN(8, 0, 2) RUN(4, 1, 2) RUN(8, 1, 2) RUN(16, 1, 2)
  }
  return 0;
}
