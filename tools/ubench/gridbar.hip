// Micro-benchmark: cost of a software grid barrier inside one persistent kernel on MI355X, with and without a
// cross-workgroup data exchange (producer store -> barrier -> consumer load from another XCD's workgroup).
// Compare with tools/ubench/chain.hip (1.54 us per empty dependent launch in a hipGraph).
// build: hipcc --offload-arch=gfx950 -O3 gridbar.hip -o gridbar ; run: ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// monotonically increasing arrival counter; round r completes when counter >= (r+1)*G
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// mode 0: barrier only.  mode 1: every workgroup stores one value (plain store), and after the barrier loads its
// neighbour's (blockIdx+1: another XCD under round-robin placement) and checks it.  mode 2: like 1 but the
// exchange uses agent-scope relaxed atomics for the data (sc1 write-through / L2-bypassing read), relaxed barrier.
__global__ __launch_bounds__(256) void k_bar(unsigned* ctr, unsigned* buf, int rounds, int mode, unsigned* errs) {
  const unsigned G = gridDim.x;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    unsigned* slot = buf + (size_t)(r & 1) * G;
    if (mode == 1 && threadIdx.x == 0) slot[blockIdx.x] = (unsigned)r + 1u;
    if (mode == 2 && threadIdx.x == 0) __hip_atomic_store(slot + blockIdx.x, (unsigned)r + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mode == 2) {
      __syncthreads();
      if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1) * G) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
    } else {
      grid_barrier(ctr, (unsigned)(r + 1) * G);
    }
    if (mode == 1 && threadIdx.x == 0) bad += slot[(blockIdx.x + 1) % G] != (unsigned)r + 1u;
    if (mode == 2 && threadIdx.x == 0)
      bad += __hip_atomic_load(slot + (blockIdx.x + 1) % G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r + 1u;
  }
  if (threadIdx.x == 0 && bad) atomicAdd(errs, bad);
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned *ctr, *buf, *errs; CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&buf, 1 << 16)); CK(hipMalloc(&errs, 256));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int R = 2000;
  for (int G : {64, 256, 512, 1024}) {
    for (int mode = 0; mode < 3; ++mode) {
      CK(hipMemsetAsync(ctr, 0, 256, st)); CK(hipMemsetAsync(buf, 0, 1 << 16, st)); CK(hipMemsetAsync(errs, 0, 256, st));
      hipLaunchKernelGGL(k_bar, dim3(G), dim3(256), 0, st, ctr, buf, 10, mode, errs);   // warm
      CK(hipMemsetAsync(ctr, 0, 256, st));
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(k_bar, dim3(G), dim3(256), 0, st, ctr, buf, R, mode, errs);
      hipEventRecord(e1, st); CK(hipStreamSynchronize(st));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned h = 0; CK(hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost));
      printf("grid %4d mode %d: %.3f us per barrier round (%u exchange errors)\n", G, mode, ms * 1000.f / R, h);
    }
  }
  return 0;
}
