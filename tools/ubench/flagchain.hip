// Micro-benchmark (MI355X): a chain of dependent small kernels, (a) the ordinary way -- one stream, every launch waits
// for the previous one at the kernel boundary -- against (b) launches alternating over TWO streams that synchronise
// through memory: kernel n+1 is dispatched while kernel n runs, requests its weights, then spins on per-XCD arrival
// counters that the workgroups of kernel n bump after their (write-through) stores.  Question: does overlapping the
// dispatch ramp and the weight round trip of launch n+1 with launch n beat the kernel boundary (1.5-2 us) by more than
// the arrival / poll latency costs?  Every spin is bounded (gives up after ~2 ms and raises an error flag).
// Kernel body ~ a decoder QKV / o_proj launch at B = 1: W bytes of weights (2-4 MB), a 4 KB activation vector written
// by the previous launch, a 4 KB output.
// build: hipcc --offload-arch=gfx950 -O3 flagchain.hip -o flagchain ; run: ./flagchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f4;

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

struct Args {
  const f4* W;          // this launch's weights: grid * 256 * LPT f4
  const float* xin;     // 1024 floats written by the previous launch
  float* xout;          // 1024 floats
  const unsigned* wait; // 8 per-XCD counters of the previous launch (mode 1) or nullptr
  unsigned* done;       // 8 per-XCD counters of this launch or nullptr
  unsigned expect;      // workgroups of the previous launch
  unsigned* err;
};

template <int LPT>
__global__ __launch_bounds__(256) void k_chain(Args a) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  f4 w[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) w[i] = __builtin_nontemporal_load(a.W + ((size_t)blockIdx.x * LPT + i) * 256 + tid);   // in flight during the wait
  if (a.wait) {
    if (tid == 0) {
      const long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        unsigned s = 0;
#pragma unroll
        for (int x = 0; x < 8; ++x) s += __hip_atomic_load(a.wait + x * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s >= a.expect) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000) { atomicAdd(a.err, 1u); break; }   // 2 ms at 100 MHz
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  // the activation vector: agent-scope loads (the producer ran on other XCDs)
  f4 x;
  const float* xp = a.xin + (tid & 255) * 4;
  x[0] = __hip_atomic_load(xp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  x[1] = __hip_atomic_load(xp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  x[2] = __hip_atomic_load(xp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  x[3] = __hip_atomic_load(xp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LPT; ++i) s += w[i][0] * x[0] + w[i][1] * x[1] + w[i][2] * x[2] + w[i][3] * x[3];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid < 4) {
    const float v = (red[0] + red[1]) + (red[2] + red[3]);
    __hip_atomic_store(a.xout + ((blockIdx.x * 4 + tid) & 1023), v * 1e-3f + 0.5f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
  }
  if (a.done) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.done + xcc_id() * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int LPT>
static double run(int mode, int grid, int n, const f4* W, size_t wstride, int nmat, float* xa, float* xb, unsigned* ctr, unsigned* err,
                  hipStream_t s0, hipStream_t s1) {
  CK(hipMemsetAsync(ctr, 0, (size_t)(n + 1) * 128 * sizeof(unsigned), s0));
  CK(hipStreamSynchronize(s0));
  hipEvent_t e0, e1, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  CK(hipEventRecord(e0, s0));
  if (mode == 1) { CK(hipEventRecord(ej, s0)); CK(hipStreamWaitEvent(s1, ej, 0)); }
  for (int i = 0; i < n; ++i) {
    Args a{};
    a.W = W + (size_t)(i % nmat) * wstride; a.xin = (i & 1) ? xb : xa; a.xout = (i & 1) ? xa : xb; a.err = err;
    a.expect = (unsigned)grid;
    if (mode == 1) { a.wait = i ? ctr + (size_t)i * 128 : nullptr; a.done = ctr + (size_t)(i + 1) * 128; }
    hipLaunchKernelGGL(k_chain<LPT>, dim3(grid), dim3(256), 0, (mode == 1 && (i & 1)) ? s1 : s0, a);
  }
  if (mode == 1) { CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); }
  CK(hipEventRecord(e1, s0));
  CK(hipStreamSynchronize(s0));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / n;
}

int main() {
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const int n = 2000, nmat = 64;
  const size_t wmax = (size_t)1024 * 256 * 8;   // f4 per matrix at the largest shape (32 MB)
  f4* W; CK(hipMalloc(&W, wmax * nmat * sizeof(f4))); CK(hipMemset(W, 0, wmax * nmat * sizeof(f4)));
  float *xa, *xb; CK(hipMalloc(&xa, 4096)); CK(hipMalloc(&xb, 4096)); CK(hipMemset(xa, 0, 4096)); CK(hipMemset(xb, 0, 4096));
  unsigned *ctr, *err; CK(hipMalloc(&ctr, (size_t)(n + 1) * 128 * sizeof(unsigned))); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  struct Shape { int grid, lpt; } shapes[] = {{256, 2}, {512, 2}, {512, 4}, {1024, 8}};
  for (auto sh : shapes) {
    const double mb = (double)sh.grid * 256 * sh.lpt * 16 / 1e6;
    for (int mode = 0; mode < 2; ++mode) {
      double us = 0;
      for (int rep = 0; rep < 2; ++rep) {
        switch (sh.lpt) {
          case 2: us = run<2>(mode, sh.grid, n, W, wmax, nmat, xa, xb, ctr, err, s0, s1); break;
          case 4: us = run<4>(mode, sh.grid, n, W, wmax, nmat, xa, xb, ctr, err, s0, s1); break;
          default: us = run<8>(mode, sh.grid, n, W, wmax, nmat, xa, xb, ctr, err, s0, s1); break;
        }
      }
      unsigned h_err = 0; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
      printf("grid %4d  %5.1f MB per launch  %s : %.2f us per launch   (give-ups so far %u)\n", sh.grid, mb,
             mode ? "two streams + arrival counters" : "one stream (kernel boundary)   ", us, h_err);
    }
  }
  return 0;
}
