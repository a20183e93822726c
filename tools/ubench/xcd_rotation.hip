// Where does workgroup 0 of a dispatch land?  The weight streamer (csrc/prefetch.h) assumes workgroup b of EVERY dispatch of the engine
// stream runs on XCD (b + rot) % 8 with one `rot` measured at engine creation.  This probe launches grids of 8 / 257 / 385 / 1024 / 4
// workgroups back to back (eagerly and as one hipGraph) and prints the XCD of workgroups 0..7 of each launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void where(unsigned* out) {
  if (threadIdx.x == 0 && blockIdx.x < 8) { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); out[blockIdx.x] = v & 15u; }
}
int main() {
  const int grids[] = {8, 8, 257, 8, 385, 8, 1024, 8, 4, 8, 192, 8, 257, 257, 8};
  const int n = sizeof(grids) / sizeof(int);
  unsigned* d; CK(hipMalloc(&d, n * 8 * 4)); 
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(d, 0xff, n * 8 * 4));
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    if (mode) CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(where, dim3(grids[i]), dim3(256), 0, st, d + i * 8);
    if (mode) { CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphLaunch(ge, st)); }
    CK(hipStreamSynchronize(st));
    std::vector<unsigned> h(n * 8); CK(hipMemcpy(h.data(), d, n * 8 * 4, hipMemcpyDeviceToHost));
    printf("%s\n", mode ? "hipGraph replay:" : "eager launches:");
    for (int i = 0; i < n; ++i) { printf("  grid %4d  XCD of workgroups 0..7:", grids[i]); for (int b = 0; b < 8 && b < grids[i]; ++b) printf(" %u", h[i * 8 + b]); printf("\n"); }
  }
  return 0;
}
