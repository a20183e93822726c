// Micro-benchmark (MI355X): can the HBM idle time of a dependent launch chain be used to pull the NEXT launches'
// weights into the XCD-local L2?  Two mechanisms:
//   A. in-kernel prefetch: every kernel of the chain, besides streaming its own matrix, touches (a part of) the
//      matrices of the following launches with the SAME piece -> XCD mapping the consumer will use;
//   B. a persistent prefetcher kernel on a second stream walks the chain's matrices in consumption order, paced by a
//      progress counter the chain's kernels bump, so HBM also streams during the launch boundaries.
// A "layer" mimics one decoder layer of csm-1b at B = 1: QKV 3 MiB, attention (latency only), o_proj 2 MiB,
// gate/up 32 MiB, down 16 MiB; the matrices cycle over a pool larger than the 256 MiB memory-side cache.
// Piece = 4 KiB (one 16-byte load per thread of a 256-thread workgroup); piece p is always touched from XCD p % 8
// (workgroup b of a launch is assumed on XCD b % 8 -- experiment C checks that with HW_REG_XCC_ID).
// build: hipcc --offload-arch=gfx950 -O3 prefetch.hip -o prefetch ; run: ./prefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u4;

struct Seg { const u4* base; unsigned p0, p1; };   // pieces [p0, p1) of the matrix at `base`
struct KArgs {
  Seg own;             // streamed and "consumed"
  Seg pf[3];           // touched only (prefetch for later launches)
  const float* xin;    // 4 KiB written by the previous launch (the data dependency of the chain)
  float* xout;
  unsigned* prog;      // launch counter (bumped by workgroup 0), nullable
  int nt_own;          // consume with non-temporal loads
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

// pieces p in [p0, p1) with p % 8 == x, dealt round-robin to the `nl` workgroups of this XCD
template <bool NT>
__device__ __forceinline__ void touch(const Seg& s, unsigned x, unsigned bl, unsigned nl, u4& acc) {
  if (s.p1 <= s.p0) return;
  unsigned first = s.p0 + ((x + 8u - (s.p0 & 7u)) & 7u);
  const unsigned tid = threadIdx.x;
  // 4 loads in flight per thread per iteration
  for (unsigned p = first + 8u * bl; p < s.p1; p += 8u * nl * 4u) {
    u4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned pj = p + 8u * nl * j;
      v[j] = u4{0, 0, 0, 0};
      if (pj < s.p1) {
        const u4* q = s.base + (size_t)pj * 256 + tid;
        v[j] = NT ? __builtin_nontemporal_load(q) : *q;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
  }
}

__global__ __launch_bounds__(256) void k_layer(KArgs a) {
  const unsigned b = blockIdx.x, x = b & 7u, bl = b >> 3, nl = gridDim.x >> 3;
  if (a.prog && b == 0 && threadIdx.x == 0) atomicAdd(a.prog, 1u);
  float xv = a.xin[threadIdx.x * 4 % 1024];
  u4 acc = {0, 0, 0, 0};
  if (a.nt_own) touch<true>(a.own, x, bl, nl, acc); else touch<false>(a.own, x, bl, nl, acc);
  // "reduce + store": the dependent tail of a real kernel
  float r = xv + (float)(acc.x ^ acc.y ^ acc.z ^ acc.w) * 1e-30f;
  for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
  if (threadIdx.x == 0) a.xout[b & 1023] = r * 1e-30f;
  u4 acc2 = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 3; ++i) touch<false>(a.pf[i], x, bl, nl, acc2);
  if ((acc2.x ^ acc2.y ^ acc2.z ^ acc2.w) == 0x12345678u && xv == 3.f) a.xout[1024 + b] = 1.f;
}

// ---- experiment C: where do workgroups land? -------------------------------------------------------------------
__global__ void k_where(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// ---- experiment B: persistent prefetcher -------------------------------------------------------------------------
struct PArgs {
  const Seg* segs;        // n segments in consumption order
  const int* need;        // segment e may be fetched once the launch counter >= rep * n_launch + need[e]
  int n, n_launch, reps;
  const unsigned* prog;
  unsigned* xcd_ticket;   // [8] zeroed
  unsigned* status;       // [0] = number of workgroups that gave up (spin budget), [1] = finished
  long long budget_ticks; // s_memrealtime ticks (100 MHz) a workgroup may spin in total
};
__global__ __launch_bounds__(256) void k_prefetcher(PArgs a) {
  __shared__ unsigned s_bl;
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) s_bl = atomicAdd(a.xcd_ticket + x, 1u);
  __syncthreads();
  const unsigned bl = s_bl, nl = gridDim.x >> 3;
  if (bl >= nl) return;   // uneven placement: the surplus workgroup idles (its pieces are covered by the modulo below)
  u4 acc = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int rep = 0; rep < a.reps; ++rep) {
    for (int e = 0; e < a.n; ++e) {
      const int want = rep * a.n_launch + a.need[e];
      if (want > 0) {
        while ((int)__hip_atomic_load(a.prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(8);
          if (__builtin_amdgcn_s_memrealtime() - t0 > a.budget_ticks) {
            if (threadIdx.x == 0) atomicAdd(a.status, 1u);
            return;
          }
        }
      }
      touch<false>(a.segs[e], x, bl, nl, acc);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) a.status[2] = 1;
  if (threadIdx.x == 0) atomicAdd(a.status + 1, 1u);
}

struct Mat { size_t off_bytes; unsigned pieces; };

int main(int argc, char** argv) {
  hipStream_t st, st2;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  float *xa, *xb;
  CK(hipMalloc(&xa, 1 << 16)); CK(hipMalloc(&xb, 1 << 16)); CK(hipMemset(xa, 0, 1 << 16)); CK(hipMemset(xb, 0, 1 << 16));
  unsigned* d_misc; CK(hipMalloc(&d_misc, 4096 * 4)); CK(hipMemset(d_misc, 0, 4096 * 4));

  // ---- C: workgroup -> XCD --------------------------------------------------------------------------------------
  for (int grid : {256, 512, 1024, 8}) {
    hipLaunchKernelGGL(k_where, dim3(grid), dim3(256), 0, st, d_misc);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned> h(grid);
    CK(hipMemcpy(h.data(), d_misc, grid * 4, hipMemcpyDeviceToHost));
    int ok = 0;
    for (int b = 0; b < grid; ++b) ok += (h[b] == (unsigned)(b & 7));
    printf("C: grid %4d: %d / %d workgroups on XCD blockIdx %% 8  (first 16:", grid, ok, grid);
    for (int b = 0; b < 16 && b < grid; ++b) printf(" %u", h[b]);
    printf(")\n");
  }

  // ---- pool of layers -----------------------------------------------------------------------------------------------
  const size_t MiB = 1 << 20;
  const size_t sz[5] = {3 * MiB, 0, 2 * MiB, 32 * MiB, 16 * MiB};   // qkv, attn, o, gate/up, down
  const int grids[5] = {384, 256, 256, 1024, 512};
  const size_t layer_bytes = 53 * MiB;
  const int NL = argc > 1 ? atoi(argv[1]) : 24;   // layers in the pool (24 -> 1.27 GiB > MALL; 4 -> 212 MiB, like the decoder)
  char* pool; CK(hipMalloc(&pool, layer_bytes * NL)); CK(hipMemset(pool, 1, layer_bytes * NL));
  auto mat = [&](int layer, int k) {
    size_t off = (size_t)(layer % NL) * layer_bytes;
    for (int i = 0; i < k; ++i) off += sz[i];
    return Mat{off, (unsigned)(sz[k] / 4096)};
  };
  auto seg = [&](Mat m, double f0, double f1) {
    Seg s; s.base = reinterpret_cast<const u4*>(pool + m.off_bytes);
    s.p0 = (unsigned)(m.pieces * f0) & ~7u; s.p1 = f1 >= 1.0 ? m.pieces : ((unsigned)(m.pieces * f1) & ~7u);
    return s;
  };
  const int LAYERS = 48, REPS = 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned* prog = d_misc + 2048;

  // plan: for launch (layer l, kernel k) the prefetch segments.  plan id:
  //  0 none
  //  1 next launch's matrix, whole (gate/up capped at `cap` of it)
  //  2 balanced: qkv/attn/o each pull a third of `cap` x gate/up, gate/up pulls down, down pulls next qkv + o
  auto run_chain = [&](int plan, double cap, bool with_prog, int nt_own) -> float {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int i = 0;
    for (int l = 0; l < LAYERS; ++l)
      for (int k = 0; k < 5; ++k, ++i) {
        KArgs a{};
        a.own = seg(mat(l, k), 0, 1);
        a.xin = (i & 1) ? xa : xb; a.xout = (i & 1) ? xb : xa; a.prog = with_prog ? prog : nullptr; a.nt_own = nt_own;
        if (plan == 1) {
          if (k == 0) a.pf[0] = seg(mat(l, 2), 0, 1);                 // qkv -> o (attention has no matrix)
          if (k == 1) a.pf[0] = seg(mat(l, 3), 0, cap * 0.5);
          if (k == 2) a.pf[0] = seg(mat(l, 3), cap * 0.5, cap);
          if (k == 3) a.pf[0] = seg(mat(l, 4), 0, 1);
          if (k == 4) a.pf[0] = seg(mat(l + 1, 0), 0, 1);
        } else if (plan == 2) {
          if (k == 0) { a.pf[0] = seg(mat(l, 2), 0, 1); a.pf[1] = seg(mat(l, 3), 0, cap / 3); }
          if (k == 1) a.pf[0] = seg(mat(l, 3), cap / 3, 2 * cap / 3);
          if (k == 2) a.pf[0] = seg(mat(l, 3), 2 * cap / 3, cap);
          if (k == 3) a.pf[0] = seg(mat(l, 4), 0, 1);
          if (k == 4) a.pf[0] = seg(mat(l + 1, 0), 0, 1);
        }
        hipLaunchKernelGGL(k_layer, dim3(grids[k]), dim3(256), 0, st, a);
      }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.f / (REPS * LAYERS);
  };

  printf("A: us per layer (5 launches: qkv 3 MiB, attn, o 2 MiB, gate/up 32 MiB, down 16 MiB; 53 MiB -> %.2f us at 6.3 TB/s)\n", 53 * MiB / 6.3e6);
  for (int nt : {0, 1}) {
    printf("  nt_own=%d  plan 0 (no prefetch): %.2f us/layer\n", nt, run_chain(0, 0, false, nt));
    for (double cap : {0.25, 0.5, 0.75, 1.0}) {
      printf("  nt_own=%d  plan 1 cap %.2f: %.2f us/layer", nt, cap, run_chain(1, cap, false, nt));
      printf("   plan 2 cap %.2f: %.2f us/layer\n", cap, run_chain(2, cap, false, nt));
    }
  }

  // ---- uniform chains: one matrix size, every launch prefetches the next launch's matrix ------------------------------------
  for (size_t mb : {2, 4, 8, 16, 24}) {
    for (int pf : {0, 1}) {
      const int grid = mb <= 4 ? 256 : (mb <= 8 ? 512 : 1024);
      const size_t bytes = mb * MiB; const int nslots = (int)(layer_bytes * NL / bytes); const int N = 200;
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < N; ++i) {
        KArgs a{};
        a.own = Seg{reinterpret_cast<const u4*>(pool + (size_t)(i % nslots) * bytes), 0, (unsigned)(bytes / 4096)};
        if (pf) a.pf[0] = Seg{reinterpret_cast<const u4*>(pool + (size_t)((i + 1) % nslots) * bytes), 0, (unsigned)(bytes / 4096)};
        a.xin = (i & 1) ? xa : xb; a.xout = (i & 1) ? xb : xa;
        hipLaunchKernelGGL(k_layer, dim3(grid), dim3(256), 0, st, a);
      }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const float us = ms * 1000.f / (REPS * N);
      printf("A-uniform: %2zu MiB grid %4d prefetch-next %d: %.2f us/launch -> %.2f TB/s\n", mb, grid, pf, us, bytes / us / 1e6);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }

  // ---- B: persistent prefetcher on a second stream ---------------------------------------------------------------------
  {
    // segments = the matrices of the chain in consumption order (attention has none)
    std::vector<Seg> segs; std::vector<int> launch_of; std::vector<size_t> bytes_of;
    for (int l = 0; l < LAYERS; ++l)
      for (int k = 0; k < 5; ++k) if (sz[k]) { segs.push_back(seg(mat(l, k), 0, 1)); launch_of.push_back(l * 5 + k); bytes_of.push_back(sz[k]); }
    const int n = (int)segs.size(), n_launch = LAYERS * 5;
    Seg* d_segs; int* d_need; unsigned* d_status = d_misc + 3000; unsigned* d_ticket = d_misc + 3100;
    CK(hipMalloc(&d_segs, n * sizeof(Seg))); CK(hipMalloc(&d_need, n * sizeof(int)));
    CK(hipMemcpy(d_segs, segs.data(), n * sizeof(Seg), hipMemcpyHostToDevice));
    printf("B: persistent prefetcher (second stream), us per layer vs plan 0; ahead = bytes it may run ahead of the chain\n");
    for (int pgrid : {256, 512}) {
      for (size_t ahead_mb : {8, 16, 24, 32, 48}) {
        // need[e] = first launch index k such that the bytes of segments belonging to launches k..launch_of[e] (inclusive
        // of e) fit in `ahead`: segment e may be fetched once launch k has started (everything before k is consumed)
        std::vector<int> need(n);
        for (int e = 0; e < n; ++e) {
          size_t acc = 0; int j = e;
          while (j >= 0 && acc + bytes_of[j] <= ahead_mb * MiB) { acc += bytes_of[j]; --j; }
          // segments j+1..e fit; segment j must be consumed, i.e. the launch after launch_of[j] must have started
          need[e] = j < 0 ? launch_of[0] - n_launch + 0 : launch_of[j] + 1;
          if (j < 0) need[e] = -1000000;
        }
        // later replays: the same table shifted by rep * n_launch (segments of the previous replay are long consumed)
        CK(hipMemcpy(d_need, need.data(), n * sizeof(int), hipMemcpyHostToDevice));
        CK(hipMemset(d_misc + 2048, 0, 2048 * 4));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int i = 0;
        for (int l = 0; l < LAYERS; ++l)
          for (int k = 0; k < 5; ++k, ++i) {
            KArgs a{};
            a.own = seg(mat(l, k), 0, 1);
            a.xin = (i & 1) ? xa : xb; a.xout = (i & 1) ? xb : xa; a.prog = prog;
            hipLaunchKernelGGL(k_layer, dim3(grids[k]), dim3(256), 0, st, a);
          }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        PArgs p{};
        p.segs = d_segs; p.need = d_need; p.n = n; p.n_launch = n_launch; p.reps = REPS; p.prog = prog;
        p.xcd_ticket = d_ticket; p.status = d_status; p.budget_ticks = 100000000LL / 10;   // 100 ms
        CK(hipStreamSynchronize(st));
        hipLaunchKernelGGL(k_prefetcher, dim3(pgrid), dim3(256), 0, st2, p);
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned hs[2]; CK(hipMemcpy(hs, d_status, 8, hipMemcpyDeviceToHost));
        printf("  prefetcher grid %3d ahead %2zu MiB: %.2f us/layer  (prefetcher workgroups finished %u, gave up %u)\n", pgrid,
               ahead_mb, ms * 1000.f / (REPS * LAYERS), hs[1], hs[0]);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
      }
    }
  }
  return 0;
}
