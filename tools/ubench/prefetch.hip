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
#include <algorithm>
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
  int use_xcc;         // piece -> XCD by the ACTUAL XCD id of the workgroup (HW_REG_XCC_ID)
  unsigned* where;     // nullable: XCD ids of workgroups 0..7 of this launch
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

// prefetcher flavour: U loads of 16 bytes in flight per thread (a CU needs ~50 KB in flight to draw its share of HBM)
template <int U>
__device__ __forceinline__ void touch_deep(const Seg& s, unsigned x, unsigned bl, unsigned nl, u4& acc) {
  if (s.p1 <= s.p0) return;
  const unsigned first = s.p0 + ((x + 8u - (s.p0 & 7u)) & 7u);
  const unsigned tid = threadIdx.x;
  for (unsigned p = first + 8u * bl; p < s.p1; p += 8u * nl * U) {
    u4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const unsigned pj = p + 8u * nl * j;
      v[j] = u4{0, 0, 0, 0};
      if (pj < s.p1) v[j] = *(s.base + (size_t)pj * 256 + tid);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
  }
}

__global__ __launch_bounds__(256) void k_layer(KArgs a) {
  const unsigned b = blockIdx.x, x = a.use_xcc ? xcc_id() : (b & 7u), bl = b >> 3, nl = gridDim.x >> 3;
  if (a.prog && b == 0 && threadIdx.x == 0) atomicAdd(a.prog, 1u);
  if (a.where && b < 8 && threadIdx.x == 0) a.where[b] = xcc_id();
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
  const int* owner;       // launch that consumes segment e: skipped if that launch has already started
  const int* need;        // segment e may be fetched once the launch counter >= rep * n_launch + need[e]
  int n, n_launch, reps;
  int deep;               // 1: 16 loads in flight per thread, cached launch counter
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
  int cur = 0;
  for (int rep = 0; rep < a.reps; ++rep) {
    for (int e = 0; e < a.n; ++e) {
      const int want = rep * a.n_launch + a.need[e];
      // the counter only grows: poll (a ~1 us round trip) only when the cached value does not already clear the segment
      if (!a.deep || cur < want) cur = (int)__hip_atomic_load(a.prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (cur < want) {
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > a.budget_ticks) {
          if (threadIdx.x == 0) atomicAdd(a.status, 1u);
          return;
        }
        cur = (int)__hip_atomic_load(a.prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (a.owner && cur > rep * a.n_launch + a.owner[e]) {   // too late: its consumer is already running
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(a.status + 3, 1u);
        continue;
      }
      if (a.deep) touch_deep<16>(a.segs[e], x, bl, nl, acc);
      else touch<false>(a.segs[e], x, bl, nl, acc);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) a.status[2] = 1;
  if (threadIdx.x == 0) atomicAdd(a.status + 1, 1u);
}

// ---- prefetcher v3: one poller wave + three loader waves per workgroup.  Loaders issue LDS-DMA loads
// (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPR destination, never waited for) so the only brake is
// the memory pipeline itself; the poller keeps the launch counter fresh in LDS so loaders never put a vector load
// (and its in-order vmcnt wait) between their prefetch loads. ------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prefetcher3(PArgs a) {
  __shared__ __attribute__((aligned(16))) char dump[4 * 1024];
  __shared__ unsigned s_bl;
  __shared__ int s_cur, s_state, s_done;
  const unsigned x = xcc_id();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) { s_bl = atomicAdd(a.xcd_ticket + x, 1u); s_cur = 0; s_state = 0; s_done = 0; }
  __syncthreads();
  const unsigned bl = s_bl, nl = gridDim.x >> 3;
  if (bl >= nl) return;
  volatile int* v_cur = &s_cur; volatile int* v_state = &s_state; volatile int* v_done = &s_done;
  if (wave == 0) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (*v_done < 3) {
      const int c = (int)__hip_atomic_load(a.prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) *v_cur = c;
      if (__builtin_amdgcn_s_memrealtime() - t0 > a.budget_ticks) {
        if (lane == 0) { *v_state = 2; atomicAdd(a.status, 1u); }
        return;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) atomicAdd(a.status + 1, 1u);
    return;
  }
  const unsigned lw = wave - 1, nslot = nl * 3u, slot = bl * 3u + lw;
  auto* ldst = (__attribute__((address_space(3))) void*)(dump + wave * 1024);
  unsigned skipped = 0;
  for (int rep = 0; rep < a.reps; ++rep) {
    for (int e = 0; e < a.n; ++e) {
      const int want = rep * a.n_launch + a.need[e];
      int cur = *v_cur;
      while (cur < want) {
        if (*v_state == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
        __builtin_amdgcn_s_sleep(1);
        cur = *v_cur;
      }
      if (a.owner && cur > rep * a.n_launch + a.owner[e]) { ++skipped; continue; }
      const Seg sg = a.segs[e];
      const unsigned first = sg.p0 + ((x + 8u - (sg.p0 & 7u)) & 7u);
      for (unsigned p = first + 8u * slot; p < sg.p1; p += 8u * nslot) {
        const char* src = reinterpret_cast<const char*>(sg.base) + (size_t)p * 4096 + lane * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024), ldst, 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) { atomicAdd((int*)&s_done, 1); if (skipped && blockIdx.x == 0 && lw == 0) atomicAdd(a.status + 3, skipped); }
}

struct Mat { size_t off_bytes; unsigned pieces; };


static hipStream_t st, st2;
static hipEvent_t e0, e1;
static float *xa, *xb;

template <typename F>
static float time_chain(int n, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) launch(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ms * 1000.f / (reps * n);
}

int main(int argc, char** argv) {
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipMalloc(&xa, 1 << 16)); CK(hipMalloc(&xb, 1 << 16)); CK(hipMemset(xa, 0, 1 << 16)); CK(hipMemset(xb, 0, 1 << 16));
  unsigned* d_misc; CK(hipMalloc(&d_misc, 8192 * 4)); CK(hipMemset(d_misc, 0, 8192 * 4));
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
  auto xio = [&](KArgs& a, int i) { a.xin = (i & 1) ? xa : xb; a.xout = (i & 1) ? xb : xa; };

  // ---- E1: XCD of workgroups 0..7, launch by launch inside a replayed graph ------------------------------------------
  {
    const int n = 20;
    unsigned* where = d_misc + 4096;
    for (int odd : {0, 1}) {
      time_chain(n, 2, [&](int i) {
        KArgs a{}; xio(a, i); a.where = where + 8 * i;
        int g = grids[i % 5]; if (odd && i % 3 == 0) g += 3;
        hipLaunchKernelGGL(k_layer, dim3(g), dim3(256), 0, st, a); });
      std::vector<unsigned> h(8 * n); CK(hipMemcpy(h.data(), where, 8 * n * 4, hipMemcpyDeviceToHost));
      printf("E1 (%s grids): XCD of workgroup 0 per launch:", odd ? "some non-multiple-of-8" : "multiple-of-8");
      for (int i = 0; i < n; ++i) printf(" %u", h[8 * i]);
      printf("   | workgroups 0..7 of launch 0:");
      for (int b = 0; b < 8; ++b) printf(" %u", h[b]);
      printf("\n");
    }
  }

  // ---- E2: the SAME matrix re-read by every launch vs a matrix pool: what survives a kernel boundary? --------------
  for (int use_xcc : {0, 1})
    for (size_t mb : {2, 16, 32}) {
      const int grid = mb <= 4 ? 256 : (mb <= 8 ? 512 : 1024);
      const size_t bytes = mb * MiB; const int nslots = (int)(layer_bytes * NL / bytes);
      float us[2];
      for (int same : {0, 1})
        us[same] = time_chain(100, 8, [&](int i) {
          KArgs a{}; xio(a, i); a.use_xcc = use_xcc;
          a.own = Seg{reinterpret_cast<const u4*>(pool + (same ? 0 : (size_t)(i % nslots) * bytes)), 0, (unsigned)(bytes / 4096)};
          hipLaunchKernelGGL(k_layer, dim3(grid), dim3(256), 0, st, a); });
      printf("E2 use_xcc=%d %3zu MiB grid %4d: pool %.2f us (%.2f TB/s)   same matrix %.2f us (%.2f TB/s)\n", use_xcc, mb, grid,
             us[0], bytes / us[0] / 1e6, us[1], bytes / us[1] / 1e6);
    }

  // ---- E3: every launch prefetches the next launch's matrix (pool) ----------------------------------------------------
  for (int use_xcc : {0, 1})
    for (size_t mb : {2, 8, 16}) {
      const int grid = mb <= 4 ? 256 : (mb <= 8 ? 512 : 1024);
      const size_t bytes = mb * MiB; const int nslots = (int)(layer_bytes * NL / bytes);
      float us[2];
      for (int pf : {0, 1})
        us[pf] = time_chain(100, 8, [&](int i) {
          KArgs a{}; xio(a, i); a.use_xcc = use_xcc;
          a.own = Seg{reinterpret_cast<const u4*>(pool + (size_t)(i % nslots) * bytes), 0, (unsigned)(bytes / 4096)};
          if (pf) a.pf[0] = Seg{reinterpret_cast<const u4*>(pool + (size_t)((i + 1) % nslots) * bytes), 0, (unsigned)(bytes / 4096)};
          hipLaunchKernelGGL(k_layer, dim3(grid), dim3(256), 0, st, a); });
      printf("E3 use_xcc=%d %2zu MiB: no prefetch %.2f us   prefetch-next %.2f us\n", use_xcc, mb, us[0], us[1]);
    }

  // ---- E4: decoder-layer chain with in-kernel prefetch plans ------------------------------------------------------------
  const int LAYERS = 48, REPS = 8;
  auto layer_chain = [&](int plan, double cap, int use_xcc, unsigned* prog) {
    return 5.f * time_chain(LAYERS * 5, REPS, [&](int i) {
      const int l = i / 5, k = i % 5;
      KArgs a{}; xio(a, i); a.use_xcc = use_xcc; a.prog = prog;
      a.own = seg(mat(l, k), 0, 1);
      if (plan == 1) {
        if (k == 0) a.pf[0] = seg(mat(l, 2), 0, 1);
        if (k == 1) a.pf[0] = seg(mat(l, 3), 0, cap * 0.5);
        if (k == 2) a.pf[0] = seg(mat(l, 3), cap * 0.5, cap);
        if (k == 3) a.pf[0] = seg(mat(l, 4), 0, 1);
        if (k == 4) a.pf[0] = seg(mat(l + 1, 0), 0, 1);
      }
      hipLaunchKernelGGL(k_layer, dim3(grids[k]), dim3(256), 0, st, a); });
  };
  printf("E4: us per layer (qkv 3 MiB, attn, o 2 MiB, gate/up 32 MiB, down 16 MiB; 53 MiB = %.2f us at 6.3 TB/s)\n", 53 * MiB / 6.3e6);
  for (int use_xcc : {0, 1}) {
    printf("  use_xcc=%d plan 0: %.2f", use_xcc, layer_chain(0, 0, use_xcc, nullptr));
    for (double cap : {0.25, 0.5, 0.75}) printf("   plan 1 cap %.2f: %.2f", cap, layer_chain(1, cap, use_xcc, nullptr));
    printf("\n");
  }

  // ---- E5: persistent prefetcher on a second stream (consumers map pieces by their actual XCD) --------------------------
  {
    std::vector<Seg> segs; std::vector<int> launch_of; std::vector<size_t> bytes_of;
    for (int l = 0; l < LAYERS; ++l)
      for (int k = 0; k < 5; ++k) if (sz[k]) { segs.push_back(seg(mat(l, k), 0, 1)); launch_of.push_back(l * 5 + k); bytes_of.push_back(sz[k]); }
    const int n = (int)segs.size(), n_launch = LAYERS * 5;
    Seg* d_segs; int* d_need; unsigned* prog = d_misc + 2048; unsigned* d_status = d_misc + 3000; unsigned* d_ticket = d_misc + 3100;
    CK(hipMalloc(&d_segs, n * sizeof(Seg))); CK(hipMalloc(&d_need, n * sizeof(int)));
    CK(hipMemcpy(d_segs, segs.data(), n * sizeof(Seg), hipMemcpyHostToDevice));
    printf("E5: persistent prefetcher on a second stream; us per layer (plan 0 with the launch counter: %.2f)\n", layer_chain(0, 0, 1, prog));
    for (int pgrid : {256}) {
      for (size_t ahead_mb : {8}) {
        std::vector<int> need(n);
        for (int e = 0; e < n; ++e) {
          size_t acc = 0; int j = e;
          while (j >= 0 && acc + bytes_of[j] <= ahead_mb * MiB) { acc += bytes_of[j]; --j; }
          if (j == e) j = e - 1;   // a matrix larger than the window: fetch it once its predecessor is consumed
          need[e] = j < 0 ? -1000000 : launch_of[j] + 1;
        }
        CK(hipMemcpy(d_need, need.data(), n * sizeof(int), hipMemcpyHostToDevice));
        CK(hipMemset(d_misc + 2048, 0, 2048 * 4));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n_launch; ++i) {
          const int l = i / 5, k = i % 5;
          KArgs a{}; xio(a, i); a.use_xcc = 1; a.prog = prog; a.own = seg(mat(l, k), 0, 1);
          hipLaunchKernelGGL(k_layer, dim3(grids[k]), dim3(256), 0, st, a);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        PArgs p{};
        p.segs = d_segs; p.need = d_need; p.n = n; p.n_launch = n_launch; p.reps = REPS; p.prog = prog;
        p.deep = 1; p.xcd_ticket = d_ticket; p.status = d_status; p.budget_ticks = 100000000LL / 10;   // 100 ms
        CK(hipStreamSynchronize(st));
        hipLaunchKernelGGL(k_prefetcher, dim3(pgrid), dim3(256), 0, st2, p);
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned hs[2]; CK(hipMemcpy(hs, d_status, 8, hipMemcpyDeviceToHost));
        printf("  prefetcher grid %3d ahead %2zu MiB: %.2f us/layer  (prefetcher workgroups finished %u, gave up %u)\n", pgrid,
               ahead_mb, ms * 1000.f / (REPS * LAYERS), hs[1], hs[0]);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      }
    }
  }
  // ---- E6: paced persistent prefetcher with 2 MiB sub-segments; second stream vs a parallel branch of the graph ------------
  {
    unsigned* prog = d_misc + 2048; unsigned* d_status = d_misc + 3000; unsigned* d_ticket = d_misc + 3100;
    const int n_launch = LAYERS * 5;
    printf("E6: paced prefetcher, 2 MiB sub-segments; us per layer (no prefetcher: %.2f)\n", layer_chain(0, 0, 1, prog));
    const int in_graph = 0;
    for (int deep : {3})
      for (int pgrid : {256})
        for (size_t SUB : {1 * MiB, 4 * MiB})
        for (size_t ahead_mb : {8, 12, 16, 20, 24, 32}) {
          std::vector<Seg> segs; std::vector<int> owner; std::vector<size_t> bytes_of;
          for (int l = 0; l < LAYERS; ++l)
            for (int k = 0; k < 5; ++k)
              for (size_t o = 0; o < sz[k]; o += SUB) {
                Mat m = mat(l, k);
                Seg sg; sg.base = reinterpret_cast<const u4*>(pool + m.off_bytes);
                sg.p0 = (unsigned)(o / 4096); sg.p1 = (unsigned)(std::min(o + SUB, sz[k]) / 4096);
                segs.push_back(sg); owner.push_back(l * 5 + k); bytes_of.push_back((size_t)(sg.p1 - sg.p0) * 4096);
              }
          // need[e] = smallest launch k such that the sub-segments of launches >= k up to e fit in the window
          std::vector<Seg> s2; std::vector<int> o2, need;
          for (size_t e = 0; e < segs.size(); ++e) {
            size_t acc = bytes_of[e]; int k = owner[e];     // window holds e itself
            long j = (long)e - 1;
            while (j >= 0 && acc + bytes_of[j] <= ahead_mb * MiB) { acc += bytes_of[j]; k = owner[j]; --j; }
            // everything of launches < owner[j+1] .. must have started: launch owner[j] must be running or done -> counter > owner[j]
            int nd = j < 0 ? -1000000 : owner[j] + 1;
            if (nd > owner[e]) continue;   // cannot be fetched before its own consumer starts: leave it to the consumer
            s2.push_back(segs[e]); o2.push_back(owner[e]); need.push_back(nd);
          }
          const int n = (int)s2.size();
          Seg* d_segs; int *d_need, *d_owner;
          CK(hipMalloc(&d_segs, n * sizeof(Seg))); CK(hipMalloc(&d_need, n * sizeof(int))); CK(hipMalloc(&d_owner, n * sizeof(int)));
          CK(hipMemcpy(d_segs, s2.data(), n * sizeof(Seg), hipMemcpyHostToDevice));
          CK(hipMemcpy(d_need, need.data(), n * sizeof(int), hipMemcpyHostToDevice));
          CK(hipMemcpy(d_owner, o2.data(), n * sizeof(int), hipMemcpyHostToDevice));
          CK(hipMemset(d_misc + 2048, 0, 2048 * 4));
          PArgs p{};
          p.segs = d_segs; p.need = d_need; p.owner = d_owner; p.n = n; p.n_launch = n_launch; p.reps = in_graph ? 1 : REPS; p.prog = prog;
          p.deep = deep; p.xcd_ticket = d_ticket; p.status = d_status; p.budget_ticks = 100000000LL / 20;   // 50 ms
          hipGraph_t g; hipGraphExec_t ge;
          hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
          CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
          if (in_graph) {
            // frame begin: reset the launch counter and the per-XCD tickets, then fork the prefetcher branch
            CK(hipMemsetAsync(d_misc + 2048, 0, 2048 * 4, st));
            CK(hipEventRecord(fork, st));
            CK(hipStreamWaitEvent(st2, fork, 0));
            hipLaunchKernelGGL(k_prefetcher, dim3(pgrid), dim3(256), 0, st2, p);
            CK(hipEventRecord(join, st2));
          }
          for (int i = 0; i < n_launch; ++i) {
            const int l = i / 5, k = i % 5;
            KArgs a{}; xio(a, i); a.use_xcc = 1; a.prog = prog; a.own = seg(mat(l, k), 0, 1);
            hipLaunchKernelGGL(k_layer, dim3(grids[k]), dim3(256), 0, st, a);
          }
          if (in_graph) CK(hipStreamWaitEvent(st, join, 0));
          CK(hipStreamEndCapture(st, &g));
          CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
          CK(hipStreamSynchronize(st));
          if (!in_graph) { if (deep == 3) hipLaunchKernelGGL(k_prefetcher3, dim3(pgrid), dim3(256), 0, st2, p); else hipLaunchKernelGGL(k_prefetcher, dim3(pgrid), dim3(256), 0, st2, p); }
          CK(hipEventRecord(e0, st));
          for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, st));
          CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          unsigned hs[4]; CK(hipMemcpy(hs, d_status, 16, hipMemcpyDeviceToHost));
          printf("  v%d prefetcher grid %3d sub %zu MiB window %2zu MiB (%d of %zu sub-segments scheduled): %.2f us/layer  (gave up %u, skipped-late(wg0) %u)\n",
                 deep, pgrid, SUB / MiB, ahead_mb, n, segs.size(), ms * 1000.f / (REPS * LAYERS), hs[0], hs[3]);
          CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
          CK(hipFree(d_segs)); CK(hipFree(d_need)); CK(hipFree(d_owner));
        }
  }
  return 0;
}
