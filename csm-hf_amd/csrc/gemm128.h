// Matrix-core GEMM for the FFN launches of a 65..128-row batched decode step (BASELINE configs[3], the 128-row strong leg).
//
// gemm32.h splits K across the eight waves of a workgroup, so every wave walks its own slice of ALL activation planes: at 128
// rows the planes (6 bytes per row and k in exact mode) are the larger operand and a decoder gate/up launch moved 268 MB through
// the CUs' vector-memory path for 33.5 MB of weights (21.4 us against a 5.2 us matrix-pipe time).  Here the waves of a
// workgroup split the WEIGHT ROWS and share the activation planes through LDS:
//   workgroup tile = (4 waves x PT weight tiles of 16 rows) x 64 batch rows x one or all 1 024-wide k groups,
//   planes: LDS-DMA (`global_load_lds`, 1 KiB per wave instruction: the plane copy is already in fragment order) into a three-stage
//           ring of 128-wide k chunks (48 KiB each in exact mode), two chunks ahead of the math,
//   weights: fragment-order copy (tile16_kernel) straight into ACCUMULATION registers (the matrix instructions read their A operand
//           from there), three chunks ahead (four register sets),
//   one bare s_barrier per chunk behind a hand-counted vmcnt (the loads of the next chunks stay in flight across it); the loads a chunk
//   issues are spread over its matrix instructions; B fragments leave LDS two groups ahead of their use (hand-counted lgkmcnt).
// A decoder gate/up launch then moves 166 MB (z = 2: 2 x 33.5 MB weights + 256 x 393 KB planes) and each plane fragment read from
// LDS feeds PT matrix instructions.
//
// Every load and wait of the chunk loop is inline assembly.  What the compiler did with its own (tools/ubench/g128_bench.hip, each a
// measured step of profiles/r05_g128.md): LDS reads sunk to their use, one live fragment, lgkmcnt(0) per pair of matrix instructions;
// LDS reads behind an LDS-DMA write: vmcnt(0) (possible alias); weight loads beside the DMA stream: vmcnt(0) at the first use of every
// register set (it does not count across the two kinds of vector-memory operation); loads inside `if`: a wait each; accumulators in
// accumulation registers: 64 v_accvgpr moves per chunk for the group sum (this unit is built with --amdgpu-mfma-vgpr-form).
//
// Results are bit-identical to gemm16.h / gemm32.h (a row's value must not depend on the batch size): per 128-wide chunk a fresh
// accumulator (j ascending; lo, mid, hi inside a step), chunks summed eight at a time in order, the groups of eight then in order from
// zero -- the association of the eight-wave LDS reduction + slab sum there.  A workgroup therefore takes ONE group (K split across
// workgroups, slab exchange as in gemm16.h) or ALL of them in order (gate/up at K = 2 048: two groups, no exchange).
// Prologues / epilogues: (plain | RMS scale from the producer's sums of squares) x (residual + output planes + sums of squares | SwiGLU).
//
// Measured (MI355X, 128 rows, exact planes; in-step rocprofv3 averages): decoder gate/up 21.4 -> 18.1 us, decoder down_proj 17.8 -> 17.6,
// backbone gate/up 51.3 -> 37 (two groups in one workgroup), backbone down_proj 33.4 -> 32.1; frame-step 11.17 -> 10.56 ms (fp32 KV),
// 10.44 with the default bf16 cache.  The chunk period is 1.2-1.5 us against 0.7 us of matrix-pipe time: the per-CU operand stream
// (80 KiB per chunk, two to three chunks in flight) is what the period follows -- eight waves per workgroup (H = 2: two waves per SIMD)
// measure the same, a 64-row launch (half the workgroups, half the bytes) takes 15.4 us of the 18.9.
#pragma once
#include <type_traits>
#include "gemm32.h"

// The launch's arguments, compact (144 bytes instead of the 600-byte GemvArgs by value) and requested in ONE batch at kernel entry
// (the asm statement at the top of the kernel): the compiler otherwise requests a field of the argument segment where it is first used,
// a scalar-cache miss each, the epilogue's among them.
struct G128Args {
  const void* Wt;          // fragment-order weights (tile16_kernel)
  const bf16_t* xplanes;   // input planes
  const float* xss;        // input sums of squares (PRO_NORM)
  bf16_t* oplanes;         // output planes, nullable
  const float* oln;        // the consumer's norm weight folded into the output planes, nullable
  float* oss;              // output sums of squares (EPI_RESID), nullable
  float* out;              // residual stream (EPI_RESID) | fp32 output (EPI_SWIGLU without planes)
  const float* wscale;     // per-row scale of fp8 weights, nullable
  float* slabs;
  int* tickets;
  int* bump_a;
  int* bump_b;
  uint32_t* dbg;           // tools/ubench/g128_bench.hip time stamps
  int xss_n, xss_ld, oss_ld, ldo, K, N, M, KB;
  float eps;
  int xcdmap;
};

#ifndef CSM_ARGS_ONLY
template <int I, int N, typename F>
__device__ __forceinline__ void g128_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    g128_static_for<I + 1, N>(f);
  }
}

// Weight fragment of this kernel: requested by inline assembly straight into ACCUMULATION registers (the matrix instructions read
// their A operand from there; the 128 registers of the four sets stay out of the 256 architectural ones) and made visible by the
// hand-counted vmcnt of the chunk loop.  As compiler-visible loads beside the LDS-DMA stream, every first use of a register set drained
// the whole queue (vmcnt(0): the compiler does not count across the two kinds of vector-memory operations).
template <typename WT>
struct G128Frag;
template <>
struct G128Frag<bf16_t> {
  u32x4 r;
  static constexpr int JB = 1024;   // bytes between the k steps of a chunk
  template <int OFF>
  __device__ __forceinline__ void request(const bf16_t* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(r) : "v"(p), "n"(OFF) : "memory"); }
  __device__ __forceinline__ bf16x8 get() const { return __builtin_bit_cast(bf16x8, r); }
};
template <>
struct G128Frag<fp8_t> {
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
  u32x2_t r;
  static constexpr int JB = 512;
  template <int OFF>
  __device__ __forceinline__ void request(const fp8_t* p) { asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=a"(r) : "v"(p), "n"(OFF) : "memory"); }
  __device__ __forceinline__ bf16x8 get() const {  // e4m3 -> fp32 (exact) -> bf16 by truncation (exact), as AFrag<fp8_t>
    bf16x8 f;
    uint32_t* pf = reinterpret_cast<uint32_t*>(&f);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int w = (int)(h < 2 ? r[0] : r[1]);
      const f32x2 v = (h & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8(w, true) : __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
      pf[h] = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xffff0000u);
    }
    return f;
  }
};

template <typename WT, int PRO, int EPI, int PT, bool ONE, int H = 1>
__global__ __launch_bounds__(256 * H) void gemm128_kernel(const G128Args a) {
  // every field of the argument segment requested in ONE batch at entry
  asm volatile("" ::"s"(a.Wt), "s"(a.xplanes), "s"(a.xss), "s"(a.oplanes), "s"(a.oln), "s"(a.oss), "s"(a.out), "s"(a.wscale));
  asm volatile("" ::"s"(a.slabs), "s"(a.tickets), "s"(a.bump_a), "s"(a.bump_b), "s"(a.dbg), "s"(a.xss_n), "s"(a.xss_ld), "s"(a.oss_ld), "s"(a.ldo), "s"(a.K),
               "s"(a.N), "s"(a.M), "s"(a.KB), "s"(a.eps), "s"(a.xcdmap));
  const int M = a.M, KB = a.KB;
  float* const slabs = a.slabs;
  int* const tickets = a.tickets;
  static_assert(EPI == EPI_RESID || EPI == EPI_SWIGLU, "FFN epilogues only");
#ifdef CSM_G128_VARIANT   // TIMING-ONLY builds of tools/ubench/g128_bench.hip (wrong results): 1 no plane DMA, 2 no weight loads, 4 no MFMAs, 8 no LDS fragment
  constexpr int ko = CSM_G128_VARIANT;   // reads, 16 no barriers, 32 every workgroup starts its plane walk at another chunk, 64 chunk time stamps to a.dbg
#else
  constexpr int ko = 0;
#endif
  constexpr int NP = ONE ? 1 : 3;        // activation planes
  // H = 1: four waves, each all four batch tiles of the workgroup (one wave per SIMD).  H = 2: eight waves -- waves w and w + 4 share
  // a panel of weight tiles (both request it: twice the weight instructions per CU) and take two batch tiles each, so that a SIMD holds
  // two waves and one issues matrix instructions while the other stands at a barrier, a full vector-memory queue or an LDS wait
  // H = 2 is a MEASUREMENT form (tools/ubench/g128_bench.hip -DCSM_G128_H=2: 18.7 against 18.2 us for gate/up, 17.3 against 17.6 for
  // down_proj at 128 full rows); the library instantiates H = 1 only -- the eight-wave form was never validated on partial batches
  static_assert(H == 1 || H == 2, "one or two waves per SIMD");
  constexpr int MT = 4 / H;              // batch tiles (16 rows) per WAVE; the workgroup always holds four
  constexpr int STAGE = 4 * NP * 4 * 1024;   // bytes of one 128-wide chunk of planes: fragment (mt, plane, j) at ((mt * NP + plane) * 4 + j) KiB
  constexpr int NPF = NP * 4 / H;        // plane DMA instructions per wave and chunk (waves 4 h' + .. : batch tile `wave / H`, k steps (wave % H) * 4 / H ..)
  constexpr int NWF = PT * 4;            // weight loads per wave and chunk
  extern __shared__ __attribute__((aligned(1024))) uint8_t g128_lds[];   // 3 stages | stat[64]
  float* stat = reinterpret_cast<float*>(g128_lds + 3 * STAGE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((ko & 64) && lane == 0 && wave < 4) a.dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + wave * 16 + 10] = (uint32_t)__builtin_amdgcn_s_memrealtime();
  if ((ko & 64) && lane == 0 && wave < 4) a.dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + wave * 16 + 12] = (uint32_t)__builtin_amdgcn_s_memtime();
  const int K = a.K;
  const int m = lane & 15, g = lane >> 4;
  const int mtiles = (M + 15) >> 4;
  // K split across workgroups (KB > 1): the k group is the FASTEST grid index, so that the workgroups of one group -- they read the same
  // 1 024-wide slice of the planes -- run on one XCD (workgroup id mod 8 with KB = 8) and that slice (786 KB at 128 rows) stays in its
  // L2; with the panels fastest every XCD walked all 6.3 MB of a down_proj's planes (FETCH_SIZE 75 MB per launch for 17 MB of weights)
  int pid = KB > 1 ? (int)blockIdx.y : (int)blockIdx.x;                                                        // panel of 4 x PT weight tiles
  const int np = KB > 1 ? (int)gridDim.y : (int)gridDim.x;
  if (EPI == EPI_SWIGLU && a.xcdmap && KB == 1 && (np & 7) == 0) pid = (pid & 7) * (np >> 3) + (pid >> 3);   // gate/up: the XCD that reads these h columns as a k group of the down_proj launch (128 rows 10.28 -> 10.18 ms)
  const int kid = KB > 1 ? (int)blockIdx.x : 0;                                                                 // k group
  const int mt0 = (int)blockIdx.z * 4;
  const int mtw = (wave >> 2) * MT;                                    // this wave's first batch tile among the workgroup's four
  const int bx = (int)blockIdx.z * np + pid;                           // slab / ticket slot
  const int wp = pid * 4 + (wave & 3);                     // this wave's panel of PT weight tiles
  const int G = (K >> 10) / KB;                                        // 1 024-wide k groups of this workgroup: one (K split across workgroups) or all (KB == 1)
  const int c0 = kid * 8 * G;                              // first 128-wide chunk of this workgroup
  const size_t ps = (size_t)K * 16;

  // ---- epilogue inputs first (oldest in the vmcnt queue): residual quads, the consumer's norm weight, the sums of squares ----------
  f32x4 rq[PT][MT], lq[PT];
  if (EPI == EPI_RESID) {
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int n0 = ((wp * PT + t) * 16 + g * 4);
      // unconditional loads on clamped indices (a load inside `if` is followed by its own wait: eight dependent round trips)
      const int nc = n0 < a.N ? n0 : 0;
      lq[t] = (f32x4)(1.f);
      if (a.oplanes && a.oln) lq[t] = *reinterpret_cast<const f32x4*>(a.oln + nc);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int mm = min((mt0 + mtw + mt) * 16 + m, M - 1);
        rq[t][mt] = *reinterpret_cast<const f32x4*>(a.out + (size_t)mm * a.ldo + nc);   // rows / columns beyond the matrix are never stored
      }
    }
  }
  // sums of squares of the rows of batch tile `wave` (gemm32.h's arithmetic): requested by inline assembly as the oldest loads of the
  // wave and consumed behind chunk 0's counted wait -- a compiler-visible load consumed ahead of the loop drains the whole prologue
  // (three chunks of operands, vmcnt(0): entry -> first matrix instruction 4.3 us instead of 3.3)
  f32x4 pv8[8];
  if (PRO == PRO_NORM && wave < 4) {
    const int row = min((mt0 + wave) * 16 + m, M - 1);
    const float* sp = a.xss + (size_t)row * a.xss_ld;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = g * 4 + 16 * i;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pv8[i]) : "v"(sp + (t < a.xss_n ? t : 0)) : "memory");
    }
  }
  if ((ko & 64) && lane == 0 && wave < 4) a.dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + wave * 16 + 15] = (uint32_t)__builtin_amdgcn_s_memrealtime();
  // ---- operand streams ---------------------------------------------------------------------------------------------------------------
  const int ntiles = (a.N + 15) >> 4;
  const WT* wbase[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const size_t tile = (size_t)min((wp * PT + t), ntiles - 1);
    wbase[t] = reinterpret_cast<const WT*>(a.Wt) + ((tile * (size_t)(K >> 7) + c0) * 4) * 512 + lane * 8;
  }
  const int ptile = wave / H, pj0 = (wave % H) * (4 / H);             // the batch tile and the first k step this wave fetches
  const bf16_t* pbase = a.xplanes + (size_t)min(mt0 + ptile, mtiles - 1) * 3 * ps + ((size_t)c0 * 256 + lane) * 8 + (size_t)pj0 * 512;
  G128Frag<WT> wf[4][PT][4];
  auto issue_w = [&](int c, auto& dst) {
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const WT* p = wbase[t] + (size_t)c * 2048;
      dst[t][0].template request<0>(p);
      dst[t][1].template request<G128Frag<WT>::JB>(p);
      dst[t][2].template request<2 * G128Frag<WT>::JB>(p);
      dst[t][3].template request<3 * G128Frag<WT>::JB>(p);
    }
  };
  auto issue_p = [&](int c, int st) {
    uint8_t* dst = g128_lds + st * STAGE + ptile * (NP * 4 * 1024) + pj0 * 1024;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int j = 0; j < 4 / H; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pbase + p * ps + ((size_t)c * 4 + j) * 512),
                                         (__attribute__((address_space(3))) void*)(dst + (p * 4 + j) * 1024), 16, 0, 0);
  };
  // the k-th of the NPF + NWF loads a chunk issues for later chunks (planes of chunk cp first, then the weights of chunk cw): spread over
  // the chunk's matrix instructions.  As a burst at the chunk's start they filled the vector-memory queue (1 KiB per instruction, 16
  // clocks each, four waves on one path) and the ONE wave of a SIMD stood at the next load instead of issuing matrix instructions:
  // +0.5 us per chunk (tools/ubench/g128_bench.hip, variants 64 / 65)
  auto issue_kth = [&](auto K_, auto IP_, auto IW_, int cp, int cw, int st, auto& wdst) {
    constexpr int k = decltype(K_)::value;
    if constexpr (k < NPF) {
      if constexpr (decltype(IP_)::value) {
        constexpr int p = k / (4 / H), j = k % (4 / H);
        uint8_t* dst = g128_lds + st * STAGE + ptile * (NP * 4 * 1024) + pj0 * 1024;
        if (!(ko & 1))
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pbase + p * ps + ((size_t)cp * 4 + j) * 512),
                                           (__attribute__((address_space(3))) void*)(dst + (p * 4 + j) * 1024), 16, 0, 0);
      }
    } else if constexpr (k < NPF + PT * 4) {
      if constexpr (decltype(IW_)::value) {
        constexpr int t = (k - NPF) / 4, j = (k - NPF) % 4;
        if (!(ko & 2)) wdst[t][j].template request<j * G128Frag<WT>::JB>(wbase[t] + (size_t)cw * 2048);
      }
    }
  };
  // issue order: W(0) P(0) W(1) P(1) W(2), then per chunk c: P(c + 2) W(c + 3) -- behind P(c) there are W(c+1), P(c+1), W(c+2)
  f32x4 sg[PT][MT];   // the group's sum: chunks in order, each from a fresh accumulator
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) sg[t][mt] = (f32x4)(0.f);
  int st_use = 0, st_iss = 2;
  int cg = 0;   // first chunk of the current group, relative to c0
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) uint8_t*)g128_lds;   // LDS byte address of the ring
  // One 128-wide chunk.  U = register set of its weights, IP / IW = whether the planes of chunk c + 2 / the weights of chunk c + 3 exist,
  // WAIT = loads issued behind the planes of chunk c -- all compile-time, so that the compiler's own vmcnt bookkeeping for the weight
  // registers stays exact (with run-time conditions around the issues it drained the queue, vmcnt(0), once per four chunks: 20 us per launch)
  using std::integral_constant;
  f32x4 accs[2][PT][MT];
  auto chunk = [&](int c, auto U_, auto IP_, auto IW_, auto WAIT_, auto PEND_) {
    constexpr int u = decltype(U_)::value, WAIT = decltype(WAIT_)::value;
    constexpr bool PEND = decltype(PEND_)::value;
    constexpr bool IP = decltype(IP_)::value, IW = decltype(IW_)::value;
    // planes of chunk c landed (this wave's share), then everybody's; the loads issued behind them stay in flight
    if ((ko & 64) && c == 0 && lane == 0 && wave < 4) a.dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + wave * 16 + 14] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    // (the wait names the weight registers of this chunk -- requested before its planes -- so that nothing that reads them moves above it)
#pragma unroll
    for (int t = 0; t < PT; ++t)
      asm volatile("s_waitcnt vmcnt(%4)" : "+a"(wf[u][t][0].r), "+a"(wf[u][t][1].r), "+a"(wf[u][t][2].r), "+a"(wf[u][t][3].r) : "n"(WAIT) : "memory");
    if (!(ko & 16)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if ((ko & 64) && lane == 0 && wave < 4) a.dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + wave * 16 + c] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    const int st_now = st_iss;
    st_iss = st_iss == 2 ? 0 : st_iss + 1;
    unsigned sb = lds0 + (unsigned)(st_use * STAGE + mtw * (NP * 4 * 1024) + lane * 16);
    st_use = st_use == 2 ? 0 : st_use + 1;
    f32x4(&acc)[PT][MT] = accs[u & 1];   // two accumulator sets: the previous chunk's is added into the group sum while this chunk multiplies
    // B fragments: LDS reads as inline assembly with hand-counted lgkmcnt.  As compiler-visible LDS loads they may alias the LDS-DMA
    // writes in flight (run-time stage index), and the compiler then drains vmcnt(0) ahead of their first use -- the whole operand queue,
    // every chunk.  A group = the MT fragments of one plane of one k step (lo, mid, hi inside a step: small terms first); group q + 2 is
    // requested while group q multiplies (three register sets; one wave per SIMD, nothing else hides the LDS latency).
    constexpr int NG = 4 * NP;
    bf16x8 bq[3][MT];
    auto lds_group = [&](auto Q_, unsigned sbv) {   // fragment (mt, plane, j) at ((mt * NP + plane) * 4 + j) KiB of the stage: immediate offsets, ONE address register
      constexpr int q = decltype(Q_)::value, j = q / NP, p = NP - 1 - q % NP;   // (sbv: a variable used only as an asm operand is not captured by a nested generic lambda)
      bf16x8(&d)[MT] = bq[q % 3];
      if (ko & 8) {
#pragma unroll
        for (int i = 0; i < MT; ++i) d[i] = (bf16x8)(short)(q + 1);
        return;
      }
      if constexpr (MT == 4)
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                     : "v"(sbv), "n"(((0 * NP + p) * 4 + j) * 1024), "n"(((1 * NP + p) * 4 + j) * 1024), "n"(((2 * NP + p) * 4 + j) * 1024),
                       "n"(((3 * NP + p) * 4 + j) * 1024));
      else
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                     : "=&v"(d[0]), "=&v"(d[1])
                     : "v"(sbv), "n"(((0 * NP + p) * 4 + j) * 1024), "n"(((1 * NP + p) * 4 + j) * 1024));
    };
    lds_group(integral_constant<int, 0>{}, sb);
    lds_group(integral_constant<int, 1>{}, sb);
    g128_static_for<0, NG>([&](auto Q_) {
      constexpr int q = decltype(Q_)::value, j = q / NP;
      if constexpr (q + 2 < NG) lds_group(integral_constant<int, (q + 2 < NG ? q + 2 : 0)>{}, sb);
      bf16x8(&b)[MT] = bq[q % 3];
      // ties the wait to the registers: the MFMAs that read them cannot move above it
      if constexpr (MT == 4) {
        if constexpr (q + 2 < NG) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        else if constexpr (q + 1 < NG) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      } else {
        if constexpr (q + 2 < NG) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b[0]), "+v"(b[1]));
        else if constexpr (q + 1 < NG) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(b[0]), "+v"(b[1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]));
      }
      {   // this group's share of the chunk's loads, ahead of its matrix instructions
        constexpr int L = NPF + NWF, k0 = q * L / NG, k1 = (q + 1) * L / NG;
        g128_static_for<k0, k1>([&](auto K_) { issue_kth(K_, IP_, IW_, cg + c + 2, cg + c + 3, st_now, wf[(u + 3) & 3]); });
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < PT; ++t)   // a fresh accumulator per chunk: the first product starts from a literal zero
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][t][j].get(), b[mt], q == 0 ? (f32x4)(0.f) : acc[t][mt], 0, 0, 0);
      if constexpr (q == 1 && PEND) {   // the previous chunk's accumulators: VALU work in the shadow of this chunk's matrix instructions
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) sg[t][mt] += accs[(u & 1) ^ 1][t][mt];
      }
    });
  };
  // Straight-line code over the eight chunks of a 1 024-wide k group, in a loop over the workgroup's groups (K = 2 048 gate/up launches: two
  // groups in ONE workgroup instead of two workgroups + a slab exchange; the sum "groups in order from zero" is the same).  The pipeline
  // restarts at a group boundary.  Every wait of the loop is hand-counted: a back-edge costs nothing (with compiler-counted weight loads
  // it meant vmcnt(0) once per trip).
  constexpr integral_constant<bool, true> yes{};
  constexpr integral_constant<bool, false> no{};
  // loads that may stay in flight across a chunk's first barrier: everything the PREVIOUS chunk issued (planes and weights are interleaved
  // there); chunks 0 and 1 stand behind the prologue's order W(0) P(0) W(1) P(1) W(2)
  constexpr integral_constant<int, NWF + NPF + NWF> w_pro{};
  constexpr integral_constant<int, NPF + NWF> w_full{};
  f32x4 s[PT][MT];
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) s[t][mt] = (f32x4)(0.f);
  for (int grp = 0; grp < G; ++grp) {
  {   // issue order: W(0) P(0) W(1) P(1) W(2); stages: chunk c of the walk sits in stage c % 3
    const int s0 = st_use, s1 = s0 == 2 ? 0 : s0 + 1;
    issue_w(cg, wf[0]);
    issue_p(cg, s0);
    issue_w(cg + 1, wf[1]);
    issue_p(cg + 1, s1);
    issue_w(cg + 2, wf[2]);
    st_iss = s1 == 2 ? 0 : s1 + 1;
  }
  chunk(0, integral_constant<int, 0>{}, yes, yes, w_pro, no);
  if (PRO == PRO_NORM && wave < 4 && grp == 0) {   // chunk 1's barrier publishes it (read by the epilogue)
    asm volatile("" : "+v"(pv8[0]), "+v"(pv8[1]), "+v"(pv8[2]), "+v"(pv8[3]), "+v"(pv8[4]), "+v"(pv8[5]), "+v"(pv8[6]), "+v"(pv8[7]));   // behind chunk 0's wait
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = g * 4 + 16 * i;
      const f32x4 v = t < a.xss_n ? pv8[i] : (f32x4)(0.f);
      q += (v[0] + v[1]) + (v[2] + v[3]);
    }
    q = xor32_sum(xor16_sum(q));
    if (lane < 16) stat[wave * 16 + lane] = __builtin_amdgcn_rsqf(q * __builtin_amdgcn_rcpf((float)K) + a.eps);
  }
  chunk(1, integral_constant<int, 1>{}, yes, yes, w_pro, yes);
  chunk(2, integral_constant<int, 2>{}, yes, yes, w_full, yes);
  chunk(3, integral_constant<int, 3>{}, yes, yes, w_full, yes);
  chunk(4, integral_constant<int, 0>{}, yes, yes, w_full, yes);
  chunk(5, integral_constant<int, 1>{}, yes, no, w_full, yes);    // the planes of chunk 7 are the last
  chunk(6, integral_constant<int, 2>{}, no, no, integral_constant<int, NPF>{}, yes);
  chunk(7, integral_constant<int, 3>{}, no, no, integral_constant<int, 0>{}, yes);
  if ((ko & 64) && lane == 0 && wave < 4) a.dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + wave * 16 + 8] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {   // chunk 7's accumulators close the group's sum; the groups in order, from zero (the narrower kernels' slab sum)
      s[t][mt] += sg[t][mt] + accs[1][t][mt];
      sg[t][mt] = (f32x4)(0.f);
    }
  cg += 8;
  }
  if (KB > 1) {   // split-K across workgroups: sc1 slab stores from the registers + ticket, the last arriver sums the slabs in order (gemm16.h)
    __shared__ int flag;
    constexpr int U = 4 * H * PT * MT;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, 0x7ffffff0, 0x00020000);
    const unsigned tile_b = (unsigned)((wave * PT * MT) * 1024 + lane * 16);
    const unsigned slab_off = (unsigned)(((size_t)bx * KB + kid) * (U * 256) * sizeof(float)) + tile_b;
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s[t][mt]), rs, slab_off + (unsigned)((t * MT + mt) * 1024), 0, /*sc1*/ 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int tk = __hip_atomic_fetch_add(tickets + bx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = tk == KB - 1;
      if (last) __hip_atomic_store(tickets + bx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag = last;
    }
    __syncthreads();
    if (!flag) return;
    const unsigned base_off = (unsigned)((size_t)bx * KB * (U * 256) * sizeof(float)) + tile_b;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      f32x4 v[MT][8];   // KB <= 8 (launcher)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
          v[mt][kb] = kb < KB ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + (unsigned)(kb * U * 1024 + (t * MT + mt) * 1024), 0, /*sc1*/ 16))
                              : (f32x4)(0.f);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 sum = (f32x4)(0.f);
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) sum += v[mt][kb];
#pragma unroll
        for (int kb = 8; kb < 16; ++kb) sum += 0.f;   // the narrower kernels add sixteen slots (-0 -> +0 in the first already; kept for the record)
        s[t][mt] = sum;
      }
    }
  }

  if ((ko & 64) && lane == 0 && wave < 4) a.dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + wave * 16 + 11] = (uint32_t)__builtin_amdgcn_s_memrealtime();
  // ---- epilogue from the registers: lane (m, g) of accumulator (t, mt) holds batch row mt * 16 + m, weight rows g * 4 .. g * 4 + 3 ----
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int n0 = ((wp * PT + t) * 16 + g * 4);
    f32x4 ws = (f32x4)(1.f);
    if (a.wscale && n0 < a.N) ws = *reinterpret_cast<const f32x4*>(a.wscale + n0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int mm = (mt0 + mtw + mt) * 16 + m;
      const bool live = mm < M && n0 < a.N;
      f32x4 pv = s[t][mt];
      const float rsc = (PRO == PRO_NORM) ? stat[(mtw + mt) * 16 + m] : 1.f;
      if (a.wscale) { pv[0] *= ws[0]; pv[1] *= ws[1]; pv[2] *= ws[2]; pv[3] *= ws[3]; }
      pv[0] *= rsc; pv[1] *= rsc; pv[2] *= rsc; pv[3] *= rsc;
      if (EPI == EPI_RESID) {
        const f32x4 xn = rq[t][mt] + pv;
        float ssq = 0.f;
        if (live) {
          *reinterpret_cast<f32x4*>(a.out + (size_t)mm * a.ldo + n0) = xn;
          if (a.oplanes) {
            f32x4 xt;
            xt[0] = xn[0] * lq[t][0]; xt[1] = xn[1] * lq[t][1]; xt[2] = xn[2] * lq[t][2]; xt[3] = xn[3] * lq[t][3];
            const size_t ops = (size_t)a.N * 16;
            store_planes4(a.oplanes + (size_t)(mt0 + mtw + mt) * 3 * ops, ops, n0, m, xt, ONE);
            ssq = (xn[0] * xn[0] + xn[1] * xn[1]) + (xn[2] * xn[2] + xn[3] * xn[3]);
          }
        }
        if (a.oplanes && a.oss) {   // per-tile sums of x_new^2 for the consumer's RMS scale: the four row groups in gemm32.h's order
          const float q0 = __shfl(ssq, m), q1 = __shfl(ssq, m + 16), q2 = __shfl(ssq, m + 32), q3 = __shfl(ssq, m + 48);
          if (lane < 16 && live) a.oss[(size_t)mm * a.oss_ld + (n0 >> 4)] = (q0 + q1) + (q2 + q3);
        }
      } else {   // SwiGLU: (gate, up) pairs
        if (live) {
          const float h0 = (pv[0] / (1.f + __expf(-pv[0]))) * pv[1];
          const float h1 = (pv[2] / (1.f + __expf(-pv[2]))) * pv[3];
          if (a.oplanes) {
            const size_t ops = (size_t)(a.N >> 1) * 16;
            store_planes2(a.oplanes + (size_t)(mt0 + mtw + mt) * 3 * ops, ops, n0 >> 1, m, h0, h1, ONE);
          } else {
            *reinterpret_cast<f32x2*>(a.out + (size_t)mm * a.ldo + (n0 >> 1)) = f32x2{h0, h1};
          }
        }
      }
    }
  }
  if ((ko & 64) && lane == 0 && wave < 4) a.dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + wave * 16 + 9] = (uint32_t)__builtin_amdgcn_s_memrealtime();
  if ((ko & 64) && lane == 0 && wave < 4) a.dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64 + wave * 16 + 13] = (uint32_t)__builtin_amdgcn_s_memtime();
  if (a.bump_a && pid == 0 && blockIdx.z == 0 && tid == 0) {
    *a.bump_a += 1;
    if (a.bump_b) *a.bump_b += 1;
  }
}
#endif  // CSM_ARGS_ONLY

// 65..128 rows, FFN launches on planes (gate/up: norm + SwiGLU; down_proj / o_proj: plain + residual); -2 when the shape is not covered
int gemm128_configure_all();   // dynamic-LDS attributes, once per engine, outside capture
int launch_gemm128(hipStream_t st, int wdtype, int M, int pro, int epi, const GemvArgs& a, float* slabs, size_t slab_floats,
                   int* tickets, int n_tickets);
