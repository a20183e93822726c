// Weight-streaming skinny GEMM ("GEMV") for decode: y[M,N] = f(x)[M,K] @ W[N,K]^T, M <= 4 per launch.
//
// Roofline: HBM.  Algorithmic bytes = N*K*sizeof(WT) (each weight read exactly once per launch); x is
// M*K*4 bytes re-read per block from L2.  A *task* is a PAIR of output rows so that the fused epilogues
// that need two outputs (RoPE pair i / i+hd/2, SwiGLU gate/up) stay wave-local.  KS waves of a block
// cooperate on one task (K split KS ways) so that even a 1024-row matrix spreads over >= 256 workgroups;
// 64 lanes x 16 B = 1 KiB per load instruction, up to 8 loads in flight per wave, issued BEFORE the
// x/RMSNorm/attention prologue so the weight latency overlaps it; the next task's loads are issued before
// the current task's reduction.  No LDS round trip for the weights (cdna_hip_programming.md "GEMV / M<=16
// decode weights" row).
//
// Replaces (reference call sites): q/k/v/o_proj, gate/up/down_proj inside transformers.LlamaModel as
// called from modeling_csm.py:345-354,545-552,568-576; codebook0_head (:361), projection (:542),
// audio_head matmul (:557); RMSNorm (transformers modeling_llama.py:62-67) as prologue; RoPE
// (modeling_llama.py:130-160) + DynamicCache append (cache_utils.py:144-145) as the QKV epilogue.
// (SDPA as the prologue of the o_proj launch -- PRO_ATTN, round 1 -- measured 4 % slower per frame than the separate
// launch and was removed in round 4; the B = 1 decoder runs attention + o_proj as attn_oproj_kernel.)
#pragma once
#include <type_traits>
#include "common.h"
#include "prefetch.h"
#include "sample_wave.h"
#ifndef CSM_ARGS_ONLY
#include "attn_tile.h"
#endif

enum { PRO_PLAIN = 0, PRO_NORM = 1, PRO_COMBINE = 2, PRO_TOKNORM = 3, PRO_SAMPLE = 4 };   // PRO_COMBINE: gemv1_combine_kernel (the B = 1 backbone o_proj)
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_QKV = 3, EPI_ARGMAX = 4 };

struct GemvArgs {
  const void* W;
  const float* wscale;  // per-output-row scale (fp8 weights), nullable
  int N, K;
  const float* x;  // [M][ldx]
  int ldx;
  const float* ln;  // PRO_NORM weight [K]
  float eps;
  float* out;  // EPI_STORE/RESID: [M][ldo] indexed by output row; EPI_SWIGLU: [M][ldo] indexed by pair
  int ldo;
  // EPI_QKV
  int n_q, n_kv, hd;
  float qscale;
  const float* cos_tab;  // [pos][hd/2]
  const float* sin_tab;
  const int* pos_ptr;    // device scalar (backbone decode: current length), or
  int pos_const;         // constant position (decoder pass), used when pos_ptr == nullptr
  const int* row_pos;    // optional per-row positions (overrides both)
  const int* row_seq;    // optional row -> sequence slot (default: seq_base + row index)
  int seq_base;
  float* qbuf;           // [M][n_q*hd]
  void* kcache;          // [B][n_kv][hd/4][lmax][4]
  void* vcache;          // [B][n_kv][lmax][hd]
  int lmax;
  // end-of-kernel counter bumps (thread 0 of block 0), used by the head kernel to advance the
  // device-side frame index / backbone length inside a graph
  int* bump_a;
  int* bump_b;
  int nt;  // non-temporal weight loads
  int prio;  // 1 = s_setprio 3 at kernel entry: issue priority over the weight streamer's resident waves (round 5)
  const void* Wt;  // MFMA kernel only: the same weights in 16-row x 32-k fragment order (tile16_kernel), nullable
  // MFMA kernel only -- activations handed from launch to launch as ready-made B operands ("planes"): the exact
  // 3-way bf16 split of x (* the consumer's norm weight), in fragment order [3][K/128][4][64 lanes][8], plus per-tile
  // partial sums of x^2 [16][xss_ld] from which the consumer derives the RMS scale in its epilogue.
  const bf16_t* xplanes;   // input planes (nullable: then x is read as fp32 and normed / split in the kernel)
  const float* xss;        // input partial sums of squares (PRO_NORM with planes)
  int xss_n, xss_ld;
  bf16_t* oplanes;         // output planes for the next launch (EPI_RESID / EPI_SWIGLU), nullable
  const float* oln;        // norm weight of the consumer folded into the output planes (nullable = 1)
  float* oss;              // output partial sums of squares [16][oss_ld], one column per 16-row tile (EPI_RESID)
  int oss_ld;
  int pl1;                 // decode_precision = bf16: planes in and out are ONE plane of nearest-even bf16 values (common.h store_planes)
  // ---- fused greedy sampling (B == 1): EPI_ARGMAX writes one (max value, row index) pair per task instead of the
  // logits; PRO_TOKNORM (the next pass's first QKV launch) reduces the pairs to the token, takes its input row
  // from the projected-embedding table and records the token -- replacing sample_kernel for codebooks 1..30.
  float2* am_out;          // EPI_ARGMAX: [ceil((N - am_from) / 2)] partials
  int am_from;             // rows below am_from are stored normally (EPI_STORE semantics)
  const float2* am_in;     // PRO_TOKNORM: partials of the previous head launch
  int am_n;
  const float* tok_table;  // [C*V][K] fp32 projected audio embeddings
  int tok_row_base;        // cb * V
  const int64_t* tok_forced;  // teacher-forced tokens [B][max_frames][C], nullable
  int64_t* tok_ring;          // generated-frame ring  [B][max_frames][C]
  const int* tok_frame_ptr;
  int tok_max_frames, tok_C, tok_cb;
  float* tok_x_out;        // [K] residual stream of the new pass (written by workgroup 0)
  // ---- fused top-k sampling (B == 1, round 5): PRO_SAMPLE is PRO_TOKNORM with the token drawn by every wave of the launch
  // from the previous head launch's logits (sample_wave.h) instead of reduced from arg-max pairs
  WaveSampleArgs smp;      // logits / V / temperature / topk / rng / noise / cb (frame is read from tok_frame_ptr)
  const int* smp_row_done; // nullable: per-row stop flag of row 0 (a finished row emits token 0)
  float store_div;         // EPI_STORE: != 0 -> the stored value is v / store_div (the head launch in front of a PRO_SAMPLE launch divides
                           // its logits by the temperature: the same IEEE division sample_kernel does, once per logit instead of in every wave)
  // ---- PRO_COMBINE (B == 1 backbone o_proj): x is the split-KV attention output, merged here from the per-split partials
  const float* cmb_part;   // [K / 64 heads][cmb_ns][64 + 4]: acc[64], m, l (attn.h: attn_decode_*_kernel with nsplit > 1)
  int cmb_ns;              // splits (<= 8)
  int configure_only;  // host-side: only set the kernel's dynamic-LDS attribute, do not launch
  int force_generic;   // host-side: skip the M == 1 register fast path (A/B measurements)
  int no_mfma;         // host-side: rows >= 2 stay on the fp32-FMA kernel (two-token decoder pass: a row's arithmetic is then
                       // bitwise that of the single-row kernel)
  int v2_tasks;        // host-side: 1 = force one task per wave in the fast path (A/B measurements)
  int norm_ks;         // host-side: 2 = single-row normed launches with K = 2048 take the register path with two waves per task (K split in the workgroup)
  int grid_cap;        // host-side: max workgroups of the generic kernel (0 = 1024)
  int g16_nw, g16_kb, g16_pt;  // host-side: override waves / K-splits / panel tiles of the MFMA kernel (0 = auto)
  int xcdmap;                  // gemm128.h gate/up launches: panel p on XCD p / (panels / 8), i.e. the XCD that reads those 1 024 h columns as a k group of the down_proj launch (kfast).
                               // (The same map in gemm16.h / gemm32.h: 16 rows 5.21 -> 5.33 ms, 4 rows 5.01 -> 5.12, 32 / 64 rows nothing -- not used there.)
  int kfast;                   // gemm16.h / gemm32.h with K split across workgroups: the k split as the fastest grid index (planes of a k slice in one XCD's L2)
  int g128_shape;              // host-side A/B override of gemm128.hip's shape choice (0 = auto): low nibble = weight tiles per wave, bit 4 = split K across workgroups
  int g16_slab;  // bits 4-7: TIMING-ONLY knock-outs of the -DCSM_G16_KO variant build (gemm16.h); the split-K slab exchange is write-through (sc1) stores + sc1 loads
  // weight streamer (prefetch.h): launches-started counter bumped by workgroup 0 (nullable), and a host-side slot
  // the launcher fills with this launch's workgroup -> rows geometry
  unsigned* prog;
  PfGeom* geom_out;

  // debug (tools/streamer_probe.py): per workgroup {XCD id, shader clocks from kernel entry until the weights were
  // consumed} of this launch, [grid][2] uint32; nullable
  uint32_t* dbg;
};

// ---- kernel-argument preload (round 6) --------------------------------------------------------------------------------------
// A kernel whose arguments are ONE struct by value starts every wave with an s_load of the kernarg segment: a scalar-cache miss
// to memory (each launch of a replayed graph has its own kernarg block) that stands in front of the first weight load of every
// launch of the chain.  gfx950's dispatcher can instead initialise up to 14 SGPRs from the first 14 dwords of the kernarg segment
// ("kernarg preload", -mllvm -amdgpu-kernarg-preload-count: build.py) -- but only for leading SCALAR parameters, not for struct
// members.  The hot kernels of the decode chain therefore take what they need to issue their first loads as 14 leading dwords
// (5 pointers + 4 ints) in front of the struct and overwrite the struct's copies with them; everything else is still read from the
// struct, behind the loads.  tools/ubench/kernarg_preload.hip: 3.32 -> 3.15 us per dependent 4 MB launch (-0.16 us, x 614 launches
// of the B = 1 frame-step); profiles/r06_kernarg_preload.md.
//   p4 = out (the residual is prefetched first of all) -- EPI_QKV: row_pos if there is one, else pos_ptr (the position load is the first
//        of the launch; a select on a flag would not do: the compiler turns `flag ? a.row_pos : nullptr` into an s_load + s_cselect)
//   flags: bit 0 nt, 1 prio, 3 p4 is row_pos; bits 8-15 hd, 16-23 n_q, 24-31 n_kv
#define GEMV_HOT_PARAMS const void* hW, const float* hx, const float* hln, unsigned* hprog, void* hp4, int hN, int hK, uint32_t hflags, int hi3
#define GEMV_HOT_ARGS(a, PRO_, EPI_)                                                                                                  \
  (a).W,                                                                                                                              \
  ((PRO_) == PRO_TOKNORM ? reinterpret_cast<const float*>((a).am_in) : ((PRO_) == PRO_COMBINE ? (a).cmb_part : (a).x)),              \
  (a).ln, (a).prog,                                                                                                                   \
  ((PRO_) == PRO_TOKNORM ? (void*)const_cast<float*>((a).tok_table + (size_t)(a).tok_row_base * (size_t)(a).K)                       \
                         : ((EPI_) == EPI_QKV ? (void*)((a).row_pos ? (a).row_pos : (a).pos_ptr) : (void*)(a).out)),                  \
  (a).N, (a).K,                                                                                                                       \
  (uint32_t)(((a).nt ? 1u : 0u) | ((a).prio ? 2u : 0u) | ((a).row_pos ? 8u : 0u) |                                                    \
             (((uint32_t)(a).hd & 255u) << 8) | (((uint32_t)(a).n_q & 255u) << 16) | (((uint32_t)(a).n_kv & 255u) << 24)),             \
  ((PRO_) == PRO_TOKNORM ? (a).am_n : ((PRO_) == PRO_COMBINE ? (a).cmb_ns : (a).pos_const))
// (hd, n_q, n_kv <= 255 is checked by the launchers)
// PRO_TOKNORM (the decoder's token-indirect QKV launch; its position is a constant, pos_ptr / row_pos are null): the launch's critical path
// is pairs -> token -> table row, so the pair array travels in the x slot, the table (already offset to the codebook's rows) in the p4
// slot and the pair count in the last int; pos_const stays in the struct (used behind the weight loads).  PRO_COMBINE: the split-KV
// partials in the x slot, their count in the last int.
#define GEMV_HOT_TAKE(a, PRO_, EPI_)                                                              \
  do {                                                                                            \
    (a).W = hW; (a).ln = hln; (a).prog = hprog; (a).N = hN; (a).K = hK;                           \
    (a).nt = (int)(hflags & 1u); (a).prio = (int)((hflags >> 1) & 1u);                            \
    if ((PRO_) == PRO_TOKNORM) {                                                                  \
      (a).am_in = reinterpret_cast<const float2*>(hx); (a).am_n = hi3;                            \
      (a).tok_table = reinterpret_cast<const float*>(hp4); (a).tok_row_base = 0;                  \
      (a).row_pos = nullptr; (a).pos_ptr = nullptr;                                               \
    } else {                                                                                      \
      if ((PRO_) == PRO_COMBINE) { (a).cmb_part = hx; (a).cmb_ns = hi3; } else (a).x = hx;        \
      if ((EPI_) == EPI_QKV) {                                                                    \
        (a).pos_const = hi3;                                                                      \
        if (hflags & 8u) { (a).row_pos = (const int*)hp4; (a).pos_ptr = nullptr; }                \
        else { (a).row_pos = nullptr; (a).pos_ptr = (const int*)hp4; }                            \
      } else {                                                                                    \
        (a).out = (float*)hp4;                                                                    \
      }                                                                                           \
    }                                                                                             \
    if ((EPI_) == EPI_QKV) {                                                                      \
      (a).hd = (int)((hflags >> 8) & 255u); (a).n_q = (int)((hflags >> 16) & 255u); (a).n_kv = (int)(hflags >> 24); \
    }                                                                                             \
  } while (0)

#ifndef CSM_ARGS_ONLY
template <typename KT>
__device__ __forceinline__ size_t k_index(int b, int j, int d, int t, int n_kv, int hd, int lmax) {
  return ((((size_t)b * n_kv + j) * (hd >> 2) + (d >> 2)) * lmax + t) * 4 + (d & 3);
}
__device__ __forceinline__ size_t v_index(int b, int j, int t, int d, int n_kv, int hd, int lmax) {
  return (((size_t)b * n_kv + j) * lmax + t) * hd + d;
}

struct GemvTask {
  int task, r0, r1, head, hi;
  bool live, has1;
};

template <int EPI>
__device__ __forceinline__ GemvTask gemv_map_task(const GemvArgs& a, int t, int ntask) {
  GemvTask k;
  k.task = t;
  k.live = t < ntask;
  k.head = 0;
  k.hi = 0;
  if (EPI == EPI_QKV) {
    const int half = a.hd >> 1;
    k.head = t / half;
    k.hi = t - k.head * half;
    if (k.head < a.n_q + a.n_kv) { k.r0 = k.head * a.hd + k.hi; k.r1 = k.r0 + half; }
    else { k.r0 = k.head * a.hd + 2 * k.hi; k.r1 = k.r0 + 1; }
  } else {
    k.r0 = 2 * t;
    k.r1 = k.r0 + 1;
  }
  k.has1 = k.live && k.r1 < a.N;
  return k;
}

// Epilogue of one task.  Everything the epilogue must READ (the residual values for EPI_RESID; the row
// position and its cos/sin for EPI_QKV) is requested by `prefetch` when the task's weight loads are issued,
// so no memory round trip is left at the tail of the kernel.
template <typename KT, int EPI, int M>
struct GemvEpi {
  float a0[M], a1[M];
  float sc0, sc1;  // row scales of the two outputs (fp8 weights), 1 otherwise
  int pos[M];
  // row scales of fp8 weights: requested BEHIND the weight loads (round 6) -- they are consumed last, and their pointer is the one early
  // argument that does not fit the 14 preloaded dwords: in front of the weights it put the kernarg read back on the launch's critical path
  __device__ __forceinline__ void prefetch_scale(const GemvArgs& a, const GemvTask& k) {
    if (a.wscale && k.live) {
      sc0 = a.wscale[k.r0];
      if (k.has1) sc1 = a.wscale[k.r1];
    }
  }
  __device__ __forceinline__ void prefetch(const GemvArgs& a, const GemvTask& k) {
    sc0 = sc1 = 1.f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      a0[m] = a1[m] = 0.f;
      pos[m] = 0;
      if (!k.live) continue;
      if (EPI == EPI_RESID) {
        a0[m] = a.out[(size_t)m * a.ldo + k.r0];
        if (k.has1) a1[m] = a.out[(size_t)m * a.ldo + k.r1];
      } else if (EPI == EPI_QKV) {
        pos[m] = row_position(a.row_pos, m, a.pos_ptr, a.pos_const);   // consumed by prefetch_late
      }
    }
  }
  // EPI_QKV: cos/sin of the row position.  Issued BEHIND the weight loads: their address depends on a loaded position
  // (backbone: the device-resident length), and vmcnt retires in issue order -- requested first, that dependent
  // round trip stood in front of every other load of the launch; requested last, it hides under the weight stream.
  __device__ __forceinline__ void prefetch_late(const GemvArgs& a, const GemvTask& k) {
    if (EPI != EPI_QKV || !k.live || k.head >= a.n_q + a.n_kv) return;
    const int half = a.hd >> 1;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      a0[m] = a.cos_tab[(size_t)pos[m] * half + k.hi];
      a1[m] = a.sin_tab[(size_t)pos[m] * half + k.hi];
    }
  }
  // one lane writes the two outputs of the task for batch row m
  __device__ __forceinline__ void store(const GemvArgs& a, const GemvTask& k, int m, float v0, float v1) const {
    v0 *= sc0;
    v1 *= sc1;
    if (EPI == EPI_STORE) {
      if (a.store_div != 0.f) { v0 = v0 / a.store_div; v1 = v1 / a.store_div; }
      a.out[(size_t)m * a.ldo + k.r0] = v0;
      if (k.has1) a.out[(size_t)m * a.ldo + k.r1] = v1;
    } else if (EPI == EPI_RESID) {
      a.out[(size_t)m * a.ldo + k.r0] = a0[m] + v0;
      if (k.has1) a.out[(size_t)m * a.ldo + k.r1] = a1[m] + v1;
    } else if (EPI == EPI_SWIGLU) {
      a.out[(size_t)m * a.ldo + k.task] = (v0 / (1.f + __expf(-v0))) * v1;
    } else if (EPI == EPI_ARGMAX) {  // M == 1
      if (k.r0 < a.am_from) {
        a.out[k.r0] = v0;
        if (k.has1) a.out[k.r1] = v1;
      } else {
        const int i0 = k.r0 - a.am_from;
        float bv = v0;
        int bi = i0;
        if (k.has1 && v1 > v0) { bv = v1; bi = i0 + 1; }   // tie -> lower index
        a.am_out[i0 >> 1] = make_float2(bv, __int_as_float(bi));
      }
    } else {  // EPI_QKV
      const int half = a.hd >> 1;
      const int b = a.row_seq ? a.row_seq[m] : a.seq_base + m;
      KT* kc = reinterpret_cast<KT*>(a.kcache);
      KT* vc = reinterpret_cast<KT*>(a.vcache);
      if (k.head < a.n_q + a.n_kv) {
        const float c = a0[m], s = a1[m];
        const float o0 = __fmaf_rn(v0, c, -__fmul_rn(v1, s));   // explicit contraction: the form of every RoPE site (misc.h rope_scatter_kernel)
        const float o1 = __fmaf_rn(v1, c, __fmul_rn(v0, s));
        if (k.head < a.n_q) {
          float* q = a.qbuf + (size_t)m * a.n_q * a.hd + k.head * a.hd;
          q[k.hi] = __fmul_rn(o0, a.qscale);
          q[k.hi + half] = __fmul_rn(o1, a.qscale);
        } else {
          const int j = k.head - a.n_q;
          store_kv(kc + k_index<KT>(b, j, k.hi, pos[m], a.n_kv, a.hd, a.lmax), o0);
          store_kv(kc + k_index<KT>(b, j, k.hi + half, pos[m], a.n_kv, a.hd, a.lmax), o1);
        }
      } else {
        const int j = k.head - a.n_q - a.n_kv;
        store_kv(vc + v_index(b, j, pos[m], 2 * k.hi, a.n_kv, a.hd, a.lmax), v0);
        store_kv(vc + v_index(b, j, pos[m], 2 * k.hi + 1, a.n_kv, a.hd, a.lmax), v1);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// M = 1 fast path ("v2"): no LDS staging and no barrier for KS == 1.  One wave = one task (pair of rows);
// the wave's 2*U weight loads, its x slice and the norm weights are all requested before anything is
// consumed, so the launch costs ONE memory round trip.  Every wave recomputes the RMS statistic from its
// own registers (K floats -- cheaper than a workgroup barrier).  Grid = all tasks (one per wave), so a
// matrix of <= 64 KiB per CU is entirely in flight at once.  K must equal 512*U*KS.
// ---------------------------------------------------------------------------------------------------
// amdgpu_waves_per_eu(1, 4): without it the compiler aims at 8 waves per SIMD (64 VGPRs) and, to get there, SINKS
// loads below the RMS prologue (the last norm-weight loads were issued one by one behind the statistic: three extra
// dependent L2 round trips per normed launch).  These launches run at <= 4 waves per SIMD anyway (192 .. 1024
// workgroups on 256 CUs), so registers are free: every load of a wave is issued before anything is consumed.
// M = rows per launch: 1 (every single-sequence launch) or 2 (the two-token first decoder pass: both rows share every
// weight register; a row's arithmetic -- pair accumulators, ascending chunks, wave_sum2 -- is exactly the M = 1 form's).
template <typename WT, typename KT, int PRO, int EPI, int U, int KS, int T, int M = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4))) void gemv1_kernel(GEMV_HOT_PARAMS, GemvArgs a) {
  GEMV_HOT_TAKE(a, PRO, EPI);
  if constexpr (!std::is_same<WT, fp8_t>::value) a.wscale = nullptr;   // only fp8 weights carry row scales: no kernarg read in front of the loads
  static_assert(M == 1 || (PRO != PRO_TOKNORM && PRO != PRO_SAMPLE && EPI != EPI_ARGMAX), "fused sampling is single-row");
  constexpr bool TOK = PRO == PRO_TOKNORM || PRO == PRO_SAMPLE;   // the input row comes from the projected-embedding table
  __shared__ float part[4][2 * T * M];
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];   // PRO_SAMPLE: the sampler's histogram / candidate / survivor arrays
  constexpr int TPB = 4 / KS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TL_BEGIN(a.dbg);
  if (a.prio) __builtin_amdgcn_s_setprio(3);
#ifdef CSM_PROBE   // tools/streamer_probe.py builds libcsm_hip_probe.so with -DCSM_PROBE; even a disabled probe costs 6 % per frame
  const unsigned long long dbg_t0 = __builtin_readcyclecounter();   // before the first kernel argument is read
#endif
  if (a.prog && blockIdx.x == 0 && tid == 0) atomicAdd(a.prog, 1u);   // weight streamer pacing: this launch has started
  const int kw = wave % KS, tw = wave / KS;
  const int K = a.K;
  const int ntask = (EPI == EPI_QKV) ? (a.N >> 1) : ((a.N + 1) >> 1);
  const WT* W = reinterpret_cast<const WT*>(a.W);
  const int e0 = kw * (U * 512) + lane * 8;  // first element of this lane's chunk 0; chunk u at + u*512
  GemvTask k[T];
  GemvEpi<KT, EPI, M> epi[T];
  W8<WT> w0[T][U], w1[T][U];
  // vmcnt retires in issue order: the x slice and the norm weights (L2 hits, consumed first by the RMS
  // prologue) are requested ahead of the weight stream so the prologue runs while the weights are in flight.
  // PRO_TOKNORM cannot: its x row address depends on the argmax below, so there the weights go first.
  // The epilogue prefetch goes first of all: at M = 1 the residual values requested behind the weights arrive too
  // late (dec o_proj 3.13 vs 2.94 us); EPI_QKV requests only its position here and the cos/sin row behind the weights.
  f32x4 xa[M][U], xb[M][U], la[U], lb[U];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    k[t] = gemv_map_task<EPI>(a, (blockIdx.x * TPB + tw) * T + t, ntask);
    if (kw == 0) epi[t].prefetch(a, k[t]);
  }
  // PRO_TOKNORM: the head launch's per-task (value, index) pairs are requested FIRST of all (round 6): vmcnt retires in issue order, so
  // pairs requested behind the weights (rounds 2-5) could only be consumed once every weight of the wave had landed -- the launch's
  // critical chain pairs -> token -> table row started a whole weight flight late.  The pair array's address is a preloaded argument.
  constexpr int NP = 17;  // up to 1088 pairs (V = 2051 -> 1026)
  float2 pv[PRO == PRO_TOKNORM ? NP : 1];
  if (PRO == PRO_TOKNORM) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {   // unconditional loads (a slot beyond am_n re-reads the last pair and is masked below): no branch per load
      const int idx = lane + 64 * j;
      pv[j] = a.am_in[idx < a.am_n ? idx : a.am_n - 1];
    }
  }
  if (!TOK) {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xa[m][u] = *reinterpret_cast<const f32x4*>(a.x + (size_t)m * a.ldx + e0 + u * 512);
        xb[m][u] = *reinterpret_cast<const f32x4*>(a.x + (size_t)m * a.ldx + e0 + u * 512 + 4);
      }
  }
  if (PRO == PRO_NORM || TOK) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      la[u] = *reinterpret_cast<const f32x4*>(a.ln + e0 + u * 512);
      lb[u] = *reinterpret_cast<const f32x4*>(a.ln + e0 + u * 512 + 4);
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {  // T tasks per wave share one x slice; every weight load is issued up front
    const WT* w0p = W + (size_t)(k[t].live ? k[t].r0 : 0) * K + e0;
    const WT* w1p = W + (size_t)(k[t].has1 ? k[t].r1 : (k[t].live ? k[t].r0 : 0)) * K + e0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (a.nt) { w0[t][u].load_nt(w0p + u * 512); w1[t][u].load_nt(w1p + u * 512); }
      else { w0[t][u].load(w0p + u * 512); w1[t][u].load(w1p + u * 512); }
    }
  }
  if (kw == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t) epi[t].prefetch_scale(a, k[t]);
  }
  if (EPI == EPI_QKV && kw == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t) epi[t].prefetch_late(a, k[t]);
  }
  // (The machine scheduler sinks half of these loads below the RMS prologue to stay at 63 VGPRs = 8 waves per
  // SIMD; pinning them here with __builtin_amdgcn_sched_barrier(0) costs 69 VGPRs and measured the same: with
  // every wave of the launch resident at once the other waves cover.)
  int tok_bi = 0;
  if (PRO == PRO_TOKNORM) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if (lane + 64 * j >= a.am_n) pv[j] = make_float2(-INFINITY, __int_as_float(0x7fffffff));
      const int ci = __float_as_int(pv[j].y);
      if (pv[j].x > bv || (pv[j].x == bv && ci < bi)) { bv = pv[j].x; bi = ci; }
    }
    wave_argmax(bv, bi);
    tok_bi = bi;
  }
  if (PRO == PRO_SAMPLE) {
    WaveSampleArgs sa = a.smp;
    sa.frame = *a.tok_frame_ptr;
    tok_bi = wg_sample_topk(sa, dyn_lds, tid);
    if (a.smp_row_done && *a.smp_row_done) tok_bi = 0;   // per-row stop: a finished row stays silent
  }
  if (TOK) {
    const int f = *a.tok_frame_ptr;
    const size_t slot = (size_t)f * a.tok_C + a.tok_cb;   // B == 1: row 0
    int64_t feed = tok_bi;
    if (a.tok_forced) feed = a.tok_forced[slot];
    if (blockIdx.x == 0 && tid == 0) a.tok_ring[slot] = tok_bi;
    const float* xsrc = a.tok_table + ((size_t)feed + (size_t)a.tok_row_base) * K;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xa[0][u] = *reinterpret_cast<const f32x4*>(xsrc + e0 + u * 512);
      xb[0][u] = *reinterpret_cast<const f32x4*>(xsrc + e0 + u * 512 + 4);
    }
  }
  if (TOK && blockIdx.x == 0 && wave == 0) {   // the new pass's residual stream
#pragma unroll
    for (int u = 0; u < U; ++u) {
      *reinterpret_cast<f32x4*>(a.tok_x_out + e0 + u * 512) = xa[0][u];
      *reinterpret_cast<f32x4*>(a.tok_x_out + e0 + u * 512 + 4) = xb[0][u];
    }
  }
  // The arithmetic below is written on float pairs (v_pk_mul_f32 / v_pk_fma_f32): the gate/up launch is VALU-issue
  // bound once its weights arrive (SQ counters: 52 % of wave time stalled at issue), so instructions per weight matter.
  f32x2 xp[M][U][4];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xp[m][u][0] = f32x2{xa[m][u][0], xa[m][u][1]};
      xp[m][u][1] = f32x2{xa[m][u][2], xa[m][u][3]};
      xp[m][u][2] = f32x2{xb[m][u][0], xb[m][u][1]};
      xp[m][u][3] = f32x2{xb[m][u][2], xb[m][u][3]};
    }
  if (PRO == PRO_NORM || TOK) {
    // KS == 1: the wave holds the whole row.  KS > 1 (round 3: the backbone's K = 2048 normed launches on this register path
    // with two waves per task): every wave sums its K slice and the KS waves of a task meet through LDS (LDS-only barrier:
    // the weight loads stay in flight) -- a fixed order, so the statistic is the same in every wave of the task.
    __shared__ float ssx[4][M];
    float ssw[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      f32x2 ss2 = f32x2{0.f, 0.f};
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) ss2 = PKFMA(xp[m][u][i], xp[m][u][i], ss2);
      ssw[m] = wave_sum(ss2[0] + ss2[1]);
      if (KS > 1 && lane == 0) ssx[wave][m] = ssw[m];
    }
    if (KS > 1) {
      lds_barrier();
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) t += ssx[tw * KS + s][m];
        ssw[m] = t;
      }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      // mean = sum * rcp(K): identical to sum / K for the power-of-two widths of this model family; raw v_rsq_f32
      // (the argument is >= eps, far from the denormal range the library wrapper rescales for)
      const float sc = __builtin_amdgcn_rsqf(ssw[m] * __builtin_amdgcn_rcpf((float)K) + a.eps);
      const f32x2 sc2 = f32x2{sc, sc};
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xp[m][u][0] = (xp[m][u][0] * sc2) * f32x2{la[u][0], la[u][1]};
        xp[m][u][1] = (xp[m][u][1] * sc2) * f32x2{la[u][2], la[u][3]};
        xp[m][u][2] = (xp[m][u][2] * sc2) * f32x2{lb[u][0], lb[u][1]};
        xp[m][u][3] = (xp[m][u][3] * sc2) * f32x2{lb[u][2], lb[u][3]};
      }
    }
  }
#ifdef CSM_PROBE
  // T1: the activations (and, for normed launches, the RMS statistic) are in registers
  const float probe_x = xp[0][0][0][0] + xp[0][U - 1][3][1];
  asm volatile("" :: "v"(probe_x));
  const unsigned long long dbg_t1 = a.dbg ? __builtin_readcyclecounter() : 0ull;
#endif
  float s0[T][M], s1[T][M];
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      f32x2 c0 = f32x2{0.f, 0.f}, c1 = f32x2{0.f, 0.f};
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          c0 = PKFMA(w0[t][u].pair(i), xp[m][u][i], c0);
          c1 = PKFMA(w1[t][u].pair(i), xp[m][u][i], c1);
        }
      }
      s0[t][m] = c0[0] + c0[1];
      s1[t][m] = c1[0] + c1[1];
      wave_sum2(s0[t][m], s1[t][m]);
    }
  }
#ifdef CSM_PROBE   // tools/streamer_probe.py builds libcsm_hip_probe.so with -DCSM_PROBE; even a disabled probe costs 6 % per frame
  if (a.dbg && tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.dbg[2 * blockIdx.x] = (xcc & 15u) | ((uint32_t)(dbg_t1 - dbg_t0) << 8);   // bits 8..: clocks until the activations arrived
    a.dbg[2 * blockIdx.x + 1] = (uint32_t)(__builtin_readcyclecounter() - dbg_t0);
  }
#endif
  if (KS > 1) {
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int m = 0; m < M; ++m) { part[wave][2 * (t * M + m)] = s0[t][m]; part[wave][2 * (t * M + m) + 1] = s1[t][m]; }
    }
    __syncthreads();
    if (kw == 0 && lane == 0) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
          for (int s = 1; s < KS; ++s) { s0[t][m] += part[wave + s][2 * (t * M + m)]; s1[t][m] += part[wave + s][2 * (t * M + m) + 1]; }
    }
  }
  if (lane == 0 && kw == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t)
      if (k[t].live) {
#pragma unroll
        for (int m = 0; m < M; ++m) epi[t].store(a, k[t], m, s0[t][m], s1[t][m]);
      }
  }
  if (a.bump_a && blockIdx.x == 0 && tid == 0) {
    *a.bump_a += 1;
    if (a.bump_b) *a.bump_b += 1;
  }
  TL_END(0x10 + EPI + 8 * PRO);
}

// ---------------------------------------------------------------------------------------------------
// B = 1 backbone o_proj with the split-KV merge as its prologue (round 5): out[n] += W[n, :] . att, where
// att[h][d] = sum_s e^{m_s - M} acc_s[h][d] / sum_s e^{m_s - M} l_s is never written to memory.  Replaces the
// attn_combine launch + the plain o_proj GEMV (one launch boundary and one cold activation round trip less per
// backbone layer).  K == 2048 == 32 heads x 64: wave w of a workgroup merges heads 8w .. 8w + 7 (a lane: 8 dims of one
// head, <= 8 splits -- the partial loads are issued with the weight loads, in front of them: they are consumed first),
// the four quarters meet in LDS behind an LDS-only barrier (the weight loads stay in flight), then every wave multiplies
// its row pair exactly like gemv1_kernel<PRO_PLAIN, EPI_RESID, U = 4> (pair accumulators, ascending chunks, wave_sum2).
// ---------------------------------------------------------------------------------------------------
template <typename WT, int SMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemv1_combine_kernel(GEMV_HOT_PARAMS, GemvArgs a) {
  GEMV_HOT_TAKE(a, PRO_COMBINE, EPI_RESID);
  if constexpr (!std::is_same<WT, fp8_t>::value) a.wscale = nullptr;
  constexpr int U = 4, HD = 64, K = 2048;
  __shared__ __attribute__((aligned(16))) float xs[K];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TL_BEGIN(a.dbg);
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  if (a.prog && blockIdx.x == 0 && tid == 0) atomicAdd(a.prog, 1u);   // weight streamer pacing: this launch has started
  const int ntask = (a.N + 1) >> 1;
  const GemvTask k = gemv_map_task<EPI_RESID>(a, blockIdx.x * 4 + wave, ntask);
  GemvEpi<float, EPI_RESID, 1> epi;
  epi.prefetch(a, k);
  const int S = a.cmb_ns;
  const int h = wave * 8 + (lane >> 3), d0 = (lane & 7) * 8;
  const float* pp = a.cmb_part + (size_t)h * S * (HD + 4);
  f32x4 pa[SMAX], pb[SMAX];
  f32x2 st[SMAX];
  // every load is unconditional (a split beyond cmb_ns re-reads the last one and gets weight 0 below): with the loads inside
  // `if (s < S)` the compiler put a vmcnt wait behind each split's loads -- eight dependent round trips, 8.6 us per launch
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    const int sc = s < S ? s : S - 1;
    pa[s] = *reinterpret_cast<const f32x4*>(pp + sc * (HD + 4) + d0);
    pb[s] = *reinterpret_cast<const f32x4*>(pp + sc * (HD + 4) + d0 + 4);
    st[s] = *reinterpret_cast<const f32x2*>(pp + sc * (HD + 4) + HD);
  }
  const WT* W = reinterpret_cast<const WT*>(a.W);
  const int e0 = lane * 8;
  W8<WT> w0[U], w1[U];
  {
    const WT* w0p = W + (size_t)(k.live ? k.r0 : 0) * K + e0;
    const WT* w1p = W + (size_t)(k.has1 ? k.r1 : (k.live ? k.r0 : 0)) * K + e0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (a.nt) { w0[u].load_nt(w0p + u * 512); w1[u].load_nt(w1p + u * 512); }
      else { w0[u].load(w0p + u * 512); w1[u].load(w1p + u * 512); }
    }
  }
  epi.prefetch_scale(a, k);
  {
    float Mx = st[0][0];
#pragma unroll
    for (int s = 1; s < SMAX; ++s) Mx = fmaxf(Mx, st[s][0]);   // (a repeated split does not change the max)
    float L = 0.f;
    f32x4 xa = (f32x4)(0.f), xb = (f32x4)(0.f);
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      const float al = (s >= S || st[s][0] == -INFINITY) ? 0.f : __expf(st[s][0] - Mx);
      L = fmaf(st[s][1], al, L);
#pragma unroll
      for (int c = 0; c < 4; ++c) { xa[c] = fmaf(al, pa[s][c], xa[c]); xb[c] = fmaf(al, pb[s][c], xb[c]); }
    }
    const float inv = 1.f / L;
    *reinterpret_cast<f32x4*>(xs + wave * 512 + lane * 8) = xa * inv;
    *reinterpret_cast<f32x4*>(xs + wave * 512 + lane * 8 + 4) = xb * inv;
  }
  lds_barrier();
  f32x2 c0 = f32x2{0.f, 0.f}, c1 = f32x2{0.f, 0.f};
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + e0 + u * 512), xb = *reinterpret_cast<const f32x4*>(xs + e0 + u * 512 + 4);
    const f32x2 xp[4] = {f32x2{xa[0], xa[1]}, f32x2{xa[2], xa[3]}, f32x2{xb[0], xb[1]}, f32x2{xb[2], xb[3]}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      c0 = PKFMA(w0[u].pair(i), xp[i], c0);
      c1 = PKFMA(w1[u].pair(i), xp[i], c1);
    }
  }
  float s0 = c0[0] + c0[1], s1 = c1[0] + c1[1];
  wave_sum2(s0, s1);
  if (lane == 0 && k.live) epi.store(a, k, 0, s0, s1);
  TL_END(0x10 + EPI_RESID + 8 * PRO_COMBINE);
}

template <typename WT, typename KT, int M, int PRO, int EPI, int KS>
__global__ __launch_bounds__(256) void gemv_kernel(GEMV_HOT_PARAMS, GemvArgs a) {
  GEMV_HOT_TAKE(a, PRO, EPI);
  if constexpr (!std::is_same<WT, fp8_t>::value) a.wscale = nullptr;
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [M][K] | red[M][4] | part[4][2M]
  constexpr int U = 4;
  constexpr int TPB = 4 / KS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TL_BEGIN(a.dbg);
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  if (a.prog && blockIdx.x == 0 && tid == 0) atomicAdd(a.prog, 1u);   // weight streamer pacing: this launch has started
  const int kw = wave % KS, tw = wave / KS;
  const int K = a.K;
  float* red = xs + (size_t)M * K;
  float* part = red + M * 4;
  const WT* W = reinterpret_cast<const WT*>(a.W);
  const int nch_w = (K >> 3) / KS;        // chunks (of 8 weights) per wave-slice
  const int ch0 = kw * nch_w;
  const int half = a.hd >> 1;
  const int ntask = (EPI == EPI_QKV) ? (a.N >> 1) : ((a.N + 1) >> 1);
  const int stride = gridDim.x * TPB;
  const int iters = (ntask - (int)blockIdx.x * TPB + stride - 1) / stride;  // block-uniform

  const int task = blockIdx.x * TPB + tw;
  // two register sets: the loads of task t+1 are in flight while task t is consumed
  W8<WT> wa0[U], wa1[U], wb0[U], wb1[U];
  auto issue = [&](const GemvTask& t, W8<WT> (&w0)[U], W8<WT> (&w1)[U], int cb) {
    const WT* w0p = W + (size_t)t.r0 * K;
    const WT* w1p = W + (size_t)(t.has1 ? t.r1 : t.r0) * K;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = cb + u * 64 + lane;
      if (t.live && c < nch_w) {
        if (a.nt) {
          w0[u].load_nt(w0p + (size_t)(ch0 + c) * 8);
          w1[u].load_nt(w1p + (size_t)(ch0 + c) * 8);
        } else {
          w0[u].load(w0p + (size_t)(ch0 + c) * 8);
          w1[u].load(w1p + (size_t)(ch0 + c) * 8);
        }
      } else {
        w0[u].zero();
        w1[u].zero();
      }
    }
  };
  GemvTask ta = gemv_map_task<EPI>(a, task, ntask);
  GemvTask tb = gemv_map_task<EPI>(a, task + stride, ntask);
  GemvEpi<KT, EPI, M> ea, eb;
  // Issue order = consumption order (vmcnt retires in order): epilogue inputs (EPI_QKV: a dependent chain),
  // then the first 2048 columns of x and of the norm weight (L2 hits, staged by the prologue), then the two
  // weight register sets, which stay in flight across the prologue's LDS-only barriers.
  if (kw == 0) {
    ea.prefetch(a, ta);
    ea.prefetch_scale(a, ta);
    if (iters > 1) { eb.prefetch(a, tb); eb.prefetch_scale(a, tb); }
  }
  constexpr int XR = 2;  // x column blocks of 1024 held in registers (K <= 2048: every normed input of the model)
  f32x4 xv[M][XR], lnw[XR];
  {
#pragma unroll
    for (int r = 0; r < XR; ++r) {
      const int kk = tid * 4 + r * 1024;
      lnw[r] = (f32x4)(1.f);
      if (PRO == PRO_NORM && kk < K) lnw[r] = *reinterpret_cast<const f32x4*>(a.ln + kk);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        xv[m][r] = (f32x4)(0.f);
        if (kk < K) xv[m][r] = *reinterpret_cast<const f32x4*>(a.x + (size_t)m * a.ldx + kk);
      }
    }
  }
  issue(ta, wa0, wa1, 0);  // in flight while the prologue runs
  if (iters > 1) issue(tb, wb0, wb1, 0);
  if (kw == 0) {
    ea.prefetch_late(a, ta);
    if (iters > 1) eb.prefetch_late(a, tb);
  }

  // ---- prologue: stage x into LDS (plain | RMS-normalised | short-cache attention output) -----------
  if (PRO == PRO_NORM && K <= XR * 1024) {
    // normed inputs: statistic and scaling entirely from registers, one LDS-only barrier for the wave exchange
    float ss[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      ss[m] = 0.f;
#pragma unroll
      for (int r = 0; r < XR; ++r) {
        const f32x4 v = xv[m][r];
        ss[m] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      }
      const float s = wave_sum(ss[m]);
      if (lane == 0) red[m * 4 + wave] = s;
    }
    lds_barrier();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const float s = red[m * 4 + 0] + red[m * 4 + 1] + red[m * 4 + 2] + red[m * 4 + 3];
      const float sc = rsqrtf(s / (float)K + a.eps);
#pragma unroll
      for (int r = 0; r < XR; ++r) {
        const int kk = tid * 4 + r * 1024;
        f32x4 v = xv[m][r];
        v[0] = (v[0] * sc) * lnw[r][0];
        v[1] = (v[1] * sc) * lnw[r][1];
        v[2] = (v[2] * sc) * lnw[r][2];
        v[3] = (v[3] * sc) * lnw[r][3];
        if (kk < K) *reinterpret_cast<f32x4*>(xs + (size_t)m * K + kk) = v;
      }
    }
    lds_barrier();
  } else {
    float ss[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      ss[m] = 0.f;
      const float* xr = a.x + (size_t)m * a.ldx;
#pragma unroll
      for (int r = 0; r < XR; ++r) {
        const int kk = tid * 4 + r * 1024;
        const f32x4 v = xv[m][r];
        if (kk < K) *reinterpret_cast<f32x4*>(xs + (size_t)m * K + kk) = v;
        if (PRO == PRO_NORM) ss[m] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      }
      for (int k = tid * 4 + XR * 1024; k < K; k += 1024) {   // K > 2048 (down_proj): these wait behind the weights
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
        *reinterpret_cast<f32x4*>(xs + (size_t)m * K + k) = v;
        if (PRO == PRO_NORM) ss[m] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      }
    }
    if (PRO == PRO_NORM) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float s = wave_sum(ss[m]);
        if (lane == 0) red[m * 4 + wave] = s;
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float s = red[m * 4 + 0] + red[m * 4 + 1] + red[m * 4 + 2] + red[m * 4 + 3];
        float sc = rsqrtf(s / (float)K + a.eps);
        for (int k = tid * 4; k < K; k += 1024) {
          f32x4 v = *reinterpret_cast<f32x4*>(xs + (size_t)m * K + k);
          const f32x4 w = *reinterpret_cast<const f32x4*>(a.ln + k);
          v[0] = (v[0] * sc) * w[0];
          v[1] = (v[1] * sc) * w[1];
          v[2] = (v[2] * sc) * w[2];
          v[3] = (v[3] * sc) * w[3];
          *reinterpret_cast<f32x4*>(xs + (size_t)m * K + k) = v;
        }
      }
    }
    if (K <= XR * 1024) lds_barrier();
    else __syncthreads();
  }

  // consume one task from register set (w0, w1); re-arm the set with task `nxt` (if `more`) before reducing
  auto process = [&](GemvTask& t, GemvEpi<KT, EPI, M>& ep, W8<WT> (&w0)[U], W8<WT> (&w1)[U], bool more) {
    // same arithmetic as gemv1_kernel (pair accumulators, ascending chunks, wave_sum2), so a row's result is
    // bit-identical whichever of the two fp32-FMA kernels -- i.e. whichever batch size -- computes it
    f32x2 c0[M], c1[M];
#pragma unroll
    for (int m = 0; m < M; ++m) c0[m] = c1[m] = f32x2{0.f, 0.f};
    for (int cb = 0; cb < nch_w; cb += 64 * U) {
      if (cb > 0) issue(t, w0, w1, cb);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int c = cb + u * 64 + lane;
        c = ch0 + (c < nch_w ? c : nch_w - 1);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + (size_t)m * K + c * 8);
          const f32x4 xb = *reinterpret_cast<const f32x4*>(xs + (size_t)m * K + c * 8 + 4);
          const f32x2 x0 = f32x2{xa[0], xa[1]}, x1 = f32x2{xa[2], xa[3]}, x2 = f32x2{xb[0], xb[1]}, x3 = f32x2{xb[2], xb[3]};
          c0[m] = PKFMA(w0[u].pair(0), x0, c0[m]);
          c1[m] = PKFMA(w1[u].pair(0), x0, c1[m]);
          c0[m] = PKFMA(w0[u].pair(1), x1, c0[m]);
          c1[m] = PKFMA(w1[u].pair(1), x1, c1[m]);
          c0[m] = PKFMA(w0[u].pair(2), x2, c0[m]);
          c1[m] = PKFMA(w1[u].pair(2), x2, c1[m]);
          c0[m] = PKFMA(w0[u].pair(3), x3, c0[m]);
          c1[m] = PKFMA(w1[u].pair(3), x3, c1[m]);
        }
      }
    }
    float acc0[M], acc1[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      acc0[m] = c0[m][0] + c0[m][1];
      acc1[m] = c1[m][0] + c1[m][1];
    }
    const GemvTask cur = t;
    const GemvEpi<KT, EPI, M> cep = ep;
    t = gemv_map_task<EPI>(a, cur.task + 2 * stride, ntask);
    if (more) {
      issue(t, w0, w1, 0);
      if (kw == 0) { ep.prefetch(a, t); ep.prefetch_scale(a, t); ep.prefetch_late(a, t); }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) wave_sum2(acc0[m], acc1[m]);
    if (KS > 1) {
      if (lane == 0) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          part[wave * 2 * M + 2 * m] = acc0[m];
          part[wave * 2 * M + 2 * m + 1] = acc1[m];
        }
      }
      lds_barrier();  // LDS-only: the next task's weight loads (issued above) stay in flight
      if (kw == 0 && lane == 0) {
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
          for (int s = 1; s < KS; ++s) {
            acc0[m] += part[(wave + s) * 2 * M + 2 * m];
            acc1[m] += part[(wave + s) * 2 * M + 2 * m + 1];
          }
      }
      lds_barrier();
    }
    if (lane == 0 && kw == 0 && cur.live) {
#pragma unroll
      for (int m = 0; m < M; ++m) cep.store(a, cur, m, acc0[m], acc1[m]);
    }
  };
  for (int it = 0; it < iters; it += 2) {
    process(ta, ea, wa0, wa1, it + 2 < iters);
    if (it + 1 < iters) process(tb, eb, wb0, wb1, it + 3 < iters);
  }
  if (a.bump_a && blockIdx.x == 0 && tid == 0) {
    *a.bump_a += 1;
    if (a.bump_b) *a.bump_b += 1;
  }
  TL_END(0x40 + EPI + 8 * PRO);
}

#endif  // CSM_ARGS_ONLY
// host-side launcher (defined in gemv.hip)
int launch_gemv(hipStream_t st, int wdtype, int kvdtype, int M, int pro, int epi, const GemvArgs& a);
