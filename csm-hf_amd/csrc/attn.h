// KV-cached GQA attention for single-query rows (decode; also used row-by-row for prefill v1).
//
// Replaces: transformers sdpa_attention_forward as reached from LlamaAttention.forward
// (sdpa_attention.py:97-163, modeling_llama.py:254-281) for q_len == 1 (all cached keys visible)
// and, with per-row positions, the causal prefill (key t visible to query at position p iff
// kv_start[b] <= t <= p).  Softmax in fp32 (as SDPA), GQA: q-head h reads kv-head h / (n_q/n_kv).
//
// Cache layout (written by the QKV epilogue, gemv.h / rope_scatter):
//   K [B][n_kv][hd/4][lmax][4]  -- position-major inside a 4-dim group, so that "lane = key position"
//                                  loads are 16-byte-per-lane coalesced (QK^T phase)
//   V [B][n_kv][lmax][hd]       -- "lane = output dim" loads are coalesced (PV phase)
// Roofline: HBM/L2 (KV streaming): algorithmic bytes = kv_len * n_kv * hd * 2 * sizeof(KT) per row.
// One workgroup = (row, kv-head, split); its 4 waves take the G = n_q/n_kv query heads that share the
// kv-head, so each K/V byte is fetched from HBM once per workgroup and re-served by L1/L2.
// Wavefront shuffles carry the softmax reductions; no LDS except the broadcast copy of q.
#pragma once
#include "common.h"

struct AttnArgs {
  const float* q;  // [rows][n_q*hd], pre-scaled by hd^-0.5, RoPE applied
  const void* kcache;
  const void* vcache;
  int n_q, n_kv, hd, lmax;
  const int* row_seq;   // nullable: row -> sequence slot (default row)
  const int* row_pos;   // nullable: per-row query position
  const int* pos_ptr;   // device scalar position (backbone decode) or
  int pos_const;        // constant
  const int* kv_start;  // nullable: per-sequence first valid key (left padding)
  int nsplit;
  int one_wave;  // 1 = one 64-thread workgroup per (row, q-head, split): the tile loads of the G heads of a kv-head
                 // go through G texture units instead of one (the short decoder caches are issue-, not byte-bound)
  float* out;   // [rows][n_q*hd]                  (nsplit == 1)
  float* part;  // [rows][n_q][nsplit][hd+4]       (nsplit > 1): acc[hd], m, l, pad
  bf16_t* oplanes;  // nullable: the output also as MFMA B-operand planes for a batched o_proj (rows <= 16)
  int pl1;          // decode_precision = bf16: one nearest-even plane (common.h store_planes)
  int tile_prefetch;  // host-side: 1 = the kernel variant that requests tile i+1 before it consumes tile i (head_dim 64)
  int* tickets;       // nullable, [rows * n_q] zeroed once (self-resetting): with nsplit > 1 the LAST split of a (row, head) to arrive
                      // merges the partials itself -- no attn_combine launch (round 4: one launch less per backbone layer)
  int prio;           // 1 = s_setprio 3 at kernel entry (issue priority over the weight streamer's resident waves)
  int gqa;            // host-side: 1 = attn_decode_gqa_kernel (head_dim 64, n_q / n_kv = 4: the key quarters of a split on the workgroup's four waves)
  int no_combine;     // host-side: 1 = with nsplit > 1 leave the partials to the consumer (gemv.h gemv1_combine_kernel): no attn_combine launch
  uint32_t* dbg;      // timeline probe slots (common.h TL_BEGIN) of the attention launch and, 4096 words on, of the combine launch; nullable
};

// kernel-argument preload (gemv.h GEMV_HOT_PARAMS has the why): 6 pointers + the cache pitch + a packed word = the 14 dwords the
// kernels need to request their first K / V tile.  p_pos = row_pos if there is one, else pos_ptr.
//   packed: bit 0 prio, 1 p_pos is row_pos, 2 one_wave; bits 3-9 pos_const (0..127), 10-17 nsplit, 18-25 n_q, 26-31 n_kv
//   hlmax: bits 0-23 the cache pitch, bits 24-31 the rows of the launch (the one-wave form derives its grid from it: gridDim is a
//          hidden kernel argument, i.e. another s_load)
#define ATTN_HOT_PARAMS const float* hq, const void* hk, const void* hv, const int* hpos, const int* hkvs, const int* hseq, int hlmax, uint32_t hpk
#define ATTN_HOT_ARGS(a, rows_)                                                                                                           \
  (a).q, (a).kcache, (a).vcache, ((a).row_pos ? (a).row_pos : (a).pos_ptr), (a).kv_start, (a).row_seq, (int)((uint32_t)(a).lmax | ((uint32_t)((rows_) <= 255 ? (rows_) : 0) << 24)), \
  (uint32_t)(((a).prio ? 1u : 0u) | ((a).row_pos ? 2u : 0u) | ((a).one_wave ? 4u : 0u) |                                             \
             ((((a).row_pos || (a).pos_ptr) ? 0u : (uint32_t)(a).pos_const) << 3) | ((uint32_t)(a).nsplit << 10) | ((uint32_t)(a).n_q << 18) | ((uint32_t)(a).n_kv << 26))
#define ATTN_HOT_TAKE(a)                                                                                   \
  do {                                                                                                     \
    (a).q = hq; (a).kcache = hk; (a).vcache = hv; (a).kv_start = hkvs; (a).row_seq = hseq; (a).lmax = (int)((uint32_t)hlmax & 0xffffffu); \
    if (hpk & 2u) { (a).row_pos = hpos; (a).pos_ptr = nullptr; } else { (a).row_pos = nullptr; (a).pos_ptr = hpos; } \
    (a).prio = (int)(hpk & 1u); (a).one_wave = (int)((hpk >> 2) & 1u); (a).pos_const = (int)((hpk >> 3) & 127u);   \
    (a).nsplit = (int)((hpk >> 10) & 255u); (a).n_q = (int)((hpk >> 18) & 255u); (a).n_kv = (int)(hpk >> 26);      \
  } while (0)

#ifndef CSM_ARGS_ONLY
#include "attn_tile.h"

// HD = head_dim (64 or 128).  One workgroup = (row, kv-head, split); wave g handles query head j*G+g.
// PF = second register set: the next tile is in flight while the current one is consumed.
template <typename KT, int HD, bool PF = false>
__global__ __launch_bounds__(256) void attn_decode_kernel(ATTN_HOT_PARAMS, AttnArgs a) {
  ATTN_HOT_TAKE(a);
  using Tile = AttnTile32<KT, HD>;
  __shared__ __attribute__((aligned(16))) float qs[16 * HD];  // up to 16 q-heads per kv-head
  __shared__ float pb[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TL_BEGIN(a.dbg);
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  const int G = a.n_q / a.n_kv;
  int blk = blockIdx.x;
  int g0 = wave, gstep = 4;
  if (a.one_wave) {
    // the G one-wave workgroups of a (row, kv-head) read the same K / V tile: `per` workgroups apart in the grid, i.e. on ONE XCD when
    // rows x kv-heads is a multiple of 8 (workgroup id mod 8), so that three of the four reads hit that XCD's L2.  As the fastest index
    // (round 4) they sat on four XCDs: 19 MB fetched per decoder attention launch of a 128-row step for 4-8 MB of K / V
    const int r8 = (int)((uint32_t)hlmax >> 24);
    const int per = r8 ? r8 * a.n_kv * a.nsplit : (int)gridDim.x / G;   // = gridDim.x / G; launches of more than 255 rows (context rows) read the hidden argument
    g0 = blk / per;
    blk -= g0 * per;
    gstep = G;
  }
  const int sp = blk % a.nsplit;
  blk /= a.nsplit;
  const int j = blk % a.n_kv;
  const int row = blk / a.n_kv;
  const int b = a.row_seq ? a.row_seq[row] : row;
  const int pos = row_position(a.row_pos, row, a.pos_ptr, a.pos_const);
  const int t_lo0 = a.kv_start ? a.kv_start[b] : 0;
  const int len = pos + 1 - t_lo0;
  int span = (len + a.nsplit - 1) / a.nsplit;
  span = (span + 15) & ~15;
  const int t_lo = t_lo0 + sp * span;
  int t_hi = t_lo + span;
  if (t_hi > pos + 1) t_hi = pos + 1;

  const KT* kc = reinterpret_cast<const KT*>(a.kcache) + ((size_t)b * a.n_kv + j) * (size_t)(HD / 4) * a.lmax * 4;
  const KT* vc = reinterpret_cast<const KT*>(a.vcache) + ((size_t)b * a.n_kv + j) * (size_t)a.lmax * HD;
  Tile tile;
  if (t_lo < t_hi) tile.load(kc, vc, a.lmax, t_lo, min(32, t_hi - t_lo), lane);  // in flight during the q staging

  for (int g = g0; g < G; g += gstep) {
    // this wave's query head goes through a wave-private LDS strip (no workgroup barrier)
    const float* qsrc = a.q + (size_t)row * a.n_q * HD + (size_t)(j * G + g) * HD;
#pragma unroll
    for (int i = 0; i < HD / 64; ++i) qs[g * HD + lane + 64 * i] = qsrc[lane + 64 * i];
    __builtin_amdgcn_wave_barrier();
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc = (f32x4)(0.f);
    if constexpr (PF) {
      // the NEXT tile is requested before this one is consumed (second register set): a workgroup that walks several
      // tiles pays one memory latency, not one per tile (B = 16 backbone, 3 tiles per split: DESIGN.md section 5)
      if (g != g0 && t_lo < t_hi) tile.load(kc, vc, a.lmax, t_lo, min(32, t_hi - t_lo), lane);
      for (int t0 = t_lo; t0 < t_hi; t0 += 32) {
        const int cnt = min(32, t_hi - t0);
        Tile nxt;
        const bool more = t0 + 32 < t_hi;
        if (more) nxt.load(kc, vc, a.lmax, t0 + 32, min(32, t_hi - t0 - 32), lane);
        tile.accumulate(qs + g * HD, pb[wave], cnt, lane, m_run, l_run, acc);
        if (more) tile = nxt;
      }
    } else {
      for (int t0 = t_lo; t0 < t_hi; t0 += 32) {
        const int cnt = min(32, t_hi - t0);
        if (t0 != t_lo || g != g0) tile.load(kc, vc, a.lmax, t0, cnt, lane);
        tile.accumulate(qs + g * HD, pb[wave], cnt, lane, m_run, l_run, acc);
      }
    }
    acc = Tile::reduce(acc);
    const int h = j * G + g;
    if (a.nsplit == 1) {
      const float inv = 1.f / l_run;
      if (lane < Tile::LPR) {
        const f32x4 o = acc * inv;
        *reinterpret_cast<f32x4*>(a.out + (size_t)row * a.n_q * HD + (size_t)h * HD + 4 * lane) = o;
        if (a.oplanes) {   // rows 16..31: second plane group
          const size_t ps = (size_t)a.n_q * HD * 16;
          store_planes4(a.oplanes + (size_t)(row >> 4) * 3 * ps, ps, h * HD + 4 * lane, row & 15, o, a.pl1 != 0);
        }
      }
    } else {
      float* pp = a.part + (((size_t)row * a.n_q + h) * a.nsplit + sp) * (HD + 4);
      if (!a.tickets) {
        if (lane < Tile::LPR) *reinterpret_cast<f32x4*>(pp + 4 * lane) = acc;
        if (lane == 0) {
          pp[HD] = m_run;
          pp[HD + 1] = l_run;
        }
        continue;
      }
      // ---- fused merge: write-through (sc1) partial + ticket; the last arriver of (row, head) reads all partials back (sc1: the
      // other splits ran on other XCDs, each with its own L2) and does attn_combine_kernel's arithmetic, term for term
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, 0x7ffffff0, 0x00020000);
      const unsigned off0 = (unsigned)(((size_t)row * a.n_q + h) * a.nsplit * (HD + 4) * sizeof(float));
      const unsigned off = off0 + (unsigned)(sp * (HD + 4) * sizeof(float));
      if (lane < Tile::LPR) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), rs, off + 16 * lane, 0, /*sc1*/ 16);
      if (lane == 0) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m_run), rs, off + HD * 4, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l_run), rs, off + HD * 4 + 4, 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int last = 0;
      if (lane == 0) {
        int* tk = a.tickets + (size_t)row * a.n_q + h;
        last = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.nsplit - 1;
        if (last) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      last = __builtin_amdgcn_readfirstlane(last);
      if (!last) continue;
      constexpr int NS = 32;   // splits merged per register batch (B = 1 at a 512-frame context: 32)
      const float ms = lane < a.nsplit ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off0 + (unsigned)(lane * (HD + 4) + HD) * 4, 0, 16)) : -INFINITY;
      const float ls = lane < a.nsplit ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off0 + (unsigned)(lane * (HD + 4) + HD + 1) * 4, 0, 16)) : 0.f;
      float num[HD / 64];
#pragma unroll
      for (int i = 0; i < HD / 64; ++i) num[i] = 0.f;
      float w = 0.f, inv = 0.f;
      for (int sb = 0; sb < a.nsplit; sb += NS) {
        float pv[HD / 64][NS];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int i = 0; i < HD / 64; ++i)
            pv[i][s] = sb + s < a.nsplit ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off0 + (unsigned)((sb + s) * (HD + 4) + lane + 64 * i) * 4, 0, 16)) : 0.f;
        if (sb == 0) {
          const float M = wave_max(ms);
          w = (ms == -INFINITY) ? 0.f : __expf(ms - M);
          inv = 1.f / wave_sum(ls * w);
        }
#pragma unroll
        for (int i = 0; i < HD / 64; ++i)
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            // (lanes >= nsplit hold w = 0 and pv = 0: fmaf(0, 0, num) = num, exactly attn_combine_kernel's 64-term chain)
            const float ws = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), sb + s));
            num[i] = fmaf(ws, pv[i][s], num[i]);
          }
      }
#pragma unroll
      for (int i = 0; i < HD / 64; ++i) {
        const float o = num[i] * inv;
        a.out[((size_t)row * a.n_q + h) * HD + lane + 64 * i] = o;
        if (a.oplanes) {
          const size_t ps = (size_t)a.n_q * HD * 16;
          store_planes(a.oplanes + (size_t)(row >> 4) * 3 * ps, ps, h * HD + lane + 64 * i, row & 15, o, a.pl1 != 0);
        }
      }
    }
  }
  TL_END(3);
}

// ---- round 5: single-sequence backbone attention with the K/V tiles SHARED by the G = 4 query heads of a kv-head ------------
// attn_decode_kernel gives every query head of a kv-head its own wave, and each of those waves pulls the same K and V tile:
// the workgroup's vector-memory instruction count (16 clocks each through the CU's one texture path) is what bounds it once a
// workgroup walks more than one tile (profiles/r05_b1_timeline.md).  Here the 4 waves of a workgroup (row, kv-head, split) each
// take a QUARTER of the split's keys for ALL G query heads -- a quarter of the tile loads per workgroup -- and their per-wave
// online-softmax partials meet in LDS, one wave per head merging them.  That makes long splits affordable: 8 splits (64
// workgroups) at a 512-frame context instead of 32, which in turn lets the o_proj launch merge the split partials in its own
// prologue (gemv.h gemv1_combine_kernel) -- the attn_combine launch of the B = 1 backbone layer is gone.  head_dim 64, G = 4.
template <typename KT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void attn_decode_gqa_kernel(ATTN_HOT_PARAMS, AttnArgs a) {
  ATTN_HOT_TAKE(a);
  constexpr int HD = 64, G = 4;
  using Tile = AttnTile32<KT, HD>;
  __shared__ __attribute__((aligned(16))) float qs[4][G * HD];   // wave-private copies of the G query heads
  __shared__ float pb[4][32];
  __shared__ __attribute__((aligned(16))) float wpart[4][G][HD];
  __shared__ float wstat[4][G][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TL_BEGIN(a.dbg);
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  int blk = blockIdx.x;
  const int sp = blk % a.nsplit;
  blk /= a.nsplit;
  const int j = blk % a.n_kv;
  const int row = blk / a.n_kv;
  const int b = a.row_seq ? a.row_seq[row] : row;
  const int pos = row_position(a.row_pos, row, a.pos_ptr, a.pos_const);
  const int t_lo0 = a.kv_start ? a.kv_start[b] : 0;
  const int len = pos + 1 - t_lo0;
  int span = (len + a.nsplit - 1) / a.nsplit;
  span = (span + 15) & ~15;                       // a multiple of 16: the waves' quarters start on 64-byte boundaries of the K rows
  const int t_lo = t_lo0 + sp * span;
  const int t_hi = min(t_lo + span, pos + 1);
  const int wspan = span >> 2;
  const int w_lo = t_lo + wave * wspan;
  const int w_hi = min(w_lo + wspan, t_hi);
  const KT* kc = reinterpret_cast<const KT*>(a.kcache) + ((size_t)b * a.n_kv + j) * (size_t)(HD / 4) * a.lmax * 4;
  const KT* vc = reinterpret_cast<const KT*>(a.vcache) + ((size_t)b * a.n_kv + j) * (size_t)a.lmax * HD;
  Tile tile;
  if (w_lo < w_hi) tile.load(kc, vc, a.lmax, w_lo, min(32, w_hi - w_lo), lane);   // in flight during the q staging
  {
    const float* qsrc = a.q + (size_t)row * a.n_q * HD + (size_t)(j * G) * HD + lane * 4;
    *reinterpret_cast<f32x4*>(&qs[wave][lane * 4]) = *reinterpret_cast<const f32x4*>(qsrc);
  }
  __builtin_amdgcn_wave_barrier();
  float m_run[G], l_run[G];
  f32x4 acc[G];
#pragma unroll
  for (int h = 0; h < G; ++h) { m_run[h] = -INFINITY; l_run[h] = 0.f; acc[h] = (f32x4)(0.f); }
  for (int t0 = w_lo; t0 < w_hi; t0 += 32) {
    const int cnt = min(32, w_hi - t0);
    Tile nxt;
    const bool more = t0 + 32 < w_hi;
    if (more) nxt.load(kc, vc, a.lmax, t0 + 32, min(32, w_hi - t0 - 32), lane);   // the next tile is requested before this one is consumed
#pragma unroll
    for (int h = 0; h < G; ++h) tile.accumulate(&qs[wave][h * HD], pb[wave], cnt, lane, m_run[h], l_run[h], acc[h]);
    if (more) tile = nxt;
  }
#pragma unroll
  for (int h = 0; h < G; ++h) {
    const f32x4 o = Tile::reduce(acc[h]);
    if (lane < Tile::LPR) *reinterpret_cast<f32x4*>(&wpart[wave][h][4 * lane]) = o;
    if (lane == 0) { wstat[wave][h][0] = m_run[h]; wstat[wave][h][1] = l_run[h]; }
  }
  __syncthreads();
  {   // wave w merges query head j * G + w over the four key quarters; lane = output dim
    const int h = wave;
    float mj[4], lj[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { mj[q] = wstat[q][h][0]; lj[q] = wstat[q][h][1]; }
    const float M = fmaxf(fmaxf(mj[0], mj[1]), fmaxf(mj[2], mj[3]));
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float al = mj[q] == -INFINITY ? 0.f : __expf(mj[q] - M);
      L = fmaf(lj[q], al, L);
      o = fmaf(al, wpart[q][h][lane], o);
    }
    const int hq = j * G + h;
    if (a.nsplit == 1) {
      const float on = o * (1.f / L);
      a.out[(size_t)row * a.n_q * HD + (size_t)hq * HD + lane] = on;
      if (a.oplanes) {   // batched decode: the output also as MFMA B-operand planes for o_proj (rows 16.. : further plane groups)
        const size_t ps = (size_t)a.n_q * HD * 16;
        store_planes(a.oplanes + (size_t)(row >> 4) * 3 * ps, ps, hq * HD + lane, row & 15, on, a.pl1 != 0);
      }
    } else {
      float* pp = a.part + (((size_t)row * a.n_q + hq) * a.nsplit + sp) * (HD + 4);
      pp[lane] = o;
      if (lane == 0) { pp[HD] = M; pp[HD + 1] = L; }   // an empty split: M = -inf, L = 0, o = 0
    }
  }
  TL_END(8);
}

// merge the per-split partials: out[row][h][d] = sum_s acc_s e^{m_s-M} / sum_s l_s e^{m_s-M}
// one wave per (row, head); lane s owns split s (nsplit <= 64), then lane = output dim; all partial
// loads are issued up front (predicated full unroll).
template <int HD>
__global__ __launch_bounds__(64) void attn_combine_kernel(const float* part, int n_q, int nsplit, float* out, bf16_t* oplanes, int pl1, uint32_t* dbg) {
  const int rh = blockIdx.x, lane = threadIdx.x;
  TL_BEGIN(dbg);
  const float* pp = part + (size_t)rh * nsplit * (HD + 4);
  float pv[HD / 64][64];
#pragma unroll
  for (int s = 0; s < 64; ++s)
#pragma unroll
    for (int i = 0; i < HD / 64; ++i) pv[i][s] = s < nsplit ? pp[s * (HD + 4) + lane + 64 * i] : 0.f;
  const float ms = lane < nsplit ? pp[lane * (HD + 4) + HD] : -INFINITY;
  const float ls = lane < nsplit ? pp[lane * (HD + 4) + HD + 1] : 0.f;
  const float M = wave_max(ms);
  const float w = (ms == -INFINITY) ? 0.f : __expf(ms - M);
  const float inv = 1.f / wave_sum(ls * w);
#pragma unroll
  for (int i = 0; i < HD / 64; ++i) {
    float num = 0.f;
#pragma unroll
    for (int s = 0; s < 64; ++s) num = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), s)), pv[i][s], num);
    out[(size_t)rh * HD + lane + 64 * i] = num * inv;
    if (oplanes) {
      const size_t ps = (size_t)n_q * HD * 16;
      const int row = rh / n_q;
      store_planes(oplanes + (size_t)(row >> 4) * 3 * ps, ps, (rh % n_q) * HD + lane + 64 * i, row & 15, num * inv, pl1 != 0);
    }
  }
  TL_END(4);
}

#endif  // CSM_ARGS_ONLY
int launch_attn(hipStream_t st, int kvdtype, int rows, const AttnArgs& a);
