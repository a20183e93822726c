// Decoder attention + o_proj in ONE launch for single-sequence decode (M = 1, cache of <= 32 positions).
//
// Replaces, per decoder layer and pass, the pair (attn_decode_kernel, o_proj GEMV) -- two dependent launches whose cost
// is a launch boundary plus a cold activation round trip each, for 2 MB of weights and a few KB of cache (the reference
// call sites: sdpa_attention_forward + LlamaAttention.o_proj, modeling_llama.py:254-281).  One workgroup owns a few output
// rows (8 for csm-1b); its n_q waves compute the n_q heads' attention outputs IN PARALLEL (exactly the one-wave tile code of
// attn_decode_kernel; redundant across the workgroups, a few KB from L2 each) while the workgroup's slice of the
// o_proj matrix, requested at kernel start, is in flight; then every thread multiplies its 8- or 16-wide k piece and the
// lanes of a row meet by DPP / permlane swaps.  The attention output never goes to memory and nothing crosses workgroups.
// (A first version split o_proj over heads -- workgroup (head, row slice), partial products to a slab, a ticket electing
// the last arriver to add them: 7.6 us per launch against 4.8 + 4.9 for the pair it replaced, and 0.23 ms per frame
// SLOWER in the graph: the release / ticket / acquire tail costs what a launch boundary costs.  The older `PRO_ATTN`
// prologue ran the heads one after the other inside every o_proj workgroup and lost 4 %.)
// Roofline: launch latency (2 MB of weights per launch); what it buys is one launch boundary per decoder layer-pass:
// B = 1 frame-step 3.28 -> 3.16 ms.  The launch is left out of the weight streamer's schedule (prefetch.h): with its
// 2 MB streamed like the o_proj GEMV's were, the step measured 3.19 ms.
#pragma once
#include <type_traits>
#include "attn_tile.h"

struct AttnOprojArgs {
  const float* q;        // [n_q * hd] of the row (pre-scaled, RoPE applied: the QKV launch's output)
  const void* kcache;    // sequence 0
  const void* vcache;
  int n_q, n_kv, hd, lmax;
  const int* pos_ptr;    // device scalar position or
  int pos_const;         // constant (decoder pass index)
  const void* W;         // o_proj [N][n_q * hd]
  const float* wscale;   // per-row scale (fp8 weights), nullable
  int N;
  float* out;            // residual stream [N], updated in place (batched form: [rows][ldo])
  int beside_streamer;   // host-side: refuse (-2) when a workgroup of this launch does not fit beside a resident streamer wave
  int dbg_onekey;        // TIMING ONLY (wrong results): every lane loads key 0 -- the launch without its K/V traffic
  uint32_t* dbg;         // timeline probe slot (common.h TL_BEGIN), nullable
  int prio;              // 1 = s_setprio 3 at kernel entry
  int gqa;               // host-side: 1 = the key-split form below (attn_oproj_gqa_kernel) where the shape allows it
};
// (A batched form of this fusion -- one workgroup per (64-output slice, batch row) -- was measured SLOWER than the stand-alone
//  attention + matrix-core o_proj pair at B = 16 (5.58 vs 5.09 ms per step: every workgroup pulls its row's K/V tiles once per
//  head) and removed in round 4; numbers in profiles/r03_b16_step_timeline.md and DESIGN.md's appendix.)

// kernel-argument preload (gemv.h GEMV_HOT_PARAMS has the why): the 14 leading dwords are everything the kernels need to request
// their weights, the residual, q and the K / V tiles -- 6 pointers, the cache pitch and a packed word (bit 0 prio, bits 8.. pos_const)
#define AO_HOT_PARAMS const void* hW, float* hout, const int* hpos_ptr, const void* hk, const void* hv, const float* hq, int hlmax, uint32_t hpk
#define AO_HOT_ARGS(a) (a).W, (a).out, (a).pos_ptr, (a).kcache, (a).vcache, (a).q, (a).lmax, (uint32_t)(((a).prio ? 1u : 0u) | ((a).pos_ptr ? 0u : ((uint32_t)(a).pos_const << 8)))
#define AO_HOT_TAKE(a)                                                                                              \
  do {                                                                                                              \
    (a).W = hW; (a).out = hout; (a).pos_ptr = hpos_ptr; (a).kcache = hk; (a).vcache = hv; (a).q = hq; (a).lmax = hlmax; \
    (a).prio = (int)(hpk & 1u); (a).pos_const = (int)(hpk >> 8);                                                    \
  } while (0)

#ifdef CSM_ATTN_OPROJ_KERNEL
// blockDim = 64 n_q (n_q in {2, 4, 8}: the K/V tile alone is 128 registers).  A thread multiplies KPT = max(8, K / 64)
// consecutive k of one output row, so a row takes K / KPT lanes (a whole wave for csm-1b's decoder: 8 rows per
// workgroup, grid = N / 8 = 128 workgroups -- the 2 MB of weights must be pulled by many CUs at once: a CU draws only
// ~11 B/clk from HBM, and the 32-rows-per-workgroup form of this kernel, 64 KB on each of 32 CUs, took 12 us).
template <typename KT, typename WT, int HD>
__global__ __launch_bounds__(512) void attn_oproj_kernel(AO_HOT_PARAMS, AttnOprojArgs a) {
  AO_HOT_TAKE(a);
  if constexpr (!std::is_same<WT, fp8_t>::value) a.wscale = nullptr;
  using Tile = AttnTile32<KT, HD>;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // q[n_q][HD] | att[n_q][HD] | p[n_q][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TL_BEGIN(a.dbg);
  const int nq = a.n_q, K = nq * HD;
  const int kpt = K >= 1024 ? 16 : 8, tpr = K / kpt;   // k per thread, threads per output row (32 or 64)
  float* qs = lds + wave * HD;
  float* att = lds + nq * HD;
  float* pb = lds + 2 * nq * HD + wave * 32;
  const int rloc = tid / tpr, part = tid - rloc * tpr;
  const int n = blockIdx.x * (64 * nq / tpr) + rloc;
  // everything this workgroup reads is requested before anything is consumed: weights, residual, q, the K/V tiles
  const WT* wp = reinterpret_cast<const WT*>(a.W) + (size_t)n * K + part * kpt;
  W8<WT> w0, w1;
  w0.load(wp);
  w1.zero();
  if (kpt == 16) w1.load(wp + 8);
  const float ws = a.wscale ? a.wscale[n] : 1.f;
  float resid = 0.f;
  if (part == 0) resid = a.out[n];
  {
    const int h = wave;
    const int pos = row_position(nullptr, 0, a.pos_ptr, a.pos_const);
    const int cnt = min(pos + 1, 32);
    const int j = h / (nq / a.n_kv);
    const KT* kc = reinterpret_cast<const KT*>(a.kcache) + (size_t)j * (size_t)(HD / 4) * a.lmax * 4;
    const KT* vc = reinterpret_cast<const KT*>(a.vcache) + (size_t)j * (size_t)a.lmax * HD;
    Tile tile;
    tile.load(kc, vc, a.lmax, 0, a.dbg_onekey ? 1 : cnt, lane);
    const float* qsrc = a.q + (size_t)h * HD;
#pragma unroll
    for (int i = 0; i < HD / 64; ++i) qs[lane + 64 * i] = qsrc[lane + 64 * i];
    __builtin_amdgcn_wave_barrier();
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc = (f32x4)(0.f);
    tile.accumulate(qs, pb, cnt, lane, m_run, l_run, acc);
    acc = Tile::reduce(acc);
    if (lane < Tile::LPR) *reinterpret_cast<f32x4*>(att + h * HD + 4 * lane) = acc * (1.f / l_run);
  }
  __syncthreads();
  const float* xp = att + part * kpt;
  const f32x4 x0 = *reinterpret_cast<const f32x4*>(xp), x1 = *reinterpret_cast<const f32x4*>(xp + 4);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s0 = fmaf(w0.get(e), x0[e], s0);
    s1 = fmaf(w0.get(4 + e), x1[e], s1);
  }
  if (kpt == 16) {
    const f32x4 x2 = *reinterpret_cast<const f32x4*>(xp + 8), x3 = *reinterpret_cast<const f32x4*>(xp + 12);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s0 = fmaf(w1.get(e), x2[e], s0);
      s1 = fmaf(w1.get(4 + e), x3[e], s1);
    }
  }
  float s = s0 + s1;
  // the lanes of a row (32: half a wave, 64: the wave): DPP inside the 16-lane rows, then the permlane swaps
  s += dpp_all<0xB1>(s);    // quad_perm [1,0,3,2]
  s += dpp_all<0x4E>(s);    // quad_perm [2,3,0,1]
  s += dpp_all<0x141>(s);   // row_half_mirror
  s += dpp_all<0x140>(s);   // row_mirror
  s = xor16_sum(s);
  if (tpr == 64) s = xor32_sum(s);
  if (part == 0) a.out[n] = resid + s * ws;
  TL_END(2);
}

// ---- round 5: the same launch with the K/V tiles SHARED by the query heads of a kv-head --------------------------------------
// In-step timeline (profiles/r05_b1_timeline.md): the launch above has a 4.1 us body against 2.0-2.1 us for a plain GEMV of the same
// size.  Every one of its 8 waves pulls its head's whole 32-key K and V tile (32 x 1 KiB wave-loads), the 4 query heads of a kv-head
// pulling the SAME tile: 272 vector-memory instructions per workgroup, 16 clocks each through the CU's one texture path = 2.1 us
// before any arithmetic -- the launch is bound by the CU's L1 request rate, not by bytes or latency.  Here a wave takes 8 KEYS of
// its kv-head for all G = 4 query heads: 4 + 4 tile loads per wave instead of 16 + 16 (the workgroup: 104 instead of 272 memory
// instructions), the same multiply-adds per wave, per-wave online-softmax partials (acc, m, l) merged through LDS by one wave
// per head.  Shape: head_dim 128, n_q = 8, n_kv = 2, cache <= 32 positions (csm-1b's decoder); other shapes keep the kernel above.
template <typename KT, typename WT>
__global__ __launch_bounds__(512) void attn_oproj_gqa_kernel(AO_HOT_PARAMS, AttnOprojArgs a) {
  AO_HOT_TAKE(a);
  if constexpr (!std::is_same<WT, fp8_t>::value) a.wscale = nullptr;
  constexpr int HD = 128, NQ = 8, G = 4, K = NQ * HD, KPW = 8;   // KPW = keys per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];   // q[8 waves][G * HD] | part[8 waves][G][HD] | att[NQ][HD] | stat[8][G][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TL_BEGIN(a.dbg);
  if (a.prio) __builtin_amdgcn_s_setprio(3);   // issue priority over the resident weight-streamer waves of the CU (B = 1 step -1.3 %)
  float* const qs = lds + wave * (G * HD);
  float* const partb = lds + 8 * G * HD;
  float* const att = partb + 8 * G * HD;
  float* const stat = att + NQ * HD;
  const int kvj = wave >> 2, wq = wave & 3;
  // o_proj: output row n = 8 * block + wave, this lane's 16 consecutive k
  const int n = blockIdx.x * 8 + wave;
  const WT* wp = reinterpret_cast<const WT*>(a.W) + (size_t)n * K + lane * 16;
  W8<WT> w0, w1;
  w0.load(wp);
  w1.load(wp + 8);
  float resid = 0.f;
  if (lane == 0) resid = a.out[n];
  const int pos = row_position(nullptr, 0, a.pos_ptr, a.pos_const);
  const int cnt = min(pos + 1, 32);
  const int t0 = KPW * wq;
  const int my = min(max(cnt - t0, 0), KPW);   // keys of this wave (wave-uniform)
  const int t = lane & 7, e = lane >> 3;        // QK^T: lane = (key t, 16-dim slice e)
  const int dg = lane & 31, tpar = lane >> 5;   // PV: lane = (4-dim group dg, key phase tpar)
  f32x4 k[4], v[4];
  if (my > 0) {
    const KT* kc = reinterpret_cast<const KT*>(a.kcache) + (size_t)kvj * (size_t)(HD / 4) * a.lmax * 4;
    const KT* vc = reinterpret_cast<const KT*>(a.vcache) + (size_t)kvj * (size_t)a.lmax * HD;
    const size_t tc = t0 + (t < my ? t : my - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] = ld_k4<KT>(kc + ((size_t)(e * 4 + i) * a.lmax + tc) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tt = tpar + 2 * i;
      const size_t tv = t0 + (tt < my ? tt : my - 1);
      v[i] = ld_k4<KT>(vc + tv * HD + 4 * dg);
    }
  }
  {   // the G query heads of this kv-head into the wave's own LDS strip (no workgroup barrier in front of the scores)
    const float* qsrc = a.q + (size_t)kvj * G * HD + lane * 8;
    const f32x4 qa = *reinterpret_cast<const f32x4*>(qsrc), qb = *reinterpret_cast<const f32x4*>(qsrc + 4);
    *reinterpret_cast<f32x4*>(qs + lane * 8) = qa;
    *reinterpret_cast<f32x4*>(qs + lane * 8 + 4) = qb;
  }
  const float ws = a.wscale ? a.wscale[n] : 1.f;   // fp8 row scale: behind every other request (its pointer is not among the preloaded arguments)
  __builtin_amdgcn_wave_barrier();
  float* const mypart = partb + wave * (G * HD);
  if (my > 0) {
    float m[G], l[G], p[G];
#pragma unroll
    for (int h = 0; h < G; ++h) {
      f32x2 sa = f32x2{0.f, 0.f}, sb = f32x2{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + h * HD + e * 16 + 4 * i);
        sa = PKFMA((f32x2{qv[0], qv[1]}), (f32x2{k[i][0], k[i][1]}), sa);
        sb = PKFMA((f32x2{qv[2], qv[3]}), (f32x2{k[i][2], k[i][3]}), sb);
      }
      float sc = (sa[0] + sa[1]) + (sb[0] + sb[1]);
      sc += dpp_all<0x128>(sc);   // row_ror:8 -- the other 8-lane half of the 16-lane row (slices e ^ 1)
      sc = xor16_sum(sc);
      sc = xor32_sum(sc);         // every lane of key t now holds the full 128-dim score
      const bool valid = t < my;
      float mx = valid ? sc : -INFINITY;
      mx = fmaxf(mx, dpp_all<0xB1>(mx));
      mx = fmaxf(mx, dpp_all<0x4E>(mx));
      mx = fmaxf(mx, dpp_all<0x141>(mx));   // max over the 8 keys (every 8-lane group holds the same 8 scores)
      const float pe = valid ? __expf(sc - mx) : 0.f;
      float su = pe;
      su += dpp_all<0xB1>(su);
      su += dpp_all<0x4E>(su);
      su += dpp_all<0x141>(su);
      m[h] = mx; l[h] = su; p[h] = pe;
    }
    f32x2 a01[G], a23[G];
#pragma unroll
    for (int h = 0; h < G; ++h) a01[h] = a23[h] = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int h = 0; h < G; ++h) {
        // p of keys 2i (lanes 0-31) and 2i + 1 (lanes 32-63): lanes 2i and 2i + 1 hold them (slice e = 0)
        const float p0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[h]), 2 * i));
        const float p1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[h]), 2 * i + 1));
        const float pv = tpar ? p1 : p0;
        const f32x2 p2 = f32x2{pv, pv};
        a01[h] = PKFMA(p2, (f32x2{v[i][0], v[i][1]}), a01[h]);
        a23[h] = PKFMA(p2, (f32x2{v[i][2], v[i][3]}), a23[h]);
      }
    }
#pragma unroll
    for (int h = 0; h < G; ++h) {
      f32x4 o;
      o[0] = xor32_sum(a01[h][0]); o[1] = xor32_sum(a01[h][1]); o[2] = xor32_sum(a23[h][0]); o[3] = xor32_sum(a23[h][1]);
      if (lane < 32) *reinterpret_cast<f32x4*>(mypart + h * HD + 4 * dg) = o;
      if (lane == 0) { stat[(wave * G + h) * 2] = m[h]; stat[(wave * G + h) * 2 + 1] = l[h]; }
    }
  } else {
    // no key of this wave is present: a zero partial with weight exp(-inf) = 0
#pragma unroll
    for (int h = 0; h < G; ++h) {
      if (lane < 32) *reinterpret_cast<f32x4*>(mypart + h * HD + 4 * dg) = (f32x4)(0.f);
      if (lane == 0) { stat[(wave * G + h) * 2] = -INFINITY; stat[(wave * G + h) * 2 + 1] = 0.f; }
    }
  }
  lds_barrier();   // LDS only: the o_proj weights stay in flight
  {   // merge: wave w owns query head w = (kv-head w / 4, local head w % 4); lane = dims 2 lane, 2 lane + 1
    const int g4 = (wave >> 2) * 4, hl = wave & 3;
    float mj[4], lj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { mj[j] = stat[((g4 + j) * G + hl) * 2]; lj[j] = stat[((g4 + j) * G + hl) * 2 + 1]; }
    const float M = fmaxf(fmaxf(mj[0], mj[1]), fmaxf(mj[2], mj[3]));
    float L = 0.f;
    f32x2 o = f32x2{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float al = mj[j] == -INFINITY ? 0.f : __expf(mj[j] - M);
      L = fmaf(lj[j], al, L);
      const f32x2 pj = *reinterpret_cast<const f32x2*>(partb + (g4 + j) * (G * HD) + hl * HD + 2 * lane);
      o = PKFMA((f32x2{al, al}), pj, o);
    }
    const float inv = 1.f / L;
    *reinterpret_cast<f32x2*>(att + wave * HD + 2 * lane) = o * f32x2{inv, inv};
  }
  lds_barrier();
  const float* xp = att + lane * 16;
  const f32x4 x0 = *reinterpret_cast<const f32x4*>(xp), x1 = *reinterpret_cast<const f32x4*>(xp + 4);
  const f32x4 x2 = *reinterpret_cast<const f32x4*>(xp + 8), x3 = *reinterpret_cast<const f32x4*>(xp + 12);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    s0 = fmaf(w0.get(c), x0[c], s0);
    s1 = fmaf(w0.get(4 + c), x1[c], s1);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    s0 = fmaf(w1.get(c), x2[c], s0);
    s1 = fmaf(w1.get(4 + c), x3[c], s1);
  }
  float s = s0 + s1;
  s += dpp_all<0xB1>(s);
  s += dpp_all<0x4E>(s);
  s += dpp_all<0x141>(s);
  s += dpp_all<0x140>(s);
  s = xor16_sum(s);
  s = xor32_sum(s);
  if (lane == 0) a.out[n] = resid + s * ws;
  TL_END(7);
}

#endif  // CSM_ATTN_OPROJ_KERNEL
// returns -2 (caller runs the two-launch form) when the shape is not covered (head_dim not 64 / 128, n_q not 2 / 4 / 8, N % 32, cache longer than 32 positions)
int launch_attn_oproj(hipStream_t st, int wdtype, int kvdtype, const AttnOprojArgs& a);
