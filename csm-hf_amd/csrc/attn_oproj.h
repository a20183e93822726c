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
};
// (A batched form of this fusion -- one workgroup per (64-output slice, batch row) -- was measured SLOWER than the stand-alone
//  attention + matrix-core o_proj pair at B = 16 (5.58 vs 5.09 ms per step: every workgroup pulls its row's K/V tiles once per
//  head) and removed in round 4; numbers in profiles/r03_b16_step_timeline.md and DESIGN.md's appendix.)

#ifdef CSM_ATTN_OPROJ_KERNEL
// blockDim = 64 n_q (n_q in {2, 4, 8}: the K/V tile alone is 128 registers).  A thread multiplies KPT = max(8, K / 64)
// consecutive k of one output row, so a row takes K / KPT lanes (a whole wave for csm-1b's decoder: 8 rows per
// workgroup, grid = N / 8 = 128 workgroups -- the 2 MB of weights must be pulled by many CUs at once: a CU draws only
// ~11 B/clk from HBM, and the 32-rows-per-workgroup form of this kernel, 64 KB on each of 32 CUs, took 12 us).
template <typename KT, typename WT, int HD>
__global__ __launch_bounds__(512) void attn_oproj_kernel(AttnOprojArgs a) {
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

#endif  // CSM_ATTN_OPROJ_KERNEL
// returns -2 (caller runs the two-launch form) when the shape is not covered (head_dim not 64 / 128, n_q not 2 / 4 / 8, N % 32, cache longer than 32 positions)
int launch_attn_oproj(hipStream_t st, int wdtype, int kvdtype, const AttnOprojArgs& a);
