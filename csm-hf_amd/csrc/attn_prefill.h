// Flash-style causal GQA attention for prefill rows on the matrix cores (head_dim 64, up to 4 q-heads per
// kv-head), fp32 throughout (v_mfma_f32_32x32x2_f32 for Q K^T and for P V: bitwise fmaf chains), online
// softmax in fp32 like SDPA.  Replaces sdpa_attention_forward for q_len > 1 (transformers
// sdpa_attention.py:97-163: is_causal path, and the boolean-mask path for left-padded rows).
//
// One workgroup = (32 query positions, one kv-head, one sequence); wave g = query head j*G+g.  K and V tiles
// of 32 keys are staged ONCE per workgroup in LDS (shared by the 4 query heads); Q and P live in registers.
// Roofline: fp32 MFMA (157 TFLOP/s): 4*S^2*hd*n_q/2 flops per sequence per layer.
#pragma once
#include "common.h"

struct PrefillAttnArgs {
  const float* q;  // [B*S][n_q*64], pre-scaled by hd^-0.5, RoPE applied
  const void* kcache;
  const void* vcache;
  int n_q, n_kv, lmax;
  int S, past;          // row r = b*S + s sits at position past + s
  const int* kv_start;  // nullable: first valid key of each sequence (left padding)
  float* out;           // [B*S][n_q*64]
  bf16_t* oplanes;      // nullable: output as row-major planes [3][B*S][n_q*64] for the o_proj GEMM instead of `out`
  size_t plane_stride;
  const int* seq_slot;  // nullable: sequence b of this launch lives in cache slot seq_slot[b] (continuous batching: several rows of a
                        // running batch prefilled at once); kv_start is indexed by the slot too.  q / out rows stay b * S + s
  int map;              // attn_prefill_bf16_kernel: 1 = alternate waves of 256 workgroups walk the query tiles in ascending order (see the kernel)
  uint8_t* oq;          // nullable (attn_prefill_bf16_kernel only): output as MX-fp8 rows [B*S][n_q*64] e4m3 + os [B*S][n_q*2] E8M0 scales
  int kvfast;           // attn_prefill_bf16_kernel: grid (n_kv, query tiles, B) instead of (query tiles, n_kv, B)
  uint8_t* os;          // (the OCP recipe of mx_quant_rows_kernel, gemm_mx.h; a 32-block = half a head = the lane pair of a query row) instead of `out`
};

#ifdef CSM_ATTN_PREFILL_KERNELS   // defined by attn_prefill.hip only
// Both products are computed TRANSPOSED so that a lane owns one query row end to end:
//   S^T[key][row] = K Q^T   (A = K tile from LDS, B = Q from registers)  -> lane (row = lane&31, half) holds 16 keys
//   O^T[d][row]   = V^T P^T (A = V tile from LDS, B = P straight from the score accumulator registers)
// The contraction over keys in the second product runs in the order the score accumulator already holds them
// (key(t, half) = (t&3) + 8*(t>>2) + 4*half), so P never leaves its registers: no LDS transpose, and the softmax
// statistics are 16 in-lane max/add plus ONE exchange with the partner lane (lane ^ 32).
template <typename KT>
__global__ __launch_bounds__(256) void attn_prefill_kernel(PrefillAttnArgs a) {
  constexpr int HD = 64;
  __shared__ float Ks[32][HD + 1];
  __shared__ __attribute__((aligned(16))) float Vs[32][HD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = a.n_q / a.n_kv;
  const int qt = blockIdx.x, j = blockIdx.y, b = blockIdx.z;
  const int s0 = qt * 32;
  const int li = lane & 31, lh = lane >> 5;
  const bool head_live = wave < G;
  const int h = j * G + (head_live ? wave : 0);
  const int bs = a.seq_slot ? a.seq_slot[b] : b;   // cache slot of this sequence
  const int kv_lo = a.kv_start ? a.kv_start[bs] : 0;
  const int s_last = min(a.S - 1, s0 + 31);
  const int kmax = a.past + s_last;  // last key any row of this tile may see
  const int s = s0 + li;             // this lane's query row
  const bool row_live = s < a.S && head_live;
  const int row_kmax = a.past + s;

  // Q in the B-operand layout of 32x32x2 (= the A layout): lane holds Q[row li][dim 2t + lh], t = 0..31
  float qreg[32];
  {
    const float* qrow = a.q + ((size_t)b * a.S + (s < a.S ? s : a.S - 1)) * a.n_q * HD + (size_t)h * HD + lh;
#pragma unroll
    for (int t = 0; t < 32; ++t) qreg[t] = row_live ? qrow[2 * t] : 0.f;
  }
  const KT* kc = reinterpret_cast<const KT*>(a.kcache) + ((size_t)bs * a.n_kv + j) * (size_t)(HD / 4) * a.lmax * 4;
  const KT* vc = reinterpret_cast<const KT*>(a.vcache) + ((size_t)bs * a.n_kv + j) * (size_t)a.lmax * HD;

  f32x16 o0 = (f32x16)(0.f), o1 = (f32x16)(0.f);   // O^T tiles: d = (reg&3) + 8*(reg>>2) + 4*lh (+32), row = li
  float m_run = -INFINITY, l_run = 0.f;

  for (int kt0 = kv_lo & ~31; kt0 <= kmax; kt0 += 32) {
    // ---- stage the K / V tile (keys kt0 .. kt0+31) once for the 4 query heads -------------------------
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int d4 = idx >> 5, t = idx & 31;
      const int key = min(kt0 + t, a.lmax - 1);
      const KT* src = kc + ((size_t)d4 * a.lmax + key) * 4;
      Ks[t][d4 * 4 + 0] = to_f32(src[0]);
      Ks[t][d4 * 4 + 1] = to_f32(src[1]);
      Ks[t][d4 * 4 + 2] = to_f32(src[2]);
      Ks[t][d4 * 4 + 3] = to_f32(src[3]);
      const int tv = idx >> 4, c4 = idx & 15;
      const KT* vsrc = vc + (size_t)min(kt0 + tv, a.lmax - 1) * HD + c4 * 4;
      f32x4 vv;
      vv[0] = to_f32(vsrc[0]); vv[1] = to_f32(vsrc[1]); vv[2] = to_f32(vsrc[2]); vv[3] = to_f32(vsrc[3]);
      *reinterpret_cast<f32x4*>(&Vs[tv][c4 * 4]) = vv;
    }
    __syncthreads();
    // ---- S^T[key][row] = sum_d K[key][d] Q[row][d] --------------------------------------------------------
    f32x16 sc = (f32x16)(0.f);
#pragma unroll
    for (int t = 0; t < 32; ++t) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[li][2 * t + lh], qreg[t], sc, 0, 0, 0);
    // accumulator layout: column = query row li, accumulator row r -> key (r&3) + 8*(r>>2) + 4*lh
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const bool valid = row_live && key >= kv_lo && key <= row_kmax;
      sc[r] = valid ? sc[r] : -INFINITY;
      mx = fmaxf(mx, sc[r]);
    }
    mx = xor32_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const bool any = m_new > -INFINITY;
    const float alpha = any ? __expf(m_run - m_new) : 1.f;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc[r] = any ? __expf(sc[r] - m_new) : 0.f;   // exp(-inf) = 0 for masked keys
      sum += sc[r];
    }
    sum = xor32_sum(sum);
    l_run = l_run * alpha + sum;
    m_run = m_new;
    o0 *= alpha;
    o1 *= alpha;
    // ---- O^T[d][row] += sum_key V[key][d] P[row][key], keys in accumulator order -----------------------------
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int kk = (t & 3) + 8 * (t >> 2) + 4 * lh;
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kk][li], sc[t], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kk][32 + li], sc[t], o1, 0, 0, 0);
    }
    __syncthreads();   // the K/V tile is rewritten next iteration
  }
  if (!row_live) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;   // a fully masked (pad) row yields zeros
  float* dst = a.out + ((size_t)b * a.S + s) * a.n_q * HD + (size_t)h * HD;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {   // accumulator rows 4*r4 .. 4*r4+3 -> 4 consecutive d
    const int d = 8 * r4 + 4 * lh;
    f32x4 v0, v1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v0[i] = o0[4 * r4 + i] * inv; v1[i] = o1[4 * r4 + i] * inv; }
    if (a.oplanes) {
      bf16_t* pd = a.oplanes + ((size_t)b * a.S + s) * a.n_q * HD + (size_t)h * HD;
      store_rowplanes4(pd + d, a.plane_stride, v0);
      store_rowplanes4(pd + 32 + d, a.plane_stride, v1);
    } else {
      *reinterpret_cast<f32x4*>(dst + d) = v0;
      *reinterpret_cast<f32x4*>(dst + 32 + d) = v1;
    }
  }
}

// ---- prefill_precision = bf16 ---------------------------------------------------------------------------------------
// The same flash loop on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 2.5 PFLOP/s dense): Q, K, V and the
// probabilities are rounded to the nearest bf16 (what the reference's bf16 SDPA does), scores, running max / sum and
// the output accumulate in fp32.  64 keys per tile, both products transposed as above, so a lane still owns one query
// row.  The second product contracts over keys in the order the two score accumulators hold them: the 8 keys of lane
// half lh at 16-key step (hh, u) are 32hh + 16u + 4lh + {0..3} and 32hh + 16u + 8 + 4lh + {0..3}; the V tile is stored
// transposed in LDS with its keys permuted so that those 8 are contiguous (one 16-byte read per operand).
typedef __attribute__((ext_vector_type(8))) short ap_bf16x8;
typedef __attribute__((ext_vector_type(2))) float ap_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 ap_bf16x2;

__device__ __forceinline__ uint32_t ap_pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32 (round to nearest even)
  const ap_f32x2 v = {lo, hi};
  const ap_bf16x2 r = __builtin_convertvector(v, ap_bf16x2);
  return *reinterpret_cast<const uint32_t*>(&r);
}
__device__ __forceinline__ ap_bf16x8 ap_pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
  u32x4 w;
  w[0] = ap_pack2(a0, a1); w[1] = ap_pack2(a2, a3); w[2] = ap_pack2(a4, a5); w[3] = ap_pack2(a6, a7);
  return *reinterpret_cast<const ap_bf16x8*>(&w);
}
// position of key k (0..63) inside a row of the transposed V tile
__device__ __forceinline__ int ap_vperm(int k) {
  const int w = k & 15;
  return (((k >> 4) * 2 + ((w >> 2) & 1)) * 8) + (w & 3) + 4 * (w >> 3);
}

template <typename KT>
struct ApStage;   // one thread's share of a 64-key K / V tile, in flight in registers while the previous tile is computed
template <>
struct ApStage<float> {
  f32x4 k[4], v[4];
  __device__ __forceinline__ void load(const float* kc, const float* vc, int lmax, int kt0, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int key = min(kt0 + (idx & 63), lmax - 1);
      k[i] = *reinterpret_cast<const f32x4*>(kc + ((size_t)(idx >> 6) * lmax + key) * 4);
    }
    const float* vsrc = vc + (size_t)min(kt0 + (tid & 63), lmax - 1) * 64 + (tid >> 6) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(vsrc + 4 * i);
  }
  __device__ __forceinline__ uint2 kword(int i) const { return make_uint2(ap_pack2(k[i][0], k[i][1]), ap_pack2(k[i][2], k[i][3])); }
  __device__ __forceinline__ float kval(int i, int e) const { return k[i][e]; }   // exact mode: the fp32 value itself
  __device__ __forceinline__ float vval(int d) const { return v[d >> 2][d & 3]; }   // V[key][16c + d]
  // bf16 bits of V[key][16c + 2q] (low half) and V[key][16c + 2q + 1] (high half)
  __device__ __forceinline__ uint32_t vpair(int q) const { return ap_pack2(v[q >> 1][(2 * q) & 3], v[q >> 1][(2 * q + 1) & 3]); }
};
template <>
struct ApStage<bf16_t> {
  uint2 k[4];
  u32x4 v[2];
  __device__ __forceinline__ void load(const bf16_t* kc, const bf16_t* vc, int lmax, int kt0, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int key = min(kt0 + (idx & 63), lmax - 1);
      k[i] = *reinterpret_cast<const uint2*>(kc + ((size_t)(idx >> 6) * lmax + key) * 4);
    }
    const bf16_t* vsrc = vc + (size_t)min(kt0 + (tid & 63), lmax - 1) * 64 + (tid >> 6) * 16;
    v[0] = *reinterpret_cast<const u32x4*>(vsrc);
    v[1] = *reinterpret_cast<const u32x4*>(vsrc + 8);
  }
  __device__ __forceinline__ uint2 kword(int i) const { return k[i]; }
  __device__ __forceinline__ uint32_t vpair(int q) const { return v[q >> 2][q & 3]; }
  __device__ __forceinline__ float kval(int, int) const { return 0.f; }   // (one-plane cache: never split)
  __device__ __forceinline__ float vval(int) const { return 0.f; }
};

__device__ __forceinline__ float ap_max3(float a, float b, float c) {   // no NaN canonicalisation round trips
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Measured and NOT kept (profiles/r04_attn_prefill_knockout.txt; the variants are in the history): two key groups per workgroup
// on alternate key tiles (round 3: slower), two or three K / V tiles in flight instead of one, eight waves with waves 4-7 on keys
// 32-63 of every tile (round 4: both the same as this form).  Knock-outs: no single stream (exp2, MFMAs, LDS writes, loads,
// barriers) is worth more than 0.4 of 5.7 ms per 2 048-frame prefill; with all of them gone 23 of 60 us per launch remain.
template <typename KT>
__global__ __launch_bounds__(256) void attn_prefill_bf16_kernel(PrefillAttnArgs a) {
  constexpr int HD = 64, LDK = 72;   // 144-byte LDS rows: 16-byte reads of 32 consecutive rows cover all banks evenly
  __shared__ __attribute__((aligned(16))) bf16_t kv_all[2 * 64 * LDK];
  bf16_t* const Ks = kv_all;                  // [key][d]
  bf16_t* const Vt = kv_all + 64 * LDK;       // [d][ap_vperm(key)]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = a.n_q / a.n_kv;
  // query tile of this workgroup.  Every workgroup of a 2 048-frame context is resident at once (two per CU), so nothing balances
  // itself: a CU holding two of the latest tiles (32 key tiles each) ran twice as long as the average one (rounds 2-3).  map = 1:
  // every other wave of 256 workgroups walks the tiles in ascending order, so that the two workgroups of a CU (dispatch slots n and
  // n + 256) hold a long and a short tile: 2 048 frames 5.69 -> 5.51 ms (bf16), 4.36 -> 4.17 ms (mxfp8).  (Pairing neighbours in
  // dispatch order instead: no change -- the dispatcher is round-robin over the CUs.)
  // a.kvfast (n_kv == 8): the kv-head is the FASTEST grid index, i.e. workgroup id mod 8 = XCD = kv-head: every query tile of a head reads
  // that head's K / V through ONE L2 instead of eight (round 5; the decode kernels' k-split / query-head grouping, same reason)
  const int bxq = a.kvfast ? (int)blockIdx.y : (int)blockIdx.x, nqx = a.kvfast ? (int)gridDim.y : (int)gridDim.x;
  int qt = nqx - 1 - bxq;   // longest (latest) query tiles are dispatched first
  if (a.map == 1 && a.kvfast) {
    // rounds of 256 workgroups = 32 query positions x 8 heads: even rounds walk down from the latest tile, odd rounds up from the first,
    // so that dispatch slots n and n + 256 (one CU) hold a long and a short tile; a bijection on [0, nqx) for every nqx
    const int r = bxq >> 5, i = bxq & 31;
    qt = (r & 1) ? (r >> 1) * 32 + i : nqx - 1 - (r >> 1) * 32 - i;
  } else if (a.map == 1) {
    // the direction is chosen ONCE per (kv-head, sequence) row of the grid -- from the linear id of the row's first workgroup --
    // so that every query tile of the row is computed exactly once whatever gridDim.x is (round-4 form flipped on bit 8 of
    // the workgroup's own linear id: a 256-boundary inside a row left tiles uncomputed when ceil(S/32) did not divide 256)
    const unsigned row0 = gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if ((row0 >> 8) & 1) qt = blockIdx.x;
  }
  const int j = a.kvfast ? (int)blockIdx.x : (int)blockIdx.y, b = blockIdx.z;
  const int s0 = qt * 32;
  const int li = lane & 31, lh = lane >> 5;
  const bool head_live = wave < G;
  const int h = j * G + (head_live ? wave : 0);
  const int bs = a.seq_slot ? a.seq_slot[b] : b;   // cache slot of this sequence
  const int kv_lo = a.kv_start ? a.kv_start[bs] : 0;
  const int s_last = min(a.S - 1, s0 + 31);
  const int kmax = a.past + s_last;
  const int s = s0 + li;
  const bool row_live = s < a.S && head_live;
  // keys this lane's row may see: [row_lo, row_lo + row_span]; a dead row sees none
  const int row_lo = kv_lo;
  const unsigned row_span = row_live && a.past + s >= kv_lo ? (unsigned)(a.past + s - kv_lo) : 0u;
  const bool row_any = row_live && a.past + s >= kv_lo;
  const bool tile_full = s0 + 31 < a.S && G == 4;   // every lane of the workgroup owns a real row

  ap_bf16x8 qf[4];   // B operand of S^T = K Q^T: lane (row li, half lh) holds Q[row][16t + 8lh .. +7]
  {
    const float* qrow = a.q + ((size_t)b * a.S + (s < a.S ? s : a.S - 1)) * a.n_q * HD + (size_t)h * HD + 8 * lh;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(qrow + 16 * t), x1 = *reinterpret_cast<const f32x4*>(qrow + 16 * t + 4);
      qf[t] = row_live ? ap_pack8(x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]) : (ap_bf16x8)(0);
    }
  }
  const KT* kc = reinterpret_cast<const KT*>(a.kcache) + ((size_t)bs * a.n_kv + j) * (size_t)(HD / 4) * a.lmax * 4;
  const KT* vc = reinterpret_cast<const KT*>(a.vcache) + ((size_t)bs * a.n_kv + j) * (size_t)a.lmax * HD;

  f32x16 o0 = (f32x16)(0.f), o1 = (f32x16)(0.f);
  // running reference in log2 units.  It is NOT the running max: it only moves when a tile's max exceeds it by more than
  // 2^8 (probabilities stay <= 256: harmless in bf16 / fp32), so the accumulator rescale is rare instead of per tile.
  float m_run = -INFINITY, l_run = 0.f;
  constexpr float L2E = 1.4426950408889634f, SLACK = 8.f;

  ApStage<KT> st;
  int kt0 = kv_lo & ~63;
  if (kt0 <= kmax) st.load(kc, vc, a.lmax, kt0, tid);
  // where this thread's V share lands: rows d = 16*wave + 2q (+1 for odd lanes), word = the (key, key^1) pair
  const int vkey = tid & 63;
  uint32_t* const vdst = reinterpret_cast<uint32_t*>(Vt) + ((16 * wave + (vkey & 1)) * LDK + ap_vperm(vkey & ~1)) / 2;
  // even lanes keep dimension 2q of (own key, next key), odd lanes dimension 2q+1 of (previous key, own key)
  const uint32_t vsel = (vkey & 1) ? 0x03020706u : 0x05040100u;   // v_perm_b32 byte selector over {theirs, mine}
  for (; kt0 <= kmax; kt0 += 64) {
    // ---- registers -> LDS ------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      *reinterpret_cast<uint2*>(&Ks[(idx & 63) * LDK + (idx >> 6) * 4]) = st.kword(i);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t mine = st.vpair(q);
      const uint32_t theirs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, true);   // lane ^ 1
      vdst[q * LDK] = __builtin_amdgcn_perm(theirs, mine, vsel);
    }
    __syncthreads();
    if (kt0 + 64 <= kmax) st.load(kc, vc, a.lmax, kt0 + 64, tid);   // next tile in flight behind this tile's math
    // ---- S^T[key][row] = sum_d K[key][d] Q[row][d], two 32-key halves ----------------------------------------------
    f32x16 sc0 = (f32x16)(0.f), sc1 = (f32x16)(0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const ap_bf16x8 k0 = *reinterpret_cast<const ap_bf16x8*>(&Ks[li * LDK + 16 * t + 8 * lh]);
      const ap_bf16x8 k1 = *reinterpret_cast<const ap_bf16x8*>(&Ks[(32 + li) * LDK + 16 * t + 8 * lh]);
      sc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[t], sc0, 0, 0, 0);
      sc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[t], sc1, 0, 0, 0);
    }
    // interior tiles (every key visible to every row) skip the mask
    const bool interior = tile_full && kt0 >= kv_lo && kt0 + 63 <= a.past + s0;
    if (!interior) {
      const int base = kt0 + 4 * lh - row_lo;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rel = base + (r & 3) + 8 * (r >> 2);   // key - row_lo: valid iff 0 <= rel <= row_span
        sc0[r] = (row_any && (unsigned)rel <= row_span) ? sc0[r] : -INFINITY;
        sc1[r] = (row_any && (unsigned)(rel + 32) <= row_span) ? sc1[r] : -INFINITY;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = ap_max3(mx, sc0[r], sc1[r]);
    mx = xor32_max(mx) * L2E;
    const bool move = mx > m_run + SLACK;   // also the first tile with a visible key (m_run = -inf)
    if (__builtin_amdgcn_ballot_w64(move) != 0) {
      const float m_new = move ? mx : m_run;
      const float alpha = (move && m_run > -INFINITY) ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.f;
      m_run = m_new;
      l_run *= alpha;
      o0 *= alpha;
      o1 *= alpha;
    }
    const float neg_m = m_run > -INFINITY ? -m_run : 0.f;   // a row with no visible key so far: exp2(-inf) = 0 anyway
    ap_f32x2 sum2 = {0.f, 0.f};
    const ap_f32x2 l2e2 = {L2E, L2E}, nm2 = {neg_m, neg_m};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      ap_f32x2 x0 = {sc0[r], sc0[r + 1]}, x1 = {sc1[r], sc1[r + 1]};
      x0 = x0 * l2e2 + nm2;
      x1 = x1 * l2e2 + nm2;
      x0[0] = __builtin_amdgcn_exp2f(x0[0]); x0[1] = __builtin_amdgcn_exp2f(x0[1]);
      x1[0] = __builtin_amdgcn_exp2f(x1[0]); x1[1] = __builtin_amdgcn_exp2f(x1[1]);
      sum2 += x0 + x1;
      sc0[r] = x0[0]; sc0[r + 1] = x0[1];
      sc1[r] = x1[0]; sc1[r + 1] = x1[1];
    }
    l_run += xor32_sum(sum2[0] + sum2[1]);
    // ---- O^T[d][row] += sum_key V[key][d] P[row][key] -------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // u = 2*hh + step
      const f32x16& sc = (u < 2) ? sc0 : sc1;
      const int r0 = 8 * (u & 1);
      const ap_bf16x8 pb = ap_pack8(sc[r0], sc[r0 + 1], sc[r0 + 2], sc[r0 + 3], sc[r0 + 4], sc[r0 + 5], sc[r0 + 6], sc[r0 + 7]);
      const ap_bf16x8 v0 = *reinterpret_cast<const ap_bf16x8*>(&Vt[li * LDK + (2 * u + lh) * 8]);
      const ap_bf16x8 v1 = *reinterpret_cast<const ap_bf16x8*>(&Vt[(32 + li) * LDK + (2 * u + lh) * 8]);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb, o1, 0, 0, 0);
    }
    __syncthreads();   // the tile is rewritten at the top of the next iteration
  }
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  if (a.oq) {   // MX-fp8 output for the o_proj GEMM: dims 0-31 / 32-63 of the head are one scale block each, held by this lane and lane ^ 32
    float m0 = 0.f, m1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= inv; o1[r] *= inv; m0 = fmaxf(m0, fabsf(o0[r])); m1 = fmaxf(m1, fabsf(o1[r])); }
    m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
    m1 = fmaxf(m1, __shfl_xor(m1, 32, 64));
    if (!row_live) return;
    int e0 = (int)((__float_as_uint(m0) >> 23) & 0xff) - 8, e1 = (int)((__float_as_uint(m1) >> 23) & 0xff) - 8;
    e0 = e0 < 0 ? 0 : (e0 > 254 ? 254 : e0);
    e1 = e1 < 0 ? 0 : (e1 > 254 ? 254 : e1);
    const float i0 = __uint_as_float((uint32_t)(254 - e0) << 23), i1 = __uint_as_float((uint32_t)(254 - e1) << 23);
    const size_t row = (size_t)b * a.S + s;
    uint8_t* qd = a.oq + row * a.n_q * HD + (size_t)h * HD;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int d = 8 * r4 + 4 * lh;
      float w0[4], w1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w0[i] = fminf(fmaxf(o0[4 * r4 + i] * i0, -448.f), 448.f);
        w1[i] = fminf(fmaxf(o1[4 * r4 + i] * i1, -448.f), 448.f);
      }
      int p0 = 0, p1 = 0;
      p0 = __builtin_amdgcn_cvt_pk_fp8_f32(w0[0], w0[1], p0, false);
      p0 = __builtin_amdgcn_cvt_pk_fp8_f32(w0[2], w0[3], p0, true);
      p1 = __builtin_amdgcn_cvt_pk_fp8_f32(w1[0], w1[1], p1, false);
      p1 = __builtin_amdgcn_cvt_pk_fp8_f32(w1[2], w1[3], p1, true);
      *reinterpret_cast<uint32_t*>(qd + d) = (uint32_t)p0;
      *reinterpret_cast<uint32_t*>(qd + 32 + d) = (uint32_t)p1;
    }
    if (lh == 0) {
      uint8_t* sd = a.os + row * (a.n_q * 2) + h * 2;
      sd[0] = (uint8_t)e0;
      sd[1] = (uint8_t)e1;
    }
    return;
  }
  if (!row_live) return;
  float* dst = a.out + ((size_t)b * a.S + s) * a.n_q * HD + (size_t)h * HD;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int d = 8 * r4 + 4 * lh;
    f32x4 v0, v1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v0[i] = o0[4 * r4 + i] * inv; v1[i] = o1[4 * r4 + i] * inv; }
    if (a.oplanes) {
      bf16_t* pd = a.oplanes + ((size_t)b * a.S + s) * a.n_q * HD + (size_t)h * HD;
      store_rowplanes4(pd + d, a.plane_stride, v0);
      store_rowplanes4(pd + 32 + d, a.plane_stride, v1);
    } else {
      *reinterpret_cast<f32x4*>(dst + d) = v0;
      *reinterpret_cast<f32x4*>(dst + 32 + d) = v1;
    }
  }
}

// ---- exact mode on the bf16 matrix pipe ------------------------------------------------------------------------------
// The fp32 flash kernel above pays 64 matrix clocks per 32x32x2 product; the bf16 pipe is 16 times faster per flop.  Here
// every fp32 operand is split EXACTLY into three bf16 pieces by truncation (x = hi + mid + lo, 8 + 8 + 8 mantissa bits:
// what the exact prefill GEMM does with its activations) and a product a.b is evaluated as the six piece products whose
// weight is >= 2^-16 relative (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi -- small terms first; the three dropped terms
// are below 2^-24, the fp32 rounding unit), each exact in the fp32 accumulator.  Scores, statistics and the output stay
// fp32: the result carries fp32 summation-order error only, like the fp32-MFMA kernel, at 96 bf16 MFMAs (3 072 matrix
// clocks) per 64-key tile and head instead of 128 fp32 MFMAs (8 192).  Q is split once, K / V while they are staged to
// LDS (a bf16 cache is its own hi piece: one plane), P per tile in registers.  Same tiling, layouts and lazy rescale as
// attn_prefill_bf16_kernel.
__device__ __forceinline__ void ap_split3(const float* x, ap_bf16x8& hi, ap_bf16x8& mid, ap_bf16x8& lo) {   // 8 values
  uint32_t* ph = reinterpret_cast<uint32_t*>(&hi);
  uint32_t* pm = reinterpret_cast<uint32_t*>(&mid);
  uint32_t* pl = reinterpret_cast<uint32_t*>(&lo);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float v0 = x[2 * p], v1 = x[2 * p + 1];
    const uint32_t h0 = __float_as_uint(v0) & 0xffff0000u, h1 = __float_as_uint(v1) & 0xffff0000u;
    const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
    const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);
    ph[p] = (h0 >> 16) | h1;
    pm[p] = (m0 >> 16) | m1;
    pl[p] = (__float_as_uint(s0) >> 16) | (__float_as_uint(s1) & 0xffff0000u);
  }
}
// the three pieces of one value as bf16 bit patterns
__device__ __forceinline__ void ap_split1(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
  const uint32_t hb = __float_as_uint(v) & 0xffff0000u;
  const float r = v - __uint_as_float(hb);
  const uint32_t mb = __float_as_uint(r) & 0xffff0000u;
  const float t = r - __uint_as_float(mb);
  h = hb >> 16; m = mb >> 16; l = __float_as_uint(t) >> 16;
}

template <typename KT>
__global__ __launch_bounds__(256) void attn_prefill_x3_kernel(PrefillAttnArgs a) {
  constexpr int HD = 64, LDK = 72;
  constexpr int NPK = sizeof(KT) == 4 ? 3 : 1;   // planes of K / V (a bf16 cache is exact in one)
  extern __shared__ __attribute__((aligned(16))) bf16_t xs[];   // Ks[NPK][64 * LDK] | Vt[NPK][64 * LDK]
  auto Ks = [&](int p) { return xs + (size_t)p * 64 * LDK; };
  auto Vt = [&](int p) { return xs + (size_t)(NPK + p) * 64 * LDK; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = a.n_q / a.n_kv;
  const int qt = gridDim.x - 1 - blockIdx.x;
  const int j = blockIdx.y, b = blockIdx.z;
  const int s0 = qt * 32;
  const int li = lane & 31, lh = lane >> 5;
  const bool head_live = wave < G;
  const int h = j * G + (head_live ? wave : 0);
  const int bs = a.seq_slot ? a.seq_slot[b] : b;   // cache slot of this sequence
  const int kv_lo = a.kv_start ? a.kv_start[bs] : 0;
  const int s_last = min(a.S - 1, s0 + 31);
  const int kmax = a.past + s_last;
  const int s = s0 + li;
  const bool row_live = s < a.S && head_live;
  const int row_lo = kv_lo;
  const unsigned row_span = row_live && a.past + s >= kv_lo ? (unsigned)(a.past + s - kv_lo) : 0u;
  const bool row_any = row_live && a.past + s >= kv_lo;
  const bool tile_full = s0 + 31 < a.S && G == 4;

  ap_bf16x8 qf[3][4];   // three pieces of Q[row][16t + 8lh .. +7]
  {
    const float* qrow = a.q + ((size_t)b * a.S + (s < a.S ? s : a.S - 1)) * a.n_q * HD + (size_t)h * HD + 8 * lh;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = row_live ? qrow[16 * t + i] : 0.f;
      ap_split3(x, qf[0][t], qf[1][t], qf[2][t]);
    }
  }
  const KT* kc = reinterpret_cast<const KT*>(a.kcache) + ((size_t)bs * a.n_kv + j) * (size_t)(HD / 4) * a.lmax * 4;
  const KT* vc = reinterpret_cast<const KT*>(a.vcache) + ((size_t)bs * a.n_kv + j) * (size_t)a.lmax * HD;

  f32x16 o0 = (f32x16)(0.f), o1 = (f32x16)(0.f);
  float m_run = -INFINITY, l_run = 0.f;
  constexpr float L2E = 1.4426950408889634f, SLACK = 8.f;

  ApStage<KT> st;
  int kt0 = kv_lo & ~63;
  st.load(kc, vc, a.lmax, kt0, tid);
  const int vkey = tid & 63;
  const int voff = ((16 * wave + (vkey & 1)) * LDK + ap_vperm(vkey & ~1)) / 2;   // in 32-bit words
  const uint32_t vsel = (vkey & 1) ? 0x03020706u : 0x05040100u;
  for (; kt0 <= kmax; kt0 += 64) {
    // ---- registers -> LDS, split into pieces on the way ---------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int off = (idx & 63) * LDK + (idx >> 6) * 4;
      if (NPK == 1) {
        *reinterpret_cast<uint2*>(Ks(0) + off) = st.kword(i);
      } else {
        uint32_t hh[4], mm[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ap_split1(st.kval(i, e), hh[e], mm[e], ll[e]);
        *reinterpret_cast<uint2*>(Ks(0) + off) = make_uint2(hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16));
        *reinterpret_cast<uint2*>(Ks(1) + off) = make_uint2(mm[0] | (mm[1] << 16), mm[2] | (mm[3] << 16));
        *reinterpret_cast<uint2*>(Ks(2) + off) = make_uint2(ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16));
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint32_t mine[3];
      if (NPK == 1) {
        mine[0] = st.vpair(q);
      } else {
        uint32_t a0, a1, a2, b0, b1, b2;
        ap_split1(st.vval(2 * q), a0, a1, a2);
        ap_split1(st.vval(2 * q + 1), b0, b1, b2);
        mine[0] = a0 | (b0 << 16); mine[1] = a1 | (b1 << 16); mine[2] = a2 | (b2 << 16);
      }
#pragma unroll
      for (int p = 0; p < NPK; ++p) {
        const uint32_t theirs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine[p], 0xB1, 0xf, 0xf, true);   // lane ^ 1
        reinterpret_cast<uint32_t*>(Vt(p))[voff + q * LDK] = __builtin_amdgcn_perm(theirs, mine[p], vsel);
      }
    }
    __syncthreads();
    if (kt0 + 64 <= kmax) st.load(kc, vc, a.lmax, kt0 + 64, tid);
    // ---- S^T = K Q^T: piece products, small terms first ---------------------------------------------------------
    f32x16 sc0 = (f32x16)(0.f), sc1 = (f32x16)(0.f);
    // (K piece, Q piece): with a one-plane cache only K's hi piece exists
    constexpr int NT = NPK == 3 ? 6 : 3;
    constexpr int KP[6] = {2, 0, 1, 1, 0, 0}, QP[6] = {0, 2, 1, 0, 1, 0};   // lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi
    constexpr int QP1[3] = {2, 1, 0};                                       // hi_K . (lo, mid, hi)_Q
#pragma unroll
    for (int term = 0; term < NT; ++term) {
      const int kp = NPK == 3 ? KP[term] : 0, qp = NPK == 3 ? QP[term] : QP1[term];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const ap_bf16x8 k0 = *reinterpret_cast<const ap_bf16x8*>(Ks(kp) + li * LDK + 16 * t + 8 * lh);
        const ap_bf16x8 k1 = *reinterpret_cast<const ap_bf16x8*>(Ks(kp) + (32 + li) * LDK + 16 * t + 8 * lh);
        sc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[qp][t], sc0, 0, 0, 0);
        sc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[qp][t], sc1, 0, 0, 0);
      }
    }
    const bool interior = tile_full && kt0 >= kv_lo && kt0 + 63 <= a.past + s0;
    if (!interior) {
      const int base = kt0 + 4 * lh - row_lo;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rel = base + (r & 3) + 8 * (r >> 2);
        sc0[r] = (row_any && (unsigned)rel <= row_span) ? sc0[r] : -INFINITY;
        sc1[r] = (row_any && (unsigned)(rel + 32) <= row_span) ? sc1[r] : -INFINITY;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = ap_max3(mx, sc0[r], sc1[r]);
    mx = xor32_max(mx) * L2E;
    const bool move = mx > m_run + SLACK;
    if (__builtin_amdgcn_ballot_w64(move) != 0) {
      const float m_new = move ? mx : m_run;
      const float alpha = (move && m_run > -INFINITY) ? exp2f(m_run - m_new) : 1.f;
      m_run = m_new;
      l_run *= alpha;
      o0 *= alpha;
      o1 *= alpha;
    }
    const float neg_m = m_run > -INFINITY ? -m_run : 0.f;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc0[r] = exp2f(fmaf(sc0[r], L2E, neg_m));   // the library exp2 (the fp32 kernel's accuracy class), not the raw instruction
      sc1[r] = exp2f(fmaf(sc1[r], L2E, neg_m));
      sum += sc0[r] + sc1[r];
    }
    l_run += xor32_sum(sum);
    // ---- O^T += V^T P^T: piece products, small terms first ---------------------------------------------------------
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x16& sc = (u < 2) ? sc0 : sc1;
      const int r0 = 8 * (u & 1);
      float pv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pv[i] = sc[r0 + i];
      ap_bf16x8 pp[3];
      ap_split3(pv, pp[0], pp[1], pp[2]);
#pragma unroll
      for (int term = 0; term < NT; ++term) {
        const int vp = NPK == 3 ? KP[term] : 0, ppi = NPK == 3 ? QP[term] : QP1[term];
        const ap_bf16x8 v0 = *reinterpret_cast<const ap_bf16x8*>(Vt(vp) + li * LDK + (2 * u + lh) * 8);
        const ap_bf16x8 v1 = *reinterpret_cast<const ap_bf16x8*>(Vt(vp) + (32 + li) * LDK + (2 * u + lh) * 8);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pp[ppi], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pp[ppi], o1, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (!row_live) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  float* dst = a.out + ((size_t)b * a.S + s) * a.n_q * HD + (size_t)h * HD;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int d = 8 * r4 + 4 * lh;
    f32x4 v0, v1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v0[i] = o0[4 * r4 + i] * inv; v1[i] = o1[4 * r4 + i] * inv; }
    if (a.oplanes) {
      bf16_t* pd = a.oplanes + ((size_t)b * a.S + s) * a.n_q * HD + (size_t)h * HD;
      store_rowplanes4(pd + d, a.plane_stride, v0);
      store_rowplanes4(pd + 32 + d, a.plane_stride, v1);
    } else {
      *reinterpret_cast<f32x4*>(dst + d) = v0;
      *reinterpret_cast<f32x4*>(dst + 32 + d) = v1;
    }
  }
}
#endif  // CSM_ATTN_PREFILL_KERNELS

// returns -2 when the shape is not covered (head_dim != 64 or more than 4 q-heads per kv-head)
// bf16_math: Q / K / V / P rounded to bf16 on the bf16 matrix pipe (prefill_precision = bf16) instead of exact fp32
// bf16_math: 1 = Q / K / V / P rounded to bf16 (prefill_precision = bf16); 2 = exact, three-piece products on the bf16 pipe
int launch_attn_prefill(hipStream_t st, int kvdtype, int B, int hd, const PrefillAttnArgs& a, int bf16_math);
