// Small HBM/L2-bound kernels of the path: frame-embedding gather-sum (K1), row RMSNorm (K2, prefill),
// RoPE + KV scatter for prefill rows (K4/K5), fused sample + feedback (K12/K13).
#pragma once
#include "common.h"
#include "sample_wave.h"

// ---------------------------------------------------------------------------------------------------
// K1: out[r,:] = sum_{c<C} mask[r,c] * audio_emb[ids[r,c] + c*V, :] + mask[r,C] * text_emb[ids[r,C], :]
// Reference: modeling_csm.py:261-282 (_embed_tokens) + :327-334 (mask, sum(dim=2)) -- without the
// [B,S,33,H] intermediate.  fp32 accumulate in codebook order, one output write.
// Ring mode (ids == nullptr): row b takes its 32 audio tokens from the on-device frame ring at the
// current frame index (the frame generate() feeds back, modeling_csm.py:675-687: text column masked).
// Algorithmic bytes: live_tokens * H * sizeof(WT) read + H*4 written per row.
// ---------------------------------------------------------------------------------------------------
struct EmbedArgs {
  const void* text_emb;
  const void* audio_emb;
  int H, C, V;
  const int64_t* ids;   // [rows][C+1] or nullptr (ring mode)
  const uint8_t* mask;  // [rows][C+1] or nullptr (all live)
  const int64_t* ring;  // [B][max_frames][C]
  const int* frame_ptr;
  int max_frames;
  float* out;  // [rows][H]
  // ring mode: stop bookkeeping on the device (no host sync per frame).  zero_count[f] += 1 for every row whose
  // frame f is all-zero; row_done[row] = 1 from then on (read by the sampler when per-row stop is on)
  int* zero_count;   // [max_frames] nullable
  int* row_done;     // [rows] nullable
  // batched decode on activation planes: the summed row also leaves as the B operands of the first layer's QKV launch
  // (x * oln as three bf16 planes in fragment order, 16 rows per group) plus per-16-column sums of x^2 for its RMS scale
  bf16_t* oplanes;   // nullable
  int pl1;           // decode_precision = bf16: one nearest-even plane
  const float* oln;  // [H] the consumer's norm weight
  float* oss;        // [rows][oss_ld]
  int oss_ld;
  uint32_t* dbg;     // timeline probe slot (common.h TL_BEGIN), nullable
};

#ifndef CSM_ARGS_ONLY
// grid = (rows, H / 512): one workgroup sums a 512-column slab of one frame; its 4 waves split the 33 tokens
// (9 + 8 + 8 + 8), every row load of a wave is issued before any is consumed, partial sums meet in LDS in a
// fixed order (wave 0 first: codebook order is only approximately the reference's, fp32 accumulate either way).
template <typename WT>
__global__ __launch_bounds__(256) void embed_sum_kernel(EmbedArgs a) {
  __shared__ float part[4][512];
  __shared__ int nz[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TL_BEGIN(a.dbg);
  const int k = blockIdx.y * 512 + lane * 8;
  const WT* te = reinterpret_cast<const WT*>(a.text_emb);
  const WT* ae = reinterpret_cast<const WT*>(a.audio_emb);
  const int f = a.ids ? 0 : *a.frame_ptr;
  const int nt = a.C + 1;                       // tokens of a frame (32 audio + text)
  const int per = (nt + 3) / 4;                 // 9 for 33
  const int c0 = wave * per, c1 = min(nt, c0 + per);
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (!a.ids && a.zero_count && blockIdx.y == 0) {   // any non-zero audio token among this wave's share of the frame?
    int any = 0;
    for (int c = c0 + lane; c < c1 && c < a.C; c += 64) any |= a.ring[((size_t)row * a.max_frames + f) * a.C + c] != 0;
    const unsigned long long bal = __ballot(any);
    if (lane == 0) nz[wave] = bal != 0ull;
  }
  if (k < a.H) {
    constexpr int MAXT = 16;
    W8<WT> w[MAXT];
    bool lv[MAXT];
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
      const int c = c0 + u;
      lv[u] = false;
      const WT* src = ae + k;
      if (c < c1) {
        if (c < a.C) {
          int64_t tok;
          if (a.ids) {
            tok = a.ids[(size_t)row * nt + c];
            lv[u] = a.mask ? a.mask[(size_t)row * nt + c] != 0 : true;
          } else {
            tok = a.ring[((size_t)row * a.max_frames + f) * a.C + c];
            lv[u] = true;
          }
          if (lv[u]) src = ae + ((size_t)tok + (size_t)c * a.V) * a.H + k;
        } else if (a.ids) {                     // text column (masked out in ring mode)
          lv[u] = a.mask ? a.mask[(size_t)row * nt + c] != 0 : true;
          if (lv[u]) src = te + (size_t)a.ids[(size_t)row * nt + c] * a.H + k;
        }
      }
      w[u].load(src);
    }
#pragma unroll
    for (int u = 0; u < MAXT; ++u)
      if (lv[u]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += w[u].get(i);
      }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) part[wave][lane * 8 + i] = acc[i];
  __syncthreads();
  for (int i = tid; i < 512; i += 256) {
    const int kk = blockIdx.y * 512 + i;
    const float v = ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i];
    if (kk < a.H) {
      a.out[(size_t)row * a.H + kk] = v;
      if (a.oplanes) store_planes(a.oplanes + (size_t)(row >> 4) * 3 * 16 * a.H, (size_t)a.H * 16, kk, row & 15, v * a.oln[kk], a.pl1 != 0);
    }
    if (a.oplanes) part[0][i] = kk < a.H ? v * v : 0.f;   // same thread read part[0][i] above
  }
  if (a.oplanes) {   // per-tile (16 columns) sums of squares, fixed order
    __syncthreads();
    if (tid < 32) {
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) q += part[0][tid * 16 + j];
      const int tile = blockIdx.y * 32 + tid;
      if (tile * 16 < a.H) a.oss[(size_t)row * a.oss_ld + tile] = q;
    }
  }
  if (!a.ids && a.zero_count && blockIdx.y == 0 && tid == 0 && !(nz[0] | nz[1] | nz[2] | nz[3])) {
    atomicAdd(a.zero_count + f, 1);
    if (a.row_done) a.row_done[row] = 1;
  }
  TL_END(6);
}

#endif  // CSM_ARGS_ONLY

// ---------------------------------------------------------------------------------------------------
// K2 (stand-alone form, prefill + last_hidden_state): out = w * (x * rsqrt(mean(x^2) + eps))
// Reference: transformers LlamaRMSNorm.forward (modeling_llama.py:62-67).
// `frame_ptr`/`frame_stride`/`frame_add`: optional output offset (trace ring) = (*frame_ptr+add)*stride.
// ---------------------------------------------------------------------------------------------------
#ifndef CSM_ARGS_ONLY
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* x, int ldx, const float* w, int H, float eps,
                                                      float* out, int ldo, const int* frame_ptr,
                                                      size_t frame_stride, int frame_add, bf16_t* planes,
                                                      size_t plane_stride, const float* part, int nsplit,
                                                      size_t part_stride, int ldp, uint8_t* mxq, uint8_t* mxs) {
  // mxq != nullptr (prefill_precision = mxfp8, H % 32 == 0): the normed row goes out as OCP MX-fp8 -- e4m3 elements
  // [rows][H] + one E8M0 scale per 32 [rows][H/32] (gemm_mx.h: the recipe of mx_quant_rows_kernel, 8 lanes per block) --
  // the A operand of the next GEMM; `out` is then not written
  // part != nullptr: the row first takes up the split-K partial products of the preceding residual GEMM
  // (gemm.h GEPI_PARTIAL): x[row] += part[0][row] + part[1][row] + ... in that order, written back in place
  // planes != nullptr: the normed row goes out as three exact bf16 planes [3][rows][H] for the prefill GEMM
  // (split once here instead of once per column block of every GEMM that reads it); `out` is then not written
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (size_t)row * ldx;
  float* o = out + (size_t)row * ldo;
  if (frame_ptr) o += (size_t)(*frame_ptr + frame_add) * frame_stride;
  float ss = 0.f;
  const bool late_store = gridDim.x <= 128;   // measured: 192 ... 512 rows are 0.5-1.5 % faster with the immediate store, 64 rows with the late one
  // the row stays in registers between the two passes (round 3: one read of x instead of two; rows wider than 8192 re-read)
  constexpr int KEEP = 8;
  f32x4 keep[KEEP];
  // the partial products of a row piece are requested TOGETHER (eight at a time) and then added in split order: a short prefill's
  // launch is one or two memory round trips long instead of one per split (64 rows x 8 splits: 8 -> 5 us per launch)
#pragma unroll
  for (int it = 0; it < KEEP; ++it) {
    const int k = tid * 4 + it * 1024;
    keep[it] = (f32x4)(0.f);
    if (k < H) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
      if (part) {
        const float* pr = part + (size_t)row * ldp + k;
        for (int s0 = 0; s0 < nsplit; s0 += 8) {
          f32x4 q[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (s0 + j < nsplit) q[j] = *reinterpret_cast<const f32x4*>(pr + (size_t)(s0 + j) * part_stride);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (s0 + j < nsplit) { v[0] += q[j][0]; v[1] += q[j][1]; v[2] += q[j][2]; v[3] += q[j][3]; }
        }
      }
      keep[it] = v;
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      // more than 128 rows (bandwidth-bound launch): the folded piece goes back at once, under the next piece's loads
      if (part && !late_store) *reinterpret_cast<f32x4*>(const_cast<float*>(xr) + k) = v;
    }
  }
  if (part && late_store) {   // few rows (latency-bound launch): after every load of this thread, so that no store sits between the load batches
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
      const int k = tid * 4 + it * 1024;
      if (k < H) *reinterpret_cast<f32x4*>(const_cast<float*>(xr) + k) = keep[it];
    }
  }
  for (int k = tid * 4 + KEEP * 1024; k < H; k += 1024) {
    f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
    if (part) {
      const float* pr = part + (size_t)row * ldp + k;
      for (int sp = 0; sp < nsplit; ++sp) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(pr + (size_t)sp * part_stride);
        v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3];
      }
      *reinterpret_cast<f32x4*>(const_cast<float*>(xr) + k) = v;   // same thread re-reads it below
    }
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float sc = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
  auto emit = [&](int k, f32x4 v) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(w + k);
    v[0] = (v[0] * sc) * g[0];
    v[1] = (v[1] * sc) * g[1];
    v[2] = (v[2] * sc) * g[2];
    v[3] = (v[3] * sc) * g[3];
    if (mxq) {
      float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      m = fmaxf(m, __shfl_xor(m, 4, 64));
      int eb = (int)((__float_as_uint(m) >> 23) & 0xff) - 8;
      eb = eb < 0 ? 0 : (eb > 254 ? 254 : eb);
      const float inv = __uint_as_float((uint32_t)(254 - eb) << 23);
      int pk = 0;
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[0] * inv, -448.f), 448.f), fminf(fmaxf(v[1] * inv, -448.f), 448.f), pk, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[2] * inv, -448.f), 448.f), fminf(fmaxf(v[3] * inv, -448.f), 448.f), pk, true);
      *reinterpret_cast<uint32_t*>(mxq + (size_t)row * H + k) = (uint32_t)pk;
      if ((tid & 7) == 0) mxs[(size_t)row * (H >> 5) + (k >> 5)] = (uint8_t)eb;
    } else if (planes) store_rowplanes4(planes + (size_t)row * H + k, plane_stride, v);
    else *reinterpret_cast<f32x4*>(o + k) = v;
  };
#pragma unroll
  for (int it = 0; it < KEEP; ++it) {
    const int k = tid * 4 + it * 1024;
    if (k < H) emit(k, keep[it]);
  }
  for (int k = tid * 4 + KEEP * 1024; k < H; k += 1024) emit(k, *reinterpret_cast<const f32x4*>(xr + k));
}

#endif  // CSM_ARGS_ONLY

// ---------------------------------------------------------------------------------------------------
// K4/K5 for prefill rows: qkv [R][(n_q+2n_kv)*hd] raw projections -> q (RoPE, scaled) and the KV cache.
// Reference: apply_rotary_pos_emb (modeling_llama.py:130-160), DynamicCache.update (cache_utils.py:144).
// ---------------------------------------------------------------------------------------------------
struct RopeArgs {
  const float* qkv;
  int n_q, n_kv, hd;
  float qscale;
  const float* cos_tab;
  const float* sin_tab;
  const int* row_seq;
  const int* row_pos;
  float* qbuf;
  void* kcache;
  void* vcache;
  int lmax;
  const int* rope_pos;  // nullable: caller-supplied RoPE positions (reference `position_ids`, modeling_csm.py:296,349);
                        // the cache slot stays row_pos -- HF masks by cache index and rotates by position_ids
  // split-K QKV GEMM of a short prefill: nsplit > 1 partial products [nsplit][rows][(n_q + 2 n_kv) hd], part_stride apart,
  // summed here in fixed order (qkv is then ignored)
  const float* part;
  int nsplit;
  size_t part_stride;
};

// ---- KV-cache interchange with the HF layout [B][n_kv][len][hd] fp32 (transformers DynamicCache layers) ----------------
struct KvConvArgs {
  void* kcache;    // engine layout K [B][n_kv][hd/4][lmax][4], V [B][n_kv][lmax][hd]
  void* vcache;
  float* k_hf;
  float* v_hf;
  int B, n_kv, hd, lmax, len;
  int to_engine;   // 1: HF -> engine (import), 0: engine -> HF (export)
};

// ---- continuous batching: move every cached position of the resident batch `delta` slots up (csm_shift_context) so that a
// context LONGER than the batch's current length can join right-aligned.  K is stored post-RoPE, rotated by its slot index;
// attention only sees position differences, so the moved keys are rotated by `delta` more (cos/sin row `delta` of the
// engine's own table) and every later query / key of the row -- rotated by its new slot -- keeps the same distances.
struct KvShiftArgs {
  const void* kcache;   // engine layout K [B][n_kv][hd/4][lmax][4], V [B][n_kv][lmax][hd]
  const void* vcache;
  void* ktmp;           // compact [B][n_kv][hd/4][len][4]
  void* vtmp;           // compact [B][n_kv][len][hd]
  const float* cos_row; // cos/sin of angle delta * inv_freq[i], i < hd/2
  const float* sin_row;
  int B, n_kv, hd, lmax, len;
};

#ifndef CSM_ARGS_ONLY
template <typename KT>
__global__ __launch_bounds__(256) void kv_shift_kernel(KvShiftArgs a) {
  const int half = a.hd >> 1;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)a.B * a.n_kv * a.len * half;
  if (i >= n) return;
  const int d = (int)(i % half);
  const int t = (int)((i / half) % a.len);
  const size_t bj = i / ((size_t)half * a.len);
  const KT* kc = reinterpret_cast<const KT*>(a.kcache);
  const KT* vc = reinterpret_cast<const KT*>(a.vcache);
  KT* kt = reinterpret_cast<KT*>(a.ktmp);
  KT* vt = reinterpret_cast<KT*>(a.vtmp);
  const int d1 = d + half;
  const float k0 = to_f32(kc[((bj * (a.hd >> 2) + (d >> 2)) * a.lmax + t) * 4 + (d & 3)]);
  const float k1 = to_f32(kc[((bj * (a.hd >> 2) + (d1 >> 2)) * a.lmax + t) * 4 + (d1 & 3)]);
  const float c = a.cos_row[d], s = a.sin_row[d];
  store_kv(kt + ((bj * (a.hd >> 2) + (d >> 2)) * a.len + t) * 4 + (d & 3), k0 * c - k1 * s);
  store_kv(kt + ((bj * (a.hd >> 2) + (d1 >> 2)) * a.len + t) * 4 + (d1 & 3), k1 * c + k0 * s);
  vt[(bj * a.len + t) * a.hd + d] = vc[(bj * a.lmax + t) * a.hd + d];
  vt[(bj * a.len + t) * a.hd + d1] = vc[(bj * a.lmax + t) * a.hd + d1];
}
__global__ void add_ints_kernel(int* p, int n, int delta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += delta;
}

template <typename KT>
__global__ __launch_bounds__(256) void kv_convert_kernel(KvConvArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)a.B * a.n_kv * a.len * a.hd;
  if (i >= n) return;
  const int d = (int)(i % a.hd);
  const int t = (int)((i / a.hd) % a.len);
  const size_t bj = i / ((size_t)a.hd * a.len);    // b * n_kv + j
  KT* kc = reinterpret_cast<KT*>(a.kcache);
  KT* vc = reinterpret_cast<KT*>(a.vcache);
  const size_t ki = ((bj * (a.hd >> 2) + (d >> 2)) * a.lmax + t) * 4 + (d & 3);
  const size_t vi = (bj * a.lmax + t) * a.hd + d;
  if (a.to_engine) {
    store_kv(kc + ki, a.k_hf[i]);
    store_kv(vc + vi, a.v_hf[i]);
  } else {
    a.k_hf[i] = to_f32(kc[ki]);
    a.v_hf[i] = to_f32(vc[vi]);
  }
}

// Split-K gate/up GEMM of a SHORT prefill (GEPI_PARTIAL, gemm.h): a context of <= 128 rows has only 128 output tiles of
// 128 x 128 for the 16 384-wide gate/up projection -- half the chip, one k-step in flight per CU: 67 MB of weights at 2.4 TB/s.
// Split 4 ways over K it fills the chip twice; this launch sums the partial products in fixed order and applies SwiGLU
// (act_fn(gate) * up, modeling_llama.py:155-159; (gate, up) rows are interleaved in the packed matrix, so columns 2c, 2c + 1),
// writing what the GEMM's SWIGLU epilogue would have written: fp32 rows, or bf16 planes (one or three) for the down_proj GEMM.
// mxq != nullptr (F % 32 == 0; rows * F / 4 a multiple of 8: every 32-block is held by 8 whole lanes): MX-fp8 output, the recipe of rmsnorm_kernel.
__global__ __launch_bounds__(256) void swiglu_reduce_kernel(const float* part, int nsplit, size_t part_stride, int rows, int F,
                                                            float* out, int ldo, bf16_t* planes, size_t plane_stride, uint8_t* mxq, uint8_t* mxs) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;       // 4 output columns each
  const int qpr = F >> 2;
  if (q >= (size_t)rows * qpr) return;   // (whole groups of 8 lanes leave together)
  const size_t row = q / qpr;
  const int c = (int)(q - row * qpr) * 4;
  const float* pr = part + row * (size_t)(2 * F) + 2 * c;
  f32x4 a = *reinterpret_cast<const f32x4*>(pr), b = *reinterpret_cast<const f32x4*>(pr + 4);
  for (int sp = 1; sp < nsplit; ++sp) {
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(pr + (size_t)sp * part_stride), b2 = *reinterpret_cast<const f32x4*>(pr + (size_t)sp * part_stride + 4);
    a[0] += a2[0]; a[1] += a2[1]; a[2] += a2[2]; a[3] += a2[3];
    b[0] += b2[0]; b[1] += b2[1]; b[2] += b2[2]; b[3] += b2[3];
  }
  f32x4 h;
  h[0] = (a[0] / (1.f + __expf(-a[0]))) * a[1];
  h[1] = (a[2] / (1.f + __expf(-a[2]))) * a[3];
  h[2] = (b[0] / (1.f + __expf(-b[0]))) * b[1];
  h[3] = (b[2] / (1.f + __expf(-b[2]))) * b[3];
  if (mxq) {
    float m = fmaxf(fmaxf(fabsf(h[0]), fabsf(h[1])), fmaxf(fabsf(h[2]), fabsf(h[3])));
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    m = fmaxf(m, __shfl_xor(m, 4, 64));
    int eb = (int)((__float_as_uint(m) >> 23) & 0xff) - 8;
    eb = eb < 0 ? 0 : (eb > 254 ? 254 : eb);
    const float inv = __uint_as_float((uint32_t)(254 - eb) << 23);
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(h[0] * inv, -448.f), 448.f), fminf(fmaxf(h[1] * inv, -448.f), 448.f), pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(h[2] * inv, -448.f), 448.f), fminf(fmaxf(h[3] * inv, -448.f), 448.f), pk, true);
    *reinterpret_cast<uint32_t*>(mxq + row * (size_t)F + c) = (uint32_t)pk;
    if ((threadIdx.x & 7) == 0) mxs[row * (size_t)(F >> 5) + (c >> 5)] = (uint8_t)eb;
  } else if (planes) store_rowplanes4(planes + row * (size_t)F + c, plane_stride, h);
  else *reinterpret_cast<f32x4*>(out + row * (size_t)ldo + c) = h;
}

template <typename KT>
__global__ __launch_bounds__(256) void rope_scatter_kernel(RopeArgs a) {
  const int row = blockIdx.x;
  const int b = a.row_seq[row], pos = a.row_pos[row];
  const int rpos = a.rope_pos ? a.rope_pos[row] : pos;
  const int half = a.hd >> 1;
  const int nh = a.n_q + 2 * a.n_kv;
  const bool parts = a.nsplit > 1;
  const float* src = (parts ? a.part : a.qkv) + (size_t)row * nh * a.hd;
  auto ld = [&](int col) {   // the splits of an element are requested together, then added in split order
    float v = src[col];
    if (parts) {
      for (int s0 = 1; s0 < a.nsplit; s0 += 8) {
        float q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (s0 + j < a.nsplit) q[j] = src[(size_t)(s0 + j) * a.part_stride + col];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (s0 + j < a.nsplit) v += q[j];
      }
    }
    return v;
  };
  KT* kc = reinterpret_cast<KT*>(a.kcache);
  KT* vc = reinterpret_cast<KT*>(a.vcache);
  // gridDim.y workgroups share a row (short prefills summing split-K partials: rows alone would leave most CUs idle)
  for (int p = blockIdx.y * 256 + threadIdx.x; p < nh * half; p += 256 * gridDim.y) {
    const int head = p / half, i = p - head * half;
    if (head < a.n_q + a.n_kv) {
      const float v0 = ld(head * a.hd + i), v1 = ld(head * a.hd + i + half);
      const float c = a.cos_tab[(size_t)rpos * half + i], s = a.sin_tab[(size_t)rpos * half + i];
      // explicit contraction (what the compiler chose before it was spelled out), shared with the GEMM's RoPE epilogue
      // (gemm.h: rope_epilogue_row) so that the two are bitwise interchangeable
      const float o0 = __fmaf_rn(v0, c, -__fmul_rn(v1, s)), o1 = __fmaf_rn(v1, c, __fmul_rn(v0, s));
      if (head < a.n_q) {
        float* q = a.qbuf + (size_t)row * a.n_q * a.hd + head * a.hd;
        q[i] = __fmul_rn(o0, a.qscale);
        q[i + half] = __fmul_rn(o1, a.qscale);
      } else {
        const int j = head - a.n_q;
        const size_t base = (((size_t)b * a.n_kv + j) * (a.hd >> 2)) * a.lmax;
        store_kv(kc + (base + (size_t)(i >> 2) * a.lmax + pos) * 4 + (i & 3), o0);
        const int i2 = i + half;
        store_kv(kc + (base + (size_t)(i2 >> 2) * a.lmax + pos) * 4 + (i2 & 3), o1);
      }
    } else {
      const int j = head - a.n_q - a.n_kv;
      KT* vr = vc + (((size_t)b * a.n_kv + j) * a.lmax + pos) * a.hd;
      store_kv(vr + 2 * i, ld(head * a.hd + 2 * i));
      store_kv(vr + 2 * i + 1, ld(head * a.hd + 2 * i + 1));
    }
  }
}

#endif  // CSM_ARGS_ONLY

// ---------------------------------------------------------------------------------------------------
// K12 + K13: sample one codebook from a logits row and feed it back.
// Reference: sample_topk / _multinomial_sample_one_no_sync (modeling_csm.py:170-189):
//   x = logits / T; kth = k-th largest; x[x < kth] = -inf (ties at kth survive);
//   p = softmax(log_softmax(x)); q ~ Exp(1) per vocabulary entry; idx = argmax(p / q) (first max).
// Greedy (topk == 1, or T == 0 which the reference cannot do) = argmax, lowest index on exact ties.
// Then (modeling_csm.py:535,542 / 564-565): next decoder input = projection(audio_emb[tok + cb*V]),
// served from the precomputed fp32 table `proj_table`.
// One workgroup per row; wavefront shuffles for the reductions; 8-bit radix select for the k-th value.
// ---------------------------------------------------------------------------------------------------
struct SampleArgs {
  const float* logits;  // [rows][ldl]
  int ldl, V;
  float temperature;
  int topk;
  uint64_t seed;
  const uint64_t* rng;  // nullable: device-resident {seed, global index of row 0} -- overrides `seed`, so that a captured
                        // graph replays with a new seed / shard offset without re-capture
  const float* noise;  // nullable, [rows][noise_ld] (already offset to this codebook)
  size_t noise_ld;
  int cb, C, B;
  const int* frame_ptr;  // nullable (standalone call)
  int max_frames;
  int64_t* ring;          // [B][max_frames][C] nullable
  const int64_t* forced;  // [B][max_frames][C] nullable
  int32_t* idx_out;       // [rows] nullable
  const float* proj_table;  // [C*V][Hd] nullable
  int Hd;
  float* dec_x;         // [rows][Hd]
  float* logits_trace;  // [max_frames][B][C][V] nullable
  // batched decode: the fed-back row also goes out as MFMA B-operand planes (gemv.h: xplanes), folded with the norm
  // weight of the decoder's first layer, with its sum of squares in column 0 of the row's partial sums
  bf16_t* oplanes;      // nullable
  int pl1;              // decode_precision = bf16: one nearest-even plane
  const float* oln;
  float* oss;
  int oss_ld, oss_n;
  const int* row_done;  // nullable: rows flagged here are frozen -- they emit token 0 (per-row stop)
  // two-token first decoder pass (B == 1): the launch also copies copy_n floats copy_src -> copy_dst (the projected backbone
  // state, position 0's input, next to the sampled token's row, position 1's input)
  const float* copy_src;
  float* copy_dst;
  int copy_n;
  uint32_t* dbg;        // timeline probe slot (common.h TL_BEGIN), nullable
  int prio;             // 1 = s_setprio 3 at kernel entry
  int spin_ticks;       // TIMING ONLY: the launch idles this many 10 ns ticks before it starts (how a long launch in the chain affects the weight streamer)
  int legacy;           // TEST HOOK (engine option "sample_legacy"): top-k on the histogram / radix path of rounds 1-4 instead of sample_wave.h's selection
};

#ifndef CSM_ARGS_ONLY
// philox4x32_10 and f32_key: sample_wave.h (shared with the in-launch wave sampler)

// block-wide argmax with lowest-index tie-break; result broadcast through LDS
__device__ __forceinline__ int block_argmax(float v, int idx, float* s_val, int* s_idx) {
  wave_argmax(v, idx);
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) { s_val[tid >> 6] = v; s_idx[tid >> 6] = idx; }
  __syncthreads();
  float bv = s_val[0];
  int bi = s_idx[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) { bv = s_val[w]; bi = s_idx[w]; }
  }
  return bi;
}

__device__ __forceinline__ float block_sum(float v, float* s_val) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_val[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_val[0] + s_val[1] + s_val[2] + s_val[3];
}
__device__ __forceinline__ float block_max(float v, float* s_val) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_val[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(s_val[0], s_val[1]), fmaxf(s_val[2], s_val[3]));
}

// kernel-argument preload (gemv.h GEMV_HOT_PARAMS has the why): logits, frame counter, table, trace and copy destination pointers,
// the row pitch, V, a packed word (bit 0 prio, bit 1 spin_ticks > 0, bits 8.. topk) and the temperature
#define SMP_HOT_PARAMS const float* hlogits, const int* hframe, const float* htable, float* htrace, float* hcopy, int hldl, int hV, uint32_t hpk, float htemp
#define SMP_HOT_ARGS(a) (a).logits, (a).frame_ptr, (a).proj_table, (a).logits_trace, (a).copy_dst, (a).ldl, (a).V,                     \
  (uint32_t)(((a).prio ? 1u : 0u) | ((a).spin_ticks > 0 ? 2u : 0u) | ((uint32_t)((a).topk > 0xffffff ? 0xffffff : ((a).topk < 0 ? 0 : (a).topk)) << 8)), (a).temperature
__global__ __launch_bounds__(256) void sample_kernel(SMP_HOT_PARAMS, SampleArgs a) {
  a.logits = hlogits; a.frame_ptr = hframe; a.proj_table = htable; a.logits_trace = htrace; a.copy_dst = hcopy; a.ldl = hldl; a.V = hV;
  a.prio = (int)(hpk & 1u); a.topk = (int)(hpk >> 8); a.temperature = htemp;
  if (!(hpk & 2u)) a.spin_ticks = 0;
  extern __shared__ __attribute__((aligned(16))) float sx[];  // [V] scaled logits
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  __shared__ unsigned hist[256];
  __shared__ unsigned s_sel[4];
  __shared__ int s_wsum[4];
  __shared__ float s_mn[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  TL_BEGIN(a.dbg);
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  if (a.spin_ticks > 0) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < a.spin_ticks) __builtin_amdgcn_s_sleep(4);
  }
  const int V = a.V;
  const float* lg = a.logits + (size_t)row * a.ldl;
  const int f = a.frame_ptr ? *a.frame_ptr : 0;
  const bool greedy = (a.topk <= 1) || (a.temperature == 0.f);

  if (a.logits_trace) {
    float* tr = a.logits_trace + (((size_t)f * a.B + row) * a.C + a.cb) * V;
    for (int i = tid; i < V; i += 256) tr[i] = lg[i];
  }
  if (a.copy_dst && row == 0)
    for (int i = tid; i < a.copy_n; i += 256) a.copy_dst[i] = a.copy_src[i];

  int choice;
  if (!greedy && V >= WS_VMIN && V <= WS_VMAX && a.temperature > 0.f && !a.legacy) {
    // round 5: the barrier-light selection of sample_wave.h (values in registers, one histogram pass, LDS-only barriers): 9.8 -> ~5 us
    // per launch at top-k 50; term for term the arithmetic of the path below (tests: the two agree token for token)
    WaveSampleArgs w{};
    w.logits = lg; w.V = V; w.temperature = a.temperature; w.topk = a.topk; w.rng = a.rng; w.seed = a.seed;
    w.noise = a.noise ? a.noise + (size_t)row * a.noise_ld : nullptr;
    w.cb = a.cb; w.frame = f; w.row = row;
    choice = wg_sample_topk<true>(w, sx, tid);
  } else if (greedy) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (V <= 256 * 9) {
      // round 6: the thread's <= 9 logits are requested at once (a slot beyond V re-reads the last logit and is masked); the loop form
      // below issued one load per iteration behind the previous compare -- nine dependent round trips, most of the launch's 5.2 us
      float v[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) { const int i = tid + 256 * j; v[j] = lg[i < V ? i : V - 1]; }
#pragma unroll
      for (int j = 0; j < 9; ++j) { const int i = tid + 256 * j; if (i < V && v[j] > bv) { bv = v[j]; bi = i; } }   // ascending i: the lowest index of a tie wins, as below
    } else {
      for (int i = tid; i < V; i += 256) {
        const float v = lg[i];
        if (v > bv) { bv = v; bi = i; }
      }
    }
    choice = block_argmax(bv, bi, s_val, s_idx);
  } else {
    for (int i = tid; i < V; i += 256) sx[i] = lg[i] / a.temperature;
    __syncthreads();
    // ---- k-th largest value.  Fast path: ONE 256-bin histogram over the value range [min, max] (monotone linear
    // bins: logits spread evenly, so no LDS-atomic pile-up like on an exponent byte), a parallel suffix scan to the bin
    // that holds the k-th largest, then the exact value by rank counting among that bin's few entries in one wave.
    // Fallback (degenerate range, > 256 entries in the bin): 4 passes of 8-bit radix select on the monotone key.
    const int ktop = a.topk < V ? a.topk : V;
    uint32_t kth_fast = 0;
    bool fast = false;
    {
      float mxv = -INFINITY, mnv = INFINITY;
      for (int i = tid; i < V; i += 256) { const float v = sx[i]; mxv = fmaxf(mxv, v); mnv = fminf(mnv, v); }
      {   // max and min in one exchange (two barriers instead of four)
        mxv = wave_max(mxv);
        mnv = -wave_max(-mnv);
        __syncthreads();
        if ((tid & 63) == 0) { s_val[tid >> 6] = mxv; s_mn[tid >> 6] = mnv; }
        __syncthreads();
        mxv = fmaxf(fmaxf(s_val[0], s_val[1]), fmaxf(s_val[2], s_val[3]));
        mnv = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
      }
      const float bs = 255.99f / (mxv - mnv);
      if (mxv > mnv && bs < INFINITY) {   // block-uniform
        float* cand = sx + V;
        hist[tid] = 0;
        if (tid == 0) s_sel[2] = 0;
        __syncthreads();
        for (int i = tid; i < V; i += 256) atomicAdd(&hist[min(255, (int)((sx[i] - mnv) * bs))], 1u);
        __syncthreads();
        {
          const int lane = tid & 63, wv = tid >> 6;
          const int h = (int)hist[tid];
          int incl = h;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_down(incl, o, 64);
            if (lane + o < 64) incl += v;
          }
          if (lane == 0) s_wsum[wv] = incl;
          __syncthreads();
          int above = incl - h;
          for (int w = wv + 1; w < 4; ++w) above += s_wsum[w];
          if (above < ktop && above + h >= ktop) {
            s_sel[0] = (unsigned)tid;
            s_sel[1] = (unsigned)(ktop - above);
          }
        }
        __syncthreads();
        const int bsel = (int)s_sel[0], kin = (int)s_sel[1];
        for (int i = tid; i < V; i += 256) {
          const float v = sx[i];
          if (min(255, (int)((v - mnv) * bs)) == bsel) cand[atomicAdd(&s_sel[2], 1u)] = v;   // order is irrelevant: only a VALUE is derived
        }
        __syncthreads();
        const int nb = (int)s_sel[2];
        if (nb <= 256) {
          if (tid < 64) {
            for (int j = tid; j < nb; j += 64) {
              const float vj = cand[j];
              int gt = 0, ge = 0;
              for (int i = 0; i < nb; ++i) {
                const float vi = cand[i];
                gt += vi > vj;
                ge += vi >= vj;
              }
              if (gt < kin && kin <= ge) s_sel[3] = f32_key(vj);   // every lane that hits writes the same key
            }
          }
          __syncthreads();
          kth_fast = s_sel[3];
          fast = true;
        }
      }
    }
    uint32_t prefix = 0, pmask = 0;
    int krem = ktop;
    if (!fast) {
      hist[tid] = 0;
      __syncthreads();
    }
    for (int pass = fast ? -1 : 3; pass >= 0; --pass) {   // three barriers per pass: every thread re-zeroes its own bin after reading it
      for (int i0 = 0; i0 < V; i0 += 256) {
        const int i = i0 + tid;
        const uint32_t k = i < V ? f32_key(sx[i]) : 0u;
        bool live = i < V && (k & pmask) == prefix;
        const uint32_t bin = (k >> (pass * 8)) & 255u;
        if (pass == 3) {
          // sign + exponent byte: a handful of distinct values per wave -- one LDS atomic per distinct value
          // (a plain per-element atomic serialises ~2000 adds on 3-4 addresses)
          uint64_t todo = __ballot(live);
          while (todo) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
            const uint64_t same = __ballot(live && bin == lb);
            if ((int)(tid & 63) == leader) atomicAdd(&hist[lb], (unsigned)__builtin_popcountll(same));
            todo &= ~same;
          }
        } else if (live) {
          atomicAdd(&hist[bin], 1u);
        }
      }
      __syncthreads();
      {
        // digit d with  #(keys in higher digits) < krem <= that + hist[d] : suffix sums of the 256 bins in parallel
        // (a single thread walking the bins cost up to 255 dependent LDS reads per pass)
        const int lane = tid & 63, wv = tid >> 6;
        const int h = (int)hist[tid];
        hist[tid] = 0;  // for the next pass (its atomics start two barriers from here)
        int incl = h;   // sum over this wave's bins with index >= own
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_down(incl, o, 64);
          if (lane + o < 64) incl += v;
        }
        if (lane == 0) s_wsum[wv] = incl;
        __syncthreads();
        int above = incl - h;
        for (int w = wv + 1; w < 4; ++w) above += s_wsum[w];
        if (above < krem && above + h >= krem) {
          s_sel[0] = (unsigned)tid;
          s_sel[1] = (unsigned)(krem - above);
        }
      }
      __syncthreads();
      prefix |= s_sel[0] << (pass * 8);
      pmask |= 255u << (pass * 8);
      krem = (int)s_sel[1];
    }
    const uint32_t kth_key = fast ? kth_fast : prefix;  // key of the k-th largest value
    // ---- survivors (>= k-th value; ties included, like the reference's `x < kth` mask) compacted in index order:
    // every thread owns a contiguous chunk, one block-wide exclusive scan of the per-thread counts, then wave 0
    // finishes alone -- the two normalisations and the race touch ~k entries, not V, and need no more barriers.
    float* cv = sx + V;                                   // [V] survivor values
    int* ci = reinterpret_cast<int*>(sx + 2 * V);         // [V] survivor indices
    const int per = (V + 255) / 256;
    const int i0 = tid * per, i1 = min(V, i0 + per);
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += f32_key(sx[i]) >= kth_key;
    int incl = cnt;
    {
      const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      if (lane == 63) s_wsum[wv] = incl;
      __syncthreads();
      for (int w = 0; w < wv; ++w) incl += s_wsum[w];
    }
    int pos = incl - cnt;
    for (int i = i0; i < i1; ++i) {
      const float v = sx[i];
      if (f32_key(v) >= kth_key) { cv[pos] = v; ci[pos] = i; ++pos; }
    }
    if (tid == 255) s_sel[0] = (unsigned)incl;            // number of survivors
    __syncthreads();
    const int ns = (int)s_sel[0];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (tid < 64) {
      // log_softmax then softmax over the survivors (two normalisations, like the reference), then the race
      float mx = -INFINITY;
      for (int j = tid; j < ns; j += 64) mx = fmaxf(mx, cv[j]);
      mx = wave_max(mx);
      float se = 0.f;
      for (int j = tid; j < ns; j += 64) se += expf(cv[j] - mx);
      const float lse = logf(wave_sum(se));
      float m2 = -INFINITY;
      for (int j = tid; j < ns; j += 64) {
        const float lp = (cv[j] - mx) - lse;
        cv[j] = lp;
        m2 = fmaxf(m2, lp);
      }
      m2 = wave_max(m2);
      float s2 = 0.f;
      for (int j = tid; j < ns; j += 64) s2 += expf(cv[j] - m2);
      s2 = wave_sum(s2);
      for (int j = tid; j < ns; j += 64) {
        const int i = ci[j];
        const float p = expf(cv[j] - m2) / s2;
        float q;
        if (a.noise) {
          q = a.noise[(size_t)row * a.noise_ld + i];
        } else {
          uint32_t r[4];
          const uint64_t seed = a.rng ? a.rng[0] : a.seed;
          const uint32_t grow = (uint32_t)row + (a.rng ? (uint32_t)a.rng[1] : 0u);   // global row: shards draw distinct streams
          philox4x32_10((uint32_t)i, grow, (uint32_t)a.cb, (uint32_t)f, (uint32_t)seed, (uint32_t)(seed >> 32), r);
          const float u = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
          q = -logf(u);
        }
        const float v = p / q;
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
      }
    }
    choice = block_argmax(bv, bi, s_val, s_idx);
  }

  if (a.row_done && a.row_done[row]) choice = 0;   // per-row stop: a finished row stays silent
  int64_t feed = choice;
  const size_t slot = ((size_t)row * a.max_frames + f) * a.C + a.cb;
  if (a.forced) feed = a.forced[slot];
  if (tid == 0) {
    if (a.ring) a.ring[slot] = choice;
    if (a.idx_out) a.idx_out[row] = choice;
  }
  if (a.proj_table && a.cb < a.C - 1) {
    const float* src = a.proj_table + ((size_t)feed + (size_t)a.cb * V) * a.Hd;
    float* dst = a.dec_x + (size_t)row * a.Hd;
    float sq = 0.f;
    for (int k = tid * 4; k < a.Hd; k += 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + k);
      *reinterpret_cast<f32x4*>(dst + k) = v;
      if (a.oplanes && row < 128) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.oln + k);
        f32x4 t;
        t[0] = v[0] * w[0]; t[1] = v[1] * w[1]; t[2] = v[2] * w[2]; t[3] = v[3] * w[3];
        store_planes4(a.oplanes + (size_t)(row >> 4) * 3 * a.Hd * 16, (size_t)a.Hd * 16, k, row & 15, t, a.pl1 != 0);
        sq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    }
    if (a.oplanes && row < 128) {   // block-uniform
      sq = wave_sum(sq);
      __syncthreads();
      if ((tid & 63) == 0) s_val[tid >> 6] = sq;
      __syncthreads();
      if (tid < a.oss_n) a.oss[(size_t)row * a.oss_ld + tid] = tid == 0 ? (s_val[0] + s_val[1]) + (s_val[2] + s_val[3]) : 0.f;
    }
  }
  TL_END(5);
}
#endif  // CSM_ARGS_ONLY

// ---------------------------------------------------------------------------------------------------
// Training forward, labels branch (reference modeling_csm.py:367-465): cross-entropy over logits rows, the decoder
// inputs of the labelled frames.  Forward only: the loss values of the reference, no backward pass.
// ---------------------------------------------------------------------------------------------------
struct CeArgs {
  const float* logits;  // [rows][ld]
  int ld, V, rows;
  const int* labels;    // [rows]; < 0 = ignore_index (-100)
  float* row_loss;      // [rows]: logsumexp(logits) - logits[label], 0 for ignored rows
};
struct DecInArgs {
  const float* head_rows;   // [B*S][ld_head]: columns [0, Hd) = projection(final-normed backbone state) of every position
  int ld_head, Hd, P;       // P = decoder positions per frame (32)
  const int* prev_row;      // [frames] context row whose backbone state predicts the frame (t - 1, wrapping like the reference's index)
  const int* tok_row;       // [frames] context row of the frame itself (its audio tokens are the inputs of positions 1..)
  const int64_t* ids;       // [B*S][C+1]
  int C, V;
  const float* proj_table;  // [C*V][Hd] = projection(audio_embeddings)
  float* out;               // [frames * P][Hd]
};

#ifndef CSM_ARGS_ONLY
// one workgroup per row (nn.CrossEntropyLoss(ignore_index=-100) per element, modeling_csm.py:374-386, 458-463)
__global__ __launch_bounds__(256) void ce_rows_kernel(CeArgs a) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int lab = a.labels[row];
  if (lab < 0 || lab >= a.V) {   // ignored
    if (tid == 0) a.row_loss[row] = 0.f;
    return;
  }
  const float* lr = a.logits + (size_t)row * a.ld;
  float mx = -INFINITY;
  for (int i = tid; i < a.V; i += 256) mx = fmaxf(mx, lr[i]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
  for (int i = tid; i < a.V; i += 256) se += __expf(lr[i] - mx);
  se = wave_sum(se);
  if ((tid & 63) == 0) red[tid >> 6] = se;
  __syncthreads();
  if (tid == 0) a.row_loss[row] = (mx + logf((red[0] + red[1]) + (red[2] + red[3]))) - lr[lab];
}

// acc[0] += sum of the row losses, acc[1] += labelled rows: one workgroup, fixed order (deterministic), fp64 inside
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* row_loss, const int* labels, int V, int rows, double* acc) {
  __shared__ double ss[256];
  __shared__ double sc[256];
  const int tid = threadIdx.x;
  double s = 0.0, c = 0.0;
  for (int i = tid; i < rows; i += 256) {
    const int lab = labels[i];
    if (lab >= 0 && lab < V) { s += (double)row_loss[i]; c += 1.0; }
  }
  ss[tid] = s; sc[tid] = c;
  __syncthreads();
  for (int o = 128; o; o >>= 1) {
    if (tid < o) { ss[tid] += ss[tid + o]; sc[tid] += sc[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) { acc[0] += ss[0]; acc[1] += sc[0]; }
}

// out3 = (loss, backbone_loss, decoder_loss): means over the labelled elements (0 / 0 = NaN like torch for an all-ignored
// backbone target; the decoder term is 0 when no frame is fully labelled, modeling_csm.py:464-465)
__global__ void loss_finalize_kernel(const double* acc, int frames, float* out3) {
  const float bl = (float)(acc[0] / acc[1]);
  const float dl = frames > 0 ? (float)(acc[2] / acc[3]) : 0.f;
  out3[0] = bl + dl; out3[1] = bl; out3[2] = dl;
}

// grid = (frames, P): decoder input rows (modeling_csm.py:399-440)
__global__ __launch_bounds__(256) void dec_input_kernel(DecInArgs a) {
  const int f = blockIdx.x, p = blockIdx.y;
  const float* src;
  if (p == 0) {
    src = a.head_rows + (size_t)a.prev_row[f] * a.ld_head;
  } else {
    const int64_t tok = a.ids[(size_t)a.tok_row[f] * (a.C + 1) + (p - 1)];
    src = a.proj_table + ((size_t)tok + (size_t)(p - 1) * a.V) * a.Hd;
  }
  float* dst = a.out + ((size_t)f * a.P + p) * a.Hd;
  for (int i = threadIdx.x * 4; i < a.Hd; i += 1024) *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(src + i);
}
#endif  // CSM_ARGS_ONLY

