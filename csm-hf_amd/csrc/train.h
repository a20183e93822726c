// Backward pass of the reference's training objective (CSMModel.forward with labels, modeling_csm.py:367-465; consumer
// train.py:308-326: CSMTrainer.compute_loss -> loss.backward()).  SURVEY.md section 8 row f-3.
//
// Everything here is fp32 and built for CORRECTNESS, not speed: a training forward that keeps its activations, and the
// reverse pass over it.  The matrix products reuse the prefill GEMM of gemm.h (C = A W^T with fp32 A: exact three-plane bf16
// MFMA for bf16 weights, fp32 MFMA for fp32 operands) in three forms:
//   forward   Y  = X  W^T            launch_gemm(A = X,    W)
//   dX        dX = dY W              launch_gemm(A = dY,   W = W^T copy [K][N] in the weight dtype)        (train_impl.inc)
//   dW        dW += dY^T X           launch_gemm(A = dY^T, W = X^T as fp32 "weights", GEPI_RESID)         (accumulates)
// so the only new arithmetic is in the small kernels below: SwiGLU, RMSNorm backward, llama3-RoPE (forward / inverse),
// causal GQA attention forward + backward (one wavefront per (row, head); recomputation from the saved log-sum-exp, no
// atomics: deterministic), cross-entropy backward, transposes, the decoder-input gather / scatter and the embedding
// scatter-add (fp32 atomics: the one non-deterministic summation order of the pass, as in every framework's embedding backward).
// Each kernel cites the forward lines it differentiates.
#pragma once
#include "common.h"

struct TrAttnArgs {
  float* qkv;            // [R][nqkv]: q heads | k heads | v heads, q and k already rotated (rope_train_kernel)
  const float* lse;      // [R][n_q]  log-sum-exp of the scaled scores (written by the forward)
  float* lse_out;
  float* out;            // forward: [R][n_q * hd]
  const float* dout;     // backward: [R][n_q * hd]
  float* dqkv;           // backward: [R][nqkv] (gradients w.r.t. the ROTATED q, k and v)
  float* dsum;           // backward scratch [R][n_q]: D_i = sum_d dO_id O_id
  const float* o;        // backward: the forward output
  int R, S, n_q, n_kv, hd;
  float scale;
  const int* kv_start;   // nullable, per sequence: keys below it are pads (a pad query sees itself only), oracle semantics
};

#ifndef CSM_ARGS_ONLY
__device__ __forceinline__ float tr_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float tr_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// act = silu(g) * u (modeling_llama.py:174-176); gu rows hold (gate, up) interleaved like the packed wgu matrix
__global__ void tr_swiglu_fwd_kernel(const float* gu, float* act, size_t n) {   // n = rows * F
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = gu[2 * i], u = gu[2 * i + 1];
  act[i] = (g / (1.f + __expf(-g))) * u;
}
__global__ void tr_swiglu_bwd_kernel(const float* gu, const float* dact, float* dgu, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = gu[2 * i], u = gu[2 * i + 1], d = dact[i];
  const float sg = 1.f / (1.f + __expf(-g));
  dgu[2 * i] = d * u * (sg * (1.f + g * (1.f - sg)));
  dgu[2 * i + 1] = d * (g * sg);
}

// y = w * x * r, r = rsqrt(mean(x^2) + eps) (modeling_llama.py:62-67).  dx = r (w dy) - x r^3 mean(w dy x);
// gw[row][k] = dy x r (summed over rows by tr_colsum_kernel into dw).  One workgroup per row.
__global__ __launch_bounds__(256) void tr_rmsnorm_bwd_kernel(const float* x, const float* w, const float* dy, int H, float eps,
                                                             float* dx, int accumulate, float* gw) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (size_t)row * H;
  const float* dr = dy + (size_t)row * H;
  float ss = 0.f, dot = 0.f;
  for (int k = tid; k < H; k += 256) {
    const float xv = xr[k];
    ss += xv * xv;
    dot += w[k] * dr[k] * xv;
  }
  ss = tr_wave_sum(ss);
  dot = tr_wave_sum(dot);
  if ((tid & 63) == 0) { red[tid >> 6] = ss; red[4 + (tid >> 6)] = dot; }
  __syncthreads();
  const float r = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
  const float c = (red[4] + red[5] + red[6] + red[7]) * r * r * r / (float)H;
  for (int k = tid; k < H; k += 256) {
    const float xv = xr[k], d = dr[k];
    const float v = r * w[k] * d - xv * c;
    float* o = dx + (size_t)row * H + k;
    *o = accumulate ? *o + v : v;
    gw[(size_t)row * H + k] = d * xv * r;
  }
}
// dst[c] += sum over rows of src[row][c], rows in ascending order (deterministic)
__global__ void tr_colsum_kernel(const float* src, int rows, int cols, float* dst) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += src[(size_t)r * cols + c];
  dst[c] += s;
}

// llama3 RoPE on the q and k heads of a [R][nqkv] buffer, in place (modeling_llama.py:130-160: half-split pairing,
// x' = x cos + rotate_half(x) sin); inverse = 1 applies the transposed rotation (the backward of the forward one)
__global__ void tr_rope_kernel(float* qkv, int R, int n_rot_heads, int nqkv, int hd, const int* row_pos, const float* cos_tab,
                               const float* sin_tab, int inverse) {
  const int half = hd >> 1;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)R * n_rot_heads * half;
  if (i >= n) return;
  const int d = (int)(i % half);
  const int h = (int)((i / half) % n_rot_heads);
  const int row = (int)(i / ((size_t)half * n_rot_heads));
  const int pos = row_pos[row];
  const float c = cos_tab[(size_t)pos * half + d], s = inverse ? -sin_tab[(size_t)pos * half + d] : sin_tab[(size_t)pos * half + d];
  float* p = qkv + (size_t)row * nqkv + h * hd + d;
  const float v0 = p[0], v1 = p[half];
  p[0] = v0 * c - v1 * s;
  p[half] = v1 * c + v0 * s;
}

// Causal GQA attention over whole sequences (sdpa_attention.py:97-163 as called by LlamaAttention, modeling_llama.py:254-281):
// row = seq * S + i; query (row, head h) sees keys j <= i of its sequence with j >= kv_start[seq] or j == i.
// One wavefront per (row, head): lanes over keys for the scores, lanes over the head dimension for the weighted sums.
__global__ __launch_bounds__(64) void tr_attn_fwd_kernel(TrAttnArgs a) {
  extern __shared__ float tr_p[];   // [S]
  const int row = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int seq = row / a.S, i = row - seq * a.S;
  const int nqkv = (a.n_q + 2 * a.n_kv) * a.hd, kvh = h / (a.n_q / a.n_kv);
  const int lo = a.kv_start ? a.kv_start[seq] : 0;
  const float* q = a.qkv + (size_t)row * nqkv + h * a.hd;
  const float* kb = a.qkv + (size_t)seq * a.S * nqkv + (a.n_q + kvh) * a.hd;
  const float* vb = kb + a.n_kv * a.hd;
  float mx = -INFINITY;
  for (int j = lane; j <= i; j += 64) {
    float s = -INFINITY;
    if (j >= lo || j == i) {
      const float* k = kb + (size_t)j * nqkv;
      float acc = 0.f;
      for (int d = 0; d < a.hd; ++d) acc += q[d] * k[d];
      s = acc * a.scale;
    }
    tr_p[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = tr_wave_max(mx);
  float l = 0.f;
  for (int j = lane; j <= i; j += 64) {
    const float p = tr_p[j] == -INFINITY ? 0.f : __expf(tr_p[j] - mx);
    tr_p[j] = p;
    l += p;
  }
  l = tr_wave_sum(l);
  __syncthreads();
  const float inv = 1.f / l;
  for (int d = lane; d < a.hd; d += 64) {
    float acc = 0.f;
    for (int j = 0; j <= i; ++j) acc += tr_p[j] * vb[(size_t)j * nqkv + d];
    a.out[(size_t)row * a.n_q * a.hd + h * a.hd + d] = acc * inv;
  }
  if (lane == 0) a.lse_out[(size_t)row * a.n_q + h] = mx + __logf(l);
}
// dQ and D_i = sum_d dO O per (row, head)
__global__ __launch_bounds__(64) void tr_attn_bwd_q_kernel(TrAttnArgs a) {
  extern __shared__ float tr_p[];   // ds [S]
  const int row = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int seq = row / a.S, i = row - seq * a.S;
  const int nqkv = (a.n_q + 2 * a.n_kv) * a.hd, kvh = h / (a.n_q / a.n_kv);
  const int lo = a.kv_start ? a.kv_start[seq] : 0;
  const float* q = a.qkv + (size_t)row * nqkv + h * a.hd;
  const float* kb = a.qkv + (size_t)seq * a.S * nqkv + (a.n_q + kvh) * a.hd;
  const float* vb = kb + a.n_kv * a.hd;
  const float* dO = a.dout + (size_t)row * a.n_q * a.hd + h * a.hd;
  const float* O = a.o + (size_t)row * a.n_q * a.hd + h * a.hd;
  float D = 0.f;
  for (int d = lane; d < a.hd; d += 64) D += dO[d] * O[d];
  D = tr_wave_sum(D);
  const float L = a.lse[(size_t)row * a.n_q + h];
  for (int j = lane; j <= i; j += 64) {
    float ds = 0.f;
    if (j >= lo || j == i) {
      const float* k = kb + (size_t)j * nqkv;
      const float* v = vb + (size_t)j * nqkv;
      float s = 0.f, dp = 0.f;
      for (int d = 0; d < a.hd; ++d) { s += q[d] * k[d]; dp += dO[d] * v[d]; }
      const float p = __expf(s * a.scale - L);
      ds = p * (dp - D);
    }
    tr_p[j] = ds;
  }
  __syncthreads();
  for (int d = lane; d < a.hd; d += 64) {
    float acc = 0.f;
    for (int j = 0; j <= i; ++j) acc += tr_p[j] * kb[(size_t)j * nqkv + d];
    a.dqkv[(size_t)row * nqkv + h * a.hd + d] = acc * a.scale;
  }
  if (lane == 0) a.dsum[(size_t)row * a.n_q + h] = D;
}
// dK_j, dV_j per (key row, kv head): sums over the G query heads of the group and the queries i >= j of the sequence
__global__ __launch_bounds__(64) void tr_attn_bwd_kv_kernel(TrAttnArgs a) {
  const int row = blockIdx.x, kvh = blockIdx.y, lane = threadIdx.x;
  const int seq = row / a.S, j = row - seq * a.S;
  const int nqkv = (a.n_q + 2 * a.n_kv) * a.hd, G = a.n_q / a.n_kv;
  const int lo = a.kv_start ? a.kv_start[seq] : 0;
  const float* k = a.qkv + (size_t)row * nqkv + (a.n_q + kvh) * a.hd;
  const float* v = k + a.n_kv * a.hd;
  // lanes own head dimensions d = lane, lane + 64
  const int d0 = lane, d1 = lane + 64;
  const bool h1 = d1 < a.hd, h0 = d0 < a.hd;
  const float k0 = h0 ? k[d0] : 0.f, k1 = h1 ? k[d1] : 0.f, v0 = h0 ? v[d0] : 0.f, v1 = h1 ? v[d1] : 0.f;
  float dk0 = 0.f, dk1 = 0.f, dv0 = 0.f, dv1 = 0.f;
  for (int i = j; i < a.S; ++i) {
    if (!(j >= lo || j == i)) continue;   // key j is a pad: only its own query sees it
    const size_t qrow = (size_t)seq * a.S + i;
    for (int g = 0; g < G; ++g) {
      const int h = kvh * G + g;
      const float* q = a.qkv + qrow * nqkv + h * a.hd;
      const float* dO = a.dout + qrow * a.n_q * a.hd + h * a.hd;
      const float q0 = h0 ? q[d0] : 0.f, q1 = h1 ? q[d1] : 0.f, o0 = h0 ? dO[d0] : 0.f, o1 = h1 ? dO[d1] : 0.f;
      const float s = tr_wave_sum(q0 * k0 + q1 * k1), dp = tr_wave_sum(o0 * v0 + o1 * v1);
      const float p = __expf(s * a.scale - a.lse[qrow * a.n_q + h]);
      const float ds = p * (dp - a.dsum[qrow * a.n_q + h]) * a.scale;
      dv0 += p * o0; dv1 += p * o1;
      dk0 += ds * q0; dk1 += ds * q1;
    }
  }
  float* dk = a.dqkv + (size_t)row * nqkv + (a.n_q + kvh) * a.hd;
  float* dv = dk + a.n_kv * a.hd;
  if (h0) { dk[d0] = dk0; dv[d0] = dv0; }
  if (h1) { dk[d1] = dk1; dv[d1] = dv1; }
}

// d logits of nn.CrossEntropyLoss(ignore_index=-100, reduction="mean") (modeling_csm.py:374-386, 458-463):
// (softmax(logits) - onehot(label)) * scale for labelled rows, 0 otherwise; columns V..ldo-1 are zeroed (GEMM padding)
__global__ __launch_bounds__(256) void tr_ce_bwd_kernel(const float* logits, int ld, int V, const int* labels, float scale,
                                                        float* dl, int ldo) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int lab = labels[row];
  float* o = dl + (size_t)row * ldo;
  if (lab < 0) {
    for (int i = tid; i < ldo; i += 256) o[i] = 0.f;
    return;
  }
  const float* lg = logits + (size_t)row * ld;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 256) mx = fmaxf(mx, lg[i]);
  mx = tr_wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = tid; i < V; i += 256) s += __expf(lg[i] - mx);
  s = tr_wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float inv = scale / (red[0] + red[1] + red[2] + red[3]);
  for (int i = tid; i < ldo; i += 256) {
    float v = 0.f;
    if (i < V) v = __expf(lg[i] - mx) * inv - (i == lab ? scale : 0.f);
    o[i] = v;
  }
}

// dst[c][r] = src[r][c] (r < rows, c < cols; src row stride lds; dst row stride ldd >= rows, columns rows..ldd-1 zeroed)
template <typename T>
__global__ __launch_bounds__(256) void tr_transpose_kernel(const T* src, int rows, int cols, int lds, T* dst, int ldd) {
  __shared__ T tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < rows && c < cols) ? src[(size_t)r * lds + c] : T(0);
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < cols && r < ldd) dst[(size_t)c * ldd + r] = tile[tx][k];
  }
}

// decoder inputs of the labelled frames before the projection (modeling_csm.py:402-440): row (f, 0) = final-normed backbone
// state of the frame's predecessor, row (f, p >= 1) = audio embedding of codebook p-1's token; fp32 [frames * P][Hb]
template <typename WT>
__global__ void tr_dec_gather_kernel(const float* hb, const WT* audio_emb, const int64_t* ids, const int* prev_row, const int* tok_row,
                                     int frames, int P, int C, int V, int Hb, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)frames * P * Hb;
  if (i >= n) return;
  const int d = (int)(i % Hb);
  const int p = (int)((i / Hb) % P);
  const int f = (int)(i / ((size_t)Hb * P));
  float v;
  if (p == 0) v = hb[(size_t)prev_row[f] * Hb + d];
  else {
    const int64_t tok = ids[(size_t)tok_row[f] * (C + 1) + (p - 1)];
    v = to_f32(audio_emb[((size_t)tok + (size_t)(p - 1) * V) * Hb + d]);
  }
  out[i] = v;
}
// the backward of that gather: d hb[prev] += dE(f, 0) (every predecessor row is distinct: plain add); d audio_emb += (atomic)
__global__ void tr_dec_scatter_kernel(const float* dE, const int64_t* ids, const int* prev_row, const int* tok_row, int frames, int P,
                                      int C, int V, int Hb, float* dhb, float* d_audio_emb) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)frames * P * Hb;
  if (i >= n) return;
  const int d = (int)(i % Hb);
  const int p = (int)((i / Hb) % P);
  const int f = (int)(i / ((size_t)Hb * P));
  if (p == 0) dhb[(size_t)prev_row[f] * Hb + d] += dE[i];
  else {
    const int64_t tok = ids[(size_t)tok_row[f] * (C + 1) + (p - 1)];
    atomicAdd(d_audio_emb + ((size_t)tok + (size_t)(p - 1) * V) * Hb + d, dE[i]);
  }
}
// backward of the frame embedding sum (modeling_csm.py:247-282, 327-334): every unmasked token's table row receives the row's gradient
__global__ void tr_embed_bwd_kernel(const float* dx, const int64_t* ids, const uint8_t* mask, int rows, int C, int V, int Hb,
                                    float* d_text, float* d_audio) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)rows * (C + 1) * Hb;
  if (i >= n) return;
  const int d = (int)(i % Hb);
  const int c = (int)((i / Hb) % (C + 1));
  const int row = (int)(i / ((size_t)Hb * (C + 1)));
  if (mask && !mask[(size_t)row * (C + 1) + c]) return;
  const int64_t tok = ids[(size_t)row * (C + 1) + c];
  const float g = dx[(size_t)row * Hb + d];
  if (c == C) atomicAdd(d_text + (size_t)tok * Hb + d, g);
  else atomicAdd(d_audio + ((size_t)tok + (size_t)c * V) * Hb + d, g);
}
__global__ void tr_add_kernel(float* dst, const float* src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
// rows of `src` (row stride lds, `cols` columns) gathered at a fixed column offset into a dense [rows][cols] buffer and back
__global__ void tr_copy2d_kernel(const float* src, size_t lds, float* dst, size_t ldd, int rows, int cols, int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  float* o = dst + (size_t)r * ldd + c;
  const float v = src[(size_t)r * lds + c];
  *o = accumulate ? *o + v : v;
}
#endif  // CSM_ARGS_ONLY

// launchers (train.hip)
int tr_launch_swiglu(hipStream_t st, const float* gu, float* act, const float* dact, float* dgu, size_t n);   // dgu ? backward : forward
int tr_launch_rmsnorm_bwd(hipStream_t st, const float* x, const float* w, const float* dy, int rows, int H, float eps, float* dx,
                          int accumulate, float* gw_scratch, float* dw);
int tr_launch_rope(hipStream_t st, float* qkv, int R, int n_rot_heads, int nqkv, int hd, const int* row_pos, const float* cos_tab,
                   const float* sin_tab, int inverse);
int tr_launch_attn_fwd(hipStream_t st, const TrAttnArgs& a);
int tr_launch_attn_bwd(hipStream_t st, const TrAttnArgs& a);
int tr_launch_ce_bwd(hipStream_t st, const float* logits, int ld, int V, const int* labels, int rows, float scale, float* dl, int ldo);
int tr_launch_transpose_f32(hipStream_t st, const float* src, int rows, int cols, int lds, float* dst, int ldd);
int tr_launch_transpose_w(hipStream_t st, int wdtype, const void* src, int rows, int cols, void* dst, int ldd);
int tr_launch_dec_gather(hipStream_t st, int wdtype, const float* hb, const void* audio_emb, const int64_t* ids, const int* prev_row,
                         const int* tok_row, int frames, int P, int C, int V, int Hb, float* out);
int tr_launch_dec_scatter(hipStream_t st, const float* dE, const int64_t* ids, const int* prev_row, const int* tok_row, int frames,
                          int P, int C, int V, int Hb, float* dhb, float* d_audio_emb);
int tr_launch_embed_bwd(hipStream_t st, const float* dx, const int64_t* ids, const uint8_t* mask, int rows, int C, int V, int Hb,
                        float* d_text, float* d_audio);
int tr_launch_add(hipStream_t st, float* dst, const float* src, size_t n);
int tr_launch_copy2d(hipStream_t st, const float* src, size_t lds, float* dst, size_t ldd, int rows, int cols, int accumulate);
