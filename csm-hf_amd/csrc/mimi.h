// Mimi decode: element-wise / gather / attention kernels around the exact-fp32 MFMA GEMM
// (gemm.h: gemm_f32mfma_kernel), which carries every convolution and linear of the codec's decode path.
// Replaces `audio_tokenizer.decode(codes)` of the reference's call site (/root/reference/README.md:114-118; codec =
// third-party moshi==0.2.2, architecture as in transformers 5.15 models/mimi/modeling_mimi.py, cited per kernel).
// Activations are channels-last ([time][channels], fp32): a causal convolution's k shifted input rows are then ONE contiguous
// operand row of K = k * C_in, so conv1d = GEMM with lda = C_in < K over a buffer that starts with k - 1 zero rows, and a
// stride-r transposed convolution (kernel 2r) is one GEMM with K = 2 C_in and N = r * C_out whose output row q holds the r
// output positions r q .. r q + r - 1.
#pragma once
#include "common.h"

#ifdef CSM_MIMI_KERNELS
// split RVQ decode (modeling_mimi.py:1004-1007, 1070-1082, 1128-1139): out[t][0:D] = sum of the semantic codebooks' rows,
// out[t][D:2D] = sum of the acoustic codebooks' rows (the two 1x1 output projections follow as one GEMM with K = 2 D)
__global__ __launch_bounds__(256) void mimi_rvq_gather_kernel(const int64_t* codes, const float* embed, int n_q, int n_sem, int csize,
                                                              int D, int T, float* out) {
  const int t = blockIdx.x;
  for (int d = threadIdx.x; d < 2 * D; d += 256) {
    const int acoustic = d >= D, dd = acoustic ? d - D : d;
    float s = 0.f;
    for (int k = acoustic ? n_sem : 0; k < (acoustic ? n_q : n_sem); ++k) {
      const int64_t c = codes[(size_t)k * T + t];
      s += embed[((size_t)k * csize + c) * D + dd];
    }
    out[(size_t)t * 2 * D + d] = s;
  }
}

// depthwise transposed convolution, kernel 2 s, stride s, causal trim (modeling_mimi.py:399-405, MimiModel.upsample):
// out[s q + p][c] = x[q][c] w[c][p] + x[q - 1][c] w[c][p + s]
__global__ __launch_bounds__(256) void mimi_upsample_kernel(const float* x, const float* w, int L, int C, int s, const float* prev, float* out) {
  // prev (nullable): the row before x[0] (streaming: the last frame of the previous call)
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)L * s * C) return;
  const int c = (int)(i % C);
  const size_t lo = i / C;
  const int q = (int)(lo / s), p = (int)(lo % s);
  float v = x[(size_t)q * C + c] * w[(size_t)c * 2 * s + p];
  if (q > 0) v += x[(size_t)(q - 1) * C + c] * w[(size_t)c * 2 * s + p + s];
  else if (prev) v += prev[c] * w[(size_t)c * 2 * s + p + s];
  out[i] = v;
}

// split-K GEMM (gemm_f32mfma_kernel, GEPI_PARTIAL): C[r][n] = sum over the ks partial products [ks][R][N], fixed order
__global__ __launch_bounds__(256) void mimi_splitk_reduce_kernel(const float* part, int ks, size_t stride, float* C, int ldc, int N, size_t total4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(part + i * 4);
  for (int s = 1; s < ks; ++s) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(part + (size_t)s * stride + i * 4);
    v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
  }
  const size_t e = i * 4, r = e / N;
  *reinterpret_cast<f32x4*>(C + r * ldc + (e - r * N)) = v;
}

// nn.LayerNorm with bias (modeling_mimi.py:737-738): one workgroup per row
__global__ __launch_bounds__(256) void mimi_layernorm_kernel(const float* x, const float* w, const float* b, int C, float eps, float* out) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (size_t)row * C;
  float s = 0.f;
  for (int i = tid; i < C; i += 256) s += xr[i];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C;
  __syncthreads();
  float v = 0.f;
  for (int i = tid; i < C; i += 256) { const float d = xr[i] - mean; v += d * d; }
  v = wave_sum(v);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const float rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)C + eps);
  for (int i = tid; i < C; i += 256) out[(size_t)row * C + i] = (xr[i] - mean) * rstd * w[i] + b[i];
}

// rotary embedding, default rope, rotate_half convention (modeling_mimi.py:524-600), in place on the q and k parts of a
// [L][3 A] projection buffer; position = pos0 + row (pos0 > 0 when a stream continues)
__global__ __launch_bounds__(256) void mimi_rope_kernel(float* qkv, int L, int heads, int hd, float theta, int pos0) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int half = hd / 2, A = heads * hd;
  if (i >= (size_t)L * 2 * heads * half) return;
  const int f = (int)(i % half);
  const size_t r = i / half;
  const int h = (int)(r % (2 * heads));      // q heads, then k heads
  const int pos = (int)(r / (2 * heads));
  float* p = qkv + (size_t)pos * 3 * A + (size_t)h * hd;
  const float ang = (float)(pos0 + pos) * powf(theta, -2.f * (float)f / (float)hd);
  const float c = cosf(ang), s = sinf(ang);
  const float a = p[f], b = p[f + half];
  p[f] = a * c - b * s;
  p[f + half] = b * c + a * s;
}

// the rotated K and the V rows of the new positions appended to the layer's history [rows][2 A] (K | V)
__global__ __launch_bounds__(256) void mimi_kv_append_kernel(const float* qkv, float* hist, size_t L, int A) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= L * 2 * A) return;
  const size_t r = i / (2 * A);
  const int c = (int)(i % (2 * A));
  hist[i] = qkv[r * 3 * A + A + c];
}
// rows [from, from + n) of src to the front of dst (the part of the history the next call can still see)
__global__ __launch_bounds__(256) void mimi_rows_copy_kernel(const float* src, float* dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// causal sliding-window attention (modeling_mimi.py:687-726, create_sliding_window_causal_mask): the query at history
// index nh + i sees the keys max(0, nh + i - window + 1) .. nh + i of hist [nh + L][2 A] (nh = positions kept from earlier
// calls of a stream, 0 otherwise).  grid = (L, heads), 64 threads, fp32 softmax.
__global__ __launch_bounds__(64) void mimi_attn_kernel(const float* qkv, const float* hist, int nh, int heads, int hd, int window, float* out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // q[hd] | scores[window]
  float* qs = sm;
  float* sc = sm + hd;
  const int i = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int A = heads * hd;
  const int hi = nh + i;
  const int j0 = max(0, hi - window + 1), n = hi - j0 + 1;
  const float* q = qkv + (size_t)i * 3 * A + (size_t)h * hd;
  const float scale = rsqrtf((float)hd);
  for (int d = tid; d < hd; d += 64) qs[d] = q[d] * scale;
  __syncthreads();
  float mx = -INFINITY;
  for (int j = tid; j < n; j += 64) {   // one key per thread and round: its hd-wide row by 16-byte loads against q in LDS
    const float* k = hist + (size_t)(j0 + j) * 2 * A + (size_t)h * hd;
    float s0 = 0.f, s1 = 0.f;
    for (int d = 0; d < hd; d += 8) {
      const f32x4 k0 = *reinterpret_cast<const f32x4*>(k + d), k1 = *reinterpret_cast<const f32x4*>(k + d + 4);
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(qs + d), q1 = *reinterpret_cast<const f32x4*>(qs + d + 4);
      s0 += (k0[0] * q0[0] + k0[1] * q0[1]) + (k0[2] * q0[2] + k0[3] * q0[3]);
      s1 += (k1[0] * q1[0] + k1[1] * q1[1]) + (k1[2] * q1[2] + k1[3] * q1[3]);
    }
    const float sv = s0 + s1;
    sc[j] = sv;
    mx = fmaxf(mx, sv);
  }
  mx = wave_max(mx);
  float se = 0.f;
  for (int j = tid; j < n; j += 64) { const float p = expf(sc[j] - mx); sc[j] = p; se += p; }
  se = wave_sum(se);
  __syncthreads();
  const float inv = 1.f / se;
  for (int d = tid; d < hd; d += 64) {
    float o0 = 0.f, o1 = 0.f;
    const float* v = hist + (size_t)j0 * 2 * A + A + (size_t)h * hd + d;
    int j = 0;
    for (; j + 1 < n; j += 2) {
      o0 = fmaf(sc[j], v[(size_t)j * 2 * A], o0);
      o1 = fmaf(sc[j + 1], v[(size_t)(j + 1) * 2 * A], o1);
    }
    if (j < n) o0 = fmaf(sc[j], v[(size_t)j * 2 * A], o0);
    out[(size_t)i * A + (size_t)h * hd + d] = (o0 + o1) * inv;
  }
}

// x[r][c] += scale[c] * y[r][c]   (MimiLayerScale + residual, modeling_mimi.py:506-507, 765, 771)
__global__ __launch_bounds__(256) void mimi_scale_add_kernel(float* x, const float* y, const float* scale, size_t n, int C) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] += scale[i % C] * y[i];
}
// exact (erf) GELU in place (ACT2FN["gelu"], modeling_mimi.py:606)
__global__ __launch_bounds__(256) void mimi_gelu_kernel(float* x, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float v = x[i]; x[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
}
// out[r][c] = act(src[r * lds + c] + bias[c % nb] (+ res[r][c])), c < C: takes a GEMM result with padded columns to its
// exact-width channels-last buffer; act: 0 none, 1 ELU (nn.ELU, alpha 1)
__global__ __launch_bounds__(256) void mimi_bias_act_kernel(const float* src, int lds, const float* bias, int nb, const float* res, int C,
                                                            size_t rows, int act, float* out, int ldo) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const size_t r = i / C;
  const int c = (int)(i % C);
  float v = src[r * lds + c] + (bias ? bias[c % nb] : 0.f);
  if (res) v += res[r * C + c];
  if (act == 1) v = v > 0.f ? v : expm1f(v);
  out[r * ldo + c] = v;
}
// dst[r][c] = ELU(src[r][c]) (or a plain copy): the padded input buffer of the next convolution
__global__ __launch_bounds__(256) void mimi_elu_copy_kernel(const float* src, float* dst, size_t n, int elu) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float v = src[i]; dst[i] = (elu && v <= 0.f) ? expm1f(v) : v; }
}
// last convolution, one output channel (modeling_mimi.py:955): audio[l] = b + sum_j sum_c w[j * C + c] * xa[(l + j) * C + c]
// over the ELU'd, front-padded input xa (k - 1 zero rows first)
__global__ __launch_bounds__(256) void mimi_last_conv_kernel(const float* xa, const float* w, const float* b, int C, int k, size_t L, float* audio) {
  // a wave takes 16 consecutive outputs; its lanes walk the contiguous k * C operand row of each (coalesced), the weights
  // stay in registers (k * C <= 64 * 4) or are re-read from L1 otherwise
  const int lane = threadIdx.x & 63;
  const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int K = k * C;
  float wr[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) wr[u] = lane + 64 * u < K ? w[lane + 64 * u] : 0.f;
  const float bias = b[0];
  for (int o = 0; o < 16; ++o) {
    const size_t l = wv * 16 + o;
    if (l >= L) return;
    const float* p = xa + l * C;
    float s = 0.f;
    if (K <= 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (lane + 64 * u < K) s = fmaf(wr[u], p[lane + 64 * u], s);
    } else {
      for (int i = lane; i < K; i += 64) s = fmaf(w[i], p[i], s);
    }
    s = wave_sum(s);
    if (lane == 0) audio[l] = s + bias;
  }
}
// ---- stream GROUPS (round 3: csm_mimi_streams_*): S streams advance in lockstep, T frames each per call, every launch
// covers all of them -- the one-frame streaming call is launch-bound (~120 launches of a few rows each), so S streams in the
// same launches cost little more than one.  Compact activations are [S * L][C] (row = s * L + l); a convolution's input is
// [S][PADR + L][C] (every stream's own left context in front of its rows); per-stream state (rotary position, history rows,
// "has a previous frame", history rows to keep) comes from small device arrays, so streams restarted at different times
// (continuous batching) share a call.
__global__ __launch_bounds__(256) void mimi_rvq_gather_g_kernel(const int64_t* codes, const float* embed, int n_q, int n_sem, int csize,
                                                                int D, int T, float* out) {
  const int t = blockIdx.x, s = blockIdx.y;
  const int64_t* cb = codes + (size_t)s * n_q * T;
  for (int d = threadIdx.x; d < 2 * D; d += 256) {
    const int acoustic = d >= D, dd = acoustic ? d - D : d;
    float v = 0.f;
    for (int k = acoustic ? n_sem : 0; k < (acoustic ? n_q : n_sem); ++k) {
      const int64_t c = cb[(size_t)k * T + t];
      v += embed[((size_t)k * csize + c) * D + dd];
    }
    out[((size_t)s * T + t) * 2 * D + d] = v;
  }
}
__global__ __launch_bounds__(256) void mimi_upsample_g_kernel(const float* x, const float* w, int T, int C, int st, const float* prev,
                                                              const int* hasprev, int S, float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t per = (size_t)T * st * C;
  if (i >= per * S) return;
  const int s = (int)(i / per);
  const size_t j = i - (size_t)s * per;
  const int c = (int)(j % C);
  const size_t lo = j / C;
  const int q = (int)(lo / st), p = (int)(lo % st);
  const float* xs = x + (size_t)s * T * C;
  float v = xs[(size_t)q * C + c] * w[(size_t)c * 2 * st + p];
  if (q > 0) v += xs[(size_t)(q - 1) * C + c] * w[(size_t)c * 2 * st + p + st];
  else if (hasprev[s]) v += prev[(size_t)s * C + c] * w[(size_t)c * 2 * st + p + st];
  out[i] = v;
}
__global__ __launch_bounds__(256) void mimi_rope_g_kernel(float* qkv, int L, int heads, int hd, float theta, const int* pos0, int S) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int half = hd / 2, A = heads * hd;
  const size_t per = (size_t)L * 2 * heads * half;
  if (i >= per * S) return;
  const int s = (int)(i / per);
  const size_t j = i - (size_t)s * per;
  const int f = (int)(j % half);
  const size_t r = j / half;
  const int h = (int)(r % (2 * heads));
  const int pos = (int)(r / (2 * heads));
  float* p = qkv + ((size_t)s * L + pos) * 3 * A + (size_t)h * hd;
  const float ang = (float)(pos0[s] + pos) * powf(theta, -2.f * (float)f / (float)hd);
  const float c = cosf(ang), sn = sinf(ang);
  const float a = p[f], b = p[f + half];
  p[f] = a * c - b * sn;
  p[f + half] = b * c + a * sn;
}
__global__ __launch_bounds__(256) void mimi_kv_append_g_kernel(const float* qkv, float* hist, size_t hstride, const int* nh, size_t L, int A, int S) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t per = L * 2 * A;
  if (i >= per * S) return;
  const int s = (int)(i / per);
  const size_t j = i - (size_t)s * per;
  const size_t r = j / (2 * A);
  const int c = (int)(j % (2 * A));
  hist[(size_t)s * hstride + ((size_t)nh[s] + r) * 2 * A + c] = qkv[((size_t)s * L + r) * 3 * A + A + c];
}
// per stream: rows [nh + L - keep, nh + L) of src to the front of dst
__global__ __launch_bounds__(256) void mimi_rows_copy_g_kernel(const float* src, float* dst, size_t hstride, const int* nh, const int* keep, int L,
                                                               int A2) {
  const int s = blockIdx.y;
  const size_t n = (size_t)keep[s] * A2;
  const size_t from = (size_t)(nh[s] + L - keep[s]) * A2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dst[(size_t)s * hstride + i] = src[(size_t)s * hstride + from + i];
}
// mimi_attn_kernel with the stream taken from the row: grid = (S * L, heads)
__global__ __launch_bounds__(64) void mimi_attn_g_kernel(const float* qkv, const float* hist_all, size_t hstride, const int* nhs, int L, int heads, int hd,
                                                         int window, float* out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // q[hd] | scores[window]
  float* qs = sm;
  float* sc = sm + hd;
  const int row = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int s = row / L, i = row - s * L;
  const float* hist = hist_all + (size_t)s * hstride;
  const int A = heads * hd;
  const int hi = nhs[s] + i;
  const int j0 = max(0, hi - window + 1), n = hi - j0 + 1;
  const float* q = qkv + (size_t)row * 3 * A + (size_t)h * hd;
  const float scale = rsqrtf((float)hd);
  for (int d = tid; d < hd; d += 64) qs[d] = q[d] * scale;
  __syncthreads();
  float mx = -INFINITY;
  for (int j = tid; j < n; j += 64) {
    const float* k = hist + (size_t)(j0 + j) * 2 * A + (size_t)h * hd;
    float s0 = 0.f, s1 = 0.f;
    for (int d = 0; d < hd; d += 8) {
      const f32x4 k0 = *reinterpret_cast<const f32x4*>(k + d), k1 = *reinterpret_cast<const f32x4*>(k + d + 4);
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(qs + d), q1 = *reinterpret_cast<const f32x4*>(qs + d + 4);
      s0 += (k0[0] * q0[0] + k0[1] * q0[1]) + (k0[2] * q0[2] + k0[3] * q0[3]);
      s1 += (k1[0] * q1[0] + k1[1] * q1[1]) + (k1[2] * q1[2] + k1[3] * q1[3]);
    }
    const float sv = s0 + s1;
    sc[j] = sv;
    mx = fmaxf(mx, sv);
  }
  mx = wave_max(mx);
  float se = 0.f;
  for (int j = tid; j < n; j += 64) { const float p = expf(sc[j] - mx); sc[j] = p; se += p; }
  se = wave_sum(se);
  __syncthreads();
  const float inv = 1.f / se;
  for (int d = tid; d < hd; d += 64) {
    float o0 = 0.f, o1 = 0.f;
    const float* v = hist + (size_t)j0 * 2 * A + A + (size_t)h * hd + d;
    int j = 0;
    for (; j + 1 < n; j += 2) {
      o0 = fmaf(sc[j], v[(size_t)j * 2 * A], o0);
      o1 = fmaf(sc[j + 1], v[(size_t)(j + 1) * 2 * A], o1);
    }
    if (j < n) o0 = fmaf(sc[j], v[(size_t)j * 2 * A], o0);
    out[(size_t)row * A + (size_t)h * hd + d] = (o0 + o1) * inv;
  }
}
// convolution input of a group, [S][PR + L][C]: mode 0: the PR cached rows of every stream to the front of its segment;
// mode 1: (ELU of) the compact rows [S * L][C] behind them; mode 2: the last PR rows of every segment back to the cache
__global__ __launch_bounds__(256) void mimi_pad_g_kernel(float* pad, float* cache, const float* src, int S, int PR, size_t L, int C, int mode, int elu) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t seg = ((size_t)PR + L) * C;
  if (mode == 1) {
    const size_t per = L * C;
    if (i >= per * S) return;
    const size_t s = i / per, j = i - s * per;
    const float v = src[i];
    pad[s * seg + (size_t)PR * C + j] = (elu && v <= 0.f) ? expm1f(v) : v;
  } else {
    const size_t per = (size_t)PR * C;
    if (i >= per * S) return;
    const size_t s = i / per, j = i - s * per;
    if (mode == 0) pad[s * seg + j] = cache[i];
    else cache[i] = pad[s * seg + L * C + j];
  }
}
// mimi_bias_act_kernel for a GEMM over a group's padded input: the result row of (stream s, row l) is s * (PR + L) + l
__global__ __launch_bounds__(256) void mimi_bias_act_g_kernel(const float* src, int lds, const float* bias, int nb, int C, int S, int PR, size_t L,
                                                              int act, float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)S * L * C) return;
  const size_t r = i / C;
  const int c = (int)(i % C);
  const size_t s = r / L, l = r - s * L;
  float v = src[(s * ((size_t)PR + L) + l) * lds + c] + (bias ? bias[c % nb] : 0.f);
  if (act == 1) v = v > 0.f ? v : expm1f(v);
  out[i] = v;
}
// mimi_last_conv_kernel over a group's padded input [S][PR + L][C]; audio [S][L]
__global__ __launch_bounds__(256) void mimi_last_conv_g_kernel(const float* pad, const float* w, const float* b, int C, int k, int PR, size_t L, float* audio) {
  const int lane = threadIdx.x & 63, s = blockIdx.y;
  const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int K = k * C;
  const float* xa = pad + ((size_t)s * ((size_t)PR + L) + (size_t)(PR - (k - 1))) * C;
  float wr[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) wr[u] = lane + 64 * u < K ? w[lane + 64 * u] : 0.f;
  const float bias = b[0];
  for (int o = 0; o < 16; ++o) {
    const size_t l = wv * 16 + o;
    if (l >= L) return;
    const float* p = xa + l * C;
    float v = 0.f;
    if (K <= 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (lane + 64 * u < K) v = fmaf(wr[u], p[lane + 64 * u], v);
    } else {
      for (int i = lane; i < K; i += 64) v = fmaf(w[i], p[i], v);
    }
    v = wave_sum(v);
    if (lane == 0) audio[(size_t)s * L + l] = v + bias;
  }
}
#endif  // CSM_MIMI_KERNELS
