// Launchers of the training-backward kernels (train.h).
#include <hip/hip_runtime.h>

#include "train.h"

static inline unsigned nb(size_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }
#define TRL return (int)hipGetLastError()

int tr_launch_swiglu(hipStream_t st, const float* gu, float* act, const float* dact, float* dgu, size_t n) {
  if (n == 0) return 0;
  if (dgu) hipLaunchKernelGGL(tr_swiglu_bwd_kernel, dim3(nb(n)), dim3(256), 0, st, gu, dact, dgu, n);
  else hipLaunchKernelGGL(tr_swiglu_fwd_kernel, dim3(nb(n)), dim3(256), 0, st, gu, act, n);
  TRL;
}
int tr_launch_rmsnorm_bwd(hipStream_t st, const float* x, const float* w, const float* dy, int rows, int H, float eps, float* dx,
                          int accumulate, float* gw_scratch, float* dw) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(tr_rmsnorm_bwd_kernel, dim3(rows), dim3(256), 0, st, x, w, dy, H, eps, dx, accumulate, gw_scratch);
  hipLaunchKernelGGL(tr_colsum_kernel, dim3(nb(H)), dim3(256), 0, st, gw_scratch, rows, H, dw);
  TRL;
}
int tr_launch_rope(hipStream_t st, float* qkv, int R, int n_rot_heads, int nqkv, int hd, const int* row_pos, const float* cos_tab,
                   const float* sin_tab, int inverse) {
  const size_t n = (size_t)R * n_rot_heads * (hd / 2);
  if (n == 0) return 0;
  hipLaunchKernelGGL(tr_rope_kernel, dim3(nb(n)), dim3(256), 0, st, qkv, R, n_rot_heads, nqkv, hd, row_pos, cos_tab, sin_tab, inverse);
  TRL;
}
int tr_launch_attn_fwd(hipStream_t st, const TrAttnArgs& a) {
  if (a.R == 0) return 0;
  if ((size_t)a.S * sizeof(float) > 60 * 1024 || a.hd > 128) return -1;
  hipLaunchKernelGGL(tr_attn_fwd_kernel, dim3(a.R, a.n_q), dim3(64), (size_t)a.S * sizeof(float), st, a);
  TRL;
}
int tr_launch_attn_bwd(hipStream_t st, const TrAttnArgs& a) {
  if (a.R == 0) return 0;
  if ((size_t)a.S * sizeof(float) > 60 * 1024 || a.hd > 128) return -1;
  hipLaunchKernelGGL(tr_attn_bwd_q_kernel, dim3(a.R, a.n_q), dim3(64), (size_t)a.S * sizeof(float), st, a);
  hipLaunchKernelGGL(tr_attn_bwd_kv_kernel, dim3(a.R, a.n_kv), dim3(64), 0, st, a);
  TRL;
}
int tr_launch_ce_bwd(hipStream_t st, const float* logits, int ld, int V, const int* labels, int rows, float scale, float* dl, int ldo) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(tr_ce_bwd_kernel, dim3(rows), dim3(256), 0, st, logits, ld, V, labels, scale, dl, ldo);
  TRL;
}
int tr_launch_transpose_f32(hipStream_t st, const float* src, int rows, int cols, int lds, float* dst, int ldd) {
  if (cols == 0 || ldd == 0) return 0;
  hipLaunchKernelGGL((tr_transpose_kernel<float>), dim3((cols + 31) / 32, (ldd + 31) / 32), dim3(256), 0, st, src, rows, cols, lds, dst, ldd);
  TRL;
}
int tr_launch_transpose_w(hipStream_t st, int wdtype, const void* src, int rows, int cols, void* dst, int ldd) {
  const dim3 grid((cols + 31) / 32, (ldd + 31) / 32);
  if (wdtype == 1) hipLaunchKernelGGL((tr_transpose_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)src, rows, cols, cols, (bf16_t*)dst, ldd);
  else if (wdtype == 0) hipLaunchKernelGGL((tr_transpose_kernel<float>), grid, dim3(256), 0, st, (const float*)src, rows, cols, cols, (float*)dst, ldd);
  else return -1;
  TRL;
}
int tr_launch_dec_gather(hipStream_t st, int wdtype, const float* hb, const void* audio_emb, const int64_t* ids, const int* prev_row,
                         const int* tok_row, int frames, int P, int C, int V, int Hb, float* out) {
  const size_t n = (size_t)frames * P * Hb;
  if (n == 0) return 0;
  if (wdtype == 1) hipLaunchKernelGGL((tr_dec_gather_kernel<bf16_t>), dim3(nb(n)), dim3(256), 0, st, hb, (const bf16_t*)audio_emb, ids, prev_row, tok_row, frames, P, C, V, Hb, out);
  else if (wdtype == 0) hipLaunchKernelGGL((tr_dec_gather_kernel<float>), dim3(nb(n)), dim3(256), 0, st, hb, (const float*)audio_emb, ids, prev_row, tok_row, frames, P, C, V, Hb, out);
  else return -1;
  TRL;
}
int tr_launch_dec_scatter(hipStream_t st, const float* dE, const int64_t* ids, const int* prev_row, const int* tok_row, int frames,
                          int P, int C, int V, int Hb, float* dhb, float* d_audio_emb) {
  const size_t n = (size_t)frames * P * Hb;
  if (n == 0) return 0;
  hipLaunchKernelGGL(tr_dec_scatter_kernel, dim3(nb(n)), dim3(256), 0, st, dE, ids, prev_row, tok_row, frames, P, C, V, Hb, dhb, d_audio_emb);
  TRL;
}
int tr_launch_embed_bwd(hipStream_t st, const float* dx, const int64_t* ids, const uint8_t* mask, int rows, int C, int V, int Hb,
                        float* d_text, float* d_audio) {
  const size_t n = (size_t)rows * (C + 1) * Hb;
  if (n == 0) return 0;
  hipLaunchKernelGGL(tr_embed_bwd_kernel, dim3(nb(n)), dim3(256), 0, st, dx, ids, mask, rows, C, V, Hb, d_text, d_audio);
  TRL;
}
int tr_launch_add(hipStream_t st, float* dst, const float* src, size_t n) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(tr_add_kernel, dim3(nb(n)), dim3(256), 0, st, dst, src, n);
  TRL;
}
int tr_launch_copy2d(hipStream_t st, const float* src, size_t lds, float* dst, size_t ldd, int rows, int cols, int accumulate) {
  const size_t n = (size_t)rows * cols;
  if (n == 0) return 0;
  hipLaunchKernelGGL(tr_copy2d_kernel, dim3(nb(n)), dim3(256), 0, st, src, lds, dst, ldd, rows, cols, accumulate);
  TRL;
}
