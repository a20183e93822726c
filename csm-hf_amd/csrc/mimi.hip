// Mimi decode behind the C ABI (include/csm_hip.h: csm_mimi_*): the orchestration of one decode call.
#define CSM_MIMI_KERNELS 1
#define CSM_ARGS_ONLY 1   // gemm.h: argument structs and launch_gemm only (the kernels live in launchers.hip)
#include "../../include/csm_hip.h"
#include "gemm.h"
#include "gemv.h"
#include "mimi.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int csm_set_error(int code, const char* msg);   // engine.hip: stores the message csm_last_error() returns
int gemv_configure_all();                       // gemv.hip: dynamic-LDS limits of the skinny-GEMM kernels (once per process)

static int mfail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return csm_set_error(code, buf);
}
#define MHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return mfail((int)e_, "%s: %s", #x, hipGetErrorString(e_)); } while (0)
#define MLCK(x) do { int r_ = (x); if (r_) return r_ < 0 ? mfail(CSM_ERR_ARG, "launch refused (%d): %s", r_, #x) : mfail(r_, "%s: %s", #x, hipGetErrorString((hipError_t)r_)); } while (0)

struct csm_mimi {
  csm_mimi_config_t c{};
  csm_mimi_weights_t w{};
  bool bound = false;
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  // scratch for ONE sequence (sequences of a batch are decoded one after the other)
  float *q2 = nullptr, *e0 = nullptr, *x = nullptr, *hn = nullptr, *qkv = nullptr, *ao = nullptr, *tmp = nullptr, *ff = nullptr;
  float *bufa = nullptr, *bufb = nullptr, *pad = nullptr, *scr = nullptr;
  size_t big = 0;   // floats in each of bufa / bufb / pad / scr
  // K/V history of the transformer, per layer, ping-pong: [window - 1 + max positions per call][2 A]
  std::vector<float*> hist[2];
  // streaming state (csm_mimi_stream_*): frames decoded so far, history rows kept, the frame before the next call's first,
  // and the last PADR input rows of every convolution (conv0, then per stage: transposed conv, residual conv; last conv)
  int s_frames = 0, s_nh = 0, s_cur = 0;
  float* s_up_prev = nullptr;
  std::vector<float*> s_conv;
  // stream GROUP (csm_mimi_streams_*): S streams in lockstep, every launch covers all of them (mimi.h: *_g kernels)
  int g_S = 0, g_T = 0, g_cur = 0;     // streams, most frames per call (max_frames / S), ping-pong index of the histories
  std::vector<int> g_frames, g_nh, g_host;   // per stream: frames decoded, history rows kept; staging of the device arrays
  std::vector<float*> g_hist[2];       // per layer: [S][g_hrows][2 A]
  size_t g_hrows = 0;
  float* g_up_prev = nullptr;          // [S][hidden]
  std::vector<float*> g_conv;          // per convolution: [S][PADR][C_in]
  float *g_pad = nullptr, *g_scr = nullptr;
  int* g_meta = nullptr;               // device [4][S]: rotary position | history rows | has a previous frame | rows to keep
  std::vector<void*> g_allocs;
  float* part = nullptr;               // split-K partial products of a GEMM with too few tiles (gemm())
  size_t part_floats = (size_t)8 << 20;
  bool splitk = true;                  // csm_mimi_set_option("splitk", 0): no K split (A/B; with "skinny_rows" 0: one kernel, one order)
  int skinny_rows = 16;   // GEMMs of at most this many rows take the skinny path; csm_mimi_set_option("skinny_rows", n) (0: none --
                          // every GEMM on the 128 x 128 tile: A/B measurements, bitwise stream == one-shot)
};
constexpr int PADR = 8;   // zero rows in front of every convolution input (>= kernel_size - 1)

static inline int pad128(int n) { return (n + 127) & ~127; }
static inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

template <typename T>
static int malloc_f(csm_mimi* m, T** p, size_t n) {
  void* q = nullptr;
  if (hipMalloc(&q, n * sizeof(T) + 256) != hipSuccess) return mfail(CSM_ERR_NOMEM, "hipMalloc(%zu bytes) failed", n * sizeof(T));
  m->allocs.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return 0;
}

extern "C" int csm_mimi_create(const csm_mimi_config_t* cfg, csm_mimi_t** out) {
  if (!cfg || !out) return mfail(CSM_ERR_ARG, "null argument");
  if (cfg->abi_version != CSM_ABI_VERSION) return mfail(CSM_ERR_ARG, "ABI version mismatch");
  const csm_mimi_config_t& c = *cfg;
  if (c.layers > CSM_MIMI_MAX_LAYERS || c.n_ratios > CSM_MIMI_MAX_RATIOS || c.n_ratios < 1) return mfail(CSM_ERR_ARG, "too many layers / ratios");
  if (c.kernel_size - 1 > PADR || c.res_kernel_size - 1 > PADR || c.last_kernel_size - 1 > PADR) return mfail(CSM_ERR_ARG, "kernel sizes above %d", PADR + 1);
  const int A = c.heads * c.head_dim;
  // GEMM shape rules (gemm.h: N % 128 == 0 is met by padding the weights' rows, K % 32 == 0 must hold as is)
  int ch = c.num_filters << c.n_ratios;
  bool ok = c.hidden % 128 == 0 && (2 * c.codebook_dim) % 32 == 0 && (3 * A) % 128 == 0 && A % 32 == 0 && c.ffn % 128 == 0 &&
            (c.kernel_size * c.hidden) % 32 == 0 && ch % 128 == 0 && c.head_dim % 2 == 0 && c.window >= 1;
  for (int i = 0; i < c.n_ratios && ok; ++i) {
    const int co = ch / 2, hid = co / c.compress;
    ok = (2 * ch) % 32 == 0 && (c.ratios[i] * co) % 128 == 0 && (c.res_kernel_size * co) % 32 == 0 && hid % 32 == 0 && co % 4 == 0;
    ch = co;
  }
  if (!ok) return mfail(CSM_ERR_ARG, "shape outside the GEMM path's rules (N %% 128, K %% 32)");
  csm_mimi* m = new csm_mimi();
  m->c = c;
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; return mfail(CSM_ERR_STATE, "stream creation failed"); }
  const size_t T = (size_t)c.max_frames, L1 = T * c.up_stride;
  size_t Lf = L1;
  size_t big = (L1 + PADR) * (size_t)std::max(c.hidden, c.num_filters << c.n_ratios);
  ch = c.num_filters << c.n_ratios;
  for (int i = 0; i < c.n_ratios; ++i) {
    const int co = ch / 2, hid = co / c.compress;
    big = std::max(big, (Lf + PADR) * (size_t)ch);                       // padded input of the transposed conv
    big = std::max(big, Lf * (size_t)(c.ratios[i] * co));                // its GEMM result
    Lf *= c.ratios[i];
    big = std::max(big, (Lf + PADR) * (size_t)co);
    big = std::max(big, Lf * (size_t)std::max(pad128(hid), pad128(co)));
    ch = co;
  }
  m->big = big;
  int r = 0;
  if ((r = malloc_f(m, &m->q2, T * 2 * c.codebook_dim)) || (r = malloc_f(m, &m->e0, T * c.hidden)) || (r = malloc_f(m, &m->x, L1 * c.hidden)) ||
      (r = malloc_f(m, &m->hn, L1 * c.hidden)) || (r = malloc_f(m, &m->qkv, L1 * 3 * A)) || (r = malloc_f(m, &m->ao, L1 * A)) ||
      (r = malloc_f(m, &m->tmp, L1 * c.hidden)) || (r = malloc_f(m, &m->ff, L1 * c.ffn)) || (r = malloc_f(m, &m->bufa, big)) ||
      (r = malloc_f(m, &m->bufb, big)) || (r = malloc_f(m, &m->pad, big)) || (r = malloc_f(m, &m->scr, big))) {
    for (void* p : m->allocs) hipFree(p);
    hipStreamDestroy(m->stream);
    delete m;
    return r;
  }
  {
    const size_t hrows = (size_t)(c.window - 1) + L1;
    int chs = c.num_filters << c.n_ratios;
    std::vector<int> cins{c.hidden};
    for (int i = 0; i < c.n_ratios; ++i) { cins.push_back(chs); cins.push_back(chs / 2); chs /= 2; }
    cins.push_back(chs);
    for (int l = 0; l < c.layers && !r; ++l)
      for (int pp = 0; pp < 2 && !r; ++pp) {
        float* h = nullptr;
        r = malloc_f(m, &h, hrows * 2 * A);
        m->hist[pp].push_back(h);
      }
    if (!r) r = malloc_f(m, &m->s_up_prev, (size_t)c.hidden);
    if (!r) r = malloc_f(m, &m->part, m->part_floats);
    if (!r && gemv_configure_all()) r = mfail(CSM_ERR_STATE, "skinny-GEMM kernel configuration failed");
    for (size_t i = 0; i < cins.size() && !r; ++i) {
      float* q = nullptr;
      r = malloc_f(m, &q, (size_t)PADR * cins[i]);
      m->s_conv.push_back(q);
    }
    if (r) {
      for (void* p : m->allocs) hipFree(p);
      hipStreamDestroy(m->stream);
      delete m;
      return r;
    }
  }
  *out = m;
  return 0;
}

extern "C" int csm_mimi_destroy(csm_mimi_t* m) {
  if (!m) return 0;
  hipStreamSynchronize(m->stream);
  for (void* p : m->allocs) hipFree(p);
  for (void* p : m->g_allocs) hipFree(p);
  hipStreamDestroy(m->stream);
  delete m;
  return 0;
}

extern "C" int csm_mimi_bind_weights(csm_mimi_t* m, const csm_mimi_weights_t* w) {
  if (!m || !w) return mfail(CSM_ERR_ARG, "null argument");
  if (!w->embed || !w->out_proj || !w->upsample || !w->conv0_w || !w->conv0_b || !w->last_w || !w->last_b) return mfail(CSM_ERR_ARG, "null weight");
  m->w = *w;
  m->bound = true;
  return 0;
}

// C[R][N] = A (rows of K floats, lda apart) @ W[N][K]^T, fp32 weights, exact-fp32 MFMA.
// The codec's GEMMs have few rows (2 T in the transformer, the first convolution and the first transposed convolution):
// on the plain 128 x 128 tile they are N / 128 = 4..32 workgroups, each walking K alone at one CU's fp32 matrix rate --
// measured 83-87 % of a decode's kernel time at ~80 us per launch (profiles/r02_mimi_stream_kernel_stats.md,
// r02_mimi_short_decode_kernel_stats.md).  Two remedies, both measured (profiles/r02_mimi_splitk.txt, r02_mimi_threshold.txt):
//  * <= skinny_rows rows: the weight-streaming skinny GEMM of the frame generator (gemv.h: fp32 FMA, <= 4 rows per
//    launch, weights spread over the chip), in groups of 4 rows: a one-frame streaming call 4.0 -> 0.8-1.0 ms;
//  * more rows but fewer than 128 tiles: K split over grid.y, partial products summed in fixed order
//    (mimi_splitk_reduce_kernel): one-shot decodes of 50 / 100 / 200 / 500 frames 5.2 / 5.5 / 6.3 / 9.1 -> 2.1 / 2.5 /
//    3.6 / 6.8 ms.
// With the split available the best row threshold is 16 (swept 4 / 16 / 32 / 64: 25 frames one-shot 1.7 ms at 16, 3.4 ms
// at 64; 4-frame calls 1.3 ms at 16, 1.6 ms at 4).  Neither path sums in the order of the unsplit MFMA chain: a stream is
// ~2e-6 of the peak away from the one-shot decode (bitwise equal with options skinny_rows = 0, splitk = 0); everything
// stays within the codec's 1e-4 of the reference implementation.
static int gemm(csm_mimi* m, const float* A, int lda, const float* W, int N, int K, size_t R, float* C, int ldc) {
  if (R <= (size_t)m->skinny_rows && K % 8 == 0) {   // groups of <= 4 rows, the weights streamed once per group
    for (size_t m0 = 0; m0 < R; m0 += 4) {
      GemvArgs a{};
      a.W = W; a.N = N; a.K = K; a.x = A + m0 * lda; a.ldx = lda; a.out = C + m0 * ldc; a.ldo = ldc; a.nt = 1;
      if (int r = launch_gemv(m->stream, CSM_DTYPE_F32, CSM_DTYPE_F32, (int)std::min<size_t>(4, R - m0), PRO_PLAIN, EPI_STORE, a)) return r;
    }
    return 0;
  }
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.R = (int)R; g.N = N; g.K = K; g.C = C; g.ldc = ldc;
  // too few 128 x 128 tiles to fill the chip (a decode of a few hundred frames: 2 T rows x N = 512..2048): split K over
  // grid.y, partial products to m->part, summed in fixed order (deterministic).  Every split keeps >= 128 of K.
  const size_t tiles = ((R + 127) / 128) * (size_t)(N / 128);
  if (m->splitk && tiles < 128 && ldc % 4 == 0) {
    int ks = 1;
    while (ks < 16 && tiles * ks < 192 && K % (32 * ks * 2) == 0 && K / (ks * 2) >= 128 && (size_t)(ks * 2) * R * N <= m->part_floats) ks *= 2;
    if (ks > 1) {
      g.ksplit = ks; g.Cpart = m->part; g.part_stride = R * (size_t)N;
      MLCK(launch_gemm(m->stream, CSM_DTYPE_F32, GEPI_PARTIAL, g));
      const size_t total4 = R * (size_t)N / 4;
      hipLaunchKernelGGL(mimi_splitk_reduce_kernel, dim3(nblk(total4)), dim3(256), 0, m->stream, m->part, ks, g.part_stride, C, ldc, N, total4);
      return 0;
    }
  }
  return launch_gemm(m->stream, CSM_DTYPE_F32, GEPI_STORE, g);
}

// causal conv1d, stride 1, dilation 1 (MimiConv1d, modeling_mimi.py:327-347): xin = exact-width channels-last input
// [L][Cin]; elu: apply nn.ELU to the input first; result (+ bias, + res, act) to out [L][Cout] (row stride ldo)
// the PADR rows in front of a convolution's input: zeros, or (streaming) the last PADR input rows of the previous call
static int pad_in(csm_mimi* m, const float* cache, int Cin) {
  if (cache) MHIP(hipMemcpyAsync(m->pad, cache, (size_t)PADR * Cin * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
  else MHIP(hipMemsetAsync(m->pad, 0, (size_t)PADR * Cin * sizeof(float), m->stream));
  return 0;
}
// (streaming) keep the last PADR rows of the PADR + L rows now in m->pad for the next call
static int pad_out(csm_mimi* m, float* cache, int Cin, size_t L) {
  if (cache) MHIP(hipMemcpyAsync(cache, m->pad + L * Cin, (size_t)PADR * Cin * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
  return 0;
}

// causal conv1d, stride 1, dilation 1 (MimiConv1d, modeling_mimi.py:327-347): xin = exact-width channels-last input
// [L][Cin]; elu: apply nn.ELU to the input first; result (+ bias, + res, act) to out [L][Cout] (row stride ldo)
static int conv1d(csm_mimi* m, const float* xin, size_t L, int Cin, int k, int elu, const float* Wp, int Cout, const float* bias,
                  const float* res, int act, float* out, int ldo, float* cache) {
  hipStream_t st = m->stream;
  if (int r = pad_in(m, cache, Cin)) return r;
  hipLaunchKernelGGL(mimi_elu_copy_kernel, dim3(nblk(L * Cin)), dim3(256), 0, st, xin, m->pad + (size_t)PADR * Cin, L * Cin, elu);
  if (int r = pad_out(m, cache, Cin, L)) return r;
  const int Np = pad128(Cout);
  MLCK(gemm(m, m->pad + (size_t)(PADR - (k - 1)) * Cin, Cin, Wp, Np, k * Cin, L, m->scr, Np));
  hipLaunchKernelGGL(mimi_bias_act_kernel, dim3(nblk(L * Cout)), dim3(256), 0, st, m->scr, Np, bias, Cout, res, Cout, L, act, out, ldo);
  const hipError_t le = hipGetLastError();
  return le != hipSuccess ? mfail((int)le, "conv1d launch failed: %s", hipGetErrorString(le)) : 0;
}

// one sequence: T frames of codes [n_q][T] -> T * samples_per_frame samples.  streaming: continue the handle's stream
// (history of the transformer, left context of every convolution, positions) instead of starting from silence
static int decode_one(csm_mimi* m, const int64_t* cb, int T, float* audio, bool streaming) {
  const csm_mimi_config_t& c = m->c;
  hipStream_t st = m->stream;
  const int H = c.hidden, D = c.codebook_dim, A = c.heads * c.head_dim, F = c.ffn;
  const int nh = streaming ? m->s_nh : 0;
  const int pos0 = streaming ? m->s_frames * c.up_stride : 0;
  int ci = 0;   // convolution cache index
  auto cache = [&]() -> float* { float* q = streaming ? m->s_conv[ci] : nullptr; ++ci; return q; };
  // ---- split RVQ decode + the two 1x1 output projections (one GEMM, K = 2 D) ----
  hipLaunchKernelGGL(mimi_rvq_gather_kernel, dim3(T), dim3(256), 0, st, cb, m->w.embed, c.n_q, c.n_sem, c.codebook_size, D, T, m->q2);
  MLCK(gemm(m, m->q2, 2 * D, m->w.out_proj, H, 2 * D, T, m->e0, H));
  // ---- upsample to the transformer's rate ----
  const size_t L1 = (size_t)T * c.up_stride;
  hipLaunchKernelGGL(mimi_upsample_kernel, dim3(nblk(L1 * H)), dim3(256), 0, st, m->e0, m->w.upsample, T, H, c.up_stride,
                     (streaming && m->s_frames > 0) ? m->s_up_prev : nullptr, m->x);
  if (streaming) MHIP(hipMemcpyAsync(m->s_up_prev, m->e0 + (size_t)(T - 1) * H, (size_t)H * sizeof(float), hipMemcpyDeviceToDevice, st));
  // ---- transformer ----
  const int keep = std::min<int>(c.window - 1, nh + (int)L1);   // history rows the next call can still see
  const int cur = streaming ? m->s_cur : 0;
  for (int l = 0; l < c.layers; ++l) {
    float* hist = m->hist[cur][l];
    hipLaunchKernelGGL(mimi_layernorm_kernel, dim3((unsigned)L1), dim3(256), 0, st, m->x, m->w.ln1_w[l], m->w.ln1_b[l], H, c.norm_eps, m->hn);
    MLCK(gemm(m, m->hn, H, m->w.wqkv[l], 3 * A, H, L1, m->qkv, 3 * A));
    hipLaunchKernelGGL(mimi_rope_kernel, dim3(nblk(L1 * 2 * c.heads * (c.head_dim / 2))), dim3(256), 0, st, m->qkv, (int)L1, c.heads, c.head_dim, c.rope_theta, pos0);
    hipLaunchKernelGGL(mimi_kv_append_kernel, dim3(nblk(L1 * 2 * A)), dim3(256), 0, st, m->qkv, hist + (size_t)nh * 2 * A, L1, A);
    hipLaunchKernelGGL(mimi_attn_kernel, dim3((unsigned)L1, c.heads), dim3(64), (size_t)(c.head_dim + c.window) * sizeof(float), st, m->qkv, hist, nh,
                       c.heads, c.head_dim, c.window, m->ao);
    if (streaming && keep > 0)
      hipLaunchKernelGGL(mimi_rows_copy_kernel, dim3(nblk((size_t)keep * 2 * A)), dim3(256), 0, st, hist + (size_t)(nh + L1 - keep) * 2 * A,
                         m->hist[cur ^ 1][l], (size_t)keep * 2 * A);
    MLCK(gemm(m, m->ao, A, m->w.wo[l], H, A, L1, m->tmp, H));
    hipLaunchKernelGGL(mimi_scale_add_kernel, dim3(nblk(L1 * H)), dim3(256), 0, st, m->x, m->tmp, m->w.ls1[l], L1 * H, H);
    hipLaunchKernelGGL(mimi_layernorm_kernel, dim3((unsigned)L1), dim3(256), 0, st, m->x, m->w.ln2_w[l], m->w.ln2_b[l], H, c.norm_eps, m->hn);
    MLCK(gemm(m, m->hn, H, m->w.w1[l], F, H, L1, m->ff, F));
    hipLaunchKernelGGL(mimi_gelu_kernel, dim3(nblk(L1 * F)), dim3(256), 0, st, m->ff, L1 * F);
    MLCK(gemm(m, m->ff, F, m->w.w2[l], H, F, L1, m->tmp, H));
    hipLaunchKernelGGL(mimi_scale_add_kernel, dim3(nblk(L1 * H)), dim3(256), 0, st, m->x, m->tmp, m->w.ls2[l], L1 * H, H);
  }
  // ---- SEANet decoder ----
  int ch = c.num_filters << c.n_ratios;
  size_t L = L1;
  float *cbuf = m->bufa, *nxt = m->bufb;
  if (int r = conv1d(m, m->x, L, H, c.kernel_size, 0, m->w.conv0_w, ch, m->w.conv0_b, nullptr, 0, cbuf, ch, cache())) return r;
  for (int i = 0; i < c.n_ratios; ++i) {
    const int rr = c.ratios[i], co = ch / 2, hid = co / c.compress;
    // ELU -> transposed conv (kernel 2 r, stride r, causal trim): one GEMM, K = 2 C_in, N = r C_out, output row q = positions r q ..
    float* cc = cache();
    if (int r = pad_in(m, cc, ch)) return r;
    hipLaunchKernelGGL(mimi_elu_copy_kernel, dim3(nblk(L * ch)), dim3(256), 0, st, cbuf, m->pad + (size_t)PADR * ch, L * ch, 1);
    if (int r = pad_out(m, cc, ch, L)) return r;
    MLCK(gemm(m, m->pad + (size_t)(PADR - 1) * ch, ch, m->w.up_w[i], rr * co, 2 * ch, L, m->scr, rr * co));
    hipLaunchKernelGGL(mimi_bias_act_kernel, dim3(nblk(L * rr * co)), dim3(256), 0, st, m->scr, rr * co, m->w.up_b[i], co, nullptr, rr * co, L, 0, nxt, rr * co);
    L *= rr;
    std::swap(cbuf, nxt);   // cbuf = [L][co]
    // residual block: x + conv1(ELU(conv3(ELU(x))))
    float* hb = nxt;        // [L][hid]
    if (int r = conv1d(m, cbuf, L, co, c.res_kernel_size, 1, m->w.res1_w[i], hid, m->w.res1_b[i], nullptr, 1, hb, hid, cache())) return r;
    {
      const int Np = pad128(co);
      MLCK(gemm(m, hb, hid, m->w.res2_w[i], Np, hid, L, m->scr, Np));
      hipLaunchKernelGGL(mimi_bias_act_kernel, dim3(nblk(L * co)), dim3(256), 0, st, m->scr, Np, m->w.res2_b[i], co, cbuf, co, L, 0, cbuf, co);
    }
    ch = co;
  }
  // ---- ELU, last convolution to one channel ----
  float* cc = cache();
  if (int r = pad_in(m, cc, ch)) return r;
  hipLaunchKernelGGL(mimi_elu_copy_kernel, dim3(nblk(L * ch)), dim3(256), 0, st, cbuf, m->pad + (size_t)PADR * ch, L * ch, 1);
  if (int r = pad_out(m, cc, ch, L)) return r;
  hipLaunchKernelGGL(mimi_last_conv_kernel, dim3((unsigned)((L + 63) / 64)), dim3(256), 0, st, m->pad + (size_t)(PADR - (c.last_kernel_size - 1)) * ch, m->w.last_w,
                     m->w.last_b, ch, c.last_kernel_size, L, audio);
  MHIP(hipGetLastError());
  if (streaming) {
    m->s_frames += T;
    m->s_nh = keep;
    m->s_cur ^= 1;
  }
  return 0;
}

extern "C" int csm_mimi_decode(csm_mimi_t* m, const int64_t* codes, int B, int T, float* audio) {
  if (!m || !m->bound || !codes || !audio) return mfail(CSM_ERR_ARG, "null argument / weights not bound");
  const csm_mimi_config_t& c = m->c;
  if (B < 1 || T < 1 || T > c.max_frames) return mfail(CSM_ERR_CAPACITY, "T = %d outside 1..max_frames %d", T, c.max_frames);
  size_t spf = (size_t)c.up_stride;
  for (int i = 0; i < c.n_ratios; ++i) spf *= c.ratios[i];
  for (int b = 0; b < B; ++b)
    if (int r = decode_one(m, codes + (size_t)b * c.n_q * T, T, audio + (size_t)b * T * spf, false)) return r;
  MHIP(hipStreamSynchronize(m->stream));
  return 0;
}

// ---- streaming decode (modeling_mimi.py:1388-1406 with decoder_past_key_values, MimiConv1dPaddingCache :73-166): one
// sequence decoded a few frames at a time; the concatenated output equals one decode of the whole sequence ----
// tuning switches of one handle (like csm_set_option): "skinny_rows" = GEMMs of at most this many rows go to the
// weight-streaming skinny GEMM (0..256; 0 = none), "splitk" = K split of GEMMs with too few tiles (0 / 1)
extern "C" int csm_mimi_set_option(csm_mimi_t* m, const char* name, int value) {
  if (!m || !name) return mfail(CSM_ERR_ARG, "null argument");
  if (!strcmp(name, "skinny_rows")) m->skinny_rows = std::max(0, std::min(256, value));
  else if (!strcmp(name, "splitk")) m->splitk = value != 0;
  else return mfail(CSM_ERR_ARG, "unknown option %s", name);
  return 0;
}

extern "C" int csm_mimi_stream_reset(csm_mimi_t* m) {
  if (!m) return mfail(CSM_ERR_ARG, "null argument");
  m->s_frames = m->s_nh = m->s_cur = 0;
  MHIP(hipMemsetAsync(m->s_up_prev, 0, (size_t)m->c.hidden * sizeof(float), m->stream));
  int chs = m->c.num_filters << m->c.n_ratios;
  std::vector<int> cins{m->c.hidden};
  for (int i = 0; i < m->c.n_ratios; ++i) { cins.push_back(chs); cins.push_back(chs / 2); chs /= 2; }
  cins.push_back(chs);
  for (size_t i = 0; i < cins.size(); ++i) MHIP(hipMemsetAsync(m->s_conv[i], 0, (size_t)PADR * cins[i] * sizeof(float), m->stream));
  MHIP(hipStreamSynchronize(m->stream));
  return 0;
}
extern "C" int csm_mimi_stream_decode(csm_mimi_t* m, const int64_t* codes, int T, float* audio) {
  if (!m || !m->bound || !codes || !audio) return mfail(CSM_ERR_ARG, "null argument / weights not bound");
  if (T < 1 || T > m->c.max_frames) return mfail(CSM_ERR_CAPACITY, "T = %d outside 1..max_frames %d", T, m->c.max_frames);
  if (int r = decode_one(m, codes, T, audio, true)) return r;
  MHIP(hipStreamSynchronize(m->stream));
  return 0;
}


// ---- stream groups: S streams decoded in lockstep, T frames each per call, every launch covering all of them -------------
// (one-frame streaming calls are launch-bound: S streams through csm_mimi_stream_decode cost S x 0.8 ms, through a group
// about one call).  Streams may be restarted one by one (csm_mimi_streams_reset(m, s)): rotary position, history length and
// left contexts are per stream.  Same arithmetic as decode_one; GEMMs see S times the rows, so they may take another of the
// three GEMM paths (fp32 summation order: ~1e-6 of the peak against the single-stream result).
template <typename T>
static int malloc_g(csm_mimi* m, T** p, size_t n) {
  void* q = nullptr;
  if (hipMalloc(&q, n * sizeof(T) + 256) != hipSuccess) return mfail(CSM_ERR_NOMEM, "hipMalloc(%zu bytes) failed", n * sizeof(T));
  m->g_allocs.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return 0;
}
static std::vector<int> conv_cins(const csm_mimi_config_t& c) {
  int chs = c.num_filters << c.n_ratios;
  std::vector<int> cins{c.hidden};
  for (int i = 0; i < c.n_ratios; ++i) { cins.push_back(chs); cins.push_back(chs / 2); chs /= 2; }
  cins.push_back(chs);
  return cins;
}

extern "C" int csm_mimi_streams_reset(csm_mimi_t* m, int stream) {
  if (!m || m->g_S < 1) return mfail(CSM_ERR_STATE, "no stream group (csm_mimi_streams_open first)");
  if (stream < -1 || stream >= m->g_S) return mfail(CSM_ERR_ARG, "stream %d outside the group of %d", stream, m->g_S);
  const std::vector<int> cins = conv_cins(m->c);
  const int s0 = stream < 0 ? 0 : stream, s1 = stream < 0 ? m->g_S : stream + 1;
  for (int s = s0; s < s1; ++s) { m->g_frames[s] = 0; m->g_nh[s] = 0; }
  MHIP(hipMemsetAsync(m->g_up_prev + (size_t)s0 * m->c.hidden, 0, (size_t)(s1 - s0) * m->c.hidden * sizeof(float), m->stream));
  for (size_t i = 0; i < cins.size(); ++i)
    MHIP(hipMemsetAsync(m->g_conv[i] + (size_t)s0 * PADR * cins[i], 0, (size_t)(s1 - s0) * PADR * cins[i] * sizeof(float), m->stream));
  MHIP(hipStreamSynchronize(m->stream));
  return 0;
}

extern "C" int csm_mimi_streams_open(csm_mimi_t* m, int S) {
  if (!m) return mfail(CSM_ERR_ARG, "null argument");
  const csm_mimi_config_t& c = m->c;
  if (S < 1 || S > c.max_frames) return mfail(CSM_ERR_CAPACITY, "a group of %d streams needs max_frames >= %d (one frame per stream and call)", S, S);
  MHIP(hipStreamSynchronize(m->stream));
  for (void* p : m->g_allocs) hipFree(p);
  m->g_allocs.clear();
  m->g_hist[0].clear(); m->g_hist[1].clear(); m->g_conv.clear();
  m->g_S = 0;
  const int A = c.heads * c.head_dim, Tg = c.max_frames / S;
  const size_t L1 = (size_t)Tg * c.up_stride;
  m->g_hrows = (size_t)(c.window - 1) + L1;
  // padded convolution inputs [S][PADR + L][C] and the GEMM results over them, stage by stage
  size_t Lf = L1, padf = (size_t)S * (PADR + L1) * c.hidden;
  int ch = c.num_filters << c.n_ratios;
  size_t scrf = (size_t)S * (PADR + L1) * pad128(ch);
  for (int i = 0; i < c.n_ratios; ++i) {
    const int co = ch / 2, hid = co / c.compress;
    padf = std::max(padf, (size_t)S * (PADR + Lf) * ch);
    scrf = std::max(scrf, (size_t)S * (PADR + Lf) * (size_t)(c.ratios[i] * co));
    Lf *= c.ratios[i];
    padf = std::max(padf, (size_t)S * (PADR + Lf) * co);
    scrf = std::max(scrf, (size_t)S * (PADR + Lf) * (size_t)std::max(pad128(hid), pad128(co)));
    ch = co;
  }
  padf = std::max(padf, (size_t)S * (PADR + Lf) * ch);
  int r = 0;
  for (int l = 0; l < c.layers && !r; ++l)
    for (int pp = 0; pp < 2 && !r; ++pp) {
      float* h = nullptr;
      r = malloc_g(m, &h, (size_t)S * m->g_hrows * 2 * A);
      m->g_hist[pp].push_back(h);
    }
  if (!r) r = malloc_g(m, &m->g_up_prev, (size_t)S * c.hidden);
  const std::vector<int> cins = conv_cins(c);
  for (size_t i = 0; i < cins.size() && !r; ++i) {
    float* q = nullptr;
    r = malloc_g(m, &q, (size_t)S * PADR * cins[i]);
    m->g_conv.push_back(q);
  }
  if (!r) r = malloc_g(m, &m->g_pad, padf);
  if (!r) r = malloc_g(m, &m->g_scr, scrf);
  if (!r) r = malloc_g(m, &m->g_meta, (size_t)4 * S);
  if (r) {
    for (void* p : m->g_allocs) hipFree(p);
    m->g_allocs.clear();
    return r;
  }
  m->g_S = S; m->g_T = Tg; m->g_cur = 0;
  m->g_frames.assign(S, 0); m->g_nh.assign(S, 0); m->g_host.assign((size_t)4 * S, 0);
  return csm_mimi_streams_reset(m, -1);
}

// group form of conv1d(): xin / out compact [S * L][C]; cache [S][PADR][Cin]
static int conv1d_g(csm_mimi* m, const float* xin, size_t L, int Cin, int k, int elu, const float* Wp, int Cout, const float* bias,
                    int act, float* out, float* cache) {
  hipStream_t st = m->stream;
  const int S = m->g_S;
  hipLaunchKernelGGL(mimi_pad_g_kernel, dim3(nblk((size_t)S * PADR * Cin)), dim3(256), 0, st, m->g_pad, cache, (const float*)nullptr, S, PADR, L, Cin, 0, 0);
  hipLaunchKernelGGL(mimi_pad_g_kernel, dim3(nblk((size_t)S * L * Cin)), dim3(256), 0, st, m->g_pad, (float*)nullptr, xin, S, PADR, L, Cin, 1, elu);
  hipLaunchKernelGGL(mimi_pad_g_kernel, dim3(nblk((size_t)S * PADR * Cin)), dim3(256), 0, st, m->g_pad, cache, (const float*)nullptr, S, PADR, L, Cin, 2, 0);
  const int Np = pad128(Cout);
  MLCK(gemm(m, m->g_pad + (size_t)(PADR - (k - 1)) * Cin, Cin, Wp, Np, k * Cin, (size_t)S * (PADR + L) - PADR, m->g_scr, Np));
  hipLaunchKernelGGL(mimi_bias_act_g_kernel, dim3(nblk((size_t)S * L * Cout)), dim3(256), 0, st, m->g_scr, Np, bias, Cout, Cout, S, PADR, L, act, out);
  const hipError_t le = hipGetLastError();
  return le != hipSuccess ? mfail((int)le, "conv1d_g launch failed: %s", hipGetErrorString(le)) : 0;
}

extern "C" int csm_mimi_streams_decode(csm_mimi_t* m, const int64_t* codes, int T, float* audio) {
  if (!m || !m->bound || !codes || !audio) return mfail(CSM_ERR_ARG, "null argument / weights not bound");
  if (m->g_S < 1) return mfail(CSM_ERR_STATE, "no stream group (csm_mimi_streams_open first)");
  if (T < 1 || T > m->g_T) return mfail(CSM_ERR_CAPACITY, "T = %d outside 1..%d (max_frames / streams)", T, m->g_T);
  const csm_mimi_config_t& c = m->c;
  hipStream_t st = m->stream;
  const int S = m->g_S, H = c.hidden, D = c.codebook_dim, A = c.heads * c.head_dim, F = c.ffn;
  const size_t L1 = (size_t)T * c.up_stride, R1 = (size_t)S * L1;
  int* hm = m->g_host.data();
  for (int s = 0; s < S; ++s) {
    hm[s] = m->g_frames[s] * c.up_stride;
    hm[S + s] = m->g_nh[s];
    hm[2 * S + s] = m->g_frames[s] > 0;
    hm[3 * S + s] = std::min<int>(c.window - 1, m->g_nh[s] + (int)L1);
  }
  MHIP(hipMemcpyAsync(m->g_meta, hm, (size_t)4 * S * sizeof(int), hipMemcpyHostToDevice, st));
  const int *d_pos0 = m->g_meta, *d_nh = m->g_meta + S, *d_prev = m->g_meta + 2 * S, *d_keep = m->g_meta + 3 * S;
  int ci = 0;
  auto cache = [&]() -> float* { return m->g_conv[ci++]; };
  hipLaunchKernelGGL(mimi_rvq_gather_g_kernel, dim3(T, S), dim3(256), 0, st, codes, m->w.embed, c.n_q, c.n_sem, c.codebook_size, D, T, m->q2);
  MLCK(gemm(m, m->q2, 2 * D, m->w.out_proj, H, 2 * D, (size_t)S * T, m->e0, H));
  hipLaunchKernelGGL(mimi_upsample_g_kernel, dim3(nblk(R1 * H)), dim3(256), 0, st, m->e0, m->w.upsample, T, H, c.up_stride, m->g_up_prev, d_prev, S, m->x);
  MHIP(hipMemcpy2DAsync(m->g_up_prev, (size_t)H * sizeof(float), m->e0 + (size_t)(T - 1) * H, (size_t)T * H * sizeof(float), (size_t)H * sizeof(float), S,
                        hipMemcpyDeviceToDevice, st));
  const size_t hstride = m->g_hrows * 2 * A;
  for (int l = 0; l < c.layers; ++l) {
    float* hist = m->g_hist[m->g_cur][l];
    hipLaunchKernelGGL(mimi_layernorm_kernel, dim3((unsigned)R1), dim3(256), 0, st, m->x, m->w.ln1_w[l], m->w.ln1_b[l], H, c.norm_eps, m->hn);
    MLCK(gemm(m, m->hn, H, m->w.wqkv[l], 3 * A, H, R1, m->qkv, 3 * A));
    hipLaunchKernelGGL(mimi_rope_g_kernel, dim3(nblk(R1 * 2 * c.heads * (c.head_dim / 2))), dim3(256), 0, st, m->qkv, (int)L1, c.heads, c.head_dim, c.rope_theta, d_pos0, S);
    hipLaunchKernelGGL(mimi_kv_append_g_kernel, dim3(nblk(R1 * 2 * A)), dim3(256), 0, st, m->qkv, hist, hstride, d_nh, L1, A, S);
    hipLaunchKernelGGL(mimi_attn_g_kernel, dim3((unsigned)R1, c.heads), dim3(64), (size_t)(c.head_dim + c.window) * sizeof(float), st, m->qkv, hist, hstride,
                       d_nh, (int)L1, c.heads, c.head_dim, c.window, m->ao);
    hipLaunchKernelGGL(mimi_rows_copy_g_kernel, dim3(64, S), dim3(256), 0, st, hist, m->g_hist[m->g_cur ^ 1][l], hstride, d_nh, d_keep, (int)L1, 2 * A);
    MLCK(gemm(m, m->ao, A, m->w.wo[l], H, A, R1, m->tmp, H));
    hipLaunchKernelGGL(mimi_scale_add_kernel, dim3(nblk(R1 * H)), dim3(256), 0, st, m->x, m->tmp, m->w.ls1[l], R1 * H, H);
    hipLaunchKernelGGL(mimi_layernorm_kernel, dim3((unsigned)R1), dim3(256), 0, st, m->x, m->w.ln2_w[l], m->w.ln2_b[l], H, c.norm_eps, m->hn);
    MLCK(gemm(m, m->hn, H, m->w.w1[l], F, H, R1, m->ff, F));
    hipLaunchKernelGGL(mimi_gelu_kernel, dim3(nblk(R1 * F)), dim3(256), 0, st, m->ff, R1 * F);
    MLCK(gemm(m, m->ff, F, m->w.w2[l], H, F, R1, m->tmp, H));
    hipLaunchKernelGGL(mimi_scale_add_kernel, dim3(nblk(R1 * H)), dim3(256), 0, st, m->x, m->tmp, m->w.ls2[l], R1 * H, H);
  }
  // ---- SEANet decoder ----
  int ch = c.num_filters << c.n_ratios;
  size_t L = L1;
  float *cbuf = m->bufa, *nxt = m->bufb;
  if (int r = conv1d_g(m, m->x, L, H, c.kernel_size, 0, m->w.conv0_w, ch, m->w.conv0_b, 0, cbuf, cache())) return r;
  for (int i = 0; i < c.n_ratios; ++i) {
    const int rr = c.ratios[i], co = ch / 2, hid = co / c.compress;
    float* cc = cache();
    hipLaunchKernelGGL(mimi_pad_g_kernel, dim3(nblk((size_t)S * PADR * ch)), dim3(256), 0, st, m->g_pad, cc, (const float*)nullptr, S, PADR, L, ch, 0, 0);
    hipLaunchKernelGGL(mimi_pad_g_kernel, dim3(nblk((size_t)S * L * ch)), dim3(256), 0, st, m->g_pad, (float*)nullptr, (const float*)cbuf, S, PADR, L, ch, 1, 1);
    hipLaunchKernelGGL(mimi_pad_g_kernel, dim3(nblk((size_t)S * PADR * ch)), dim3(256), 0, st, m->g_pad, cc, (const float*)nullptr, S, PADR, L, ch, 2, 0);
    MLCK(gemm(m, m->g_pad + (size_t)(PADR - 1) * ch, ch, m->w.up_w[i], rr * co, 2 * ch, (size_t)S * (PADR + L) - PADR, m->g_scr, rr * co));
    hipLaunchKernelGGL(mimi_bias_act_g_kernel, dim3(nblk((size_t)S * L * rr * co)), dim3(256), 0, st, m->g_scr, rr * co, m->w.up_b[i], co, rr * co, S, PADR, L, 0, nxt);
    L *= rr;
    std::swap(cbuf, nxt);   // cbuf = [S * L][co]
    float* hb = nxt;        // [S * L][hid]
    if (int r = conv1d_g(m, cbuf, L, co, c.res_kernel_size, 1, m->w.res1_w[i], hid, m->w.res1_b[i], 1, hb, cache())) return r;
    {
      const int Np = pad128(co);
      MLCK(gemm(m, hb, hid, m->w.res2_w[i], Np, hid, (size_t)S * L, m->g_scr, Np));
      hipLaunchKernelGGL(mimi_bias_act_kernel, dim3(nblk((size_t)S * L * co)), dim3(256), 0, st, m->g_scr, Np, m->w.res2_b[i], co, cbuf, co, (size_t)S * L, 0, cbuf, co);
    }
    ch = co;
  }
  float* cc = cache();
  hipLaunchKernelGGL(mimi_pad_g_kernel, dim3(nblk((size_t)S * PADR * ch)), dim3(256), 0, st, m->g_pad, cc, (const float*)nullptr, S, PADR, L, ch, 0, 0);
  hipLaunchKernelGGL(mimi_pad_g_kernel, dim3(nblk((size_t)S * L * ch)), dim3(256), 0, st, m->g_pad, (float*)nullptr, (const float*)cbuf, S, PADR, L, ch, 1, 1);
  hipLaunchKernelGGL(mimi_pad_g_kernel, dim3(nblk((size_t)S * PADR * ch)), dim3(256), 0, st, m->g_pad, cc, (const float*)nullptr, S, PADR, L, ch, 2, 0);
  hipLaunchKernelGGL(mimi_last_conv_g_kernel, dim3((unsigned)((L + 63) / 64), S), dim3(256), 0, st, m->g_pad, m->w.last_w, m->w.last_b, ch, c.last_kernel_size, PADR, L, audio);
  MHIP(hipGetLastError());
  MHIP(hipStreamSynchronize(st));
  for (int s = 0; s < S; ++s) {
    m->g_frames[s] += T;
    m->g_nh[s] = hm[3 * S + s];
  }
  m->g_cur ^= 1;
  return 0;
}
