// Stage (a)-(c) bench + checker of the persistent decoder engine (dec_persist.h, this directory) on synthetic csm-1b
// decoder weights: ONE launch runs n_pass positions x n_layers layers (+ the fused arg-max head of every position >= 1)
// at B = 1.  The launch chain it replaces costs ~19 us per decoder layer-pass in the replaying frame graph
// (profiles/r03_b1_step_timeline.md, streamer on); VERDICT r3's gate: go if a layer-pass costs <= 16 us here.
//
//   ./persist_dec [n_pass=32] [n_layers=4] [reps=20] [check_passes=32]
// prints: give-ups, the CPU (double) check of the residual stream after every pass + the greedy tokens, us per launch and
// per layer-pass for {default, nt weights, no tag waits, no DMA, neither}, and the per-edge stamp breakdown of CU 0.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../csm-hf_amd/csrc persist_dec.hip -o persist_dec -lpthread
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#define CSM_DEC_PERSIST_KERNEL 1
#include "dec_persist.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using namespace dpk;
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Host {
  std::vector<uint16_t> wqkv[4], wo[4], wgu[4], wd[4], head;
  std::vector<float> ln1[4], ln2[4], fin, table, cosv, sinv, x0, x1;
  int V = 2051, C = 32, lmax = 32;
};

template <typename T>
static T* up(const std::vector<T>& h) {
  T* d;
  CK(hipMalloc(&d, h.size() * sizeof(T)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

static void fill_bf(std::vector<uint16_t>& v, size_t n, float std, uint64_t seed) {
  v.resize(n);
  const int nt = 8;
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      std::mt19937_64 g(seed * 977 + t);
      std::normal_distribution<float> d(0.f, std);
      for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) v[i] = f2bf(d(g));
    });
  for (auto& x : th) x.join();
}
static void fill_f(std::vector<float>& v, size_t n, float mean, float std, uint64_t seed) {
  v.resize(n);
  const int nt = 8;
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      std::mt19937_64 g(seed * 131 + t);
      std::normal_distribution<float> d(mean, std);
      for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) v[i] = d(g);
    });
  for (auto& x : th) x.join();
}

// y[n] = sum_k W[n][k] x[k] in double, rows split over threads
static void matvec(const std::vector<uint16_t>& W, size_t row0, int N, int K, const std::vector<double>& x, std::vector<double>& y) {
  y.assign(N, 0.0);
  const int nt = 16;
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      for (int n = N * t / nt; n < N * (t + 1) / nt; ++n) {
        const uint16_t* w = W.data() + (row0 + n) * (size_t)K;
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)bf2f(w[k]) * x[k];
        y[n] = s;
      }
    });
  for (auto& x_ : th) x_.join();
}
static void rms(const std::vector<double>& x, const std::vector<float>& w, double eps, std::vector<double>& y) {
  double ss = 0;
  for (double v : x) ss += v * v;
  const double sc = 1.0 / std::sqrt(ss / x.size() + eps);
  y.resize(x.size());
  for (size_t i = 0; i < x.size(); ++i) y[i] = x[i] * sc * w[i];
}

struct Ref {
  std::vector<std::vector<double>> xs;   // residual stream leaving every pass
  std::vector<int> tok;                  // greedy token of every pass >= 1
  std::vector<double> margin;
};

static Ref reference(const Host& h, int n_pass, int n_layers, bool kv_only0, double eps) {
  Ref r;
  std::vector<std::vector<double>> kc(n_layers, std::vector<double>(NKV * 32 * HD)), vc(n_layers, std::vector<double>(NKV * 32 * HD));
  int token = 0;
  for (int p = 0; p < n_pass; ++p) {
    std::vector<double> x(H);
    for (int i = 0; i < H; ++i) x[i] = p == 0 ? h.x0[i] : (p == 1 ? h.x1[i] : h.table[((size_t)(p - 1) * h.V + token) * H + i]);
    for (int l = 0; l < n_layers; ++l) {
      std::vector<double> xn, qkv;
      rms(x, h.ln1[l], eps, xn);
      matvec(h.wqkv[l], 0, NQKV, H, xn, qkv);
      for (int hd = 0; hd < NQ + NKV; ++hd)
        for (int i = 0; i < 64; ++i) {
          const double c = h.cosv[p * 64 + i], s = h.sinv[p * 64 + i];
          const double a = qkv[hd * HD + i], b = qkv[hd * HD + i + 64];
          qkv[hd * HD + i] = a * c - b * s;
          qkv[hd * HD + i + 64] = b * c + a * s;
        }
      for (int j = 0; j < NKV; ++j)
        for (int d = 0; d < HD; ++d) {
          kc[l][(j * 32 + p) * HD + d] = qkv[(NQ + j) * HD + d];
          vc[l][(j * 32 + p) * HD + d] = qkv[(NQ + NKV + j) * HD + d];
        }
      if (kv_only0 && p == 0 && l == n_layers - 1) break;
      std::vector<double> att(H);
      for (int hq = 0; hq < NQ; ++hq) {
        const int j = hq / (NQ / NKV);
        double sc[32], mx = -1e300, den = 0;
        for (int t = 0; t <= p; ++t) {
          double s = 0;
          for (int d = 0; d < HD; ++d) s += qkv[hq * HD + d] * kc[l][(j * 32 + t) * HD + d];
          sc[t] = s / std::sqrt((double)HD);
          mx = std::max(mx, sc[t]);
        }
        for (int t = 0; t <= p; ++t) { sc[t] = std::exp(sc[t] - mx); den += sc[t]; }
        for (int d = 0; d < HD; ++d) {
          double o = 0;
          for (int t = 0; t <= p; ++t) o += sc[t] * vc[l][(j * 32 + t) * HD + d];
          att[hq * HD + d] = o / den;
        }
      }
      std::vector<double> o;
      matvec(h.wo[l], 0, H, H, att, o);
      for (int i = 0; i < H; ++i) x[i] += o[i];
      std::vector<double> gu, act(F), dn;
      rms(x, h.ln2[l], eps, xn);
      matvec(h.wgu[l], 0, 2 * F, H, xn, gu);
      for (int i = 0; i < F; ++i) act[i] = gu[2 * i] / (1.0 + std::exp(-gu[2 * i])) * gu[2 * i + 1];
      matvec(h.wd[l], 0, H, F, act, dn);
      for (int i = 0; i < H; ++i) x[i] += dn[i];
    }
    r.xs.push_back(x);
    if (p == 0) { r.tok.push_back(-1); r.margin.push_back(0); continue; }
    std::vector<double> xn, lg;
    rms(x, h.fin, eps, xn);
    matvec(h.head, (size_t)(p - 1) * h.V, h.V, H, xn, lg);
    int b = 0;
    for (int i = 1; i < h.V; ++i) if (lg[i] > lg[b]) b = i;
    double second = -1e300;
    for (int i = 0; i < h.V; ++i) if (i != b) second = std::max(second, lg[i]);
    token = b;
    r.tok.push_back(b);
    r.margin.push_back(lg[b] - second);
  }
  return r;
}

int main(int argc, char** argv) {
  const int n_pass = argc > 1 ? atoi(argv[1]) : 32, n_layers = argc > 2 ? atoi(argv[2]) : 4, reps = argc > 3 ? atoi(argv[3]) : 20;
  const int check_passes = argc > 4 ? atoi(argv[4]) : n_pass;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, LDS per block %zu\n", prop.name, prop.multiProcessorCount, (size_t)prop.sharedMemPerBlock);
  if (prop.multiProcessorCount < NCU) { printf("needs %d CUs\n", NCU); return 1; }
  Host h;
  for (int l = 0; l < 4; ++l) {
    fill_bf(h.wqkv[l], (size_t)NQKV * H, 0.03f, 10 + l);
    fill_bf(h.wo[l], (size_t)H * H, 0.03f, 20 + l);
    fill_bf(h.wgu[l], (size_t)2 * F * H, 0.03f, 30 + l);
    fill_bf(h.wd[l], (size_t)H * F, 0.02f, 40 + l);
    fill_f(h.ln1[l], H, 1.f, 0.1f, 50 + l);
    fill_f(h.ln2[l], H, 1.f, 0.1f, 60 + l);
  }
  fill_f(h.fin, H, 1.f, 0.1f, 70);
  fill_bf(h.head, (size_t)(h.C - 1) * h.V * H, 0.03f, 80);
  fill_f(h.table, (size_t)h.C * h.V * H, 0.f, 0.5f, 90);
  fill_f(h.x0, H, 0.f, 0.5f, 91);
  fill_f(h.x1, H, 0.f, 0.5f, 92);
  h.cosv.resize(32 * 64);
  h.sinv.resize(32 * 64);
  for (int p = 0; p < 32; ++p)
    for (int i = 0; i < 64; ++i) {
      const float inv = powf(500000.f, -2.f * i / 128.f);
      h.cosv[p * 64 + i] = cosf(p * inv);
      h.sinv[p * 64 + i] = sinf(p * inv);
    }

  DecPersistArgs a{};
  for (int l = 0; l < 4; ++l) {
    a.wqkv[l] = up(h.wqkv[l]); a.wo[l] = up(h.wo[l]); a.wgu[l] = up(h.wgu[l]); a.wd[l] = up(h.wd[l]);
    a.ln1[l] = up(h.ln1[l]); a.ln2[l] = up(h.ln2[l]);
    CK(hipMalloc(&a.kcache[l], (size_t)NKV * 32 * HD * 4 + 1024));
    CK(hipMalloc(&a.vcache[l], (size_t)NKV * 32 * HD * 4 + 1024));
    CK(hipMemset(a.kcache[l], 0xff, (size_t)NKV * 32 * HD * 4));   // NaN patterns: nothing unwritten may be consumed
    CK(hipMemset(a.vcache[l], 0xff, (size_t)NKV * 32 * HD * 4));
  }
  a.final_norm = up(h.fin); a.head = up(h.head); a.tok_table = up(h.table); a.cos_tab = up(h.cosv); a.sin_tab = up(h.sinv);
  a.lmax = 32; a.x_pos0 = up(h.x0); a.x_pos1 = up(h.x1); a.forced = nullptr; a.C = h.C; a.V = h.V;
  int64_t* ring; CK(hipMalloc(&ring, 64 * 32 * 8)); CK(hipMemset(ring, 0xff, 64 * 32 * 8)); a.ring = ring;
  int* frame; CK(hipMalloc(&frame, 4)); CK(hipMemset(frame, 0, 4)); a.frame_ptr = frame;
  CK(hipMalloc(&a.gran, (size_t)GTOT * 8));
  CK(hipMalloc(&a.err, 64)); CK(hipMemset(a.err, 0, 64));
  a.eps = 1e-5f; a.qscale = 1.0f / sqrtf(128.f); a.n_pass = n_pass; a.n_layers = n_layers; a.kv_only_pass0 = 1; a.flags = 0;
  unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)(32 * 5 * 16 + 16) * 8)); CK(hipMemset(dbg, 0, (size_t)(32 * 5 * 16 + 16) * 8));
  float* dbg_x; CK(hipMalloc(&dbg_x, (size_t)32 * H * 4)); CK(hipMemset(dbg_x, 0, (size_t)32 * H * 4));
  if (int e = dpk::configure()) { printf("configure failed: %d\n", e); return 1; }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  auto run = [&](int flags, int nt, int n, bool stamps) -> double {
    a.flags = flags; a.dbg = stamps ? dbg : nullptr; a.dbg_x = stamps ? dbg_x : nullptr;
    double tot = 0;
    for (int i = 0; i < n; ++i) {
      CK(hipMemsetAsync(a.gran, 0, (size_t)GTOT * 8, st));
      CK(hipEventRecord(e0, st));
      const int r = dpk::launch(st, a, nt);
      if (r) { printf("launch failed: %d\n", r); exit(1); }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2 || n < 3) tot += ms;
    }
    return tot / (n < 3 ? n : n - 2) * 1000.0;
  };
  unsigned err[2];
  // ---- correctness ------------------------------------------------------------------------------------------------
  const double us1 = run(0, 0, 1, true);
  CK(hipMemcpy(err, a.err, 8, hipMemcpyDeviceToHost));
  printf("first launch: %.1f us, give-ups %u (first code 0x%x)\n", us1, err[0], err[1]);
  std::vector<float> gx((size_t)32 * H);
  std::vector<int64_t> gring(32);
  CK(hipMemcpy(gx.data(), dbg_x, gx.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(gring.data(), ring, 32 * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  if (check_passes > 0) {
    const int np = std::min(check_passes, n_pass);
    Ref r = reference(h, np, n_layers, a.kv_only_pass0 != 0, a.eps);
    for (int p = 0; p < np; ++p) {
      double num = 0, den = 0;
      for (int i = 0; i < H; ++i) { const double d = gx[(size_t)p * H + i] - r.xs[p][i]; num += d * d; den += r.xs[p][i] * r.xs[p][i]; }
      const double rel = std::sqrt(num / (den + 1e-30));
      const bool skip0 = p == 0 && a.kv_only_pass0;   // position 0's stream stops after its last K/V append
      const bool tok_ok = p == 0 || gring[p] == r.tok[p] || r.margin[p] < 1e-4;
      if ((!skip0 && !(rel < 2e-4)) || !tok_ok) ++bad;
      if (p < 4 || p == np - 1 || (!skip0 && !(rel < 2e-4)) || !tok_ok)
        printf("  pass %2d: rel-L2 of the residual stream %.2e%s   token %lld (reference %d, margin %.3g)%s\n", p, rel, skip0 ? " (kv-only)" : "",
               (long long)gring[p], r.tok[p], r.margin[p], tok_ok ? "" : "  MISMATCH");
      if (!tok_ok) break;   // later passes start from another token
    }
    printf("check: %s (%d bad of %d passes)\n", bad ? "FAILED" : "ok", bad, np);
  }
  // ---- timing + per-edge breakdown (CU 0, consumer 0; s_memrealtime = 10 ns ticks) -----------------------------------
  const int lp = n_pass * n_layers - (a.kv_only_pass0 ? 1 : 0);
  struct V { const char* name; int flags, nt; } vars[] = {
      {"default", 0, 0}, {"loader not thinned during gathers", 64, 0}, {"poll sleep 1", 1 << 8, 0}, {"poll sleep 3", 3 << 8, 0}, {"nt weight stream", 0, 1}, {"no tag waits", 1, 0}, {"no DMA", 2, 0}, {"no tag waits, no DMA", 3, 0},
      {"no tag waits, no DMA, no consumer barriers", 7, 0}, {"... and no act gather", 15, 0}, {"... and no attention arithmetic", 31, 0}};
  const char* names[] = {"QKV (norm, dot, RoPE, publish)", "gather q/k/v", "attention + barrier", "o_proj + publish", "gather x'",
                         "gate/up + publish", "gather act quarter", "down + combine + publish", "gather x"};
  for (const V& v : vars) {
    const double us = run(v.flags, v.nt, reps, false);
    CK(hipMemcpy(err, a.err, 8, hipMemcpyDeviceToHost));
    printf("%-44s %8.1f us per launch = %6.2f us per layer-pass (%d layer-passes + %d heads)  give-ups %u\n", v.name, us, us / lp, lp, n_pass - 1, err[0]);
    run(v.flags, v.nt, 2, true);
    std::vector<unsigned long long> ts((size_t)32 * 5 * 16 + 16);
    CK(hipMemcpy(ts.data(), dbg, ts.size() * 8, hipMemcpyDeviceToHost));
    double sum[9] = {0}, sub[4] = {0}, tot = 0;
    int cnt = 0;
    for (int p = 2; p < n_pass; ++p)
      for (int l = 0; l < n_layers; ++l) {
        const unsigned long long* t = &ts[((size_t)p * (n_layers + 1) + l) * 16];
        for (int e = 0; e < 9; ++e) sum[e] += (double)(t[e + 1] - t[e]) * 0.01;
        sub[1] += (double)(t[11] - t[2]) * 0.01;    // tile wait + patch
        sub[2] += (double)(t[12] - t[11]) * 0.01;   // two heads
        sub[3] += (double)(t[3] - t[12]) * 0.01;    // consumer barrier
        sub[0] += (double)((t[13] - t[5]) + (t[14] - t[13] > 0 ? 0 : 0)) * 0.01;   // gate/up: until the first group of slots has landed
        ++cnt;
      }
    if (cnt) {
      printf("    ");
      for (int e = 0; e < 9; ++e) { printf("%s %.2f | ", names[e], sum[e] / cnt); tot += sum[e] / cnt; }
      const unsigned long long* lp_ = &ts[(size_t)n_pass * (n_layers + 1) * 16];
      printf("total %.2f\n    sub: K/V tile wait + patch %.2f, two heads %.2f, barrier %.2f, gate/up until slots 0-15 landed %.2f; loader of CU 0: %.0f us blocked on a full ring, %.0f us in the 32-in-flight limit, %.0f us in all, %llu fills;", tot, sub[1] / cnt, sub[2] / cnt, sub[3] / cnt, sub[0] / cnt, lp_[0] * 0.01, lp_[1] * 0.01, lp_[2] * 0.01, lp_[3]);
      double hs[3] = {0};
      int hc = 0;
      for (int p = 2; p < n_pass; ++p) {
        const unsigned long long* t = &ts[((size_t)p * (n_layers + 1) + n_layers) * 16];
        hs[0] += (double)(t[1] - t[0]) * 0.01; hs[1] += (double)(t[2] - t[1]) * 0.01;
        if (p + 1 < n_pass) hs[2] += (double)(ts[((size_t)(p + 1) * (n_layers + 1)) * 16] - t[2]) * 0.01;
        ++hc;
      }
      printf(" head: dot + publish %.2f, pairs sweep + arg-max %.2f, table row + sync %.2f\n", hs[0] / hc, hs[1] / hc, hs[2] / hc);
    }
  }
  return bad ? 2 : 0;
}
