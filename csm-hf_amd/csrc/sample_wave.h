// Top-k sampling of ONE logits row inside a 256-thread workgroup whose vector-memory loads must stay in flight (LDS-only
// barriers): the form of `sample_topk` that runs as the PROLOGUE of the next decoder pass's first QKV launch (gemv.h PRO_SAMPLE),
// redundantly in every workgroup of that launch, under the launch's weight loads -- the B = 1 sampled frame-step then has no
// sampler launch for codebooks 1..30 (round 5; the reference's default call is sampled: modeling_csm.py:591-600 `topk=50`,
// sampler :170-189).  (First form: every WAVE sampled on its own, 33 values per lane and no barrier at all -- 7 us of
// dependent single-wave arithmetic per launch; the four waves sharing the row take 9 values per thread and seven barriers.)
//
// Same arithmetic, term for term, as sample_kernel (misc.h) -- which the tests pin against the reference's draws -- so a token
// sampled here equals the token the stand-alone launch samples, bit for bit:
//   scaled = logits / T; kth = the k-th largest scaled value; survivors = { i : scaled_i >= kth } in index order (ties kept, like
//   the reference's `x < kth` mask); lp = log_softmax over the survivors, p = softmax(lp) (two normalisations, :181-187), q ~ Exp(1)
//   (explicit noise, or -log(u) of the Philox stream keyed (index, global row, codebook, frame | seed)), token = argmax p / q with
//   the lowest index on ties.  The per-lane partial sums run over j = lane, lane + 64, ... and meet in wave_sum / wave_max exactly
//   as in sample_kernel's wave 0 (wave 0 of the workgroup runs that tail; the token reaches the other waves through LDS).
// k-th largest: a 256-bin histogram over [min, max] of the still-active values in LDS, suffix scan to the bin holding
// the k-th, then -- once that bin holds <= 256 values -- the exact value by rank counting; a fuller bin is re-binned over its own
// [min, max] (each round strictly narrows the range or ends on all-equal values), so the select is exact for any input.
#pragma once
#include "common.h"

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ uint32_t f32_key(float x) {  // monotone float -> uint map
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct WaveSampleArgs {
  const float* logits;     // [V] of the row, already divided by the temperature
  int V;
  float temperature;
  int topk;
  const uint64_t* rng;     // nullable: device-resident {seed, global index of row 0} (overrides `seed`)
  const float* noise;      // nullable: explicit Exp(1) draws [V] of this row and codebook
  int cb, frame;
  int row;                 // batch row (Philox counter word 1 = row + rng[1]); 0 for the B = 1 in-launch form
  uint64_t seed;           // used when rng == nullptr (stand-alone csm_sample_topk)
};

constexpr int WS_NJ = 9;                        // values per thread: 2048 <= V <= 256 * 9 = 2304 (only a thread's LAST value can lie beyond V:
                                                // the launcher sends every other vocabulary size to the stand-alone sampler launch)
constexpr int WS_VMIN = 256 * (WS_NJ - 1), WS_VMAX = 256 * WS_NJ;
// LDS of the workgroup (words): hist[256 bins + 256 dump bins] | cand[256 + 4] | misc[64] | cnt[48] | cv[V + 4] | ci[V + 4]; the dump
// bins and `+ 4` slots swallow the writes of threads that have nothing to write -- every LDS access is unconditional
__host__ __device__ inline size_t wave_sample_lds_words(int V) { return 884 + 2 * ((size_t)V + 4); }

__device__ __forceinline__ int ws_prefix(uint64_t bal) {   // set bits of `bal` below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
}
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // LDS only: vmcnt untouched

// returns the sampled index (uniform over the 256 threads, all of which must call).  `lds` = wave_sample_lds_words(V) words.
// DIV = false: a.logits holds the logits ALREADY divided by the temperature (the head launch's epilogue divides: GemvArgs::store_div);
// DIV = true (sample_kernel): raw logits, divided here.
template <bool DIV = false>
__device__ __forceinline__ int wg_sample_topk(const WaveSampleArgs& a, float* lds, int tid) {
  const int V = a.V;
  const int lane = tid & 63, wave = tid >> 6;
  unsigned* hist = reinterpret_cast<unsigned*>(lds);
  float* cand = lds + 512;
  float* misc = lds + 772;
  int* cnt = reinterpret_cast<int*>(lds + 836);
  float* cv = lds + 884;
  int* ci = reinterpret_cast<int*>(lds + 884 + V + 4);
  float x[WS_NJ];
#pragma unroll
  for (int j = 0; j < WS_NJ - 1; ++j) x[j] = a.logits[tid + 256 * j];
  {
    const int i = tid + 256 * (WS_NJ - 1);
    const float v = a.logits[i < V ? i : V - 1];
    x[WS_NJ - 1] = i < V ? v : -INFINITY;      // padding: below every finite value, outside every [lo, hi] below
  }
  if (DIV) {
#pragma unroll
    for (int j = 0; j < WS_NJ; ++j) x[j] = x[j] / a.temperature;   // (-inf stays -inf for T > 0)
  }
  const int ktop = a.topk < V ? a.topk : V;
  int krem = ktop;
  uint32_t kth_key = 0;   // (if 64 rounds ever ended without a decision, key 0 keeps every value: a superset, never a dropped member)
  float lo = -3.402823466e38f, hi = 3.402823466e38f;   // value range still holding the k-th largest (bins are monotone in the value)
  for (int round = 0; round < 64; ++round) {   // (a round narrows [lo, hi] to one of its 256 bins, or ends; every decision is workgroup-uniform)
    float mxv = -INFINITY, mnv = INFINITY;
#pragma unroll
    for (int j = 0; j < WS_NJ; ++j) {
      const bool ac = x[j] >= lo && x[j] <= hi;
      mxv = fmaxf(mxv, ac ? x[j] : -INFINITY);
      mnv = fminf(mnv, ac ? x[j] : INFINITY);
    }
    mxv = wave_max(mxv);
    mnv = -wave_max(-mnv);
    if (lane == 0) { misc[wave] = mxv; misc[4 + wave] = mnv; }
    hist[tid] = 0u;
    hist[256 + tid] = 0u;
    ws_barrier();
    mxv = fmaxf(fmaxf(misc[0], misc[1]), fmaxf(misc[2], misc[3]));
    mnv = fminf(fminf(misc[4], misc[5]), fminf(misc[6], misc[7]));
    const float bs = 255.99f / (mxv - mnv);
    // every remaining candidate has the same value -- or a range too narrow to bin (mxv - mnv denormal: 255.99 / range overflows).  In both
    // cases the candidates that are left are all KEPT (key of the minimum): for equal values that is the reference's tie rule; for the
    // unbinnable range it keeps a few values beyond the k-th instead of dropping top-k members (round 5 took the maximum's key there: ADVICE r5)
    if (!(mxv > mnv) || !(bs < INFINITY)) { kth_key = f32_key(mnv); break; }
#pragma unroll
    for (int j = 0; j < WS_NJ; ++j) {
      const bool ac = x[j] >= lo && x[j] <= hi;
      const int b = min(255, (int)(((ac ? x[j] : mnv) - mnv) * bs));
      atomicAdd(&hist[ac ? b : 256 + tid], 1u);
    }
    ws_barrier();
    const uint4 h4 = *reinterpret_cast<const uint4*>(hist + 4 * lane);   // bins 4 lane .. 4 lane + 3 (every wave scans for itself)
    const int tot = (int)(h4.x + h4.y + h4.z + h4.w);
    int incl = tot;   // sum over the lanes >= own
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_down(incl, o, 64);
      incl += lane + o < 64 ? v : 0;
    }
    const int above = incl - tot;   // values in bins beyond this lane's four
    const int a3 = above, a2 = a3 + (int)h4.w, a1 = a2 + (int)h4.z, a0 = a1 + (int)h4.y;
    int bsel = -1, kin = 0;
    if (a0 < krem && a0 + (int)h4.x >= krem) { bsel = 4 * lane; kin = krem - a0; }
    if (a1 < krem && a1 + (int)h4.y >= krem) { bsel = 4 * lane + 1; kin = krem - a1; }
    if (a2 < krem && a2 + (int)h4.z >= krem) { bsel = 4 * lane + 2; kin = krem - a2; }
    if (a3 < krem && a3 + (int)h4.w >= krem) { bsel = 4 * lane + 3; kin = krem - a3; }
    const uint64_t hit = __ballot(bsel >= 0);      // exactly one lane
    if (!hit) { kth_key = f32_key(mnv); break; }   // (unreachable for finite inputs: krem never exceeds the active count; NaN logits land here)
    const int src = __builtin_ctzll(hit);
    bsel = __builtin_amdgcn_readlane(bsel, src);
    kin = __builtin_amdgcn_readlane(kin, src);
    // the selected bin's values: per-wave counts -> LDS -> every wave knows its base in cand[]
    uint64_t bal[WS_NJ];
    int mine = 0;
    float nlo = INFINITY, nhi = -INFINITY;
#pragma unroll
    for (int j = 0; j < WS_NJ; ++j) {
      const bool ac = x[j] >= lo && x[j] <= hi;
      const bool in = ac && min(255, (int)(((ac ? x[j] : mnv) - mnv) * bs)) == bsel;
      bal[j] = __ballot(in);
      mine += __builtin_popcountll(bal[j]);
      nlo = fminf(nlo, in ? x[j] : INFINITY);
      nhi = fmaxf(nhi, in ? x[j] : -INFINITY);
    }
    if (lane == 0) cnt[wave] = mine;
    ws_barrier();
    const int c0 = cnt[0], c1 = cnt[1], c2 = cnt[2], c3 = cnt[3];
    const int nb = c0 + c1 + c2 + c3;
    int pos0 = wave == 0 ? 0 : (wave == 1 ? c0 : (wave == 2 ? c0 + c1 : c0 + c1 + c2));
#pragma unroll
    for (int j = 0; j < WS_NJ; ++j) {
      const bool in = (bal[j] >> lane) & 1ull;
      const int pos = pos0 + ws_prefix(bal[j]);
      cand[in && pos < 256 ? pos : 256 + (lane & 3)] = x[j];
      pos0 += __builtin_popcountll(bal[j]);
    }
    krem = kin;
    if (nb <= 256) {
      ws_barrier();
      uint32_t found = 0;
      bool have = false;
      for (int c = lane; c < nb; c += 64) {     // (every wave ranks for itself: nb is a handful)
        const float vj = cand[c];
        int gt = 0, ge = 0;
        for (int i = 0; i < nb; ++i) {
          const float vi = cand[i];
          gt += vi > vj;
          ge += vi >= vj;
        }
        if (gt < kin && kin <= ge) { found = f32_key(vj); have = true; }
      }
      const uint64_t hb = __ballot(have);
      kth_key = (uint32_t)__builtin_amdgcn_readlane((int)found, hb ? __builtin_ctzll(hb) : 0);
      break;
    }
    nlo = -wave_max(-nlo);
    nhi = wave_max(nhi);
    if (lane == 0) { misc[8 + wave] = nlo; misc[12 + wave] = nhi; }
    ws_barrier();
    lo = fminf(fminf(misc[8], misc[9]), fminf(misc[10], misc[11]));
    hi = fmaxf(fmaxf(misc[12], misc[13]), fmaxf(misc[14], misc[15]));
  }
  // ---- survivors (key >= k-th key; ties kept) in index order: element tid + 256 j -> ordered by (j, wave, lane)
  uint64_t sb[WS_NJ];
#pragma unroll
  for (int j = 0; j < WS_NJ; ++j) {
    sb[j] = __ballot(f32_key(x[j]) >= kth_key);      // (the -inf padding has the smallest key of all)
    if (lane == 0) cnt[4 + 4 * j + wave] = __builtin_popcountll(sb[j]);
  }
  ws_barrier();
  int ns = 0;
#pragma unroll
  for (int j = 0; j < WS_NJ; ++j) {
    const int4 c = *reinterpret_cast<const int4*>(cnt + 4 + 4 * j);
    const int before = wave == 0 ? 0 : (wave == 1 ? c.x : (wave == 2 ? c.x + c.y : c.x + c.y + c.z));
    const bool in = (sb[j] >> lane) & 1ull;
    const int pos = in ? ns + before + ws_prefix(sb[j]) : V + (lane & 3);
    cv[pos] = x[j];
    ci[pos] = tid + 256 * j;
    ns += c.x + c.y + c.z + c.w;
  }
  ws_barrier();
  if (wave == 0) {
    // ---- sample_kernel's wave-0 tail, verbatim
    float mx = -INFINITY;
    for (int j = lane; j < ns; j += 64) mx = fmaxf(mx, cv[j]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int j = lane; j < ns; j += 64) se += expf(cv[j] - mx);
    const float lse = logf(wave_sum(se));
    float m2 = -INFINITY;
    for (int j = lane; j < ns; j += 64) {
      const float lp = (cv[j] - mx) - lse;
      cv[j] = lp;
      m2 = fmaxf(m2, lp);
    }
    m2 = wave_max(m2);
    float s2 = 0.f;
    for (int j = lane; j < ns; j += 64) s2 += expf(cv[j] - m2);
    s2 = wave_sum(s2);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < ns; j += 64) {
      const int i = ci[j];
      const float p = expf(cv[j] - m2) / s2;
      float q;
      if (a.noise) {
        q = a.noise[i];
      } else {
        uint32_t r[4];
        const uint64_t seed = a.rng ? a.rng[0] : a.seed;
        const uint32_t grow = (uint32_t)a.row + (a.rng ? (uint32_t)a.rng[1] : 0u);   // global row: shards draw distinct streams
        philox4x32_10((uint32_t)i, grow, (uint32_t)a.cb, (uint32_t)a.frame, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        const float u = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        q = -logf(u);
      }
      const float v = p / q;
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    wave_argmax(bv, bi);
    if (lane == 0) cnt[44] = bi;
  }
  ws_barrier();
  return cnt[44];
}
