// EXPERIMENT (round 4, VERDICT r3 "next round" item 1, stage (a)-(c)) -- measured, lost, NOT part of libcsm_hip.so: see
// profiles/r04_persistent_decoder.md.  Kept with its bench (persist_dec.hip) so that the numbers can be reproduced.
//
// Persistent decoder engine: the reference's 31-step inner decoder loop of one frame
// (modeling_csm.py:555-576, and its first two-position forward :534-552 run as positions 0 and 1) as ONE launch at
// B = 1, greedy -- instead of ~500 dependent launches (QKV -> attention + o_proj -> gate/up -> down per layer-pass,
// head per pass), each of which pays a kernel boundary plus a cold activation round trip for 2-33 MB of weights.
//
// Geometry: 256 workgroups = one per CU (160 KB of LDS each forces that), 5 waves: wave 0 is the LOADER, waves 1-4 are
// CONSUMERS.  The loader streams this CU's share of every weight matrix, in consumption order, into a 30 x 4 KiB LDS
// ring with LDS-DMA (`global_load_lds_dwordx4`: 1 KiB per wave instruction, no VGPR round trip), runs up to 32 KiB
// ahead in flight and as far ahead as the ring allows ACROSS the dependency edges of the layer -- that run-ahead is
// what a launch chain cannot do.  The weights stay in the engine's own row-major packing (no second copy): a 4 KiB
// slot is one task of the launch chain (a RoPE / SwiGLU / head row pair at K = 1024, a quarter of a down_proj row).
// Consumers multiply slots with EXACTLY the per-row arithmetic of gemv1_kernel / attn_oproj_kernel (same lane -> k
// mapping, same accumulator pairs, same DPP reduction order), so a greedy token stream is bitwise the launch chain's.
//
// Hand-offs between CUs: every op's output vector goes to every CU as 8-byte {value, tag} granules written by ONE
// agent-scope (sc1, write-through) store each and swept with agent-scope loads until every tag equals the edge's epoch
// (MI355X guide, Guideline 16 R2: the data is the flag, no fence) -- x (1024), q/k/v (1536), x after o_proj (1024), the
// SwiGLU output (8192) per layer-pass, and 256 (max, index) pairs per head.  The granule arrays are zeroed by a memset
// node before every launch; tags are edge numbers within the launch (never 0).  Every spin is bounded.
// The K/V of earlier positions are read from the global cache with sc1 loads (their producers stored sc1 a pass ago);
// the current position's K/V arrive with the q granules.
#pragma once
#include "attn_tile.h"
#include "common.h"

namespace dpk {
constexpr int H = 1024, F = 8192, NQ = 8, NKV = 2, HD = 128, NQKV = (NQ + 2 * NKV) * HD;   // csm-1b decoder
constexpr int NCU = 256, NCW = 4, NTHREADS = 64 * (NCW + 1);
constexpr int RING = 34, SLOT = 4096, DEPTH = 8;   // ring slots, bytes per slot, slots in flight
constexpr int S_QKV = 0, S_O = 3, S_GU = 5, S_DN = 37, SLOTS_LAYER = 53, SLOTS_HEAD = 5;
// granule arrays (8 bytes each)
constexpr int GX = 0, GQ = GX + H, GO = GQ + NQKV, GA = GO + H, GH = GA + F, GTOT = GH + 2 * NCU;
// LDS map (bytes)
// ring | per-consumer strip (q of two heads, new k, new v: 2 KiB each) | attention output | softmax strips; the 1024- and
// 8192-vectors handed between CUs live in REGISTERS of the waves that multiply them (no LDS copy, no barrier)
constexpr int LDS_SQW = RING * SLOT, LDS_ATT = LDS_SQW + NCW * 2048, LDS_P = LDS_ATT + H * 4, LDS_XA = LDS_P + NCW * 256, LDS_XB = LDS_XA + H * 4, LDS_BYTES = LDS_XB + H * 4;   // + the static control block
constexpr unsigned ABORT = 0x40000000u;
}  // namespace dpk

struct DecPersistArgs {
  const bf16_t* wqkv[4];   // [1536][1024]
  const bf16_t* wo[4];     // [1024][1024]
  const bf16_t* wgu[4];    // [16384][1024] gate / up rows interleaved
  const bf16_t* wd[4];     // [1024][8192]
  const float* ln1[4];
  const float* ln2[4];
  const float* final_norm;
  const bf16_t* head;        // audio_head_t [C - 1][V][1024]
  const float* tok_table;    // [C * V][1024] fp32 projected audio embeddings
  const float* cos_tab;      // [pos][64]
  const float* sin_tab;
  float* kcache[4];          // sequence 0: [n_kv][hd / 4][lmax][4]
  float* vcache[4];          //             [n_kv][lmax][hd]
  int lmax;
  const float* x_pos0;       // [1024] input of position 0 (projected backbone state)
  const float* x_pos1;       // [1024] input of position 1 (projected embedding of codebook 0)
  const int64_t* forced;     // teacher-forced tokens [max_frames][C], nullable
  int64_t* ring;             // generated-frame ring   [max_frames][C]
  const int* frame_ptr;
  int C, V;
  unsigned long long* gran;  // dpk::GTOT granules, zeroed before the launch
  unsigned* err;             // [0] give-ups, [1] first give-up code
  float eps, qscale;         // qscale = 1 / sqrt(head_dim) as the engine computes it
  int n_pass;                // positions 0 .. n_pass-1 (C for a frame)
  int n_layers;              // 4
  int kv_only_pass0;         // position 0's last layer only appends K/V (its hidden state is never read)
  int flags;                 // TIMING ONLY (wrong results): bit 0 gathers do not wait for tags, bit 1 the loader issues no DMA, bit 2 no consumer
                             // barriers, bit 3 no SwiGLU-vector gather, bit 4 no attention arithmetic
  unsigned long long* dbg;   // [n_pass][n_layers + 1][16] s_memrealtime stamps of CU 0 / consumer 0, nullable
  float* dbg_x;              // [n_pass][1024] the residual stream leaving every pass (CU 0), nullable
};

#ifdef CSM_DEC_PERSIST_KERNEL
namespace dpk {
typedef unsigned long long u64;
struct Misc {
  unsigned landed, sync, pcnt, dead, gathering, pad_[3];
  unsigned slot_gen[40];
  float part[4][4];
  float amv[4];
  int ami[4];
};

__shared__ Misc g_misc;   // control words of the workgroup (static LDS: its address space is known at every use)
// every access names LDS explicitly: through a generic `volatile` pointer these words are read with flat sc0 sc1 loads
typedef __attribute__((address_space(3))) unsigned lu32;
typedef __attribute__((address_space(3))) float lf32;
typedef __attribute__((address_space(3))) int li32;
#define LW(field) ((volatile dpk::lu32*)(&dpk::g_misc.field))
#define LF(field) ((volatile dpk::lf32*)(&dpk::g_misc.field))
#define LI(field) ((volatile dpk::li32*)(&dpk::g_misc.field))
__device__ __forceinline__ unsigned lds_ld(const volatile lu32* p) { return *p; }

// give up: make every LDS wait of this workgroup pass, count it
__device__ __forceinline__ void give_up(const DecPersistArgs& a, unsigned code, int lane) {
  if (lane == 0) {
    *LW(dead) = 1u;
    *LW(landed) = ABORT; *LW(sync) = ABORT; *LW(pcnt) = ABORT;
#pragma unroll 1
    for (int i = 0; i < 40; ++i) LW(slot_gen)[i] = ABORT;
    if (atomicAdd(a.err, 1u) == 0u) a.err[1] = code;
  }
}

// wait until an LDS word reaches `target`; bounded (100 ms) like every spin of this kernel
__device__ __forceinline__ void lds_wait(const DecPersistArgs& a, const volatile lu32* p, unsigned target, unsigned code,
                                         int lane) {
  if (__builtin_amdgcn_readfirstlane(lds_ld(p)) >= target) { asm volatile("" ::: "memory"); return; }
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  while (__builtin_amdgcn_readfirstlane(lds_ld(p)) < target) {
    __builtin_amdgcn_s_sleep(1);   // leave the LDS queue to the DMA and to the waves that compute
    if ((++spins & 1023u) == 0u && (__builtin_amdgcn_s_memrealtime() - t0 > 10000000ll || lds_ld(LW(dead)))) { give_up(a, code, lane); break; }
  }
  asm volatile("" ::: "memory");
}
// consumer-only barrier: an LDS arrival counter (s_barrier would tie in the loader wave)
__device__ __forceinline__ void csync(const DecPersistArgs& a, unsigned& target, int lane) {
  target += NCW;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (a.flags & 4) return;   // TIMING ONLY
  if (lane == 0) __hip_atomic_fetch_add((lu32*)LW(sync), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  lds_wait(a, LW(sync), target, 0x10u, lane);
}
__device__ __forceinline__ void wait_landed(const DecPersistArgs& a, unsigned i, int lane) {
  lds_wait(a, LW(landed), i + 1u, 0x20u, lane);
}
__device__ __forceinline__ void release_slot(unsigned i, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of the slot has returned
  if (lane == 0) LW(slot_gen)[i % RING] = i / RING + 1u;
}
__device__ __forceinline__ const char* slot_ptr(const char* smem, unsigned i) { return smem + (i % RING) * SLOT; }

// one sweep loop over NL granules per lane (granule k * 64 + lane of g); values land in dst[k * 64 + lane]
template <int NL>
__device__ __forceinline__ void gather_regs(const DecPersistArgs& a, const u64* g, unsigned epoch, u64 (&v)[NL], int lane,
                                            unsigned code) {
  unsigned pending = NL >= 32 ? 0xffffffffu : ((1u << NL) - 1u);
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  while (pending) {
#pragma unroll
    for (int k = 0; k < NL; ++k)
      if ((pending >> k) & 1u) v[k] = __hip_atomic_load(g + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < NL; ++k)
      if ((pending >> k) & 1u) {
        const bool ok = (unsigned)(v[k] >> 32) == epoch;
        if (__all(ok) || (a.flags & 1)) pending &= ~(1u << k);
      }
    if (pending && (++spins & 63u) == 0u) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ll || __builtin_amdgcn_readfirstlane(lds_ld(LW(dead)))) {   // 50 ms
        give_up(a, code, lane);
        break;
      }
    }
  }
}
template <int NL>
__device__ __forceinline__ void gather(const DecPersistArgs& a, const u64* g, unsigned epoch, float* dst, int lane,
                                       unsigned code) {
  u64 v[NL];
  gather_regs<NL>(a, g, epoch, v, lane, code);
#pragma unroll
  for (int k = 0; k < NL; ++k) dst[k * 64 + lane] = __uint_as_float((unsigned)v[k]);
}

__device__ __forceinline__ void publish(u64* g, unsigned epoch, float v) {
  __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ f32x2 wpair(const u32x4& r, int i) { return f32x2{bf16_lo(r[i]), bf16_hi(r[i])}; }

// activation slice of a K = 1024 normed launch exactly as gemv1_kernel<.., U = 2, KS = 1> holds it: lane l owns
// k = 8 l + 512 u .. + 7; RMS statistic, scale and norm weight applied in the kernel's order
__device__ __forceinline__ void normed_x(const float* sx, const f32x4 (&la)[2], const f32x4 (&lb)[2], float eps, int lane,
                                         f32x2 (&xp)[2][4]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const f32x4 xa = *reinterpret_cast<const f32x4*>(sx + lane * 8 + u * 512);
    const f32x4 xb = *reinterpret_cast<const f32x4*>(sx + lane * 8 + u * 512 + 4);
    xp[u][0] = f32x2{xa[0], xa[1]}; xp[u][1] = f32x2{xa[2], xa[3]};
    xp[u][2] = f32x2{xb[0], xb[1]}; xp[u][3] = f32x2{xb[2], xb[3]};
  }
  f32x2 ss2 = f32x2{0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) ss2 = PKFMA(xp[u][i], xp[u][i], ss2);
  const float ssw = wave_sum(ss2[0] + ss2[1]);
  const float sc = __builtin_amdgcn_rsqf(ssw * __builtin_amdgcn_rcpf((float)H) + eps);
  const f32x2 sc2 = f32x2{sc, sc};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    xp[u][0] = (xp[u][0] * sc2) * f32x2{la[u][0], la[u][1]};
    xp[u][1] = (xp[u][1] * sc2) * f32x2{la[u][2], la[u][3]};
    xp[u][2] = (xp[u][2] * sc2) * f32x2{lb[u][0], lb[u][1]};
    xp[u][3] = (xp[u][3] * sc2) * f32x2{lb[u][2], lb[u][3]};
  }
}
__device__ __forceinline__ void load_ln(const float* ln, int lane, f32x4 (&la)[2], f32x4 (&lb)[2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    la[u] = *reinterpret_cast<const f32x4*>(ln + lane * 8 + u * 512);
    lb[u] = *reinterpret_cast<const f32x4*>(ln + lane * 8 + u * 512 + 4);
  }
}
// a row pair at K = 1024 from one slot (row r0 at +0, row r1 at +2048): gemv1_kernel's accumulation
__device__ __forceinline__ void dot_pair(const char* slot, const f32x2 (&xp)[2][4], int lane, float& s0, float& s1) {
  u32x4 w0[2], w1[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    w0[u] = *reinterpret_cast<const u32x4*>(slot + u * 1024 + lane * 16);
    w1[u] = *reinterpret_cast<const u32x4*>(slot + 2048 + u * 1024 + lane * 16);
  }
  f32x2 c0 = f32x2{0.f, 0.f}, c1 = f32x2{0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      c0 = PKFMA(wpair(w0[u], i), xp[u][i], c0);
      c1 = PKFMA(wpair(w1[u], i), xp[u][i], c1);
    }
  s0 = c0[0] + c0[1];
  s1 = c1[0] + c1[1];
  wave_sum2(s0, s1);
}

// One LDS-DMA instruction: 64 lanes x 16 bytes from `src` (per lane) to 1 KiB of LDS at byte address `dst` (wave-uniform).
// Inline asm on purpose: issued through the builtin, hipcc orders every later LDS access of the wave (the ring's control
// words) behind the DMA with s_waitcnt vmcnt(0) -- ONE fill in flight, 0.66 us per 4 KiB, the whole engine starved (first
// measurements of this file).  The asm is invisible to that pass; landing is counted by the loader's own vmcnt(32).
template <bool NT>
__device__ __forceinline__ void dma1k(const char* src, unsigned dst) {
  unsigned keep;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

template <bool NT>
__device__ __forceinline__ void loader(const DecPersistArgs& a, char* smem, int cu, int lane) {
  unsigned i = 0, pos = 0, gen = 0;
  const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const bool prof = a.dbg && cu == 0;
  long long t_full = 0, t_vm = 0;
  const long long t_begin = __builtin_amdgcn_s_memrealtime();
  // snapshot of the release words: lane L holds slot_gen[L], lanes >= 40 hold `dead` -- ONE LDS read serves many fills (an LDS
  // round trip per fill sat behind the consumers' read bursts and polls)
  unsigned sg = 0;
  bool alive = true;
  auto snap = [&]() {
    sg = lds_ld(lane < 40 ? LW(slot_gen) + lane : (lane == 62 ? LW(gathering) : LW(dead)));
    alive = __builtin_amdgcn_readlane(sg, 63) == 0u;
  };
  auto fill = [&](const char* s0, const char* s1, const char* s2, const char* s3) {
    if (__builtin_amdgcn_readlane(sg, pos) < gen) {
      snap();
      if (__builtin_amdgcn_readlane(sg, pos) < gen) {   // ring full: publish what is in flight, then wait
        const long long tf0 = prof ? __builtin_amdgcn_s_memrealtime() : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && alive) *LW(landed) = i;
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        while (__builtin_amdgcn_readlane(sg, pos) < gen) {
          __builtin_amdgcn_s_sleep(1);
          snap();
          if (!alive) break;
          if ((++spins & 255u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > 10000000ll) {
            // (no returning atomic on this path: hipcc would guard its result register with s_waitcnt vmcnt(0) in EVERY fill)
            if (lane == 0) {
              *LW(dead) = 1u;
              *LW(landed) = ABORT; *LW(sync) = ABORT; *LW(pcnt) = ABORT;
              __hip_atomic_store(a.err + 2, 0x40u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            alive = false;
            break;
          }
        }
        if (prof) t_full += __builtin_amdgcn_s_memrealtime() - tf0;
      }
    }
    if (!(a.flags & 2)) {
      const unsigned d = lbase + pos * SLOT;
      dma1k<NT>(s0 + lane * 16, d);
      dma1k<NT>(s1 + lane * 16, d + 1024);
      dma1k<NT>(s2 + lane * 16, d + 2048);
      dma1k<NT>(s3 + lane * 16, d + 3072);
    }
    ++i;
    if (++pos == RING) { pos = 0; ++gen; }
    const long long tv0 = prof ? __builtin_amdgcn_s_memrealtime() : 0;
    // while consumers of this CU sweep granules the loader is THINNED to one fill in flight: their polls and publishes
    // share the CU's memory queue with the DMA (MI355X guide, gather-pass row: 0.3-0.65 us per pass with the own DMA quiet,
    // 1.0-1.7 behind an unthrottled refill burst).  The snapshot is refreshed by every fill; it is one fill old when used.
    const bool thin = !(a.flags & 64) && __builtin_amdgcn_readlane(sg, 62) != 0u;
    snap();
    if (thin) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (lane == 0 && alive && i > 1u) *LW(landed) = i - 1u;
    } else {
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");   // 4 * DEPTH: every slot below i - DEPTH has landed
      if (i > DEPTH && lane == 0 && alive) *LW(landed) = i - DEPTH;
    }
    if (prof) t_vm += __builtin_amdgcn_s_memrealtime() - tv0;
  };
  for (int p = 0; p < a.n_pass; ++p) {
    for (int l = 0; l < a.n_layers; ++l) {
      const char* wq = reinterpret_cast<const char*>(a.wqkv[l]);
#pragma unroll 1
      for (int j = 0; j < 3; ++j) {
        const int t = 3 * cu + j, head = t >> 6, hi = t & 63;
        const int r0 = head < NQ + NKV ? head * HD + hi : head * HD + 2 * hi;
        const int r1 = head < NQ + NKV ? r0 + 64 : r0 + 1;
        fill(wq + (size_t)r0 * 2048, wq + (size_t)r0 * 2048 + 1024, wq + (size_t)r1 * 2048, wq + (size_t)r1 * 2048 + 1024);
      }
      if (a.kv_only_pass0 && p == 0 && l == a.n_layers - 1) continue;
      const char* wo = reinterpret_cast<const char*>(a.wo[l]) + (size_t)(4 * cu) * 2048;
#pragma unroll 1
      for (int j = 0; j < 2; ++j) fill(wo + j * 4096, wo + j * 4096 + 1024, wo + j * 4096 + 2048, wo + j * 4096 + 3072);
      const char* wg = reinterpret_cast<const char*>(a.wgu[l]) + (size_t)(32 * cu) * 4096;
#pragma unroll 1
      for (int j = 0; j < 32; ++j) fill(wg + j * 4096, wg + j * 4096 + 1024, wg + j * 4096 + 2048, wg + j * 4096 + 3072);
      const char* wd = reinterpret_cast<const char*>(a.wd[l]) + (size_t)(4 * cu) * 16384;
#pragma unroll 1
      for (int j = 0; j < 16; ++j) fill(wd + j * 4096, wd + j * 4096 + 1024, wd + j * 4096 + 2048, wd + j * 4096 + 3072);
    }
    if (p >= 1) {
      const char* hs = reinterpret_cast<const char*>(a.head) + (size_t)(p - 1) * a.V * 2048;
#pragma unroll 1
      for (int j = 0; j < SLOTS_HEAD; ++j) {
        int t = j < 4 ? 4 * cu + j : (H + cu);   // tasks 1024, 1025 (rows 2048 .. 2050) on CUs 0 and 1
        if (2 * t >= a.V) t = 0;
        const int r0 = 2 * t, r1 = 2 * t + 1 < a.V ? 2 * t + 1 : 2 * t;
        fill(hs + (size_t)r0 * 2048, hs + (size_t)r0 * 2048 + 1024, hs + (size_t)r1 * 2048, hs + (size_t)r1 * 2048 + 1024);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && alive) *LW(landed) = i;
  if (prof && lane == 0) {   // loader profile of CU 0 behind the stamps: ticks blocked on a full ring, in the in-flight limit, in total, fills
    unsigned long long* d = a.dbg + (size_t)a.n_pass * (a.n_layers + 1) * 16;
    d[0] = (unsigned long long)t_full; d[1] = (unsigned long long)t_vm; d[2] = (unsigned long long)(__builtin_amdgcn_s_memrealtime() - t_begin); d[3] = i;
  }
}

__device__ __forceinline__ f32x4 ld_sc1(const float* base, unsigned off_floats) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7ffffff0, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off_floats * 4u, 0, /*sc1*/ 16));
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sweep NP granule PAIRS per lane (pair p = granules idx(p), idx(p) + 1, idx even: one 16-byte sc1 load, each 8-byte half
// written by one store) until every tag equals `epoch`; out[2 p], out[2 p + 1] = the two values.  `mid` runs once, between
// the issue of the first sweep and its check: loads issued there return under the wait for remote data (a first sweep
// almost never finds everything).
template <int NP, typename IdxF, typename MidF>
__device__ __forceinline__ void gather_pairs(const DecPersistArgs& a, unsigned epoch, IdxF idx, MidF mid, float (&out)[2 * NP], int lane, unsigned code) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.gran, 0, 0x7ffffff0, 0x00020000);
  u32x4 v[NP];
  unsigned pending = (1u << NP) - 1u;
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  if (lane == 0) __hip_atomic_fetch_add((lu32*)LW(gathering), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // thins the loader
  // first sweep, `mid`, first check -- unconditionally, so that what `mid` defines is defined on every path
#pragma unroll
  for (int p = 0; p < NP; ++p) v[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)idx(p) * 8u, 0, /*sc1*/ 16);
  mid();
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const bool ok = v[p][1] == epoch && v[p][3] == epoch;
    if (__all(ok) || (a.flags & 1)) pending &= ~(1u << p);
  }
  while (pending) {
    for (int z = (a.flags >> 8) & 15; z > 0; --z) __builtin_amdgcn_s_sleep(2);
    if ((++spins & 63u) == 0u && (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ll || __builtin_amdgcn_readfirstlane(lds_ld(LW(dead))))) {   // 50 ms
      give_up(a, code, lane);
      break;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if ((pending >> p) & 1u) v[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)idx(p) * 8u, 0, /*sc1*/ 16);
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if ((pending >> p) & 1u) {
        const bool ok = v[p][1] == epoch && v[p][3] == epoch;
        if (__all(ok) || (a.flags & 1)) pending &= ~(1u << p);
      }
  }
  if (lane == 0) __hip_atomic_fetch_add((lu32*)LW(gathering), 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
  for (int p = 0; p < NP; ++p) { out[2 * p] = __uint_as_float(v[p][0]); out[2 * p + 1] = __uint_as_float(v[p][2]); }
}
struct NoMid { __device__ __forceinline__ void operator()() const {} };

// A 1024-vector to every consumer in the layout gemv1_kernel<.., U = 2> holds it (x16[8 u + e] = x[8 lane + 512 u + e]):
// each consumer sweeps ONE quarter of the granules (every wave sweeping all of them put four times the polling traffic
// into the CU's memory pipe, next to the weight stream: 41 us per layer-pass instead of 23), the quarters meet in LDS.
__device__ __forceinline__ void gather_x16(const DecPersistArgs& a, int g0, unsigned epoch, float* sx, unsigned& sync_t, int w, float (&x16)[16],
                                           int lane, unsigned code) {
  float q[4];
  gather_pairs<2>(a, epoch, [&](int p) { return g0 + 256 * w + 128 * p + 2 * lane; }, NoMid(), q, lane, code);
  *reinterpret_cast<f32x2*>(sx + 256 * w + 2 * lane) = f32x2{q[0], q[1]};
  *reinterpret_cast<f32x2*>(sx + 256 * w + 128 + 2 * lane) = f32x2{q[2], q[3]};
  csync(a, sync_t, lane);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const f32x4 xa = *reinterpret_cast<const f32x4*>(sx + lane * 8 + u * 512), xb = *reinterpret_cast<const f32x4*>(sx + lane * 8 + u * 512 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { x16[8 * u + e] = xa[e]; x16[8 * u + 4 + e] = xb[e]; }
  }
}
// the four values x[4 cu .. 4 cu + 3] out of that register layout, wave-uniform
__device__ __forceinline__ void pick4(const float (&x16)[16], int cu, float (&r)[4]) {
  const int i0 = 4 * cu, u = i0 >> 9, src = (i0 & 511) >> 3, e0 = i0 & 7;
  // (lane reads first, then scalar selects: a select between register ELEMENTS became an indexed access through scratch)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x16[k]), src));
    const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x16[4 + k]), src));
    const float c2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x16[8 + k]), src));
    const float c3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x16[12 + k]), src));
    r[k] = u ? (e0 ? c3 : c2) : (e0 ? c1 : c0);
  }
}
__device__ __forceinline__ void normed_x16(const float (&x16)[16], const f32x4 (&la)[2], const f32x4 (&lb)[2], float eps, f32x2 (&xp)[2][4]) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) xp[u][i] = f32x2{x16[8 * u + 2 * i], x16[8 * u + 2 * i + 1]};
  f32x2 ss2 = f32x2{0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) ss2 = PKFMA(xp[u][i], xp[u][i], ss2);
  const float ssw = wave_sum(ss2[0] + ss2[1]);
  const float sc = __builtin_amdgcn_rsqf(ssw * __builtin_amdgcn_rcpf((float)H) + eps);
  const f32x2 sc2 = f32x2{sc, sc};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    xp[u][0] = (xp[u][0] * sc2) * f32x2{la[u][0], la[u][1]};
    xp[u][1] = (xp[u][1] * sc2) * f32x2{la[u][2], la[u][3]};
    xp[u][2] = (xp[u][2] * sc2) * f32x2{lb[u][0], lb[u][1]};
    xp[u][3] = (xp[u][3] * sc2) * f32x2{lb[u][2], lb[u][3]};
  }
}
// NT row pairs at K = 1024 from NT slots, interleaved (independent accumulator chains and reductions: one wave per SIMD
// has nothing else to hide their latencies behind); per pair exactly dot_pair's arithmetic
template <int NT_>
__device__ __forceinline__ void dot_pairs(const char* const (&slot)[NT_], const f32x2 (&xp)[2][4], int lane, float (&s0)[NT_], float (&s1)[NT_]) {
  f32x2 c0[NT_], c1[NT_];
#pragma unroll
  for (int t = 0; t < NT_; ++t) { c0[t] = f32x2{0.f, 0.f}; c1[t] = f32x2{0.f, 0.f}; }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    u32x4 w0[NT_], w1[NT_];
#pragma unroll
    for (int t = 0; t < NT_; ++t) {
      w0[t] = *reinterpret_cast<const u32x4*>(slot[t] + u * 1024 + lane * 16);
      w1[t] = *reinterpret_cast<const u32x4*>(slot[t] + 2048 + u * 1024 + lane * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT_; ++t) {
        c0[t] = PKFMA(wpair(w0[t], i), xp[u][i], c0[t]);
        c1[t] = PKFMA(wpair(w1[t], i), xp[u][i], c1[t]);
      }
  }
#pragma unroll
  for (int t = 0; t < NT_; ++t) { s0[t] = c0[t][0] + c0[t][1]; s1[t] = c1[t][0] + c1[t][1]; }
#pragma unroll
  for (int t = 0; t < NT_; ++t) wave_sum2(s0[t], s1[t]);
}

// two query heads against one 32-key register tile, interleaved; per head exactly AttnTile32::accumulate + reduce from
// (m_run, l_run, acc) = (-inf, 0, 0), i.e. attn_oproj_kernel's arithmetic
template <typename Tile>
__device__ __forceinline__ void attn2(const Tile& tile, const float* qA, const float* qB, float* pA, float* pB, int cnt, int lane, float* oA, float* oB) {
  const int t = lane & 31, half = lane >> 5;
  f32x2 saA = f32x2{0.f, 0.f}, sbA = saA, saB = saA, sbB = saA;
#pragma unroll
  for (int i = 0; i < Tile::NK; ++i) {
    const f32x4 qa = *reinterpret_cast<const f32x4*>(qA + (half * Tile::NK + i) * 4);
    const f32x4 qb = *reinterpret_cast<const f32x4*>(qB + (half * Tile::NK + i) * 4);
    const f32x2 k01 = f32x2{tile.k[i][0], tile.k[i][1]}, k23 = f32x2{tile.k[i][2], tile.k[i][3]};
    if (i & 1) {
      sbA = PKFMA((f32x2{qa[0], qa[1]}), k01, sbA); sbA = PKFMA((f32x2{qa[2], qa[3]}), k23, sbA);
      sbB = PKFMA((f32x2{qb[0], qb[1]}), k01, sbB); sbB = PKFMA((f32x2{qb[2], qb[3]}), k23, sbB);
    } else {
      saA = PKFMA((f32x2{qa[0], qa[1]}), k01, saA); saA = PKFMA((f32x2{qa[2], qa[3]}), k23, saA);
      saB = PKFMA((f32x2{qb[0], qb[1]}), k01, saB); saB = PKFMA((f32x2{qb[2], qb[3]}), k23, saB);
    }
  }
  float sA = (saA[0] + saA[1]) + (sbA[0] + sbA[1]), sB = (saB[0] + saB[1]) + (sbB[0] + sbB[1]);
  sA = xor32_sum(sA); sB = xor32_sum(sB);
  const bool valid = t < cnt;
  if (!valid) { sA = -INFINITY; sB = -INFINITY; }
  float m_runA = -INFINITY, l_runA = 0.f, m_runB = -INFINITY, l_runB = 0.f;
  const float m_newA = fmaxf(m_runA, wave_max(sA)), m_newB = fmaxf(m_runB, wave_max(sB));
  const float pa = valid ? __expf(sA - m_newA) : 0.f, pb = valid ? __expf(sB - m_newB) : 0.f;
  const float alphaA = __expf(m_runA - m_newA), alphaB = __expf(m_runB - m_newB);
  l_runA = l_runA * alphaA + wave_sum(half == 0 ? pa : 0.f);
  l_runB = l_runB * alphaB + wave_sum(half == 0 ? pb : 0.f);
  __builtin_amdgcn_wave_barrier();
  if (half == 0) { pA[t] = pa; pB[t] = pb; }
  __builtin_amdgcn_wave_barrier();
  const int tpar = lane / Tile::LPR;
  f32x4 accA = (f32x4)(0.f), accB = (f32x4)(0.f);
  accA *= alphaA; accB *= alphaB;
  f32x2 a01 = f32x2{accA[0], accA[1]}, a23 = f32x2{accA[2], accA[3]}, b01 = f32x2{accB[0], accB[1]}, b23 = f32x2{accB[2], accB[3]};
#pragma unroll
  for (int i = 0; i < Tile::NK; ++i) {
    const float va = pA[tpar + Tile::TP * i], vb = pB[tpar + Tile::TP * i];
    const f32x2 v01 = f32x2{tile.v[i][0], tile.v[i][1]}, v23 = f32x2{tile.v[i][2], tile.v[i][3]};
    a01 = PKFMA((f32x2{va, va}), v01, a01); a23 = PKFMA((f32x2{va, va}), v23, a23);
    b01 = PKFMA((f32x2{vb, vb}), v01, b01); b23 = PKFMA((f32x2{vb, vb}), v23, b23);
  }
  accA[0] = a01[0]; accA[1] = a01[1]; accA[2] = a23[0]; accA[3] = a23[1];
  accB[0] = b01[0]; accB[1] = b01[1]; accB[2] = b23[0]; accB[3] = b23[1];
  __builtin_amdgcn_wave_barrier();
  accA = Tile::reduce(accA);
  accB = Tile::reduce(accB);
  if (lane < Tile::LPR) {
    *reinterpret_cast<f32x4*>(oA + 4 * lane) = accA * (1.f / l_runA);
    *reinterpret_cast<f32x4*>(oB + 4 * lane) = accB * (1.f / l_runB);
  }
}

__device__ __forceinline__ void consumer(const DecPersistArgs& a, char* smem, int cu, int w, int lane_in) {
  using Tile = AttnTile32<float, HD>;
  int lane = lane_in;
  float* sqw = reinterpret_cast<float*>(smem + LDS_SQW) + w * 512;   // wave-private: q of heads 2 w, 2 w + 1 | k | v of the new position
  float* satt = reinterpret_cast<float*>(smem + LDS_ATT);
  float* pbuf = reinterpret_cast<float*>(smem + LDS_P) + w * 64;
  float* sxa = reinterpret_cast<float*>(smem + LDS_XA);   // two landing buffers: a fast wave may already sweep the next vector
  float* sxb = reinterpret_cast<float*>(smem + LDS_XB);   // while a slow one still reads the previous one
  u64* G = a.gran;
  unsigned sync_t = 0, pcnt_t = 0, ep = 0;
  unsigned base = 0;   // global slot index of the current layer's first slot
  const int frame = a.frame_ptr ? *a.frame_ptr : 0;
  int token = 0;
  const float qscale = a.qscale;
  auto stamp = [&](int p, int l, int e) {
    if (a.dbg && cu == 0 && w == 0 && lane == 0) a.dbg[((size_t)p * (a.n_layers + 1) + l) * 16 + e] = __builtin_amdgcn_s_memrealtime();
  };
  for (int p = 0; p < a.n_pass; ++p) {
    // ---- the position's input row, straight into the register layout of the K = 1024 launches ---------------------
    float x16[16];
    {
      const float* src = p == 0 ? a.x_pos0 : (p == 1 ? a.x_pos1 : a.tok_table + ((size_t)(p - 1) * a.V + token) * H);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 xa = *reinterpret_cast<const f32x4*>(src + lane * 8 + u * 512), xb = *reinterpret_cast<const f32x4*>(src + lane * 8 + u * 512 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x16[8 * u + e] = xa[e]; x16[8 * u + 4 + e] = xb[e]; }
      }
    }
    const int pos = p, cnt = p + 1;
    for (int l = 0; l < a.n_layers; ++l) {
      // per-lane addresses are recomputed in every layer-pass (a few VALU ops) instead of being hoisted out of the
      // pass / layer loops, where ~100 of them stayed live and spilled
      asm volatile("" : "+v"(lane));
      stamp(p, l, 0);
      const bool kv_only = a.kv_only_pass0 && p == 0 && l == a.n_layers - 1;
      float res[4];
      pick4(x16, cu, res);   // the residual values of this CU's o_proj rows
      // ---- QKV: tasks 3 cu + w on consumers 0-2 (RMSNorm prologue, RoPE + cache append epilogue) ------------------
      ++ep;
      if (w < 3) {
        f32x4 la[2], lb[2];
        load_ln(a.ln1[l], lane, la, lb);
        const int t = 3 * cu + w, head = t >> 6, hi = t & 63;
        const float cs = a.cos_tab[pos * 64 + hi], sn = a.sin_tab[pos * 64 + hi];
        f32x2 xp[2][4];
        normed_x16(x16, la, lb, a.eps, xp);
        const unsigned si = base + S_QKV + w;
        wait_landed(a, si, lane);
        float v0, v1;
        dot_pair(slot_ptr(smem, si), xp, lane, v0, v1);
        release_slot(si, lane);
        float o0, o1;
        int i0, i1;
        if (head < NQ + NKV) {
          o0 = v0 * cs - v1 * sn;
          o1 = v1 * cs + v0 * sn;
          if (head < NQ) {
            o0 = o0 * qscale; o1 = o1 * qscale;
            i0 = head * HD + hi; i1 = i0 + 64;
          } else {
            const int j = head - NQ;
            i0 = NQ * HD + j * HD + hi; i1 = i0 + 64;
            if (lane == 0) {
              st_sc1(a.kcache[l] + ((size_t)(j * (HD / 4) + (hi >> 2)) * a.lmax + pos) * 4 + (hi & 3), o0);
              st_sc1(a.kcache[l] + ((size_t)(j * (HD / 4) + ((hi + 64) >> 2)) * a.lmax + pos) * 4 + (hi & 3), o1);
            }
          }
        } else {
          const int j = head - NQ - NKV;
          o0 = v0; o1 = v1;
          i0 = (NQ + NKV) * HD + j * HD + 2 * hi; i1 = i0 + 1;
          if (lane == 0) {
            st_sc1(a.vcache[l] + ((size_t)j * a.lmax + pos) * HD + 2 * hi, o0);
            st_sc1(a.vcache[l] + ((size_t)j * a.lmax + pos) * HD + 2 * hi + 1, o1);
          }
        }
        if (lane < 2) publish(G + GQ + (lane ? i1 : i0), ep, lane ? o1 : o0);
      }
      stamp(p, l, 1);
      if (kv_only) {   // nothing of this position is read again: the next position starts from its own input row
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // its K / V reach later passes through the cache: they have left this wave
        base += 3;
        continue;
      }
      // ---- this wave's q (two heads) + the new k / v of its kv-head, into its own LDS strip; the K / V of earlier positions
      // (sc1 loads from the cache) are requested behind the first sweep and land while the granules are still on their way ----
      const int jkv = w >> 1;
      Tile tile;
      const int tk = lane & 31, half = lane >> 5, dg = lane & 31, tpar = lane >> 5;
      {
        float qv[8];
        gather_pairs<4>(a, ep,
                        [&](int q) { return GQ + (q < 2 ? 256 * w + 128 * q : (q == 2 ? NQ * HD : (NQ + NKV) * HD) + HD * jkv) + 2 * lane; },
                        [&]() {
                          if (pos > 0) {
                            const float* kc = a.kcache[l] + (size_t)jkv * (HD / 4) * a.lmax * 4;
                            const float* vc = a.vcache[l] + (size_t)jkv * a.lmax * HD;
                            const int tp = min(tk, pos - 1);
#pragma unroll
                            for (int i = 0; i < 16; ++i) tile.k[i] = ld_sc1(kc, (unsigned)(((half * 16 + i) * a.lmax + tp) * 4));
#pragma unroll
                            for (int i = 0; i < 16; ++i) tile.v[i] = ld_sc1(vc, (unsigned)(min(tpar + 2 * i, pos - 1) * HD + 4 * dg));
                          } else {   // position 0: every lane takes the current key below; defined values keep the tile out of the loop-carried state
#pragma unroll
                            for (int i = 0; i < 16; ++i) { tile.k[i] = (f32x4)(0.f); tile.v[i] = (f32x4)(0.f); }
                          }
                        },
                        qv, lane, 0x100u | (unsigned)(p << 16) | (unsigned)(l << 12));
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x2*>(sqw + 128 * q + 2 * lane) = f32x2{qv[2 * q], qv[2 * q + 1]};
      }
      stamp(p, l, 2);
      // ---- attention: heads 2 w, 2 w + 1 (attn_oproj_kernel's tile arithmetic) -----------------------------------
      {
        const float* kn = sqw + 256;
        const float* vn = sqw + 384;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (tk >= pos) tile.k[i] = *reinterpret_cast<const f32x4*>(kn + (half * 16 + i) * 4);
          if (tpar + 2 * i >= pos) tile.v[i] = *reinterpret_cast<const f32x4*>(vn + 4 * dg);
        }
        stamp(p, l, 11);
        if (!(a.flags & 16)) attn2(tile, sqw, sqw + HD, pbuf, pbuf + 32, cnt, lane, satt + (2 * w) * HD, satt + (2 * w + 1) * HD);
      }
      stamp(p, l, 12);
      csync(a, sync_t, lane);
      stamp(p, l, 3);
      // ---- o_proj + residual: rows 4 cu + 2 w, + 1 on consumers 0, 1 (attn_oproj_kernel's row arithmetic) -----------
      ++ep;
      if (w < 2) {
        const unsigned si = base + S_O + w;
        wait_landed(a, si, lane);
        const char* slot = slot_ptr(smem, si);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(satt + lane * 16), x1 = *reinterpret_cast<const f32x4*>(satt + lane * 16 + 4);
        const f32x4 x2 = *reinterpret_cast<const f32x4*>(satt + lane * 16 + 8), x3 = *reinterpret_cast<const f32x4*>(satt + lane * 16 + 12);
        float o2[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const u32x4 w0 = *reinterpret_cast<const u32x4*>(slot + rr * 2048 + lane * 32);
          const u32x4 w1 = *reinterpret_cast<const u32x4*>(slot + rr * 2048 + lane * 32 + 16);
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0 = fmaf((e & 1) ? bf16_hi(w0[e >> 1]) : bf16_lo(w0[e >> 1]), x0[e], s0);
            s1 = fmaf((e & 1) ? bf16_hi(w0[2 + (e >> 1)]) : bf16_lo(w0[2 + (e >> 1)]), x1[e], s1);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0 = fmaf((e & 1) ? bf16_hi(w1[e >> 1]) : bf16_lo(w1[e >> 1]), x2[e], s0);
            s1 = fmaf((e & 1) ? bf16_hi(w1[2 + (e >> 1)]) : bf16_lo(w1[2 + (e >> 1)]), x3[e], s1);
          }
          float s = s0 + s1;
          s += dpp_all<0xB1>(s);
          s += dpp_all<0x4E>(s);
          s += dpp_all<0x141>(s);
          s += dpp_all<0x140>(s);
          s = xor16_sum(s);
          s = xor32_sum(s);
          o2[rr] = (w ? res[2 + rr] : res[rr]) + s * 1.f;
        }
        release_slot(si, lane);
        if (lane < 2) publish(G + GO + 4 * cu + 2 * w + lane, ep, lane ? o2[1] : o2[0]);
      }
      stamp(p, l, 4);
      f32x4 la[2], lb[2];
      load_ln(a.ln2[l], lane, la, lb);
      gather_x16(a, GO, ep, sxa, sync_t, w, x16, lane, 0x200u | (unsigned)(p << 16) | (unsigned)(l << 12));
      pick4(x16, cu, res);   // the residual values of this CU's down_proj rows
      stamp(p, l, 5);
      // ---- gate / up + SwiGLU: tasks 32 cu + w + 4 i, four at a time --------------------------------------------------
      ++ep;
      {
        f32x2 xp[2][4];
        normed_x16(x16, la, lb, a.eps, xp);
        float act[8];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          wait_landed(a, base + S_GU + w + 4 * (4 * g + 3), lane);   // slots land in order
          stamp(p, l, 13 + g);
          const char* sl[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) sl[i] = slot_ptr(smem, base + S_GU + w + 4 * (4 * g + i));
          float v0[4], v1[4];
          dot_pairs<4>(sl, xp, lane, v0, v1);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane < 4) {
            const unsigned si = base + S_GU + w + 4 * (4 * g + lane);
            LW(slot_gen)[si % RING] = si / RING + 1u;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) act[4 * g + i] = (v0[i] / (1.f + __expf(-v0[i]))) * v1[i];
        }
        float mine = act[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) mine = lane == i ? act[i] : mine;
        if (lane < 8) publish(G + GA + 32 * cu + w + 4 * lane, ep, mine);
      }
      stamp(p, l, 6);
      // ---- down_proj + residual: consumer kw gathers ITS quarter of the SwiGLU vector into registers (no LDS, no barrier)
      // and multiplies quarter kw of rows 4 cu .. + 3 (gemv1_kernel<.., U = 4, KS = 4>) ----------------------------------
      {
        float aq[32];
        if (!(a.flags & 8))
          gather_pairs<16>(a, ep, [&](int q) { return GA + w * 2048 + 512 * (q >> 2) + 8 * lane + 2 * (q & 3); }, NoMid(), aq, lane,
                           0x300u | (unsigned)(p << 16) | (unsigned)(l << 12));
        else
#pragma unroll
          for (int i = 0; i < 32; ++i) aq[i] = 0.f;
        stamp(p, l, 7);
        ++ep;
        wait_landed(a, base + S_DN + 12 + w, lane);
        f32x2 c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = f32x2{0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          u32x4 wr[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) wr[r] = *reinterpret_cast<const u32x4*>(slot_ptr(smem, base + S_DN + 4 * r + w) + u * 1024 + lane * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = PKFMA(wpair(wr[r], i), (f32x2{aq[8 * u + 2 * i], aq[8 * u + 2 * i + 1]}), c[r]);
        }
        float s[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = c[r][0] + c[r][1];
        wave_sum2(s[0], s[1]);
        wave_sum2(s[2], s[3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 4) {
          const unsigned si = base + S_DN + 4 * lane + w;
          LW(slot_gen)[si % RING] = si / RING + 1u;
          LF(part)[lane * 4 + w] = lane == 0 ? s[0] : (lane == 1 ? s[1] : (lane == 2 ? s[2] : s[3]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((lu32*)LW(pcnt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pcnt_t += NCW;
        if (w == 0) {
          lds_wait(a, LW(pcnt), pcnt_t, 0x30u, lane);
          if (lane < 4) {
            float v = LF(part)[lane * 4];
            v += LF(part)[lane * 4 + 1];
            v += LF(part)[lane * 4 + 2];
            v += LF(part)[lane * 4 + 3];
            const float r = lane == 0 ? res[0] : (lane == 1 ? res[1] : (lane == 2 ? res[2] : res[3]));
            publish(G + GX + 4 * cu + lane, ep, r + v * 1.f);
          }
        }
      }
      stamp(p, l, 8);
      gather_x16(a, GX, ep, sxb, sync_t, w, x16, lane, 0x400u | (unsigned)(p << 16) | (unsigned)(l << 12));
      stamp(p, l, 9);
      base += SLOTS_LAYER;
    }
    if (a.dbg_x && cu == 0 && w == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        *reinterpret_cast<f32x4*>(a.dbg_x + (size_t)p * H + lane * 8 + u * 512) = f32x4{x16[8 * u], x16[8 * u + 1], x16[8 * u + 2], x16[8 * u + 3]};
        *reinterpret_cast<f32x4*>(a.dbg_x + (size_t)p * H + lane * 8 + u * 512 + 4) = f32x4{x16[8 * u + 4], x16[8 * u + 5], x16[8 * u + 6], x16[8 * u + 7]};
      }
    }
    if (p == 0) continue;
    // ---- head of codebook p: final norm, rows 2 t, 2 t + 1 of audio_head[p - 1], fused arg-max (EPI_ARGMAX / PRO_TOKNORM) ----
    asm volatile("" : "+v"(lane));
    stamp(p, a.n_layers, 0);
    ++ep;
    {
      f32x4 la[2], lb[2];
      load_ln(a.final_norm, lane, la, lb);
      f32x2 xp[2][4];
      normed_x16(x16, la, lb, a.eps, xp);
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll 1
      for (int j = w; j < SLOTS_HEAD; j += 4) {
        const unsigned si = base + j;
        wait_landed(a, si, lane);
        float v0, v1;
        dot_pair(slot_ptr(smem, si), xp, lane, v0, v1);
        release_slot(si, lane);
        const int t = j < 4 ? 4 * cu + j : (H + cu);
        if (2 * t < a.V) {
          float tv = v0;
          int ti = 2 * t;
          if (2 * t + 1 < a.V && v1 > v0) { tv = v1; ti = 2 * t + 1; }
          if (tv > bv || (tv == bv && ti < bi)) { bv = tv; bi = ti; }
        }
      }
      if (lane == 0) { LF(amv)[w] = bv; LI(ami)[w] = bi; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add((lu32*)LW(pcnt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      pcnt_t += NCW;
      if (w == 0) {
        lds_wait(a, LW(pcnt), pcnt_t, 0x31u, lane);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const float tv = LF(amv)[k];
          const int ti = LI(ami)[k];
          if (tv > bv || (tv == bv && ti < bi)) { bv = tv; bi = ti; }
        }
        if (lane < 2) publish(G + GH + 2 * cu + lane, ep, lane ? __int_as_float(bi) : bv);
      }
      base += SLOTS_HEAD;
    }
    stamp(p, a.n_layers, 1);
    {
      // every consumer reduces all 256 (value, index) pairs itself: no LDS exchange, the token is wave-uniform in all four
      float pv[8];
      gather_pairs<4>(a, ep, [&](int q) { return GH + 128 * q + 2 * lane; }, NoMid(), pv, lane, 0x500u | (unsigned)(p << 16));
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float val = pv[2 * k];
        const int idx = __float_as_int(pv[2 * k + 1]);
        if (val > bv || (val == bv && idx < bi)) { bv = val; bi = idx; }
      }
      wave_argmax(bv, bi);
      token = min(max(__builtin_amdgcn_readfirstlane(bi), 0), a.V - 1);   // (a give-up leaves garbage: stay inside the table)
    }
    stamp(p, a.n_layers, 2);
    {
      const size_t slot = (size_t)frame * a.C + p;
      if (cu == 0 && w == 0 && lane == 0) a.ring[slot] = token;
      if (a.forced) token = min(max(__builtin_amdgcn_readfirstlane((int)a.forced[slot]), 0), a.V - 1);
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(NTHREADS) void dec_persist_kernel(DecPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x < sizeof(Misc) / 4) reinterpret_cast<unsigned*>(&g_misc)[threadIdx.x] = 0u;
  __syncthreads();
  if (wave == 0) loader<NT>(a, smem, blockIdx.x, lane);
  else consumer(a, smem, blockIdx.x, wave - 1, lane);
}
static inline int configure() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_persist_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_persist_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}
static inline int launch(hipStream_t st, const DecPersistArgs& a, int nt) {
  if (a.n_layers < 1 || a.n_layers > 4 || a.n_pass < 1 || a.n_pass > a.lmax || a.V < 8 || a.V > 2 * (H + 2)) return -2;
  if (nt) hipLaunchKernelGGL(dec_persist_kernel<true>, dim3(NCU), dim3(NTHREADS), LDS_BYTES, st, a);
  else hipLaunchKernelGGL(dec_persist_kernel<false>, dim3(NCU), dim3(NTHREADS), LDS_BYTES, st, a);
  return (int)hipGetLastError();
}
}  // namespace dpk
#endif  // CSM_DEC_PERSIST_KERNEL

