// Shared device helpers for the gfx950 kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CSM_WAVE 64

// ---- in-step timeline probe (tools/b1_timeline.py builds libcsm_hip_timeline.so with -DCSM_TIMELINE): every workgroup of every
// decode-path launch records the 100 MHz constant clock (s_memrealtime: the same counter on all XCDs) at kernel entry and after
// its last store into this launch's slot [2048 workgroups][2] of the debug buffer (csm_set_debug_buffer); workgroup 0 tags the
// slot with the kernel kind and grid size.  Compiled out of the product build (an s_memrealtime at entry is an SMEM wait).
#ifdef CSM_TIMELINE
// (round 6: the time stamp is taken unconditionally and the slot pointer is only looked at in TL_END -- `ptr ? stamp : 0` made every
//  kernel of the probe build start with a kernarg read and its wait, i.e. exactly the entry cost the product build no longer has)
#define TL_BEGIN(ptr) const uint32_t tl_t0_ = (uint32_t)__builtin_amdgcn_s_memrealtime(); uint32_t* const tl_p_ = (ptr)
#define TL_END(kind)                                                                                                   \
  do {                                                                                                                 \
    if (tl_p_ && threadIdx.x == 0) {                                                                                   \
      const unsigned b_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                              \
      if (b_ < 2047u) { tl_p_[2 * b_] = tl_t0_; tl_p_[2 * b_ + 1] = (uint32_t)__builtin_amdgcn_s_memrealtime(); }      \
      if (b_ == 0) tl_p_[4094] = (uint32_t)(kind) | ((gridDim.x * gridDim.y * gridDim.z) << 8);                        \
    }                                                                                                                  \
  } while (0)
#else
#define TL_BEGIN(ptr)
#define TL_END(kind)
#endif

typedef uint16_t bf16_t;  // raw bf16 bits; all arithmetic is done in fp32
struct fp8_t { uint8_t v; };  // OCP e4m3fn byte (gfx950 v_cvt_*_fp8 is OCP, not fnuz); distinct type for overloads
typedef __attribute__((ext_vector_type(2))) float f32x2;

typedef __attribute__((ext_vector_type(4))) float f32x4;
// fused multiply-add on a float pair (v_pk_fma_f32): explicit, so the rounding does not hang on -ffp-contract
#define PKFMA(a, b, c) __builtin_elementwise_fma((a), (b), (c))
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even fp32 -> bf16 (finite inputs; NaN stays NaN)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// Load 8 consecutive weights as fp32.  bf16: one 16-byte load; fp32: two.
template <typename WT>
struct W8;
template <>
struct W8<bf16_t> {
  u32x4 r;
  __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const u32x4*>(p); }
  __device__ __forceinline__ void load_nt(const bf16_t* p) {
    r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  }
  __device__ __forceinline__ void zero() { r = (u32x4)(0u); }
  __device__ __forceinline__ float get(int i) const {
    uint32_t u = r[i >> 1];
    return (i & 1) ? bf16_hi(u) : bf16_lo(u);
  }
  __device__ __forceinline__ f32x2 pair(int i) const {  // weights 2i, 2i+1 (one dword): two unpack ops
    f32x2 p;
    p[0] = bf16_lo(r[i]);
    p[1] = bf16_hi(r[i]);
    return p;
  }
};
template <>
struct W8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const f32x4*>(p);
    b = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ void load_nt(const float* p) {
    a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
  }
  __device__ __forceinline__ void zero() { a = (f32x4)(0.f); b = (f32x4)(0.f); }
  __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b[i - 4]; }
  __device__ __forceinline__ f32x2 pair(int i) const {
    f32x2 p;
    p[0] = get(2 * i);
    p[1] = get(2 * i + 1);
    return p;
  }
};

template <>
struct W8<fp8_t> {  // 8 e4m3 weights = one 8-byte load; widened by v_cvt_pk_f32_fp8 (exact)
  uint2 r;
  __device__ __forceinline__ void load(const fp8_t* p) { r = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void load_nt(const fp8_t* p) {
    const uint64_t u = __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(p));
    r.x = (uint32_t)u;
    r.y = (uint32_t)(u >> 32);
  }
  __device__ __forceinline__ void zero() { r.x = 0u; r.y = 0u; }
  __device__ __forceinline__ float get(int i) const {
    const int w = (int)(i < 4 ? r.x : r.y);   // word_sel must be a literal
    const f32x2 f = (i & 2) ? __builtin_amdgcn_cvt_pk_f32_fp8(w, true) : __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
    return (i & 1) ? f[1] : f[0];
  }
  __device__ __forceinline__ f32x2 pair(int i) const {  // one v_cvt_pk_f32_fp8 per pair
    const int w = (int)(i < 2 ? r.x : r.y);
    return (i & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8(w, true) : __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
  }
};

// Wavefront reductions on the DPP path (row = 16 lanes): quad swaps, half-row and row mirrors, then the two
// cross-row broadcasts of gfx9 (row_bcast15/31); the total lands in lane 63 and is returned wave-uniform through
// v_readlane.  Six dependent VALU ops instead of six ds_bpermute round trips through the LDS crossbar -- these
// reductions sit on the critical path of every latency-bound decode kernel.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_all(float v) {  // every lane has a valid source: lets the move fold into its user
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_all<0xB1>(v);            // quad_perm [1,0,3,2]
  v += dpp_all<0x4E>(v);            // quad_perm [2,3,0,1]
  v += dpp_all<0x141>(v);           // row_half_mirror
  v += dpp_all<0x140>(v);           // row_mirror: every lane of a row holds the row sum
  v += dpp_mov<0x142, 0xa>(0.f, v); // row_bcast15 -> rows 1, 3
  v += dpp_mov<0x143, 0xc>(0.f, v); // row_bcast31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// two independent sums for the price of one: fold a's and b's upper halves onto the lower ones with one
// v_permlane32_swap (lanes 0-31 carry a, lanes 32-63 carry b), reduce the 32-lane halves, read lanes 31 and 63
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  float v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  v += dpp_all<0xB1>(v);
  v += dpp_all<0x4E>(v);
  v += dpp_all<0x141>(v);
  v += dpp_all<0x140>(v);
  v += dpp_mov<0x142, 0xa>(0.f, v);
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_all<0xB1>(v));
  v = fmaxf(v, dpp_all<0x4E>(v));
  v = fmaxf(v, dpp_all<0x141>(v));
  v = fmaxf(v, dpp_all<0x140>(v));
  v = fmaxf(v, dpp_mov<0x142, 0xa>(-INFINITY, v));
  v = fmaxf(v, dpp_mov<0x143, 0xc>(-INFINITY, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ int wave_min_i32(int v) {
  auto mv = [](auto tag, int old, int x) {
    return __builtin_amdgcn_update_dpp(old, x, decltype(tag)::ctrl, decltype(tag)::mask, 0xf, false);
  };
  struct Q1 { enum { ctrl = 0xB1, mask = 0xf }; }; struct Q2 { enum { ctrl = 0x4E, mask = 0xf }; };
  struct HM { enum { ctrl = 0x141, mask = 0xf }; }; struct RM { enum { ctrl = 0x140, mask = 0xf }; };
  struct B15 { enum { ctrl = 0x142, mask = 0xa }; }; struct B31 { enum { ctrl = 0x143, mask = 0xc }; };
  v = min(v, mv(Q1{}, v, v));
  v = min(v, mv(Q2{}, v, v));
  v = min(v, mv(HM{}, v, v));
  v = min(v, mv(RM{}, v, v));
  v = min(v, mv(B15{}, 0x7fffffff, v));
  v = min(v, mv(B31{}, 0x7fffffff, v));
  return __builtin_amdgcn_readlane(v, 63);
}
// wave-uniform argmax with lowest-index tie-break: max value first, then the smallest index that attains it
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
  const float m = wave_max(v);
  idx = wave_min_i32(v == m ? idx : 0x7fffffff);
  v = m;
}
// butterfly partner sums across the two halves / across odd-even rows of 16 lanes (gfx950 v_permlane*_swap):
// every lane ends with x[l] + x[l ^ 32] (resp. x[l] + x[l ^ 16]) without the LDS crossbar
__device__ __forceinline__ float xor32_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_sum(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// One activation value into the three planes of the next launch's B operand: exact truncation split, element
// (row m, column k) at fragment position ((k/128 * 4 + (k/32)%4) * 64 + ((k/8)%4) * 16 + m) * 8 + k%8.
// `one` (decode_precision = bf16, the reference's own arithmetic class, README.md:73): ONE plane, the value rounded to
// nearest-even bf16 like the reference's eager bf16 execution rounds every activation; planes 1 and 2 are not written
// and the consumer multiplies plane 0 only.
__device__ __forceinline__ void store_planes(bf16_t* planes, size_t plane_stride, int k, int m, float v, bool one = false) {
  if (one) {
    planes[((size_t)((k >> 7) * 4 + ((k >> 5) & 3)) * 64 + ((k >> 3) & 3) * 16 + m) * 8 + (k & 7)] = f32_to_bf16(v);
    return;
  }
  const uint32_t h = __float_as_uint(v) & 0xffff0000u;
  const float r = v - __uint_as_float(h);
  const uint32_t md = __float_as_uint(r) & 0xffff0000u;
  const float l = r - __uint_as_float(md);
  const size_t off = ((size_t)((k >> 7) * 4 + ((k >> 5) & 3)) * 64 + ((k >> 3) & 3) * 16 + m) * 8 + (k & 7);
  planes[off] = (bf16_t)(h >> 16);
  planes[plane_stride + off] = (bf16_t)(md >> 16);
  planes[2 * plane_stride + off] = (bf16_t)(__float_as_uint(l) >> 16);
}

// prefill planes are plain row-major [3][rows][K] bf16 (the prefill GEMM stages them to LDS itself): 4 consecutive
// columns of one row, one 8-byte store per plane; `p` points at plane 0, element (row, k0)
// plane_stride == 0 selects the ONE-plane form (prefill_precision = bf16): the value rounded to nearest-even bf16,
// like the reference's own bf16 execution rounds every activation -- one MFMA per weight fragment instead of three.
__device__ __forceinline__ void store_rowplanes4(bf16_t* p, size_t plane_stride, const f32x4& v) {
  if (plane_stride == 0) {
    *reinterpret_cast<uint2*>(p) = make_uint2((uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16),
                                              (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16));
    return;
  }
  uint32_t hw[2], mw[2], lw[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float v0 = v[2 * q], v1 = v[2 * q + 1];
    const uint32_t h0 = __float_as_uint(v0) & 0xffff0000u, h1 = __float_as_uint(v1) & 0xffff0000u;
    const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
    const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);
    hw[q] = (h0 >> 16) | h1;
    mw[q] = (m0 >> 16) | m1;
    lw[q] = (__float_as_uint(s0) >> 16) | (__float_as_uint(s1) & 0xffff0000u);
  }
  *reinterpret_cast<uint2*>(p) = make_uint2(hw[0], hw[1]);
  *reinterpret_cast<uint2*>(p + plane_stride) = make_uint2(mw[0], mw[1]);
  *reinterpret_cast<uint2*>(p + 2 * plane_stride) = make_uint2(lw[0], lw[1]);
}
__device__ __forceinline__ void store_rowplane1(bf16_t* p, size_t plane_stride, float v) {
  if (plane_stride == 0) { p[0] = f32_to_bf16(v); return; }
  const uint32_t h = __float_as_uint(v) & 0xffff0000u;
  const float r = v - __uint_as_float(h);
  const uint32_t md = __float_as_uint(r) & 0xffff0000u;
  const float l = r - __uint_as_float(md);
  p[0] = (bf16_t)(h >> 16);
  p[plane_stride] = (bf16_t)(md >> 16);
  p[2 * plane_stride] = (bf16_t)(__float_as_uint(l) >> 16);
}
// two consecutive columns k0, k0+1 (k0 even) of row m: one 4-byte store per plane
__device__ __forceinline__ void store_planes2(bf16_t* planes, size_t plane_stride, int k0, int m, float v0, float v1, bool one = false) {
  if (one) {
    const size_t off1 = ((size_t)((k0 >> 7) * 4 + ((k0 >> 5) & 3)) * 64 + ((k0 >> 3) & 3) * 16 + m) * 8 + (k0 & 7);
    *reinterpret_cast<uint32_t*>(planes + off1) = (uint32_t)f32_to_bf16(v0) | ((uint32_t)f32_to_bf16(v1) << 16);
    return;
  }
  const uint32_t h0 = __float_as_uint(v0) & 0xffff0000u, h1 = __float_as_uint(v1) & 0xffff0000u;
  const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
  const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
  const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);
  const size_t off = ((size_t)((k0 >> 7) * 4 + ((k0 >> 5) & 3)) * 64 + ((k0 >> 3) & 3) * 16 + m) * 8 + (k0 & 7);
  *reinterpret_cast<uint32_t*>(planes + off) = (h0 >> 16) | h1;
  *reinterpret_cast<uint32_t*>(planes + plane_stride + off) = (m0 >> 16) | m1;
  *reinterpret_cast<uint32_t*>(planes + 2 * plane_stride + off) = (__float_as_uint(s0) >> 16) | (__float_as_uint(s1) & 0xffff0000u);
}
// four consecutive columns k0..k0+3 (k0 % 4 == 0) of row m: one 8-byte store per plane
__device__ __forceinline__ void store_planes4(bf16_t* planes, size_t plane_stride, int k0, int m, const f32x4& v, bool one = false) {
  if (one) {
    const size_t off1 = ((size_t)((k0 >> 7) * 4 + ((k0 >> 5) & 3)) * 64 + ((k0 >> 3) & 3) * 16 + m) * 8 + (k0 & 7);
    *reinterpret_cast<uint2*>(planes + off1) = make_uint2((uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16),
                                                          (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16));
    return;
  }
  uint32_t hw[2], mw[2], lw[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float v0 = v[2 * p], v1 = v[2 * p + 1];
    const uint32_t h0 = __float_as_uint(v0) & 0xffff0000u, h1 = __float_as_uint(v1) & 0xffff0000u;
    const float r0 = v0 - __uint_as_float(h0), r1 = v1 - __uint_as_float(h1);
    const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);
    hw[p] = (h0 >> 16) | h1;
    mw[p] = (m0 >> 16) | m1;
    lw[p] = (__float_as_uint(s0) >> 16) | (__float_as_uint(s1) & 0xffff0000u);
  }
  const size_t off = ((size_t)((k0 >> 7) * 4 + ((k0 >> 5) & 3)) * 64 + ((k0 >> 3) & 3) * 16 + m) * 8 + (k0 & 7);
  *reinterpret_cast<uint2*>(planes + off) = make_uint2(hw[0], hw[1]);
  *reinterpret_cast<uint2*>(planes + plane_stride + off) = make_uint2(mw[0], mw[1]);
  *reinterpret_cast<uint2*>(planes + 2 * plane_stride + off) = make_uint2(lw[0], lw[1]);
}

// Workgroup barrier that only waits for the LDS counter.  __syncthreads() also drains vmcnt, i.e. it would wait
// for every weight load in flight; use this one between a prologue's LDS exchange and the weight consumption.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Position of a row: per-row table | device scalar (backbone length) | launch constant (decoder pass).  Written as
// branches around VOLATILE loads on purpose: from `p ? *p : (q ? *q : c)` the compiler built ONE load through a selected
// pointer and parked the constant in scratch memory to have something to point at -- a dependent scratch round trip
// (~1 500 clocks) ahead of every load of the QKV launches, constant-position decoder passes included.
// (relaxed agent-scope atomic loads: a vector load that bypasses the non-coherent caches like the `volatile` loads of rounds 1-5 did, but
//  without the s_waitcnt vmcnt(0) the compiler puts directly behind a volatile access -- that wait stood in front of every weight load of
//  the backbone's QKV launches.  The counters are written by earlier launches of the same stream.)
__device__ __forceinline__ int row_position(const int* row_pos, int m, const int* pos_ptr, int pos_const) {
  if (row_pos) return __hip_atomic_load(row_pos + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (pos_ptr) return __hip_atomic_load(pos_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return pos_const;
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }

__device__ __forceinline__ void store_kv(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_kv(bf16_t* p, float v) { *p = f32_to_bf16(v); }
