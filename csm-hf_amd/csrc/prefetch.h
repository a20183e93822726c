// Weight streamer: a persistent kernel on a second HIP stream that walks the weight matrices of a captured frame-step
// in consumption order and pulls them into the XCD-local L2 ahead of the launches that read them.
//
// Why: a frame-step at small batch is a chain of ~770 dependent launches; each launch boundary (~1.5 us) and each
// latency-bound small launch leaves HBM idle, so the chain streams 9 GB in ~3.4 ms (2.6 TB/s) although the memory
// system sustains 6.3 TB/s.  L2 contents survive kernel boundaries (tools/ubench/prefetch.hip, E2: a 32 MiB matrix
// re-read by the next launch runs at 9.2 TB/s, a fresh one at 4.1-4.8), and a kernel on another stream runs concurrently
// with a replaying hipGraph (E5/E6; a parallel BRANCH of the graph does not: ROCm 7.2 serialises it).  The streamer
// therefore fetches, with fire-and-forget LDS-DMA loads (`global_load_lds_dwordx4`: 1 KiB per wave instruction, no
// VGPR destination, never waited for), exactly the rows each consumer workgroup will read, FROM THE XCD that workgroup
// will run on (workgroup b of every dispatch lands on XCD (b + rot) % 8; `rot` is measured at engine creation), paced
// by a launch counter the consumers bump: a segment may be fetched once the bytes fetched-but-not-yet-consumed fit a
// window (default 24 MiB of the 32 MiB aggregate L2), and is skipped when its consumer has already started.
//
// The streamer never writes model state, so results are bit-identical with it on or off; a late or absent streamer
// only costs speed.  It ENDS deterministically (round 6): the poller of every workgroup sees the launch counter reach
// `reps * n_launch` -- the last streamed launch of the last replay has started, nothing is left worth fetching -- and
// retires the workgroup as `finished` wherever its loaders are.  "No launch started for `budget_ticks`" while the
// counter is below that total therefore means a STALLED chain (two streams on one hardware queue, a launch that cannot
// become resident beside the streamer): the workgroup gives up, the give-up is counted in a lifetime counter the host
// reads before the next csm_generate (engine.hip: pf_harvest), which re-runs the stream-concurrency probe and switches
// the streamer off for the engine instead of stalling every call.
// No reference counterpart: the reference (modeling_csm.py) issues torch ops one by one.
#pragma once
#include "common.h"

// How one weight-streaming launch maps its workgroups to matrix rows; filled by the launcher (gemv_inst.inc).
struct PfGeom {
  const void* W;   // row-major [N][K]
  int N, K, esz;
  int kind;        // 0: task t -> rows 2t, 2t+1;  1: QKV epilogue (RoPE pairs, gemv.h gemv_map_task);  -1: not streamed
                   // 2: matrix-core skinny GEMM on the fragment-order copy (gemm16.h): W = the copy, grid = gx * KB
                   //    linear workgroups (x fastest), tpb = PT tiles per panel, iters = NW chunks per workgroup,
                   //    stride = gx, ntask = K / 128 chunks per tile row, K * esz / (K / 128) = bytes per chunk;
                   //    n_rope_heads > 0 marks the QKV panel -> tile map (g16_row)
  int grid, tpb;   // workgroup b, iteration it covers tasks [b*tpb + it*stride, +tpb)
  int iters, stride;
  int ntask;
  int hd, n_rope_heads;   // kind 1: head_dim, n_q + n_kv
  int exclusive;          // this launch's workgroups fill a CU's register file: nothing (no streamer) can stay resident beside it
};
constexpr int PF_STREAMER_VGPRS = 72;   // registers of weight_prefetch_kernel (checked by the build log); one wave per SIMD

// A range of consumer workgroups of one launch (device-side schedule entry).
struct PfSeg {
  const char* W;
  uint32_t row_bytes;
  int32_t N, kind, tpb, iters, stride, ntask, hd, n_rope_heads;
  int32_t b0, b1;    // consumer workgroups [b0, b1)
  int32_t owner;     // index of the consuming launch within the frame-step
  int32_t need;      // may be fetched once (launches started) >= rep * n_launch + need
};

struct PfArgs {
  const PfSeg* segs;
  int n, n_launch, reps;
  int rot;                 // workgroup b of a dispatch runs on XCD (b + rot) % 8
  const unsigned* prog;    // launches started so far (bumped by workgroup 0 of every streamed launch)
  unsigned* ticket;        // [8] per-XCD arrival counters, zeroed before the launch
  unsigned* status;        // [0] workgroups that gave up, [1] finished, [2] segments skipped as late (workgroup 0 of XCD 0),
                           // [3] workgroups retired by the end-of-chain rule, [4..7] stop record of a give-up
  unsigned* lifetime;      // never reset: [0] give-ups, [1] finished, [2] streamer launches (workgroup 0 of XCD 0)
  long long budget_ticks;  // s_memrealtime ticks (100 MHz) without a launch starting, counter below the total: give up
  int skip_late;           // 1: a segment whose consumer has already started is skipped
  int poll_sleep;          // s_sleep units between two polls of the launch counter
  int seg_sleep;           // s_sleep units every loader wave idles before each segment (rate limiter)
  int stride;              // 0: contiguous 16-byte loads (every byte crosses the CU);  64 | 128: ONE dword per `stride`
                           // bytes -- the L2 fills whole lines, the CU receives 1/16 or 1/32 of the bytes
  int depth;               // prefetch loads (1 KiB each) a loader wave keeps in flight: 4 | 8 | 16 | 32, else unlimited.
                           // An unthrottled streamer floods the memory queues and the chain's latency-critical loads
                           // (activations, a few KB per launch) wait behind it
};

#ifdef CSM_PREFETCH_KERNELS   // defined by launchers.hip only: the kernels live in one translation unit
__device__ __forceinline__ unsigned pf_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

__global__ void pf_where_kernel(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = pf_xcc_id();
}

// concurrency self-test: the waiter (second stream) spins until the setter (engine stream, submitted AFTER it) has
// raised the flag, or until `budget_ticks` of the 100 MHz clock pass; out = 1 if it saw the flag.  Two HIP streams can
// share one hardware queue, and then the streamer would hold up the very chain it is meant to feed.
__global__ void pf_wait_kernel(const unsigned* flag, unsigned* out, long long budget_ticks) {
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned seen = 0;
  while (!seen && __builtin_amdgcn_s_memrealtime() - t0 < budget_ticks) {
    seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_sleep(8);
  }
  *out = seen;
}
__global__ void pf_set_kernel(unsigned* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void pf_nop_kernel(unsigned* sink) { if (sink && threadIdx.x == 1024) *sink = 1u; }

// one poller wave + three loader waves per workgroup
__global__ __launch_bounds__(256) void weight_prefetch_kernel(PfArgs a) {
  __shared__ __attribute__((aligned(16))) char dump[4 * 1024];
  __shared__ unsigned s_bl;
  __shared__ int s_cur, s_state, s_done;
  const unsigned x = pf_xcc_id();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) {
    s_bl = atomicAdd(a.ticket + x, 1u);
    s_cur = 0; s_state = 0; s_done = 0;
  }
  __syncthreads();
  const unsigned bl = s_bl, nl = gridDim.x >> 3;
  if (bl >= nl) return;   // uneven placement: surplus workgroups idle (their share is dealt by the modulo below)
  volatile int* v_cur = &s_cur;
  volatile int* v_state = &s_state;
  volatile int* v_done = &s_done;
  if (wave == 0) {
    // poller: keeps the launch counter fresh in LDS, so the loaders never put a vector load (whose in-order vmcnt
    // wait would cover every prefetch load in flight) between their prefetches
    long long t_last = __builtin_amdgcn_s_memrealtime();
    int last = -1;
    const int total = a.reps * a.n_launch;
    if (lane == 0 && x == 0 && bl == 0) atomicAdd(a.lifetime + 2, 1u);
    while (*v_done < 3) {
      const int c = (int)__hip_atomic_load(a.prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == 0) *v_cur = c;
      if (c >= total) {
        // deterministic end: the last streamed launch of the last replay has started.  Loaders that are still walking
        // the schedule (a chain of launches shorter than the loaders' per-segment bookkeeping, a slow clock) stop at
        // their next segment: this is `finished`, never a give-up
        if (lane == 0) { *v_state = 3; atomicAdd(a.status + 3, 1u); }
        break;
      }
      const long long now = __builtin_amdgcn_s_memrealtime();
      if (c != last) { last = c; t_last = now; }
      // one budget without a launch starting while launches are outstanding: the chain is stalled (or has not been
      // submitted: the host submits the replays right behind the streamer).  Rounds 2-5 waited ten budgets before the
      // first launch; a chain held up BY the streamer then lost 200 ms per call
      if (now - t_last > a.budget_ticks) {
        if (lane == 0) { *v_state = 2; atomicAdd(a.status, 1u); atomicAdd(a.lifetime, 1u); }
        return;
      }
      for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) { atomicAdd(a.status + 1, 1u); atomicAdd(a.lifetime + 1, 1u); }
    return;
  }
  const unsigned slot = bl * 3u + (unsigned)(wave - 1), nslot = nl * 3u;
  auto* ldst = (__attribute__((address_space(3))) void*)(dump + wave * 1024);
  unsigned skipped = 0;
  PfSeg nxt = a.segs[0];
  for (int rep = 0; rep < a.reps; ++rep) {
    const int base = rep * a.n_launch;
    for (int e = 0; e < a.n; ++e) {
      // the descriptor of the NEXT segment is requested (scalar loads) before this one is processed: a dependent
      // ~1 us descriptor fetch per segment would cap the streamer at a few MB per microsecond
      const PfSeg seg = nxt;
      nxt = a.segs[e + 1 < a.n ? e + 1 : 0];
      const PfSeg* sp = &seg;
      const int want = base + sp->need;
      int cur = *v_cur;
      int st = *v_state;
      while (cur < want && st == 0) {
        __builtin_amdgcn_s_sleep(1);
        cur = *v_cur;
        st = *v_state;
      }
      if (st != 0) {   // 3: the chain has started its last launch (poller);  2: the poller gave up
        if (st == 2 && lane == 0 && wave == 1) {   // record of where a workgroup stopped: {segment, want, seen, rep}
          a.status[4] = (unsigned)e; a.status[5] = (unsigned)want; a.status[6] = (unsigned)cur; a.status[7] = (unsigned)rep;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && skipped && x == 0 && bl == 0 && wave == 1) atomicAdd(a.status + 2, skipped);   // the late-segment sample, also on this exit
        return;
      }
      if (a.skip_late && cur > base + sp->owner) { ++skipped; continue; }   // its consumer is already running: leave it alone
      for (int z = 0; z < a.seg_sleep; z += 8) __builtin_amdgcn_s_sleep(8);
      const int b0 = sp->b0, b1 = sp->b1, tpb = sp->tpb, iters = sp->iters, ntask = sp->ntask, N = sp->N;
      const unsigned rb = sp->row_bytes;
      const char* W = sp->W;
      // consumer workgroups of this XCD inside [b0, b1): b = bf + 8 i
      const int bf = b0 + (int)((x + 16u - (unsigned)((b0 + a.rot) & 7)) & 7u);
      if (bf >= b1) continue;
      const unsigned nbx = (unsigned)(b1 - bf + 7) >> 3;
      if (sp->kind == 2) {
        // fragment-order weights: workgroup (bx, by) reads, for each of its `tpb` tiles, `iters` consecutive chunks
        const unsigned cb = rb;                          // bytes of one (tile, chunk) fragment block
        const unsigned RB = (unsigned)iters * cb;        // contiguous bytes per tile
        const unsigned FB2 = (unsigned)tpb * RB;
        const unsigned ppb2 = (FB2 + 4095u) >> 12;
        const unsigned n_units2 = nbx * ppb2;
        const int gx = sp->stride, half2 = sp->hd >> 1, spp = half2 >> 4;
        for (unsigned u = slot; u < n_units2; u += nslot) {
          const unsigned piece = u % ppb2;
          const int b = bf + 8 * (int)(u / ppb2);
          const int bx = b % gx, by = b / gx;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned po = piece * 4096u + (unsigned)j * 1024u;
            if (po >= FB2) break;
            unsigned vo = po + (unsigned)lane * 16u;
            if (vo >= FB2) vo = po;
            const int t = (int)(vo / RB);
            const unsigned off = vo - (unsigned)t * RB;
            int tile;
            if (sp->n_rope_heads > 0) { const int head = bx / spp, s_ = bx - head * spp; tile = (head * sp->hd + t * half2 + s_ * 16) >> 4; }
            else tile = bx * tpb + t;
            tile = min(tile, ((N + 15) >> 4) - 1);     // partial last panel: stay inside the copy
            const char* src = W + ((size_t)tile * (size_t)ntask + (size_t)by * (size_t)iters) * cb + off;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, ldst, 16, 0, 0);
          }
        }
        continue;
      }
      const unsigned FB = 2u * (unsigned)tpb * rb;                 // footprint of one (workgroup, iteration)
      const unsigned PB = a.stride ? 64u * (unsigned)a.stride : 4096u;   // bytes one unit covers
      const unsigned ppb = (FB + PB - 1u) / PB;
      const unsigned n_units = nbx * (unsigned)iters * ppb;
      const int half = sp->hd >> 1;
      for (unsigned u = slot; u < n_units; u += nslot) {
        const unsigned piece = u % ppb, r = u / ppb;
        const int it = (int)(r % (unsigned)iters), i = (int)(r / (unsigned)iters);
        const int b = bf + 8 * i;
        const int t0 = b * tpb + it * sp->stride;
        if (t0 >= ntask) continue;
        const int nt = min(tpb, ntask - t0);
        // virtual footprint [0, fb) -> addresses
        unsigned fb;
        size_t base0, base1 = 0;
        unsigned split = 0xffffffffu;   // virtual offset where chunk 1 starts (RoPE pairs)
        if (sp->kind == 1) {
          const int head = t0 / half, hi = t0 - head * half;
          if (head < sp->n_rope_heads) {
            base0 = (size_t)(head * sp->hd + hi) * rb;
            base1 = base0 + (size_t)half * rb;
            split = (unsigned)nt * rb;
            fb = 2u * split;
          } else {
            base0 = (size_t)(head * sp->hd + 2 * hi) * rb;
            fb = 2u * (unsigned)nt * rb;
          }
        } else {
          base0 = (size_t)(2 * t0) * rb;
          const int rows = min(2 * nt, N - 2 * t0);
          fb = (unsigned)rows * rb;
        }
        if (a.stride) {
          unsigned vo = piece * PB + (unsigned)lane * (unsigned)a.stride;
          if (piece * PB < fb) {
            if (vo >= fb) vo = 0;
            const char* src = W + (vo >= split ? base1 + (vo - split) : base0 + vo);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, ldst, 4, 0, 0);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned vo = piece * 4096u + (unsigned)j * 1024u + (unsigned)lane * 16u;
            if (piece * 4096u + (unsigned)j * 1024u >= fb) break;    // wave-uniform
            if (vo >= fb) vo = 0;
            const char* src = W + (vo >= split ? base1 + (vo - split) : base0 + vo);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, ldst, 16, 0, 0);
          }
        }
        // throttle: at most `depth` loads of this wave outstanding before the next unit is issued
        if (a.depth == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (a.depth == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (a.depth == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (a.depth == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (a.depth == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
    atomicAdd((int*)&s_done, 1);
    if (skipped && x == 0 && bl == 0 && wave == 1) atomicAdd(a.status + 2, skipped);
  }
}
#endif  // CSM_PREFETCH_KERNELS

int launch_pf_where(hipStream_t st, unsigned* out8);
int launch_pf_concurrency_probe(hipStream_t waiter_stream, hipStream_t setter_stream, unsigned* flag, unsigned* out);
// dispatch-rate probe: `n` dependent empty launches on the chain's stream between two events, with (waiter_stream != nullptr) or without a
// kernel resident on the streamer's stream (it spins until the flag the chain raises behind the launches, or for 5 ms)
int launch_pf_rate_probe(hipStream_t waiter_stream, hipStream_t chain_stream, unsigned* flag, unsigned* out, int n, hipEvent_t ev0, hipEvent_t ev1);
int launch_weight_prefetch(hipStream_t st, int grid, const PfArgs& a);
