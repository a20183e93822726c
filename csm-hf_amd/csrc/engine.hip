// libcsm_hip.so -- engine + C ABI (include/csm_hip.h).  Orchestrates the gfx950 kernels of gemv.h,
// attn.h, gemm.h, misc.h into the three phases of the reference's generate():
//   prefill        = CSMModel.forward on the context         (modeling_csm.py:321-365)
//   decode_frame   = generate_frame after its forward        (modeling_csm.py:522-589)
//   backbone_step  = the forward of the next generate_frame  (modeling_csm.py:508-520, S = 1)
// decode_frame + backbone_step are captured once into a hipGraph and replayed per frame; the frame
// index and the backbone length live in device memory so a replay takes no parameters.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/csm_hip.h"
#define CSM_ARGS_ONLY 1  // kernel definitions live in gemv.hip / launchers.hip
#include "attn.h"
#include "attn_prefill.h"
#include "attn_oproj.h"
#include "gemm.h"
#include "gemm16.h"
#include "gemm128.h"
#include "gemm_mx.h"
#include "gemv.h"
#include "misc.h"
#include "train.h"

int launch_embed(hipStream_t st, int wdtype, int rows, const EmbedArgs& a);
int launch_rmsnorm(hipStream_t st, const float* x, int ldx, const float* w, int rows, int H, float eps, float* out,
                   int ldo, const int* frame_ptr, size_t frame_stride, int frame_add, bf16_t* planes = nullptr,
                   size_t plane_stride = 0, const float* part = nullptr, int nsplit = 0, size_t part_stride = 0, int ldp = 0,
                   uint8_t* mxq = nullptr, uint8_t* mxs = nullptr);
int launch_rope_scatter(hipStream_t st, int kvdtype, int rows, const RopeArgs& a);
int launch_swiglu_reduce(hipStream_t st, const float* part, int nsplit, size_t part_stride, int rows, int F, float* out, int ldo,
                         bf16_t* planes, size_t plane_stride, uint8_t* mxq, uint8_t* mxs);
int launch_sample(hipStream_t st, int rows, const SampleArgs& a);
int launch_kv_convert(hipStream_t st, int kvdtype, const KvConvArgs& a);
int launch_rows_iota(hipStream_t st, int* row_seq, int* row_pos, int R, int S, int past);
int launch_rows_slots(hipStream_t st, int* row_seq, int* row_pos, int R, int S, int past, const int* slots);
int launch_set_int(hipStream_t st, int* p, int v);
int launch_set_rng(hipStream_t st, uint64_t* p, uint64_t seed, uint64_t row_offset);
int configure_sample();
int launch_tile16(hipStream_t st, const void* W, void* Wt, int N, int K, int esz);
int launch_widen(hipStream_t st, int wdtype, const void* src, float* dst, size_t n);
int gemv_configure_all();
int gemm32_configure_all();
int launch_attn_prefill(hipStream_t st, int kvdtype, int B, int hd, const PrefillAttnArgs& a, int bf16_math);
int launch_ce_rows(hipStream_t st, const CeArgs& a);
int launch_ce_reduce(hipStream_t st, const float* row_loss, const int* labels, int V, int rows, double* acc);
int launch_loss_finalize(hipStream_t st, const double* acc, int frames, float* out3);
int launch_dec_input(hipStream_t st, int frames, const DecInArgs& a);
int launch_kv_shift(hipStream_t st, int kvdtype, const KvShiftArgs& a);
int launch_add_ints(hipStream_t st, int* p, int n, int delta);
int launch_gemm_dma_bf16(hipStream_t st, int epi, const GemmArgs& a);                  // gemm_mx.hip
int launch_gemm256_bf16(hipStream_t st, int epi, const GemmArgs& a, int min_wgs);     // gemm_mx.hip (gemm256.h)

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
// other translation units of the library (mimi.hip) report through the same thread-local message
int csm_set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
#define HIPCK(x)                                                                              \
  do {                                                                                        \
    hipError_t _e = (x);                                                                      \
    if (_e != hipSuccess) return fail((int)_e, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define LCK(x)                                                                                \
  do {                                                                                        \
    int _e = (x);                                                                             \
    if (_e != 0) return fail(_e > 0 ? _e : CSM_ERR_ARG, "%s failed with %d (%s:%d)", #x, _e, __FILE__, __LINE__); \
  } while (0)

struct Stack {
  csm_llama_cfg_t c{};
  std::vector<csm_layer_weights_t> layers;
  const float* final_norm = nullptr;
  const float* cos = nullptr;
  const float* sin = nullptr;
  int rope_positions = 0;
  std::vector<void*> kc, vc;
  int lmax = 0;
  int nqkv() const { return (c.n_q + 2 * c.n_kv) * c.head_dim; }
};

// What a captured frame-step depends on.  The sampling seed and the shard's row offset are NOT here: they live in
// device memory (d_rng) and are rewritten before every launch, like the frame index and the backbone length.
struct GraphKey {
  int B, topk, nsplit, per_row;
  float temperature;
  const void *noise, *forced, *ltrace, *htrace;
  bool operator<(const GraphKey& o) const { return memcmp(this, &o, sizeof(GraphKey)) < 0; }
};
struct GraphEntry {
  hipGraphExec_t exec;
  uint64_t last_use;
  // weight-streamer schedule of this frame-step (prefetch.h): device array of segments in consumption order
  PfSeg* d_segs = nullptr;
  int n_segs = 0, n_launch = 0;
  size_t sched_bytes = 0, step_bytes = 0;
};
constexpr size_t MAX_GRAPHS = 8;   // LRU bound on cached frame-step graphs

struct csm_engine {
  csm_config_t cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t pooled_own = nullptr;   // the engine's own stream of its pooled pair (== stream when the caller passed none)
  csm_weights_t w{};
  bool bound = false;
  Stack bb, dec;
  int esz_kv = 4;
  // device state
  int* d_len = nullptr;
  int* d_frame = nullptr;
  int* d_kv_start = nullptr;
  int* d_zero_count = nullptr;   // [max_frames] rows whose frame was all-zero (stop test without a per-frame host sync)
  int* d_row_done = nullptr;     // [max_batch] rows that have emitted an all-zero frame
  uint64_t* d_rng = nullptr;   // {seed, global index of row 0}: read by the sampler, written before every launch
  int64_t* ring = nullptr;
  // decode scratch
  int* attn_tickets = nullptr;   // [B * n_q]: split-merge tickets of the backbone attention (attn.h), zeroed once, self-resetting
  int fuse_attn_combine = 1;     // 2-32 rows: the last split of a (row, head) merges the partials inside the attention launch (0 = always the
                                 // attn_combine launch).  Measured (round 4): B = 16 5.09 -> 5.05 ms; B = 1 3.12 -> 3.16 and 128 rows 11.52 -> 11.59 the other way
                                 // (the ticket round trip on the tail of 32 splits costs what the launch costs), so those keep the launch
  float *q_bb = nullptr, *att_bb = nullptr, *part_bb = nullptr, *act_bb = nullptr, *h_bb = nullptr;
  float* head_out = nullptr;  // [B][ld_head]: [0,Hd) decoder pos-0 input, [Hd, Hd+V) codebook-0 logits
  int ld_head = 0;
  float *dec_x = nullptr, *q_dec = nullptr, *att_dec = nullptr, *act_dec = nullptr, *logits_dec = nullptr;
  float* last_h = nullptr;
  // two-token first decoder pass (B == 1; reference modeling_csm.py:534-552 runs positions 0 and 1 as ONE forward)
  float *dec_x2 = nullptr, *q_dec2 = nullptr, *att_dec2 = nullptr, *act_dec2 = nullptr;
  int *d_pos01 = nullptr, *d_seq00 = nullptr;
  int two_token_pass = 1;   // B == 1: 3.169 -> 3.114-3.129 ms per step on the 2-row register kernel (profiles/r03_b1_ab.txt)
  int64_t* ids_stage = nullptr;
  uint8_t* mask_stage = nullptr;
  // prefill scratch
  float *p_h = nullptr, *p_xn = nullptr, *p_qkv = nullptr, *p_q = nullptr, *p_att = nullptr, *p_act = nullptr;
  int *p_row_seq = nullptr, *p_row_pos = nullptr;
  // prefill activations as row-major bf16 planes [3][rows][K] (bf16 / fp8 weights): split once by the producer
  bf16_t *p_pl_h = nullptr, *p_pl_act = nullptr;
  // training forward (csm_forward_loss), allocated at the first call: head rows of every context position, scratch KV
  // caches of the decoder pass over labelled frames (LOSS_FRAMES sequences of 32 positions), logits / labels / losses
  float* loss_head = nullptr; size_t loss_head_rows = 0;
  std::vector<void*> loss_kc, loss_vc;
  float* loss_logits = nullptr; float* loss_rows = nullptr; int* loss_lab = nullptr; int* loss_idx = nullptr; double* loss_acc = nullptr;
  size_t loss_lab_n = 0, loss_rows_n = 0;
  float* p_part = nullptr;   // split-K partial products of the residual prefill GEMMs: [4][max_prefill_rows][Hb]
  int prefill_splitk = 1;
  static constexpr int prefill_splitk_qkv = 4;   // most K splits of the QKV GEMM (swept 0 / 2 / 4 / 8: 4 best or tied at 32-512 frames) of a short prefill split over K too (partials summed by the RoPE launch)
  float* p_part_gu = nullptr;   // [4][min(128, max_prefill_rows)][2 F] partial products of a short prefill's split-K gate/up GEMM (allocated at first use)
  int prefill_splitk_gu = 2;    // most K splits of the gate/up GEMM of a prefill of <= 64 rows (<= 128 with one activation plane); partials summed + SwiGLU by swiglu_reduce_kernel; 0 / 1 = off.  Measured 2 / 4 ways at 32 / 64 / 128 rows: bf16 1.49 -> 1.37 / 1.38, 1.54 -> 1.42 / 1.47, 1.83 -> 1.76 / 1.86 ms; exact 1.87 -> 1.72 / 1.74, 1.98 -> 1.84 / 1.89, 2.39 -> 2.50 / 2.59
  static constexpr int prefill_plane_pad = 2176; // elements between the bf16 planes of an exact-mode operand beyond R K (multiple of 8, <= 8192).  Measured over 4 processes each at 2 048 rows: pad 0 16.1 / 18.2 / 16.1 / 18.3 ms (bimodal by process), 2176: 16.2 / 16.2 / 16.7 / 16.9; 1088: 17.5; 4224: 16.3 / 17.4; no effect at 512 / 1 024 / 16 x 512 rows
  int gemm_mx_skinny = 256;     // GemmMxArgs::skinny: the same for the MX-fp8 GEMM, as a row bound (0 = off)
  int gemm_dma_skinny = 1;      // GemmArgs::dma_skinny: 64 / 32-row workgroups of the LDS-DMA GEMM for the split-K / SwiGLU launches of a prefill of up to 256 rows (bf16 mode) / 768 rows (exact mode)
  int prefill_fuse_rope = 1;    // QKV GEMM with the RoPE / q-scale / cache-append epilogue (GEPI_ROPE, gemm.h) where an LDS-DMA tile takes the launch and head_dim is 64
  int prefill_fuse_quant = 1;   // mxfp8 mode: the context attention writes its output already MX-quantised (no mx_quant_rows launch)
  size_t p_part_h = 0;          // p_part holds 4 x max_prefill_rows x p_part_h floats
  static constexpr int prefill_splitk_max = 8;   // most K splits of a residual prefill GEMM (the partials buffer holds 4 at max_prefill_rows: more only for fewer rows)
  static constexpr int prefill_planes = 1;
  int prefill_bf16 = 0;   // prefill_precision: 0 = exact (fp32 activations as three bf16 planes), 1 = activations rounded to bf16 (one plane)
  int gemm_wide = 1; static constexpr int gemm_wide_depth = 1, gemm_wide_exact = 0;   // gemm_wide_kernel switches (GemmArgs::wide ...)
  int gemm_256 = 256;   // gemm256_kernel (256 x 256 tile): minimum workgroup count (tiles x K splits) of a launch it takes (one per CU); 0 = off;
                        // bits 24-25 select a schedule variant (A/B).  csm-1b prefill, bf16 / mxfp8: 2048 frames 6.50 / 5.07 -> 5.86 / 4.71 ms,
                        // 16 x 512 frames 19.7 / 16.0 -> 18.8 / 13.6 ms (profiles/r03_gemm256.txt)
  static constexpr int gemm_dma_min_wgs = 200;   // exact (three-plane) LDS-DMA GEMM only for launches of at least this many 128 x 128 workgroups: below, the 64 x 64
                                // square tile fills the chip better (256-frame exact prefill 3.26 -> 2.92 ms; 512 / 1024 frames unchanged: profiles/r03_prefill.txt)
  int gemm_dma = 5; static constexpr int gemm_dma_max_rows = 4096;   // gemm_dma_bf16_kernel (GemmArgs::dma): bit 0 one plane, bit 2 three planes (exact), launches of up to 4096 rows
  // prefill_precision = mxfp8 (BASELINE configs[4]): MX-fp8 copies of the backbone linears (csm_bind_mx_weights, borrowed) and
  // the quantised-activation scratch [max_prefill_rows][widest K] + scales
  std::unordered_map<const void*, void*> train_wT;   // transposed weight copies of the training backward (train_impl.inc)
  std::vector<csm_mx_layer_t> mx_layers;
  int prefill_mx = 0;
  uint8_t *p_mx_q = nullptr, *p_mx_s = nullptr, *p_mx_q2 = nullptr, *p_mx_s2 = nullptr;   // q2 / s2: the SwiGLU output (down_proj's operand)
  int mx_fuse_swiglu = 1;   // gate/up writes its SwiGLU output as MX-fp8 itself (0: fp32 + quantiser launch, A/B)
  static constexpr int prefill_x3_attn = 1;     // exact mode: context attention as three-piece products on the bf16 matrix pipe (0 = fp32 MFMA)
  int prefill_bf16_attn = 1;   // with prefill_bf16: the context attention on the bf16 matrix pipe too (0 = keep the fp32-MFMA flash kernel)
  // host mirrors
  int B = 0;
  int h_len = 0, h_frame = 0;
  bool ready = false;   // head_out holds valid c0 logits
  int nsplit_bb = 0;  // 0 = auto: ~256 workgroups per attention launch
  int fuse_attn_oproj = 1;   // decoder attention + o_proj as one launch (attn_oproj.h): bit 0 single sequence (B = 1: 3.28 -> 3.16 ms),
                             // bit 1 batched rows (measured SLOWER at B = 16, 5.58 vs 5.09 ms: off)
  static constexpr int use_mfma = 1;
  static constexpr int flash_prefill = 1;
  int kernel_prio = 7;         // s_setprio 3 at kernel entry (issue priority over the resident weight-streamer waves): bit 0 the fused decoder
                               // attention + o_proj launch, bit 1 the GEMV family, bit 2 backbone attention and the samplers
  int dbg_sample_spin = 0;     // TIMING ONLY: every sampler launch idles this many 10 ns ticks first
  int prefill_attn_kvfast = 1;   // bf16-class context attention: kv-head as the fastest grid index (one XCD per kv-head)
  int g16_xcdmap = 1;  // gemm128 gate/up panels on the XCD that reads their h columns in the down_proj launch (needs g16_kfast)
  int g16_kfast = 1;   // K-split matrix-core launches: the k split as the fastest grid index
  int g128 = 1, g128_min = 64, g128_shape = 0;   // gemm128.h for the FFN launches of batches beyond g128_min rows (shape: A/B override)
  int attn_gqa_wide = 1;     // backbone attention of > 32 rows on attn_decode_gqa_kernel
  int oproj_combine = 1;     // B = 1 backbone: split-KV merge folded into the o_proj launch (gemv1_combine_kernel), attention on bb_nsplit_b1 long splits
  int cmb_splits = 8;        // its split count (<= 8)
  int attn_oproj_gqa = 1;    // the fused launch in its key-split form (attn_oproj_gqa_kernel: K/V tiles shared by the query heads of a kv-head)
  int fuse_sample = 1;   // B == 1 greedy: argmax folded into the head launch + next QKV prologue
  float2* am_part = nullptr;
  float* g16_slabs = nullptr;
  size_t g16_slab_floats = 0;
  int* g16_tickets = nullptr;
  static constexpr int nt_backbone = 1, nt_decoder = 2;   // decoder: the large streams (gate/up, down) non-temporal -- beside the weight streamer their
                                           // consumed lines are then the first victims in L2 (3.32 -> 3.30 ms per step)
  // KV splits of the backbone decode attention: the kernel is latency-bound per 32-key tile, so aim for
  // <= 2 tiles per workgroup at the current length (+ headroom for the frames of this generate call) while
  // keeping at least ~256 workgroups; frozen into the graph at capture time.
  int nsplit_eff() const {
    if (nsplit_bb > 0) return nsplit_bb;
    if (B == 1 && oproj_combine && cfg.backbone.head_dim == 64 && cfg.backbone.n_q == 4 * cfg.backbone.n_kv && cfg.backbone.n_q * 64 == 2048) return cmb_splits;
    int by_len = (h_len + 256 + 63) / 64;
    if (B >= 32) by_len = (by_len + 7) / 8;      // 32-64 rows: 256-512 (row, kv-head) pairs already fill the chip -- 2 splits at a 512-frame
                                                 // context: B = 64 frame-step 9.48 -> 9.25 ms (8 splits -> 2; profiles/r03_b64_rows64.txt)
    else if (B >= 8) by_len = std::min(4, (by_len + 3) / 4);  // enough rows to fill the chip: fewer, longer splits (less combine work); B = 16:
                                                 // 4 splits 5.04 ms against 5.08 (8) / 5.09 (2) at 512 frames, 5.40 / 5.44 / 5.47 at 2048
    const int by_fill = 256 / ((B > 0 ? B : 1) * cfg.backbone.n_kv);
    const int ns = by_len > by_fill ? by_len : by_fill;
    int p = 1;
    while (p < ns && p < 64) p *= 2;
    return p;
  }
  std::map<GraphKey, GraphEntry> graphs;
  uint64_t graph_tick = 0;
  int graphs_captured = 0;
  // weight streamer (prefetch.h): a persistent kernel on `stream2` pulls the weights of the replaying graph into L2
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  unsigned* d_prog = nullptr;      // launches started (bumped by the streamed launches)
  unsigned* d_pf_misc = nullptr;   // [0..7] per-XCD tickets, [8..11] status, [12..15] stop record, [16..23] XCD probe, [32..33] concurrency probe,
                                   // [40..42] lifetime counters (never reset): give-ups, finished, streamer launches
  // streamer health (round 6): the status words of every streamer launch are copied to pinned host memory behind the kernel on its own
  // stream; the NEXT csm_generate reads them once `ev_join` has completed (no extra synchronisation) and, after a give-up, re-runs the
  // stream-concurrency probe and switches the streamer off for this engine rather than stalling every call (pf_harvest)
  unsigned* h_pf = nullptr;        // pinned mirror of d_pf_misc[8..47]
  int pf_pending = 0;              // streamer launches whose mirror has not been read yet
  unsigned pf_seen_gaveup = 0;     // lifetime give-ups already accounted for
  int pf_strikes = 0, pf_clean = 0, pf_probe_runs = 0;
  int pf_disabled = 0;             // 0 on;  1 the two streams share a hardware queue (probe);  2 repeated give-ups with the probe passing;
                                   // 3 dispatch not round-robin over the XCDs;  4 the probe at engine creation failed
  float pf_rate_base_us = 0.f, pf_rate_beside_us = 0.f;   // dispatch-rate probe: us per empty launch alone / beside a resident kernel on the streamer's stream
  int pf_budget_us = 20000;        // no launch starting for this long while launches are outstanding = a stalled chain
  int sample_legacy = 0;           // test hook: csm_sample_topk's top-k on sample_kernel's histogram / radix selection of rounds 1-4
  int pf_force_serial = 0;         // test hook: streamer (and probe) on the ENGINE stream -- the failure mode of two streams on one hardware queue
  int pf_rot = -1;                 // workgroup b of a dispatch runs on XCD (b + pf_rot) % 8; -1 = not round-robin: streamer off
  int pf_enable = 1, pf_window_mb = 6; int pf_sub_kb = 8192; static constexpr int pf_grid = 256;   // sub: round 6 -- runs of 8 MiB (rounds 2-5: 4): B = 1 2.82 -> 2.72 ms; plateau 6-10 MiB, window 4-12 MiB alike (profiles/r06_streamer_grid.txt)
  //   // window: round 5 -- 24 MiB (rounds 2-4) only works while the chain
  // never lets the streamer get a full window ahead: after any launch longer than ~3 us (a sampler) the data fetched first is gone again by the time it is read
  // (B = 1 top-k 50: 3.53 ms at 24 MiB, 3.36 at 6; greedy 3.096 -> 3.067); profiles/r05_streamer_window.txt
  int pf_lead = 1;   // 1: the data of the RUNNING launch counts as consumed (all its workgroups issue their loads at once)
  int g16_k16 = 0;      // (option "g16_k16", A/B) nw | kb << 8 for the K = 2048 (16-chunk) matrix-core launches on planes; 0 = one 16-wave workgroup per panel
  int pf_batched = 0;   // (option "prefetch_batched", A/B) 1: also pace / stream the matrix-core launches of batched decode (they are `exclusive` for csm-1b: gemm16.h)
  int pf_cofetch = 1, pf_skip_late = 1, pf_stride = 0;
  int pf_depth = 0, pf_poll_sleep = 2;   // (options again in round 6: re-swept after the kernel-argument preload shortened every launch)
  int pf_seg_sleep = 0;   // seg_sleep: 16 until the decode kernels got issue priority (kernel_prio); with it an unthrottled
  // streamer no longer slows the chain's latency-bound launches: B = 1 3.09 -> 2.95 ms (profiles/r05_b1_budget.md)   // options again in round 5 (prefetch_depth / prefetch_seg_sleep): re-swept at the 6 MiB window
  static constexpr int pf_max_kb = 0;        // > 0: only launches whose matrix is at most this large are streamed whole
  static constexpr int pf_part_kb = 0;       // > 0: of larger matrices, stream only the first this-many KiB (round 5 sweep: monotone worse)
  std::vector<PfGeom>* pf_rec = nullptr;   // non-null while a frame-step is being captured
  std::vector<PfGeom> last_geoms;          // launches of the last captured frame-step (debug / tools)
  uint32_t* dbg_buf = nullptr;             // debug probe: [launch][2048 workgroups][2] (csm_set_debug_buffer)
  int dbg_launches = 0;
  int tl_n = 0;                            // timeline probe: slots handed out during the current capture
  long long pf_last[4] = {0, 0, 0, 0};
  int pf_last_frames = 0;   // schedule of the last replayed graph: segments, launches, scheduled bytes, streamed-launch bytes
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_ms = 0.f;
  std::vector<void*> allocs;
  // fragment-order copies of the linear weights for the MFMA skinny GEMM (batched decode), keyed by the bound pointer
  std::unordered_map<const void*, void*> tiled;
  std::vector<void*> tiled_allocs;
  int tile_weights = 1;
  // activations handed between the batched-decode launches as ready-made MFMA B operands (gemv.h: xplanes)
  bf16_t *pl_h = nullptr, *pl_act = nullptr;
  float* pl_ss = nullptr;
  int use_planes = 31;  // bit 0: residual stream, bit 1: SwiGLU output, bit 2: attention output, bit 3: sampler feedback row, bit 4: backbone input row (embedding sum)
  static constexpr int g16_gu = 0;     // A/B: panel tiles of the batched gate/up launch (0 = auto, 1 | 2 | 4)
  static constexpr int attn_prefetch = 0;   // backbone decode attention requests tile i+1 before consuming tile i: bit 0 at B = 1, bit 1 at B >= 2
  static constexpr int gemv_norm_ks = 1;   // B = 1, K = 2048 normed launches on the register GEMV with two waves per task: bit 0 the backbone QKV (3.134 -> 3.103 ms
                          // per step, tokens unchanged), bit 1 gate/up (measured slower: off); profiles/r03_b1_ab.txt
  int* p_seq_slot = nullptr;   // set around stack_rows by csm_prefill_slots: sequence b of the prefill lives in cache slot p_seq_slot[b]
  int* d_slots = nullptr;      // [max_batch] device copy of the slots of one csm_prefill_slots call
  int rows64 = 1;   // batches of 33..64 rows: one matrix-core launch per linear (gemm32_kernel with four batch tiles) instead of two 32-row launches
  char *shift_kt = nullptr, *shift_vt = nullptr;   // scratch of csm_shift_context (one layer of the resident batch), kept between calls
  size_t shift_bytes = 0;
  int decode_bf16 = 0;   // decode_precision = bf16 (the reference's own arithmetic class for batched decode, README.md:73): activations handed
                         // between the matrix-core launches as ONE nearest-even bf16 plane, one MFMA per weight fragment instead of three
  int dbg_skip = 0;   // TIMING ONLY (results are wrong): knock launches out of a decode layer -- bits 0-4 decoder QKV / attention / o_proj / gate-up / down_proj, bit 5 the fused B = 1 attention + o_proj launch, bit 6 the B = 1 fused-argmax heads, bits 8-12 the same five for the backbone
  int g16_slab = 0;   // bits 4-7: TIMING-ONLY knock-outs of the -DCSM_G16_KO variant build (dbg_skip bits 16-19)
  static constexpr int g16_down = 0;   // A/B: panel shape override of the batched down_proj (nw | kb << 8 | pt << 16), 0 = auto
  static constexpr int attn_one_wave = 1;  // bit 0: decoder attention, bit 1: backbone attention as one-wave workgroups (measured: B=1
                          // 3.54 / 3.49 / 3.56 / 3.51 ms per step for 0 / 1 / 2 / 3)
};

static void drop_tiled(csm_engine* e);
constexpr int PL_GROUPS = 8;     // plane groups of 16 rows: batches up to 128 rows run on activation planes
constexpr int PL_SS_LD = 512;   // partial-sum columns per row: hidden / 16 tiles, hidden <= 8192
static inline int emb_dtype(const csm_engine* e) { return e->cfg.weight_dtype == CSM_DTYPE_FP8 ? CSM_DTYPE_BF16 : e->cfg.weight_dtype; }
static inline size_t w_esz(const csm_engine* e) { return e->cfg.weight_dtype == CSM_DTYPE_FP8 ? 1 : (e->cfg.weight_dtype == CSM_DTYPE_BF16 ? 2 : 4); }

template <typename T>
static int dalloc(csm_engine* e, T** p, size_t n) {
  void* q = nullptr;
  hipError_t r = hipMalloc(&q, n * sizeof(T) + 256);
  if (r != hipSuccess) return fail(CSM_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(r));
  e->allocs.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return 0;
}

extern "C" const char* csm_last_error(void) { return g_err; }
extern "C" int csm_abi_version(void) { return CSM_ABI_VERSION; }

static int check_stack(const csm_llama_cfg_t& c, const char* name) {
  if (c.hidden % 8 || c.ffn % 8 || c.layers < 1) return fail(CSM_ERR_ARG, "%s: hidden/ffn must be multiples of 8", name);
  if (c.head_dim != 64 && c.head_dim != 128) return fail(CSM_ERR_ARG, "%s: head_dim must be 64 or 128, got %d", name, c.head_dim);
  if (c.n_q % c.n_kv || c.n_q / c.n_kv > 16) return fail(CSM_ERR_ARG, "%s: unsupported GQA ratio", name);
  return 0;
}


// ---- stream pool (round 6) ----
// The streams of a destroyed engine are kept and handed to the next engine on the same device instead of being destroyed and created
// anew: which hardware queue slot a HIP stream gets depends on everything the process created before, and the frame-step time depends on
// the slots of the engine stream and of the streamer's stream (profiles/r06_queue_layout_states.md: an engine that replaced a smaller one
// -- a short generate() first, then a longer one -- ran at 2.87 ms per step against 2.75 for the first engine of the process).  The first
// engine of a process gets the good layout; its successors inherit it.  Engines alive at the same time get pairs of their own.
struct StreamPair { int device; hipStream_t own, s2; };
static std::mutex g_stream_pool_mu;
static std::vector<StreamPair> g_stream_pool;

// ---- weight-streamer health (prefetch.h) ----
// Stream-concurrency probe: a waiter on the streamer's stream must see a flag raised by a kernel submitted AFTER it on the engine
// stream.  If the two HIP streams share a hardware queue the waiter times out (3 ms) and the streamer, which is submitted ahead of the
// replays it feeds, would hold them up for one budget per call.  Synchronises both streams.
static int pf_probe_concurrency(csm_engine* e, int* seen_out) {
  unsigned seen = 0;
  hipStream_t waiter = e->pf_force_serial ? e->stream : e->stream2;
  HIPCK(hipStreamSynchronize(e->stream2));
  HIPCK(hipMemsetAsync(e->d_pf_misc + 32, 0, 2 * sizeof(unsigned), e->stream));
  HIPCK(hipStreamSynchronize(e->stream));
  LCK(launch_pf_concurrency_probe(waiter, e->stream, e->d_pf_misc + 32, e->d_pf_misc + 33));
  HIPCK(hipStreamSynchronize(e->stream2));
  HIPCK(hipStreamSynchronize(e->stream));
  HIPCK(hipMemcpy(&seen, e->d_pf_misc + 33, sizeof(seen), hipMemcpyDeviceToHost));
  e->pf_probe_runs++;
  *seen_out = (int)seen;
  if (seen && !e->pf_force_serial) {
    // Concurrent is not enough (round 6): with the two streams on ONE hardware queue slot (a process with several other active streams:
    // the runtime multiplexes HIP streams over a few queues) a kernel resident on the streamer's stream let the chain run -- at ~30 us per
    // launch instead of ~3: an 18 ms frame-step with `gave_up 0`.  So the probe also measures the DISPATCH RATE of the engine stream:
    // a hipGraph of 128 dependent empty launches alone, then beside a resident spinning kernel (one workgroup per CU) on the streamer's stream.
    float base = 1e30f, beside = 1e30f;   // the best of three each: a host hiccup between two launches must not look like a slow queue
    for (int rep = 0; rep < 3; ++rep)
      for (int pass = 0; pass < 2; ++pass) {
        HIPCK(hipMemsetAsync(e->d_pf_misc + 32, 0, 2 * sizeof(unsigned), e->stream));
        HIPCK(hipStreamSynchronize(e->stream));
        LCK(launch_pf_rate_probe(pass ? e->stream2 : nullptr, e->stream, e->d_pf_misc + 32, e->d_pf_misc + 33, 128, e->ev0, e->ev1));
        HIPCK(hipStreamSynchronize(e->stream));
        HIPCK(hipStreamSynchronize(e->stream2));
        float ms = 0.f;
        HIPCK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
        if (pass) beside = std::min(beside, ms); else base = std::min(base, ms);
      }
    e->pf_rate_base_us = base * 1e3f / 128.f;
    e->pf_rate_beside_us = beside * 1e3f / 128.f;
    // measured: 1.70 us alone, 1.80 beside (+6 %) on separate queues; > 2 x on a shared one (the real chain: 2.7 -> 18-21 ms per step).
    // (A third state exists that this probe does not see: section 8 of DESIGN.md, the null-stream order.)
    if (beside > 1.6f * base) *seen_out = 2;   // concurrent, but the chain's launches crawl beside the resident kernel
  }
  return 0;
}

// Reads the pinned status mirror of the streamer launches that have completed (`wait`: block until the last one has) and acts on
// give-ups: the probe is run again; if the streams turn out to be serialised the streamer is off for this engine (reason 1), and so it
// is after two give-ups within 64 calls with the probe passing (reason 2: a chain that stalls beside the resident streamer for another
// reason).  A single give-up with a passing probe (a host that was descheduled between two graph launches) costs nothing further.
static int pf_harvest(csm_engine* e, bool wait) {
  if (!e->pf_pending || !e->h_pf) return 0;
  if (wait) HIPCK(hipEventSynchronize(e->ev_join));
  else if (hipEventQuery(e->ev_join) != hipSuccess) { (void)hipGetLastError(); return 0; }
  e->pf_pending = 0;
  const unsigned gave = e->h_pf[32];   // d_pf_misc[40]
  if (gave == e->pf_seen_gaveup) {
    if (++e->pf_clean >= 64) { e->pf_clean = 0; if (e->pf_strikes > 0) e->pf_strikes--; }
    return 0;
  }
  e->pf_seen_gaveup = gave;
  e->pf_strikes++;
  e->pf_clean = 0;
  int seen = 0;
  if (int r = pf_probe_concurrency(e, &seen)) return r;
  if (!seen) e->pf_disabled = 1;
  else if (seen == 2) e->pf_disabled = 5;
  else if (e->pf_strikes >= 2) e->pf_disabled = 2;
  return 0;
}
static const char* pf_reason(int r) {
  switch (r) {
    case 0: return "on";
    case 1: return "off: the streamer's stream and the engine stream share a hardware queue (give-up, then probe)";
    case 2: return "off: two give-ups within 64 calls although the streams run concurrently";
    case 3: return "off: workgroups are not dispatched round-robin over the XCDs";
    case 4: return "off: the stream-concurrency probe failed at engine creation";
    case 5: return "off: launches on the engine stream slow down > 1.6 x beside a kernel resident on the streamer's stream (shared hardware queue)";
  }
  return "?";
}

extern "C" int csm_engine_create(const csm_config_t* cfg, int device, void* stream, csm_engine_t** out) {
  if (!cfg || !out) return fail(CSM_ERR_ARG, "null argument");
  if (cfg->abi_version != CSM_ABI_VERSION) return fail(CSM_ERR_ARG, "ABI version mismatch: %d vs %d", cfg->abi_version, CSM_ABI_VERSION);
  if (int r = check_stack(cfg->backbone, "backbone")) return r;
  if (int r = check_stack(cfg->decoder, "decoder")) return r;
  if (cfg->weight_dtype < 0 || cfg->weight_dtype > 2 || cfg->kv_dtype < 0 || cfg->kv_dtype > 1) return fail(CSM_ERR_ARG, "bad dtype");
  if (cfg->max_batch < 1 || cfg->max_len < 1 || cfg->max_frames < 1 || cfg->max_prefill_rows < 1)
    return fail(CSM_ERR_ARG, "max_batch/max_len/max_frames/max_prefill_rows must be >= 1");
  if (cfg->n_codebooks < 2) return fail(CSM_ERR_ARG, "n_codebooks must be >= 2");
  HIPCK(hipSetDevice(device));
  LCK(gemv_configure_all());
  LCK(gemm32_configure_all());
  LCK(gemm128_configure_all());
  LCK(configure_sample());
  csm_engine* e = new csm_engine();
  e->cfg = *cfg;
  e->device = device;
  int prio_least = 0, prio_greatest = 0;
  HIPCK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  {
    // The legacy null stream must own its hardware queue BEFORE the weight streamer's stream is created (round 6, measured:
    // tools/probes/stream_alloc_probe2.py).  In a process whose first GPU work runs on a side stream (`with torch.cuda.stream(s):` around
    // everything) the null stream gets its queue only later -- at the first synchronous hipMemcpy of this file -- and the B = 1 frame-step
    // then runs at 3.16 ms with the streamer on against 3.03 off and 2.73 when the null stream came first; one null-stream operation
    // here, before the streams below exist, gives 2.72-2.73 in every order.  (Why the order of queue creation matters is not established.)
    void* z = nullptr;
    HIPCK(hipMalloc(&z, 256));
    HIPCK(hipMemsetAsync(z, 0, 256, nullptr));
    HIPCK(hipStreamSynchronize(nullptr));
    HIPCK(hipFree(z));
  }
  {
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    for (size_t i = 0; i < g_stream_pool.size(); ++i)
      if (g_stream_pool[i].device == device) { e->pooled_own = g_stream_pool[i].own; e->stream2 = g_stream_pool[i].s2; g_stream_pool.erase(g_stream_pool.begin() + i); break; }
  }
  if (!e->pooled_own) HIPCK(hipStreamCreateWithPriority(&e->pooled_own, hipStreamNonBlocking, prio_greatest));   // (created even when the caller brings a stream: the pair stays together)
  if (stream) {
    e->stream = reinterpret_cast<hipStream_t>(stream);
  } else {
    e->stream = e->pooled_own;
    e->own_stream = true;
  }
  HIPCK(hipEventCreate(&e->ev0));
  HIPCK(hipEventCreate(&e->ev1));
  e->bb.c = cfg->backbone;
  e->dec.c = cfg->decoder;
  e->esz_kv = cfg->kv_dtype == CSM_DTYPE_BF16 ? 2 : 4;
  const int B = cfg->max_batch, C = cfg->n_codebooks, V = cfg->audio_vocab;
  const int Hb = cfg->backbone.hidden, Hd = cfg->decoder.hidden;
  int r;
  // KV caches: +64 positions of slack so 16-byte tail reads of the last tile stay in bounds
  e->bb.lmax = cfg->max_len;
  e->dec.lmax = C;
  for (Stack* s : {&e->bb, &e->dec}) {
    const size_t per = (size_t)B * s->c.n_kv * s->lmax * s->c.head_dim * e->esz_kv;
    for (int l = 0; l < s->c.layers; ++l) {
      char *k = nullptr, *v = nullptr;
      if ((r = dalloc(e, &k, per))) return r;
      if ((r = dalloc(e, &v, per))) return r;
      HIPCK(hipMemsetAsync(k, 0, per, e->stream));
      HIPCK(hipMemsetAsync(v, 0, per, e->stream));
      s->kc.push_back(k);
      s->vc.push_back(v);
    }
  }
  if ((r = dalloc(e, &e->d_len, 1)) || (r = dalloc(e, &e->d_frame, 1)) || (r = dalloc(e, &e->d_kv_start, (size_t)B)) ||
      (r = dalloc(e, &e->d_rng, 2)))
    return r;
  HIPCK(hipMemsetAsync(e->d_rng, 0, 2 * sizeof(uint64_t), e->stream));
  if ((r = dalloc(e, &e->d_zero_count, (size_t)cfg->max_frames)) || (r = dalloc(e, &e->d_row_done, (size_t)B))) return r;
  HIPCK(hipMemsetAsync(e->d_zero_count, 0, (size_t)cfg->max_frames * sizeof(int), e->stream));
  HIPCK(hipMemsetAsync(e->d_row_done, 0, (size_t)B * sizeof(int), e->stream));
  if ((r = dalloc(e, &e->ring, (size_t)B * cfg->max_frames * C))) return r;
  HIPCK(hipMemsetAsync(e->ring, 0, (size_t)B * cfg->max_frames * C * sizeof(int64_t), e->stream));
  HIPCK(hipMemsetAsync(e->d_kv_start, 0, B * sizeof(int), e->stream));
  e->ld_head = (Hd + V + 3) & ~3;
  const int nqb = cfg->backbone.n_q * cfg->backbone.head_dim, nqd = cfg->decoder.n_q * cfg->decoder.head_dim;
  if ((r = dalloc(e, &e->h_bb, (size_t)B * Hb)) || (r = dalloc(e, &e->q_bb, (size_t)B * nqb)) ||
      (r = dalloc(e, &e->att_bb, (size_t)B * nqb)) ||
      (r = dalloc(e, &e->part_bb, (size_t)B * cfg->backbone.n_q * 64 * (cfg->backbone.head_dim + 4))) ||
      (r = dalloc(e, &e->attn_tickets, (size_t)B * cfg->backbone.n_q)) ||
      (r = dalloc(e, &e->act_bb, (size_t)B * cfg->backbone.ffn)) || (r = dalloc(e, &e->head_out, (size_t)B * e->ld_head)) ||
      (r = dalloc(e, &e->dec_x, (size_t)B * Hd)) || (r = dalloc(e, &e->q_dec, (size_t)B * nqd)) ||
      (r = dalloc(e, &e->att_dec, (size_t)B * nqd)) || (r = dalloc(e, &e->act_dec, (size_t)B * cfg->decoder.ffn)) ||
      (r = dalloc(e, &e->logits_dec, (size_t)B * ((V + 3) & ~3))) || (r = dalloc(e, &e->last_h, (size_t)B * Hb)) ||
      (r = dalloc(e, &e->ids_stage, (size_t)B * (C + 1))) || (r = dalloc(e, &e->mask_stage, (size_t)B * (C + 1))))
    return r;
  if ((r = dalloc(e, &e->dec_x2, (size_t)2 * Hd)) || (r = dalloc(e, &e->q_dec2, (size_t)2 * nqd)) || (r = dalloc(e, &e->att_dec2, (size_t)2 * nqd)) ||
      (r = dalloc(e, &e->act_dec2, (size_t)2 * cfg->decoder.ffn)) || (r = dalloc(e, &e->d_pos01, (size_t)4)))
    return r;
  e->d_seq00 = e->d_pos01 + 2;
  {
    const int h01[4] = {0, 1, 0, 0};
    HIPCK(hipMemcpyAsync(e->d_pos01, h01, sizeof(h01), hipMemcpyHostToDevice, e->stream));
    HIPCK(hipStreamSynchronize(e->stream));
  }
  const size_t R = cfg->max_prefill_rows;
  // the prefill scratch serves BOTH stacks (stack_rows: backbone context; decoder pass of the training forward), so every
  // buffer takes the wider of the two shapes (the tiny test model's decoder QKV, 512 wide, is wider than its backbone's 384)
  const size_t Hm_ = std::max(Hb, Hd), qkvm = std::max(e->bb.nqkv(), e->dec.nqkv()), nqm = std::max(nqb, nqd);
  const size_t ffm = std::max(cfg->backbone.ffn, cfg->decoder.ffn);
  if ((r = dalloc(e, &e->p_h, R * Hm_)) || (r = dalloc(e, &e->p_xn, R * Hm_)) ||
      (r = dalloc(e, &e->p_qkv, R * qkvm)) || (r = dalloc(e, &e->p_q, R * nqm)) ||
      (r = dalloc(e, &e->p_att, R * nqm)) || (r = dalloc(e, &e->p_act, R * ffm)) ||
      (r = dalloc(e, &e->p_row_seq, R)) || (r = dalloc(e, &e->p_row_pos, R)))
    return r;
  if (cfg->weight_dtype != CSM_DTYPE_F32) {
    const size_t Km = std::max(Hm_, nqm);
    if ((r = dalloc(e, &e->p_pl_h, 3 * R * Km + 3 * 8192)) || (r = dalloc(e, &e->p_pl_act, 3 * R * ffm + 3 * 8192))) return r;   // + room for the plane pad (prefill_plane_pad <= 8192 elements)
    if (R <= 4096 && (r = dalloc(e, &e->p_part, 4 * R * Hm_))) return r;   // only small prefills are short of workgroups
    e->p_part_h = Hm_;
  }
  if ((r = dalloc(e, &e->am_part, (size_t)2048))) return r;
  // split-K scratch of the MFMA skinny GEMM: panels x K-splits x 64x16 floats (4 MiB covers N = 4096, K = 8192)
  {
    const size_t Hm = std::max(cfg->backbone.hidden, cfg->decoder.hidden), Fm = std::max(cfg->backbone.ffn, cfg->decoder.ffn);
    const size_t G = PL_GROUPS;
    if ((r = dalloc(e, &e->pl_h, G * 3 * 16 * Hm)) || (r = dalloc(e, &e->pl_act, G * 3 * 16 * Fm)) || (r = dalloc(e, &e->pl_ss, G * 16 * PL_SS_LD))) return r;
    HIPCK(hipMemsetAsync(e->pl_h, 0, G * 3 * 16 * Hm * sizeof(bf16_t), e->stream));
    HIPCK(hipMemsetAsync(e->pl_act, 0, G * 3 * 16 * Fm * sizeof(bf16_t), e->stream));
    HIPCK(hipMemsetAsync(e->pl_ss, 0, G * 16 * PL_SS_LD * sizeof(float), e->stream));
  }
  e->g16_slab_floats = (size_t)1 << 23;   // split-K slabs: panels x K splits x batch tiles x 256 floats (backbone gate/up at 128 rows: 4 M floats)
  if ((r = dalloc(e, &e->g16_slabs, e->g16_slab_floats)) || (r = dalloc(e, &e->g16_tickets, (size_t)4096))) return r;
  HIPCK(hipMemsetAsync(e->g16_tickets, 0, 4096 * sizeof(int), e->stream));
  HIPCK(hipMemsetAsync(e->attn_tickets, 0, (size_t)B * cfg->backbone.n_q * sizeof(int), e->stream));

  LCK(launch_set_int(e->stream, e->d_len, 0));
  LCK(launch_set_int(e->stream, e->d_frame, 0));
  // weight streamer: second stream, fork/join events, launch counter, and the dispatcher's workgroup -> XCD rotation
  if (!e->stream2) HIPCK(hipStreamCreateWithPriority(&e->stream2, hipStreamNonBlocking, prio_least));
  HIPCK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  HIPCK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
  if ((r = dalloc(e, &e->d_prog, 16)) || (r = dalloc(e, &e->d_pf_misc, 64))) return r;
  HIPCK(hipMemsetAsync(e->d_prog, 0, 16 * sizeof(unsigned), e->stream));
  HIPCK(hipMemsetAsync(e->d_pf_misc, 0, 64 * sizeof(unsigned), e->stream));
  {
    unsigned where[8];
    LCK(launch_pf_where(e->stream, e->d_pf_misc + 16));
    HIPCK(hipMemcpyAsync(where, e->d_pf_misc + 16, sizeof(where), hipMemcpyDeviceToHost, e->stream));
    HIPCK(hipStreamSynchronize(e->stream));
    bool rr = true;
    for (int b = 0; b < 8; ++b) rr = rr && where[b] == ((where[0] + b) & 7u);
    e->pf_rot = rr ? (int)where[0] : -1;
    if (!rr) e->pf_disabled = 3;
    // the streamer must run BESIDE the engine stream: a waiter on stream2 has to see a flag raised by a kernel
    // submitted later on the engine stream.  If the two streams share a hardware queue it times out (3 ms): streamer off.
    // (Creating another stream for the streamer when the probes fail -- the next one lands on the next hardware queue -- was tried: the
    // probes then pass, but the frame-step runs at 3.15 ms with the streamer on against 3.03 off; so a failed probe switches it off.)
    int seen = 0;
    if (int pr = pf_probe_concurrency(e, &seen)) return pr;
    if (!seen && e->pf_rot >= 0) { e->pf_rot = -1; e->pf_disabled = 4; }
    if (seen == 2 && e->pf_rot >= 0) e->pf_disabled = 5;
    HIPCK(hipHostMalloc((void**)&e->h_pf, 40 * sizeof(unsigned), hipHostMallocDefault));
    memset(e->h_pf, 0, 40 * sizeof(unsigned));
  }
  HIPCK(hipStreamSynchronize(e->stream));
  *out = e;
  return 0;
}

static void drop_graphs(csm_engine* e) {
  for (auto& kv : e->graphs) {
    hipGraphExecDestroy(kv.second.exec);
    if (kv.second.d_segs) hipFree(kv.second.d_segs);
  }
  e->graphs.clear();
}

extern "C" int csm_engine_destroy(csm_engine_t* e) {
  if (!e) return 0;
  hipSetDevice(e->device);
  hipStreamSynchronize(e->stream);
  if (e->stream2) hipStreamSynchronize(e->stream2);
  drop_graphs(e);
  if (e->ev_fork) hipEventDestroy(e->ev_fork);
  if (e->ev_join) hipEventDestroy(e->ev_join);
  if (e->h_pf) hipHostFree(e->h_pf);
  for (void* p : e->allocs) hipFree(p);
  if (e->shift_kt) hipFree(e->shift_kt);
  if (e->shift_vt) hipFree(e->shift_vt);
  drop_tiled(e);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  if (e->pooled_own && e->stream2) {   // the pair goes back to the pool (both streams are idle: synchronised above)
    hipStreamSynchronize(e->pooled_own);
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    g_stream_pool.push_back(StreamPair{e->device, e->pooled_own, e->stream2});
  } else {
    if (e->stream2) hipStreamDestroy(e->stream2);
    if (e->pooled_own) hipStreamDestroy(e->pooled_own);
  }
  delete e;
  return 0;
}

static int bind_stack(Stack& s, const csm_stack_weights_t& w, const char* name) {
  if (!w.layers || !w.final_norm || !w.rope_cos || !w.rope_sin) return fail(CSM_ERR_ARG, "%s: null weight pointer", name);
  s.layers.assign(w.layers, w.layers + s.c.layers);
  for (auto& l : s.layers)
    if (!l.wqkv || !l.wo || !l.wgu || !l.wd || !l.ln1 || !l.ln2) return fail(CSM_ERR_ARG, "%s: null layer weight", name);
  s.final_norm = w.final_norm;
  s.cos = w.rope_cos;
  s.sin = w.rope_sin;
  s.rope_positions = w.rope_positions;
  if (s.rope_positions < s.lmax) return fail(CSM_ERR_ARG, "%s: RoPE table covers %d positions, cache needs %d", name, s.rope_positions, s.lmax);
  return 0;
}

// Batched decode (2 <= B) runs on the MFMA skinny GEMM, which streams the weights in fragment order: one extra
// copy of every linear (2.3 GB for csm-1b in bf16 -- HBM capacity is not the constraint on this part) buys fully
// coalesced 1 KiB wavefront loads instead of 64 sixteen-byte pieces spread over 32 cache lines.
static void drop_tiled(csm_engine* e) {
  for (void* p : e->tiled_allocs) hipFree(p);
  e->tiled_allocs.clear();
  e->tiled.clear();
}
static int tile_one(csm_engine* e, const void* W, int N, int K) {
  if (!W || K % 128 || e->tiled.count(W)) return 0;
  const size_t esz = w_esz(e);
  const size_t bytes = (size_t)((N + 15) / 16) * 16 * K * esz;
  void* q = nullptr;
  hipError_t r = hipMalloc(&q, bytes + 256);
  if (r != hipSuccess) return fail(CSM_ERR_NOMEM, "hipMalloc(%zu bytes) for a fragment-order weight copy failed: %s", bytes, hipGetErrorString(r));
  e->tiled_allocs.push_back(q);
  LCK(launch_tile16(e->stream, W, q, N, K, (int)esz));
  e->tiled[W] = q;
  return 0;
}
static int build_tiled(csm_engine* e) {
  drop_tiled(e);
  if (!e->tile_weights || e->cfg.weight_dtype == CSM_DTYPE_F32) return 0;
  // the batched decode kernels (gemm16 / gemm32) stream every matrix from its copy; a one-sequence engine only needs the
  // backbone's, for the wide prefill GEMM (gemm_wide_kernel), and only when its prefills can be large enough to use it
  // (round 3: bf16 prefills of up to gemm_dma_max_rows rows run on the LDS-DMA GEMM, which reads the ROW-MAJOR weights -- a
  // one-sequence bf16 engine then needs no copy at all: -1.9 GB for csm-1b; fp8 weights still take the wide tile)
  const bool dma_covers = e->cfg.weight_dtype == CSM_DTYPE_BF16 && (e->gemm_dma & 1) && e->cfg.max_prefill_rows <= e->gemm_dma_max_rows;
  const bool decode_tiles = e->cfg.max_batch >= 2, prefill_tiles = e->cfg.max_prefill_rows >= 512 && !dma_covers;
  if (!decode_tiles && !prefill_tiles) return 0;
  for (Stack* s : {&e->bb, &e->dec}) {
    if (s == &e->dec && !decode_tiles) continue;
    const int H = s->c.hidden, F = s->c.ffn;
    for (auto& l : s->layers) {
      if (int r = tile_one(e, l.wqkv, s->nqkv(), H)) return r;
      if (int r = tile_one(e, l.wo, H, s->c.n_q * s->c.head_dim)) return r;
      if (int r = tile_one(e, l.wgu, 2 * F, H)) return r;
      if (int r = tile_one(e, l.wd, H, F)) return r;
    }
  }
  const int Hb = e->bb.c.hidden, Hd = e->dec.c.hidden, V = e->cfg.audio_vocab, C = e->cfg.n_codebooks;
  if (!decode_tiles) { HIPCK(hipStreamSynchronize(e->stream)); return 0; }
  if (int r = tile_one(e, e->w.proj_head0, Hd + V, Hb)) return r;
  for (int i = 0; i < C - 1; ++i)
    if (int r = tile_one(e, (const char*)e->w.audio_head_t + (size_t)i * V * Hd * w_esz(e), V, Hd)) return r;
  HIPCK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int csm_bind_weights(csm_engine_t* e, const csm_weights_t* w) {
  if (!e || !w) return fail(CSM_ERR_ARG, "null argument");
  if (!w->text_emb || !w->audio_emb || !w->proj_head0 || !w->audio_head_t) return fail(CSM_ERR_ARG, "null top-level weight");
  if (e->cfg.weight_dtype == CSM_DTYPE_FP8 && (!w->s_proj_head0 || !w->s_audio_head)) return fail(CSM_ERR_ARG, "fp8 weights need row scales");
  if (int r = bind_stack(e->bb, w->backbone, "backbone")) return r;
  if (int r = bind_stack(e->dec, w->decoder, "decoder")) return r;
  e->w = *w;
  e->w.backbone.layers = e->bb.layers.data();
  e->w.decoder.layers = e->dec.layers.data();
  e->bound = true;
  e->train_wT.clear();
  drop_graphs(e);
  return build_tiled(e);
}

extern "C" int csm_set_proj_table(csm_engine_t* e, const float* t) {
  if (!e) return fail(CSM_ERR_ARG, "null engine");
  e->w.proj_table = t;
  drop_graphs(e);
  return 0;
}

extern "C" int csm_reset(csm_engine_t* e) {
  if (!e) return fail(CSM_ERR_ARG, "null engine");
  LCK(launch_set_int(e->stream, e->d_len, 0));
  LCK(launch_set_int(e->stream, e->d_frame, 0));
  HIPCK(hipMemsetAsync(e->d_kv_start, 0, e->cfg.max_batch * sizeof(int), e->stream));
  HIPCK(hipMemsetAsync(e->d_zero_count, 0, (size_t)e->cfg.max_frames * sizeof(int), e->stream));
  HIPCK(hipMemsetAsync(e->d_row_done, 0, (size_t)e->cfg.max_batch * sizeof(int), e->stream));
  e->h_len = e->h_frame = 0;
  e->ready = false;
  e->B = 0;
  return 0;
}

extern "C" int csm_set_kv_start(csm_engine_t* e, const int32_t* kv_start_host, int B) {
  if (!e || !kv_start_host || B < 1 || B > e->cfg.max_batch) return fail(CSM_ERR_ARG, "bad kv_start arguments");
  HIPCK(hipMemcpyAsync(e->d_kv_start, kv_start_host, B * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIPCK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int csm_set_option(csm_engine_t* e, const char* name, int value) {
  if (!e || !name) return fail(CSM_ERR_ARG, "null argument");
  if (!strcmp(name, "nsplit_backbone")) e->nsplit_bb = value < 0 ? 0 : (value > 64 ? 64 : value);
  else if (!strcmp(name, "fuse_attn_oproj")) e->fuse_attn_oproj = value;
  else if (!strcmp(name, "attn_oproj_gqa")) e->attn_oproj_gqa = value ? 1 : 0;
  else if (!strcmp(name, "kernel_prio")) e->kernel_prio = value & 7;
  else if (!strcmp(name, "dbg_sample_spin")) e->dbg_sample_spin = value < 0 ? 0 : value;
  else if (!strcmp(name, "prefill_attn_kvfast")) e->prefill_attn_kvfast = value != 0;
  else if (!strcmp(name, "g16_xcdmap")) e->g16_xcdmap = value != 0;
  else if (!strcmp(name, "g16_kfast")) e->g16_kfast = value != 0;
  else if (!strcmp(name, "g128")) e->g128 = value != 0;
  else if (!strcmp(name, "g128_min")) e->g128_min = value < 32 ? 32 : value;
  else if (!strcmp(name, "g128_shape")) e->g128_shape = value;
  else if (!strcmp(name, "attn_gqa_wide")) e->attn_gqa_wide = value ? 1 : 0;
  else if (!strcmp(name, "oproj_combine")) e->oproj_combine = value ? 1 : 0;
  else if (!strcmp(name, "combine_splits")) e->cmb_splits = value < 2 ? 2 : (value > 8 ? 8 : value);
  else if (!strcmp(name, "fuse_sample")) e->fuse_sample = value;
  else if (!strcmp(name, "two_token_pass")) e->two_token_pass = value;
  else if (!strcmp(name, "use_planes")) e->use_planes = value;
  else if (!strcmp(name, "decode_bf16")) e->decode_bf16 = value ? 1 : 0;
  else if (!strcmp(name, "fuse_attn_combine")) e->fuse_attn_combine = value < 0 ? 0 : (value > 2 ? 2 : value);   // 2: at every batch size (A/B)
  else if (!strcmp(name, "prefill_bf16_attn")) e->prefill_bf16_attn = value;
  else if (!strcmp(name, "gemm_wide")) e->gemm_wide = value;
  else if (!strcmp(name, "gemm_dma")) e->gemm_dma = value;
  else if (!strcmp(name, "gemm_256")) e->gemm_256 = value < 0 ? 0 : value;
  else if (!strcmp(name, "prefill_bf16")) e->prefill_bf16 = value ? 1 : 0;
  else if (!strcmp(name, "mx_fuse_swiglu")) e->mx_fuse_swiglu = value;
  else if (!strcmp(name, "prefill_mx")) {
    if (value && e->mx_layers.empty()) return fail(CSM_ERR_STATE, "prefill_mx needs MX-fp8 weights (csm_bind_mx_weights)");
    e->prefill_mx = value ? 1 : 0;
  }
  else if (!strcmp(name, "prefill_splitk")) e->prefill_splitk = value ? 1 : 0;
  else if (!strcmp(name, "prefill_splitk_gu")) e->prefill_splitk_gu = value < 0 ? 0 : value;
  else if (!strcmp(name, "gemm_mx_skinny")) e->gemm_mx_skinny = value < 0 ? 0 : value;
  else if (!strcmp(name, "gemm_dma_skinny")) e->gemm_dma_skinny = value < 0 ? 0 : value;   // 2 = A/B: exact mode up to 4096 rows
  else if (!strcmp(name, "prefill_fuse_rope")) e->prefill_fuse_rope = value ? 1 : 0;
  else if (!strcmp(name, "prefill_fuse_quant")) e->prefill_fuse_quant = value ? 1 : 0;
  else if (!strcmp(name, "dbg_skip")) { e->dbg_skip = value & 0xffff; e->g16_slab = (value >> 16) << 4; }   // bits 16-19: in-kernel knock-outs of the CSM_G16_KO build
  else if (!strcmp(name, "rows64")) e->rows64 = value < 0 ? -1 : (value ? 1 : 0);   // -1: 16-row launches only (tests: every wider form against gemm16_kernel)
  else if (!strcmp(name, "sample_legacy")) { e->sample_legacy = value ? 1 : 0; return 0; }   // test hook: the stand-alone sampler's old selection path
  else if (!strcmp(name, "weight_prefetch")) e->pf_enable = value;
  else if (!strcmp(name, "prefetch_window_mb")) e->pf_window_mb = value < 1 ? 1 : value;
  else if (!strcmp(name, "prefetch_seg_sleep")) e->pf_seg_sleep = value < 0 ? 0 : value;
  else if (!strcmp(name, "prefetch_batched")) e->pf_batched = value ? 1 : 0;
  else if (!strcmp(name, "g16_k16")) e->g16_k16 = value < 0 ? 0 : value;
  else if (!strcmp(name, "prefetch_sub_kb")) e->pf_sub_kb = value < 64 ? 64 : value;
  else if (!strcmp(name, "prefetch_lead")) e->pf_lead = value ? 1 : 0;
  else if (!strcmp(name, "prefetch_cofetch")) e->pf_cofetch = value ? 1 : 0;
  else if (!strcmp(name, "prefetch_skip_late")) e->pf_skip_late = value ? 1 : 0;
  else if (!strcmp(name, "prefetch_depth")) e->pf_depth = value;
  else if (!strcmp(name, "prefetch_poll_sleep")) e->pf_poll_sleep = value < 0 ? 0 : value;
  else if (!strcmp(name, "prefetch_stride")) e->pf_stride = (value == 64 || value == 128) ? value : 0;
  // streamer health: launch parameters of the streamer, not of the captured frame-step -- the graphs stay
  else if (!strcmp(name, "prefetch_budget_us")) { e->pf_budget_us = value < 100 ? 100 : value; return 0; }
  else if (!strcmp(name, "prefetch_force_serial")) { e->pf_force_serial = value ? 1 : 0; return 0; }
  else if (!strcmp(name, "prefetch_rearm")) {   // clears a run-time switch-off (1 / 2) and the strike count; the creation-time verdicts (3 / 4) stay
    if (e->pf_disabled == 1 || e->pf_disabled == 2) e->pf_disabled = 0;
    e->pf_strikes = 0; e->pf_clean = 0;
    return 0;
  }
  else if (!strcmp(name, "tile_weights")) {   // A/B: 0 drops the fragment-order copies (row-major MFMA path)
    e->tile_weights = value;
    if (e->bound) { if (int r = build_tiled(e)) return r; }
  }
  else return fail(CSM_ERR_ARG, "unknown option %s", name);
  drop_graphs(e);
  return 0;
}

// timeline probe (-DCSM_TIMELINE build, tools/b1_timeline.py): while a frame-step is captured with a debug buffer set, every
// decode-path launch gets the next 4096-word slot, in launch order; null otherwise
static uint32_t* tl_slot(csm_engine* e, int n = 1) {
#ifdef CSM_TIMELINE
  if (!e->pf_rec || !e->dbg_buf || e->tl_n + n > e->dbg_launches) return nullptr;
  uint32_t* p = e->dbg_buf + (size_t)e->tl_n * 4096;
  e->tl_n += n;
  return p;
#else
  (void)e; (void)n;
  return nullptr;
#endif
}

// ---- decode-side GEMV with row grouping (M <= 4 per launch) -----------------------------------------
static int gemv_rows(csm_engine* e, int M, int pro, int epi, GemvArgs a) {
  const float* x = a.x;
  float* out = a.out;
  float* q = a.qbuf;
  const int* rs = a.row_seq;
  const int* rp = a.row_pos;
  int* ba = a.bump_a;
  int* bbp = a.bump_b;
  const bf16_t* xpl = a.xplanes;
  const float* xss = a.xss;
  bf16_t* opl = a.oplanes;
  float* oss = a.oss;
  const size_t opl_group = (size_t)3 * 16 * (epi == EPI_SWIGLU ? a.N / 2 : a.N);   // plane group of 16 rows
  a.pl1 = e->decode_bf16;
  // rows are processed in groups: up to 16 on the matrix-core kernel (bf16 / fp8 weights, eligible shapes),
  // otherwise up to 4 on the fp32-FMA kernels; every group re-streams the weights
  int m0 = 0;
  while (m0 < M) {
    const int left = M - m0;
    auto slice = [&](int m) {
      a.x = x + (size_t)m0 * a.ldx;
      a.out = out ? out + (size_t)m0 * a.ldo : nullptr;
      a.qbuf = q ? q + (size_t)m0 * a.n_q * a.hd : nullptr;
      a.row_seq = rs ? rs + m0 : nullptr;
      a.row_pos = rp ? rp + m0 : nullptr;
      a.seq_base = m0;
      const bool last = m0 + m >= M;
      a.bump_a = last ? ba : nullptr;
      a.bump_b = last ? bbp : nullptr;
      a.xplanes = xpl ? xpl + (size_t)(m0 / 16) * 3 * 16 * a.K : nullptr;
      a.xss = xss ? xss + (size_t)m0 * a.xss_ld : nullptr;
      a.oplanes = opl ? opl + (size_t)(m0 / 16) * opl_group : nullptr;
      a.oss = oss ? oss + (size_t)m0 * a.oss_ld : nullptr;
    };
    const auto tl = e->tiled.find(a.W);
    a.Wt = tl == e->tiled.end() ? nullptr : tl->second;
    if (left > 16 && e->rows64 >= 0 && e->use_mfma && !a.no_mfma && xpl && a.Wt && m0 % 32 == 0) {   // 17..32 rows on planes (33..64 with rows64): one launch, weights streamed once
      const int cap = (e->rows64 > 0 && m0 % 128 == 0 && left > 64) ? 128 : ((e->rows64 > 0 && m0 % 64 == 0 && left > 32) ? 64 : 32);
      const int m = left < cap ? left : cap;
      slice(m);
      a.kfast = e->g16_kfast; a.xcdmap = e->g16_kfast && e->g16_xcdmap;
      int r = -2;
      if (e->g128 && m > e->g128_min) {   // 65..128 rows, FFN launches: weight rows split over the waves, planes shared through LDS (gemm128.h)
        a.g128_shape = e->g128_shape;
        r = launch_gemm128(e->stream, e->cfg.weight_dtype, m, pro, epi, a, e->g16_slabs, e->g16_slab_floats, e->g16_tickets, 4096);
      }
      if (r == -2)
        r = launch_gemm32(e->stream, e->cfg.weight_dtype, e->cfg.kv_dtype, m, pro, epi, a, e->g16_slabs,
                          e->g16_slab_floats, e->g16_tickets, 4096);
      if (r != -2) {
        if (r) return r;
        m0 += m;
        continue;
      }
    }
    if (left >= 2 && e->use_mfma && !a.no_mfma) {
      const int m = left < 16 ? left : 16;
      slice(m);
      if (!a.Wt) a.xplanes = nullptr;   // unbound weights (hooks): fp32 activations
      PfGeom geom{};
      geom.kind = -1;
      // capture of a single-group batch (M <= 16): the launch is paced / streamed (prefetch.h); larger batches re-stream
      // every matrix once per group and are left alone
      const bool rec = e->pf_rec && M <= 16 && e->pf_batched;
      if (!a.g16_nw && e->g16_k16 && a.K == 2048 && a.xplanes) { a.g16_nw = e->g16_k16 & 0xff; a.g16_kb = (e->g16_k16 >> 8) & 0xff; }
      if (rec) { a.prog = e->d_prog; a.geom_out = &geom; }
      if (e->g16_slab >> 4) a.g16_slab = (a.g16_slab & 3) | (e->g16_slab & ~15);   // TIMING-ONLY in-kernel knock-outs (gemm16.h), every matrix-core launch
      a.dbg = tl_slot(e);
      a.prio = (e->kernel_prio >> 1) & 1;
      a.kfast = rec ? 0 : e->g16_kfast;   // (the streamer's recorded geometry is panels-fastest)
      const int r = launch_gemm16(e->stream, e->cfg.weight_dtype, e->cfg.kv_dtype, m, pro, epi, a, e->g16_slabs,
                                  e->g16_slab_floats, e->g16_tickets, 4096);
      a.prog = nullptr; a.geom_out = nullptr;
      if (r != -2) {
        if (r) return r;
        if (rec) e->pf_rec->push_back(geom);
        m0 += m;
        continue;
      }
    }
    if (a.xplanes || a.oplanes) return fail(CSM_ERR_STATE, "activation planes requested but the matrix-core kernel does not cover N=%d K=%d M=%d", a.N, a.K, M);
    const int m = left < 4 ? left : 4;
    slice(m);
    PfGeom geom{};
    geom.kind = -1;
    if (e->pf_rec) { a.prog = e->d_prog; a.geom_out = &geom; }   // capture: this launch is paced / streamed (prefetch.h)
    a.prio = (e->kernel_prio >> 1) & 1;
#ifdef CSM_TIMELINE
    a.dbg = tl_slot(e);
#else
    if (e->pf_rec && e->dbg_buf && (int)e->pf_rec->size() < e->dbg_launches) a.dbg = e->dbg_buf + e->pf_rec->size() * 4096;
#endif
    const int r = launch_gemv(e->stream, e->cfg.weight_dtype, e->cfg.kv_dtype, m, pro, epi, a);
    a.prog = nullptr; a.geom_out = nullptr; a.dbg = nullptr;
    if (r) return r;
    if (e->pf_rec) e->pf_rec->push_back(geom);
    m0 += m;
  }
  return 0;
}

// activation planes are used when every launch of the stack runs on the matrix-core kernel with fragment-order weights
static bool planes_on(const csm_engine* e, const Stack& s, int M) {
  const int H = s.c.hidden, F = s.c.ffn, A = s.c.n_q * s.c.head_dim;
  return e->use_planes && e->use_mfma && M >= 2 && M <= 16 * PL_GROUPS && !e->tiled.empty() && H % 512 == 0 && F % 512 == 0 &&
         A % 512 == 0 && H <= 16 * PL_SS_LD && (e->cfg.weight_dtype == CSM_DTYPE_BF16 || e->cfg.weight_dtype == CSM_DTYPE_FP8);
}

// one Llama layer on M single-token rows (decode)
static int layer_decode(csm_engine* e, Stack& s, int l, int M, float* h, int ldh, const int* pos_ptr, int pos_const,
                        float* qb, float* att, float* part, int nsplit, float* act, int nt,
                        const GemvArgs* tok = nullptr, bool kv_only = false, bool in_planes = false,
                        const float* next_ln = nullptr) {
  const csm_layer_weights_t& w = s.layers[l];
  const int H = s.c.hidden, nq = s.c.n_q, nkv = s.c.n_kv, hd = s.c.head_dim, F = s.c.ffn;
  // planes: o_proj and down_proj emit the residual stream as B operands for the next normed launch (folded with that
  // launch's norm weight, plus per-tile sums of squares); gate/up emits the SwiGLU output the same way for down_proj
  const bool planes = planes_on(e, s, M);
  // nt: 0 = plain loads, 1 = every matrix non-temporal, 2 = only the large streams (gate/up, down) non-temporal so
  // that the small per-pass matrices (qkv, o) can stay in the XCD-local L2 between decoder passes
  const int nt_small = nt == 1, nt_big = nt >= 1;
  const int sk = (&s == &e->dec) ? e->dbg_skip & 0xff : (e->dbg_skip >> 8) & 0xff;
  GemvArgs a{};
  a.nt = nt_small;
  a.W = w.wqkv; a.wscale = w.sqkv; a.N = s.nqkv(); a.K = H; a.x = h; a.ldx = ldh; a.ln = w.ln1; a.eps = s.c.rms_eps;
  a.n_q = nq; a.n_kv = nkv; a.hd = hd; a.qscale = 1.0f / sqrtf((float)hd);
  a.cos_tab = s.cos; a.sin_tab = s.sin; a.pos_ptr = pos_ptr; a.pos_const = pos_const;
  a.qbuf = qb; a.kcache = s.kc[l]; a.vcache = s.vc[l]; a.lmax = s.lmax;
  a.norm_ks = e->gemv_norm_ks & 1 ? 2 : 0;
  if (planes && in_planes) { a.xplanes = e->pl_h; a.xss = e->pl_ss; a.xss_n = H / 16; a.xss_ld = PL_SS_LD; }
  if (tok) {  // the row comes from (partials -> token -> projected-embedding table); h is written by workgroup 0
    a.am_in = tok->am_in; a.am_n = tok->am_n; a.tok_table = tok->tok_table; a.tok_row_base = tok->tok_row_base;
    a.tok_forced = tok->tok_forced; a.tok_ring = tok->tok_ring; a.tok_frame_ptr = tok->tok_frame_ptr;
    a.tok_max_frames = tok->tok_max_frames; a.tok_C = tok->tok_C; a.tok_cb = tok->tok_cb; a.tok_x_out = h;
    a.smp = tok->smp; a.smp_row_done = tok->smp_row_done;
    LCK(gemv_rows(e, M, tok->am_in ? PRO_TOKNORM : PRO_SAMPLE, EPI_QKV, a));
  } else if (!(sk & 1)) {
    LCK(gemv_rows(e, M, PRO_NORM, EPI_QKV, a));
  }
  if (kv_only) return 0;  // this position's hidden state is never read again: only its K/V had to be appended

  GemvArgs o{};
  o.nt = nt_small;
  o.W = w.wo; o.wscale = w.so; o.N = H; o.K = nq * hd; o.ldx = nq * hd; o.out = h; o.ldo = ldh;
  int ao = -2;
  if (M == 1 && (e->fuse_attn_oproj & 1) && &s == &e->dec && s.lmax <= 32 && (sk & 32)) {
    ao = 0;   // dbg_skip bit 5: the fused attention + o_proj launch knocked out (timing only)
  } else if (M == 1 && (e->fuse_attn_oproj & 1) && &s == &e->dec && s.lmax <= 32) {
    // single sequence, short cache: one launch for SDPA + o_proj (heads in parallel on the waves of each o_proj workgroup)
    AttnOprojArgs f{};
    f.q = qb; f.kcache = s.kc[l]; f.vcache = s.vc[l]; f.n_q = nq; f.n_kv = nkv; f.hd = hd; f.lmax = s.lmax;
    f.pos_ptr = pos_ptr; f.pos_const = pos_const; f.W = w.wo; f.wscale = w.so; f.N = H; f.out = h;
    f.beside_streamer = e->pf_enable && e->pf_rot >= 0;
    f.dbg_onekey = (e->dbg_skip >> 7) & 1;
    f.dbg = tl_slot(e);
    f.gqa = e->attn_oproj_gqa;
    f.prio = e->kernel_prio & 1;
    ao = launch_attn_oproj(e->stream, e->cfg.weight_dtype, e->cfg.kv_dtype, f);
    if (ao != -2) LCK(ao);
  }
  // B = 1 backbone (round 5): attention on few long splits with the K/V tiles shared by the query heads of a kv-head
  // (attn_decode_gqa_kernel), the split merge folded into the o_proj launch (gemv1_combine_kernel): five launches per layer
  const bool cmb = M == 1 && &s == &e->bb && e->oproj_combine && hd == 64 && nq == 4 * nkv && nq * hd == 2048 && nsplit > 1 && nsplit <= 8;
  if (ao != -2) {
  } else if (cmb) {
    AttnArgs t{};
    t.q = qb; t.kcache = s.kc[l]; t.vcache = s.vc[l]; t.n_q = nq; t.n_kv = nkv; t.hd = hd; t.lmax = s.lmax;
    t.pos_ptr = pos_ptr; t.pos_const = pos_const; t.kv_start = e->d_kv_start;
    t.nsplit = nsplit; t.out = att; t.part = part; t.gqa = 1; t.no_combine = 1;
    t.dbg = tl_slot(e);
    t.prio = (e->kernel_prio >> 2) & 1;
    if (!(sk & 2)) LCK(launch_attn(e->stream, e->cfg.kv_dtype, 1, t));
    o.cmb_part = part; o.cmb_ns = nsplit; o.x = nullptr;
    if (!(sk & 4)) LCK(gemv_rows(e, 1, PRO_COMBINE, EPI_RESID, o));
  } else {
    AttnArgs t{};
    t.q = qb; t.kcache = s.kc[l]; t.vcache = s.vc[l]; t.n_q = nq; t.n_kv = nkv; t.hd = hd; t.lmax = s.lmax;
    t.pos_ptr = pos_ptr; t.pos_const = pos_const; t.kv_start = (&s == &e->bb) ? e->d_kv_start : nullptr;
    t.nsplit = nsplit; t.out = att; t.part = part;
    t.tickets = (&s == &e->bb && nsplit > 1 && (e->fuse_attn_combine == 2 || (e->fuse_attn_combine && M >= 2 && M <= 32))) ? e->attn_tickets : nullptr;
    t.one_wave = (&s == &e->bb) ? (e->attn_one_wave >> 1) & 1 : e->attn_one_wave & 1;
    t.tile_prefetch = (M == 1 ? e->attn_prefetch & 1 : (e->attn_prefetch >> 1) & 1);
    // the attention output goes to o_proj as planes too (staged in the SwiGLU plane buffer, which is free here)
    const bool att_planes = planes && (e->use_planes & 4);
    t.oplanes = att_planes ? e->pl_act : nullptr;
    t.pl1 = e->decode_bf16;
    // round 5: batches too wide for the in-launch merge (> 32 rows) take the key-quarter kernel -- the four query heads of a kv-head share every
    // K / V tile load; the backbone attention of a 128-row step is bound by its load instructions and bytes, not its arithmetic
    t.gqa = (&s == &e->bb) && e->attn_gqa_wide && !t.tickets && M > 32;
    t.prio = 0;
    t.dbg = tl_slot(e, (nsplit > 1 && !t.tickets) ? 2 : 1);
    if (!(sk & 2)) LCK(launch_attn(e->stream, e->cfg.kv_dtype, M, t));
    o.x = att;
    if (att_planes) o.xplanes = e->pl_act;
    if (planes) { o.oplanes = e->pl_h; o.oln = w.ln2; o.oss = e->pl_ss; o.oss_ld = PL_SS_LD; }
    if (!(sk & 4)) LCK(gemv_rows(e, M, PRO_PLAIN, EPI_RESID, o));
  }

  GemvArgs g{};
  g.nt = nt_big;
  g.W = w.wgu; g.wscale = w.sgu; g.N = 2 * F; g.K = H; g.x = h; g.ldx = ldh; g.ln = w.ln2; g.eps = s.c.rms_eps; g.out = act; g.ldo = F;
  g.norm_ks = e->gemv_norm_ks & 2 ? 2 : 0;
  {  // panel tiles of the gate/up launch: low byte decoder, next byte backbone (0 = auto)
    const int pt = (&s == &e->bb) ? (e->g16_gu >> 8) & 0xff : e->g16_gu & 0xff;
    if (pt) g.g16_pt = pt;
  }
  const bool act_planes = planes && (e->use_planes & 2);   // A/B: bit 1 = SwiGLU output handed over as planes too
  if (planes) { g.xplanes = e->pl_h; g.xss = e->pl_ss; g.xss_n = H / 16; g.xss_ld = PL_SS_LD; g.oplanes = act_planes ? e->pl_act : nullptr; }
  if (!(sk & 8)) LCK(gemv_rows(e, M, PRO_NORM, EPI_SWIGLU, g));

  GemvArgs d{};
  d.nt = nt_big;
  d.W = w.wd; d.wscale = w.sd; d.N = H; d.K = F; d.x = act; d.ldx = F; d.out = h; d.ldo = ldh;
  if (e->g16_down) { d.g16_nw = e->g16_down & 0xff; d.g16_kb = (e->g16_down >> 8) & 0xff; d.g16_pt = (e->g16_down >> 16) & 0xff; }
  if (planes) { d.xplanes = act_planes ? e->pl_act : nullptr; d.oplanes = e->pl_h; d.oln = next_ln; d.oss = e->pl_ss; d.oss_ld = PL_SS_LD; }
  if (!(sk & 16)) LCK(gemv_rows(e, M, PRO_PLAIN, EPI_RESID, d));
  return 0;
}

// final norm + [projection ; codebook0_head] on the backbone residual rows -> head_out
static int backbone_head(csm_engine* e, const float* h, int ldh, int M, bool bump_len, bool bump_frame) {
  GemvArgs a{};
  a.nt = e->nt_backbone;
  a.W = e->w.proj_head0; a.wscale = e->w.s_proj_head0; a.N = e->cfg.decoder.hidden + e->cfg.audio_vocab; a.K = e->cfg.backbone.hidden;
  a.x = h; a.ldx = ldh; a.ln = e->bb.final_norm; a.eps = e->bb.c.rms_eps; a.out = e->head_out; a.ldo = e->ld_head;
  if (planes_on(e, e->bb, M) && h == e->h_bb) {   // the last layer's down_proj left the planes (final norm folded in)
    a.xplanes = e->pl_h; a.xss = e->pl_ss; a.xss_n = e->cfg.backbone.hidden / 16; a.xss_ld = PL_SS_LD;
  }
  if (bump_len) {
    a.bump_a = e->d_len;
    a.bump_b = bump_frame ? e->d_frame : nullptr;
  } else if (bump_frame) {
    a.bump_a = e->d_frame;
  }
  return gemv_rows(e, M, PRO_NORM, EPI_STORE, a);
}

static int backbone_step_impl(csm_engine* e, const csm_sampling_t* s, bool from_ids, bool advance_frame, bool want_last_h) {
  const int B = e->B, Hb = e->cfg.backbone.hidden;
  EmbedArgs em{};
  em.text_emb = e->w.text_emb; em.audio_emb = e->w.audio_emb; em.H = Hb; em.C = e->cfg.n_codebooks; em.V = e->cfg.audio_vocab;
  if (from_ids) {
    em.ids = e->ids_stage;
    em.mask = e->mask_stage;
  } else {
    em.ring = (s && s->forced) ? s->forced : e->ring;
    em.frame_ptr = e->d_frame;
    em.max_frames = e->cfg.max_frames;
    em.zero_count = e->d_zero_count;
    em.row_done = e->d_row_done;
  }
  em.out = e->h_bb;
  // batched decode on planes: the embedding sum hands layer 0 its operands like every later producer does
  const bool em_planes = planes_on(e, e->bb, B) && (e->use_planes & 16);
  if (em_planes) { em.oplanes = e->pl_h; em.oln = e->bb.layers[0].ln1; em.oss = e->pl_ss; em.oss_ld = PL_SS_LD; em.pl1 = e->decode_bf16; }
  em.dbg = tl_slot(e);
  LCK(launch_embed(e->stream, emb_dtype(e), B, em));
  for (int l = 0; l < e->bb.c.layers; ++l)
    LCK(layer_decode(e, e->bb, l, B, e->h_bb, Hb, e->d_len, 0, e->q_bb, e->att_bb, e->part_bb, e->nsplit_eff(), e->act_bb, e->nt_backbone,
                     nullptr, false, l > 0 || em_planes, l + 1 < e->bb.c.layers ? e->bb.layers[l + 1].ln1 : e->bb.final_norm));
  if (want_last_h) {
    LCK(launch_rmsnorm(e->stream, e->h_bb, Hb, e->bb.final_norm, B, Hb, e->bb.c.rms_eps, e->last_h, Hb, nullptr, 0, 0));
    if (s && s->last_h_trace)
      LCK(launch_rmsnorm(e->stream, e->h_bb, Hb, e->bb.final_norm, B, Hb, e->bb.c.rms_eps, s->last_h_trace, Hb, e->d_frame,
                         (size_t)B * Hb, advance_frame ? 1 : 0));
  }
  LCK(backbone_head(e, e->h_bb, Hb, B, true, advance_frame));
  return 0;
}

// The reference's first decoder forward of a frame takes TWO positions at once -- the projected backbone state (position 0)
// and the projected embedding of codebook 0 (position 1), modeling_csm.py:534-552 -- and only position 1's output is used.
// B == 1: rows x2[0] (position 0) and x2[1] (position 1) go through the decoder stack as ONE pass of 2-row launches (the
// M <= 4 skinny GEMM, causal 2-row attention); the last layer appends both positions' K/V and then continues with row 1
// alone on the single-row kernels.  Against two one-token passes: one weight pass (222 MB) and 10 launches fewer per frame;
// a row's arithmetic is that of the single-row kernels (gemv.h), so the greedy stream is unchanged (bench parity: all
// 3 520 tokens equal).  MEASURED (round 3, profiles/r03_b1_ab.txt): on the LDS-staged 2-row kernel (gemv_kernel<M = 2>) the
// merged pass LOST (3.200 vs 3.164 ms per frame-step: 15 slow launches against 10 launches + one weight pass saved); with a
// 2-row form of the register kernel (gemv1_kernel<..., M = 2>: both rows share every weight register) it wins:
// 3.114-3.129 vs 3.169 ms (-1.5 %).  Default on (`two_token_pass`).
static int decoder_two_token_pass(csm_engine* e, float* x2) {
  Stack& s = e->dec;
  const int H = s.c.hidden, nq = s.c.n_q, nkv = s.c.n_kv, hd = s.c.head_dim, F = s.c.ffn, A = nq * hd;
  const int nt = e->nt_decoder, nt_small = nt == 1, nt_big = nt >= 1;
  for (int l = 0; l < s.c.layers; ++l) {
    const csm_layer_weights_t& w = s.layers[l];
    const bool last = l + 1 == s.c.layers;
    GemvArgs a{};
    a.nt = nt_small;
    a.W = w.wqkv; a.wscale = w.sqkv; a.N = s.nqkv(); a.K = H; a.x = x2; a.ldx = H; a.ln = w.ln1; a.eps = s.c.rms_eps;
    a.n_q = nq; a.n_kv = nkv; a.hd = hd; a.qscale = 1.0f / sqrtf((float)hd);
    a.cos_tab = s.cos; a.sin_tab = s.sin; a.row_pos = e->d_pos01; a.row_seq = e->d_seq00;
    a.qbuf = e->q_dec2; a.kcache = s.kc[l]; a.vcache = s.vc[l]; a.lmax = s.lmax;
    a.no_mfma = 1;
    LCK(gemv_rows(e, 2, PRO_NORM, EPI_QKV, a));
    if (!last) {
      AttnArgs t{};
      t.q = e->q_dec2; t.kcache = s.kc[l]; t.vcache = s.vc[l]; t.n_q = nq; t.n_kv = nkv; t.hd = hd; t.lmax = s.lmax;
      t.row_seq = e->d_seq00; t.row_pos = e->d_pos01; t.nsplit = 1; t.out = e->att_dec2; t.one_wave = e->attn_one_wave & 1;
      t.dbg = tl_slot(e);
      LCK(launch_attn(e->stream, e->cfg.kv_dtype, 2, t));
      GemvArgs o{};
      o.nt = nt_small; o.W = w.wo; o.wscale = w.so; o.N = H; o.K = A; o.x = e->att_dec2; o.ldx = A; o.out = x2; o.ldo = H;
      o.no_mfma = 1;
      LCK(gemv_rows(e, 2, PRO_PLAIN, EPI_RESID, o));
      GemvArgs g{};
      g.nt = nt_big; g.W = w.wgu; g.wscale = w.sgu; g.N = 2 * F; g.K = H; g.x = x2; g.ldx = H; g.ln = w.ln2; g.eps = s.c.rms_eps;
      g.out = e->act_dec2; g.ldo = F;
      g.no_mfma = 1;
      LCK(gemv_rows(e, 2, PRO_NORM, EPI_SWIGLU, g));
      GemvArgs d{};
      d.nt = nt_big; d.W = w.wd; d.wscale = w.sd; d.N = H; d.K = F; d.x = e->act_dec2; d.ldx = F; d.out = x2; d.ldo = H;
      d.no_mfma = 1;
      LCK(gemv_rows(e, 2, PRO_PLAIN, EPI_RESID, d));
    } else {
      // position 0's hidden state is never read again: only row 1 continues (single-row kernels, position 1)
      float* h1 = x2 + H;
      const float* q1 = e->q_dec2 + A;
      int ao = -2;
      if ((e->fuse_attn_oproj & 1) && s.lmax <= 32) {
        AttnOprojArgs f{};
        f.q = q1; f.kcache = s.kc[l]; f.vcache = s.vc[l]; f.n_q = nq; f.n_kv = nkv; f.hd = hd; f.lmax = s.lmax;
        f.pos_ptr = nullptr; f.pos_const = 1; f.W = w.wo; f.wscale = w.so; f.N = H; f.out = h1;
        f.beside_streamer = e->pf_enable && e->pf_rot >= 0;
        f.dbg = tl_slot(e);
        f.gqa = e->attn_oproj_gqa;
        f.prio = e->kernel_prio & 1;
        ao = launch_attn_oproj(e->stream, e->cfg.weight_dtype, e->cfg.kv_dtype, f);
        if (ao != -2) LCK(ao);
      }
      if (ao == -2) {
        AttnArgs t{};
        t.q = q1; t.kcache = s.kc[l]; t.vcache = s.vc[l]; t.n_q = nq; t.n_kv = nkv; t.hd = hd; t.lmax = s.lmax;
        t.pos_ptr = nullptr; t.pos_const = 1; t.nsplit = 1; t.out = e->att_dec2; t.one_wave = e->attn_one_wave & 1;
        t.dbg = tl_slot(e);
        LCK(launch_attn(e->stream, e->cfg.kv_dtype, 1, t));
        GemvArgs o{};
        o.nt = nt_small; o.W = w.wo; o.wscale = w.so; o.N = H; o.K = A; o.x = e->att_dec2; o.ldx = A; o.out = h1; o.ldo = H;
        LCK(gemv_rows(e, 1, PRO_PLAIN, EPI_RESID, o));
      }
      GemvArgs g{};
      g.nt = nt_big; g.W = w.wgu; g.wscale = w.sgu; g.N = 2 * F; g.K = H; g.x = h1; g.ldx = H; g.ln = w.ln2; g.eps = s.c.rms_eps;
      g.out = e->act_dec2; g.ldo = F;
      LCK(gemv_rows(e, 1, PRO_NORM, EPI_SWIGLU, g));
      GemvArgs d{};
      d.nt = nt_big; d.W = w.wd; d.wscale = w.sd; d.N = H; d.K = F; d.x = e->act_dec2; d.ldx = F; d.out = h1; d.ldo = H;
      LCK(gemv_rows(e, 1, PRO_PLAIN, EPI_RESID, d));
    }
  }
  return 0;
}

static int decode_frame_impl(csm_engine* e, const csm_sampling_t* s) {
  const int B = e->B, C = e->cfg.n_codebooks, V = e->cfg.audio_vocab, Hd = e->cfg.decoder.hidden;
  // B == 1: positions 0 and 1 of the decoder as one two-row pass (decoder_two_token_pass); x2[0] = position 0's input,
  // x2[1] = the row every later pass works on
  const bool two_tok = e->two_token_pass && B == 1 && C >= 2 && e->dec.lmax >= 2;
  float* const decx = two_tok ? e->dec_x2 + Hd : e->dec_x;
  auto sample = [&](int cb, const float* logits, int ldl) -> int {
    SampleArgs a{};
    a.logits = logits; a.ldl = ldl; a.V = V; a.temperature = s->temperature; a.topk = s->topk; a.rng = e->d_rng;
    if (s->noise) {
      a.noise = s->noise + (size_t)cb * V;
      a.noise_ld = (size_t)C * V;
    }
    a.cb = cb; a.C = C; a.B = B; a.frame_ptr = e->d_frame; a.max_frames = e->cfg.max_frames;
    a.ring = e->ring; a.forced = s->forced; a.proj_table = e->w.proj_table; a.Hd = Hd; a.dec_x = decx;
    if (two_tok && cb == 0) { a.copy_src = e->head_out; a.copy_dst = e->dec_x2; a.copy_n = Hd; }
    a.logits_trace = s->logits_trace;
    a.row_done = s->per_row_stop ? e->d_row_done : nullptr;
    if (planes_on(e, e->dec, B) && (e->use_planes & 8)) {
      a.oplanes = e->pl_h; a.oln = e->dec.layers[0].ln1; a.oss = e->pl_ss; a.oss_ld = PL_SS_LD; a.oss_n = Hd / 16; a.pl1 = e->decode_bf16;
    }
    a.dbg = tl_slot(e);
    a.spin_ticks = e->dbg_sample_spin;
    a.prio = (e->kernel_prio >> 2) & 1;
    return launch_sample(e->stream, B, a);
  };
  // B == 1 greedy without traces: codebooks 1..C-2 need no sampler launch -- the head writes per-task argmax
  // pairs and the next pass's first QKV launch turns them into the token and its input row
  const bool greedy = s->topk <= 1 || s->temperature == 0.f;
  const bool fused = e->fuse_sample && B == 1 && greedy && !s->noise && !s->logits_trace && Hd % 512 == 0 && Hd <= 1024 &&
                     (V + 1) / 2 <= 1088;
  // B == 1 top-k sampling without traces (round 5): codebooks 1..C-2 need no sampler launch either -- the head writes its logits
  // and every wave of the next pass's first QKV launch draws the token itself (sample_wave.h: sample_kernel's arithmetic, no
  // workgroup barrier), under that launch's weight loads
  const bool fused_smp = e->fuse_sample && B == 1 && !greedy && !s->logits_trace && Hd % 512 == 0 && Hd <= 1024 && V <= WS_VMAX && V >= WS_VMIN;
  LCK(sample(0, e->head_out + Hd, e->ld_head));
  if (two_tok) LCK(decoder_two_token_pass(e, e->dec_x2));
  for (int p = two_tok ? 1 : 0; p < C; ++p) {
    float* h = p == 0 ? e->head_out : decx;
    const int ldh = p == 0 ? e->ld_head : Hd;
    GemvArgs tok{};
    const bool use_tok = (fused || fused_smp) && p >= 2;   // input of pass p = token of codebook p-1 (sampled by head p-1)
    if (use_tok) {
      tok.am_in = e->am_part; tok.am_n = (V + 1) / 2; tok.tok_table = e->w.proj_table; tok.tok_row_base = (p - 1) * V;
      tok.tok_forced = s->forced; tok.tok_ring = e->ring; tok.tok_frame_ptr = e->d_frame;
      tok.tok_max_frames = e->cfg.max_frames; tok.tok_C = C; tok.tok_cb = p - 1;
      if (fused_smp) {
        tok.am_in = nullptr;
        tok.smp.logits = e->logits_dec; tok.smp.V = V; tok.smp.temperature = s->temperature; tok.smp.topk = s->topk;
        tok.smp.rng = e->d_rng; tok.smp.noise = s->noise ? s->noise + (size_t)(p - 1) * V : nullptr; tok.smp.cb = p - 1;
        tok.smp_row_done = s->per_row_stop ? e->d_row_done : nullptr;
      }
    }
    for (int l = 0; l < e->dec.c.layers && !(two_tok && p == 1); ++l)
      LCK(layer_decode(e, e->dec, l, B, h, ldh, nullptr, p, e->q_dec, e->att_dec, nullptr, 1, e->act_dec, e->nt_decoder,
                       (use_tok && l == 0) ? &tok : nullptr,
                       // pass 0 (the backbone state at position 0) produces no logits: its last layer only has to
                       // append K/V -- the attention, o_proj and MLP of that layer are dead work
                       p == 0 && l == e->dec.c.layers - 1, l > 0 || (p >= 2 && (e->use_planes & 8) && !use_tok),   // p == 1: codebook 0 was sampled before pass 0 overwrote the planes
                       l + 1 < e->dec.c.layers ? e->dec.layers[l + 1].ln1 : e->dec.final_norm));
    if (p >= 1 && fused && p < C - 1) {
      GemvArgs a{};
      a.nt = e->nt_backbone;
      a.W = (const char*)e->w.audio_head_t + (size_t)(p - 1) * V * Hd * w_esz(e);
      a.wscale = e->w.s_audio_head ? e->w.s_audio_head + (size_t)(p - 1) * V : nullptr;
      a.N = V; a.K = Hd; a.x = h; a.ldx = ldh; a.ln = e->dec.final_norm; a.eps = e->dec.c.rms_eps;
      a.out = e->logits_dec; a.ldo = (V + 3) & ~3; a.am_out = e->am_part; a.am_from = 0;
      if (!(e->dbg_skip & 64)) LCK(gemv_rows(e, B, PRO_NORM, EPI_ARGMAX, a));   // dbg_skip bit 6: the fused-argmax head launches (timing only)
    } else if (p >= 1 && fused_smp && p < C - 1) {
      GemvArgs a{};
      a.nt = e->nt_backbone;
      a.W = (const char*)e->w.audio_head_t + (size_t)(p - 1) * V * Hd * w_esz(e);
      a.wscale = e->w.s_audio_head ? e->w.s_audio_head + (size_t)(p - 1) * V : nullptr;
      a.N = V; a.K = Hd; a.x = h; a.ldx = ldh; a.ln = e->dec.final_norm; a.eps = e->dec.c.rms_eps;
      a.out = e->logits_dec; a.ldo = (V + 3) & ~3;
      a.store_div = s->temperature;                      // logits / T, the sampler's first step, done here once per logit
      LCK(gemv_rows(e, B, PRO_NORM, EPI_STORE, a));      // logits only: the next pass's first launch samples from them
    } else if (p >= 1) {
      GemvArgs a{};
      a.nt = e->nt_backbone;  // each audio_head slice is read once per frame
      a.W = (const char*)e->w.audio_head_t + (size_t)(p - 1) * V * Hd * w_esz(e);
      a.wscale = e->w.s_audio_head ? e->w.s_audio_head + (size_t)(p - 1) * V : nullptr;
      a.N = V; a.K = Hd; a.x = h; a.ldx = ldh; a.ln = e->dec.final_norm; a.eps = e->dec.c.rms_eps;
      a.out = e->logits_dec; a.ldo = (V + 3) & ~3;
      if (planes_on(e, e->dec, B)) { a.xplanes = e->pl_h; a.xss = e->pl_ss; a.xss_n = Hd / 16; a.xss_ld = PL_SS_LD; }
      LCK(gemv_rows(e, B, PRO_NORM, EPI_STORE, a));
      LCK(sample(p, e->logits_dec, (V + 3) & ~3));
    }
  }
  return 0;
}

static int check_ready(csm_engine* e, const csm_sampling_t* s) {
  if (!e || !s) return fail(CSM_ERR_ARG, "null argument");
  if (!e->bound || !e->w.proj_table) return fail(CSM_ERR_STATE, "weights / projection table not bound");
  if (!e->ready) return fail(CSM_ERR_STATE, "no codebook-0 logits pending: call csm_prefill or csm_backbone_step first");
  if (s->topk < 1) return fail(CSM_ERR_ARG, "topk must be >= 1 (selected index k out of range)");
  if (s->topk > e->cfg.audio_vocab) return fail(CSM_ERR_ARG, "selected index k out of range (topk %d > vocab %d)", s->topk, e->cfg.audio_vocab);
  return 0;
}

extern "C" int csm_decode_frame(csm_engine_t* e, const csm_sampling_t* s) {
  if (int r = check_ready(e, s)) return r;
  if (e->h_frame >= e->cfg.max_frames) return fail(CSM_ERR_CAPACITY, "frame ring full (%d)", e->cfg.max_frames);
  LCK(launch_set_rng(e->stream, e->d_rng, s->seed, (uint64_t)(uint32_t)s->row_offset));
  LCK(decode_frame_impl(e, s));
  e->ready = false;
  return 0;
}

extern "C" int csm_backbone_step(csm_engine_t* e, const csm_sampling_t* s) {
  if (!e || !e->bound) return fail(CSM_ERR_STATE, "weights not bound");
  if (e->B < 1) return fail(CSM_ERR_STATE, "no active batch: call csm_prefill first");
  if (e->h_len >= e->cfg.max_len) return fail(CSM_ERR_CAPACITY, "KV cache full (%d positions)", e->cfg.max_len);
  LCK(backbone_step_impl(e, s, false, true, true));
  e->h_len++;
  e->h_frame++;
  e->ready = true;
  return 0;
}

extern "C" int csm_backbone_step_ids(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, int B, int advance_frame) {
  if (!e || !e->bound || !ids) return fail(CSM_ERR_STATE, "weights not bound / null ids");
  if (B < 1 || B > e->cfg.max_batch || (e->B && B != e->B)) return fail(CSM_ERR_ARG, "batch %d does not match active batch %d", B, e->B);
  if (e->h_len >= e->cfg.max_len) return fail(CSM_ERR_CAPACITY, "KV cache full (%d positions)", e->cfg.max_len);
  e->B = B;
  const int C1 = e->cfg.n_codebooks + 1;
  HIPCK(hipMemcpyAsync(e->ids_stage, ids, (size_t)B * C1 * sizeof(int64_t), hipMemcpyDeviceToDevice, e->stream));
  if (mask) HIPCK(hipMemcpyAsync(e->mask_stage, mask, (size_t)B * C1, hipMemcpyDeviceToDevice, e->stream));
  else HIPCK(hipMemsetAsync(e->mask_stage, 1, (size_t)B * C1, e->stream));
  LCK(backbone_step_impl(e, nullptr, true, advance_frame != 0, true));
  e->h_len++;
  if (advance_frame) e->h_frame++;
  e->ready = true;
  return 0;
}

static int prefill_impl(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, int B, int S, const int32_t* rope_pos,
                        float* last_h_out, float* c0_logits_out, float* all_h_out = nullptr);

extern "C" int csm_prefill(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, int B, int S, float* last_h_out,
                           float* c0_logits_out) {
  return prefill_impl(e, ids, mask, B, S, nullptr, last_h_out, c0_logits_out);
}

extern "C" int csm_prefill_pos(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, int B, int S,
                               const int32_t* position_ids, float* last_h_out, float* c0_logits_out) {
  return prefill_impl(e, ids, mask, B, S, position_ids, last_h_out, c0_logits_out);
}

static int kv_convert(csm_engine_t* e, int layer, float* K, float* V, int B, int len, int to_engine) {
  if (!e || !K || !V) return fail(CSM_ERR_ARG, "null argument");
  Stack& s = e->bb;
  if (layer < 0 || layer >= s.c.layers) return fail(CSM_ERR_ARG, "bad layer %d", layer);
  if (B < 1 || B > e->cfg.max_batch || len < 0 || len > s.lmax) return fail(CSM_ERR_CAPACITY, "batch %d / length %d exceed the engine's cache (%d, %d)", B, len, e->cfg.max_batch, s.lmax);
  KvConvArgs a{};
  a.kcache = s.kc[layer]; a.vcache = s.vc[layer]; a.k_hf = K; a.v_hf = V; a.B = B; a.n_kv = s.c.n_kv; a.hd = s.c.head_dim;
  a.lmax = s.lmax; a.len = len; a.to_engine = to_engine;
  LCK(launch_kv_convert(e->stream, e->cfg.kv_dtype, a));
  return 0;
}

extern "C" int csm_kv_export(csm_engine_t* e, int layer, float* k_out, float* v_out, int len) {
  return kv_convert(e, layer, k_out, v_out, e ? e->B : 0, len, 0);
}
extern "C" int csm_kv_import(csm_engine_t* e, int layer, const float* k_in, const float* v_in, int B, int len) {
  return kv_convert(e, layer, const_cast<float*>(k_in), const_cast<float*>(v_in), B, len, 1);
}
// after csm_kv_import of every layer: the engine continues from `len` cached positions of a batch of B
extern "C" int csm_set_length(csm_engine_t* e, int B, int len) {
  if (!e || B < 1 || B > e->cfg.max_batch || len < 0 || len > e->cfg.max_len) return fail(CSM_ERR_ARG, "bad batch / length");
  LCK(launch_set_int(e->stream, e->d_len, len));
  e->B = B; e->h_len = len; e->ready = false;
  return 0;
}

static const void* tiled_of(const csm_engine* e, const void* W) {
  const auto it = e->tiled.find(W);
  return it == e->tiled.end() ? nullptr : it->second;
}

// K splits of an MX-fp8 prefill GEMM with too few 128 x 128 tiles to fill the chip (k-steps of 128, >= 4 per split)
// K splits that let the 256 x 256 tile (gemm256.h) take a down_proj launch: enough splits for `min_wgs` workgroups, each split
// at least 2048 of K; 0 = leave the choice as it is
static inline int ksplit_256(int R, int N, int K, int cap, int min_wgs, int kstep) {
  if (min_wgs <= 0 || R % 256 || N % 256 || K < 4096) return 0;
  const long tiles = (long)(R / 256) * (N / 256);
  if (tiles >= min_wgs) return 1;
  const int ks = (int)((min_wgs + tiles - 1) / tiles);
  return (ks <= cap && K % (kstep * ks) == 0 && K / ks >= 2048) ? ks : 0;
}

static inline int mx_ksplit(int R, int N, int K, int cap) {
  const long tiles = (long)((R + 127) / 128) * (N / 128);
  if (tiles >= 384) return 1;
  int ks = (int)((512 + tiles - 1) / tiles);
  if (ks > cap) ks = cap;
  while (ks > 1 && (K % (128 * ks) || K / ks < 512)) --ks;
  return ks;
}

// GEPI_ROPE arguments of a QKV launch (gemm.h: RopeEpi) -- what launch_rope_scatter would be given
static RopeEpi rope_epi_of(csm_engine* e, Stack& s, void* kc, void* vc, int lmax, const int32_t* rope_pos) {
  RopeEpi r{};
  r.cos_tab = s.cos; r.sin_tab = s.sin; r.row_seq = e->p_row_seq; r.row_pos = e->p_row_pos; r.rope_pos = rope_pos;
  r.qbuf = e->p_q; r.kcache = kc; r.vcache = vc; r.n_q = s.c.n_q; r.n_kv = s.c.n_kv; r.lmax = lmax;
  r.kv_bf16 = e->cfg.kv_dtype == 1 ? 1 : 0; r.qscale = 1.0f / sqrtf((float)s.c.head_dim);
  return r;
}

// stack_rows with every linear on the block-scaled fp8 matrix instruction (gemm_mx.h; backbone only): the producers
// (RMSNorm, attention, SwiGLU epilogue) leave fp32 rows, mx_quant_rows_kernel turns them into e4m3 + E8M0 scales, the GEMM
// multiplies them with the MX copy of the weights.  Same residual / split-K / RoPE / attention launches as the other modes.
static int stack_rows_mx(csm_engine* e, Stack& s, void* const* kc, void* const* vc, int lmax, int B, int S, int past,
                         const int* kv_start, const int32_t* rope_pos, bool allow_split, int* pending_out, size_t* part_stride_out) {
  const size_t R = (size_t)B * S;
  const int H = s.c.hidden, nq = s.c.n_q, nkv = s.c.n_kv, hd = s.c.head_dim, F = s.c.ffn, A = nq * hd, NQKV = s.nqkv();
  if (!e->p_mx_q) {
    const size_t Kmax = std::max<size_t>(std::max(H, A), F), rows = (size_t)e->cfg.max_prefill_rows;
    if (int r = dalloc(e, &e->p_mx_q, rows * Kmax)) return r;
    if (int r = dalloc(e, &e->p_mx_s, rows * (Kmax / 32))) return r;
    if (int r = dalloc(e, &e->p_mx_q2, rows * (size_t)F)) return r;
    if (int r = dalloc(e, &e->p_mx_s2, rows * (size_t)(F / 32))) return r;
  }
  const bool can_split = allow_split && e->prefill_splitk && e->p_part && R <= 4096;
  const int cap = (int)std::min<size_t>((size_t)e->prefill_splitk_max, 4 * (size_t)e->cfg.max_prefill_rows / R);
  const int ks_o = can_split ? mx_ksplit((int)R, H, A, cap) : 1;
  int ks_d = can_split ? mx_ksplit((int)R, H, F, cap) : 1;
  if (can_split) { const int k256 = ksplit_256((int)R, H, F, cap, e->gemm_256 & 0xffffff, 128); if (k256) ks_d = k256; }
  int ks_q = 1;
  if (can_split && e->prefill_splitk_qkv) {
    const size_t room = 4 * (size_t)e->cfg.max_prefill_rows * (size_t)e->p_part_h / (R * (size_t)NQKV);
    ks_q = mx_ksplit((int)R, NQKV, H, (int)std::min<size_t>(std::min<size_t>((size_t)e->prefill_splitk_max, (size_t)e->prefill_splitk_qkv), room));
  }
  int ks_gu = 1;   // gate/up of a short prefill split over K (see stack_rows)
  if (can_split && e->prefill_splitk_gu > 1 && R <= 64 && F % 32 == 0) {   // measured: 32 / 64 rows 1.16 / 1.21 -> 1.12 / 1.18 ms, nothing at 128
    int k = std::min(e->prefill_splitk_gu, 4);
    while (k > 1 && (H % (128 * k) || H / k < 512)) --k;
    if (k > 1 && !e->p_part_gu) {
      if (int r = dalloc(e, &e->p_part_gu, (size_t)4 * std::min<size_t>(128, (size_t)e->cfg.max_prefill_rows) * (size_t)(2 * F))) return r;
    }
    ks_gu = k < 1 ? 1 : k;
  }
  const size_t part_stride = R * (size_t)H;
  auto quant = [&](const float* x, int K) {
    MxQuantArgs q{};
    q.x = x; q.ldx = K; q.rows = (int)R; q.K = K; q.q = e->p_mx_q; q.s = e->p_mx_s;
    return launch_mx_quant(e->stream, q);
  };
  auto gemm = [&](int epi, const uint8_t* Wq, const uint8_t* Ws, int N, int K, float* C, int ldc, int ks, size_t pstride) {
    GemmMxArgs g{};
    g.Aq = e->p_mx_q; g.As = e->p_mx_s; g.Wq = Wq; g.Ws = Ws; g.R = (int)R; g.N = N; g.K = K; g.C = C; g.ldc = ldc;
    g.ksplit = ks; g.Cpart = e->p_part; g.part_stride = pstride; g.big = e->gemm_256; g.skinny = e->gemm_mx_skinny;
    return launch_gemm_mx(e->stream, epi, g);
  };
  int pending = 0;
  for (int l = 0; l < s.c.layers; ++l) {
    const csm_layer_weights_t& w = s.layers[l];
    const csm_mx_layer_t& m = e->mx_layers[l];
    LCK(launch_rmsnorm(e->stream, e->p_h, H, w.ln1, (int)R, H, s.c.rms_eps, e->p_xn, H, nullptr, 0, 0, nullptr, 0,
                       pending > 0 ? e->p_part : nullptr, pending, part_stride, H, e->p_mx_q, e->p_mx_s));   // normed rows leave as MX-fp8
    pending = 0;
    RopeArgs ra{};
    bool roped = false;
    if (ks_q > 1) {
      LCK(gemm(GEPI_PARTIAL, m.qkv, m.qkv_s, NQKV, H, nullptr, 0, ks_q, R * (size_t)NQKV));
      ra.part = e->p_part; ra.nsplit = ks_q; ra.part_stride = R * (size_t)NQKV;
    } else if (e->prefill_fuse_rope && hd == 64) {   // RoPE, q scale and the cache append in the GEMM's epilogue
      GemmMxArgs g{};
      g.Aq = e->p_mx_q; g.As = e->p_mx_s; g.Wq = m.qkv; g.Ws = m.qkv_s; g.R = (int)R; g.N = NQKV; g.K = H; g.big = e->gemm_256;
      g.rope = rope_epi_of(e, s, kc[l], vc[l], lmax, rope_pos);
      LCK(launch_gemm_mx(e->stream, GEPI_ROPE, g));
      roped = true;
    } else {
      LCK(gemm(GEPI_STORE, m.qkv, m.qkv_s, NQKV, H, e->p_qkv, NQKV, 1, 0));
    }
    if (!roped) {
      ra.qkv = e->p_qkv; ra.n_q = nq; ra.n_kv = nkv; ra.hd = hd; ra.qscale = 1.0f / sqrtf((float)hd);
      ra.cos_tab = s.cos; ra.sin_tab = s.sin; ra.row_seq = e->p_row_seq; ra.row_pos = e->p_row_pos;
      ra.qbuf = e->p_q; ra.kcache = kc[l]; ra.vcache = vc[l]; ra.lmax = lmax; ra.rope_pos = rope_pos;
      LCK(launch_rope_scatter(e->stream, e->cfg.kv_dtype, (int)R, ra));
    }
    PrefillAttnArgs fa{};
    fa.kvfast = e->prefill_attn_kvfast;
    fa.q = e->p_q; fa.kcache = kc[l]; fa.vcache = vc[l]; fa.n_q = nq; fa.n_kv = nkv; fa.lmax = lmax;
    fa.S = S; fa.past = past; fa.kv_start = kv_start; fa.seq_slot = e->p_seq_slot; fa.out = e->p_att;
    // the bf16 flash kernel leaves its output already MX-quantised (a 32-block is half a head: the lane pair of a query row)
    bool att_q = e->flash_prefill && e->prefill_bf16_attn && e->prefill_fuse_quant && hd == 64;
    if (att_q) { fa.oq = e->p_mx_q; fa.os = e->p_mx_s; }
    int fr = e->flash_prefill ? launch_attn_prefill(e->stream, e->cfg.kv_dtype, B, hd, fa, e->prefill_bf16_attn ? 1 : (e->prefill_x3_attn ? 2 : 0)) : -2;
    if (fr == -2) {
      att_q = false;
      AttnArgs t{};
      t.q = e->p_q; t.kcache = kc[l]; t.vcache = vc[l]; t.n_q = nq; t.n_kv = nkv; t.hd = hd; t.lmax = lmax;
      t.row_seq = e->p_row_seq; t.row_pos = e->p_row_pos; t.kv_start = kv_start; t.nsplit = 1; t.out = e->p_att;
      fr = launch_attn(e->stream, e->cfg.kv_dtype, (int)R, t);
    }
    LCK(fr);
    if (!att_q) LCK(quant(e->p_att, A));
    if (ks_o > 1) {
      LCK(gemm(GEPI_PARTIAL, m.o, m.o_s, H, A, nullptr, 0, ks_o, part_stride));
      pending = ks_o;
    } else {
      LCK(gemm(GEPI_RESID, m.o, m.o_s, H, A, e->p_h, H, 1, 0));
    }
    LCK(launch_rmsnorm(e->stream, e->p_h, H, w.ln2, (int)R, H, s.c.rms_eps, e->p_xn, H, nullptr, 0, 0, nullptr, 0,
                       pending > 0 ? e->p_part : nullptr, pending, part_stride, H, e->p_mx_q, e->p_mx_s));
    pending = 0;
    const bool fq = e->mx_fuse_swiglu != 0;
    if (ks_gu > 1) {   // short prefill: gate/up split over K, partials summed + SwiGLU + MX quantiser in swiglu_reduce_kernel (misc.h)
      GemmMxArgs g{};
      g.Aq = e->p_mx_q; g.As = e->p_mx_s; g.Wq = m.gu; g.Ws = m.gu_s; g.R = (int)R; g.N = 2 * F; g.K = H;
      g.ksplit = ks_gu; g.Cpart = e->p_part_gu; g.part_stride = R * (size_t)(2 * F); g.big = e->gemm_256; g.skinny = e->gemm_mx_skinny;
      LCK(launch_gemm_mx(e->stream, GEPI_PARTIAL, g));
      LCK(launch_swiglu_reduce(e->stream, e->p_part_gu, ks_gu, g.part_stride, (int)R, F, e->p_act, F, nullptr, 0, fq ? e->p_mx_q2 : nullptr, fq ? e->p_mx_s2 : nullptr));
    } else {
      GemmMxArgs g{};
      g.Aq = e->p_mx_q; g.As = e->p_mx_s; g.Wq = m.gu; g.Ws = m.gu_s; g.R = (int)R; g.N = 2 * F; g.K = H; g.C = e->p_act; g.ldc = F;
      if (fq) { g.Cq = e->p_mx_q2; g.Cs = e->p_mx_s2; }
      g.big = e->gemm_256; g.skinny = e->gemm_mx_skinny;
      LCK(launch_gemm_mx(e->stream, GEPI_SWIGLU, g));
    }
    if (!fq) LCK(quant(e->p_act, F));
    {
      GemmMxArgs g{};
      g.Aq = fq ? e->p_mx_q2 : e->p_mx_q; g.As = fq ? e->p_mx_s2 : e->p_mx_s; g.Wq = m.d; g.Ws = m.d_s; g.R = (int)R; g.N = H; g.K = F;
      g.C = e->p_h; g.ldc = H; g.ksplit = ks_d; g.Cpart = e->p_part; g.part_stride = part_stride; g.big = e->gemm_256; g.skinny = e->gemm_mx_skinny;
      LCK(launch_gemm_mx(e->stream, ks_d > 1 ? GEPI_PARTIAL : GEPI_RESID, g));
      if (ks_d > 1) pending = ks_d;
    }
  }
  *pending_out = pending;
  *part_stride_out = part_stride;
  return 0;
}

// R = B * S rows (row b * S + s = position past + s of sequence b; e->p_row_seq / p_row_pos filled by the caller) through
// every layer of a stack: residual stream in e->p_h [R][hidden], K/V appended to kc[l] / vc[l] ([B][n_kv][lmax][hd]
// layouts of the engine caches).  On return *pending_out split-K partials of the LAST layer's down_proj wait in e->p_part
// (stride *part_stride_out) for the caller's next rmsnorm launch to fold in.  Used by the context prefill (backbone, the
// engine's own caches) and by the training forward's decoder pass (scratch caches, 32 positions per frame).
static int stack_rows(csm_engine* e, Stack& s, void* const* kc, void* const* vc, int lmax, int B, int S, int past,
                      const int* kv_start, const int32_t* rope_pos, bool allow_split, int* pending_out, size_t* part_stride_out) {
  const size_t R = (size_t)B * S;
  const int H = s.c.hidden, nq = s.c.n_q, nkv = s.c.n_kv, hd = s.c.head_dim, F = s.c.ffn;
  const int wd = e->cfg.weight_dtype;
  if (e->prefill_mx && &s == &e->bb && (int)e->mx_layers.size() == s.c.layers && H % 128 == 0 && F % 128 == 0 && (nq * hd) % 128 == 0 &&
      s.nqkv() % 128 == 0 && (2 * F) % 128 == 0)
    return stack_rows_mx(e, s, kc, vc, lmax, B, S, past, kv_start, rope_pos, allow_split, pending_out, part_stride_out);
  // bf16 / fp8 weights: RMSNorm, the flash attention and the SwiGLU epilogue hand their outputs to the next GEMM as
  // exact bf16 planes (split once per element instead of once per column block of the consumer)
  const bool pl = e->prefill_planes && e->p_pl_h && wd != CSM_DTYPE_F32 && H % 8 == 0 && F % 8 == 0 && (nq * hd) % 8 == 0;
  // prefill_precision = bf16: ONE plane (activations rounded to nearest bf16 by the producer), flagged by a plane
  // stride of 0; = exact: three planes, one stride apart
  const bool one = pl && e->prefill_bf16;
  // the three planes of an exact-mode operand sit a power-of-two distance apart when R K is one (2 048 x 2 048 x 2 B = 8 MiB): the three DMA
  // requests of a tile then meet in the same memory channel.  A pad between the planes (prefill_plane_pad elements) moves them apart
  const size_t ppad = (size_t)e->prefill_plane_pad;
  const size_t ps_h = one ? 0 : R * (size_t)H + ppad, ps_att = one ? 0 : R * (size_t)(nq * hd) + ppad, ps_act = one ? 0 : R * (size_t)F + ppad;
  // split-K for the residual GEMMs (o_proj, down_proj) of a small prefill: partial products go to p_part and the NEXT
  // RMSNorm launch folds them into the residual stream (fixed order: deterministic)
  const bool can_split = allow_split && pl && e->prefill_splitk && e->p_part && R <= 4096;
  // p_part holds 4 splits of max_prefill_rows rows: a shorter prefill may split further
  const int ks_cap = (int)std::min<size_t>((size_t)e->prefill_splitk_max, 4 * (size_t)e->cfg.max_prefill_rows / R);
  int ks_o = can_split ? prefill_ksplit((int)R, H, nq * hd, ks_cap) : 1, ks_d = can_split ? prefill_ksplit((int)R, H, F, ks_cap) : 1;
  if ((one || e->gemm_wide_exact) && e->gemm_wide && can_split && H % 256 == 0 && !e->tiled.empty()) {
    // one-plane activations: 128 x 256 tiles (gemm_wide_kernel) when they, times a K split that leaves each split at
    // least 16 k-steps, fill the chip; otherwise the 64 x 64 split-K choice above stands
    const long t = (long)((R + 127) / 128) * (H / 256);
    auto wide_split = [&](int K, int cur) {
      if (t >= (one ? 320 : 192)) return 1;
      const int ks = (int)((256 + t - 1) / t);
      return (ks <= 4 && K % (256 * ks) == 0 && K / ks >= 1024 && t * ks >= 256) ? ks : cur;
    };
    ks_o = wide_split(nq * hd, ks_o);
    ks_d = wide_split(F, ks_d);
  }
  if (one && can_split) { const int k256 = ksplit_256((int)R, H, F, ks_cap, e->gemm_256 & 0xffffff, 64); if (k256) ks_d = k256; }
  // QKV: the same K split, its partials summed by the RoPE / cache-append launch that reads the result anyway.  p_part is
  // free between the RMSNorm that folded the previous layer's partials and this layer's o_proj
  int ks_q = 1;
  if (can_split && e->prefill_splitk_qkv) {
    const size_t room = 4 * (size_t)e->cfg.max_prefill_rows * (size_t)e->p_part_h / (R * (size_t)s.nqkv());
    ks_q = prefill_ksplit((int)R, s.nqkv(), H, (int)std::min<size_t>(std::min<size_t>((size_t)e->prefill_splitk_max, (size_t)e->prefill_splitk_qkv), room));
  }
  // gate/up of a short prefill (<= 128 rows: 128 tiles of 128 x 128, half the chip with one k-step in flight each): the same K split,
  // as many ways as fill the chip twice and fit p_part (free between the RMSNorm that folded o_proj's partials and down_proj)
  int ks_gu = 1;
  if (can_split && e->prefill_splitk_gu > 1 && R <= (one ? 128u : 64u) && &s == &e->bb) {
    int k = std::min(e->prefill_splitk_gu, 4);
    while (k > 1 && (H % (64 * k) || H / k < 512)) --k;
    if (k > 1 && !e->p_part_gu) {
      if (int r = dalloc(e, &e->p_part_gu, (size_t)4 * std::min<size_t>(128, (size_t)e->cfg.max_prefill_rows) * (size_t)(2 * F))) return r;
    }
    ks_gu = k < 1 ? 1 : k;
  }
  const size_t part_stride = R * (size_t)H;
  int pending = 0;   // splits waiting in p_part for the next RMSNorm
  for (int l = 0; l < s.c.layers; ++l) {
    const csm_layer_weights_t& w = s.layers[l];
    LCK(launch_rmsnorm(e->stream, e->p_h, H, w.ln1, (int)R, H, s.c.rms_eps, e->p_xn, H, nullptr, 0, 0, pl ? e->p_pl_h : nullptr, ps_h,
                       pending > 0 ? e->p_part : nullptr, pending, part_stride, H));
    pending = 0;
    GemmArgs g{};
    if (pl) { g.Aplanes = e->p_pl_h; g.a_plane_stride = ps_h; }
    g.A = e->p_xn; g.lda = H; g.W = w.wqkv; g.wscale = w.sqkv; g.R = (int)R; g.N = s.nqkv(); g.K = H; g.C = e->p_qkv; g.ldc = s.nqkv();
    g.Wt = tiled_of(e, w.wqkv); g.wide = e->gemm_wide; g.wide_depth = e->gemm_wide_depth; g.wide_exact = e->gemm_wide_exact; g.dma = e->gemm_dma; g.dma_max_rows = e->gemm_dma_max_rows; g.big256 = e->gemm_256; g.dma_min_wgs = e->gemm_dma_min_wgs; g.dma_skinny = e->gemm_dma_skinny;
    RopeArgs ra{};
    bool roped = false;
    if (ks_q > 1) {
      g.ksplit = ks_q; g.Cpart = e->p_part; g.part_stride = R * (size_t)s.nqkv();
      LCK(launch_gemm(e->stream, wd, GEPI_PARTIAL, g));
      ra.part = e->p_part; ra.nsplit = ks_q; ra.part_stride = g.part_stride;
    } else {
      if (pl && e->prefill_fuse_rope && hd == 64) {   // RoPE, q scale and the cache append in the GEMM's epilogue (LDS-DMA tiles only: -2 otherwise)
        GemmArgs gr = g;
        gr.rope = rope_epi_of(e, s, kc[l], vc[l], lmax, rope_pos);
        const int rr = launch_gemm(e->stream, wd, GEPI_ROPE, gr);
        if (rr != -2) { LCK(rr); roped = true; }
      }
      if (!roped) LCK(launch_gemm(e->stream, wd, GEPI_STORE, g));
    }
    if (!roped) {
      ra.qkv = e->p_qkv; ra.n_q = nq; ra.n_kv = nkv; ra.hd = hd; ra.qscale = 1.0f / sqrtf((float)hd);
      ra.cos_tab = s.cos; ra.sin_tab = s.sin; ra.row_seq = e->p_row_seq; ra.row_pos = e->p_row_pos;
      ra.qbuf = e->p_q; ra.kcache = kc[l]; ra.vcache = vc[l]; ra.lmax = lmax; ra.rope_pos = rope_pos;
      LCK(launch_rope_scatter(e->stream, e->cfg.kv_dtype, (int)R, ra));
    }
    PrefillAttnArgs fa{};
    fa.kvfast = e->prefill_attn_kvfast;
    fa.q = e->p_q; fa.kcache = kc[l]; fa.vcache = vc[l]; fa.n_q = nq; fa.n_kv = nkv; fa.lmax = lmax;
    fa.S = S; fa.past = past; fa.kv_start = kv_start; fa.seq_slot = e->p_seq_slot; fa.out = e->p_att;
    if (pl) { fa.oplanes = e->p_pl_h; fa.plane_stride = ps_att; }
    int fr = e->flash_prefill ? launch_attn_prefill(e->stream, e->cfg.kv_dtype, B, hd, fa, (one && e->prefill_bf16_attn) ? 1 : (e->prefill_x3_attn ? 2 : 0)) : -2;
    bool att_pl = pl && fr != -2;
    if (fr == -2) {   // shapes the matrix-core kernel does not cover: one workgroup per (row, kv-head)
      AttnArgs t{};
      t.q = e->p_q; t.kcache = kc[l]; t.vcache = vc[l]; t.n_q = nq; t.n_kv = nkv; t.hd = hd; t.lmax = lmax;
      t.row_seq = e->p_row_seq; t.row_pos = e->p_row_pos; t.kv_start = kv_start; t.nsplit = 1; t.out = e->p_att;
      fr = launch_attn(e->stream, e->cfg.kv_dtype, (int)R, t);
    }
    LCK(fr);
    GemmArgs o{};
    if (att_pl) { o.Aplanes = e->p_pl_h; o.a_plane_stride = ps_att; }
    o.Wt = tiled_of(e, w.wo); o.wide = e->gemm_wide; o.wide_depth = e->gemm_wide_depth; o.wide_exact = e->gemm_wide_exact; o.dma = e->gemm_dma; o.dma_max_rows = e->gemm_dma_max_rows; o.big256 = e->gemm_256; o.dma_min_wgs = e->gemm_dma_min_wgs; o.dma_skinny = e->gemm_dma_skinny;
    o.A = e->p_att; o.lda = nq * hd; o.W = w.wo; o.wscale = w.so; o.R = (int)R; o.N = H; o.K = nq * hd; o.C = e->p_h; o.ldc = H;
    if (att_pl && ks_o > 1) {
      o.ksplit = ks_o; o.Cpart = e->p_part; o.part_stride = part_stride;
      LCK(launch_gemm(e->stream, wd, GEPI_PARTIAL, o));
      pending = ks_o;
    } else {
      LCK(launch_gemm(e->stream, wd, GEPI_RESID, o));
    }
    LCK(launch_rmsnorm(e->stream, e->p_h, H, w.ln2, (int)R, H, s.c.rms_eps, e->p_xn, H, nullptr, 0, 0, pl ? e->p_pl_h : nullptr, ps_h,
                       pending > 0 ? e->p_part : nullptr, pending, part_stride, H));
    pending = 0;
    GemmArgs gu{};
    if (pl) { gu.Aplanes = e->p_pl_h; gu.a_plane_stride = ps_h; gu.Cplanes = e->p_pl_act; gu.c_plane_stride = ps_act; }
    gu.Wt = tiled_of(e, w.wgu); gu.wide = e->gemm_wide; gu.wide_depth = e->gemm_wide_depth; gu.wide_exact = e->gemm_wide_exact; gu.dma = e->gemm_dma; gu.dma_max_rows = e->gemm_dma_max_rows; gu.big256 = e->gemm_256; gu.dma_min_wgs = e->gemm_dma_min_wgs; gu.dma_skinny = e->gemm_dma_skinny;
    gu.A = e->p_xn; gu.lda = H; gu.W = w.wgu; gu.wscale = w.sgu; gu.R = (int)R; gu.N = 2 * F; gu.K = H; gu.C = e->p_act; gu.ldc = F;
    if (ks_gu > 1) {   // short prefill: split over K, partials summed + SwiGLU by swiglu_reduce_kernel (misc.h)
      gu.ksplit = ks_gu; gu.Cpart = e->p_part_gu; gu.part_stride = R * (size_t)(2 * F);
      LCK(launch_gemm(e->stream, wd, GEPI_PARTIAL, gu));
      LCK(launch_swiglu_reduce(e->stream, e->p_part_gu, ks_gu, gu.part_stride, (int)R, F, e->p_act, F, pl ? e->p_pl_act : nullptr, ps_act, nullptr, nullptr));
    } else {
      LCK(launch_gemm(e->stream, wd, GEPI_SWIGLU, gu));
    }
    GemmArgs d{};
    if (pl) { d.Aplanes = e->p_pl_act; d.a_plane_stride = ps_act; }
    d.Wt = tiled_of(e, w.wd); d.wide = e->gemm_wide; d.wide_depth = e->gemm_wide_depth; d.wide_exact = e->gemm_wide_exact; d.dma = e->gemm_dma; d.dma_max_rows = e->gemm_dma_max_rows; d.big256 = e->gemm_256; d.dma_min_wgs = e->gemm_dma_min_wgs; d.dma_skinny = e->gemm_dma_skinny;
    d.A = e->p_act; d.lda = F; d.W = w.wd; d.wscale = w.sd; d.R = (int)R; d.N = H; d.K = F; d.C = e->p_h; d.ldc = H;
    if (pl && ks_d > 1) {
      d.ksplit = ks_d; d.Cpart = e->p_part; d.part_stride = part_stride;
      LCK(launch_gemm(e->stream, wd, GEPI_PARTIAL, d));
      pending = ks_d;
    } else {
      LCK(launch_gemm(e->stream, wd, GEPI_RESID, d));
    }
  }
  *pending_out = pending;
  *part_stride_out = part_stride;
  return 0;
}

static int prefill_impl(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, int B, int S, const int32_t* rope_pos,
                        float* last_h_out, float* c0_logits_out, float* all_h_out) {
  if (!e || !e->bound || !ids) return fail(CSM_ERR_STATE, "weights not bound / null ids");
  if (B < 1 || B > e->cfg.max_batch || S < 1) return fail(CSM_ERR_ARG, "bad batch/sequence (%d, %d)", B, S);
  if (e->B && B != e->B) return fail(CSM_ERR_ARG, "batch %d does not match active batch %d (call csm_reset)", B, e->B);
  const size_t R = (size_t)B * S;
  if (R > (size_t)e->cfg.max_prefill_rows) return fail(CSM_ERR_CAPACITY, "B*S = %zu exceeds max_prefill_rows %d", R, e->cfg.max_prefill_rows);
  if (e->h_len + S > e->cfg.max_len) return fail(CSM_ERR_CAPACITY, "context %d + %d exceeds max_len %d", e->h_len, S, e->cfg.max_len);
  e->B = B;
  Stack& s = e->bb;
  const int Hb = s.c.hidden, nq = s.c.n_q, nkv = s.c.n_kv, hd = s.c.head_dim, F = s.c.ffn;
  const int wd = e->cfg.weight_dtype;
  LCK(launch_rows_iota(e->stream, e->p_row_seq, e->p_row_pos, (int)R, S, e->h_len));
  EmbedArgs em{};
  em.text_emb = e->w.text_emb; em.audio_emb = e->w.audio_emb; em.H = Hb; em.C = e->cfg.n_codebooks; em.V = e->cfg.audio_vocab;
  em.ids = ids; em.mask = mask; em.out = e->p_h;
  LCK(launch_embed(e->stream, emb_dtype(e), (int)R, em));
  int pending = 0;
  size_t part_stride = 0;
  if (int r = stack_rows(e, s, s.kc.data(), s.vc.data(), s.lmax, B, S, e->h_len, e->d_kv_start, rope_pos, true, &pending, &part_stride)) return r;
  if (all_h_out) {   // training forward: the final-normed hidden state of EVERY row (the last layer's partials fold in here)
    LCK(launch_rmsnorm(e->stream, e->p_h, Hb, s.final_norm, (int)R, Hb, s.c.rms_eps, all_h_out, Hb, nullptr, 0, 0, nullptr, 0,
                       pending > 0 ? e->p_part : nullptr, pending, part_stride, Hb));
    pending = 0;
  }
  // last position of every sequence: rows b*S + S-1
  const float* hl = e->p_h + (size_t)(S - 1) * Hb;
  const int ldl = S * Hb;
  // (the last layer's split-K partials are folded into the last rows here, in place, before the head reads them)
  LCK(launch_rmsnorm(e->stream, hl, ldl, s.final_norm, B, Hb, s.c.rms_eps, e->last_h, Hb, nullptr, 0, 0, nullptr, 0,
                     pending > 0 ? e->p_part + (size_t)(S - 1) * Hb : nullptr, pending, part_stride, ldl));
  LCK(launch_set_int(e->stream, e->d_len, e->h_len + S));
  LCK(backbone_head(e, hl, ldl, B, false, false));
  e->h_len += S;
  e->ready = true;
  if (last_h_out) HIPCK(hipMemcpyAsync(last_h_out, e->last_h, (size_t)B * Hb * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
  if (c0_logits_out)
    HIPCK(hipMemcpy2DAsync(c0_logits_out, e->cfg.audio_vocab * sizeof(float), e->head_out + e->cfg.decoder.hidden,
                           e->ld_head * sizeof(float), e->cfg.audio_vocab * sizeof(float), B, hipMemcpyDeviceToDevice, e->stream));
  return 0;
}

extern "C" int csm_get_state(csm_engine_t* e, float* last_h_out, float* c0_logits_out) {
  if (!e || e->B < 1) return fail(CSM_ERR_STATE, "no active batch");
  const int B = e->B;
  if (last_h_out)
    HIPCK(hipMemcpyAsync(last_h_out, e->last_h, (size_t)B * e->cfg.backbone.hidden * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
  if (c0_logits_out)
    HIPCK(hipMemcpy2DAsync(c0_logits_out, e->cfg.audio_vocab * sizeof(float), e->head_out + e->cfg.decoder.hidden,
                           e->ld_head * sizeof(float), e->cfg.audio_vocab * sizeof(float), B, hipMemcpyDeviceToDevice, e->stream));
  return 0;
}

// Weight-streamer schedule of a captured frame-step: every streamed launch is cut into runs of consumer workgroups of
// about `pf_sub_kb`; run e may be fetched once the bytes of runs whose consumer has not started yet, up to and including
// e, fit the window (cyclically over consecutive frames).  Runs that could only be fetched after their own consumer has
// started are left to the consumer.
// ---- continuous batching (SURVEY f-4; no reference counterpart): a NEW utterance takes over batch row `row` of a running
// batch.  Its S context frames are prefilled right-aligned against the batch's current length L (cache positions
// L - S .. L - 1 of that row, kv_start[row] = L - S: exactly the layout of a left-padded row, which the reference positions
// the same way -- RoPE is relative, `tiny_padded`), the row's head output is replaced, its stop flag cleared; the other
// rows, the captured frame-step graphs and the frame counter are untouched.  Call between two csm_generate /
// csm_decode_frame + csm_backbone_step pairs.  The row's frames from the current frame index on belong to the new utterance.
extern "C" int csm_prefill_slot(csm_engine_t* e, int row, const int64_t* ids, const uint8_t* mask, int S) {
  if (!e || !e->bound || !ids) return fail(CSM_ERR_ARG, "null argument");
  if (!e->ready || e->B < 1) return fail(CSM_ERR_STATE, "no running batch (csm_prefill first)");
  if (row < 0 || row >= e->B) return fail(CSM_ERR_ARG, "row %d outside the running batch of %d", row, e->B);
  if (S < 1 || S > e->h_len)
    return fail(CSM_ERR_CAPACITY, "a joining context (%d frames) is longer than the batch's current length (%d): csm_shift_context first", S, e->h_len);
  Stack& s = e->bb;
  const int Hb = s.c.hidden, past0 = e->h_len - S, C1 = e->cfg.n_codebooks + 1;
  const size_t kvb = e->cfg.kv_dtype == 1 ? 2 : 4;
  const size_t slot = (size_t)s.c.n_kv * s.lmax * s.c.head_dim * kvb;
  std::vector<void*> kc(s.c.layers), vc(s.c.layers);
  for (int l = 0; l < s.c.layers; ++l) { kc[l] = (char*)s.kc[l] + (size_t)row * slot; vc[l] = (char*)s.vc[l] + (size_t)row * slot; }
  LCK(launch_set_int(e->stream, e->d_kv_start + row, past0));
  if (e->d_row_done) LCK(launch_set_int(e->stream, e->d_row_done + row, 0));
  // contexts longer than the prefill scratch go through it in chunks (causal: chunk c attends to the row's earlier chunks)
  const int chunk = e->cfg.max_prefill_rows;
  int pending = 0, n = 0;
  size_t part_stride = 0;
  for (int done = 0; done < S; done += n) {
    n = std::min(chunk, S - done);
    const int past = past0 + done;
    LCK(launch_rows_iota(e->stream, e->p_row_seq, e->p_row_pos, n, n, past));
    EmbedArgs em{};
    em.text_emb = e->w.text_emb; em.audio_emb = e->w.audio_emb; em.H = Hb; em.C = e->cfg.n_codebooks; em.V = e->cfg.audio_vocab;
    em.ids = ids + (size_t)done * C1; em.mask = mask ? mask + (size_t)done * C1 : nullptr; em.out = e->p_h;
    LCK(launch_embed(e->stream, emb_dtype(e), n, em));
    if (int r = stack_rows(e, s, kc.data(), vc.data(), s.lmax, 1, n, past, e->d_kv_start + row, nullptr, true, &pending, &part_stride)) return r;
  }
  const float* hl = e->p_h + (size_t)(n - 1) * Hb;
  LCK(launch_rmsnorm(e->stream, hl, Hb, s.final_norm, 1, Hb, s.c.rms_eps, e->last_h + (size_t)row * Hb, Hb, nullptr, 0, 0, nullptr, 0,
                     pending > 0 ? e->p_part + (size_t)(n - 1) * Hb : nullptr, pending, part_stride, Hb));
  GemvArgs a{};
  a.nt = e->nt_backbone;
  a.W = e->w.proj_head0; a.wscale = e->w.s_proj_head0; a.N = e->cfg.decoder.hidden + e->cfg.audio_vocab; a.K = Hb;
  a.x = hl; a.ldx = Hb; a.ln = s.final_norm; a.eps = s.c.rms_eps; a.out = e->head_out + (size_t)row * e->ld_head; a.ldo = e->ld_head;
  return gemv_rows(e, 1, PRO_NORM, EPI_STORE, a);
}

// Several utterances take over several rows of the running batch in ONE prefill (round 3): contexts ids / mask [n][S][C+1],
// left-padded by the caller to the common S (mask 0 on the pad frames), lens[i] = true frames of context i, rows[i] = its batch
// row.  The same result as n csm_prefill_slot calls (a left-padded row equals its solo run: kv_start hides the pad positions)
// at the cost of one short prefill instead of n -- a joining context of 64-128 frames costs ~2 ms whatever its length, and the
// whole batch waits for it.  Needs n * S <= max_prefill_rows and S <= the batch's current length (else -2: one by one).
extern "C" int csm_prefill_slots(csm_engine_t* e, const int32_t* rows, const int32_t* lens, int n, const int64_t* ids, const uint8_t* mask, int S) {
  if (!e || !e->bound || !ids || !rows || !lens) return fail(CSM_ERR_ARG, "null argument");
  if (!e->ready || e->B < 1) return fail(CSM_ERR_STATE, "no running batch (csm_prefill first)");
  if (n < 1 || n > e->B) return fail(CSM_ERR_ARG, "%d joining rows for a batch of %d", n, e->B);
  if (S < 1 || S > e->h_len || (size_t)n * S > (size_t)e->cfg.max_prefill_rows)
    return fail(CSM_ERR_CAPACITY, "joint slot prefill of %d x %d frames does not fit (batch length %d, max_prefill_rows %d)", n, S, e->h_len, e->cfg.max_prefill_rows);
  for (int i = 0; i < n; ++i) {
    if (rows[i] < 0 || rows[i] >= e->B || lens[i] < 1 || lens[i] > S) return fail(CSM_ERR_ARG, "bad row / length at %d", i);
    for (int j = 0; j < i; ++j) if (rows[j] == rows[i]) return fail(CSM_ERR_ARG, "row %d listed twice", rows[i]);
  }
  Stack& s = e->bb;
  const int Hb = s.c.hidden, past0 = e->h_len - S;
  if (!e->d_slots) { int r; if ((r = dalloc(e, &e->d_slots, (size_t)e->cfg.max_batch))) return r; }
  HIPCK(hipMemcpyAsync(e->d_slots, rows, (size_t)n * sizeof(int), hipMemcpyHostToDevice, e->stream));
  for (int i = 0; i < n; ++i) {
    LCK(launch_set_int(e->stream, e->d_kv_start + rows[i], e->h_len - lens[i]));
    if (e->d_row_done) LCK(launch_set_int(e->stream, e->d_row_done + rows[i], 0));
  }
  const int R = n * S;
  LCK(launch_rows_slots(e->stream, e->p_row_seq, e->p_row_pos, R, S, past0, e->d_slots));
  EmbedArgs em{};
  em.text_emb = e->w.text_emb; em.audio_emb = e->w.audio_emb; em.H = Hb; em.C = e->cfg.n_codebooks; em.V = e->cfg.audio_vocab;
  em.ids = ids; em.mask = mask; em.out = e->p_h;
  LCK(launch_embed(e->stream, emb_dtype(e), R, em));
  int pending = 0;
  size_t part_stride = 0;
  e->p_seq_slot = e->d_slots;
  const int sr = stack_rows(e, s, s.kc.data(), s.vc.data(), s.lmax, n, S, past0, e->d_kv_start, nullptr, true, &pending, &part_stride);
  e->p_seq_slot = nullptr;
  if (sr) return sr;
  for (int i = 0; i < n; ++i) {
    const size_t last = (size_t)i * S + (S - 1);
    const float* hl = e->p_h + last * Hb;
    LCK(launch_rmsnorm(e->stream, hl, Hb, s.final_norm, 1, Hb, s.c.rms_eps, e->last_h + (size_t)rows[i] * Hb, Hb, nullptr, 0, 0, nullptr, 0,
                       pending > 0 ? e->p_part + last * Hb : nullptr, pending, part_stride, Hb));
    GemvArgs a{};
    a.nt = e->nt_backbone;
    a.W = e->w.proj_head0; a.wscale = e->w.s_proj_head0; a.N = e->cfg.decoder.hidden + e->cfg.audio_vocab; a.K = Hb;
    a.x = hl; a.ldx = Hb; a.ln = s.final_norm; a.eps = s.c.rms_eps; a.out = e->head_out + (size_t)rows[i] * e->ld_head; a.ldo = e->ld_head;
    LCK(gemv_rows(e, 1, PRO_NORM, EPI_STORE, a));
  }
  return 0;
}

// Continuous batching, contexts longer than the running batch: every cached position of the resident batch moves `delta`
// slots up (keys re-rotated by delta, see KvShiftArgs), kv_start of every row and the shared length grow by delta.  The
// rows' attention results are unchanged up to fp32 rounding of the extra rotation (RoPE is relative); the slots below a
// row's kv_start are never read.  Call between two frame-steps, then csm_prefill_slot with a context of up to the new length.
extern "C" int csm_shift_context(csm_engine_t* e, int delta) {
  if (!e || !e->bound) return fail(CSM_ERR_ARG, "null / unbound engine");
  if (e->B < 1) return fail(CSM_ERR_STATE, "no running batch");
  if (delta < 0) return fail(CSM_ERR_ARG, "delta < 0");
  if (delta == 0) return 0;
  Stack& s = e->bb;
  const int len = e->h_len, B = e->B, nkv = s.c.n_kv, hd = s.c.head_dim;
  if (len + delta > e->cfg.max_len) return fail(CSM_ERR_CAPACITY, "length %d + shift %d exceeds max_len %d", len, delta, e->cfg.max_len);
  if (delta >= s.rope_positions) return fail(CSM_ERR_CAPACITY, "shift %d beyond the RoPE table (%d positions)", delta, s.rope_positions);
  const size_t es = e->esz_kv, cnt = (size_t)B * nkv * len * hd;
  if (len > 0) {
    // scratch for ONE layer of the resident batch, kept by the engine (a hipMalloc / hipFree pair per join otherwise)
    if (e->shift_bytes < cnt * es) {
      if (e->shift_kt) hipFree(e->shift_kt);
      if (e->shift_vt) hipFree(e->shift_vt);
      e->shift_kt = e->shift_vt = nullptr;
      e->shift_bytes = 0;
      const size_t want = cnt * es + cnt * es / 2 + 256;   // headroom: the batch keeps growing while it runs
      if (hipMalloc((void**)&e->shift_kt, want) != hipSuccess || hipMalloc((void**)&e->shift_vt, want) != hipSuccess) {
        if (e->shift_kt) hipFree(e->shift_kt);
        e->shift_kt = e->shift_vt = nullptr;
        return fail(CSM_ERR_NOMEM, "hipMalloc(%zu bytes) for the context shift failed", 2 * want);
      }
      e->shift_bytes = want - 256;
    }
    char *kt = e->shift_kt, *vt = e->shift_vt;
    for (int l = 0; l < s.c.layers; ++l) {
      KvShiftArgs a{};
      a.kcache = s.kc[l]; a.vcache = s.vc[l]; a.ktmp = kt; a.vtmp = vt; a.B = B; a.n_kv = nkv; a.hd = hd; a.lmax = s.lmax; a.len = len;
      a.cos_row = s.cos + (size_t)delta * (hd / 2); a.sin_row = s.sin + (size_t)delta * (hd / 2);
      int r = launch_kv_shift(e->stream, e->cfg.kv_dtype, a);
      hipError_t h1 = hipSuccess, h2 = hipSuccess;
      if (!r) {
        h1 = hipMemcpy2DAsync((char*)s.kc[l] + (size_t)delta * 4 * es, (size_t)s.lmax * 4 * es, kt, (size_t)len * 4 * es, (size_t)len * 4 * es,
                              (size_t)B * nkv * (hd / 4), hipMemcpyDeviceToDevice, e->stream);
        h2 = hipMemcpy2DAsync((char*)s.vc[l] + (size_t)delta * hd * es, (size_t)s.lmax * hd * es, vt, (size_t)len * hd * es, (size_t)len * hd * es,
                              (size_t)B * nkv, hipMemcpyDeviceToDevice, e->stream);
      }
      if (r || h1 != hipSuccess || h2 != hipSuccess) {
        // NOT atomic: layers 0 .. l-1 are already moved and re-rotated while the counters are not.  The resident batch is
        // unusable: drop it, so that every later call fails loudly until the caller resets and prefills again
        hipStreamSynchronize(e->stream);
        e->B = 0; e->h_len = 0; e->h_frame = 0; e->ready = false;
        return fail(r ? r : (int)(h1 != hipSuccess ? h1 : h2), "context shift failed at layer %d: the resident batch was dropped (csm_reset + csm_prefill)", l);
      }
    }
  }
  LCK(launch_add_ints(e->stream, e->d_kv_start, B, delta));
  LCK(launch_set_int(e->stream, e->d_len, len + delta));
  HIPCK(hipStreamSynchronize(e->stream));
  e->h_len = len + delta;
  return 0;
}

// ---- training forward, labels branch (reference modeling_csm.py:367-465) -------------------------------------------------
// loss = CE(codebook-0 logits of position t, label of t+1) + CE(decoder logits of codebooks 1..C-1 over the frames whose C
// audio labels are all present), both as the reference's nn.CrossEntropyLoss(ignore_index=-100) means.  FORWARD ONLY (no
// backward pass): what a caller can do with it is evaluate / monitor the reference's training objective on MI355X.
// The backbone pass is the context prefill (it fills the engine's KV cache like csm_prefill: call csm_reset first); the
// decoder pass runs the labelled frames as sequences of C positions through the same prefill kernels on scratch caches.
constexpr int LOSS_FRAMES = 256;   // labelled frames per decoder pass
extern "C" int csm_forward_loss(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, const int64_t* labels, int B, int S,
                                float* out3, float* last_h_out, float* c0_logits_out) {
  if (!e || !e->bound || !ids || !labels || !out3) return fail(CSM_ERR_ARG, "null argument");
  if (e->h_len != 0) return fail(CSM_ERR_STATE, "csm_forward_loss starts from an empty cache (csm_reset first)");
  if (!e->w.proj_table) return fail(CSM_ERR_STATE, "projection table not built");
  const int C = e->cfg.n_codebooks, V = e->cfg.audio_vocab, Hb = e->cfg.backbone.hidden, Hd = e->cfg.decoder.hidden;
  const size_t R = (size_t)B * S;
  const int P = C;   // decoder positions per frame that matter: 0 .. C-1 (the reference also runs position C, which nothing reads)
  if (P > e->dec.lmax || P > e->dec.rope_positions) return fail(CSM_ERR_CAPACITY, "decoder cache shorter than %d positions", P);
  const int FC = std::min<int>(LOSS_FRAMES, e->cfg.max_prefill_rows / P);
  if (FC < 1) return fail(CSM_ERR_CAPACITY, "max_prefill_rows %d is below one decoder frame", e->cfg.max_prefill_rows);
  int r;
  // ---- scratch (first call) ----
  if (!e->loss_head) {
    const size_t maxR = (size_t)e->cfg.max_prefill_rows;
    const size_t kvb = e->cfg.kv_dtype == 1 ? 2 : 4;
    const size_t per = (size_t)FC * e->dec.c.n_kv * e->dec.lmax * e->dec.c.head_dim * kvb;
    if ((r = dalloc(e, &e->loss_head, maxR * e->ld_head))) return r;
    e->loss_head_rows = maxR;
    for (int l = 0; l < e->dec.c.layers; ++l) {
      char *k = nullptr, *v = nullptr;
      if ((r = dalloc(e, &k, per)) || (r = dalloc(e, &v, per))) return r;
      HIPCK(hipMemsetAsync(k, 0, per, e->stream));
      HIPCK(hipMemsetAsync(v, 0, per, e->stream));
      e->loss_kc.push_back(k); e->loss_vc.push_back(v);
    }
    e->loss_rows_n = std::max(maxR, (size_t)FC);
    e->loss_lab_n = maxR + (size_t)(C - 1) * FC;
    if ((r = dalloc(e, &e->loss_logits, (size_t)FC * ((V + 3) & ~3))) || (r = dalloc(e, &e->loss_rows, e->loss_rows_n)) ||
        (r = dalloc(e, &e->loss_lab, e->loss_lab_n)) || (r = dalloc(e, &e->loss_idx, (size_t)2 * FC)) || (r = dalloc(e, &e->loss_acc, (size_t)4)))
      return r;
  }
  if (R > e->loss_head_rows) return fail(CSM_ERR_CAPACITY, "B*S = %zu exceeds max_prefill_rows", R);
  // ---- 1. backbone over the context; final-normed state of EVERY position ----
  if ((r = prefill_impl(e, ids, mask, B, S, nullptr, last_h_out, c0_logits_out, e->p_xn))) return r;
  // ---- 2. [projection ; codebook0_head] on every position (modeling_csm.py:361 and :407) ----
  {
    GemvArgs a{};
    a.nt = 0;
    a.W = e->w.proj_head0; a.wscale = e->w.s_proj_head0; a.N = Hd + V; a.K = Hb; a.x = e->p_xn; a.ldx = Hb;
    a.out = e->loss_head; a.ldo = e->ld_head;
    if ((r = gemv_rows(e, (int)R, PRO_PLAIN, EPI_STORE, a))) return r;
  }
  // ---- 3. labels on the host: shifted codebook-0 targets, the list of fully labelled frames ----
  std::vector<int64_t> lab(R * (size_t)(C + 1));
  HIPCK(hipMemcpyAsync(lab.data(), labels, lab.size() * sizeof(int64_t), hipMemcpyDeviceToHost, e->stream));
  HIPCK(hipStreamSynchronize(e->stream));
  std::vector<int> lab0(R);
  std::vector<int> fr_prev, fr_row;
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < S; ++t) {
      const size_t row = (size_t)b * S + t;
      const int64_t nx = t + 1 < S ? lab[(row + 1) * (C + 1)] : -100;
      lab0[row] = (nx >= 0 && nx < V) ? (int)nx : -100;
      bool all = true;
      for (int c = 0; c < C; ++c) all = all && lab[row * (C + 1) + c] != -100;
      if (all) { fr_prev.push_back((int)((size_t)b * S + (t + S - 1) % S)); fr_row.push_back((int)row); }
    }
  HIPCK(hipMemsetAsync(e->loss_acc, 0, 4 * sizeof(double), e->stream));
  HIPCK(hipMemcpyAsync(e->loss_lab, lab0.data(), R * sizeof(int), hipMemcpyHostToDevice, e->stream));
  {
    CeArgs ce{};
    ce.logits = e->loss_head + Hd; ce.ld = e->ld_head; ce.V = V; ce.rows = (int)R; ce.labels = e->loss_lab; ce.row_loss = e->loss_rows;
    LCK(launch_ce_rows(e->stream, ce));
    LCK(launch_ce_reduce(e->stream, e->loss_rows, e->loss_lab, V, (int)R, e->loss_acc));
  }
  HIPCK(hipStreamSynchronize(e->stream));   // lab0 is re-used below
  // ---- 4. decoder over the labelled frames, LOSS_FRAMES at a time ----
  const int nf = (int)fr_row.size();
  const int ldl = (V + 3) & ~3;
  std::vector<int> idx(2 * (size_t)FC), labc((size_t)(C - 1) * FC);
  for (int f0 = 0; f0 < nf; f0 += FC) {
    const int fc = std::min(FC, nf - f0);
    for (int i = 0; i < fc; ++i) { idx[i] = fr_prev[f0 + i]; idx[FC + i] = fr_row[f0 + i]; }
    for (int c = 1; c < C; ++c)
      for (int i = 0; i < fc; ++i) {
        const int64_t v = lab[(size_t)fr_row[f0 + i] * (C + 1) + c];
        labc[(size_t)(c - 1) * FC + i] = (v >= 0 && v < V) ? (int)v : -100;
      }
    HIPCK(hipMemcpyAsync(e->loss_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCK(hipMemcpyAsync(e->loss_lab, labc.data(), labc.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    DecInArgs di{};
    di.head_rows = e->loss_head; di.ld_head = e->ld_head; di.Hd = Hd; di.P = P; di.prev_row = e->loss_idx; di.tok_row = e->loss_idx + FC;
    di.ids = ids; di.C = C; di.V = V; di.proj_table = e->w.proj_table; di.out = e->p_h;
    LCK(launch_dec_input(e->stream, fc, di));
    LCK(launch_rows_iota(e->stream, e->p_row_seq, e->p_row_pos, fc * P, P, 0));
    int pending = 0;
    size_t part_stride = 0;
    if ((r = stack_rows(e, e->dec, e->loss_kc.data(), e->loss_vc.data(), e->dec.lmax, fc, P, 0, nullptr, nullptr, false, &pending, &part_stride)))
      return r;
    for (int c = 1; c < C; ++c) {   // codebook c from position c of every frame (modeling_csm.py:447-463)
      GemvArgs a{};
      a.nt = 0;
      a.W = (const char*)e->w.audio_head_t + (size_t)(c - 1) * V * Hd * w_esz(e);
      a.wscale = e->w.s_audio_head ? e->w.s_audio_head + (size_t)(c - 1) * V : nullptr;
      a.N = V; a.K = Hd; a.x = e->p_h + (size_t)c * Hd; a.ldx = P * Hd; a.ln = e->dec.final_norm; a.eps = e->dec.c.rms_eps;
      a.out = e->loss_logits; a.ldo = ldl;
      if ((r = gemv_rows(e, fc, PRO_NORM, EPI_STORE, a))) return r;
      CeArgs ce{};
      ce.logits = e->loss_logits; ce.ld = ldl; ce.V = V; ce.rows = fc; ce.labels = e->loss_lab + (size_t)(c - 1) * FC; ce.row_loss = e->loss_rows;
      LCK(launch_ce_rows(e->stream, ce));
      LCK(launch_ce_reduce(e->stream, e->loss_rows, ce.labels, V, fc, e->loss_acc + 2));
    }
    HIPCK(hipStreamSynchronize(e->stream));   // idx / labc are overwritten by the next chunk
  }
  LCK(launch_loss_finalize(e->stream, e->loss_acc, nf, out3));
  return 0;
}

static int build_pf_schedule(csm_engine* e, const std::vector<PfGeom>& geoms, GraphEntry& ent) {
  ent.n_launch = (int)geoms.size();
  if (geoms.empty()) return 0;
  for (const PfGeom& g : geoms)
    if (g.exclusive) return 0;   // a launch that needs whole CUs to itself: the streamer would hold the chain up (gemm16.h)
  std::vector<PfSeg> segs;
  std::vector<size_t> bytes;
  const size_t sub = (size_t)e->pf_sub_kb << 10;
  for (int li = 0; li < (int)geoms.size(); ++li) {
    const PfGeom& g = geoms[li];
    if (g.kind < 0 || !g.W || g.grid < 1 || g.tpb < 1) continue;
    size_t rb = (size_t)g.K * g.esz;
    size_t per_block = (size_t)g.iters * 2 * g.tpb * rb;
    if (g.kind == 2) {   // fragment-order copy: bytes of one (tile, chunk) block; a workgroup reads tpb x iters of them
      rb = (size_t)2048 * g.esz;
      per_block = (size_t)g.tpb * g.iters * rb;
    }
    int bps = (int)((sub + per_block - 1) / per_block);
    bps = (bps + 7) & ~7;
    if (bps < 8) bps = 8;
    const size_t total = g.kind == 2 ? (size_t)g.grid * per_block : (size_t)g.N * rb;
    size_t limit = total;
    if (e->pf_max_kb > 0 && total > ((size_t)e->pf_max_kb << 10)) limit = (size_t)e->pf_part_kb << 10;
    size_t done = 0;
    for (int b0 = 0; b0 < g.grid; b0 += bps) {
      if (done >= limit) break;
      done += (size_t)(std::min(g.grid, b0 + bps) - b0) * per_block;
      PfSeg sg{};
      sg.W = (const char*)g.W; sg.row_bytes = (uint32_t)rb; sg.N = g.N; sg.kind = g.kind; sg.tpb = g.tpb; sg.iters = g.iters;
      sg.stride = g.stride; sg.ntask = g.ntask; sg.hd = g.hd > 1 ? g.hd : 2; sg.n_rope_heads = g.n_rope_heads;
      sg.b0 = b0; sg.b1 = std::min(g.grid, b0 + bps); sg.owner = li;
      segs.push_back(sg);
      bytes.push_back((size_t)(sg.b1 - sg.b0) * per_block);
      ent.step_bytes += bytes.back();
    }
  }
  const int n = (int)segs.size();
  if (n == 0) return 0;
  const size_t window = (size_t)e->pf_window_mb << 20;
  std::vector<PfSeg> keep;
  for (int i = 0; i < n; ++i) {
    size_t acc = bytes[i];
    int j = i - 1, wrap = 0;   // walk back, cyclically into the previous frame
    int need = 0;
    bool all = false;
    for (int steps = 0; steps < n - 1; ++steps) {
      int jj = j;
      if (jj < 0) { jj += n; wrap = 1; } else wrap = 0;
      if (acc + bytes[jj] > window) { need = segs[jj].owner + 1 - e->pf_lead - (wrap || j < 0 ? ent.n_launch : 0); break; }
      acc += bytes[jj];
      --j;
      if (steps == n - 2) all = true;
    }
    if (all || n == 1) need = -ent.n_launch;   // a whole frame-step fits the window: fetched while the PREVIOUS frame-step runs (round 6: was "at once",
                                               // i.e. every replay of a small model was fetched before the first launch and the loaders outlived the chain)
    // a run that could only be fetched once its own consumer has started is left to the consumer; co-fetch keeps the
    // runs that become fetchable exactly when their consumer starts (the streamer then reads beside the consumer)
    if (need > segs[i].owner || (need == segs[i].owner && !e->pf_cofetch)) continue;
    segs[i].need = need;
    keep.push_back(segs[i]);
    ent.sched_bytes += bytes[i];
  }
  if (keep.empty()) return 0;
  void* d = nullptr;
  hipError_t r = hipMalloc(&d, keep.size() * sizeof(PfSeg));
  if (r != hipSuccess) return fail(CSM_ERR_NOMEM, "hipMalloc for the weight-streamer schedule failed: %s", hipGetErrorString(r));
  HIPCK(hipMemcpy(d, keep.data(), keep.size() * sizeof(PfSeg), hipMemcpyHostToDevice));
  ent.d_segs = (PfSeg*)d;
  ent.n_segs = (int)keep.size();
  return 0;
}

extern "C" int csm_generate(csm_engine_t* e, const csm_sampling_t* s, int n_frames, int use_graph) {
  if (int r = check_ready(e, s)) return r;
  if (n_frames < 0) return fail(CSM_ERR_ARG, "n_frames < 0");
  if (e->h_frame + n_frames > e->cfg.max_frames) return fail(CSM_ERR_CAPACITY, "frame ring too small: %d + %d > %d", e->h_frame, n_frames, e->cfg.max_frames);
  if (e->h_len + n_frames > e->cfg.max_len) return fail(CSM_ERR_CAPACITY, "KV cache too small: %d + %d > %d", e->h_len, n_frames, e->cfg.max_len);
  const bool want_h = s->last_h_trace != nullptr;
  LCK(launch_set_rng(e->stream, e->d_rng, s->seed, (uint64_t)(uint32_t)s->row_offset));
  HIPCK(hipEventRecord(e->ev0, e->stream));
  if (use_graph && n_frames > 0) {
    GraphKey k{};
    k.B = e->B; k.topk = s->topk; k.nsplit = e->nsplit_eff(); k.temperature = s->temperature; k.per_row = s->per_row_stop != 0;   // exactly what the captured sampler launches use (SampleArgs::row_done), also at B = 1
    k.noise = s->noise; k.forced = s->forced; k.ltrace = s->logits_trace; k.htrace = s->last_h_trace;
    const bool greedy = s->topk <= 1 || s->temperature == 0.f;
    if (greedy) { k.topk = 1; k.temperature = 0.f; }   // every greedy setting runs the same launches
    auto it = e->graphs.find(k);
    if (it == e->graphs.end()) {
      if (e->graphs.size() >= MAX_GRAPHS) {   // evict the least recently used graph
        auto lru = e->graphs.begin();
        for (auto jt = e->graphs.begin(); jt != e->graphs.end(); ++jt)
          if (jt->second.last_use < lru->second.last_use) lru = jt;
        HIPCK(hipStreamSynchronize(e->stream));
        hipGraphExecDestroy(lru->second.exec);
        if (lru->second.d_segs) hipFree(lru->second.d_segs);
        e->graphs.erase(lru);
      }
      hipGraph_t g = nullptr;
      std::vector<PfGeom> geoms;
      HIPCK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
      e->pf_rec = &geoms;
      e->tl_n = 0;
      int r = decode_frame_impl(e, s);
      if (!r) r = backbone_step_impl(e, s, false, true, want_h);
      e->pf_rec = nullptr;
      hipError_t ce = hipStreamEndCapture(e->stream, &g);
      if (r) {
        if (g) hipGraphDestroy(g);
        return r;
      }
      HIPCK(ce);
      hipGraphExec_t ge = nullptr;
      HIPCK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      HIPCK(hipGraphDestroy(g));
      hipGraphUpload(ge, e->stream);   // the first launch then starts like every later one (the weight streamer waits for it)
      GraphEntry ent{};
      ent.exec = ge;
      e->last_geoms = geoms;
      if (int br = build_pf_schedule(e, geoms, ent)) return br;
      it = e->graphs.emplace(k, ent).first;
      e->graphs_captured++;
    }
    it->second.last_use = ++e->graph_tick;
    if (int hr = pf_harvest(e, false)) return hr;   // status of the previous streamer launch, if it has completed: may switch the streamer off
    const bool stream_weights = e->pf_enable && !e->pf_disabled && e->pf_rot >= 0 && it->second.n_segs > 0;
    e->pf_last_frames = n_frames;
    e->pf_last[0] = it->second.n_segs; e->pf_last[1] = it->second.n_launch;
    e->pf_last[2] = (long long)it->second.sched_bytes; e->pf_last[3] = (long long)it->second.step_bytes;
    if (stream_weights) {   // the streamer runs beside the replays on stream2, paced by the launch counter
      HIPCK(hipMemsetAsync(e->d_prog, 0, sizeof(unsigned), e->stream));
      HIPCK(hipMemsetAsync(e->d_pf_misc, 0, 16 * sizeof(unsigned), e->stream));
      hipStream_t pst = e->pf_force_serial ? e->stream : e->stream2;
      HIPCK(hipEventRecord(e->ev_fork, e->stream));
      if (pst != e->stream) HIPCK(hipStreamWaitEvent(pst, e->ev_fork, 0));
      PfArgs pa{};
      pa.segs = it->second.d_segs; pa.n = it->second.n_segs; pa.n_launch = it->second.n_launch; pa.reps = n_frames;
      pa.rot = e->pf_rot; pa.prog = e->d_prog; pa.ticket = e->d_pf_misc; pa.status = e->d_pf_misc + 8; pa.lifetime = e->d_pf_misc + 40;
      pa.budget_ticks = (long long)e->pf_budget_us * 100;   // default 20 ms without a launch starting: give up (s_memrealtime runs at 100 MHz)
      pa.skip_late = e->pf_skip_late; pa.poll_sleep = e->pf_poll_sleep; pa.depth = e->pf_depth; pa.seg_sleep = e->pf_seg_sleep; pa.stride = e->pf_stride;
      LCK(launch_weight_prefetch(pst, e->pf_grid, pa));
      HIPCK(hipMemcpyAsync(e->h_pf, e->d_pf_misc + 8, 40 * sizeof(unsigned), hipMemcpyDeviceToHost, pst));
      HIPCK(hipEventRecord(e->ev_join, pst));
      e->pf_pending++;
    }
    for (int i = 0; i < n_frames; ++i) HIPCK(hipGraphLaunch(it->second.exec, e->stream));
    if (stream_weights) {
      HIPCK(hipEventRecord(e->ev1, e->stream));
      if (!e->pf_force_serial) HIPCK(hipStreamWaitEvent(e->stream, e->ev_join, 0));   // later work on the engine stream sees the streamer finished
      e->h_len += n_frames;
      e->h_frame += n_frames;
      return 0;
    }
  } else {
    for (int i = 0; i < n_frames; ++i) {
      LCK(decode_frame_impl(e, s));
      LCK(backbone_step_impl(e, s, false, true, want_h));
    }
  }
  HIPCK(hipEventRecord(e->ev1, e->stream));
  e->h_len += n_frames;
  e->h_frame += n_frames;
  return 0;
}

extern "C" int csm_last_generate_ms(csm_engine_t* e, float* ms_host) {
  if (!e || !ms_host) return fail(CSM_ERR_ARG, "null argument");
  HIPCK(hipEventSynchronize(e->ev1));
  HIPCK(hipEventElapsedTime(ms_host, e->ev0, e->ev1));
  return 0;
}

extern "C" int csm_read_frames(csm_engine_t* e, int64_t* frames_out, int first, int n) {
  if (!e || !frames_out || first < 0 || n < 0 || first + n > e->cfg.max_frames) return fail(CSM_ERR_ARG, "bad frame range");
  const int C = e->cfg.n_codebooks;
  if (n == 0) return 0;
  HIPCK(hipMemcpy2DAsync(frames_out, (size_t)n * C * sizeof(int64_t), e->ring + (size_t)first * C,
                         (size_t)e->cfg.max_frames * C * sizeof(int64_t), (size_t)n * C * sizeof(int64_t), e->B,
                         hipMemcpyDeviceToDevice, e->stream));
  return 0;
}

extern "C" int csm_read_zero_counts(csm_engine_t* e, int32_t* out_host, int first, int n) {
  if (!e || !out_host || first < 0 || n < 0 || first + n > e->cfg.max_frames) return fail(CSM_ERR_ARG, "bad frame range");
  if (n == 0) return 0;
  HIPCK(hipMemcpyAsync(out_host, e->d_zero_count + first, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int csm_rewind_frames(csm_engine_t* e) {
  if (!e) return fail(CSM_ERR_ARG, "null engine");
  LCK(launch_set_int(e->stream, e->d_frame, 0));
  HIPCK(hipMemsetAsync(e->d_zero_count, 0, (size_t)e->cfg.max_frames * sizeof(int), e->stream));
  e->h_frame = 0;
  return 0;
}

// debug probe: streamed launch i of the next captured frame-step writes {XCD id, clocks until its weights were consumed}
// per workgroup into buf[i][2048][2] (uint32); pass NULL to stop.  Drops the cached graphs.
extern "C" int csm_set_debug_buffer(csm_engine_t* e, uint32_t* buf, int n_launches) {
  if (!e) return fail(CSM_ERR_ARG, "null engine");
  e->dbg_buf = buf;
  e->dbg_launches = buf ? n_launches : 0;
  drop_graphs(e);
  return 0;
}

// geometry of the streamed launches of the last captured frame-step: out[i] = {N, K, grid, tasks per workgroup, kind}
extern "C" int csm_last_geoms(csm_engine_t* e, int32_t* out_host, int max_launches, int* n_host) {
  if (!e || !out_host || !n_host) return fail(CSM_ERR_ARG, "null argument");
  const int n = std::min((int)e->last_geoms.size(), max_launches);
  for (int i = 0; i < n; ++i) {
    const PfGeom& g = e->last_geoms[i];
    int32_t* o = out_host + 5 * i;
    o[0] = g.N; o[1] = g.K; o[2] = g.grid; o[3] = g.tpb; o[4] = g.kind;
  }
  *n_host = n;
  return 0;
}

extern "C" int csm_prefetch_stats(csm_engine_t* e, long long* out10_host) {
  if (!e || !out10_host) return fail(CSM_ERR_ARG, "null argument");
  HIPCK(hipStreamSynchronize(e->stream));
  HIPCK(hipStreamSynchronize(e->stream2));
  unsigned st[4], prog = 0;
  HIPCK(hipMemcpy(st, e->d_pf_misc + 8, sizeof(st), hipMemcpyDeviceToHost));
  HIPCK(hipMemcpy(&prog, e->d_prog, sizeof(prog), hipMemcpyDeviceToHost));
  out10_host[0] = st[0]; out10_host[1] = st[1]; out10_host[2] = st[2]; out10_host[3] = e->pf_rot;
  for (int i = 0; i < 4; ++i) out10_host[4 + i] = e->pf_last[i];
  out10_host[8] = prog;            // launches counted since the last csm_generate began
  out10_host[9] = e->pf_last_frames;
  unsigned dbg[4];
  HIPCK(hipMemcpy(dbg, e->d_pf_misc + 12, sizeof(dbg), hipMemcpyDeviceToHost));
  if (int hr = pf_harvest(e, true)) return hr;
  snprintf(g_err, sizeof(g_err), "streamer %s; dispatch-rate probe %.2f us per empty launch alone, %.2f beside a resident kernel; stop record of the last give-up: segment %u want %u seen %u rep %u; retired by the end-of-chain rule %u",
           pf_reason(e->pf_disabled), e->pf_rate_base_us, e->pf_rate_beside_us, dbg[0], dbg[1], dbg[2], dbg[3], st[3]);
  return 0;
}

// Health of the weight streamer (round 6).  out8 = {disabled reason (0 on, 1 streams share a hardware queue, 2 repeated give-ups, 3 dispatch
// not round-robin, 4 creation-time probe failed), strikes, lifetime give-ups (workgroups), lifetime finished (workgroups), streamer
// launches, budget in us, concurrency-probe runs, streamer launches not yet accounted for}.  Waits for the last streamer launch only.
extern "C" int csm_prefetch_health(csm_engine_t* e, long long* out8_host) {
  if (!e || !out8_host) return fail(CSM_ERR_ARG, "null argument");
  if (int hr = pf_harvest(e, true)) return hr;
  out8_host[0] = e->pf_disabled; out8_host[1] = e->pf_strikes;
  out8_host[2] = e->h_pf ? e->h_pf[32] : 0; out8_host[3] = e->h_pf ? e->h_pf[33] : 0; out8_host[4] = e->h_pf ? e->h_pf[34] : 0;
  out8_host[5] = e->pf_budget_us; out8_host[6] = e->pf_probe_runs; out8_host[7] = e->pf_pending;
  snprintf(g_err, sizeof(g_err), "%s", pf_reason(e->pf_disabled));
  return 0;
}

extern "C" int csm_graph_stats(csm_engine_t* e, int* captured_total_host, int* cached_host) {
  if (!e) return fail(CSM_ERR_ARG, "null engine");
  if (captured_total_host) *captured_total_host = e->graphs_captured;
  if (cached_host) *cached_host = (int)e->graphs.size();
  return 0;
}

// Re-home a live context into a larger engine (same model, same device).  K rows are [lmax][4] strips per
// (sequence, kv-head, 4-dim group), V rows are [lmax][hd] per (sequence, kv-head): 2-D copies with the two pitches.
extern "C" int csm_kv_copy(csm_engine_t* dst, csm_engine_t* src) {
  if (!dst || !src || dst == src) return fail(CSM_ERR_ARG, "bad engines");
  if (dst->device != src->device || dst->esz_kv != src->esz_kv || dst->cfg.n_codebooks != src->cfg.n_codebooks ||
      memcmp(&dst->cfg.backbone, &src->cfg.backbone, sizeof(csm_llama_cfg_t)) || memcmp(&dst->cfg.decoder, &src->cfg.decoder, sizeof(csm_llama_cfg_t)))
    return fail(CSM_ERR_ARG, "engines differ in model shape / KV dtype / device");
  const int B = src->B, len = src->h_len, nf = src->h_frame, C = src->cfg.n_codebooks;
  if (B > dst->cfg.max_batch || len > dst->cfg.max_len || nf > dst->cfg.max_frames)
    return fail(CSM_ERR_CAPACITY, "destination engine too small (batch %d, length %d, frames %d)", B, len, nf);
  HIPCK(hipStreamSynchronize(src->stream));
  hipStream_t st = dst->stream;
  const size_t es = src->esz_kv;
  if (B > 0 && len > 0) {
    Stack& a = src->bb; Stack& b = dst->bb;
    const int nkv = a.c.n_kv, hd = a.c.head_dim;
    for (int l = 0; l < a.c.layers; ++l) {
      HIPCK(hipMemcpy2DAsync(b.kc[l], (size_t)b.lmax * 4 * es, a.kc[l], (size_t)a.lmax * 4 * es, (size_t)len * 4 * es,
                             (size_t)B * nkv * (hd / 4), hipMemcpyDeviceToDevice, st));
      HIPCK(hipMemcpy2DAsync(b.vc[l], (size_t)b.lmax * hd * es, a.vc[l], (size_t)a.lmax * hd * es, (size_t)len * hd * es,
                             (size_t)B * nkv, hipMemcpyDeviceToDevice, st));
    }
  }
  if (B > 0) {
    if (nf > 0)
      HIPCK(hipMemcpy2DAsync(dst->ring, (size_t)dst->cfg.max_frames * C * sizeof(int64_t), src->ring,
                             (size_t)src->cfg.max_frames * C * sizeof(int64_t), (size_t)nf * C * sizeof(int64_t), B,
                             hipMemcpyDeviceToDevice, st));
    HIPCK(hipMemcpyAsync(dst->d_kv_start, src->d_kv_start, B * sizeof(int), hipMemcpyDeviceToDevice, st));
    HIPCK(hipMemcpyAsync(dst->d_row_done, src->d_row_done, B * sizeof(int), hipMemcpyDeviceToDevice, st));
    if (nf > 0) HIPCK(hipMemcpyAsync(dst->d_zero_count, src->d_zero_count, (size_t)nf * sizeof(int), hipMemcpyDeviceToDevice, st));
    HIPCK(hipMemcpyAsync(dst->head_out, src->head_out, (size_t)B * src->ld_head * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCK(hipMemcpyAsync(dst->last_h, src->last_h, (size_t)B * src->cfg.backbone.hidden * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  LCK(launch_set_int(st, dst->d_len, len));
  LCK(launch_set_int(st, dst->d_frame, nf));
  dst->B = B; dst->h_len = len; dst->h_frame = nf; dst->ready = src->ready;
  HIPCK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int csm_frames_done(csm_engine_t* e, int* n_host) {
  if (!e || !n_host) return fail(CSM_ERR_ARG, "null argument");
  HIPCK(hipMemcpyAsync(n_host, e->d_frame, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCK(hipStreamSynchronize(e->stream));
  return 0;
}
extern "C" int csm_cur_len(csm_engine_t* e, int* len_host) {
  if (!e || !len_host) return fail(CSM_ERR_ARG, "null argument");
  HIPCK(hipMemcpyAsync(len_host, e->d_len, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCK(hipStreamSynchronize(e->stream));
  return 0;
}
extern "C" int csm_sync(csm_engine_t* e) {
  if (!e) return fail(CSM_ERR_ARG, "null engine");
  HIPCK(hipStreamSynchronize(e->stream));
  return 0;
}

// proj_table[r,:] = projection.weight @ audio_emb[r,:], fp32, through the prefill GEMM in row chunks
extern "C" int csm_build_proj_table(csm_engine_t* e, float* out) {
  if (!e || !e->bound || !out) return fail(CSM_ERR_STATE, "weights not bound / null output");
  const int Hb = e->cfg.backbone.hidden, Hd = e->cfg.decoder.hidden;
  const size_t rows = (size_t)e->cfg.n_codebooks * e->cfg.audio_vocab;
  const size_t esz = emb_dtype(e) == CSM_DTYPE_BF16 ? 2 : 4;
  const size_t chunk = e->cfg.max_prefill_rows;
  // projection.weight = first Hd rows of proj_head0
  for (size_t r0 = 0; r0 < rows; r0 += chunk) {
    const size_t n = rows - r0 < chunk ? rows - r0 : chunk;
    LCK(launch_widen(e->stream, emb_dtype(e), (const char*)e->w.audio_emb + r0 * Hb * esz, e->p_xn, n * Hb));
    GemmArgs g{};
    g.A = e->p_xn; g.lda = Hb; g.W = e->w.proj_head0; g.wscale = e->w.s_proj_head0; g.R = (int)n; g.N = Hd; g.K = Hb; g.C = out + r0 * Hd; g.ldc = Hd;
    LCK(launch_gemm(e->stream, e->cfg.weight_dtype, GEPI_STORE, g));
  }
  HIPCK(hipStreamSynchronize(e->stream));
  e->w.proj_table = out;
  return 0;
}

// ---- per-kernel entry points (unit parity) -----------------------------------------------------------
extern "C" int csm_embed_sum(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, int rows, float* out) {
  if (!e || !e->bound) return fail(CSM_ERR_STATE, "weights not bound");
  EmbedArgs em{};
  em.text_emb = e->w.text_emb; em.audio_emb = e->w.audio_emb; em.H = e->cfg.backbone.hidden; em.C = e->cfg.n_codebooks;
  em.V = e->cfg.audio_vocab; em.ids = ids; em.mask = mask; em.out = out;
  LCK(launch_embed(e->stream, emb_dtype(e), rows, em));
  return 0;
}

extern "C" int csm_rmsnorm(csm_engine_t* e, const float* x, const float* w, int rows, int hidden, float eps, float* out) {
  if (!e) return fail(CSM_ERR_ARG, "null engine");
  LCK(launch_rmsnorm(e->stream, x, hidden, w, rows, hidden, eps, out, hidden, nullptr, 0, 0));
  return 0;
}

extern "C" int csm_gemv(csm_engine_t* e, const void* W, int wdtype, const float* wscale, int N, int K, const float* x,
                        int M, const float* ln, float eps, float* y) {
  if (!e || M < 1 || M > 256) return fail(CSM_ERR_ARG, "bad gemv arguments");
  GemvArgs a{};
  a.W = W; a.wscale = wscale; a.N = N; a.K = K; a.x = x; a.ldx = K; a.ln = ln; a.eps = eps; a.out = y; a.ldo = N;
  const int save = e->cfg.weight_dtype;
  e->cfg.weight_dtype = wdtype;
  // batched rows take the same route as bound weights: a (temporary) fragment-order copy for the MFMA kernel
  bool tmp_tile = false;
  if (M >= 2 && e->tile_weights && e->use_mfma && (wdtype == CSM_DTYPE_BF16 || wdtype == CSM_DTYPE_FP8) && K % 128 == 0 &&
      !e->tiled.count(W)) {
    if (int tr = tile_one(e, W, N, K)) { e->cfg.weight_dtype = save; return tr; }
    tmp_tile = true;
  }
  int r = gemv_rows(e, M, ln ? PRO_NORM : PRO_PLAIN, EPI_STORE, a);
  e->cfg.weight_dtype = save;
  if (tmp_tile) {
    hipStreamSynchronize(e->stream);
    e->tiled.erase(W);
    hipFree(e->tiled_allocs.back());
    e->tiled_allocs.pop_back();
  }
  if (r) return fail(r > 0 ? r : CSM_ERR_ARG, "gemv launch failed (%d): N=%d K=%d M=%d", r, N, K, M);
  return 0;
}

extern "C" int csm_gemm(csm_engine_t* e, const void* W, int wdtype, const float* wscale, int N, int K, const float* A,
                        int R, float* C) {
  if (!e) return fail(CSM_ERR_ARG, "null engine");
  GemmArgs g{};
  g.A = A; g.lda = K; g.W = W; g.wscale = wscale; g.R = R; g.N = N; g.K = K; g.C = C; g.ldc = N;
  LCK(launch_gemm(e->stream, wdtype, GEPI_STORE, g));
  return 0;
}

extern "C" int csm_gemm_bf16(csm_engine_t* e, const void* W, int N, int K, const void* A, int R, float* C, int kernel) {
  if (!e || !W || !A || !C) return fail(CSM_ERR_ARG, "null argument");
  if (kernel < 0 || kernel > 2) return fail(CSM_ERR_ARG, "kernel must be 0 (square tile), 1 (LDS-DMA 128 x 128) or 2 (LDS-DMA 256 x 256)");
  GemmArgs g{};
  g.Aplanes = reinterpret_cast<const bf16_t*>(A); g.a_plane_stride = 0; g.lda = K;
  g.W = W; g.R = R; g.N = N; g.K = K; g.C = C; g.ldc = N;
  g.dma = kernel == 1 ? 2 : 0; g.dma_max_rows = 1 << 30; g.big256 = kernel == 2 ? 1 : 0;
  // the LDS-DMA tiles are called directly: through launch_gemm a shape they do not cover would silently take another tile
  const int r = kernel == 1 ? launch_gemm_dma_bf16(e->stream, GEPI_STORE, g)
              : kernel == 2 ? launch_gemm256_bf16(e->stream, GEPI_STORE, g, 1)
                            : launch_gemm(e->stream, CSM_DTYPE_BF16, GEPI_STORE, g);
  if (r) return fail(CSM_ERR_ARG, "csm_gemm_bf16: tile %d does not cover R = %d, N = %d, K = %d (%d)", kernel, R, N, K, r);
  return 0;
}

extern "C" int csm_sample_topk(csm_engine_t* e, const float* logits, int rows, int V, float temperature, int topk,
                               uint64_t seed, const float* noise, int32_t* out_idx) {
  if (!logits || !out_idx) return fail(CSM_ERR_ARG, "null argument");
  if (topk < 1 || topk > V) return fail(CSM_ERR_ARG, "selected index k out of range");
  if (topk > 1 && temperature != 0.f && (size_t)3 * V * sizeof(float) > 160 * 1024 - 4096)
    return fail(CSM_ERR_CAPACITY, "top-k sampling keeps 3 x V floats in LDS: V = %d exceeds the 13 300-entry limit of this kernel", V);
  if (!e) LCK(configure_sample());
  hipStream_t st = e ? e->stream : nullptr;  // engine-less call (module-level sample_topk): null stream
  SampleArgs a{};
  a.logits = logits; a.ldl = V; a.V = V; a.temperature = temperature; a.topk = topk; a.seed = seed; a.noise = noise;
  a.noise_ld = V; a.cb = 0; a.C = 1; a.B = rows; a.max_frames = 1; a.idx_out = out_idx;
  a.legacy = e ? e->sample_legacy : 0;
  LCK(launch_sample(st, rows, a));
  if (!e) HIPCK(hipStreamSynchronize(st));
  return 0;
}

extern "C" int csm_attn_decode(csm_engine_t* e, int which, int layer, const float* q, const int32_t* row_seq,
                               const int32_t* row_pos, int rows, int nsplit, float* out) {
  if (!e) return fail(CSM_ERR_ARG, "null engine");
  Stack& s = which ? e->dec : e->bb;
  if (layer < 0 || layer >= s.c.layers) return fail(CSM_ERR_ARG, "bad layer");
  if (nsplit > 1 && (size_t)rows > (size_t)e->cfg.max_batch) return fail(CSM_ERR_ARG, "split attention limited to max_batch rows");
  AttnArgs t{};
  t.q = q; t.kcache = s.kc[layer]; t.vcache = s.vc[layer]; t.n_q = s.c.n_q; t.n_kv = s.c.n_kv; t.hd = s.c.head_dim;
  t.lmax = s.lmax; t.row_seq = row_seq; t.row_pos = row_pos; t.kv_start = which ? nullptr : e->d_kv_start;
  t.nsplit = nsplit < 1 ? 1 : nsplit; t.out = out; t.part = e->part_bb;
  LCK(launch_attn(e->stream, e->cfg.kv_dtype, rows, t));
  return 0;
}

// write K/V rows into a layer cache from raw qkv projections (test hook for csm_attn_decode):
// qkv [rows][(n_q+2n_kv)*hd] -> q_out [rows][n_q*hd] (rotated, scaled) + cache
extern "C" int csm_rope_scatter(csm_engine_t* e, int which, int layer, const float* qkv, const int32_t* row_seq,
                                const int32_t* row_pos, int rows, float* q_out) {
  if (!e || !e->bound) return fail(CSM_ERR_STATE, "weights not bound");
  Stack& s = which ? e->dec : e->bb;
  if (layer < 0 || layer >= s.c.layers) return fail(CSM_ERR_ARG, "bad layer");
  RopeArgs ra{};
  ra.qkv = qkv; ra.n_q = s.c.n_q; ra.n_kv = s.c.n_kv; ra.hd = s.c.head_dim; ra.qscale = 1.0f / sqrtf((float)s.c.head_dim);
  ra.cos_tab = s.cos; ra.sin_tab = s.sin; ra.row_seq = row_seq; ra.row_pos = row_pos; ra.qbuf = q_out;
  ra.kcache = s.kc[layer]; ra.vcache = s.vc[layer]; ra.lmax = s.lmax;
  LCK(launch_rope_scatter(e->stream, e->cfg.kv_dtype, rows, ra));
  return 0;
}

// ---- MX-fp8 (OCP microscaling: e4m3 elements + one E8M0 scale per 32 along K): row quantiser, GEMM hook, weight binding ----
extern "C" int csm_mx_quantize(csm_engine_t* e, const float* x, int rows, int K, uint8_t* q_out, uint8_t* s_out) {
  if (!e || !x || !q_out || !s_out || rows < 1 || K < 32 || K % 32) return fail(CSM_ERR_ARG, "bad mx_quantize arguments");
  MxQuantArgs q{};
  q.x = x; q.ldx = K; q.rows = rows; q.K = K; q.q = q_out; q.s = s_out;
  LCK(launch_mx_quant(e->stream, q));
  return 0;
}
extern "C" int csm_gemm_mx(csm_engine_t* e, const uint8_t* Wq, const uint8_t* Ws, int N, int K, const uint8_t* Aq, const uint8_t* As,
                           int R, float* C) {
  if (!e || !C) return fail(CSM_ERR_ARG, "null argument");
  GemmMxArgs g{};
  g.Aq = Aq; g.As = As; g.Wq = Wq; g.Ws = Ws; g.R = R; g.N = N; g.K = K; g.C = C; g.ldc = N; g.big = e->gemm_256;
  const int r = launch_gemm_mx(e->stream, GEPI_STORE, g);
  if (r == -2) return fail(CSM_ERR_ARG, "csm_gemm_mx covers N %% 128 == 0 and K %% 128 == 0 (got N=%d K=%d)", N, K);
  LCK(r);
  return 0;
}
extern "C" int csm_bind_mx_weights(csm_engine_t* e, const csm_mx_layer_t* layers, int n_layers) {
  if (!e) return fail(CSM_ERR_ARG, "null engine");
  if (!layers || n_layers == 0) { e->mx_layers.clear(); e->prefill_mx = 0; return 0; }
  if (n_layers != e->cfg.backbone.layers) return fail(CSM_ERR_ARG, "expected %d backbone layers, got %d", e->cfg.backbone.layers, n_layers);
  for (int l = 0; l < n_layers; ++l) {
    const csm_mx_layer_t& m = layers[l];
    if (!m.qkv || !m.qkv_s || !m.o || !m.o_s || !m.gu || !m.gu_s || !m.d || !m.d_s) return fail(CSM_ERR_ARG, "null MX weight pointer (layer %d)", l);
  }
  e->mx_layers.assign(layers, layers + n_layers);
  return 0;
}

// ---- kernel micro-benchmark hook (tools/bench_gemv.py): n_launch dependent launches of one GEMV shape,
// cycling over `n_w` weight matrices `w_stride` bytes apart, captured in a hipGraph and replayed `reps`
// times; returns microseconds per launch measured with HIP events on the engine stream. --------------
extern "C" int csm_bench_gemv(csm_engine_t* e, const void* W, size_t w_stride, int n_w, int wdtype, int N, int K,
                              const float* x, int M, const float* ln, float eps, float* y, int epi, int nt,
                              int n_launch, int reps, float* us_per_launch, int grid_cap, int v2_tasks, int force_generic) {
  if (!e || !W || !x || !y || !us_per_launch || M < 1 || M > 16 || n_w < 1) return fail(CSM_ERR_ARG, "bad bench arguments");
  const int save_wd = e->cfg.weight_dtype;
  e->cfg.weight_dtype = wdtype;
  // the pool's matrices get temporary fragment-order copies, like bound weights (batched rows only)
  std::vector<const void*> tmp_keys;
  const size_t tiled_before = e->tiled_allocs.size();
  if (M >= 2 && e->tile_weights && e->use_mfma && (wdtype == CSM_DTYPE_BF16 || wdtype == CSM_DTYPE_FP8)) {
    for (int i = 0; i < n_w; ++i) {
      const void* wi = (const char*)W + (size_t)i * w_stride;
      if (e->tiled.count(wi)) continue;
      if (int tr = tile_one(e, wi, N, K)) { e->cfg.weight_dtype = save_wd; return tr; }
      tmp_keys.push_back(wi);
    }
    HIPCK(hipStreamSynchronize(e->stream));
  }
  auto drop_tmp = [&]() {
    for (const void* k : tmp_keys) e->tiled.erase(k);
    while (e->tiled_allocs.size() > tiled_before) { hipFree(e->tiled_allocs.back()); e->tiled_allocs.pop_back(); }
  };
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  HIPCK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
  int r = 0;
  for (int i = 0; i < n_launch && !r; ++i) {
    GemvArgs a{};
    a.W = (const char*)W + (size_t)(i % n_w) * w_stride; a.N = N; a.K = K; a.x = x; a.ldx = K + (force_generic >> 8); a.ln = ln; a.eps = eps;
    a.out = y; a.ldo = (epi == EPI_SWIGLU) ? N / 2 : N; a.nt = nt;
    a.grid_cap = grid_cap & 0xffff; a.v2_tasks = v2_tasks; a.force_generic = force_generic & 0xff;
    a.g16_nw = (grid_cap >> 16) & 0xff; a.g16_kb = (grid_cap >> 24) & 0x3f; a.g16_pt = (grid_cap >> 30) & 1 ? 4 : ((grid_cap >> 16) ? 1 : 0);
    r = gemv_rows(e, M, ln ? PRO_NORM : PRO_PLAIN, epi, a);
  }
  e->cfg.weight_dtype = save_wd;
  hipError_t ce = hipStreamEndCapture(e->stream, &g);
  if (r) { if (g) hipGraphDestroy(g); drop_tmp(); return fail(CSM_ERR_ARG, "bench launch failed (%d)", r); }
  HIPCK(ce);
  HIPCK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  HIPCK(hipGraphLaunch(ge, e->stream));
  HIPCK(hipStreamSynchronize(e->stream));
  HIPCK(hipEventRecord(e->ev0, e->stream));
  for (int i = 0; i < reps; ++i) HIPCK(hipGraphLaunch(ge, e->stream));
  HIPCK(hipEventRecord(e->ev1, e->stream));
  HIPCK(hipEventSynchronize(e->ev1));
  float ms = 0.f;
  HIPCK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  *us_per_launch = ms * 1000.f / ((float)reps * n_launch);
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  drop_tmp();
  return 0;
}

#include "train_impl.inc"
