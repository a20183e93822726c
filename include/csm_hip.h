/*
 * csm_hip.h -- C ABI of libcsm_hip.so, the MI355X (gfx950) implementation of the CSM generation hot
 * path.  Plain C: pointers, sizes, POD structs -- no torch / C++ types cross this boundary.
 *
 * The reference (thomasgauthier/csm-hf) is pure Python and has NO plugin/FFI interface for this path
 * (SURVEY.md section 8-b): the path sits behind `CSMModel.forward / generate_frame / generate`
 * (/root/reference/modeling_csm.py:292-365, 484-589, 591-702).  Each entry point below names the
 * reference interface it replaces; the host-side mirror of the Python API lives in
 * `csm-hf_amd/modeling_csm.py` and binds these symbols with ctypes.
 *
 * Conventions
 *   - return 0 on success; negative = csm error code, positive = hipError_t.  Never throws.
 *     `csm_last_error()` returns a thread-local message for the last failure.
 *   - every `const void*`/`void*` is a DEVICE pointer unless the name ends in `_host`.
 *   - weights are BORROWED (caller keeps them alive, e.g. torch tensors); KV caches, activation
 *     scratch, RoPE tables and graphs are engine-owned.
 *   - an engine is bound to one device and one stream and is not thread-safe; multi-GPU = one engine
 *     (one process) per device.
 *   - nothing allocates or synchronises inside a captured region.
 */
#ifndef CSM_HIP_H
#define CSM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSM_ABI_VERSION 7

enum { CSM_DTYPE_F32 = 0, CSM_DTYPE_BF16 = 1, CSM_DTYPE_FP8 = 2 /* OCP e4m3fn + per-output-row fp32 scale (matrices only) */ };

enum {
  CSM_OK = 0,
  CSM_ERR_ARG = -1,       /* bad argument / unsupported shape */
  CSM_ERR_STATE = -2,     /* call sequence error (e.g. frame before prefill) */
  CSM_ERR_CAPACITY = -3,  /* batch / length exceeds what kv_reserve allocated */
  CSM_ERR_NOMEM = -4
};

/* One Llama stack (reference: LlamaConfig fields used at modeling_csm.py:68-109). */
typedef struct {
  int32_t hidden, ffn, layers, n_q, n_kv, head_dim;
  float rms_eps;
} csm_llama_cfg_t;

/* reference: CSMConfig (modeling_csm.py:52-143) + engine sizing knobs. */
typedef struct {
  int32_t abi_version;   /* must be CSM_ABI_VERSION */
  int32_t text_vocab, audio_vocab, n_codebooks;
  csm_llama_cfg_t backbone, decoder;
  int32_t weight_dtype;  /* CSM_DTYPE_*: dtype of all matrices; embedding tables use it too, except FP8 -> bf16 tables */
  int32_t kv_dtype;      /* CSM_DTYPE_*: backbone/decoder KV-cache storage */
  int32_t max_batch;     /* sequences per engine (per GPU) */
  int32_t max_len;       /* backbone KV positions per sequence (context + generated frames) */
  int32_t max_frames;    /* capacity of the on-device generated-frame ring */
  int32_t max_prefill_rows; /* max B*S handled by one csm_prefill call (activation scratch) */
} csm_config_t;

/* Per-layer weights, ENGINE LAYOUT (host packs once at load time, csm-hf_amd/engine.py):
 *   wqkv [(n_q+2*n_kv)*head_dim, hidden]  = cat(q_proj, k_proj, v_proj) rows
 *   wo   [hidden, n_q*head_dim]
 *   wgu  [2*ffn, hidden]                  row 2i = gate_proj row i, row 2i+1 = up_proj row i
 *   wd   [hidden, ffn]
 *   ln1, ln2 [hidden]  fp32 always
 *   sqkv, so, sgu, sd  per-output-row fp32 scales of the four matrices (CSM_DTYPE_FP8 only, else NULL):
 *                      W[n,k] = fp8[n,k] * s[n]
 */
typedef struct {
  const void *wqkv, *wo, *wgu, *wd;
  const float *ln1, *ln2;
  const float *sqkv, *so, *sgu, *sd;
} csm_layer_weights_t;

typedef struct {
  const csm_layer_weights_t* layers; /* host array [cfg.layers] */
  const float* final_norm;           /* [hidden] fp32 */
  const float* rope_cos;             /* [rope_positions, head_dim/2] fp32, host-computed llama3 table */
  const float* rope_sin;
  int32_t rope_positions;
} csm_stack_weights_t;

/* reference tensors: modeling_csm.py:222-240 (names in SURVEY.md section 8 f-1). */
typedef struct {
  csm_stack_weights_t backbone, decoder;
  const void* text_emb;      /* [text_vocab, Hb] */
  const void* audio_emb;     /* [n_codebooks*audio_vocab, Hb] */
  const void* proj_head0;    /* [Hd + audio_vocab, Hb] = cat(projection.weight, codebook0_head.weight) */
  const void* audio_head_t;  /* [n_codebooks-1, audio_vocab, Hd] = audio_head.transpose(1,2) */
  const float* proj_table;   /* [n_codebooks*audio_vocab, Hd] fp32 = projection(audio_emb); may be NULL at
                                bind time and supplied later by csm_build_proj_table / csm_set_proj_table */
  const float* s_proj_head0;   /* [Hd + audio_vocab] row scales (FP8 only) */
  const float* s_audio_head;   /* [(n_codebooks-1) * audio_vocab] row scales (FP8 only) */
} csm_weights_t;

typedef struct csm_engine csm_engine_t;

/* sampling controls of one frame (reference: sample_topk, modeling_csm.py:179-189). */
typedef struct {
  float temperature;     /* 0 => argmax (the reference would produce NaNs) */
  int32_t topk;          /* 1 => greedy */
  uint64_t seed;         /* Philox key for the Exp(1) race; ignored when noise != NULL or greedy.  Kept in device
                            memory by the engine (written before every launch), so it is NOT part of a captured graph */
  const float* noise;    /* optional explicit Exp(1) draws [max_frames?1][B][n_codebooks][audio_vocab] for
                            parity tests; indexed [b][cb][v] for the CURRENT frame */
  const int64_t* forced; /* optional teacher-forced tokens [B][max_frames][n_codebooks]: fed back instead
                            of the model's samples (samples are still recorded) */
  float* logits_trace;   /* optional [max_frames][B][n_codebooks][audio_vocab] fp32 dump of every logits row */
  float* last_h_trace;   /* optional [max_frames][B][Hb] fp32 dump of last_hidden_state per frame */
  int32_t row_offset;    /* global index of this engine's row 0 (batch-sharded generation): the
                            Philox counter uses row_offset + local row, so shards draw distinct, world-size-independent
                            streams.  No reference counterpart (torch's global generator, modeling_csm.py:175) */
  int32_t per_row_stop;  /* 1: a row that has emitted an all-zero frame is FROZEN (it keeps emitting zeros), so the all-zero
                            test fires when the last row finishes -- per-row end of utterance.
                            0: the reference's rule (modeling_csm.py:662): rows keep generating until ALL rows emit an
                            all-zero frame in the same step */
} csm_sampling_t;

/* ---- lifecycle: CSMModel.__init__ / setup_caches / reset_caches (modeling_csm.py:214-245, 284-290) */
int csm_engine_create(const csm_config_t* cfg, int device, void* stream /* hipStream_t */, csm_engine_t** out);
int csm_engine_destroy(csm_engine_t* e);
int csm_bind_weights(csm_engine_t* e, const csm_weights_t* w);
/* proj_table[r,:] = projection.weight @ audio_emb[r,:]  (fp32 accumulate, fp32 out, caller-owned
 * [n_codebooks*audio_vocab, Hd] buffer); replaces the per-codebook embed+projection of
 * modeling_csm.py:535,542,564-565 by one table row read.  Needs max_prefill_rows scratch. */
int csm_build_proj_table(csm_engine_t* e, float* proj_table_out);
int csm_set_proj_table(csm_engine_t* e, const float* proj_table);
int csm_reset(csm_engine_t* e);   /* reset_caches(): lengths, frame counter; graphs stay */
/* Engine options -- the COMPLETE list (ABI 6 removed 36 A/B knobs whose losing variants left the library; round 5 added six names,
 * round 6 the streamer's health / schedule names, "sample_legacy", "g16_k16"; an unknown name is CSM_ERR_ARG).  Defaults are the measured best; every call drops the captured graphs.
 *   precision:   "prefill_bf16" (context activations rounded to bf16), "prefill_mx" (context linears on the MX-fp8 matrix
 *                instruction; needs csm_bind_mx_weights), "prefill_bf16_attn" (context attention on the bf16 pipe in those
 *                modes), "decode_bf16" (batched decode on ONE nearest-even activation plane: the reference's own bf16 class)
 *   decode:      "nsplit_backbone" (KV splits of the backbone attention, 0 = by length), "fuse_attn_oproj" (B = 1 decoder
 *                attention + o_proj as one launch), "fuse_attn_combine" (backbone attention: the last KV split of a (row, head)
 *                merges the partials inside the launch), "fuse_sample" (B = 1: greedy arg-max folded into the head launch, top-k sampling into the next QKV launch),
 *                "two_token_pass" (positions 0 and 1 of the decoder as one 2-row pass, modeling_csm.py:534-552), "use_planes"
 *                (bit mask: batched activations as MFMA B-operand planes), "rows64" (1: 17-128 rows in ONE launch per linear; 0: 32-row
 *                launches; -1: 16-row launches -- the forms the width tests compare against, bit for bit),
 *                "tile_weights" (fragment-order weight copies of the matrix-core kernel; 0 frees them), "weight_prefetch"
 *                (weight streamer on / off), "prefetch_window_mb" (bytes it may run ahead, default 6: round 5), "prefetch_sub_kb" (size of a run of
 *                consumer workgroups, the unit the schedule is made of: default 8192 since round 6 -- 4096 before; B = 1 2.82 -> 2.73 ms), "prefetch_seg_sleep"
 *                (its pause per run, default 0 since the decode kernels run at s_setprio 3), "prefetch_lead" / "prefetch_cofetch" / "prefetch_skip_late" /
 *                "prefetch_depth" / "prefetch_poll_sleep" / "prefetch_stride" (schedule and loader details, defaults 1 / 1 / 1 / 0 / 2 / 0: swept again in round 6,
 *                profiles/r06_streamer_grid.txt), "prefetch_batched" (A/B: also stream the matrix-core launches of a 2-16-row batch; measured slower, default 0),
 *                "g16_k16" (A/B: nw | kb << 8 of the K = 2048 matrix-core launches; 0 = one 16-wave workgroup per panel), "prefetch_budget_us" / "prefetch_rearm" /
 *                "prefetch_force_serial" (streamer health: see csm_prefetch_health), "sample_legacy" (TEST HOOK: csm_sample_topk on the histogram / radix
 *                selection of rounds 1-4 instead of sample_wave.h's: the two are compared token for token on adversarial rows), "kernel_prio" (bit mask of the
 *                launch families that raise their issue priority: 1 decoder attention + o_proj, 2 GEMV / skinny GEMM, 4 backbone
 *                attention + samplers; default 7), "attn_oproj_gqa" (B = 1 decoder attention + o_proj with the K/V tiles shared by
 *                the query heads of a kv-head), "oproj_combine" / "combine_splits" (B = 1 backbone: split-KV merge inside the o_proj
 *                launch, on this many splits <= 8), "attn_gqa_wide" (backbone attention beyond 32 rows on the kernel that shares K/V tiles
 *                among the query heads of a kv-head), "g16_kfast" (K-split matrix-core launches: the k split as the fastest grid index -- one XCD per k slice of the activation planes),
 *                "g16_xcdmap" (65-128 rows: a gate/up panel runs on the XCD that reads its output columns as a k group of the down_proj launch),
 *                "g128" / "g128_min" / "g128_shape" (FFN launches of batches beyond g128_min = 64 rows on
 *                gemm128.h: weight rows split over the waves, planes shared through LDS; shape = A/B override: low nibble weight tiles per
 *                wave, bit 6 one k group per workgroup for gate/up)
 *   prefill:     "gemm_wide", "gemm_dma", "gemm_256", "gemm_dma_skinny", "gemm_mx_skinny" (tile selection of the context GEMMs:
 *                used by the bitwise tile-vs-tile tests), "prefill_splitk", "prefill_splitk_gu" (K splits of short prefills),
 *                "prefill_fuse_rope" (RoPE + cache append as the QKV GEMM's epilogue, modeling_llama.py:130-176 / 267-281),
 *                "prefill_fuse_quant" / "mx_fuse_swiglu" (MX quantisation fused into the producing kernels), "prefill_attn_kvfast" (bf16-class
 *                context attention: the kv-head as the fastest grid index -- one XCD per kv-head)
 *   measurement: "dbg_skip" (TIMING ONLY, wrong results: knock launch kinds out of the decode chain), "dbg_sample_spin" (TIMING ONLY:
 *                every sampler launch idles this many 10 ns ticks first -- how a pause in the chain affects the weight streamer) */
int csm_set_option(csm_engine_t* e, const char* name, int value);

/* ---- CSMModel.forward, S>=1 rows on an empty or partly filled cache (modeling_csm.py:321-365).
 * ids [B,S,C+1] int64, mask [B,S,C+1] uint8 (0/1).  Appends S positions to the backbone KV cache and
 * leaves the codebook-0 logits of the last position (and the decoder's position-0 input) in the
 * engine, ready for csm_decode_frame.  Optional outputs: last_h [B,Hb] fp32, c0_logits [B,V] fp32. */
int csm_prefill(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, int B, int S,
                float* last_h_out, float* c0_logits_out);

/* same with caller-supplied RoPE positions (reference `position_ids`, modeling_csm.py:296,349 -> LlamaModel): position_ids
 * [B*S] int32 on the device, each < the bound RoPE table; the cache slot of a row stays (cached length + s), like HF,
 * which masks by cache index and rotates by position_ids */
int csm_prefill_pos(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, int B, int S, const int32_t* position_ids,
                    float* last_h_out, float* c0_logits_out);
/* ---- Mimi decode: `audio_tokenizer.decode(gen_frames.permute(0, 2, 1))`, the step right
 * after the generation path (/root/reference/README.md:58-60, 114-118; train.py:363-365).  The codec is the third-party
 * package moshi==0.2.2 (absent from the reference tree and from the image); its decode path is restated from the published
 * architecture as implemented by transformers 5.15 models/mimi/modeling_mimi.py:1388-1455 (MimiModel.decode), which the
 * parity fixtures are generated with.  fp32 throughout.  Weights are repacked by the host (csm-hf_amd/mimi.py:
 * pack_mimi_weights documents every layout). */
#define CSM_MIMI_MAX_LAYERS 16
#define CSM_MIMI_MAX_RATIOS 8
typedef struct csm_mimi csm_mimi_t;
typedef struct {
  int32_t abi_version;
  int32_t n_q, n_sem, codebook_size, codebook_dim;        /* split residual VQ: n_sem semantic + (n_q - n_sem) acoustic codebooks */
  int32_t hidden, layers, heads, head_dim, ffn, window;   /* decoder transformer; window = sliding attention window */
  float rope_theta, norm_eps;
  int32_t n_ratios, ratios[CSM_MIMI_MAX_RATIOS];          /* SEANet upsampling ratios, in decode order */
  int32_t num_filters, kernel_size, last_kernel_size, res_kernel_size, compress, up_stride;
  int32_t max_frames;                                     /* capacity: codec frames per sequence */
} csm_mimi_config_t;
typedef struct {                                          /* fp32 device pointers */
  const float* embed;      /* [n_q][codebook_size][codebook_dim] = embed_sum / clamp(cluster_usage, 1e-5) */
  const float* out_proj;   /* [hidden][2 * codebook_dim] = [semantic output_proj | acoustic output_proj] */
  const float* upsample;   /* [hidden][2 * up_stride] depthwise taps */
  const float* ln1_w[CSM_MIMI_MAX_LAYERS]; const float* ln1_b[CSM_MIMI_MAX_LAYERS];
  const float* wqkv[CSM_MIMI_MAX_LAYERS];  /* [3 * heads * head_dim][hidden] = [q_proj; k_proj; v_proj] */
  const float* wo[CSM_MIMI_MAX_LAYERS];    /* [hidden][heads * head_dim] */
  const float* ls1[CSM_MIMI_MAX_LAYERS];
  const float* ln2_w[CSM_MIMI_MAX_LAYERS]; const float* ln2_b[CSM_MIMI_MAX_LAYERS];
  const float* w1[CSM_MIMI_MAX_LAYERS];    /* [ffn][hidden] */
  const float* w2[CSM_MIMI_MAX_LAYERS];    /* [hidden][ffn] */
  const float* ls2[CSM_MIMI_MAX_LAYERS];
  const float* conv0_w; const float* conv0_b;   /* [C0][k * hidden]: W'[co][j * C_in + ci] = w[co][ci][j];  [C0] */
  const float* up_w[CSM_MIMI_MAX_RATIOS];       /* [r * C_out][2 * C_in]: row s * C_out + co = [w[:, co, s + r] | w[:, co, s]] */
  const float* up_b[CSM_MIMI_MAX_RATIOS];       /* [C_out] */
  const float* res1_w[CSM_MIMI_MAX_RATIOS];     /* [pad128(hid)][k * C_out] (rows beyond hid zero), layout as conv0_w */
  const float* res1_b[CSM_MIMI_MAX_RATIOS];     /* [hid] */
  const float* res2_w[CSM_MIMI_MAX_RATIOS];     /* [pad128(C_out)][hid] */
  const float* res2_b[CSM_MIMI_MAX_RATIOS];     /* [C_out] */
  const float* last_w; const float* last_b;     /* [k * C]: w'[j * C + c] = w[0][c][j];  [1] */
} csm_mimi_weights_t;
int csm_mimi_create(const csm_mimi_config_t* cfg, csm_mimi_t** out);
int csm_mimi_destroy(csm_mimi_t* m);
/* tuning switches of one handle: "skinny_rows" (GEMMs of <= n rows on the weight-streaming skinny GEMM; 0 = none),
 * "splitk" (K split of GEMMs with too few tiles; 0 / 1).  With both 0 a stream is bitwise the one-shot decode. */
int csm_mimi_set_option(csm_mimi_t* m, const char* name, int value);
int csm_mimi_bind_weights(csm_mimi_t* m, const csm_mimi_weights_t* w);   /* borrowed pointers */
/* codes [B][n_q][T] int64 (device) -> audio [B][T * samples_per_frame] fp32 (device); samples_per_frame = up_stride * prod(ratios) */
int csm_mimi_decode(csm_mimi_t* m, const int64_t* codes, int B, int T, float* audio);
/* streaming: one sequence a few frames at a time (codes [n_q][T] of the NEW frames); the handle keeps the transformer's K/V
 * window and every convolution's left context (transformers: decoder_past_key_values + MimiConv1dPaddingCache,
 * modeling_mimi.py:73-166, 1388-1406), so the concatenated chunks equal one csm_mimi_decode of the whole sequence */
int csm_mimi_stream_reset(csm_mimi_t* m);
int csm_mimi_stream_decode(csm_mimi_t* m, const int64_t* codes, int T, float* audio);
/* stream GROUPS (round 3): S streams -- the rows of a generated batch (reference call site README.md:114-118, applied frame by
 * frame to every row of `generate`'s output) -- advance in lockstep, T <= max_frames / S frames each per call, and every launch
 * covers all of them: codes [S][n_q][T] -> audio [S][T * samples_per_frame].  csm_mimi_streams_open allocates the group's state
 * (and replaces an earlier group); csm_mimi_streams_reset(m, s) restarts stream s alone (s = -1: all), so a batch row taken
 * over by a new utterance (continuous batching) starts from silence while the others continue.  Each stream's chunks
 * concatenate to csm_mimi_decode of its whole sequence (fp32 summation order of the GEMM paths aside). */
int csm_mimi_streams_open(csm_mimi_t* m, int S);
int csm_mimi_streams_reset(csm_mimi_t* m, int stream);
int csm_mimi_streams_decode(csm_mimi_t* m, const int64_t* codes, int T, float* audio);

/* ---- continuous batching (no reference counterpart): a new utterance takes over batch row
 * `row` of the running batch between two frame-steps.  ids [S][C+1] / mask [S][C+1] on the device; S <= the batch's
 * current length (the context is placed right-aligned, like a left-padded row of the reference; longer contexts:
 * csm_shift_context first; contexts longer than max_prefill_rows are prefilled in chunks).  The row's frames from
 * the current frame index on belong to the new utterance; other rows are untouched. */
int csm_prefill_slot(csm_engine_t* e, int row, const int64_t* ids, const uint8_t* mask, int S);
/* several utterances take over several rows in ONE prefill (round 3): ids / mask [n][S][C+1] left-padded by the caller to the
 * common S (mask 0 on pad frames), lens[i] = true frames of context i, rows[i] = its batch row (host arrays).  Same result as n
 * csm_prefill_slot calls; needs n * S <= max_prefill_rows and S <= the batch's current length (CSM_ERR_CAPACITY otherwise). */
int csm_prefill_slots(csm_engine_t* e, const int32_t* rows, const int32_t* lens, int n, const int64_t* ids, const uint8_t* mask, int S);
/* A context LONGER than the running batch's current length: first move every resident row `delta` cache slots up (keys
 * re-rotated by `delta`: RoPE is relative, the rows' results change by fp32 rounding only; kv_start of every row and the
 * shared length grow by delta; length + delta <= max_len), then csm_prefill_slot.  No reference counterpart. */
int csm_shift_context(csm_engine_t* e, int delta);

/* ---- MX-fp8 on the CDNA4 block-scaled matrix instruction (BASELINE configs[4]: "fp8 weights (CDNA4 fp8 MFMA)"; no
 * reference counterpart -- the reference runs bf16).  OCP microscaling: e4m3 elements + one E8M0 scale byte (2^(b - 127))
 * per 32 consecutive elements along K.  csm_bind_mx_weights hands the engine MX copies of the backbone's packed linears
 * ([N][K] bytes + [N][K/32] scales each; borrowed pointers); `csm_set_option(e, "prefill_mx", 1)` then runs every linear
 * of a context prefill (transformers LlamaModel at q_len > 1, reference call site modeling_csm.py:345-354) on
 * v_mfma_scale_f32_16x16x128_f8f6f4 with activations quantised to the same format.  The two hooks are the unit-parity
 * entries: x fp32 [rows][K] -> (q, s) on the device; C[R][N] = dequant(A) dequant(W)^T in fp32. */
typedef struct {
  const uint8_t *qkv, *qkv_s, *o, *o_s, *gu, *gu_s, *d, *d_s;
} csm_mx_layer_t;
int csm_bind_mx_weights(csm_engine_t* e, const csm_mx_layer_t* backbone_layers, int n_layers);   /* NULL / 0: unbind */
int csm_mx_quantize(csm_engine_t* e, const float* x, int rows, int K, uint8_t* q_out, uint8_t* s_out);
int csm_gemm_mx(csm_engine_t* e, const uint8_t* Wq, const uint8_t* Ws, int N, int K, const uint8_t* Aq, const uint8_t* As,
                int R, float* C);
/* ---- CSMModel.forward with labels (modeling_csm.py:367-465): the training objective, FORWARD ONLY.
 * labels [B,S,C+1] int64 on the device, -100 = ignored.  out3 (device, 3 floats) = (loss, backbone_loss, decoder_loss):
 * cross-entropy of the codebook-0 logits of position t against labels[:, t+1, 0], plus cross-entropy of the decoder's
 * codebook 1..C-1 logits over the frames whose C audio labels are all present -- the reference's
 * nn.CrossEntropyLoss(ignore_index=-100) means.  Starts from an empty cache (csm_reset) and leaves the context prefilled
 * like csm_prefill does (last_h_out / c0_logits_out as there, nullable).  No backward pass is provided. */
int csm_forward_loss(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, const int64_t* labels, int B, int S,
                     float* out3, float* last_h_out, float* c0_logits_out);
/* ---- the same objective WITH its gradients (reference consumer: train.py:308-326, CSMTrainer.compute_loss -> loss.backward()).
 * Gradients of `loss` w.r.t. every parameter are ACCUMULATED (+=) into caller-owned fp32 device buffers laid out like the
 * engine's packed weights (csm_layer_weights_t): wqkv [(n_q + 2 n_kv) hd][H] = [q; k; v] rows, wgu [2 F][H] with gate / up
 * rows interleaved, proj_head0 [Hd + V][Hb] = [projection; codebook0_head], audio_head_t [C-1][V][Hd] (the transposed
 * slices); norm weights [H]; embedding tables in their own shapes.  fp32 or bf16 weights; fp32 arithmetic throughout
 * (csrc/train.h).  Starts from scratch: the KV cache is neither read nor written; csm_set_kv_start gives the left padding. */
typedef struct {
  float *dwqkv, *dwo, *dwgu, *dwd, *dln1, *dln2;
} csm_layer_grads_t;
typedef struct {
  const csm_layer_grads_t* layers;   /* host array [cfg.layers] */
  float* final_norm;
} csm_stack_grads_t;
typedef struct {
  csm_stack_grads_t backbone, decoder;
  float* text_emb;       /* [text_vocab, Hb] */
  float* audio_emb;      /* [n_codebooks * audio_vocab, Hb] */
  float* proj_head0;     /* [Hd + audio_vocab, Hb] */
  float* audio_head_t;   /* [n_codebooks - 1, audio_vocab, Hd] */
} csm_grads_t;
int csm_forward_backward(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, const int64_t* labels, int B, int S,
                         float* out3, const csm_grads_t* grads);
/* backbone KV cache <-> the HF layout of `past_key_values` (transformers DynamicCache: per layer keys / values
 * [B, n_kv, len, head_dim]; reference modeling_csm.py:355-358 returns it, :349 takes it back), fp32 on the device.
 * Export reads the resident batch; import + csm_set_length continue a context the caller built, forked or edited. */
int csm_kv_export(csm_engine_t* e, int layer, float* k_out, float* v_out, int len);
int csm_kv_import(csm_engine_t* e, int layer, const float* k_in, const float* v_in, int B, int len);
int csm_set_length(csm_engine_t* e, int B, int len);

/* ---- CSMModel.generate_frame minus its backbone forward (modeling_csm.py:522-589): sample c0,
 * run the 31-step decoder loop, write the frame into the on-device ring at the current frame index. */
int csm_decode_frame(csm_engine_t* e, const csm_sampling_t* s);
/* ---- the backbone step of the NEXT generate_frame call (modeling_csm.py:508-520 with S=1): embeds
 * the frame just generated (or the forced one), appends one KV position, computes c0 logits. */
int csm_backbone_step(csm_engine_t* e, const csm_sampling_t* s);
/* same step but fed with an explicit [B,1,C+1] row (API path of forward / generate_frame with
 * past_key_values); `advance_frame` != 0 when a csm_decode_frame preceded it. */
int csm_backbone_step_ids(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, int B, int advance_frame);
/* copy out the pending last_hidden_state [B,Hb] / codebook-0 logits [B,V] (either may be NULL) */
int csm_get_state(csm_engine_t* e, float* last_h_out, float* c0_logits_out);

/* ---- CSMModel.generate (modeling_csm.py:631-702) after prefill: n_frames x (decode_frame +
 * backbone_step), replayed from a hipGraph when `use_graph`.  Frames land in the ring; read them with
 * csm_read_frames.  No host sync inside. */
int csm_generate(csm_engine_t* e, const csm_sampling_t* s, int n_frames, int use_graph);
int csm_read_frames(csm_engine_t* e, int64_t* frames_out /* device [B,n,C] */, int first, int n);
/* Stop test without a host sync per frame (reference: `torch.all(new_frame == 0)`, modeling_csm.py:662, one sync per
 * frame): every backbone step counts, on the device, the rows whose frame was all-zero; out[i] = that count for frame
 * first + i.  generate() replays k frames, reads k counters once, and cuts at the first frame whose count == B. */
int csm_read_zero_counts(csm_engine_t* e, int32_t* out_host, int first, int n);   /* syncs the stream */
/* generate_frame-driven streaming (modeling_csm.py:484-589 hands every frame to the caller): restart the on-device
 * frame ring at slot 0 once the caller has read what it holds, so a stream is not limited to max_frames frames */
int csm_rewind_frames(csm_engine_t* e);
/* captured-graph bookkeeping (tests): graphs captured since creation, graphs currently cached (LRU, <= 8) */
int csm_graph_stats(csm_engine_t* e, int* captured_total_host, int* cached_host);
/* weight streamer (csrc/prefetch.h: a persistent kernel on a second stream that pulls the replaying graph's weights into
 * the XCD-local L2 ahead of the launches that read them; engine options "weight_prefetch", "prefetch_window_mb",
 * "prefetch_sub_kb", "prefetch_grid").  out8 = {workgroups that gave up, workgroups finished, segments skipped as late
 * (one sampled wave), workgroup->XCD rotation (-1: streamer disabled), segments in the schedule, streamed launches per
 * frame-step, scheduled bytes, bytes of all streamed launches, launches counted, frames} for the last csm_generate;
 * syncs both streams */
int csm_prefetch_stats(csm_engine_t* e, long long* out10_host);
/* health of the weight streamer (ABI 7).  The streamer ends when the launch counter reaches the total of the call; "no launch started
 * for `prefetch_budget_us` (default 20 000) while launches are outstanding" is a stalled chain: the workgroups give up, and the NEXT
 * csm_generate (which reads a pinned status mirror -- no synchronisation) re-runs the stream-concurrency probe and switches the
 * streamer off for this engine: the reference loop it feeds (modeling_csm.py:644-690) must never stall behind an optimisation.
 * out8 = {disabled reason: 0 on / 1 the two streams share a hardware queue / 2 two give-ups within 64 calls with the probe passing /
 * 3 workgroups not dispatched round-robin over the XCDs / 4 probe failed at engine creation / 5 the engine stream's launches slow down
 * > 1.6 x beside a kernel resident on the streamer's stream (the two share a hardware queue slot: an 18 ms frame-step otherwise), strikes, lifetime give-ups
 * (workgroups), lifetime finished (workgroups), streamer launches, budget (us), probe runs, launches not yet accounted for};
 * the reason as text in csm_last_error().  Waits for the last streamer launch only.  Options: "prefetch_budget_us",
 * "prefetch_rearm" (clears reasons 1 / 2), "prefetch_force_serial" (TEST HOOK: streamer and probe on the engine stream) */
int csm_prefetch_health(csm_engine_t* e, long long* out8_host);
/* debug probes of the weight streamer (tools/streamer_probe.py): per-workgroup {XCD id, clocks until the weights were
 * consumed} of the first n streamed launches of the next captured frame-step -> buf[n][2048][2] uint32 (device);
 * geometry {N, K, grid, tasks per workgroup, kind} of the streamed launches of the last captured frame-step */
int csm_set_debug_buffer(csm_engine_t* e, uint32_t* buf, int n_launches);
int csm_last_geoms(csm_engine_t* e, int32_t* out_host, int max_launches, int* n_host);
/* Move the live state of `src` (KV caches of the resident batch, lengths, frame ring, pending codebook-0 logits) into
 * `dst`, an engine of the same model with larger capacities -- the reference's DynamicCache simply grows
 * (transformers cache_utils.py:144-145); here a continuation that outgrows max_len / max_frames re-homes its cache */
int csm_kv_copy(csm_engine_t* dst, csm_engine_t* src);
int csm_frames_done(csm_engine_t* e, int* n_host);            /* syncs the stream */
int csm_cur_len(csm_engine_t* e, int* len_host);              /* syncs the stream */
/* per-row first valid KV position (left padding): pads are masked at every step */
int csm_set_kv_start(csm_engine_t* e, const int32_t* kv_start_host, int B);

/* ---- timing hook for bench.py: HIP events on the engine stream around the last csm_generate */
int csm_last_generate_ms(csm_engine_t* e, float* ms_host);

/* ---- per-kernel entry points (unit parity tests; all on the engine stream) --------------------- */
/* K1 frame embedding: out[r,:] = sum_c mask[r,c] * table_c[ids[r,c]]   (modeling_csm.py:261-282,327-334) */
int csm_embed_sum(csm_engine_t* e, const int64_t* ids, const uint8_t* mask, int rows, float* out);
/* K2 RMSNorm (transformers LlamaRMSNorm): out = w * (x * rsqrt(mean(x^2)+eps)) */
int csm_rmsnorm(csm_engine_t* e, const float* x, const float* w, int rows, int hidden, float eps, float* out);
/* K3/K8/K9 skinny GEMM y[M,N] = x[M,K] @ W[N,K]^T, M <= 16, optional fused RMSNorm prologue */
int csm_gemv(csm_engine_t* e, const void* W, int wdtype, const float* wscale /* fp8 row scales, nullable */, int N,
             int K, const float* x, int M, const float* ln /* nullable */, float eps, float* y);
/* prefill GEMM C[R,N] = A[R,K] @ W[N,K]^T on the MFMA path */
int csm_gemm(csm_engine_t* e, const void* W, int wdtype, const float* wscale, int N, int K, const float* A, int R,
             float* C);
/* the same product as the context prefill runs it with `prefill_precision = "bf16"`: activations ALREADY rounded to bf16 by
 * their producer (one row-major plane [R][K]), bf16 weights, fp32 accumulation on v_mfma_f32_*_bf16.  `kernel` pins the tile:
 * 0 = square tile (gemm_bf16x3_kernel, one plane), 1 = both operands staged by LDS-DMA (gemm_dma_bf16_kernel, 128 x 128),
 * 2 = the 256 x 256 LDS-DMA tile (gemm256_kernel).  Test hook for the per-GEMM check against the fp64 product of the rounded
 * operands (the nn.Linear call sites of the context forward, modeling_csm.py:345-354).  CSM_ERR_ARG when the shape is not
 * covered by the requested tile. */
int csm_gemm_bf16(csm_engine_t* e, const void* W /* bf16 [N][K] */, int N, int K, const void* A /* bf16 [R][K] */, int R, float* C,
                  int kernel);
/* K12 sampler on a [rows,V] logits matrix; noise nullable; returns int32 indices */
int csm_sample_topk(csm_engine_t* e, const float* logits, int rows, int V, float temperature, int topk,
                    uint64_t seed, const float* noise, int32_t* out_idx);
/* K4-K6 one attention step on caller-provided q and a caller-provided cache (decode kernel):
 * q [rows, n_q*hd] fp32 (already rotated, unscaled), kc/vc engine cache layout, pos[rows] */
int csm_attn_decode(csm_engine_t* e, int which /*0 backbone,1 decoder*/, int layer, const float* q,
                    const int32_t* row_seq, const int32_t* row_pos, int rows, int nsplit, float* out);
/* K4/K5 RoPE + KV append for `rows` raw qkv projections [rows,(n_q+2n_kv)*hd] into a layer cache */
int csm_rope_scatter(csm_engine_t* e, int which, int layer, const float* qkv, const int32_t* row_seq,
                     const int32_t* row_pos, int rows, float* q_out);

/* micro-benchmark hook: n_launch dependent launches of one GEMV shape in a hipGraph, us per launch */
int csm_bench_gemv(csm_engine_t* e, const void* W, size_t w_stride, int n_w, int wdtype, int N, int K,
                   const float* x, int M, const float* ln, float eps, float* y, int epi, int nt,
                   int n_launch, int reps, float* us_per_launch, int grid_cap, int v2_tasks, int force_generic);

int csm_sync(csm_engine_t* e);
const char* csm_last_error(void);
int csm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CSM_HIP_H */
