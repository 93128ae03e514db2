/*
 * chimeralm_hip.h -- C ABI of the MI355X (gfx950) inference engine for ChimeraLM's `predict` hot path.
 *
 * The drop-in boundary of the reference is the `net` module of `ClassificationLit`
 *   /root/reference/chimeralm/models/basic_module.py:14-22,38,67-77   (ClassificationLit.forward -> self.net)
 *   /root/reference/chimeralm/models/components/hyena.py:218-256      (HyenaDna.__init__ / forward)
 * i.e. `forward(input_ids int64[B,L], input_quals=None) -> logits fp32[B,2]`.  The reference is pure Python
 * on torch and has no FFI of its own; these entry points are what a ctypes binding behind
 * `HyenaDna.forward` binds (INTEGRATION.md shows the stub).  Plain pointers and sizes only -- no torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative CLM_E_* code; `clm_last_error` gives the text.
 *     No exception crosses this boundary.
 *   - one handle per GPU; a handle is not re-entrant; different handles are independent.
 *   - `clm_forward` is asynchronous on the caller's HIP stream and performs no host synchronisation,
 *     PROVIDED the workspace is already large enough (`clm_reserve`); growing it synchronises the stream first.
 *   - weight memory is copied at `clm_load_weight`; the caller keeps ownership of `data`.
 */
#ifndef CHIMERALM_HIP_H
#define CHIMERALM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLM_ABI_VERSION 5

/* error codes */
#define CLM_OK 0
#define CLM_E_INVALID (-1)     /* bad argument / shape / unknown key            */
#define CLM_E_HIP (-2)         /* a HIP runtime call failed                      */
#define CLM_E_MISSING (-3)     /* clm_finalize: a required weight was not loaded */
#define CLM_E_UNSUPPORTED (-4) /* shape or mode outside what the kernels cover   */
#define CLM_E_STATE (-5)       /* call order (e.g. forward before finalize)      */

/* element types for clm_load_weight / token ids */
#define CLM_DT_F32 0
#define CLM_DT_F64 1
#define CLM_DT_BF16 2
#define CLM_DT_F16 3
#define CLM_DT_U8 4
#define CLM_DT_I32 5
#define CLM_DT_I64 6

/* arithmetic type of the dense projections (in_proj / out_proj / fc1 / fc2 / pooling score GEMM).
 * Accumulation, LayerNorm statistics, the residual stream, the long convolution (FFT) and the softmax are
 * fp32 in every mode.  F32 uses v_mfma_f32_32x32x2_f32 (exact fp32); BF16/F16 use v_mfma_f32_32x32x16_*. */
#define CLM_PREC_F32 0
#define CLM_PREC_BF16 1
#define CLM_PREC_F16 2
/* F16C: fp16 activations; the weights of in_proj, out_proj and the score layer held as hi + lo, hi = fp16(w) and lo = e4m3((w - hi)
 * * 2^17): per 64-deep group four fp16 MFMAs with hi and one block-scaled fp8 MFMA with lo (activations truncated to e5m2 in
 * registers), one fp32 accumulator; the two MLP products on plain fp16 weights (their rounding does not show in the logits).  The mode
 * that runs at 16-bit MFMA rate AND stays within the reference's 1e-3 logit tolerance (weight rounding is the error
 * that attention pooling cannot average out, tests/error_model.py); z / y are stored as fp16 like CLM_PREC_F16. */
#define CLM_PREC_F16C 3
/* F16X3 (ABI 4): near-exact at fp16 MFMA rate / 3.  Every operand of a dense projection is split into two halfs, x = hi + lo with
 * hi = fp16(x) and lo = fp16(x - hi) (~21 bits; weights pre-scaled by 2^10 so that their lo halfs stay normal), and a product is
 * three fp16 MFMAs into the fp32 accumulator: w_hi a_hi + w_lo a_hi + w_hi a_lo.  Everything else is the exact-fp32 engine (fp32
 * z / y / residual stream in HBM, fp32 convolution, score layer on the fp32 MFMA).  Logits within ~1e-5 of exact fp32 -- three
 * orders of magnitude inside the reference's 1e-3 -- at 2x its rate; unguarded by default (csrc/tail32.hip, AR_X3).  ABI 5: also
 * the arithmetic every 16-bit handle runs its short reads (clm_set_short_read_len) and its first fall-back level in. */
#define CLM_PREC_F16X3 4

typedef struct clm_handle clm_handle;

/* Mirrors the hyper-parameters fixed by the reference at
 *   chimeralm/models/lm.py:19-31 (head: 256 -> 512, 2 layers, attention pooling, gelu, residual) and by the
 *   HyenaDNA-small-32k config.  Only these values are accepted (checked in clm_create). */
typedef struct clm_config {
    int32_t struct_size;   /* = sizeof(clm_config), ABI guard                               */
    int32_t d_model;       /* 256                                                           */
    int32_t n_layer;       /* 4                                                             */
    int32_t d_inner;       /* 1024                                                          */
    int32_t vocab_rows;    /* 16  (vocab 12 padded to a multiple of 8)                      */
    int32_t filter_order;  /* 64                                                            */
    int32_t emb_dim;       /* 5                                                             */
    int32_t max_seq_len;   /* 32770 rows of pos_emb.z / pos_emb.t                           */
    int32_t head_hidden;   /* 512                                                           */
    int32_t n_classes;     /* 2                                                             */
    float ln_eps;          /* 1e-5                                                          */
    int32_t precision;     /* CLM_PREC_*                                                    */
    int32_t chunk_reads;   /* reads pushed through all layers together (default 256; the engine lowers it for long reads so that a
                              chunk holds at most 256 x 8,256 tokens in the 16-bit modes, 64 x 8,256 in exact fp32: workspace ~ chunk) */
} clm_config;

int clm_abi_version(void);
int clm_default_config(clm_config* cfg);

/* Replaces HyenaDna.__init__ (hyena.py:218-242): allocate an engine on HIP device `device`. */
int clm_create(const clm_config* cfg, int device, clm_handle** out);

/* Replaces the state_dict load of PyTorchModelHubMixin.from_pretrained / Lightning ckpt_path
 * (lm.py:17, __main__.py:317).  `key` is the reference checkpoint key with or without the `net.` prefix, e.g.
 * `net.backbone.backbone.layers.0.mixer.in_proj.weight`, `net.head.attention.0.weight`.
 * `data` may be host or device memory (hipMemcpyDefault); `shape[ndim]` is checked against the model.
 * Unknown keys return CLM_E_INVALID; the aliases `implicit_filter.{3,5}.freq` of the shared sine module are
 * accepted and ignored in favour of `.1.freq`. */
int clm_load_weight(clm_handle* h, const char* key, const void* data, int dtype, const int64_t* shape, int ndim);

/* Packs weights for the kernels (MFMA fragment order, compute dtype) and checks completeness.
 * May be called again after further clm_load_weight calls; the cached filter spectra are dropped. */
int clm_finalize(clm_handle* h);

/* Grow the workspace for batches up to B reads of L tokens (optional; clm_forward does it on demand). */
int clm_reserve(clm_handle* h, int B, int L);

/* Replaces HyenaDna.forward(input_ids, input_quals=None) (hyena.py:244-256).
 *   ids        device pointer, [B, L] row-major with `ids_row_stride` elements between rows,
 *              dtype CLM_DT_I64 (what the reference's collator produces), CLM_DT_I32 or CLM_DT_U8.
 *   logits_out device pointer, fp32 [B, n_classes].
 *   stream     hipStream_t (NULL = default stream). */
int clm_forward(clm_handle* h, const void* ids, int ids_dtype, int64_t ids_row_stride, int B, int L,
                float* logits_out, void* stream);

/* ---- batches that arrive in host memory ------------------------------------------------------------
 * Replaces the `batch["input_ids"].to(device)` Lightning performs before predict_step (basic_module.py:177-187 receives a
 * device batch).  Two device staging buffers per handle: clm_stage_ids enqueues the H2D copy of a batch on the HANDLE'S OWN
 * copy stream (so it overlaps the forward pass running on the compute stream) and returns which buffer it used;
 * clm_forward_staged makes the compute stream wait for exactly that copy and runs the forward on it; clm_stage_wait blocks
 * the host until the copy has left the host buffer (a pinned slot of the BAM feeder, chimeralm_feed.h, can then be
 * released).  A staging buffer is not overwritten before the forward that read it has finished (event-ordered).
 * `host_ids` should be page-locked for the copy to be asynchronous. */
int clm_stage_ids(clm_handle* h, const void* host_ids, int ids_dtype, int64_t ids_row_stride, int B, int L, int* staged);
int clm_forward_staged(clm_handle* h, int staged, float* logits_out, void* stream);
int clm_stage_wait(clm_handle* h, int staged);

/* Errors a forward can only detect on the device after the call has returned: a token id outside [0, vocab_rows), for
 * which the reference's nn.Embedding raises IndexError inside HyenaDna.forward (hyena.py:249).  The id kernels clamp such an
 * id (no wild read) and flag the handle; the flag is reported ONCE, as CLM_E_INVALID with the message in clm_last_error,
 * by the next clm_forward / clm_forward_staged / clm_stage_wait, or by clm_check, which first waits for `stream`. */
int clm_check(clm_handle* h, void* stream);

/* ---- the 16-bit modes checked against, and replaced by, the reference's arithmetic -------------------------------------
 * The reference computes the whole forward in ONE precision, fp32 (hyena.py:244-256).  A 16-bit handle (CLM_PREC_F16C / F16 /
 * BF16) also holds the exact-fp32 packing of its weights and the exact-fp32 kernels, so the deviation of its mode from the
 * reference's arithmetic can be MEASURED on the weights that were loaded and on reads the caller chooses, instead of assumed
 * from other weights:
 *   clm_selfcheck   runs `ids` (device memory, as clm_forward) through the handle's mode AND through the exact-fp32 kernels
 *                   and returns the largest |logit difference| (`max_abs_diff`, host; +inf if anything is not finite) and the
 *                   number of reads whose argmax differs (`labels_differ`, may be NULL).  Synchronises `stream`.  Not affected
 *                   by clm_set_fallback; 0 for a CLM_PREC_F32 handle.  Lengths a CLM_PREC_F16C handle runs in its fp16x3 kernels
 *                   (clm_set_short_read_len) are measured like any other: fp16x3 against exact fp32.
 *   clm_set_fallback(h, level)   ABI 5: a LEVEL, not a switch -- the reference computes one precision always (hyena.py:244-256), so
 *                   the answer to "the fast mode is off" is the next-fastest arithmetic that is inside its tolerance, not the
 *                   slowest.  0 = the handle's own mode.  1 = every later clm_forward* runs in the next arithmetic inside the
 *                   gate: CLM_PREC_F16X3 on a 16-bit handle (fp32-class logits at about twice the exact-fp32 rate), exact fp32
 *                   on a CLM_PREC_F16X3 handle.  2 = exact fp32 on every handle.  (The caller's reaction to a self-check above its
 *                   threshold: chimeralm_amd/hyena.py uses level 1 at 5e-4, half the reference tolerance.)  No effect on a
 *                   CLM_PREC_F32 handle.  Other values: CLM_E_INVALID.
 *   clm_effective_precision  the CLM_PREC_* code reads of L tokens run in right now (CLM_PREC_F16X3 for the short reads of a
 *                   CLM_PREC_F16C handle and for a 16-bit handle at fall-back level 1). */
int clm_selfcheck(clm_handle* h, const void* ids, int ids_dtype, int64_t ids_row_stride, int B, int L, void* stream,
                  float* max_abs_diff, int* labels_differ);
int clm_set_fallback(clm_handle* h, int level);
int clm_effective_precision(const clm_handle* h, int L);
/* The reduction both self-checks report (pure host arithmetic, no device needed): a, b = logits [B][n_classes] of the mode and of
 * the referee; *max_abs_diff = the largest |a - b|, +inf as soon as ANY difference is not finite (and it stays +inf: a NaN in read
 * 0 must not be overwritten by a finite difference of read 1); *labels_differ = reads whose argmax differs (may be NULL). */
int clm_logit_deviation(const float* a, const float* b, int B, int n_classes, float* max_abs_diff, int* labels_differ);
/* CLM_PREC_F16C only: reads shorter than `min_len` tokens run in the fp16x3 kernels inside the mode (ABI 5; exact fp32 before, and
 * still at fall-back level 2) -- default 2,048: the mode's error is a sum of per-token fp16 roundings that the attention pooling
 * averages like 1 / sqrt(L); below some length it no longer fits half the tolerance.  That length depends on the weights: the caller
 * may MEASURE it with clm_selfcheck(mode) on reads of decreasing length after clm_set_short_read_len(h, 1) (chimeralm_amd/hyena.py
 * does, at 1,024 / 512 / 256 tokens, and lowers the switch only with a margin of 2 under its threshold) and lower or raise it. */
int clm_set_short_read_len(clm_handle* h, int min_len);
/* CLM_PREC_F16C only (round 4): the two MLP products (fc1, fc2: two thirds of the dense FLOPs) run on plain fp16 weights by default
 * -- on most weights their rounding does not show in the logits -- and on hi + lo weights, like in_proj / out_proj / the score
 * layer, after clm_set_mlp_compensation(h, 1) (both packings are held; takes effect with the next forward; ~10 % slower).  The
 * caller decides with clm_selfcheck on the loaded weights: chimeralm_amd/hyena.py switches it on when the default form measures
 * above its threshold, and falls back to exact fp32 only if this form does too. */
int clm_set_mlp_compensation(clm_handle* h, int on);

/* ---- SequenceCNNTransformer -----------------------------------------------------------
 * Multi-head self-attention of nn.TransformerEncoderLayer as the reference builds it
 * (/root/reference/chimeralm/models/components/transformer.py:64-68,98: d_model 256, 8 heads of 32, no masks):
 *   qkv  device, [B, L, 768] 16-bit, the in_proj output q | k | v per token;  out  device, [B, L, 256] 16-bit, heads concatenated
 *   precision CLM_PREC_F16 or CLM_PREC_BF16 (element type of qkv / out; statistics and accumulation are fp32). */
int clm_attention_fwd(const void* qkv, void* out, int B, int L, int precision, void* stream);

/* The whole SequenceCNNTransformer forward (transformer.py:88-104; configuration of configs/model/transformer.yaml:3-12:
 * vocab 12, d_model 256, kernel 3, 8 heads, feed-forward 1024, `n_layers` encoder layers) behind the same `net` boundary as
 * HyenaDna: forward(input_ids[B, L], input_quals=None) -> logits fp32 [B, 2].  Same conventions as the clm_* calls above;
 * weights are loaded under the reference module's state_dict keys (with or without the `net.` prefix), the buffer
 * `pos_encoder.pe` [1, max_len, 256] included; fp32 tensors only.  L >= 8; L / 8 must not exceed max_len (the reference
 * asserts the same, transformer.py:21).  clm_tf_debug_fetch names: "hidden" fp32 [B, L/8, 256] (encoder output),
 * "scores" fp32 [B, L/8] (pooling scores before the softmax), "pooled" fp32 [B, 256]. */
typedef struct clm_tf_handle clm_tf_handle;
int clm_tf_create(int device, int precision /* CLM_PREC_F32 (exact, parity mode) | CLM_PREC_F16X3 (fp32-class, the module default) | CLM_PREC_F16C | CLM_PREC_F16 | CLM_PREC_BF16 */,
                  int n_layers, clm_tf_handle** out);
int clm_tf_load_weight(clm_tf_handle* h, const char* key, const void* data, int dtype, const int64_t* shape, int ndim);
int clm_tf_finalize(clm_tf_handle* h);
int clm_tf_forward(clm_tf_handle* h, const void* ids, int ids_dtype, int64_t ids_row_stride, int B, int L, float* logits_out,
                   void* stream);
/* Round 3: the 16-bit mode on trial, as clm_selfcheck / clm_set_fallback for the Hyena engine.  A 16-bit handle keeps the raw fp32
 * tensors, so the exact-fp32 kernels (tf_fp32.hip) can run on the same handle: clm_tf_selfcheck runs `ids` through the handle's
 * mode and through them, synchronises `stream` and reports max |logit difference| and the number of reads whose label differs
 * (0 / 0 on an fp32 handle); clm_tf_set_fallback(h, level) has clm_set_fallback's levels: 1 = every later clm_tf_forward of a
 * 16-bit handle runs the fp32-path kernels on hi + lo halfs (fp16x3; exact fp32 on an fp16x3 handle), 2 = exact fp32. */
int clm_tf_selfcheck(clm_tf_handle* h, const void* ids, int ids_dtype, int64_t ids_row_stride, int B, int L, void* stream,
                     float* max_abs_diff_out, int* labels_differ_out);
int clm_tf_set_fallback(clm_tf_handle* h, int level);
int clm_tf_debug_fetch(clm_tf_handle* h, const char* name, void* host_out, size_t bytes);
/* Profiling taps of the 16-bit path (bench.py --net transformer): accumulated HIP-event time on the launch stream and number of
 * spans per stage -- 0 conv stack + positional encoding / LayerNorm, 1 attention, 2 encoder layer kernel (out_proj + LayerNorm-1
 * + feed-forward + LayerNorm-2 + next layer's QKV), 3 pooling head.  clm_tf_profile_read synchronises the device. */
int clm_tf_profile_enable(clm_tf_handle* h, int on);
int clm_tf_profile_read(clm_tf_handle* h, double* ms_out /*[4]*/, int64_t* spans_out /*[4]*/, int reset);
const char* clm_tf_last_error(const clm_tf_handle* h);
int clm_tf_destroy(clm_tf_handle* h);

/* ---- test / measurement taps (not on the product path) -------------------------------------------- */

/* Copy a named intermediate of the LAST clm_forward to host memory (synchronises the device).  Names:
 *   "hidden"        fp32 [B, L, 256]  residual stream after the final block (before ln_f)
 *   "scores"        fp32 [B, L]       pooling scores before the softmax
 *   "pooled"        fp32 [B, 256]
 *   "filter.<i>"    fp32 [L, 256]     implicit long filter of layer i for the last L
 * Only data of the last processed chunk is meaningful for "hidden"/"scores" when B > chunk_reads. */
int clm_debug_fetch(clm_handle* h, const char* name, void* host_out, size_t bytes);
/* Make clm_forward return right after stage `stage` (CLM_STAGE_*) of block `layer` (-1 with CLM_STAGE_EMBED:
 * after the embedding); the raw buffers "h", "z" [B,768,Lp], "y" [B,256,Lp], "u" [B,L,1024] can then be
 * fetched (z/y/u in the activation storage type of the precision mode).  layer = -1, stage = -1 disables. */
int clm_debug_stop_after(clm_handle* h, int layer, int stage);

/* Per-stage device timing with HIP events on the forward stream.  Stage names: clm_profile_stage_name. */
#define CLM_STAGE_EMBED 0
#define CLM_STAGE_INPROJ 1
#define CLM_STAGE_CONV 2
#define CLM_STAGE_OUTPROJ 3
#define CLM_STAGE_FC1 4
#define CLM_STAGE_FC2 5
#define CLM_STAGE_SCORE 6
#define CLM_STAGE_POOL 7
#define CLM_STAGE_HEADMLP 8
#define CLM_STAGE_FILTER 9
#define CLM_STAGE_TAIL 10   /* profile only: fused out_proj + LN2 + fc1 + GELU + fc2 (16-bit modes) */
#define CLM_STAGE_MLP 11    /* profile only: fused LN2 + fc1 + GELU + fc2 (16-bit modes, CLM_DEBUG=split_tail) */
#define CLM_N_STAGES 12
int clm_profile_enable(clm_handle* h, int on);
/* Synchronises, then returns accumulated milliseconds and launch counts per stage since the last reset. */
int clm_profile_read(clm_handle* h, double* ms_out /*[CLM_N_STAGES]*/, int64_t* launches_out /*[CLM_N_STAGES]*/,
                     int reset);
const char* clm_profile_stage_name(int stage);

const char* clm_last_error(const clm_handle* h); /* h may be NULL: error of the last failed clm_create */
int clm_destroy(clm_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* CHIMERALM_HIP_H */
