/*
 * hig.h -- C ABI of the MI355X-native motion-diffusion denoiser hot path (libhig.so).
 *
 * The reference (line/Human-Interaction-Generation) has no FFI layer: the boundary it exposes
 * is the Python object API (MotionTransformer / GaussianDiffusion / DDPMTrainer).  This ABI is
 * what sits UNDER our mirror of that API; every entry point cites the reference code whose
 * arithmetic it replaces.  Conventions:
 *   - extern "C", plain pointers and sizes; returns 0 (HIG_OK) or a negative HIG_E* code;
 *     never throws, never exits; hig_last_error() holds the message of the last failure
 *     on the calling thread.
 *   - The caller owns every buffer.  The library allocates no device memory, frees nothing,
 *     keeps no reference past return.  All device pointers must be valid on the current HIP
 *     device (hipSetDevice is the caller's job).
 *   - No global mutable state except (a) what hig_shutdown() releases (the per-thread weight-gradient stream below),
 *     (b) the diagnostic stamp pointers of hig_gemm_debug_stamps / hig_gemm_bf16_debug_stamps /
 *     hig_gemm_ws16_debug_stamps / hig_linattn16_debug_stamps (NULL unless a profiling tool sets them; never set during a timed or captured run),
 *     and (c) tuning knobs read ONCE from the environment (HIG_*: DESIGN.md section 5 lists them) into function-local
 *     constants on first use -- after that first call they never change, so calls stay re-entrant.
 *   - Every launch is ordered through the `stream` argument (a hipStream_t passed as void*): on it,
 *     or -- hig_denoiser_bwd's weight gradients only -- on a library-owned stream forked from and
 *     joined back into it with events.  No host synchronisation, device allocation, blocking copy or
 *     host read-back happens inside, so any call sequence can be captured into a hipGraph.
 *   - Matrices are row-major fp32 unless stated; nn.Linear weights are (out, in).
 */
#ifndef HIG_H
#define HIG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIG_OK 0
#define HIG_EINVAL (-1)
#define HIG_EHIP (-2)
#define HIG_EUNSUPPORTED (-3)

#define HIG_ATTN_LINEAR 0 /* LinearTemporal*Attention, transformer.py:89-155 (default) */
#define HIG_ATTN_FULL 1   /* Temporal*Attention (no_eff=True), transformer.py:196-262 */

/* GEMM compute modes */
#define HIG_PREC_F32 0    /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate */
#define HIG_PREC_BF16X3 1 /* split-bf16 (hi*hi + hi*lo + lo*hi) on v_mfma_f32_32x32x16_bf16 */
#define HIG_PREC_BF16 2   /* single bf16 product, fp32 accumulate */

typedef void* hig_stream_t; /* hipStream_t */

int hig_version(void);
/* Copies the calling thread's last error message (NUL terminated) into buf; returns its length. */
int hig_last_error(char* buf, int n);

/* ------------------------------------------------------------------------------------------
 * Problem description.  Mirrors the constructor arguments of MotionTransformer
 * (transformer.py:289-304) plus the call-time batch shape.
 * ---------------------------------------------------------------------------------------- */
typedef struct hig_dims {
  int32_t B;          /* samples in the batch */
  int32_t T;          /* motion frames per sample (<= num_frames) */
  int32_t F;          /* input_feats */
  int32_t d;          /* latent_dim */
  int32_t H;          /* num_heads (d % H == 0, head dim multiple of 4, <= 128) */
  int32_t ff;         /* ff_size */
  int32_t L;          /* num_layers */
  int32_t N;          /* text tokens (77) */
  int32_t Lt;         /* text_latent_dim */
  int32_t num_frames; /* rows of sequence_embedding */
  int32_t attn_kind;  /* HIG_ATTN_* */
  int32_t prec;       /* HIG_PREC_* */
  int32_t two_person; /* 0: MotionTransformer.  1: MotionInteractionTransformer (interaction_transformer.py
                         :397-616): B = 2 x pairs, x = [person1; person2], token 0 of every sample is the
                         init-pose row (joint_embed2 on its first 4 features, out2), and each layer has the
                         person<->person linear cross-attention `int_ca_block` (:167-207).
                         2: the same with no_cross_attn=True (no int_ca_block). */
  int32_t storage;    /* HIG_STORE_F32 (0): fp32 activations and weights.  HIG_STORE_BF16 (1): bf16 activations and a
                         bf16 shadow of the weight matrices, fp32 accumulation / LayerNorm statistics / softmax /
                         context matrices / modulation vectors -- BASELINE configs 3 and 5.  Inference
                         (hig_denoiser_fwd_bf16): single-person model (linear or full attention) and two-person model
                         (linear attention); training (hig_denoiser_fwd_bf16_train / hig_denoiser_bwd_bf16): single-person
                         model with linear attention.  Head dim 64 or 128, d, ff, Lt multiples of 32. */
} hig_dims;
#define HIG_STORE_F32 0
#define HIG_STORE_BF16 1

/* Parameter table: an array of device pointers, HIG_NGLOBAL global entries followed by
 * HIG_NLAYER entries per decoder layer, in the order below.  The same table layout is used
 * for gradients.  Three groups must be CONTIGUOUS in memory so one GEMM covers them:
 *   SA_QKV_W = [query.weight; key.weight; value.weight]  (3d, d)   SA_QKV_B (3d)
 *   CA_KV_W  = [key.weight; value.weight]                (2d, Lt)  CA_KV_B  (2d)
 *   STY_EMB_W= the 3L `emb_layers.1.weight` matrices stacked in (layer, sa|ca|ffn) order
 *              (3L*2d, E); STY_EMB_B likewise (3L*2d).  With two_person == 1 there are 4 per
 *              layer, in (sa|ca|int_ca|ffn) order.
 * Names are the reference's state-dict keys. */
enum {
  HIG_P_SEQ_EMB = 0, /* sequence_embedding (num_frames, d) */
  HIG_P_JOINT_W,     /* joint_embed.weight (d, F) */
  HIG_P_JOINT_B,
  HIG_P_TE0_W, /* time_embed.0 (E, d) */
  HIG_P_TE0_B,
  HIG_P_TE2_W, /* time_embed.2 (E, E) */
  HIG_P_TE2_B,
  HIG_P_STY_EMB_W,
  HIG_P_STY_EMB_B,
  HIG_P_OUT_W, /* out.weight (F, d) */
  HIG_P_OUT_B,
  /* two-person model only (NULL otherwise) */
  HIG_P_JOINT2_W, /* joint_embed2.weight (d, 4) */
  HIG_P_JOINT2_B,
  HIG_P_OUT2_W,   /* out2.weight (F, d) */
  HIG_P_OUT2_B,
  HIG_NGLOBAL
};
enum {
  HIG_L_SA_NORM_W = 0,
  HIG_L_SA_NORM_B,
  HIG_L_SA_QKV_W,
  HIG_L_SA_QKV_B,
  HIG_L_SA_STY_NORM_W,
  HIG_L_SA_STY_NORM_B,
  HIG_L_SA_STY_OUT_W, /* proj_out.out_layers.2 (d, d) */
  HIG_L_SA_STY_OUT_B,
  HIG_L_CA_NORM_W,
  HIG_L_CA_NORM_B,
  HIG_L_CA_TNORM_W, /* ca_block.text_norm (Lt) */
  HIG_L_CA_TNORM_B,
  HIG_L_CA_Q_W,
  HIG_L_CA_Q_B,
  HIG_L_CA_KV_W,
  HIG_L_CA_KV_B,
  HIG_L_CA_STY_NORM_W,
  HIG_L_CA_STY_NORM_B,
  HIG_L_CA_STY_OUT_W,
  HIG_L_CA_STY_OUT_B,
  HIG_L_FFN_W1, /* ffn.linear1 (ff, d) */
  HIG_L_FFN_B1,
  HIG_L_FFN_W2, /* ffn.linear2 (d, ff) */
  HIG_L_FFN_B2,
  HIG_L_FFN_STY_NORM_W,
  HIG_L_FFN_STY_NORM_B,
  HIG_L_FFN_STY_OUT_W,
  HIG_L_FFN_STY_OUT_B,
  /* two-person model with interaction attention only (NULL otherwise) */
  HIG_L_INT_NORM_W, /* int_ca_block.norm (d): applied to the own AND the partner stream */
  HIG_L_INT_NORM_B,
  HIG_L_INT_QKV_W,  /* [query; key; value] (3d, d), contiguous */
  HIG_L_INT_QKV_B,
  HIG_L_INT_STY_NORM_W,
  HIG_L_INT_STY_NORM_B,
  HIG_L_INT_STY_OUT_W,
  HIG_L_INT_STY_OUT_B,
  HIG_NLAYER
};

/* Bytes of scratch the forward needs.  training != 0 keeps every layer's activations for
 * hig_denoiser_bwd; training == 0 reuses one layer's buffers for all layers. */
int64_t hig_workspace_bytes(const hig_dims* dims, int training);
/* Bytes for the step-invariant cross-attention text context (all L layers). */
int64_t hig_textctx_bytes(const hig_dims* dims, int training);

/* Cross-attention text side for all L layers: text_norm, key/value projections, softmax over
 * the N text tokens, A = k^T v (transformer.py:146-152).  Step-invariant during sampling, so
 * p_sample_loop calls it once and passes `textctx` to every hig_denoiser_fwd. */
int hig_text_context(const hig_dims* dims, const void* const* params, const float* xf_out,
                     void* textctx, int training, hig_stream_t stream);

/* MotionTransformer.forward with text embeddings supplied (transformer.py:407-426):
 * x (B,T,F), t (B) int64, length (B) int64 or NULL (= all T), xf_proj (B,4d), out (B,T,F). */
int hig_denoiser_fwd(const hig_dims* dims, const void* const* params, const float* x,
                     const int64_t* t, const int64_t* length, const float* xf_proj,
                     const void* textctx, float* out, void* workspace, int training,
                     hig_stream_t stream);

/* hig_text_context followed by hig_denoiser_fwd as ONE call -- what MotionTransformer.forward computes per call
 * (transformer.py:144-150 inside :407-426): the text side (xf_out -> `textctx`, caller-owned scratch of hig_textctx_bytes) is
 * forked onto a library-owned stream and joined layer by layer (layer l's context matrices are first needed in front of
 * layer l's cross-attention), so its 2 L small launches run next to the first decoder layers instead of in front of them.
 * Same results as the two calls, bit for bit.  Under hipGraph capture (or HIG_TEXT_FORK=0) the text side runs first on `stream`. */
int hig_denoiser_fwd_text(const hig_dims* dims, const void* const* params, const float* x, const int64_t* t,
                          const int64_t* length, const float* xf_proj, const float* xf_out, void* textctx, float* out,
                          void* workspace, int training, hig_stream_t stream);

/* The general fp32-storage forward.  xf_out (nullable): compute the text side in this call (hig_denoiser_fwd_text), else
 * `textctx` is an input.  derived32 (nullable; inference only): 6 L + 4 device pointers the caller derives from the parameters and
 * rebuilds when they change -- [6 l + 3 k + 0 .. 2], k = 0 self-attention q/k/v (3d rows), k = 1 cross-attention query (d rows):
 * W' (fp32, rows x d) = gamma (.) W of the LayerNorm in front of the projection, colsum (rows) = row sums of W', bias' (rows) =
 * b + W beta.  With them the LayerNorm launches in front of those projections disappear (d % 128 == 0): the stylization-out
 * GEMM that produces the residual stream writes its row statistics (hig_gemm_desc.row_stats_out), the projection applies them
 * in its epilogue (row_stats_in).  Any entry may be NULL (that projection then keeps its LayerNorm kernel).
 * The table has 6 L + 4 entries: [6 L + 0 .. 3] (all four or none; linear attention only) = the text side's key/value weights of
 * ALL layers with their text_norm folded in, stacked as one (L 2d, Lt) fp32 matrix [gamma_l (.) Wk_l, l = 0 .. L-1; gamma_l (.) Wv_l,
 * l = 0 .. L-1] (every layer's key rows, then every layer's value rows), the bias' in the same order
 * b_l + W_l beta_l (L 2d), a vector of Lt ones and one of Lt zeros: with xf_out given, the L key/value GEMMs of the per-call
 * text side run as ONE product over the affine-free LayerNorm of the text rows (`textctx` must then have hig_textctx_bytes(dims,
 * 0) bytes, which includes that form's staging).  HIG_TEXT_BATCH=0 keeps the per-layer form. */
int hig_denoiser_fwd_x(const hig_dims* dims, const void* const* params, const void* const* derived32, const float* x,
                       const int64_t* t, const int64_t* length, const float* xf_proj, const float* xf_out, void* textctx,
                       float* out, void* workspace, int training, hig_stream_t stream);

/* bf16-storage forward (dims->storage == HIG_STORE_BF16, inference).  `params` is the fp32 table above (biases,
 * LayerNorm vectors and the F-wide input projection are read from it), `params16` the same table laid over the bf16
 * shadow of the flat parameter buffer (hig_cast_bf16): entry k = shadow base + 2 x (offset of entry k in floats).
 * x, xf_proj, xf_out and out stay fp32 (the DDPM state is carried in fp32); every activation in between is bf16.
 * Workspace / text-context sizes: hig_workspace_bytes / hig_textctx_bytes with the same dims (training = 0). */
int hig_text_context_bf16(const hig_dims* dims, const void* const* params, const void* const* params16,
                          const float* xf_out, void* textctx, hig_stream_t stream);
/* derived (nullable): 13 L + 5 device pointers the caller derives from the parameters and keeps next to the bf16 shadow
 * (rebuilt when the parameters change); any entry may be NULL (the library then does that piece per call / unfused).
 * [13 l + 3 k + 0 .. 2], k = 0, 1, 2, d == 512 or 1024: W' (bf16, rows x d), colsum (fp32, rows), bias' (fp32, rows) of the LayerNorm +
 * Linear pair k of layer l -- k = 0 self-attention q/k/v (3d rows), k = 1 cross-attention query (d rows), k = 2 q/k/v of the
 * person <-> person attention (two-person model, 3d rows) -- with W' = gamma (.) W of the LayerNorm in front of the Linear,
 * colsum[j] = sum_r float(W'[j][r]), bias' = b + W beta.  With them (and >= 2048 rows) the LayerNorm kernels in front of
 * those projections disappear: the GEMM that produces the residual stream also writes its row statistics, and the
 * projection applies them in its epilogue (hig_gemm16_desc).
 * [13 l + 9 + s], s = 0, 1, 2, 3: the stylization-out weight of the self- / cross- / person <-> person attention / FFN block
 * in matrix-core operand order (hig_weight_frag16) for hig_attn_out16 / hig_rows_out16.
 * [13 L]: joint_embed weight padded and rounded for hig_joint_embed_bf16_w ((d, Fp) bf16, Fp = F rounded up to 32).
 * [13 L + 1 .. 13 L + 4] (all four or none; linear attention; read by hig_denoiser_fwd_bf16_x when it computes the text side):
 * the key/value weights of ALL layers with their text_norm folded in, stacked as one (L 2d, Lt) bf16 matrix (every layer's key
 * rows gamma_l (.) Wk_l, then every layer's value rows), bias' = b_l + W_l beta_l in the same order (fp32, L 2d), Lt ones and
 * Lt zeros (fp32): the text side then runs ONE key/value GEMM over the affine-free LayerNorm of the text rows and ONE context
 * build instead of L of each (`textctx`: hig_textctx_bytes covers that form's staging).  HIG_TEXT_BATCH=0: per-layer form. */
int hig_denoiser_fwd_bf16(const hig_dims* dims, const void* const* params, const void* const* params16,
                          const void* const* derived, const float* x,
                          const int64_t* t, const int64_t* length, const float* xf_proj, const void* textctx,
                          float* out, void* workspace, hig_stream_t stream);
/* hig_text_context_bf16 + hig_denoiser_fwd_bf16 as ONE call, the reference's per-call forward (transformer.py:144-150 inside
 * :415-430): xf_out (nullable; NULL = textctx is already built, exactly hig_denoiser_fwd_bf16) is the (B, N, Lt) fp32 text
 * encoding, textctx is WRITTEN.  Launched eagerly, the B-row work -- the timestep-embedding chain with its one GEMM over every
 * stylization block's modulation weight, and the text side (one event per layer) -- runs on a library-owned stream next to
 * the frame-row launches and is joined where its results are first read; under stream capture, or without the library's
 * streams, everything runs in order on `stream`.  Results are identical either way (same kernels, same operands). */
int hig_denoiser_fwd_bf16_x(const hig_dims* dims, const void* const* params, const void* const* params16,
                            const void* const* derived, const float* x, const int64_t* t, const int64_t* length,
                            const float* xf_proj, const float* xf_out, void* textctx, float* out, void* workspace,
                            hig_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * bf16-storage TRAINING step (dims->storage == HIG_STORE_BF16; single-person model, linear attention, head dim 64 / 128).
 * Reference arithmetic: DDPMTrainer.forward / backward_G / update (trainers/ddpm_trainer.py:97-119,172-187) over
 * MotionTransformer.forward (models/transformer.py:60-194,407-426); the reference trains in fp32 -- this mode keeps fp32
 * MASTER parameters, gradients and optimizer state, and runs the step's matrix products on the bf16 matrix cores:
 *   - activations saved for the backward in bf16 (per layer: LN outputs, q/k/v, attention outputs, stylization inputs /
 *     outputs, the residual stream, FFN pre-activation z and gelu(z)); hd x hd context matrices, softmax statistics,
 *     modulation vectors and the per-sample embedding chain (time_embed, emb, scale / shift) in fp32;
 *   - data gradients dX = dC . W through hig_gemm_bf16 on a transposed bf16 copy of the layer's weights (epilogues
 *     HIG_EPI_RES: + the incoming residual gradient, HIG_EPI_DGELU: x gelu'(z));
 *   - weight gradients dW = dC^T . act as split-R bf16 GEMMs over the B T rows into fp32 slabs, summed in a fixed order
 *     (hig_gemm_bf16_split; operands transposed by hig_transpose_bf16_batch), written to the fp32 gradient table;
 *   - LayerNorm / stylization / linear-attention backward kernels with bf16 rows (hig_ln_bwd_bf16, hig_linattn_*_bwd_bf16);
 *   - the F-wide input / output projections and the B-row embedding chain on the fp32 kernels (their operands are fp32).
 * hig_workspace_bytes / hig_textctx_bytes (training = 1) and hig_bwd_workspace_bytes size the buffers for these dims.
 * `params` = fp32 master table, `params16` = the same table over the bf16 shadow (hig_cast_bf16 / hig_clip_adam_shadow).
 * ---------------------------------------------------------------------------------------- */
int hig_text_context_bf16_train(const hig_dims* dims, const void* const* params, const void* const* params16,
                                const float* xf_out, void* textctx, hig_stream_t stream);
int hig_denoiser_fwd_bf16_train(const hig_dims* dims, const void* const* params, const void* const* params16, const float* x,
                                const int64_t* t, const int64_t* length, const float* xf_proj, const void* textctx,
                                float* out, void* workspace, hig_stream_t stream);
/* Writes (does not accumulate) every entry of `grads` (fp32, the layout of `params`), dx (B,T,F) or NULL, dxf_proj (B,4d),
 * dxf_out (B,N,Lt), all fp32.  Weight gradients run on the library's second stream like hig_denoiser_bwd's. */
int hig_denoiser_bwd_bf16(const hig_dims* dims, const void* const* params, const void* const* params16, const float* x,
                          const int64_t* t, const int64_t* length, const float* xf_out, const void* textctx,
                          const void* workspace, const float* dout, void* const* grads, float* dx, float* dxf_proj,
                          float* dxf_out, void* bwd_workspace, hig_stream_t stream);
/* hig_denoiser_bwd_bf16 with the per-layer hook of hig_denoiser_bwd_hooked below (same contract: hook(user, l) once layer l's
 * parameter gradients -- its rows of HIG_P_STY_EMB_W included -- are enqueued and `comm_stream` waits for them).  Events
 * only: capturable into a hipGraph together with the collectives its hook enqueues. */
typedef void (*hig_layer_hook)(void* user, int32_t layer);
int hig_denoiser_bwd_bf16_hooked(const hig_dims* dims, const void* const* params, const void* const* params16, const float* x,
                                 const int64_t* t, const int64_t* length, const float* xf_out, const void* textctx,
                                 const void* workspace, const float* dout, void* const* grads, float* dx, float* dxf_proj,
                                 float* dxf_out, void* bwd_workspace, hig_stream_t stream, hig_layer_hook hook, void* hook_user,
                                 hig_stream_t comm_stream);
/* Pieces of the above (unit-testable).  hig_ln_bwd_bf16: hig_ln_bwd with bf16 upstream gradients `da`, rows `x` bf16 or fp32
 * (x_f32), residual / result bf16 or fp32 (dx_f32; `res` has the type of `dx`), and the LayerNorm statistics recomputed
 * from x (no `stats` argument).  partial: hig_ln_bwd_partial_floats(rows, n, rows_per_sample) floats. */
int hig_ln_bwd_bf16(const void* da, int64_t ldda, const void* x, int32_t x_f32, int64_t ldx, const float* gamma,
                    const float* beta, const float* ss, int64_t ss_ld, int32_t ss_shift_off, int32_t mod_silu,
                    const void* res, int64_t ldr, void* dx, int32_t dx_f32, int64_t lddx, int64_t rows, int32_t n,
                    int32_t rows_per_sample, float* dgamma, float* dbeta, float* dss, int64_t dss_ld, float* partial,
                    hig_stream_t stream);
/* hig_linattn_apply_bwd / hig_linattn_ctx_bwd with bf16 dY, Q, dQ / K, V, dK, dV (A, dA, kstat fp32); head dim 64 / 128. */
int hig_linattn_apply_bwd_bf16(const void* dY, int64_t lddy, const void* Q, int64_t ldq, const float* A, void* dQ,
                               int64_t lddq, float* dA, int32_t B, int32_t rows, int32_t H, int32_t hd, float* scratch,
                               hig_stream_t stream);
int hig_linattn_ctx_bwd_bf16(const float* dA, const float* A, const void* K, const void* V, int64_t ld, const float* kstat,
                             const int64_t* length, void* dK, void* dV, int64_t ldd, int32_t B, int32_t rows, int32_t H,
                             int32_t hd, hig_stream_t stream);
/* out[j] = sum_i x[i][j] over bf16 rows (bias gradients); n, ldx multiples of 8; partial as for hig_colsum. */
int hig_colsum_bf16(const void* x, int64_t ldx, int64_t rows, int32_t n, float* out, float* partial, hig_stream_t stream);
/* n <= 12 bf16 transposes in one launch: dsts[m] (cols x rows, leading dimension ldd[m] >= rows rounded up to 8) =
 * srcs[m] (rows x cols, leading dimension lds[m])^T.  cols, lds, ldd multiples of 8; the pointer / extent arrays are HOST
 * arrays read before return.  Columns [rows, round_up(rows, 8)) of a destination row receive zeros. */
int hig_transpose_bf16_batch(int32_t n, const void* const* srcs, const int64_t* lds, void* const* dsts, const int64_t* ldd,
                             const int32_t* rows, const int32_t* cols, hig_stream_t stream);
int hig_transpose_bf16(const void* src, int64_t ld, int32_t rows, int32_t cols, void* dst, int64_t ldd, hig_stream_t stream);
/* f = gelu(z) (exact erf form) elementwise on bf16; dst = float(src) elementwise.  n % 8 == 0. */
int hig_gelu_bf16(const void* z, void* f, int64_t n, hig_stream_t stream);
int hig_cast_f32(const void* src, float* dst, int64_t n, hig_stream_t stream);
/* dst (rows, ld_dst) bf16 = src (rows, ld_src) fp32 rounded, columns [cols, ld_dst) zero: the F-wide rows (motion features x,
 * the loss gradient d(out)) as operands of the bf16 matrix kernels in the bf16-storage backward -- the reference's
 * joint_embed / out Linear layers (transformer.py:343,425) under autograd.  ld_dst a multiple of 8, dst 16-byte aligned. */
int hig_cast_pad_bf16(const float* src, int64_t ld_src, int64_t rows, int32_t cols, void* dst, int64_t ld_dst, hig_stream_t stream);

/* Weight gradient of an nn.Linear over bf16 rows WITHOUT transposed operand copies (csrc/wgrad16.hip): dW (J, K) fp32 dense =
 * dC^T . act, dbias (J) fp32 (nullable) = column sums of dC, from dC (rows, J) and act (rows, K) bf16 row-major.  Row chunks
 * land row-major in LDS by DMA, ds_read_b64_tr_b16 supplies the matrix-core operands; the rows are split over `splits`
 * workgroup slices (0 = the library's rule) whose partial tiles are summed in split order (deterministic).  J, K and the
 * leading dimensions multiples of 8, 16-byte aligned buffers.  slabs: hig_wgrad_bf16_scratch_floats(J, K, splits) floats. */
int64_t hig_wgrad_bf16_scratch_floats(int32_t J, int32_t K, int32_t splits);
int hig_wgrad_bf16(const void* dC, int64_t ldd, const void* act, int64_t ldx, int64_t rows, int32_t J, int32_t K, float* dW,
                   float* dbias, int32_t splits, float* slabs, int64_t slab_floats, hig_stream_t stream);

/* Backward of hig_denoiser_fwd(training=1) for d(out) = dout.  Writes (does not accumulate)
 * every entry of `grads` (same table layout as params; NULL entries are skipped is NOT
 * supported -- all must be valid), dx (B,T,F) or NULL, dxf_proj (B,4d), dxf_out (B,N,Lt).
 * Streams: the weight-gradient GEMMs are forked onto a second stream that the library owns (one per
 * host thread and device, created on first use -- so call it once eagerly before capturing it into a
 * hipGraph) and joined back into `stream` before the function returns: callers still order everything
 * through `stream` alone.  Environment HIG_BWD_OVERLAP=0 keeps every launch on `stream`. */
int hig_denoiser_bwd(const hig_dims* dims, const void* const* params, const float* x,
                     const int64_t* t, const int64_t* length, const float* xf_out,
                     const void* textctx, const void* workspace, const float* dout,
                     void* const* grads, float* dx, float* dxf_proj, float* dxf_out,
                     void* bwd_workspace, hig_stream_t stream);
int64_t hig_bwd_workspace_bytes(const hig_dims* dims);
/* The same backward with a per-layer completion hook, for overlapping the data-parallel gradient exchange with the
 * backward (the reference's DDP reducer does this with buckets, tools/train.py:77-82).  After the launches of decoder
 * layer l (l = L-1 ... 0) are enqueued -- including that layer's rows of the stacked stylization `emb_layers.1.weight`
 * gradient, which this variant computes per layer instead of once at the end -- `comm_stream` (optional) is made to
 * wait for them and `hook(user, l)` is called on the host: every entry of `grads` that belongs to layer l, and rows
 * [l * nsty * 2d, (l+1) * nsty * 2d) of HIG_P_STY_EMB_W, are final once `comm_stream` gets there.  The remaining
 * (global) gradients are final when the call's launches on `stream` have completed.  Events only: capturable into a hipGraph
 * together with what the hook enqueues, once the library's streams / events exist (an eager warm-up call creates them). */
/* (hig_layer_hook: declared above, next to hig_denoiser_bwd_bf16_hooked) */
int hig_denoiser_bwd_hooked(const hig_dims* dims, const void* const* params, const float* x,
                            const int64_t* t, const int64_t* length, const float* xf_out,
                            const void* textctx, const void* workspace, const float* dout,
                            void* const* grads, float* dx, float* dxf_proj, float* dxf_out,
                            void* bwd_workspace, hig_stream_t stream, hig_layer_hook hook, void* hook_user,
                            hig_stream_t comm_stream);

/* ------------------------------------------------------------------------------------------
 * Text head: the trainable part of MotionTransformer.encode_text after CLIP
 * (transformer.py:324-340, 389-397): text_pre_proj -> L_text post-norm nn.TransformerEncoderLayer
 * (self-attention over the N tokens, no mask, 1/sqrt(hd) scaling; FFN with exact GELU; dropout 0)
 * -> text_ln -> xf_out; xf_proj = text_proj(xf_out[b, eot[b]]).  Batch-first (B, N, .) rows.
 * Parameter table: HIG_T_* then HIG_TL_* per layer, nn.MultiheadAttention / nn.Linear layouts.
 * ---------------------------------------------------------------------------------------- */
typedef struct hig_text_dims {
  int32_t B, N, W;  /* samples, tokens (77), CLIP width (512) */
  int32_t Lt, H, ff, L, E; /* text_latent_dim, text_num_heads, text_ff_size, num_text_layers, 4*latent_dim */
  int32_t prec;     /* HIG_PREC_* of the GEMM products */
} hig_text_dims;
enum {
  HIG_T_PRE_W = 0, HIG_T_PRE_B, /* text_pre_proj (Lt, W); both NULL = nn.Identity (needs W == Lt) */
  HIG_T_LN_W, HIG_T_LN_B,       /* text_ln */
  HIG_T_PROJ_W, HIG_T_PROJ_B,   /* text_proj.0 (E, Lt) */
  HIG_T_NGLOBAL
};
enum {
  HIG_TL_IN_W = 0, HIG_TL_IN_B,   /* self_attn.in_proj_weight (3Lt, Lt) / in_proj_bias */
  HIG_TL_OUT_W, HIG_TL_OUT_B,     /* self_attn.out_proj */
  HIG_TL_N1_W, HIG_TL_N1_B,       /* norm1 */
  HIG_TL_FF1_W, HIG_TL_FF1_B,     /* linear1 (ff, Lt) */
  HIG_TL_FF2_W, HIG_TL_FF2_B,     /* linear2 (Lt, ff) */
  HIG_TL_N2_W, HIG_TL_N2_B,       /* norm2 */
  HIG_TL_NLAYER
};
int64_t hig_text_head_workspace_bytes(const hig_text_dims* dims, int training);
int64_t hig_text_head_bwd_workspace_bytes(const hig_text_dims* dims);
/* clip_out (B, N, W): CLIP's ln_final output, batch-first; eot[b] = index of the EOT token
 * (text.argmax(-1), transformer.py:395).  Writes xf_out (B, N, Lt) and xf_proj (B, E).
 * training != 0 keeps every layer's activations in `workspace` for hig_text_head_bwd. */
int hig_text_head_fwd(const hig_text_dims* dims, const void* const* params, const float* clip_out,
                      const int64_t* eot, float* xf_out, float* xf_proj, void* workspace, int training,
                      hig_stream_t stream);
/* Adjoint for upstream d(xf_out), d(xf_proj) (either may be NULL = zero).  Writes (not accumulates)
 * every parameter gradient through `grads` (same table order) and d(clip_out) if dclip != NULL. */
int hig_text_head_bwd(const hig_text_dims* dims, const void* const* params, const float* clip_out,
                      const int64_t* eot, const float* xf_out, const void* workspace, const float* dxf_out,
                      const float* dxf_proj, void* const* grads, float* dclip, void* bwd_workspace,
                      hig_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Evaluator feature extraction: the two classifiers that score generated pairs,
 * `MotionEncoder` (interaction_transformer.py:641-741: class logits + the pooled feature FID /
 * diversity are computed on) and `MotionConsistencyEvalModel` (:743-829: real/fake logits from a
 * learned [cls] token).  Inference only (the reference calls them under no_grad,
 * datasets/evaluator.py:479-493).  Both embed each person like the two-person denoiser (token 0 =
 * joint_embed2 of the init-pose row's first 4 features, token t >= 1 = joint_embed1 +
 * sequence_embedding[t-1]), put the two persons one after the other on the token axis
 * ([cls,] person 1's T tokens, person 2's T tokens), run a post-norm nn.TransformerEncoder with
 * tokens t >= length[b] of either person excluded as keys, then
 *   MotionEncoder:  o = out2(token 0) / out1(other tokens);  feature = mean of o over the valid
 *                   tokens of both persons;  logits = fin_proj(feature)
 *   Consistency  :  logits = cls_output(encoder output of the [cls] token).
 * ---------------------------------------------------------------------------------------- */
typedef struct hig_eval_dims {
  int32_t B, T, F;         /* pairs, tokens per person (frames + 1), input_feats (dim_pose - 4) */
  int32_t d, H, ff, L;     /* latent_dim, num_heads (head dim in {8,16,32,64,128}), ff_size, num_layers */
  int32_t C;               /* class_num: logits per pair */
  int32_t cls;             /* 0 = MotionEncoder, 1 = MotionConsistencyEvalModel ([cls] token first) */
  int32_t prec;            /* HIG_PREC_* of the GEMM products */
} hig_eval_dims;
enum {
  HIG_EV_SEQ_EMB = 0,                /* sequence_embedding (num_frames, d) */
  HIG_EV_JOINT1_W, HIG_EV_JOINT1_B,  /* joint_embed1 (d, F) */
  HIG_EV_JOINT2_W, HIG_EV_JOINT2_B,  /* joint_embed2 (d, 4) */
  HIG_EV_OUT1_W, HIG_EV_OUT1_B,      /* out1 (d, d)            -- MotionEncoder only, else NULL */
  HIG_EV_OUT2_W, HIG_EV_OUT2_B,      /* out2 (d, d)            -- MotionEncoder only, else NULL */
  HIG_EV_HEAD_W, HIG_EV_HEAD_B,      /* fin_proj.0 / cls_output.0 (C, d) */
  HIG_EV_CLS_IN,                     /* cls_input (d)          -- consistency model only, else NULL */
  HIG_EV_NGLOBAL
};
/* Table = HIG_EV_NGLOBAL globals, then per layer the HIG_TL_* block of the text head (same
 * nn.TransformerEncoderLayer parameters). */
int64_t hig_eval_encoder_workspace_bytes(const hig_eval_dims* dims);
/* x1, x2 (B, T, F): the two persons; length (B): valid tokens per person.  Writes logits (B, C)
 * and, for the MotionEncoder (cls == 0), feature (B, d) (may be NULL). */
int hig_eval_encoder_fwd(const hig_eval_dims* dims, const void* const* params, const float* x1,
                         const float* x2, const int64_t* length, float* logits, float* feature,
                         void* workspace, hig_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Input pipeline: batch assembly from a device-resident bank of motions.
 * Reference arithmetic: Text2MotionMulDataset.__getitem__, datasets/mul_dataset.py:203-209.
 * out[r][t][:] = normalise(bank[seq_off[r] + frame_ix[r][t] * F ...]):  token 0 (the init-pose row)
 * is (x - init_mean) / init_std on its first 4 features and raw elsewhere; tokens >= 1 are
 * (x - mean) / std.  stats = [mean F | std F | init_mean 4 | init_std 4], float (stats_f64 == 0) or
 * double (numpy's promotion when the saved statistics are float64): bit-identical to numpy either way.
 * ---------------------------------------------------------------------------------------- */
int hig_gather_frames(const float* bank, const int64_t* seq_off, const int32_t* frame_ix,
                      const void* stats, int32_t stats_f64, int32_t rows, int32_t T, int32_t F,
                      float* out, hig_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Post-sampling joint recovery: de-normalise a generated motion and integrate it to
 * world-space joint positions.  Reference: tools/visualization.py:146-152 (x * std + mean, init row on
 * 4 features), utils/motion_process.py:362-382 recover_root_rot_pos, :418-462 recover_from_ric2
 * (one person; both persons are rows of the same call), utils/quaternion.py qinv / qrot.
 * motion (rows, T + 1, F): T body frames + one init-state row [x, z, quat_w, quat_y, ...], first
 * (init_first = 1, the sampler's token order) or last (0, recover_from_ric2's order).
 * stats = [mean F | std F | init_mean 4 | init_std 4] floats, or NULL if already de-normalised.
 * pos (rows, T, joints, 3).
 * ---------------------------------------------------------------------------------------- */
int hig_recover_joints(const float* motion, const float* stats, int32_t rows, int32_t T, int32_t F,
                       int32_t joints, int32_t init_first, float* pos, hig_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel entry points (unit-testable pieces of the above).
 * ---------------------------------------------------------------------------------------- */

/* Generic fused GEMM  C[i][j] = epi( sum_r xf(X)[i][r] * Y[j][r] ).
 * x_rs / y_rs: 0 = operand stored reduce-contiguous (X[i*ldx + r]); 1 = reduce-slow
 * (X[r*ldx + i]).  xf_* select the prologue applied to the ACTIVATION operand (X when
 * xf_on_y == 0, Y otherwise) -- the fused LayerNorm / stylization modulation / SiLU of
 * StylizationBlock.forward (transformer.py:81-85) and the pre-attention norms. */
#define HIG_XF_NONE 0
#define HIG_XF_LN 1          /* (x-mean)*rstd*gamma+beta */
#define HIG_XF_LN_MOD_SILU 2 /* silu(LN(x)*(1+scale)+shift), scale/shift per sample */
#define HIG_XF_SILU 3
#define HIG_EPI_NONE 0
#define HIG_EPI_BIAS 1
#define HIG_EPI_BIAS_GELU 2 /* out = gelu(acc+bias); aux (if set) receives acc+bias */
#define HIG_EPI_BIAS_RES 3  /* out = res + acc + bias */
#define HIG_EPI_BIAS_POS 4  /* out = acc + bias + pos[i % T] (joint_embed + sequence_embedding) */
#define HIG_EPI_RES 5       /* out = res + acc */
#define HIG_EPI_DGELU 6     /* out = acc * gelu'(aux)   (hig_gemm_bf16: the pre-activation z is passed as `res`) */
#define HIG_EPI_BIAS_SILU 7      /* out = silu(acc + bias)          (hig_gemm_bf16 only) */
#define HIG_EPI_BIAS_RES_SILU 8  /* out = silu(res + acc + bias)    (hig_gemm_bf16 only) */
/* Zero-initialise the descriptor (memset) and set what the chosen xf / epi need: optional pointers are tested
 * against NULL. */
typedef struct hig_gemm_desc {
  const float* X; int64_t ldx; int32_t x_rs;
  const float* Y; int64_t ldy; int32_t y_rs;
  float* C; int64_t ldc;
  int32_t I, J, R;
  int32_t xf, xf_on_y, epi, prec;
  const float* bias;                 /* [J] */
  const float* res; int64_t ldr;     /* [I][J] */
  float* aux; int64_t ldaux;         /* [I][J] */
  const float* stats;                /* [rows][2] mean, rstd of the activation rows */
  const float* gamma; const float* beta; /* [features] */
  const float* ss; int64_t ss_ld; int32_t ss_shift_off; int32_t rows_per_sample;
  const float* pos; int64_t ldpos; int32_t T;
  int32_t pos_shift;                 /* EPI_BIAS_POS adds pos[(i % T) - pos_shift]; rows with a negative index get none */
  float* xcolsum;                    /* optional, x_rs == 1 + fp32 products + I % 4 == 0: xcolsum[i] = sum_r X[r][i] -- the
                                        bias gradient that goes with a weight gradient dW = dC^T . act (X = dC) */
  /* LayerNorm folded into the NEXT GEMM (fp32 storage, inference forward; all NULL otherwise; both operands reduce-contiguous,
   * 16-byte aligned, no split):
   * row_stats_out (EPI_BIAS_RES, J % 64 == 0): also write, per 64-column panel of the output rows, (sum, sum of squared
   *   deviations from the panel's own mean), [I][J / 64][2] fp32.
   * row_stats_in + ln_colsum (EPI_BIAS, R % 128 == 0): X holds UN-normalised rows with their statistics [I][R / 64][2] in
   *   row_stats_in, Y is W' = gamma (.) W, ln_colsum[j] = sum_r W'[j][r], bias[j] = b[j] + sum_r beta[r] W[j][r]:
   *   C = rstd (X W'^T) - rstd mean ln_colsum + bias  ==  LayerNorm(X) W^T + b   (transformer.py:108-110,144). */
  float* row_stats_out;
  const float* row_stats_in;
  const float* ln_colsum;
} hig_gemm_desc;
int hig_gemm(const hig_gemm_desc* g, hig_stream_t stream);
/* hig_gemm with a scratch for the SPLIT TAIL of the exact-fp32 kernel, as hig_denoiser_fwd runs its GEMMs: when the
 * 64x64 tiles do not fill whole rounds of the 256 CUs (M = 12 544 never does), the tiles of the last, partly filled
 * round are cut into 2-8 slices of the reduce range; a slice parks its accumulators in `ws`, draws a ticket, and the
 * workgroup drawing a tile's last ticket sums the slices in slice order and applies the epilogue (deterministic; no
 * workgroup waits for another).  ws: hig_gemm_tail_ws_bytes() bytes, 16-byte aligned, its first 1024 bytes zero
 * before the first launch (every launch leaves them zero); one scratch per stream in flight.  ws == NULL: hig_gemm. */
int64_t hig_gemm_tail_ws_bytes(void);
int hig_gemm_ws(const hig_gemm_desc* g, void* ws, int64_t ws_bytes, hig_stream_t stream);
/* Diagnostic (tools/gemm_stamps.py): while buf != NULL, thread 0 of every workgroup (< 4096) of the exact-fp32 kernel
 * writes s_memtime stamps of its second tile's phases to buf[block * 8 + k].  Never set during a timed run. */
int hig_gemm_debug_stamps(void* buf);
/* The same contraction with the reduce range split over `splits` partial outputs in `slabs` and a deterministic
 * (fixed-order, no float atomics) slab reduction: how hig_denoiser_bwd runs its weight gradients dW = dC^T . act over
 * the M = B*T rows (autograd of nn.Linear).  EPI_NONE, dense C (ldc == J), I*J % 4 == 0.  splits == 0: the
 * library's own rule (tiles x splits fills the chip).  slabs: >= hig_gemm_split_scratch_floats(g, splits) floats. */
int64_t hig_gemm_split_scratch_floats(const hig_gemm_desc* g, int32_t splits);
int hig_gemm_split(const hig_gemm_desc* g, int32_t splits, float* slabs, int64_t slab_floats, hig_stream_t stream);

/* bf16-STORAGE GEMM (hig_dims.storage == HIG_STORE_BF16):  C[i][j] = epi( sum_r X[i][r] * Y[j][r] ) with X (I, ldx)
 * and Y (J, ldy) bf16, both reduce-contiguous (activations x nn.Linear weight (out, in)), fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16, operand tiles DMA-ed global -> LDS (no register hop, no conversion).  R % 32 == 0,
 * operands 16-byte aligned, ldx / ldy multiples of 8.  bias fp32 [J]; res / C bf16 or fp32 ([I][J], any ld; rows that
 * are not 16-byte aligned fall back to element stores).  epi: HIG_EPI_NONE, _BIAS, _BIAS_GELU, _BIAS_RES, _BIAS_SILU,
 * _BIAS_RES_SILU, and the data-gradient forms _RES (out = res + acc) and _DGELU (out = acc * gelu'(res)). */
typedef struct hig_gemm16_desc {
  const void* X; int64_t ldx;
  const void* Y; int64_t ldy;
  void* C; int64_t ldc; int32_t c_f32;          /* C stored as fp32 (1) or bf16 (0) */
  int32_t I, J, R;
  int32_t epi;
  const float* bias;
  const void* res; int64_t ldr; int32_t res_f32;
  /* LayerNorm folded into the NEXT GEMM (weight-stationary kernel only: >= 2048 rows, R == 512 or 1024; all NULL otherwise):
   * row_stats_out (EPI_BIAS_RES, J == R): also write, per 128-column panel of the bf16-rounded output rows, (sum, sum of
   * squared deviations from the panel's own mean), [I][J / 128][2] fp32 -- the consumer (R = that J) merges the panels into the row's
   * mean and variance without forming E[x^2] - mean^2, so a large common offset of a row does not cost it its variance.  row_stats_in + ln_colsum (EPI_BIAS): X holds UN-normalised rows whose statistics are in
   * row_stats_in; Y must be W' = bf16(gamma (.) W), ln_colsum[j] = sum_r float(W'[j][r]), bias[j] = b[j] + sum_r beta[r] W[j][r]:
   * C = rstd (X W'^T) - rstd mean ln_colsum + bias  ==  LayerNorm(X) W^T + b  (transformer.py:108-110,144). */
  float* row_stats_out;
  const float* row_stats_in;
  const float* ln_colsum;
  /* EPI_BIAS_GELU only (nullable): the pre-activation acc + bias as bf16 as well, from the same launch -- FFN linear1 of the
   * training forward keeps both z and gelu(z) (transformer.py:168).  16-byte aligned, ldaux % 8 == 0. */
  void* aux; int64_t ldaux;
} hig_gemm16_desc;
int hig_gemm_bf16(const hig_gemm16_desc* g, hig_stream_t stream);
/* hig_gemm_bf16 with the reduce range split over `splits` fp32 slabs and a deterministic (fixed-order) slab reduction: how
 * hig_denoiser_bwd_bf16 runs dW = dC^T . act (X = dC^T (J_out, rows), Y = act^T (K_in, rows), both reduce-contiguous over
 * the rows).  EPI_NONE, fp32 C with ldc == J, R % 64 == 0, (R / 64) % splits == 0; splits == 0: the library's rule.
 * slabs: >= hig_gemm_bf16_split_scratch_floats(g, splits) floats. */
int64_t hig_gemm_bf16_split_scratch_floats(const hig_gemm16_desc* g, int32_t splits);
int hig_gemm_bf16_split(const hig_gemm16_desc* g, int32_t splits, float* slabs, int64_t slab_floats, hig_stream_t stream);
/* Diagnostics (tools/gemm_ws16_stamps.py): while buf != NULL, thread 0 of every workgroup of the
 * tiled bf16 kernel (HIG_BF16_DBG & 16; buf[block * 8 + k], block < 4096) / of the weight-stationary kernel
 * (buf[block * 16 + k], 256 blocks) writes s_memtime stamps of its phases.  This pointer is the library's only mutable
 * global state besides what hig_shutdown() releases; it is NULL unless a tool sets it, and is never set during a timed run. */
int hig_gemm_bf16_debug_stamps(void* buf);
int hig_gemm_ws16_debug_stamps(void* buf);
int hig_gemm_wsp16_debug_stamps(void* buf);   /* gemm_wsp16.hip: buf[block * 16 + k], 256 blocks */
int hig_gemm_wsp32_debug_stamps(void* buf);   /* gemm_wsp32.hip: buf[block * 16 + k] matrix wave 0, buf[4096 + block * 16 + k] service wave 4 */
/* Launches served by the exact-fp32 weight-stationary kernel (gemm_wsp32.hip) since the library was loaded: hig_gemm routes
 * K = 512 / 1024 products with >= 2048 rows there (HIG_F32_WSP=0 switches it off); a test reads the difference around a call. */
int64_t hig_gemm_wsp32_launches(void);
int hig_wgrad16_debug_stamps(void* buf);      /* wgrad16.hip (see there): 8192 x 8 bytes */
/* the same for hig_linattn_apply_sty_mm16: 8 stamps per workgroup (see linattn16.hip) */
int hig_linattn16_debug_stamps(void* buf);
/* dst[i] = bf16(src[i]) (round to nearest even): builds the bf16 shadow of the flat fp32 parameter buffer. */
/* joint_embed + sequence_embedding of the bf16-storage forward (transformer.py:418-419) as its own kernel pair:
 * out[m][:] = bf16( x[m][:F] . W^T + bias + pos[(m % T) - pos_shift] ), x fp32 with F (e.g. 150, 263) features per
 * row, W fp32 (d, F) padded / rounded to bf16 into `w_scratch` (hig_joint_embed_bf16_scratch_bytes) by the call,
 * rows with a negative positional index get no positional term.  d % 128 == 0, F <= 512, out 16-byte aligned, ldo % 8 == 0.
 * hig_joint_embed_bf16_w takes the weight already padded and rounded: (d, Fp) bf16, Fp = F rounded up to a multiple of 32,
 * zeros beyond column F (kept by the caller next to its bf16 weight shadow: no per-call padding launch). */
int64_t hig_joint_embed_bf16_scratch_bytes(int32_t F, int32_t d);
int hig_joint_embed_bf16_w(const float* x, int64_t M, int32_t F, const void* w_padded, const float* bias, const float* pos,
                           int64_t ldpos, int32_t T, int32_t pos_shift, void* out, int64_t ldo, int32_t d, hig_stream_t stream);
int hig_joint_embed_bf16(const float* x, int64_t M, int32_t F, const float* W, const float* bias, const float* pos,
                         int64_t ldpos, int32_t T, int32_t pos_shift, void* out, int64_t ldo, int32_t d,
                         void* w_scratch, hig_stream_t stream);
int hig_cast_bf16(const float* src, void* dst, int64_t n, hig_stream_t stream);

/* bf16-storage row kernel: out (bf16) = LN(x) * gamma + beta, and with ss != NULL the stylization front
 * silu(LN(x) * (1 + scale) + shift) (scale = ss[b][0..n), shift = ss[b][ss_shift_off ..), b = row / rows_per_sample).
 * x bf16 (x_f32 == 0) or fp32 (x_f32 == 1); n % 8 == 0, n <= 1024. */
int hig_ln_bf16(const void* x, int32_t x_f32, int64_t ldx, int64_t rows, int32_t n, const float* gamma,
                const float* beta, const float* ss, int64_t ss_ld, int32_t ss_shift_off, int32_t rows_per_sample,
                void* out, int64_t ldo, hig_stream_t stream);
/* bf16-storage forms of hig_linattn_ctx / hig_linattn_apply (below): K, V, Q, Y bf16; A, kstat fp32; head dim 64 / 128. */
/* At16 (nullable): the context matrices once more, transposed and rounded -- At[l][c] = bf16(A[b][h][c][l]) -- the
 * matrix-core operand of hig_linattn_apply_sty_mm16, hd x hd elements per (b, h) in "fragment-major" order: element (l, c)
 * at ((((l / 32) (hd / 16) + c / 16) 64 + l % 32 + 32 ((c % 16) / 8)) 8 + c % 8 (hig_at16_offset, csrc/hig_common.h), so
 * that one 32 x 16 MFMA operand block is one contiguous KiB. */
int hig_linattn_ctx_bf16(const void* K, const void* V, int64_t ld, int32_t B, int32_t rows, int32_t H, int32_t hd,
                         const int64_t* length, float* A, float* kstat, float* scratch, void* At16, hig_stream_t stream);
int hig_linattn_apply_bf16(const void* Q, int64_t ldq, const float* A, void* Y, int64_t ldy, int32_t B, int32_t rows,
                           int32_t H, int32_t hd, hig_stream_t stream);
/* hig_linattn_ctx_bf16 with the k^T v product on the bf16 matrix cores (csrc/linattn16.hip: softmax_r(K) and V rounded to
 * bf16 for the product, fp32 accumulate / statistics; K and V chunks by LDS-DMA, ds_read_b64_tr_b16 transpose reads).
 * Head dim 64, one workgroup per (sample, head).  Default of the bf16-storage forward (HIG_CTX16=1). */
int hig_linattn_ctx_mm16(const void* K, const void* V, int64_t ld, int32_t B, int32_t rows, int32_t H, int32_t hd,
                         const int64_t* length, float* A, float* kstat, void* At16, hig_stream_t stream);
/* hig_linattn_apply_bf16 followed by the stylization front hig_ln_bf16(ss != NULL) over all H heads, as one kernel:
 * Out = silu( LN_d( softmax_hd(Q) . A ) * (1 + scale) + shift ) (transformer.py:111,116-118 then :81-85); the
 * (rows x d) intermediate never reaches memory.  H in {4, 8}, head dim 64 / 128; rows_per_sample == rows. */
int hig_linattn_apply_sty_bf16(const void* Q, int64_t ldq, const float* A, const float* gamma, const float* beta,
                               const float* ss, int64_t ss_ld, int32_t ss_shift_off, void* Out, int64_t ldo,
                               int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream);
/* The same operator with the hd x hd products on the bf16 matrix cores (csrc/linattn16.hip): softmax(Q) and A are rounded
 * to bf16 for the product (fp32 accumulate), a workgroup owns 32 whole rows, Q / Out move as whole rows through LDS
 * (DMA), the context operands go global -> registers.  At16: the transposed bf16 context matrices in the order
 * hig_linattn_ctx_bf16 / hig_linattn_ctx_mm16 write them.  Head dim 64 / 128 with 4 or 8 heads.  Default of the
 * bf16-storage forward (HIG_FUSE_APPLY=2). */
/* hig_linattn_apply_sty_mm16 + the stylization block's output projection + the residual update as ONE kernel (small batches:
 * a launch costs ~4.4 us on this part before its first instruction, and the projection is bound by the CU's L2 fetch rate
 * either way):  h[rows] += silu( LN( softmax_hd(Q) . A ) (1 + scale) + shift ) . W^T + bias   (transformer.py:111-118, :81-86).
 * W_frag: the (d, d) bf16 weight in matrix-core operand order: element (j, r) at ((j / 32) (d / 16) + r / 16) 512 + (j % 32 +
 * 32 ((r % 16) / 8)) 8 + r % 8 (hig_weight_frag16 builds it).  h bf16, updated in place.  stats (nullable): (sum, centred
 * sum of squares) of the new rows per 128-column panel, [B rows][4][2] fp32 -- what hig_gemm16_desc.row_stats_in consumes.  d = 512
 * with 8 heads of 64. */
int hig_attn_out16(const void* Q, int64_t ldq, const void* At16, const float* gamma, const float* beta, const float* ss,
                   int64_t ss_ld, int32_t ss_shift_off, const void* W_frag, const float* bias, void* h, int64_t ldh,
                   float* stats, int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream);
/* hig_ln_bf16 (stylization front) + the stylization-out projection + the residual update as one kernel, for a block whose
 * input rows Y are in memory (the FFN block):  h[rows] += silu( LN(Y[rows]) (1 + scale) + shift ) . W^T + bias
 * (transformer.py:81-86).  Y, h bf16 (B rows, 512); W_frag / stats as in hig_attn_out16; rows_per_sample == rows. */
int hig_rows_out16(const void* Y, int64_t ldy, const float* gamma, const float* beta, const float* ss, int64_t ss_ld,
                   int32_t ss_shift_off, const void* W_frag, const float* bias, void* h, int64_t ldh, float* stats,
                   int32_t B, int32_t rows, int32_t d, hig_stream_t stream);
/* Y_frag[((j / 32) (R / 16) + r / 16) 512 + (j % 32 + 32 ((r % 16) / 8)) 8 + r % 8] = Y[j][r] for a (J, R) row-major bf16
 * matrix with leading dimension ldy (J % 32 == 0, R % 16 == 0): a 32 x 16 MFMA operand block = 64 lanes x 8 elements = 1 KiB. */
int hig_weight_frag16(const void* Y, int64_t ldy, int32_t J, int32_t R, void* Y_frag, hig_stream_t stream);
int hig_linattn_apply_sty_mm16(const void* Q, int64_t ldq, const void* At16, const float* gamma, const float* beta,
                               const float* ss, int64_t ss_ld, int32_t ss_shift_off, void* Out, int64_t ldo,
                               int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream);
/* ... with the attention output y = softmax(q) . A itself (bf16 rows, the input of the LayerNorm) as a second output: the
 * training forward keeps y for the backward of the stylization block (transformer.py:111-118 + :81-85).  Yout NULL: the above. */
int hig_linattn_apply_sty_mm16_y(const void* Q, int64_t ldq, const void* At16, const float* gamma, const float* beta,
                                 const float* ss, int64_t ss_ld, int32_t ss_shift_off, void* Out, int64_t ldo, void* Yout, int64_t ldy,
                                 int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream);
/* The same fused kernel with fp32 storage (a standalone entry point: hig_denoiser_fwd runs hig_linattn_apply +
 * hig_ln_mod_silu, which measured equal).  Q, Out fp32, 16-byte aligned. */
int hig_linattn_apply_sty(const float* Q, int64_t ldq, const float* A, const float* gamma, const float* beta,
                          const float* ss, int64_t ss_ld, int32_t ss_shift_off, float* Out, int64_t ldo,
                          int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream);

/* Row statistics for LayerNorm: stats[m] = (mean, rstd) of x[m, :n], eps = 1e-5, biased var. */
int hig_rowstats(const float* x, int64_t ldx, int64_t rows, int32_t n, float* stats,
                 hig_stream_t stream);

/* Front half of StylizationBlock.forward (transformer.py:81-84) fused with the SiLU of its
 * out_layers and with the LayerNorm statistics:  a = silu(LN(x)*(1+scale)+shift), stats = (mean, rstd).
 * scale = ss[b][0..n), shift = ss[b][ss_shift_off ..), b = row / rows_per_sample. */
int hig_ln_mod_silu(const float* x, int64_t ldx, int64_t rows, int32_t n, const float* gamma,
                    const float* beta, const float* ss, int64_t ss_ld, int32_t ss_shift_off,
                    int32_t rows_per_sample, float* a, int64_t lda, float* stats, hig_stream_t stream);

/* Plain LayerNorm y = LN(x)*gamma+beta (eps 1e-5) that also returns stats = (mean, rstd) per row. */
int hig_layernorm(const float* x, int64_t ldx, int64_t rows, int32_t n, const float* gamma,
                  const float* beta, float* y, int64_t ldy, float* stats, hig_stream_t stream);
/* dst[b][:n] = src[b*rows_per_sample + idx[b]][:n] and its adjoint dst[...] += src[b]. */
int hig_gather_rows(const float* src, int64_t ld, int32_t B, int32_t rows_per_sample, const int64_t* idx,
                    int32_t n, float* dst, int64_t ldd, hig_stream_t stream);
int hig_scatter_add_rows(const float* src, int64_t ld, int32_t B, int32_t rows_per_sample,
                         const int64_t* idx, int32_t n, float* dst, int64_t ldd, hig_stream_t stream);

/* Linear ("efficient") attention pieces, transformer.py:110-117 / 146-153.  Channel c of
 * head h lives at column h*hd + c.
 * ctx: k = softmax over the `len[b]` leading rows of each sample (masked rows contribute 0),
 *      A[b,h] = k^T v; kstat[b][col] = (column max, column sum of exp).
 * apply: y = softmax_hd(Q) . A[b,h]. */
int hig_linattn_ctx(const float* K, const float* V, int64_t ld, int32_t B, int32_t rows,
                    int32_t H, int32_t hd, const int64_t* length, float* A, float* kstat,
                    float* scratch, hig_stream_t stream);
/* scratch (optional, hig_linattn_ctx_scratch_floats(B, rows, H, hd) floats): with it, hd 64 / 128 contexts
 * are built from 64-row chunks in parallel (online softmax) and merged; NULL = one workgroup per (b, h). */
int64_t hig_linattn_ctx_scratch_floats(int32_t B, int32_t rows, int32_t H, int32_t hd);
int hig_linattn_apply(const float* Q, int64_t ldq, const float* A, float* Y, int64_t ldy,
                      int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream);
/* The backward kernels split each sample's rows into 64-row chunks (one workgroup each) and
 * combine the chunks' partial sums in a fixed order; `scratch` holds those partials
 * (hig_linattn_bwd_scratch_floats(B, rows, H, hd) floats). */
int64_t hig_linattn_bwd_scratch_floats(int32_t B, int32_t rows, int32_t H, int32_t hd);
int hig_linattn_apply_bwd(const float* dY, int64_t lddy, const float* Q, int64_t ldq,
                          const float* A, float* dQ, int64_t lddq, float* dA, int32_t B,
                          int32_t rows, int32_t H, int32_t hd, float* scratch, hig_stream_t stream);
int hig_linattn_ctx_bwd(const float* dA, const float* A, const float* K, const float* V, int64_t ld,
                        const float* kstat, const int64_t* length, float* dK, float* dV,
                        int64_t ldd, int32_t B, int32_t rows, int32_t H, int32_t hd,
                        float* scratch, hig_stream_t stream);

/* Full softmax attention (no_eff=True), transformer.py:208-227 / 242-262, flash-style (no T x T
 * matrix in memory).  S = q.k/sqrt(hd) (+ -100000 on query rows n >= qlen[b]: the reference puts
 * its mask on the QUERY axis; qlen == NULL for cross attention), W = softmax over the Tk keys,
 * Y = W V; lse[b,h,n] = log sum_m exp(S[n,m]) is kept for the backward, `delta` is scratch
 * (B*H*Tq floats).  Head dim in {8,16,32,64,128}; 64 and 128 run on the matrix cores
 * (v_mfma_f32_32x32x2_f32, exact fp32 products), the smaller ones on the VALU. */
int hig_fullattn_fwd(const float* Q, int64_t ldq, const float* K, const float* V, int64_t ldk,
                     int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t hd, const int64_t* qlen,
                     float* Y, int64_t ldy, float* lse, hig_stream_t stream);
/* Same forward with torch's `src_key_padding_mask` (nn.MultiheadAttention): kpad (B, Tk) bytes,
 * non-zero = that key takes no part in the softmax (NULL = none).  Used by the evaluator encoders
 * (interaction_transformer.py:733,827). */
int hig_fullattn_fwd_kpad(const float* Q, int64_t ldq, const float* K, const float* V, int64_t ldk,
                          int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t hd, const int64_t* qlen,
                          const uint8_t* kpad, float* Y, int64_t ldy, float* lse, hig_stream_t stream);
/* bf16-storage form of hig_fullattn_fwd (inference, hig_dims.storage == HIG_STORE_BF16 with attn_kind FULL): Q, K, V, Y
 * bf16, logits / softmax / accumulation fp32, no log-sum-exp kept.  Head dim 64 or 128. */
int hig_fullattn_fwd_bf16(const void* Q, int64_t ldq, const void* K, const void* V, int64_t ldk, int32_t B, int32_t Tq,
                          int32_t Tk, int32_t H, int32_t hd, const int64_t* qlen, void* Y, int64_t ldy, hig_stream_t stream);
int hig_fullattn_bwd(const float* dY, int64_t lddy, const float* Y, int64_t ldy, const float* Q,
                     int64_t ldq, const float* K, const float* V, int64_t ldk, int32_t B, int32_t Tq,
                     int32_t Tk, int32_t H, int32_t hd, const int64_t* qlen, const float* lse,
                     float* delta, float* dQ, int64_t lddq, float* dK, float* dV, int64_t lddk,
                     hig_stream_t stream);

/* Backward of y = [silu](LN(x)*(1+scale)+shift) w.r.t. x for upstream gradient da, plus the
 * reductions for gamma/beta (over all rows) and scale/shift (per sample):
 *   dx = (res ? res : 0) + LNbwd(...).  partial: [rows/rows_per_sample * splits][4][n]. */
int hig_ln_bwd(const float* da, int64_t ldda, const float* x, int64_t ldx, const float* stats,
               const float* gamma, const float* beta, const float* ss, int64_t ss_ld,
               int32_t ss_shift_off, int32_t mod_silu, const float* res, int64_t ldr,
               float* dx, int64_t lddx, int64_t rows, int32_t n, int32_t rows_per_sample,
               float* dgamma, float* dbeta, float* dss, int64_t dss_ld, float* partial,
               hig_stream_t stream);
int64_t hig_ln_bwd_partial_floats(int64_t rows, int32_t n, int32_t rows_per_sample);

/* dst[c][r] = src[r][c], optionally with LayerNorm applied to the source rows first
 * (stats = (mean, rstd) per source row, gamma/beta per source column; NULL = plain transpose). */
int hig_transpose(const float* src, int64_t ld, int32_t rows, int32_t cols, float* dst, int64_t ldd,
                  const float* stats, const float* gamma, const float* beta, hig_stream_t stream);

/* n <= 12 dense transposes in one launch: dsts[m] (cols x rows) = srcs[m] (rows x cols)^T.  The pointer
 * and extent arrays are HOST arrays read before return (the W -> W^T copies of one layer's backward). */
int hig_transpose_batch(int32_t n, const float* const* srcs, float* const* dsts, const int32_t* rows,
                        const int32_t* cols, hig_stream_t stream);

/* out[j] = sum_i x[i][j]  (bias gradients).  partial: [hig_colsum_chunks(rows)][n] floats
 * (at most HIG_COLSUM_CHUNKS row chunks). */
#define HIG_COLSUM_CHUNKS 512
int hig_colsum_chunks(int64_t rows);
int hig_colsum(const float* x, int64_t ldx, int64_t rows, int32_t n, float* out, float* partial,
               hig_stream_t stream);

/* timestep_embedding (transformer.py:15-32): out[b] = [cos(t*f), sin(t*f)], fp32. */
int hig_timestep_embedding(const int64_t* t, int32_t B, int32_t d, float* out, hig_stream_t s);
/* the same values rounded to bf16 (input of the bf16-storage forward's time_embed.0 GEMM) */
int hig_timestep_embedding_bf16(const int64_t* t, int32_t B, int32_t d, void* out16, hig_stream_t s);

/* ------------------------------------------------------------------------------------------
 * DDPM arithmetic (gaussian_diffusion.py).  `tab` is a device table of 6 x nsteps fp32 rows:
 * sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod, sqrt_recip_alphas_cumprod,
 * sqrt_recipm1_alphas_cumprod, posterior_mean_coef1, posterior_mean_coef2, followed by
 * posterior_log_variance_clipped (7 rows), each float64-computed then cast like
 * _extract_into_tensor(...).float() (gaussian_diffusion.py:1137-1150).
 * ---------------------------------------------------------------------------------------- */
#define HIG_TAB_ROWS 7
/* q_sample, gaussian_diffusion.py:399-417: xt = sqrt_ac[t]*x0 + sqrt_1m_ac[t]*noise */
int hig_q_sample(const float* x0, const float* noise, const int64_t* t, const float* tab,
                 int32_t nsteps, int32_t B, int64_t per_sample, float* xt, hig_stream_t s);
/* p_sample (EPSILON / FIXED_SMALL, clip_denoised=False), gaussian_diffusion.py:443-544,606-666:
 * x_prev = coef1*x0hat + coef2*x + [t!=0]*exp(0.5*logvar)*z;  may run in place (x_prev == x). */
int hig_p_sample_step(const float* x, const float* eps, const float* z, const int64_t* t,
                      const float* tab, int32_t nsteps, int32_t B, int64_t per_sample,
                      float* x_prev, float* pred_xstart /* nullable */, hig_stream_t s);
/* t[b] -= 1 on the device (the sampling loop's step counter, graph-replayable). */
int hig_dec_timesteps(int64_t* t, int32_t B, hig_stream_t s);
/* DDPMTrainer.backward_G (ddpm_trainer.py:172-178): loss = sum_bt mask*mean_f (p-t)^2 / sum mask
 * with mask[b][t] = t < length[b];  dpred = d loss / d pred.  scratch: 2*B floats + loss. */
int hig_masked_mse(const float* pred, const float* target, const int64_t* length, int32_t B,
                   int32_t T, int32_t F, float* loss /* device scalar */, float* dpred,
                   float* scratch, hig_stream_t s);
/* clip_grad_norm_(0.5) + Adam (ddpm_trainer.py:184-185,222) on a flat fp32 buffer:
 * g *= inv_world (DDP mean); gnorm = ||g||; g *= min(1, max_norm/(gnorm+1e-6)); Adam step.
 * `state`: device {float gnorm_out; int32 step} ; scratch: HIG_NORM_BLOCKS floats. */
#define HIG_NORM_BLOCKS 1024
/* Two-person losses of DDPMMulTrainer.backward_G (mul_ddpm_trainer.py:223-247): per-token MSE with the
 * init-pose token (t == 0) scored on its first 4 features, masked by length[r].  pit == 0: the labelled loss
 * sum / sum(mask).  pit == 1: rows are [m1|c1, m1|c2, m2|c2, m2|c1] (R = 4 x pairs); per pair the cheaper of the
 * two caption assignments, / (sum(mask) / 2).  dpred (optional) = d loss / d pred.  scratch: 2 R floats. */
int hig_pair_mse(const float* pred, const float* target, const int64_t* length, int32_t R, int32_t T,
                 int32_t F, int32_t pit, float* loss, float* dpred, float* scratch, hig_stream_t s);
int hig_sumsq_partial(const float* g, int64_t n, float inv_world, float* scratch, hig_stream_t s);
int hig_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                  float b2, float eps, float max_norm, float inv_world, const float* scratch,
                  float* gnorm_out, int32_t* step_dev, hig_stream_t s);
/* The same step with the learning rate read from device memory (lr_dev[0]) when lr_dev != NULL: a captured hipGraph
 * of the training step then follows an LR schedule without being re-captured (the host writes the scalar before each
 * replay). */
int hig_clip_adam_lrdev(float* p, const float* g, float* m, float* v, int64_t n, float lr, const float* lr_dev,
                        float b1, float b2, float eps, float max_norm, float inv_world, const float* scratch,
                        float* gnorm_out, int32_t* step_dev, hig_stream_t s);
/* hig_clip_adam_lrdev that also writes the bf16 shadow of the updated parameters, shadow16[i] = bf16(p[i]) for i < shadow_n
 * (shadow_n <= n, a multiple of 4): the bf16-storage step needs no separate cast pass after the update. */
int hig_clip_adam_shadow(float* p, const float* g, float* m, float* v, int64_t n, float lr, const float* lr_dev,
                         float b1, float b2, float eps, float max_norm, float inv_world, const float* scratch,
                         float* gnorm_out, int32_t* step_dev, void* shadow16, int64_t shadow_n, hig_stream_t s);
/* Diagnostic: launches a one-thread kernel named hig_marker_kernel with a grid of `id` blocks -- a named, numbered mark in a
 * rocprofv3 kernel trace (bench.py brackets its roofline microbenchmark with markers 1 and 2; tools/summarize_profiles.py
 * averages the launches between them).  Does nothing else. */
int hig_debug_marker(int32_t id, hig_stream_t s);
/* Releases what the library itself owns on the calling host thread: the second stream and the events
 * hig_denoiser_bwd forks its weight gradients onto (created lazily, one set per host thread and device).  Call it
 * from the thread that ran the backward, with no launch of this library in flight; later calls re-create them. */
int hig_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* HIG_H */
