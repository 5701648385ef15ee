// Launch sequences of the denoiser forward / backward over the fused kernels.
// Reference: MotionTransformer.forward (codes/models/transformer.py:407-426) and the blocks it
// calls (:60-194); backward is the hand-derived adjoint of exactly that graph (the reference
// relies on torch autograd).  Host code only: every call enqueues kernels on `stream`, nothing
// else -- safe under hipGraph capture.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "hig_common.h"
#include "hig_host.h"

// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int hig_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
extern "C" int hig_last_error(char* buf, int n) {
  if (buf && n > 0) {
    strncpy(buf, g_err, (size_t)n - 1);
    buf[n - 1] = 0;
  }
  return (int)strlen(g_err);
}
extern "C" int hig_version(void) { return 100; }

namespace {

__global__ void mul_dsilu_kernel(const float* __restrict__ a, const float* __restrict__ z, int64_t n,
                                 float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] * hig_dsilu(z[i]);
}

// out[b] = in[(b + B/2) % B]  (the partner's sequence length, two-person model)
__global__ void swap_halves_i64_kernel(const int64_t* __restrict__ in, int B, int64_t fill, int64_t* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = in ? in[(b + B / 2) % B] : fill;
}
// Copies row (b, 0, :) of buf [B][T][n] to tok0 [B][n] (if tok0) and zeroes it in buf (if zero).
__global__ void tok0_kernel(float* __restrict__ buf, int64_t sample_stride, int B, int n, float* __restrict__ tok0,
                            int zero) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * n) return;
  const int b = idx / n, c = idx % n;
  float* p = buf + (int64_t)b * sample_stride + c;
  if (tok0) tok0[idx] = *p;
  if (zero) *p = 0.f;
}


struct Dims {
  int B, T, F, d, H, ff, L, N, Lt, nf, hd, E, prec, full, two, nsty, bf16;
  int64_t M, Mt;
};

int check_dims(const hig_dims* p, Dims& D) {
  HIG_REQUIRE(p, "null dims");
  D.B = p->B; D.T = p->T; D.F = p->F; D.d = p->d; D.H = p->H; D.ff = p->ff; D.L = p->L;
  D.N = p->N; D.Lt = p->Lt; D.nf = p->num_frames;
  HIG_REQUIRE(D.B > 0 && D.T > 0 && D.F > 0 && D.d > 0 && D.H > 0 && D.ff > 0 && D.L > 0 && D.N > 0 && D.Lt > 0,
              "hig_dims: every extent must be positive");
  HIG_REQUIRE(D.d % D.H == 0, "hig_dims: d=%d not divisible by H=%d", D.d, D.H);
  D.hd = D.d / D.H;
  D.E = 4 * D.d;
  HIG_REQUIRE(D.hd == 8 || D.hd == 16 || D.hd == 32 || D.hd == 64 || D.hd == 128,
              "hig_dims: head dim %d not in {8,16,32,64,128}", D.hd);
  HIG_REQUIRE(D.d % 4 == 0 && D.ff % 4 == 0 && D.Lt % 4 == 0, "hig_dims: d, ff, Lt must be multiples of 4");
  HIG_REQUIRE(D.d <= 1024 && D.Lt <= 1024, "hig_dims: d and Lt must be <= 1024");
  HIG_REQUIRE(p->attn_kind == HIG_ATTN_LINEAR || p->attn_kind == HIG_ATTN_FULL, "hig_dims: unknown attn_kind=%d",
              p->attn_kind);
  D.full = p->attn_kind == HIG_ATTN_FULL;
  if (p->prec != HIG_PREC_F32 && p->prec != HIG_PREC_BF16X3 && p->prec != HIG_PREC_BF16)
    return hig_set_error(HIG_EINVAL, "hig: unknown prec=%d", p->prec);
  D.prec = p->prec;
  HIG_REQUIRE(p->two_person >= 0 && p->two_person <= 2, "hig_dims: two_person must be 0, 1 or 2");
  D.two = p->two_person;
  D.nsty = D.two == 1 ? 4 : 3;  // stylization blocks per layer: sa, ca, [int_ca], ffn
  if (D.two) {
    HIG_REQUIRE(D.B % 2 == 0 && D.T >= 2 && D.F >= 4, "hig_dims: two-person needs even B, T >= 2, F >= 4");
    HIG_REQUIRE(D.T - 1 <= D.nf, "hig_dims: T-1=%d exceeds num_frames=%d", D.T - 1, D.nf);
    if (D.full) return hig_set_error(HIG_EUNSUPPORTED, "hig: the two-person model is built for linear attention only");
  } else {
    HIG_REQUIRE(D.T <= D.nf, "hig_dims: T=%d exceeds num_frames=%d", D.T, D.nf);
  }
  D.M = (int64_t)D.B * D.T;
  D.Mt = (int64_t)D.B * D.N;
  HIG_REQUIRE(p->storage == HIG_STORE_F32 || p->storage == HIG_STORE_BF16, "hig_dims: unknown storage=%d", p->storage);
  D.bf16 = p->storage == HIG_STORE_BF16;
  if (D.bf16) {

    if (D.hd != 64 && D.hd != 128)
      return hig_set_error(HIG_EUNSUPPORTED, "hig: bf16 storage needs head dim 64 or 128 (got %d)", D.hd);
    if (D.d % 32 || D.ff % 32 || D.Lt % 32)
      return hig_set_error(HIG_EUNSUPPORTED, "hig: bf16 storage needs d, ff, Lt multiples of 32 (got %d, %d, %d)", D.d, D.ff, D.Lt);
  }
  return HIG_OK;
}

// Forward workspace (floats).  Per-layer block repeated L times when training, once otherwise.
struct FwdLayout {
  int64_t te, te_h, emb, ss, h0, lenp, cscr, gtail;
  int64_t layer0, lstride;
  int64_t st1, qkv, A1, kst1, lse1, y1, st2, a1, h1, st3, qc, lse2, y2, st4, a2, h2, z1, f1, y3, st5, a3, h3;
  int64_t st6, iqkv, Ai, ksti, y4, st7, a4, h2b;  // two-person interaction attention block
  int64_t xn1, xn2, xn3;  // LayerNorm outputs feeding the q/k/v GEMMs (kept: wgrad operands)
  int64_t cscr2, gtail2;  // second set of per-stream scratch (two-stream forward)
  int64_t gtail3;         // split-tail scratch of the text side when it runs on the third stream (hig_denoiser_fwd_text)
  int64_t lnstats;        // LayerNorm fold (inference): (sum, centred sum of squares) per row and 64-column panel of the residual stream
  int64_t total;
};
FwdLayout fwd_layout(const Dims& D, int training) {
  FwdLayout w;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += al(n); return r; };
  w.te = take((int64_t)D.B * D.d);
  w.te_h = take((int64_t)D.B * D.E);
  w.emb = take((int64_t)D.B * D.E);
  w.ss = take((int64_t)D.B * D.nsty * D.L * 2 * D.d);
  w.h0 = take(D.M * D.d);
  w.lenp = take((int64_t)D.B * 2);  // int64 lengths with the two halves swapped (partner's mask)
  w.cscr = take(hig_linattn_ctx_scratch_floats(D.B, D.T, D.H, D.hd));   // chunk partials of the context build
  w.gtail = take(HIG_GEMM_TAIL_BYTES / 4);   // split tail of the fp32 GEMMs (hig_gemm_set_tail_scratch)
  w.cscr2 = take(hig_linattn_ctx_scratch_floats(D.B, D.T, D.H, D.hd));
  w.gtail2 = take(HIG_GEMM_TAIL_BYTES / 4);
  w.gtail3 = take(HIG_GEMM_TAIL_BYTES / 4);
  w.lnstats = take(training ? 0 : D.M * (D.d / 64 + 1) * 2);
  w.layer0 = o;
  o = 0;
  w.st1 = take(D.M * 2);
  w.xn1 = take(D.M * D.d);
  w.qkv = take(D.M * 3 * D.d);
  w.A1 = take((int64_t)D.B * D.H * D.hd * D.hd);
  w.kst1 = take((int64_t)D.B * D.d * 2);
  w.lse1 = take((int64_t)D.B * D.H * D.T);  // full attention: log-sum-exp per (b, h, query)
  w.y1 = take(D.M * D.d);
  w.st2 = take(D.M * 2);
  w.a1 = take(D.M * D.d);
  w.h1 = take(D.M * D.d);
  w.st3 = take(D.M * 2);
  w.xn2 = take(D.M * D.d);
  w.qc = take(D.M * D.d);
  w.lse2 = take((int64_t)D.B * D.H * D.T);
  w.y2 = take(D.M * D.d);
  w.st4 = take(D.M * 2);
  w.a2 = take(D.M * D.d);
  w.h2 = take(D.M * D.d);
  w.z1 = take(D.M * D.ff);
  w.f1 = take(D.M * D.ff);
  w.y3 = take(D.M * D.d);
  w.st5 = take(D.M * 2);
  w.a3 = take(D.M * D.d);
  w.h3 = take(D.M * D.d);
  w.st6 = w.iqkv = w.Ai = w.ksti = w.y4 = w.st7 = w.a4 = w.h2b = w.xn3 = 0;
  if (D.two == 1) {
    w.st6 = take(D.M * 2);
    w.xn3 = take(D.M * D.d);
    w.iqkv = take(D.M * 3 * D.d);
    w.Ai = take((int64_t)D.B * D.H * D.hd * D.hd);
    w.ksti = take((int64_t)D.B * D.d * 2);
    w.y4 = take(D.M * D.d);
    w.st7 = take(D.M * 2);
    w.a4 = take(D.M * D.d);
    w.h2b = take(D.M * D.d);
  }
  w.lstride = training ? o : 0;
  w.total = w.layer0 + (training ? o * D.L : o);
  return w;
}

// Text context (floats): LN stats of xf_out (shared by all layers), then per layer the context
// matrices + column-softmax stats, and the key/value projections (kept per layer for backward).
struct TextLayout {
  int64_t stt, cscr, layer0, lstride, Ac, kstc, kv, kv_stride, xhat, kvall, total;
};
TextLayout text_layout(const Dims& D, int training) {
  TextLayout t;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += al(n); return r; };
  t.stt = take(D.Mt * 2);
  t.cscr = take(hig_linattn_ctx_scratch_floats(D.B, D.N, D.H, D.hd));
  t.layer0 = o;
  o = 0;
  t.Ac = take((int64_t)D.B * D.H * D.hd * D.hd);
  t.kstc = take((int64_t)D.B * D.d * 2);
  t.lstride = o;
  t.kv = t.layer0 + t.lstride * D.L;
  // linear attention needs key/value only to build A_c (kept per layer just for backward);
  // full attention reads them at every step
  const bool keep = training || D.full;
  t.kv_stride = keep ? al(D.Mt * 2 * D.d) : 0;
  t.total = t.kv + (keep ? t.kv_stride * D.L : al(D.Mt * 2 * D.d));
  // inference, linear attention, fp32 storage: room for the batched form of the text side (text_context_impl: the normalised
  // text rows once, the key/value projections of ALL layers as one (Mt, L 2d) matrix)
  t.xhat = t.kvall = -1;
  if (!keep && !D.bf16) {
    o = t.total;
    t.xhat = take(D.Mt * D.Lt);
    t.kvall = take(D.Mt * 2 * D.d * D.L);
    t.total = o;
  }
  return t;
}

struct BwdLayout {
  int64_t dhA, dhB, t1, t2, tff, dqkv, dA, delta, dkv, dxfn, dss, demb, dtmp, dte_h, slabs, slab_floats,
      colpart, colpart_w, lnpart, wT, tA, tB, attn, tok0, doutm, postmp, gtail, total;
};
BwdLayout bwd_layout(const Dims& D) {
  BwdLayout w;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += al(n); return r; };
  w.dhA = take(D.M * D.d);
  w.dhB = take(D.M * D.d);
  w.t1 = take(D.M * D.d);
  w.t2 = take(D.M * D.d);
  w.tff = take(D.M * D.ff);
  w.dqkv = take(D.M * 3 * D.d);
  w.dA = take((int64_t)D.B * D.H * D.hd * D.hd);
  w.delta = take((int64_t)D.B * D.H * D.T);
  w.dkv = take(D.Mt * 2 * D.d);
  w.dxfn = take(D.Mt * D.Lt);
  w.dss = take((int64_t)D.B * D.nsty * D.L * 2 * D.d);
  w.demb = take((int64_t)D.B * D.E);
  w.dtmp = take((int64_t)D.B * D.E);
  w.dte_h = take((int64_t)D.B * D.E);
  // split-R slabs: up to 1024 (tile, split) pairs of 128x128 floats, or 16 splits of any output
  int64_t biggest = 0;
  const int64_t outs[] = {(int64_t)3 * D.d * D.d, (int64_t)D.d * D.d, (int64_t)D.ff * D.d,
                          (int64_t)2 * D.d * D.Lt, (int64_t)D.F * D.d, (int64_t)D.B * D.E,
                          (int64_t)D.E * D.E, (int64_t)D.E * D.d};
  for (int64_t v : outs) biggest = v > biggest ? v : biggest;
  w.slab_floats = biggest * 16 > (int64_t)1536 * 128 * 128 ? biggest * 16 : (int64_t)1536 * 128 * 128;
  w.slabs = take(w.slab_floats);
  // column-sum partials: [chunks(rows)][n] for every (rows, n) the backward reduces
  int64_t colp = 0;
  const int64_t uses[][2] = {{D.M, 3 * D.d}, {D.M, D.ff}, {D.M, D.F}, {D.Mt, 2 * D.d}, {D.B, D.E},
                             {D.B, (int64_t)D.nsty * D.L * 2 * D.d}, {D.B, (int64_t)D.T * D.d}};
  for (auto& u : uses) {
    const int64_t v = (int64_t)hig_colsum_chunks(u[0]) * u[1];
    colp = v > colp ? v : colp;
  }
  w.colpart = take(colp);
  w.colpart_w = take(colp);   // the weight-gradient stream's own partials (it runs next to the caller's stream)
  int64_t lp = hig_ln_bwd_partial_floats(D.M, D.d, D.T);
  const int64_t lpt = hig_ln_bwd_partial_floats(D.Mt, D.Lt, D.N);
  w.lnpart = take(lp > lpt ? lp : lpt);
  // one layer's transposed weights (reduce-contiguous operands for the data-gradient GEMMs)
  w.wT = take((int64_t)11 * D.d * D.d + (int64_t)2 * D.d * D.ff + (int64_t)2 * D.d * D.Lt);
  // transposed dC / activation operands of the weight-gradient GEMMs (bf16 product modes)
  const int64_t mrows = D.M > D.Mt ? D.M : D.Mt;
  const int64_t wide = 3 * D.d > D.ff ? 3 * D.d : D.ff;
  w.tA = take(wide * mrows);
  w.tB = take((int64_t)(D.ff > D.d ? (D.ff > D.Lt ? D.ff : D.Lt) : (D.d > D.Lt ? D.d : D.Lt)) * mrows);
  const int64_t as1 = hig_linattn_bwd_scratch_floats(D.B, D.T, D.H, D.hd);
  const int64_t as2 = hig_linattn_bwd_scratch_floats(D.B, D.N, D.H, D.hd);
  w.attn = take(as1 > as2 ? as1 : as2);
  w.tok0 = take((int64_t)D.B * D.d);       // two-person: init-pose rows of d(h0)
  w.doutm = take(D.two ? D.M * D.F : 0);   // two-person: d(out) with the init-pose rows zeroed
  w.postmp = take(D.two ? (int64_t)D.T * D.d : 0);
  w.gtail = take(HIG_GEMM_TAIL_BYTES / 4);   // split tail of the data-gradient GEMMs (caller's stream only)
  w.total = o;
  return w;
}

// Hands the fp32 GEMMs launched in this scope (same thread, ONE stream) a split-tail scratch (hig_host.h).
struct TailScratchScope {
  explicit TailScratchScope(void* p) { hig_gemm_set_tail_scratch(p, HIG_GEMM_TAIL_BYTES); }
  ~TailScratchScope() { hig_gemm_set_tail_scratch(nullptr, 0); }
  TailScratchScope(const TailScratchScope&) = delete;
  TailScratchScope& operator=(const TailScratchScope&) = delete;
};

inline const float* P(const void* const* t, int idx) { return static_cast<const float*>(t[idx]); }
inline const float* PL(const void* const* t, int l, int idx) {
  return static_cast<const float*>(t[HIG_NGLOBAL + l * HIG_NLAYER + idx]);
}
inline float* GP(void* const* t, int idx) { return static_cast<float*>(t[idx]); }
inline float* GL(void* const* t, int l, int idx) {
  return static_cast<float*>(t[HIG_NGLOBAL + l * HIG_NLAYER + idx]);
}

}  // namespace

namespace {
int64_t fwd16_total(const Dims& D);
int64_t text16_total(const Dims& D);
int64_t fwd16t_total(const Dims& D);
int64_t text16t_total(const Dims& D);
int64_t bwd16_total(const Dims& D);
// the bf16-storage TRAINING step is built for linear attention (single-person and two-person model)
int bf16_train_unsupported(const Dims& D, int training) {
  if (D.bf16 && training && D.full) {
    hig_set_error(HIG_EUNSUPPORTED, "hig: bf16-storage training is built for linear attention (attn_kind=%d: train with fp32 storage)", D.full);
    return 1;
  }
  return 0;
}
}  // namespace
extern "C" int64_t hig_workspace_bytes(const hig_dims* dims, int training) {
  Dims D;
  if (check_dims(dims, D) != HIG_OK || bf16_train_unsupported(D, training)) return -1;
  if (D.bf16) return training ? fwd16t_total(D) : fwd16_total(D);
  return fwd_layout(D, training).total * 4;
}
extern "C" int64_t hig_textctx_bytes(const hig_dims* dims, int training) {
  Dims D;
  if (check_dims(dims, D) != HIG_OK || bf16_train_unsupported(D, training)) return -1;
  if (D.bf16) return training ? text16t_total(D) : text16_total(D);
  return text_layout(D, training).total * 4;
}
extern "C" int64_t hig_bwd_workspace_bytes(const hig_dims* dims) {
  Dims D;
  if (check_dims(dims, D) != HIG_OK || bf16_train_unsupported(D, 1)) return -1;
  if (D.bf16) return bwd16_total(D);
  return bwd_layout(D).total * 4;
}

namespace {
struct SideStream;
int text_context_impl(const Dims& D, const void* const* params, const void* const* derived32, const float* xf_out, void* textctx,
                      int training, hipStream_t st, hipEvent_t* layer_done);
}
extern "C" int hig_text_context(const hig_dims* dims, const void* const* params, const float* xf_out,
                                void* textctx, int training, hig_stream_t stream) {
  Dims D;
  HIG_TRY(check_dims(dims, D));
  HIG_REQUIRE(params && xf_out && textctx, "hig_text_context: null argument");
  HIG_REQUIRE(!D.bf16, "hig_text_context: bf16 storage goes through hig_text_context_bf16");
  return text_context_impl(D, params, nullptr, xf_out, textctx, training, hig_stream(stream), nullptr);
}
namespace {
// layer_done (nullable): event l is recorded on `st` behind layer l's launches (hig_denoiser_fwd_text waits for it in front of
// layer l's cross-attention)
// Does the text side run in its batched form (one key/value GEMM + one context build for all layers)?  Inference, linear
// attention, fp32 storage, and the caller's derived-operand table carries the stacked folded weights (entries 6 L .. 6 L + 3).
bool text_batched(const Dims& D, const void* const* derived32, int training) {
  static const int batch_env = getenv("HIG_TEXT_BATCH") ? atoi(getenv("HIG_TEXT_BATCH")) : 1;   // tuning knob
  return batch_env && derived32 && !training && !D.full && !D.bf16 && derived32[6 * D.L] && derived32[6 * D.L + 1] &&
         derived32[6 * D.L + 2] && derived32[6 * D.L + 3];
}
// derived32 (nullable): the caller's derived-operand table (hig_denoiser_fwd_x); entries [6 L .. 6 L + 3] select the BATCHED form
int text_context_impl(const Dims& D, const void* const* params, const void* const* derived32, const float* xf_out, void* textctx,
                      int training, hipStream_t st, hipEvent_t* layer_done) {
  hig_stream_t stream = reinterpret_cast<hig_stream_t>(st);
  const TextLayout tl = text_layout(D, training);
  float* base = static_cast<float*>(textctx);
  float* stt = base + tl.stt;
  // Batched form (inference, linear attention): LN_text_l(x) = xhat gamma_l + beta_l with xhat = (x - mean) rstd the same for
  // every layer, so [key_l; value_l](LN_text_l(x)) = xhat (gamma_l (.) W_l)^T + (W_l beta_l + b_l): with the folded weights of
  // all layers stacked (derived per parameter version, models/transformer.py:_derived32) the L key/value GEMMs of B N rows
  // x K = Lt are ONE product of L 2d columns -- 77 row tiles per workgroup of the weight-stationary kernel instead of 8 launches
  // of 9.6 (K = 256: gemm_wsp32.hip), ~150 against ~300 us of chip time at config 2 (transformer.py:146,150).
  if (text_batched(D, derived32, training)) {
    float* xhat = base + tl.xhat;
    float* kvall = base + tl.kvall;
    const int64_t ldkv = (int64_t)D.L * 2 * D.d;
    HIG_TRY(hig_layernorm(xf_out, D.Lt, D.Mt, D.Lt, static_cast<const float*>(derived32[6 * D.L + 2]),
                          static_cast<const float*>(derived32[6 * D.L + 3]), xhat, D.Lt, stt, stream));
    HIG_TRY(hig_gemm_launch(G(xhat, D.Lt, 0, static_cast<const float*>(derived32[6 * D.L]), D.Lt, 0, kvall, ldkv, D.Mt, ldkv, D.Lt)
                                .epi(HIG_EPI_BIAS, static_cast<const float*>(derived32[6 * D.L + 1])).prec(D.prec).g, 1, nullptr, st));
    // the stacked weights put all keys in front of all values ([K_0 .. K_{L-1} | V_0 .. V_{L-1}]): head l H + h of "L H heads"
    // is layer l's head h, and ONE context-build launch serves every layer (4096 workgroups instead of 8 x 512)
    const float* Kall = kvall;
    const float* Vall = kvall + (int64_t)D.L * D.d;
    int rc = hig_linattn_ctx_groups(Kall, Vall, ldkv, D.B, D.N, D.H, D.L, D.hd, base + tl.layer0 + tl.Ac, tl.lstride,
                                    base + tl.layer0 + tl.kstc, tl.lstride, st);
    if (rc < 0) return rc;
    for (int l = 0; l < D.L; ++l) {
      if (rc != HIG_OK) {   // (head dim not on the matrix-core kernels: one launch per layer)
        float* Ac = base + tl.layer0 + tl.lstride * l + tl.Ac;
        float* kstc = base + tl.layer0 + tl.lstride * l + tl.kstc;
        HIG_TRY(hig_linattn_ctx(Kall + (int64_t)l * D.d, Vall + (int64_t)l * D.d, ldkv, D.B, D.N, D.H, D.hd, nullptr, Ac, kstc,
                                base + tl.cscr, stream));
      }
      if (layer_done && hipEventRecord(layer_done[l], st) != hipSuccess) return hig_set_error(HIG_EHIP, "hipEventRecord failed");
    }
    return HIG_OK;
  }
  HIG_TRY(hig_rowstats(xf_out, D.Lt, D.Mt, D.Lt, stt, stream));
  for (int l = 0; l < D.L; ++l) {
    float* kv = base + tl.kv + tl.kv_stride * l;
    float* Ac = base + tl.layer0 + tl.lstride * l + tl.Ac;
    float* kstc = base + tl.layer0 + tl.lstride * l + tl.kstc;
    // [key; value](LN_text(xf_out))   (transformer.py:146,150)
    HIG_TRY(hig_gemm_launch(G(xf_out, D.Lt, 0, PL(params, l, HIG_L_CA_KV_W), D.Lt, 0, kv, 2 * D.d, D.Mt, 2 * D.d, D.Lt)
                                .ln(0, stt, PL(params, l, HIG_L_CA_TNORM_W), PL(params, l, HIG_L_CA_TNORM_B))
                                .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_CA_KV_B)).prec(D.prec).g, 1, nullptr, st));
    // linear attention: softmax over the N text tokens (no mask) and A = k^T v (transformer.py:148,152);
    // full attention (:253-259) consumes key/value directly
    if (!D.full)
      HIG_TRY(hig_linattn_ctx(kv, kv + D.d, 2 * D.d, D.B, D.N, D.H, D.hd, nullptr, Ac, kstc, base + tl.cscr, stream));
    if (layer_done && hipEventRecord(layer_done[l], st) != hipSuccess) return hig_set_error(HIG_EHIP, "hipEventRecord failed");
  }
  return HIG_OK;
}
}  // namespace

namespace {

// (the library-owned second stream of the calling thread: protocol described at WgradFork below)
constexpr int kMaxTextLayers = 32;
struct SideStream {
  hipStream_t s2 = nullptr;
  hipEvent_t ready = nullptr;
  hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr};
  // third stream: the cross-attention text side of hig_denoiser_fwd_text runs next to the first layers (one event per layer)
  hipStream_t s3 = nullptr;
  hipEvent_t text_done[kMaxTextLayers] = {};
  bool ok = false, failed = false;
};

constexpr int kMaxDev = 16;
SideStream* side_stream_table() {
  static thread_local SideStream tab[kMaxDev];
  return tab;
}

hipEvent_t& layer_event() {   // hig_denoiser_bwd_hooked: "layer l is enqueued" marker on the caller's stream
  static thread_local hipEvent_t ev = nullptr;
  return ev;
}

// HIG_BWD_OVERLAP, read once for every user below: -1 unset (each form's own default), 0 everything on the caller's stream,
// 1 fork the weight gradients in every form (eager, captured, bf16 storage).
int bwd_overlap_env() {
  static const int v = getenv("HIG_BWD_OVERLAP") ? atoi(getenv("HIG_BWD_OVERLAP")) : -1;
  return v;
}

SideStream* side_stream_for_current_device(hipStream_t caller) {
  if (bwd_overlap_env() == 0) return nullptr;
  SideStream* tab = side_stream_table();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  SideStream& s = tab[dev];
  if (s.failed) return nullptr;
  if (!s.ok) {
    // first use on this thread / device.  Creating a stream is not something to do under capture: a caller that
    // captures its very first backward (no eager warm-up) simply gets the single-stream order for that graph.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(caller, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    bool good = hipStreamCreateWithFlags(&s.s2, hipStreamNonBlocking) == hipSuccess;
    good = good && hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 4 && good; ++i) good = hipEventCreateWithFlags(&s.done[i], hipEventDisableTiming) == hipSuccess;
    good = good && hipStreamCreateWithFlags(&s.s3, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < kMaxTextLayers && good; ++i) good = hipEventCreateWithFlags(&s.text_done[i], hipEventDisableTiming) == hipSuccess;
    if (!good) {   // do not retry (and leak) on every call: stay on the caller's stream for good
      s.failed = true;
      return nullptr;
    }
    s.ok = true;
  }
  return &s;
}

}  // namespace

static int denoiser_fwd_impl(const hig_dims* dims, const void* const* params, const void* const* derived32, const float* x, const int64_t* t,
                             const int64_t* length, const float* xf_proj, const float* xf_out_for_text, const void* textctx, float* out,
                             void* workspace, int training, hig_stream_t stream);
extern "C" int hig_denoiser_fwd(const hig_dims* dims, const void* const* params, const float* x,
                                const int64_t* t, const int64_t* length, const float* xf_proj,
                                const void* textctx, float* out, void* workspace, int training,
                                hig_stream_t stream) {
  return denoiser_fwd_impl(dims, params, nullptr, x, t, length, xf_proj, nullptr, textctx, out, workspace, training, stream);
}
// hig_text_context + hig_denoiser_fwd as ONE call (include/hig.h): the text side is forked onto a library-owned stream
extern "C" int hig_denoiser_fwd_text(const hig_dims* dims, const void* const* params, const float* x, const int64_t* t,
                                     const int64_t* length, const float* xf_proj, const float* xf_out, void* textctx, float* out,
                                     void* workspace, int training, hig_stream_t stream) {
  HIG_REQUIRE(xf_out, "hig_denoiser_fwd_text: null argument");
  return denoiser_fwd_impl(dims, params, nullptr, x, t, length, xf_proj, xf_out, textctx, out, workspace, training, stream);
}
// The general form (include/hig.h): derived32 (nullable) = LayerNorm-folded projection operands, xf_out (nullable) = compute the
// text side here.
extern "C" int hig_denoiser_fwd_x(const hig_dims* dims, const void* const* params, const void* const* derived32, const float* x,
                                  const int64_t* t, const int64_t* length, const float* xf_proj, const float* xf_out, void* textctx,
                                  float* out, void* workspace, int training, hig_stream_t stream) {
  return denoiser_fwd_impl(dims, params, derived32, x, t, length, xf_proj, xf_out, textctx, out, workspace, training, stream);
}
static int denoiser_fwd_impl(const hig_dims* dims, const void* const* params, const void* const* derived32, const float* x, const int64_t* t,
                             const int64_t* length, const float* xf_proj, const float* xf_out_for_text, const void* textctx, float* out,
                             void* workspace, int training, hig_stream_t stream) {
  Dims D;
  HIG_TRY(check_dims(dims, D));
  HIG_REQUIRE(params && x && t && xf_proj && textctx && out && workspace, "hig_denoiser_fwd: null argument");
  HIG_REQUIRE(!D.bf16, "hig_denoiser_fwd: bf16 storage goes through hig_denoiser_fwd_bf16");
  const FwdLayout w = fwd_layout(D, training);
  const TextLayout tl = text_layout(D, training);
  float* ws = static_cast<float*>(workspace);
  const float* tc = static_cast<const float*>(textctx);
  hipStream_t st = hig_stream(stream);
  const int d = D.d, E = D.E;
  const int64_t M = D.M;
  const int64_t ss_ld = (int64_t)D.nsty * D.L * 2 * d;
  const int Bp = D.B / 2;  // pairs (two-person)

  // the forward's GEMMs all run on `st`: they share one split-tail scratch (tickets zeroed here, left zero by each launch)
  HIG_TRY(hig_zero_async(ws + w.gtail, HIG_GEMM_TAIL_CNT_BYTES, st));
  const TailScratchScope tail_scope(ws + w.gtail);

  // The cross-attention text side (hig_denoiser_fwd_text): layer l's context matrices are first needed in front of layer l's
  // cross-attention, a third of a layer into the forward -- its 2 L launches (key/value GEMMs over B N rows, context builds)
  // run on a third stream next to the first layers, one event per layer (events only: eager launches; under capture, or
  // without the library's streams, they run first on the caller's stream as hig_text_context would).
  hipEvent_t* text_ev = nullptr;
  // Whatever the text fork has enqueued on the third stream is joined back into `st` on every exit that does not reach the
  // last layer's own wait (the caller recycles `textctx` / `xf_out` as soon as `st` gets there): disarmed on success.
  struct TextJoin {
    hipStream_t s3 = nullptr, st = nullptr;
    hipEvent_t ev = nullptr;
    bool armed = false;
    void arm(hipStream_t s3_, hipEvent_t ev_, hipStream_t st_) { s3 = s3_; ev = ev_; st = st_; armed = true; }
    ~TextJoin() {
      if (!armed) return;
      (void)hipEventRecord(ev, s3);
      (void)hipStreamWaitEvent(st, ev, 0);
    }
  } text_join;
  if (xf_out_for_text) {
    static const int text_fork = getenv("HIG_TEXT_FORK") ? atoi(getenv("HIG_TEXT_FORK")) : 1;   // tuning knob
    // (the batched form is two whole-chip launches: next to the first layers' GEMMs -- one workgroup per CU each -- they only take
    // turns with them, 6.04 ms forked against 5.98 in front at config 2; it runs first on the caller's stream)
    SideStream* ts = (text_fork && D.L <= kMaxTextLayers && !text_batched(D, derived32, training)) ? side_stream_for_current_device(st) : nullptr;
    if (ts) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) ts = nullptr;
    }
    if (ts) {
      // The text side's key/value GEMMs run NEXT TO the GEMMs on `st`: they get a split-tail scratch of their own (tickets and
      // partial sums shared between two concurrent launches would hand a workgroup another GEMM's "last arriver" ticket or
      // slices: B = 32, N = 77 text rows qualify for the split tail just like the q/k/v launch beside them).
      HIG_TRY(hig_zero_async(ws + w.gtail3, HIG_GEMM_TAIL_CNT_BYTES, st));
      if (hipEventRecord(ts->ready, st) != hipSuccess || hipStreamWaitEvent(ts->s3, ts->ready, 0) != hipSuccess)
        return hig_set_error(HIG_EHIP, "text fork failed");
      hig_gemm_set_tail_scratch(ws + w.gtail3, HIG_GEMM_TAIL_BYTES);
      const int rc = text_context_impl(D, params, derived32, xf_out_for_text, const_cast<void*>(textctx), training, ts->s3, ts->text_done);
      hig_gemm_set_tail_scratch(ws + w.gtail, HIG_GEMM_TAIL_BYTES);
      text_join.arm(ts->s3, ts->text_done[0], st);   // every error exit below joins the text stream first
      if (rc != HIG_OK) return rc;
      text_ev = ts->text_done;
    } else {
      HIG_TRY(text_context_impl(D, params, derived32, xf_out_for_text, const_cast<void*>(textctx), training, st, nullptr));
    }
  }
  // K0: emb = time_embed(timestep_embedding(t)) + xf_proj; all 3L scale/shift pairs in ONE GEMM
  HIG_TRY(hig_timestep_embedding(t, D.B, d, ws + w.te, stream));
  // per-sample (B-row) GEMMs: weight-bandwidth bound; split over the reduce range with the still unused layer
  // buffers as scratch (hig_gemm_few_rows)
  float* few_scratch = ws + w.layer0;
  const int64_t few_floats = w.total - w.layer0;
  HIG_TRY(hig_gemm_few_rows(G(ws + w.te, d, 0, P(params, HIG_P_TE0_W), d, 0, ws + w.te_h, E, D.B, E, d)
                                .epi(HIG_EPI_BIAS, P(params, HIG_P_TE0_B)).prec(D.prec).g, few_scratch, few_floats, st));
  HIG_TRY(hig_gemm_few_rows(G(ws + w.te_h, E, 0, P(params, HIG_P_TE2_W), E, 0, ws + w.emb, E, D.B, E, E)
                                .silu(0).epi(HIG_EPI_BIAS_RES, P(params, HIG_P_TE2_B)).res(xf_proj, E).prec(D.prec).g,
                            few_scratch, few_floats, st));
  HIG_TRY(hig_gemm_few_rows(G(ws + w.emb, E, 0, P(params, HIG_P_STY_EMB_W), E, 0, ws + w.ss, ss_ld, D.B, ss_ld, E)
                                .silu(0).epi(HIG_EPI_BIAS, P(params, HIG_P_STY_EMB_B)).prec(D.prec).g, few_scratch, few_floats,
                            st));
  // K1: h0 = joint_embed(x) + sequence_embedding[:T]
  {
    G ge(x, D.F, 0, P(params, HIG_P_JOINT_W), D.F, 0, ws + w.h0, d, M, d, D.F);
    ge.epi(HIG_EPI_BIAS_POS, P(params, HIG_P_JOINT_B)).pos(P(params, HIG_P_SEQ_EMB), d, D.T).prec(D.prec);
    ge.g.pos_shift = D.two ? 1 : 0;  // two-person: frame t >= 1 gets sequence_embedding[t-1] (:595)
    HIG_TRY(hig_gemm_launch(ge.g, 1, nullptr, st));
  }
  const int64_t* len_partner = nullptr;
  if (D.two) {
    // token 0 is the init-pose row: joint_embed2 on its first 4 features, no positional term (:596)
    HIG_TRY(hig_gemm_launch(G(x, (int64_t)D.T * D.F, 0, P(params, HIG_P_JOINT2_W), 4, 0, ws + w.h0, (int64_t)D.T * d,
                              D.B, d, 4).epi(HIG_EPI_BIAS, P(params, HIG_P_JOINT2_B)).g, 1, nullptr, st));
    int64_t* lp = reinterpret_cast<int64_t*>(ws + w.lenp);
    hipLaunchKernelGGL(swap_halves_i64_kernel, dim3((D.B + 255) / 256), dim3(256), 0, st, length, D.B, (int64_t)D.T, lp);
    HIG_CHECK_LAUNCH();
    len_partner = lp;
  }
  // inference + linear attention: `apply` and the stylization front that follows it can run as ONE kernel (the (M, d)
  // attention output never reaches HBM); training keeps the pair -- the backward reads y and the LayerNorm statistics.
  // Not used (a switch until round 5): measured 34.0 against 36.0 us per attention at B = 64 (head dim 64), equal at B = 32,
  // 83.8 against 71.1 us at head dim 128, forward 6.24 against 6.26 ms -- the fused kernel's fp32 MFMAs and its
  // LayerNorm / SiLU arithmetic share the SIMD lanes and a workgroup's chain (load, 2 heads, statistics, epilogue,
  // store) is 26K cycles long with two workgroups per CU to hide it (profiles/r02_notes.md section 9)
  constexpr int fuse_env = 0;   // (a former tuning knob, fixed at the value that won its A/B)
  const bool fuse_apply = fuse_env && !training && !D.full && (D.H == 4 || D.H == 8) && (D.hd == 64 || D.hd == 128);
  // One decoder layer for the samples [b0, b0 + nb) on stream `s` (every kernel of a layer is row- or sample-local, so a
  // batch range is a pointer offset).  `hin` / the returned pointer are the FULL-batch residual stream of the layer.
  // LayerNorm fold (fp32 storage): inference only, d a multiple of 128, operands derived by the caller per parameter version
  static const int fold_env = getenv("HIG_LNFOLD32") ? atoi(getenv("HIG_LNFOLD32")) : 1;   // tuning knob
  const bool fold32 = fold_env && derived32 && !training && d % 128 == 0 && d <= 1024;
  int layer_rc = HIG_OK;   // the code of the launch that failed inside `layer` (it returns NULL then)
  auto layer = [&](int l, const float* hin_full, int b0, int nb, hipStream_t s, float* cscr) -> const float* {
    hig_stream_t hs = reinterpret_cast<hig_stream_t>(s);
    float* lb = ws + w.layer0 + w.lstride * l;
    const int64_t r0 = (int64_t)b0 * D.T, Mh = (int64_t)nb * D.T;          // first row / rows of this range
    const int64_t aoff = (int64_t)b0 * D.H * D.hd * D.hd;                   // context matrices (B, H, hd, hd)
    const float* ssl = ws + w.ss + (int64_t)(D.nsty * l) * 2 * d + (int64_t)b0 * ss_ld;
    const float* ss_ffn = ssl + (int64_t)(D.nsty - 1) * 2 * d;
    const int64_t* len = length ? length + b0 : nullptr;
    const float* hin = hin_full + r0 * d;
    auto R = [&](int64_t off, int64_t ld) { return lb + off + r0 * ld; };  // rows of an (M, ld) buffer of the layer
    auto fail = [&](int rc) -> const float* { layer_rc = rc; return nullptr; };
#define HIG_L(expr) do { const int rc_ = (expr); if (rc_ != HIG_OK) return fail(rc_); } while (0)
    // ---- self attention -------------------------------------------------------------
    // LayerNorm as its own row kernel: the GEMM then stages plain operands (a fused LN prologue cost
    // the q/k/v GEMM 258 -> 207 us at config 2, the row pass 13 us; profiles/r01_notes.md)
    // LayerNorm fold (inference, `derived32`): LN(h) W^T + b = rstd (h W'^T) - rstd mean colsum + b' with the row statistics
    // written by the stylization-out GEMM that produced h -- the LayerNorm launch and its (M, d) round trip disappear
    // (layer 0's first projection has no producer: plain LayerNorm)
    const int np = d >> 6;
    float* lnst = ws + w.lnstats + r0 * np * 2;
    auto folded = [&](int k) { return fold32 && derived32[6 * l + 3 * k] != nullptr; };
    auto ln_proj = [&](int k, bool have_stats, const float* hrows, int norm_w, int norm_b, int lin_w, int lin_b, float* xn, float* st,
                       float* outp, int ncols) -> int {
      if (have_stats && folded(k)) {
        G gf(hrows, d, 0, static_cast<const float*>(derived32[6 * l + 3 * k]), d, 0, outp, ncols, Mh, ncols, d);
        gf.epi(HIG_EPI_BIAS, static_cast<const float*>(derived32[6 * l + 3 * k + 2])).prec(D.prec);
        gf.g.row_stats_in = lnst;
        gf.g.ln_colsum = static_cast<const float*>(derived32[6 * l + 3 * k + 1]);
        return hig_gemm_launch(gf.g, 1, nullptr, s);
      }
      HIG_TRY(hig_layernorm(hrows, d, Mh, d, PL(params, l, norm_w), PL(params, l, norm_b), xn, d, st, hs));
      return hig_gemm_launch(G(xn, d, 0, PL(params, l, lin_w), d, 0, outp, ncols, Mh, ncols, d)
                                 .epi(HIG_EPI_BIAS, PL(params, l, lin_b)).prec(D.prec).g, 1, nullptr, s);
    };
    HIG_L(ln_proj(0, l > 0 && folded(0), hin, HIG_L_SA_NORM_W, HIG_L_SA_NORM_B, HIG_L_SA_QKV_W, HIG_L_SA_QKV_B, R(w.xn1, d), R(w.st1, 2),
                  R(w.qkv, 3 * d), 3 * d));
    if (D.full) {
      HIG_L(hig_fullattn_fwd(R(w.qkv, 3 * d), 3 * d, R(w.qkv, 3 * d) + d, R(w.qkv, 3 * d) + 2 * d, 3 * d, nb, D.T, D.T, D.H, D.hd,
                             len, R(w.y1, d), d, lb + w.lse1 + (int64_t)b0 * D.H * D.T, hs));
    } else {
      HIG_L(hig_linattn_ctx(R(w.qkv, 3 * d) + d, R(w.qkv, 3 * d) + 2 * d, 3 * d, nb, D.T, D.H, D.hd, len,
                            lb + w.A1 + aoff, lb + w.kst1 + (int64_t)b0 * d * 2, cscr, hs));
      if (fuse_apply)   // inference: apply + LayerNorm + modulation + SiLU in one kernel, y1 is never written
        HIG_L(hig_linattn_apply_sty(R(w.qkv, 3 * d), 3 * d, lb + w.A1 + aoff, PL(params, l, HIG_L_SA_STY_NORM_W),
                                    PL(params, l, HIG_L_SA_STY_NORM_B), ssl, ss_ld, d, R(w.a1, d), d, nb, D.T, D.H, D.hd, hs));
      else
        HIG_L(hig_linattn_apply(R(w.qkv, 3 * d), 3 * d, lb + w.A1 + aoff, R(w.y1, d), d, nb, D.T, D.H, D.hd, hs));
    }
    if (!fuse_apply)
      HIG_L(hig_ln_mod_silu(R(w.y1, d), d, Mh, d, PL(params, l, HIG_L_SA_STY_NORM_W), PL(params, l, HIG_L_SA_STY_NORM_B),
                            ssl, ss_ld, d, D.T, R(w.a1, d), d, R(w.st2, 2), hs));
    {
      G gs(R(w.a1, d), d, 0, PL(params, l, HIG_L_SA_STY_OUT_W), d, 0, R(w.h1, d), d, Mh, d, d);
      gs.epi(HIG_EPI_BIAS_RES, PL(params, l, HIG_L_SA_STY_OUT_B)).res(hin, d).prec(D.prec);
      if (folded(1)) gs.g.row_stats_out = lnst;          // h1's statistics for the cross-attention query projection
      HIG_L(hig_gemm_launch(gs.g, 1, nullptr, s));
    }
    // ---- cross attention ------------------------------------------------------------
    HIG_L(ln_proj(1, folded(1), R(w.h1, d), HIG_L_CA_NORM_W, HIG_L_CA_NORM_B, HIG_L_CA_Q_W, HIG_L_CA_Q_B, R(w.xn2, d), R(w.st3, 2),
                  R(w.qc, d), d));
    const float* Acl = tc + tl.layer0 + tl.lstride * l + tl.Ac + aoff;
    if (text_ev && hipStreamWaitEvent(s, text_ev[l], 0) != hipSuccess) return fail(hig_set_error(HIG_EHIP, "text join failed"));
    if (D.full) {
      const float* kvl = tc + tl.kv + tl.kv_stride * l + (int64_t)b0 * D.N * 2 * d;
      HIG_L(hig_fullattn_fwd(R(w.qc, d), d, kvl, kvl + d, 2 * d, nb, D.T, D.N, D.H, D.hd, nullptr, R(w.y2, d), d,
                             lb + w.lse2 + (int64_t)b0 * D.H * D.T, hs));
    } else {
      if (fuse_apply)
        HIG_L(hig_linattn_apply_sty(R(w.qc, d), d, Acl, PL(params, l, HIG_L_CA_STY_NORM_W),
                                    PL(params, l, HIG_L_CA_STY_NORM_B), ssl + 2 * d, ss_ld, d, R(w.a2, d), d, nb, D.T, D.H, D.hd, hs));
      else
        HIG_L(hig_linattn_apply(R(w.qc, d), d, Acl, R(w.y2, d), d, nb, D.T, D.H, D.hd, hs));
    }
    if (!fuse_apply)
      HIG_L(hig_ln_mod_silu(R(w.y2, d), d, Mh, d, PL(params, l, HIG_L_CA_STY_NORM_W), PL(params, l, HIG_L_CA_STY_NORM_B),
                            ssl + 2 * d, ss_ld, d, D.T, R(w.a2, d), d, R(w.st4, 2), hs));
    HIG_L(hig_gemm_launch(G(R(w.a2, d), d, 0, PL(params, l, HIG_L_CA_STY_OUT_W), d, 0, R(w.h2, d), d, Mh, d, d)
                              .epi(HIG_EPI_BIAS_RES, PL(params, l, HIG_L_CA_STY_OUT_B)).res(R(w.h1, d), d).prec(D.prec).g, 1, nullptr, s));
    const float* hffn = R(w.h2, d);
    int64_t hffn_off = w.h2;
    if (D.two == 1) {
      // ---- person <-> person linear cross attention (interaction_transformer.py:181-205): queries from
      // the own stream, key/value from the partner's (same LayerNorm on both), key softmax masked with
      // the consumer's length, value unmasked (masked rows carry k == 0 anyway).  Whole batch only (b0 == 0): the two
      // persons of a pair sit in different halves of the batch.
      HIG_L(hig_layernorm(lb + w.h2, d, M, d, PL(params, l, HIG_L_INT_NORM_W), PL(params, l, HIG_L_INT_NORM_B),
                          lb + w.xn3, d, lb + w.st6, hs));
      HIG_L(hig_gemm_launch(G(lb + w.xn3, d, 0, PL(params, l, HIG_L_INT_QKV_W), d, 0, lb + w.iqkv, 3 * d, M, 3 * d, d)
                                .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_INT_QKV_B)).prec(D.prec).g, 1, nullptr, s));
      HIG_L(hig_linattn_ctx(lb + w.iqkv + d, lb + w.iqkv + 2 * d, 3 * d, D.B, D.T, D.H, D.hd, len_partner,
                            lb + w.Ai, lb + w.ksti, cscr, hs));
      const int64_t halfA = (int64_t)Bp * D.H * D.hd * D.hd, halfM = (int64_t)Bp * D.T;
      HIG_L(hig_linattn_apply(lb + w.iqkv, 3 * d, lb + w.Ai + halfA, lb + w.y4, d, Bp, D.T, D.H, D.hd, hs));
      HIG_L(hig_linattn_apply(lb + w.iqkv + halfM * 3 * d, 3 * d, lb + w.Ai, lb + w.y4 + halfM * d, d, Bp, D.T, D.H,
                              D.hd, hs));
      HIG_L(hig_ln_mod_silu(lb + w.y4, d, M, d, PL(params, l, HIG_L_INT_STY_NORM_W), PL(params, l, HIG_L_INT_STY_NORM_B),
                            ssl + 4 * d, ss_ld, d, D.T, lb + w.a4, d, lb + w.st7, hs));
      HIG_L(hig_gemm_launch(G(lb + w.a4, d, 0, PL(params, l, HIG_L_INT_STY_OUT_W), d, 0, lb + w.h2b, d, M, d, d)
                                .epi(HIG_EPI_BIAS_RES, PL(params, l, HIG_L_INT_STY_OUT_B)).res(lb + w.h2, d).prec(D.prec).g,
                            1, nullptr, s));
      hffn = lb + w.h2b;
      hffn_off = w.h2b;
    }
    (void)hffn_off;
    // ---- FFN ------------------------------------------------------------------------
    HIG_L(hig_gemm_launch(G(hffn, d, 0, PL(params, l, HIG_L_FFN_W1), d, 0, R(w.f1, D.ff), D.ff, Mh, D.ff, d)
                              .epi(HIG_EPI_BIAS_GELU, PL(params, l, HIG_L_FFN_B1))
                              .aux(training ? R(w.z1, D.ff) : nullptr, D.ff).prec(D.prec).g, 1, nullptr, s));
    HIG_L(hig_gemm_launch(G(R(w.f1, D.ff), D.ff, 0, PL(params, l, HIG_L_FFN_W2), D.ff, 0, R(w.y3, d), d, Mh, d, D.ff)
                              .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_FFN_B2)).prec(D.prec).g, 1, nullptr, s));
    HIG_L(hig_ln_mod_silu(R(w.y3, d), d, Mh, d, PL(params, l, HIG_L_FFN_STY_NORM_W), PL(params, l, HIG_L_FFN_STY_NORM_B),
                          ss_ffn, ss_ld, d, D.T, R(w.a3, d), d, R(w.st5, 2), hs));
    {
      G gs(R(w.a3, d), d, 0, PL(params, l, HIG_L_FFN_STY_OUT_W), d, 0, R(w.h3, d), d, Mh, d, d);
      gs.epi(HIG_EPI_BIAS_RES, PL(params, l, HIG_L_FFN_STY_OUT_B)).res(hffn, d).prec(D.prec);
      if (fold32 && l + 1 < D.L && derived32[6 * (l + 1)] != nullptr) gs.g.row_stats_out = lnst;   // h3's statistics for the next layer's q/k/v
      HIG_L(hig_gemm_launch(gs.g, 1, nullptr, s));
    }
#undef HIG_L
    return lb + w.h3;
  };
  // K6: out = Linear(d, F)(h_L) for the samples [b0, b0 + nb)
  auto out_proj = [&](const float* hin_full, int b0, int nb, hipStream_t s) -> int {
    const int64_t r0 = (int64_t)b0 * D.T;
    return hig_gemm_launch(G(hin_full + r0 * d, d, 0, P(params, HIG_P_OUT_W), d, 0, out + r0 * D.F, D.F, (int64_t)nb * D.T, D.F, d)
                               .epi(HIG_EPI_BIAS, P(params, HIG_P_OUT_B)).prec(D.prec).g, 1, nullptr, s);
  };

  // Two halves of the batch on two streams (single-person model, enough rows): the second half runs on the library's
  // side stream, forked after the per-sample prologue and joined before returning (events only: capturable).  An
  // in-order stream leaves the chip idle in every kernel's tail and ramp-up; two independent chains of the same
  // kernels fill those gaps (two whole forwards side by side: 5.85 ms each against 6.6 alone).
  // (Round 6: with the exact-fp32 products on the weight-stationary kernel -- one workgroup per CU, every launch fills the chip
  // by itself -- two half-batch chains no longer fit side by side, and each half pays the kernel's fixed cost on half the rows:
  // B = 64 forward 5.85 ms split against 5.78 ms on one stream.  Unset, the split is therefore kept for the bf16 product modes
  // and for chips where that kernel declines.)
  static const int split_knob = getenv("HIG_FWD_SPLIT") ? atoi(getenv("HIG_FWD_SPLIT")) : -1;   // tuning knob
  const int split_env = split_knob >= 0 ? split_knob : ((D.prec == HIG_PREC_F32 && hig_gemm_wsp32_active()) ? 0 : 1);
  // (M >= 8192: measured at B = 64.  Half batches run other tile schedules than the whole batch -- other split-tail
  // geometry, so sums in another order, last-bit differences (2e-7 rel-L2) -- and the B = 32 sampling step is expected to
  // equal its captured form bit for bit (tests/test_gpu_full_size.py), so small batches stay on one stream.)
  SideStream* side = (split_env && !D.two && D.B >= 16 && M >= 8192) ? side_stream_for_current_device(st) : nullptr;
  if (side) {   // eager launches only: replayed from a hipGraph the two branches cost more than they gain (captured
                // training step 21.2 -> 22.2 ms, against 20.4 -> 20.2 ms eager; forward 6.23 -> 6.10 ms eager)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) side = nullptr;
  }
  const float* hin = ws + w.h0;
  if (side) {
    const int nbA = D.B / 2, nbB = D.B - nbA;
    HIG_TRY(hig_zero_async(ws + w.gtail2, HIG_GEMM_TAIL_CNT_BYTES, st));
    if (hipEventRecord(side->ready, st) != hipSuccess || hipStreamWaitEvent(side->s2, side->ready, 0) != hipSuccess)
      return hig_set_error(HIG_EHIP, "forward fork failed");
    // Whatever happens inside the forked region, the side stream is joined back into `st` and the thread's GEMM tail
    // scratch points at this call's own before returning: the caller may recycle `ws` / `out` as soon as `st` gets
    // there, and a failed launch must not leave s2 writing them (nor a later GEMM parking sums in gtail2).
    int rc = HIG_OK;
    for (int l = 0; l < D.L && rc == HIG_OK; ++l) {
      hig_gemm_set_tail_scratch(ws + w.gtail, HIG_GEMM_TAIL_BYTES);
      const float* ha = layer(l, hin, 0, nbA, st, ws + w.cscr);
      const float* hb = nullptr;
      if (ha) {
        hig_gemm_set_tail_scratch(ws + w.gtail2, HIG_GEMM_TAIL_BYTES);
        hb = layer(l, hin, nbA, nbB, side->s2, ws + w.cscr2);
      }
      if (!ha || !hb) rc = layer_rc != HIG_OK ? layer_rc : HIG_EHIP;
      else hin = ha;
    }
    if (rc == HIG_OK) {
      hig_gemm_set_tail_scratch(ws + w.gtail, HIG_GEMM_TAIL_BYTES);
      rc = out_proj(hin, 0, nbA, st);
    }
    if (rc == HIG_OK) {
      hig_gemm_set_tail_scratch(ws + w.gtail2, HIG_GEMM_TAIL_BYTES);
      rc = out_proj(hin, nbA, nbB, side->s2);
    }
    hig_gemm_set_tail_scratch(ws + w.gtail, HIG_GEMM_TAIL_BYTES);
    if (hipEventRecord(side->done[0], side->s2) != hipSuccess || hipStreamWaitEvent(st, side->done[0], 0) != hipSuccess)
      return rc != HIG_OK ? rc : hig_set_error(HIG_EHIP, "forward join failed");
    if (rc == HIG_OK) text_join.armed = false;   // (both halves waited for the last layer's text event)
    return rc;
  }
  for (int l = 0; l < D.L; ++l) {
    hin = layer(l, hin, 0, D.B, st, ws + w.cscr);
    if (!hin) return layer_rc != HIG_OK ? layer_rc : HIG_EHIP;
  }
  HIG_TRY(out_proj(hin, 0, D.B, st));
  if (D.two)  // init-pose rows go through out2 instead (:613-614)
    HIG_TRY(hig_gemm_launch(G(hin, (int64_t)D.T * d, 0, P(params, HIG_P_OUT2_W), d, 0, out, (int64_t)D.T * D.F, D.B, D.F, d)
                                .epi(HIG_EPI_BIAS, P(params, HIG_P_OUT2_B)).g, 1, nullptr, st));
  text_join.armed = false;
  return HIG_OK;
}

// ==========================================================================================================
// bf16-storage forward (hig_dims.storage == HIG_STORE_BF16): the same launch sequence over bf16 activations.
// Byte layouts (256-byte granules).  Inference only: one layer's buffers serve every layer, the residual stream `h`
// is updated in place by the stylization-out GEMMs (each thread reads and writes its own 8 columns).
// ==========================================================================================================
namespace {

inline int64_t alb(int64_t bytes) { return (bytes + 255) & ~(int64_t)255; }

struct Fwd16Layout {
  int64_t te32, te16, teh16, semb16, ss, h32, h, xn, qkv, A1, At1, kst1, cscr, y, a, qc, f1, lenp, tok0, stats, total;
};
Fwd16Layout fwd16_layout(const Dims& D) {
  Fwd16Layout w;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += alb(n); return r; };
  w.te32 = take((int64_t)D.B * D.d * 4);
  w.te16 = take((int64_t)D.B * D.d * 2);
  w.teh16 = take((int64_t)D.B * D.E * 2);
  w.semb16 = take((int64_t)D.B * D.E * 2);
  w.ss = take((int64_t)D.B * D.nsty * D.L * 2 * D.d * 4);
  w.h32 = take(D.M * D.d * 4 > (int64_t)D.d * 544 * 2 ? D.M * D.d * 4 : (int64_t)D.d * 544 * 2);   // joint_embed: padded bf16
                                        // weight of hig_joint_embed_bf16 (d x Fp), or the fp32 output of the fallback GEMM
  w.h = take(D.M * D.d * 2);
  w.xn = take(D.M * D.d * 2);
  w.qkv = take(D.M * 3 * D.d * 2);
  w.A1 = take((int64_t)D.B * D.H * D.hd * D.hd * 4);
  w.At1 = take((int64_t)D.B * D.H * D.hd * D.hd * 2);   // the same, transposed, bf16 (linattn16.hip)
  w.kst1 = take((int64_t)D.B * D.d * 2 * 4);
  w.cscr = take(hig_linattn_ctx_scratch_floats(D.B, D.T, D.H, D.hd) * 4);
  w.y = take(D.M * D.d * 2);
  w.a = take(D.M * D.d * 2);
  w.qc = take(D.M * D.d * 2);
  w.f1 = take(D.M * D.ff * 2);
  w.lenp = take((int64_t)D.B * 8);           // two-person: lengths with the two halves swapped (the partner's mask)
  w.tok0 = take((int64_t)D.B * D.d * 4);     // two-person: joint_embed2 of the init-pose rows (fp32) before they enter h
  w.stats = take(D.M * (D.d / 128 > 4 ? D.d / 128 : 4) * 2 * 4);   // LayerNorm fold: (sum, centred sum of squares) per row and 128-column panel of h
  w.total = o;
  return w;
}

// dst[r][0..n) = bf16(src[r][0..n)) for `rows` rows with their own leading dimensions (the init-pose rows of h)
__global__ void cast_rows_bf16_kernel(const float* __restrict__ src, int64_t lds, __bf16* __restrict__ dst, int64_t ldd,
                                      int rows, int n) {
  const int64_t total = (int64_t)rows * n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / n, c = i % n;
    dst[r * ldd + c] = (__bf16)src[r * lds + c];
  }
}

__global__ void zero_u32_kernel(unsigned* __restrict__ p, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__global__ void copy2d_f32_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int rows, int cols) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds + c];
}

// tok0[b][0..n) = float(buf[b][0][0..n)) for a bf16 buf [B][T][n] (if tok0), then zeroes that row in buf (if zero): the init-pose
// rows of the two-person model's residual-stream gradient
__global__ void tok0_bf16_kernel(__bf16* __restrict__ buf, int64_t sample_stride, int B, int n, float* __restrict__ tok0, int zero) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * n) return;
  const int b = idx / n, c = idx % n;
  __bf16* p = buf + (int64_t)b * sample_stride + c;
  if (tok0) tok0[idx] = (float)*p;
  if (zero) *p = (__bf16)0.f;
}

struct Text16Layout {
  int64_t xfn, kv, kv_stride, cscr, layer0, lstride, Ac, Atc, kstc, kvall, total;
};
Text16Layout text16_layout(const Dims& D) {
  Text16Layout t;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += alb(n); return r; };
  t.xfn = take(D.Mt * D.Lt * 2);
  // linear attention consumes key / value right away (into A_c); full attention reads them at every step: kept per layer
  t.kv_stride = D.full ? alb(D.Mt * 2 * D.d * 2) : 0;
  t.kv = take(D.full ? t.kv_stride * D.L : D.Mt * 2 * D.d * 2);
  t.cscr = take(hig_linattn_ctx_scratch_floats(D.B, D.N, D.H, D.hd) * 4);
  t.layer0 = o;
  o = 0;
  t.Ac = take((int64_t)D.B * D.H * D.hd * D.hd * 4);
  t.Atc = take((int64_t)D.B * D.H * D.hd * D.hd * 2);   // transposed bf16 copy (linattn16.hip)
  t.kstc = take((int64_t)D.B * D.d * 2 * 4);
  t.lstride = o;
  t.total = t.layer0 + t.lstride * D.L;
  // linear attention: room for the batched form of the text side (key/value projections of ALL layers as one (Mt, L 2d) matrix)
  t.kvall = -1;
  if (!D.full) {
    o = t.total;
    t.kvall = take(D.Mt * 2 * D.d * D.L * 2);
    t.total = o;
  }
  return t;
}

struct G16 {  // small builder for bf16 gemm descriptors
  hig_gemm16_desc g;
  G16(const void* X, int64_t ldx, const void* Y, int64_t ldy, void* C, int64_t ldc, int64_t I, int64_t J, int64_t R) {
    memset(&g, 0, sizeof(g));
    g.X = X; g.ldx = ldx; g.Y = Y; g.ldy = ldy; g.C = C; g.ldc = ldc;
    g.I = (int)I; g.J = (int)J; g.R = (int)R; g.epi = HIG_EPI_NONE;
  }
  G16& epi(int e, const float* bias) { g.epi = e; g.bias = bias; return *this; }
  G16& res16(const void* r, int64_t ldr) { g.res = r; g.ldr = ldr; g.res_f32 = 0; return *this; }
  G16& res32(const float* r, int64_t ldr) { g.res = r; g.ldr = ldr; g.res_f32 = 1; return *this; }
  G16& out32() { g.c_f32 = 1; return *this; }
};

int64_t fwd16_total(const Dims& D) { return fwd16_layout(D).total; }
int64_t text16_total(const Dims& D) { return text16_layout(D).total; }

inline const void* P16(const void* const* t, int idx) { return t[idx]; }
inline const void* PL16(const void* const* t, int l, int idx) { return t[HIG_NGLOBAL + l * HIG_NLAYER + idx]; }

}  // namespace

// Context build of the bf16-storage forward: the bf16-matrix-core kernel of linattn16.hip where it is built (head dim 64 / 128),
// else the fp32-MFMA kernels with bf16 loads (HIG_CTX16=0 forces those).
static int ctx16(const Dims& D, const void* K, const void* V, int64_t ld, int B, int rows, const int64_t* length, float* A,
                 float* kstat, float* scratch, void* At16, hig_stream_t stream) {
  static const int mm16 = getenv("HIG_CTX16") ? atoi(getenv("HIG_CTX16")) : 1;   // tuning knob
  if (mm16 && (D.hd == 64 || D.hd == 128)) return hig_linattn_ctx_mm16(K, V, ld, B, rows, D.H, D.hd, length, A, kstat, At16, stream);
  return hig_linattn_ctx_bf16(K, V, ld, B, rows, D.H, D.hd, length, A, kstat, scratch, At16, stream);
}

// layer_done (nullable): event l is recorded on `st` behind layer l's launches (the forked form of hig_denoiser_fwd_bf16_x)
static int text_context16_impl(const Dims& D, const void* const* params, const void* const* params16, const void* const* derived,
                               const float* xf_out, void* textctx, hipStream_t st, hipEvent_t* layer_done);
extern "C" int hig_text_context_bf16(const hig_dims* dims, const void* const* params, const void* const* params16,
                                     const float* xf_out, void* textctx, hig_stream_t stream) {
  Dims D;
  HIG_TRY(check_dims(dims, D));
  HIG_REQUIRE(D.bf16, "hig_text_context_bf16: dims->storage must be HIG_STORE_BF16");
  HIG_REQUIRE(params && params16 && xf_out && textctx, "hig_text_context_bf16: null argument");
  return text_context16_impl(D, params, params16, nullptr, xf_out, textctx, hig_stream(stream), nullptr);
}
// derived (nullable): the caller's derived-operand table of hig_denoiser_fwd_bf16_x; entries [13 L + 1 .. 13 L + 4] select the
// BATCHED form (text_context_impl above: one key/value GEMM over the stacked text_norm-folded bf16 weights of all layers, one
// context build over L H heads)
static bool text16_batched(const Dims& D, const void* const* derived) {
  static const int batch_env = getenv("HIG_TEXT_BATCH") ? atoi(getenv("HIG_TEXT_BATCH")) : 1;   // tuning knob
  return batch_env && derived && !D.full && derived[13 * D.L + 1] && derived[13 * D.L + 2] && derived[13 * D.L + 3] && derived[13 * D.L + 4];
}
static int text_context16_impl(const Dims& D, const void* const* params, const void* const* params16, const void* const* derived,
                               const float* xf_out, void* textctx, hipStream_t st, hipEvent_t* layer_done) {
  const Text16Layout tl = text16_layout(D);
  char* base = static_cast<char*>(textctx);
  hig_stream_t stream = reinterpret_cast<hig_stream_t>(st);
  void* xfn = base + tl.xfn;
  if (text16_batched(D, derived) && tl.kvall >= 0) {
    char* kvall = base + tl.kvall;
    const int64_t ldkv = (int64_t)D.L * 2 * D.d;
    HIG_TRY(hig_ln_bf16(xf_out, 1, D.Lt, D.Mt, D.Lt, static_cast<const float*>(derived[13 * D.L + 3]),
                        static_cast<const float*>(derived[13 * D.L + 4]), nullptr, 0, 0, 0, xfn, D.Lt, stream));
    HIG_TRY(hig_gemm16_launch(G16(xfn, D.Lt, derived[13 * D.L + 1], D.Lt, kvall, ldkv, D.Mt, ldkv, D.Lt)
                                  .epi(HIG_EPI_BIAS, static_cast<const float*>(derived[13 * D.L + 2])).g, st));
    const char* Kall = kvall;
    const char* Vall = kvall + (int64_t)D.L * D.d * 2;
    int rc = hig_linattn_ctx16_groups(Kall, Vall, ldkv, D.B, D.N, D.H, D.L, D.hd,
                                      reinterpret_cast<float*>(base + tl.layer0 + tl.Ac), tl.lstride / 4,
                                      reinterpret_cast<float*>(base + tl.layer0 + tl.kstc), tl.lstride / 4,
                                      base + tl.layer0 + tl.Atc, tl.lstride / 2, st);
    if (rc < 0) return rc;
    for (int l = 0; l < D.L; ++l) {
      if (rc != HIG_OK) {   // (head dim not on the bf16 matrix-core kernel: one launch per layer)
        float* Ac = reinterpret_cast<float*>(base + tl.layer0 + tl.lstride * l + tl.Ac);
        float* kstc = reinterpret_cast<float*>(base + tl.layer0 + tl.lstride * l + tl.kstc);
        HIG_TRY(ctx16(D, Kall + (int64_t)l * D.d * 2, Vall + (int64_t)l * D.d * 2, ldkv, D.B, D.N, nullptr, Ac, kstc,
                      reinterpret_cast<float*>(base + tl.cscr), base + tl.layer0 + tl.lstride * l + tl.Atc, stream));
      }
      if (layer_done && hipEventRecord(layer_done[l], st) != hipSuccess) return hig_set_error(HIG_EHIP, "hipEventRecord failed");
    }
    return HIG_OK;
  }
  for (int l = 0; l < D.L; ++l) {
    char* kv = base + tl.kv + tl.kv_stride * l;
    float* Ac = reinterpret_cast<float*>(base + tl.layer0 + tl.lstride * l + tl.Ac);
    float* kstc = reinterpret_cast<float*>(base + tl.layer0 + tl.lstride * l + tl.kstc);
    // text_norm (per layer), [key; value] projection, softmax over the N tokens, A_c = k^T v   (transformer.py:146-152)
    HIG_TRY(hig_ln_bf16(xf_out, 1, D.Lt, D.Mt, D.Lt, PL(params, l, HIG_L_CA_TNORM_W), PL(params, l, HIG_L_CA_TNORM_B),
                        nullptr, 0, 0, 0, xfn, D.Lt, stream));
    HIG_TRY(hig_gemm16_launch(G16(xfn, D.Lt, PL16(params16, l, HIG_L_CA_KV_W), D.Lt, kv, 2 * D.d, D.Mt, 2 * D.d, D.Lt)
                                  .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_CA_KV_B)).g, st));
    if (!D.full)
      HIG_TRY(ctx16(D, kv, kv + (int64_t)D.d * 2, 2 * D.d, D.B, D.N, nullptr, Ac, kstc,
                    reinterpret_cast<float*>(base + tl.cscr), base + tl.layer0 + tl.lstride * l + tl.Atc, stream));
    if (layer_done && hipEventRecord(layer_done[l], st) != hipSuccess) return hig_set_error(HIG_EHIP, "hipEventRecord failed");
  }
  return HIG_OK;
}

static int denoiser_fwd16_impl(const hig_dims* dims, const void* const* params, const void* const* params16, const void* const* lnfold,
                               const float* x, const int64_t* t, const int64_t* length, const float* xf_proj, const float* xf_out_for_text,
                               const void* textctx, float* out, void* workspace, hig_stream_t stream);
extern "C" int hig_denoiser_fwd_bf16(const hig_dims* dims, const void* const* params, const void* const* params16,
                                     const void* const* lnfold, const float* x, const int64_t* t, const int64_t* length,
                                     const float* xf_proj,
                                     const void* textctx, float* out, void* workspace, hig_stream_t stream) {
  return denoiser_fwd16_impl(dims, params, params16, lnfold, x, t, length, xf_proj, nullptr, textctx, out, workspace, stream);
}
// The general form (include/hig.h): xf_out (nullable) = compute the text side here, on a library-owned stream next to the
// first layers.
extern "C" int hig_denoiser_fwd_bf16_x(const hig_dims* dims, const void* const* params, const void* const* params16,
                                       const void* const* lnfold, const float* x, const int64_t* t, const int64_t* length,
                                       const float* xf_proj, const float* xf_out, void* textctx, float* out, void* workspace,
                                       hig_stream_t stream) {
  return denoiser_fwd16_impl(dims, params, params16, lnfold, x, t, length, xf_proj, xf_out, textctx, out, workspace, stream);
}
static int denoiser_fwd16_impl(const hig_dims* dims, const void* const* params, const void* const* params16, const void* const* lnfold,
                               const float* x, const int64_t* t, const int64_t* length, const float* xf_proj, const float* xf_out_for_text,
                               const void* textctx, float* out, void* workspace, hig_stream_t stream) {
  Dims D;
  HIG_TRY(check_dims(dims, D));
  HIG_REQUIRE(D.bf16, "hig_denoiser_fwd_bf16: dims->storage must be HIG_STORE_BF16");
  HIG_REQUIRE(params && params16 && x && t && xf_proj && textctx && out && workspace, "hig_denoiser_fwd_bf16: null argument");
  const Fwd16Layout w = fwd16_layout(D);
  const Text16Layout tl = text16_layout(D);
  char* ws = static_cast<char*>(workspace);
  const char* tc = static_cast<const char*>(textctx);
  hipStream_t st = hig_stream(stream);
  const int d = D.d, E = D.E;
  const int64_t M = D.M;
  const int64_t ss_ld = (int64_t)D.nsty * D.L * 2 * d;
  float* ss = reinterpret_cast<float*>(ws + w.ss);

  // Everything that hangs off the B conditioning rows instead of the M frame rows -- the embedding chain (its last GEMM reads
  // every stylization block's (2 d, E) weight: 604 MB at the config-5 shape, HBM-bound) and the cross-attention text side --
  // is first needed a few launches into layer 0 / in front of layer l's cross-attention.  Launched eagerly with the library's
  // streams available, both run on the third stream next to the frame-row launches (events only; HIG_FWD16_FORK=0, a capture
  // in progress or no side streams: everything in order on the caller's stream).
  // Measured (tools/fwd16_fork.sh, same call, per-call text: off / text / both): config 2 B = 64 1.718 / 1.667 / 1.657-1.672 ms,
  // B = 32 1.158 / 1.134 / 1.121; config-5 shape 4.83 / 4.71 / 4.66-4.70; with the text side cached, forking the embedding
  // chain alone COSTS 2-6 % at config 2 (its 35 us of launches are shorter than the two event waits they add): it is forked
  // only when its modulation weight is large (>= 256 MB: the d = 1024 models).
  static const int fork_knob = getenv("HIG_FWD16_FORK") ? atoi(getenv("HIG_FWD16_FORK")) : -1;   // tuning knob: bit 0 embedding chain, bit 1 text side
  // (round 6: the BATCHED text side -- three launches, two of them chip-wide -- is faster in front of the frame-row launches than
  // next to them: B = 64 1.488 against 1.540 ms, per-layer form 1.539 forked / 1.603 in front; not forked by default)
  const bool batched_text = xf_out_for_text && text16_batched(D, lnfold);
  const int fork_env = fork_knob >= 0 ? fork_knob : ((batched_text ? 0 : 2) | (((int64_t)E * ss_ld * 2 >= (256ll << 20)) ? 1 : 0));
  SideStream* fs = (fork_env && D.L < kMaxTextLayers) ? side_stream_for_current_device(st) : nullptr;
  if (fs) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) fs = nullptr;
  }
  const bool fork_emb = fs && (fork_env & 1), fork_text = fs && (fork_env & 2) && xf_out_for_text;
  if (fork_emb || fork_text)
    if (hipEventRecord(fs->ready, st) != hipSuccess || hipStreamWaitEvent(fs->s3, fs->ready, 0) != hipSuccess)
      return hig_set_error(HIG_EHIP, "forward fork failed");
  hipStream_t se = fork_emb ? fs->s3 : st;
  hig_stream_t hse = reinterpret_cast<hig_stream_t>(se);
  hipEvent_t emb_ev = fork_emb ? fs->text_done[kMaxTextLayers - 1] : nullptr;
  // whatever was enqueued on the third stream is joined before an error leaves
  auto join_side = [&](int rc) -> int {
    if (fork_emb || fork_text) {
      (void)hipEventRecord(fs->text_done[kMaxTextLayers - 1], fs->s3);
      (void)hipStreamWaitEvent(st, fs->text_done[kMaxTextLayers - 1], 0);
    }
    return rc;
  };
#define HIG_TRY_SIDE(expr) do { const int rc_ = (expr); if (rc_ != HIG_OK) return join_side(rc_); } while (0)

  // K0: emb = time_embed(timestep_embedding(t)) + xf_proj; only silu(emb) is consumed (by every stylization block):
  //     te -> silu(Lin0) -> silu(Lin2 + xf_proj) -> ONE GEMM for all 3L (scale, shift) pairs      (transformer.py:345-349,415,81-83)
  HIG_TRY_SIDE(hig_timestep_embedding_bf16(t, D.B, d, ws + w.te16, hse));
  HIG_TRY_SIDE(hig_gemm16_launch(G16(ws + w.te16, d, P16(params16, HIG_P_TE0_W), d, ws + w.teh16, E, D.B, E, d)
                                     .epi(HIG_EPI_BIAS_SILU, P(params, HIG_P_TE0_B)).g, se));
  HIG_TRY_SIDE(hig_gemm16_launch(G16(ws + w.teh16, E, P16(params16, HIG_P_TE2_W), E, ws + w.semb16, E, D.B, E, E)
                                     .epi(HIG_EPI_BIAS_RES_SILU, P(params, HIG_P_TE2_B)).res32(xf_proj, E).g, se));
  HIG_TRY_SIDE(hig_gemm16_launch(G16(ws + w.semb16, E, P16(params16, HIG_P_STY_EMB_W), E, ss, ss_ld, D.B, ss_ld, E)
                                     .epi(HIG_EPI_BIAS, P(params, HIG_P_STY_EMB_B)).out32().g, se));
  if (fork_emb && hipEventRecord(emb_ev, se) != hipSuccess) return join_side(hig_set_error(HIG_EHIP, "hipEventRecord failed"));
  hipEvent_t* text_ev = nullptr;
  if (xf_out_for_text) {
    HIG_TRY_SIDE(text_context16_impl(D, params, params16, lnfold, xf_out_for_text, const_cast<void*>(textctx), fork_text ? fs->s3 : st,
                                     fork_text ? fs->text_done : nullptr));
    if (fork_text) text_ev = fs->text_done;
  }
  bool emb_joined = !fork_emb;
  auto need_emb = [&]() -> int {   // in front of the first launch that reads the (scale, shift) table
    if (emb_joined) return HIG_OK;
    emb_joined = true;
    return hipStreamWaitEvent(st, emb_ev, 0) == hipSuccess ? HIG_OK : hig_set_error(HIG_EHIP, "embedding join failed");
  };
  auto frame_rows = [&]() -> int {
  // K1: h0 = joint_embed(x) + sequence_embedding[:T]: fp32 operands (x is the fp32 DDPM state, F = 150 rows are not
  //     16-byte aligned), result rounded once into the bf16 residual stream
  static const int joint16 = getenv("HIG_JOINT16") ? atoi(getenv("HIG_JOINT16")) : 1;   // tuning knob
  if (joint16 && d % 128 == 0 && D.F <= 512) {
    // own kernel pair (weight padded / rounded to bf16, x rounded in LDS, bf16 MFMA): 31 -> ~8 us at B = 32
    if (lnfold && lnfold[13 * D.L])   // (weight padded / rounded once, next to the caller's bf16 shadow)
      HIG_TRY(hig_joint_embed_bf16_w(x, M, D.F, lnfold[13 * D.L], P(params, HIG_P_JOINT_B), P(params, HIG_P_SEQ_EMB), d, D.T,
                                     D.two ? 1 : 0, ws + w.h, d, d, stream));
    else
      HIG_TRY(hig_joint_embed_bf16(x, M, D.F, P(params, HIG_P_JOINT_W), P(params, HIG_P_JOINT_B), P(params, HIG_P_SEQ_EMB), d,
                                   D.T, D.two ? 1 : 0, ws + w.h, d, d, ws + w.h32, stream));
  } else {
    float* h32 = reinterpret_cast<float*>(ws + w.h32);
    G ge(x, D.F, 0, P(params, HIG_P_JOINT_W), D.F, 0, h32, d, M, d, D.F);
    ge.epi(HIG_EPI_BIAS_POS, P(params, HIG_P_JOINT_B)).pos(P(params, HIG_P_SEQ_EMB), d, D.T);
    ge.g.pos_shift = D.two ? 1 : 0;
    HIG_TRY(hig_gemm_launch(ge.g, 1, nullptr, st));
    HIG_TRY(hig_cast_bf16(h32, ws + w.h, M * d, stream));
  }
  const int64_t* len_partner = nullptr;
  const int Bp = D.B / 2;
  if (D.two) {
    // token 0 is the init-pose row: joint_embed2 on its first 4 features, no positional term
    // (interaction_transformer.py:596); fp32 GEMM over the B rows, then into the bf16 residual stream
    float* tok0 = reinterpret_cast<float*>(ws + w.tok0);
    HIG_TRY(hig_gemm_launch(G(x, (int64_t)D.T * D.F, 0, P(params, HIG_P_JOINT2_W), 4, 0, tok0, d, D.B, d, 4)
                                .epi(HIG_EPI_BIAS, P(params, HIG_P_JOINT2_B)).g, 1, nullptr, st));
    hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3((unsigned)(((int64_t)D.B * d + 255) / 256)), dim3(256), 0, st, tok0, (int64_t)d,
                       reinterpret_cast<__bf16*>(ws + w.h), (int64_t)D.T * d, D.B, d);
    HIG_CHECK_LAUNCH();
    int64_t* lp = reinterpret_cast<int64_t*>(ws + w.lenp);
    hipLaunchKernelGGL(swap_halves_i64_kernel, dim3((D.B + 255) / 256), dim3(256), 0, st, length, D.B, (int64_t)D.T, lp);
    HIG_CHECK_LAUNCH();
    len_partner = lp;
  }
  void* h = ws + w.h;
  void* xn = ws + w.xn;
  char* qkv = ws + w.qkv;
  void* y = ws + w.y;
  void* a = ws + w.a;
  void* qc = ws + w.qc;
  void* f1 = ws + w.f1;
  float* A1 = reinterpret_cast<float*>(ws + w.A1);
  float* kst1 = reinterpret_cast<float*>(ws + w.kst1);
  float* cscr = reinterpret_cast<float*>(ws + w.cscr);
  // one stylization block: h += Lin_out( silu( LN(y) (1 + scale) + shift ) )      (transformer.py:81-85)
  // LayerNorm fold (see include/hig.h): when the NEXT consumer of the residual stream is a LayerNorm + Linear pair, the
  // stylization-out GEMM also writes the row statistics of the new h (`want_stats` set by the layer loop below)
  const bool fold = lnfold && lnfold[0] && hig_gemm_ws16_lnfold_ok(M, d);
  float* stats = reinterpret_cast<float*>(ws + w.stats);
  bool want_stats = false, have_stats = false;
  auto sty_out = [&](int l, int out_w, int out_b) -> int {
    G16 g(a, d, PL16(params16, l, out_w), d, h, d, M, d, d);
    g.epi(HIG_EPI_BIAS_RES, PL(params, l, out_b)).res16(h, d);
    if (fold && want_stats) g.g.row_stats_out = stats;
    have_stats = fold && want_stats;
    return hig_gemm16_launch(g.g, st);
  };
  // xn-free projection of LN(h): out = LN(h) W^T + b through the folded operands (k = 0: q/k/v, k = 1: cross-attention query,
  // k = 2: q/k/v of the person <-> person attention)
  auto ln_proj = [&](int l, int k, int norm_w, int norm_b, int lin_w, int lin_b, void* outp, int64_t ncols) -> int {
    if (have_stats && lnfold[13 * l + 3 * k]) {
      G16 g(h, d, lnfold[13 * l + 3 * k], d, outp, ncols, M, ncols, d);
      g.epi(HIG_EPI_BIAS, static_cast<const float*>(lnfold[13 * l + 3 * k + 2]));
      g.g.row_stats_in = stats;
      g.g.ln_colsum = static_cast<const float*>(lnfold[13 * l + 3 * k + 1]);
      return hig_gemm16_launch(g.g, st);
    }
    HIG_TRY(hig_ln_bf16(h, 0, d, M, d, PL(params, l, norm_w), PL(params, l, norm_b), nullptr, 0, 0, 0, xn, d, stream));
    return hig_gemm16_launch(G16(xn, d, PL16(params16, l, lin_w), d, outp, ncols, M, ncols, d).epi(HIG_EPI_BIAS, PL(params, l, lin_b)).g, st);
  };
  auto stylize = [&](int l, int slot, int norm_w, int norm_b, int out_w, int out_b) -> int {
    HIG_TRY(need_emb());
    const float* ssl = ss + (int64_t)(D.nsty * l + slot) * 2 * d;
    HIG_TRY(hig_ln_bf16(y, 0, d, M, d, PL(params, l, norm_w), PL(params, l, norm_b), ssl, ss_ld, d, D.T, a, d, stream));
    return sty_out(l, out_w, out_b);
  };
  // attention output -> stylization block.  With 4 or 8 heads the `y = q A` product, the LayerNorm, the modulation and
  // the SiLU are ONE kernel (y never leaves the chip); otherwise apply + row kernel.
  // (opt-in: measured equal at B = 64 and 4 % slower at B = 32 -- its fp32 MFMAs serialise 128 per wave behind poorly
  // coalesced query loads; profiles/r02_notes.md)
  // HIG_FUSE_APPLY: 2 (default) = the bf16-matrix-core kernel of linattn16.hip where it is built (head dim 64), 1 = the
  // fp32-MFMA fused kernel, 0 = apply + row kernel
  static const int fuse_env = getenv("HIG_FUSE_APPLY") ? atoi(getenv("HIG_FUSE_APPLY")) : 2;   // tuning knob
  const bool fuse_mm16 = fuse_env == 2 && (D.hd == 64 || D.hd == 128) && (D.H == 4 || D.H == 8);
  const bool fuse_apply = (fuse_env == 1 || fuse_mm16) && (D.H == 4 || D.H == 8);
  // hig_attn_out16 / hig_rows_out16 (a whole stylization block as one launch) for the small batches: there the launches are
  // bound by the per-launch floor (~4.4 us) and a fetch-bound projection.  Two of their workgroups (one per 32 rows of a sample)
  // are resident per CU; same-call A/B, forward, fused vs not: B = 32 (224 workgroups) 0.996 vs 1.097 ms, B = 40 1.221 vs
  // 1.262, B = 64 (448) 1.521 vs 1.614, B = 73 1.640 vs 1.733, B = 96 (672) 2.067 vs 2.100, B = 128 (896) 2.512 vs 2.516,
  // B = 256 4.393 vs 4.377: used up to 3 workgroups per CU (768).  HIG_FUSE_OUT=0 switches it off, n >= 2 moves the limit to n per CU.
  static const int fuse_out_env = getenv("HIG_FUSE_OUT") ? atoi(getenv("HIG_FUSE_OUT")) : 1;   // tuning knob
  const bool fuse_out = fuse_out_env && fuse_mm16 && d == 512 && D.hd == 64 && D.H == 8 &&
                        (int64_t)((D.T + 31) / 32) * D.B <= (int64_t)hig_chip_cus() * (fuse_out_env >= 2 ? fuse_out_env : 3);
  auto attend = [&](int l, int slot, const void* q, int64_t ldq, const float* ctx, const void* ctx_t16, int norm_w, int norm_b,
                    int out_w, int out_b) -> int {
    HIG_TRY(need_emb());
    if (fuse_apply) {
      const float* ssl = ss + (int64_t)(D.nsty * l + slot) * 2 * d;
      const void* wfrag = (fuse_out && slot < 3 && lnfold) ? lnfold[13 * l + 9 + slot] : nullptr;
      if (wfrag) {   // apply + stylization front + output projection + residual update as ONE launch
        const bool st_out = fold && want_stats;
        HIG_TRY(hig_attn_out16(q, ldq, ctx_t16, PL(params, l, norm_w), PL(params, l, norm_b), ssl, ss_ld, d, wfrag, PL(params, l, out_b),
                               h, d, st_out ? stats : nullptr, D.B, D.T, D.H, D.hd, stream));
        have_stats = st_out;
        return HIG_OK;
      }
      if (fuse_mm16)
        HIG_TRY(hig_linattn_apply_sty_mm16(q, ldq, ctx_t16, PL(params, l, norm_w), PL(params, l, norm_b), ssl, ss_ld, d, a, d,
                                           D.B, D.T, D.H, D.hd, stream));
      else
        HIG_TRY(hig_linattn_apply_sty_bf16(q, ldq, ctx, PL(params, l, norm_w), PL(params, l, norm_b), ssl, ss_ld, d, a, d,
                                           D.B, D.T, D.H, D.hd, stream));
      return sty_out(l, out_w, out_b);
    }
    HIG_TRY(hig_linattn_apply_bf16(q, ldq, ctx, y, d, D.B, D.T, D.H, D.hd, stream));
    return stylize(l, slot, norm_w, norm_b, out_w, out_b);
  };
  for (int l = 0; l < D.L; ++l) {
    // ---- self attention (transformer.py:101-119) ----
    HIG_TRY(ln_proj(l, 0, HIG_L_SA_NORM_W, HIG_L_SA_NORM_B, HIG_L_SA_QKV_W, HIG_L_SA_QKV_B, qkv, 3 * d));
    want_stats = true;                           // the self-attention stylization block feeds the cross-attention LayerNorm
    if (D.full) {   // no_eff=True: softmax over the T keys, query-axis mask constant (transformer.py:208-227)
      HIG_TRY(hig_fullattn_fwd_bf16(qkv, 3 * d, qkv + (int64_t)d * 2, qkv + (int64_t)2 * d * 2, 3 * d, D.B, D.T, D.T, D.H, D.hd,
                                    length, y, d, stream));
      HIG_TRY(stylize(l, 0, HIG_L_SA_STY_NORM_W, HIG_L_SA_STY_NORM_B, HIG_L_SA_STY_OUT_W, HIG_L_SA_STY_OUT_B));
    } else {
      HIG_TRY(ctx16(D, qkv + (int64_t)d * 2, qkv + (int64_t)2 * d * 2, 3 * d, D.B, D.T, length, A1, kst1, cscr, ws + w.At1, stream));
      HIG_TRY(attend(l, 0, qkv, 3 * d, A1, ws + w.At1, HIG_L_SA_STY_NORM_W, HIG_L_SA_STY_NORM_B, HIG_L_SA_STY_OUT_W, HIG_L_SA_STY_OUT_B));
    }
    // ---- cross attention to the text context (transformer.py:135-155) ----
    HIG_TRY(ln_proj(l, 1, HIG_L_CA_NORM_W, HIG_L_CA_NORM_B, HIG_L_CA_Q_W, HIG_L_CA_Q_B, qc, d));
    want_stats = D.two == 1;                     // (its stylization block feeds the interaction LayerNorm of the two-person model, else nothing folded)
    if (text_ev && hipStreamWaitEvent(st, text_ev[l], 0) != hipSuccess) return hig_set_error(HIG_EHIP, "text join failed");
    if (D.full) {   // softmax over the N text tokens, no mask (transformer.py:242-262)
      const char* kvl = tc + tl.kv + tl.kv_stride * l;
      HIG_TRY(hig_fullattn_fwd_bf16(qc, d, kvl, kvl + (int64_t)d * 2, 2 * d, D.B, D.T, D.N, D.H, D.hd, nullptr, y, d, stream));
      HIG_TRY(stylize(l, 1, HIG_L_CA_STY_NORM_W, HIG_L_CA_STY_NORM_B, HIG_L_CA_STY_OUT_W, HIG_L_CA_STY_OUT_B));
    } else {
      HIG_TRY(attend(l, 1, qc, d, reinterpret_cast<const float*>(tc + tl.layer0 + tl.lstride * l + tl.Ac),
                     tc + tl.layer0 + tl.lstride * l + tl.Atc, HIG_L_CA_STY_NORM_W, HIG_L_CA_STY_NORM_B, HIG_L_CA_STY_OUT_W, HIG_L_CA_STY_OUT_B));
    }
    if (D.two == 1) {
      // ---- person <-> person linear cross attention (interaction_transformer.py:181-205): queries from the own
      // stream, key / value from the partner's (same LayerNorm on both), key softmax masked with the consumer's length
      HIG_TRY(ln_proj(l, 2, HIG_L_INT_NORM_W, HIG_L_INT_NORM_B, HIG_L_INT_QKV_W, HIG_L_INT_QKV_B, qkv, 3 * d));
      want_stats = false;                        // (the interaction stylization block: no folded consumer behind it)
      HIG_TRY(ctx16(D, qkv + (int64_t)d * 2, qkv + (int64_t)2 * d * 2, 3 * d, D.B, D.T, len_partner, A1, kst1, cscr,
                    fuse_mm16 ? ws + w.At1 : nullptr, stream));
      const int64_t halfA = (int64_t)Bp * D.H * D.hd * D.hd, halfM = (int64_t)Bp * D.T;
      if (fuse_out && lnfold && lnfold[13 * l + 11]) {
        // each half of the batch against the OTHER half's context matrices; apply + stylization block + residual update fused
        const float* ssl = ss + (int64_t)(D.nsty * l + 2) * 2 * d;
        const char* At = ws + w.At1;
        char* hb = static_cast<char*>(h);
        HIG_TRY(hig_attn_out16(qkv, 3 * d, At + halfA * 2, PL(params, l, HIG_L_INT_STY_NORM_W), PL(params, l, HIG_L_INT_STY_NORM_B), ssl,
                               ss_ld, d, lnfold[13 * l + 11], PL(params, l, HIG_L_INT_STY_OUT_B), hb, d, nullptr, Bp, D.T, D.H, D.hd, stream));
        HIG_TRY(hig_attn_out16(qkv + halfM * 3 * d * 2, 3 * d, At, PL(params, l, HIG_L_INT_STY_NORM_W), PL(params, l, HIG_L_INT_STY_NORM_B),
                               ssl + (int64_t)Bp * ss_ld, ss_ld, d, lnfold[13 * l + 11], PL(params, l, HIG_L_INT_STY_OUT_B),
                               hb + halfM * d * 2, d, nullptr, Bp, D.T, D.H, D.hd, stream));
        have_stats = false;
      } else if (fuse_mm16) {
        // each half of the batch against the OTHER half's context matrices, apply + stylization front as one kernel
        const float* ssl = ss + (int64_t)(D.nsty * l + 2) * 2 * d;
        const char* At = ws + w.At1;
        HIG_TRY(hig_linattn_apply_sty_mm16(qkv, 3 * d, At + halfA * 2, PL(params, l, HIG_L_INT_STY_NORM_W), PL(params, l, HIG_L_INT_STY_NORM_B),
                                           ssl, ss_ld, d, a, d, Bp, D.T, D.H, D.hd, stream));
        HIG_TRY(hig_linattn_apply_sty_mm16(qkv + halfM * 3 * d * 2, 3 * d, At, PL(params, l, HIG_L_INT_STY_NORM_W),
                                           PL(params, l, HIG_L_INT_STY_NORM_B), ssl + (int64_t)Bp * ss_ld, ss_ld, d,
                                           static_cast<char*>(a) + halfM * d * 2, d, Bp, D.T, D.H, D.hd, stream));
        HIG_TRY(sty_out(l, HIG_L_INT_STY_OUT_W, HIG_L_INT_STY_OUT_B));
      } else {
        HIG_TRY(hig_linattn_apply_bf16(qkv, 3 * d, A1 + halfA, y, d, Bp, D.T, D.H, D.hd, stream));
        HIG_TRY(hig_linattn_apply_bf16(qkv + halfM * 3 * d * 2, 3 * d, A1, static_cast<char*>(y) + halfM * d * 2, d, Bp, D.T, D.H, D.hd,
                                       stream));
        HIG_TRY(stylize(l, 2, HIG_L_INT_STY_NORM_W, HIG_L_INT_STY_NORM_B, HIG_L_INT_STY_OUT_W, HIG_L_INT_STY_OUT_B));
      }
    }
    // ---- FFN (transformer.py:167-170) ----
    HIG_TRY(hig_gemm16_launch(G16(h, d, PL16(params16, l, HIG_L_FFN_W1), d, f1, D.ff, M, D.ff, d)
                                  .epi(HIG_EPI_BIAS_GELU, PL(params, l, HIG_L_FFN_B1)).g, st));
    HIG_TRY(hig_gemm16_launch(G16(f1, D.ff, PL16(params16, l, HIG_L_FFN_W2), D.ff, y, d, M, d, D.ff)
                                  .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_FFN_B2)).g, st));
    want_stats = l + 1 < D.L;                    // the FFN stylization block feeds the next layer's self-attention LayerNorm
    if (fuse_out && lnfold && lnfold[13 * l + 12]) {   // stylization front + output projection + residual update as ONE launch
      const bool st_out = fold && want_stats;
      HIG_TRY(hig_rows_out16(y, d, PL(params, l, HIG_L_FFN_STY_NORM_W), PL(params, l, HIG_L_FFN_STY_NORM_B),
                             ss + (int64_t)(D.nsty * l + D.nsty - 1) * 2 * d, ss_ld, d, lnfold[13 * l + 12], PL(params, l, HIG_L_FFN_STY_OUT_B),
                             h, d, st_out ? stats : nullptr, D.B, D.T, d, stream));
      have_stats = st_out;
    } else {
      HIG_TRY(stylize(l, D.nsty - 1, HIG_L_FFN_STY_NORM_W, HIG_L_FFN_STY_NORM_B, HIG_L_FFN_STY_OUT_W, HIG_L_FFN_STY_OUT_B));
    }
  }
  // K6: out = Linear(d, F)(h_L), fp32 (the DDPM update consumes it)
  HIG_TRY(hig_gemm16_launch(G16(h, d, P16(params16, HIG_P_OUT_W), d, out, D.F, M, D.F, d)
                                .epi(HIG_EPI_BIAS, P(params, HIG_P_OUT_B)).out32().g, st));
  if (D.two)  // init-pose rows go through out2 instead (interaction_transformer.py:613-614)
    HIG_TRY(hig_gemm16_launch(G16(h, (int64_t)D.T * d, P16(params16, HIG_P_OUT2_W), d, out, (int64_t)D.T * D.F, D.B, D.F, d)
                                  .epi(HIG_EPI_BIAS, P(params, HIG_P_OUT2_B)).out32().g, st));
  return HIG_OK;
  };
  const int rc = frame_rows();
  return rc == HIG_OK ? rc : join_side(rc);
#undef HIG_TRY_SIDE
}

namespace {

// Weight gradients next to the data-gradient chain.  dW = dC^T . act and d(input) = dC . W only share dC, so the
// backward forks every weight-gradient GEMM onto a second (library-owned, per host thread and device) stream and joins
// it before returning: the chip is ~13 % idle on one in-order stream (kernel tails, ramp-up / drain between dependent
// launches, HBM idle under GEMMs; tools/concurrency_probe.py) and this is the one independent half of the work.
// Protocol (all through events, so it is captured into a hipGraph like any fork / join):
//   side waits for `ready` (recorded on the caller's stream where the weight gradient is requested: dC exists);
//   weight gradient k runs on the side stream (in order: they share the split-R slabs) and records done[k % 4];
//   the caller's stream waits for done[k-2] before it goes past request k -- every buffer a weight gradient reads
//   (dh ping/pong, t1, t2, tff, dqkv, dkv) is next overwritten at least two requests later (walk of the layer loop
//   in DESIGN.md section 4), and the saved forward activations are never written during backward;
//   join = the caller's stream waits for the last done event.
// HIG_BWD_OVERLAP=0 keeps everything on the caller's stream.
struct WgradFork {
  SideStream* side;
  hipStream_t main;
  int k = 0;
  WgradFork(SideStream* s, hipStream_t m) : side(s), main(m) {}
  hipStream_t stream() const { return side ? side->s2 : main; }
  int begin() {
    if (!side) return HIG_OK;
    if (k >= 2 && hipStreamWaitEvent(main, side->done[(k - 2) & 3], 0) != hipSuccess)
      return hig_set_error(HIG_EHIP, "hipStreamWaitEvent failed");
    if (hipEventRecord(side->ready, main) != hipSuccess || hipStreamWaitEvent(side->s2, side->ready, 0) != hipSuccess)
      return hig_set_error(HIG_EHIP, "weight-gradient fork failed");
    return HIG_OK;
  }
  int end() {
    if (!side) return HIG_OK;
    if (hipEventRecord(side->done[k & 3], side->s2) != hipSuccess) return hig_set_error(HIG_EHIP, "hipEventRecord failed");
    ++k;
    return HIG_OK;
  }
  // `other` waits for everything requested so far (the side stream is in order: the latest event)
  int wait_on(hipStream_t other) const {
    if (!side || k == 0) return HIG_OK;
    if (hipStreamWaitEvent(other, side->done[(k - 1) & 3], 0) != hipSuccess)
      return hig_set_error(HIG_EHIP, "weight-gradient join failed");
    return HIG_OK;
  }
  int join() {
    HIG_TRY(wait_on(main));
    k = 0;
    return HIG_OK;
  }
};

}  // namespace

// The library's only owned resources: the weight-gradient streams / events of the calling thread.
extern "C" int hig_shutdown(void) {
  SideStream* tab = side_stream_table();
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  int rc = HIG_OK;
  for (int dev = 0; dev < kMaxDev; ++dev) {
    SideStream& s = tab[dev];
    if (!s.ok) { s.failed = false; continue; }
    if (hipSetDevice(dev) != hipSuccess) { rc = hig_set_error(HIG_EHIP, "hig_shutdown: hipSetDevice(%d) failed", dev); continue; }
    if (hipStreamSynchronize(s.s2) != hipSuccess) rc = hig_set_error(HIG_EHIP, "hig_shutdown: side stream of device %d is in error", dev);
    for (int i = 0; i < 4; ++i) if (s.done[i]) (void)hipEventDestroy(s.done[i]);
    if (s.s3 && hipStreamSynchronize(s.s3) != hipSuccess) rc = hig_set_error(HIG_EHIP, "hig_shutdown: text stream of device %d is in error", dev);
    for (int i = 0; i < kMaxTextLayers; ++i) if (s.text_done[i]) (void)hipEventDestroy(s.text_done[i]);
    if (s.s3) (void)hipStreamDestroy(s.s3);
    if (s.ready) (void)hipEventDestroy(s.ready);
    if (s.s2) (void)hipStreamDestroy(s.s2);
    s = SideStream();
  }
  if (layer_event()) {
    (void)hipEventDestroy(layer_event());
    layer_event() = nullptr;
  }
  if (have_cur) (void)hipSetDevice(cur);
  return rc;
}

extern "C" int hig_denoiser_bwd(const hig_dims* dims, const void* const* params, const float* x,
                                const int64_t* t, const int64_t* length, const float* xf_out,
                                const void* textctx, const void* workspace, const float* dout,
                                void* const* grads, float* dx, float* dxf_proj, float* dxf_out,
                                void* bwd_workspace, hig_stream_t stream) {
  return hig_denoiser_bwd_hooked(dims, params, x, t, length, xf_out, textctx, workspace, dout, grads, dx, dxf_proj, dxf_out,
                                 bwd_workspace, stream, nullptr, nullptr, nullptr);
}

extern "C" int hig_denoiser_bwd_hooked(const hig_dims* dims, const void* const* params, const float* x,
                                       const int64_t* t, const int64_t* length, const float* xf_out,
                                       const void* textctx, const void* workspace, const float* dout,
                                       void* const* grads, float* dx, float* dxf_proj, float* dxf_out,
                                       void* bwd_workspace, hig_stream_t stream, hig_layer_hook hook, void* hook_user,
                                       hig_stream_t comm_stream) {
  (void)t;
  Dims D;
  HIG_TRY(check_dims(dims, D));
  HIG_REQUIRE(params && x && xf_out && textctx && workspace && dout && grads && dxf_proj && dxf_out && bwd_workspace,
              "hig_denoiser_bwd: null argument");
  const FwdLayout w = fwd_layout(D, 1);
  const TextLayout tl = text_layout(D, 1);
  const BwdLayout bw = bwd_layout(D);
  const float* ws = static_cast<const float*>(workspace);
  const float* tc = static_cast<const float*>(textctx);
  float* b = static_cast<float*>(bwd_workspace);
  hipStream_t st = hig_stream(stream);
  const int d = D.d, E = D.E, ff = D.ff, F = D.F, Lt = D.Lt;
  const int64_t M = D.M, Mt = D.Mt;
  const int64_t ss_ld = (int64_t)D.nsty * D.L * 2 * d;
  const int Bp = D.B / 2;
  float* slabs = b + bw.slabs;
  float* colp = b + bw.colpart;
  float* lnp = b + bw.lnpart;
  float* dss = b + bw.dss;

  // data-gradient GEMMs (caller's stream) may split their tail; the weight gradients on the side stream never do
  // (reduce-slow X operand or split-R: excluded by the rule in gemm.hip), so one scratch serves the whole backward
  HIG_TRY(hig_zero_async(b + bw.gtail, HIG_GEMM_TAIL_CNT_BYTES, st));
  const TailScratchScope tail_scope(b + bw.gtail);

  // Eager launches: the weight gradients go to the second stream.  Under stream capture they stay on the caller's (unless
  // HIG_BWD_OVERLAP=1): the replayed graph did not turn the fork into overlap -- config 2, captured fp32 step 21.6-21.7 ms forked
  // against 21.2 on one stream, while eager launches gain a millisecond from it (20.4 against 21.3).
  const int fork_env = bwd_overlap_env();
  hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap_status) != hipSuccess) cap_status = hipStreamCaptureStatusNone;
  const bool want_fork = fork_env == 1 || (fork_env != 0 && cap_status == hipStreamCaptureStatusNone);
  WgradFork fork(want_fork ? side_stream_for_current_device(st) : nullptr, st);
  hig_stream_t wstream = reinterpret_cast<hig_stream_t>(fork.stream());
  auto wgrad_on = [&](G gd) -> int {  // X, Y both reduce-slow; split over the reduce rows
    const int s = wgrad_splits(gd.g.I, gd.g.J, gd.g.R, bw.slab_floats, gd.g.prec);
    return hig_gemm_launch(gd.g, s, slabs, fork.stream());
  };
  auto wgrad = [&](G gd) -> int {
    HIG_TRY(fork.begin());
    HIG_TRY(wgrad_on(gd));
    return fork.end();
  };
  // dW[n][k] = sum_m dC[m][n] * act[m][k]  (act optionally LayerNorm'ed on the fly).
  // exact-fp32 mode: both operands read reduce-slow straight from their row-major buffers.
  // bf16 product modes: both are transposed first (LN fused into the transpose) so the GEMM gets
  // reduce-contiguous operands -- the layout the bf16 MFMA fragments need.
  // `dbias` (optional) = column sums of dC, the bias gradient of the same Linear: in the exact-fp32 path the wgrad
  // GEMM produces it while it streams dC (its reduce-slow X operand); otherwise a separate column-sum pass.
  auto wgrad_act = [&](const float* dC, int n_out, const float* act, int k_in, float* out, int64_t rows,
                       const float* stats, const float* gamma, const float* beta, float* dbias = nullptr) -> int {
    HIG_TRY(fork.begin());   // everything below (transposes, column sums, the GEMM) goes on the weight-gradient stream
    if (D.prec != HIG_PREC_F32 && rows % 32 == 0) {
      float* ta = b + bw.tA;
      float* tb = b + bw.tB;
      if (dbias) HIG_TRY(hig_colsum(dC, n_out, rows, n_out, dbias, b + bw.colpart_w, wstream));
      HIG_TRY(hig_transpose(dC, n_out, (int)rows, n_out, ta, rows, nullptr, nullptr, nullptr, wstream));
      HIG_TRY(hig_transpose(act, k_in, (int)rows, k_in, tb, rows, stats, gamma, beta, wstream));
      HIG_TRY(wgrad_on(G(ta, rows, 0, tb, rows, 0, out, k_in, n_out, k_in, rows).prec(D.prec)));
      return fork.end();
    }
    G gd(dC, n_out, 1, act, k_in, 1, out, k_in, n_out, k_in, rows);
    if (stats) gd.ln(1, stats, gamma, beta);
    if (dbias) {
      if (n_out % 4 == 0) gd.xsum(dbias);
      else HIG_TRY(hig_colsum(dC, n_out, rows, n_out, dbias, b + bw.colpart_w, wstream));
    }
    HIG_TRY(wgrad_on(gd));
    return fork.end();
  };
  // W (out, in) row-major -> W^T (in, out): the data-gradient GEMM dA = dC . W then reads both
  // operands reduce-contiguous (fast tile fetch, b128 LDS fragments, bf16 modes available)
  float* wT = b + bw.wT;
  const int64_t o_sty3 = 0, o_w2t = o_sty3 + (int64_t)d * d, o_w1t = o_w2t + (int64_t)d * ff,
                o_sty2 = o_w1t + (int64_t)d * ff, o_caq = o_sty2 + (int64_t)d * d, o_kv = o_caq + (int64_t)d * d,
                o_sty1 = o_kv + (int64_t)2 * d * Lt, o_qkv = o_sty1 + (int64_t)d * d,
                o_isty = o_qkv + (int64_t)3 * d * d, o_iqkv = o_isty + (int64_t)d * d;
  constexpr int TRB_N = 12;
  auto colsum = [&](const float* src, int64_t ld, int64_t rows, int n, float* dst) -> int {
    return hig_colsum(src, ld, rows, n, dst, colp, stream);
  };
  // Backward of one stylization block: h_out = h_in + Lin_out(silu(LN(y)*(1+scale)+shift)).
  // `dh` is d(h_out); produces dy into `dy_out`, parameter grads, and dss columns of block s.
  auto sty_bwd = [&](int l, int s, const float* dh, const float* y, const float* a_saved, const float* stats,
                     int norm_w, int norm_b, int out_w, int out_b, int64_t wt_off, float* dy_out) -> int {
    const float* ssl = ws + w.ss + (int64_t)s * 2 * d;
    HIG_TRY(wgrad_act(dh, d, a_saved, d, GL(grads, l, out_w), M, nullptr, nullptr, nullptr, GL(grads, l, out_b)));
    HIG_TRY(hig_gemm_launch(G(dh, d, 0, wT + wt_off, d, 0, b + bw.t1, d, M, d, d).prec(D.prec).g, 1, nullptr, st));
    return hig_ln_bwd(b + bw.t1, d, y, d, stats, PL(params, l, norm_w), PL(params, l, norm_b), ssl, ss_ld, d, 1,
                      nullptr, 0, dy_out, d, M, d, D.T, GL(grads, l, norm_w), GL(grads, l, norm_b),
                      dss + (int64_t)s * 2 * d, ss_ld, lnp, stream);
  };

  // ---- output projection ---------------------------------------------------------------
  const float* hL = ws + w.layer0 + w.lstride * (D.L - 1) + w.h3;
  float* dh = b + bw.dhA;
  float* dh_alt = b + bw.dhB;
  const float* dout_m = dout;  // rows that went through `out`
  if (D.two) {
    // init-pose rows went through out2 (:613): separate them from the `out` adjoint
    float* dm = b + bw.doutm;
    HIG_TRY(hig_copy_async(dm, dout, (size_t)M * F * 4, st));
    hipLaunchKernelGGL(tok0_kernel, dim3((D.B * F + 255) / 256), dim3(256), 0, st, dm, (int64_t)D.T * F, D.B, F,
                       (float*)nullptr, 1);
    HIG_CHECK_LAUNCH();
    dout_m = dm;
    HIG_TRY(colsum(dout, (int64_t)D.T * F, D.B, F, GP(grads, HIG_P_OUT2_B)));
    HIG_TRY(wgrad(G(dout, (int64_t)D.T * F, 1, hL, (int64_t)D.T * d, 1, GP(grads, HIG_P_OUT2_W), d, F, d, D.B)));
  }
  HIG_TRY(colsum(dout_m, F, M, F, GP(grads, HIG_P_OUT_B)));
  HIG_TRY(wgrad(G(dout_m, F, 1, hL, d, 1, GP(grads, HIG_P_OUT_W), d, F, d, M)));
  HIG_TRY(hig_gemm_launch(G(dout_m, F, 0, P(params, HIG_P_OUT_W), d, 1, dh, d, M, d, F).g, 1, nullptr, st));
  if (D.two)  // the out-GEMM left exact zeros in the init-pose rows of dh; fill them from out2
    HIG_TRY(hig_gemm_launch(G(dout, (int64_t)D.T * F, 0, P(params, HIG_P_OUT2_W), d, 1, dh, (int64_t)D.T * d, D.B, d, F).g,
                            1, nullptr, st));

  for (int l = D.L - 1; l >= 0; --l) {
    const float* lb = ws + w.layer0 + w.lstride * l;
    const float* hin = l == 0 ? ws + w.h0 : ws + w.layer0 + w.lstride * (l - 1) + w.h3;
    {  // all W -> W^T copies of this layer in one launch
      const float* srcs[TRB_N];
      float* dsts[TRB_N];
      int32_t rws[TRB_N], cls[TRB_N];
      int n = 0;
      auto add = [&](int idx, int out_f, int in_f, int64_t off) {
        srcs[n] = PL(params, l, idx); dsts[n] = wT + off; rws[n] = out_f; cls[n] = in_f; ++n;
      };
      add(HIG_L_FFN_STY_OUT_W, d, d, o_sty3);
      add(HIG_L_FFN_W2, d, ff, o_w2t);
      add(HIG_L_FFN_W1, ff, d, o_w1t);
      add(HIG_L_CA_STY_OUT_W, d, d, o_sty2);
      add(HIG_L_CA_Q_W, d, d, o_caq);
      add(HIG_L_CA_KV_W, 2 * d, Lt, o_kv);
      add(HIG_L_SA_STY_OUT_W, d, d, o_sty1);
      add(HIG_L_SA_QKV_W, 3 * d, d, o_qkv);
      if (D.two == 1) {
        add(HIG_L_INT_STY_OUT_W, d, d, o_isty);
        add(HIG_L_INT_QKV_W, 3 * d, d, o_iqkv);
      }
      HIG_TRY(hig_transpose_batch(n, srcs, dsts, rws, cls, stream));
    }
    const float* hffn = D.two == 1 ? lb + w.h2b : lb + w.h2;
    // ---- FFN --------------------------------------------------------------------------
    HIG_TRY(sty_bwd(l, D.nsty * l + D.nsty - 1, dh, lb + w.y3, lb + w.a3, lb + w.st5, HIG_L_FFN_STY_NORM_W, HIG_L_FFN_STY_NORM_B,
                    HIG_L_FFN_STY_OUT_W, HIG_L_FFN_STY_OUT_B, o_sty3, b + bw.t2));
    const float* dy3 = b + bw.t2;
    HIG_TRY(wgrad_act(dy3, d, lb + w.f1, ff, GL(grads, l, HIG_L_FFN_W2), M, nullptr, nullptr, nullptr,
                      GL(grads, l, HIG_L_FFN_B2)));
    HIG_TRY(hig_gemm_launch(G(dy3, d, 0, wT + o_w2t, d, 0, b + bw.tff, ff, M, ff, d).prec(D.prec)
                                .epi(HIG_EPI_DGELU).aux(const_cast<float*>(lb + w.z1), ff).g, 1, nullptr, st));
    const float* dz1 = b + bw.tff;
    HIG_TRY(wgrad_act(dz1, ff, hffn, d, GL(grads, l, HIG_L_FFN_W1), M, nullptr, nullptr, nullptr,
                      GL(grads, l, HIG_L_FFN_B1)));
    HIG_TRY(hig_gemm_launch(G(dz1, ff, 0, wT + o_w1t, ff, 0, dh_alt, d, M, d, ff).prec(D.prec)
                                .epi(HIG_EPI_RES).res(dh, d).g, 1, nullptr, st));
    { float* tmp = dh; dh = dh_alt; dh_alt = tmp; }  // dh = d(h2), or d(h2b) in the interaction model
    if (D.two == 1) {
      // ---- person <-> person cross attention ------------------------------------------------
      HIG_TRY(sty_bwd(l, D.nsty * l + 2, dh, lb + w.y4, lb + w.a4, lb + w.st7, HIG_L_INT_STY_NORM_W, HIG_L_INT_STY_NORM_B,
                      HIG_L_INT_STY_OUT_W, HIG_L_INT_STY_OUT_B, o_isty, b + bw.t2));
      const int64_t* len_partner = reinterpret_cast<const int64_t*>(ws + w.lenp);
      const int64_t halfA = (int64_t)Bp * D.H * D.hd * D.hd, halfM = (int64_t)Bp * D.T;
      float* dqkv = b + bw.dqkv;
      // consumer sample s read the context of producer (s + B/2) % B: route d(A) back the same way
      HIG_TRY(hig_linattn_apply_bwd(b + bw.t2, d, lb + w.iqkv, 3 * d, lb + w.Ai + halfA, dqkv, 3 * d, b + bw.dA + halfA,
                                    Bp, D.T, D.H, D.hd, b + bw.attn, stream));
      HIG_TRY(hig_linattn_apply_bwd(b + bw.t2 + halfM * d, d, lb + w.iqkv + halfM * 3 * d, 3 * d, lb + w.Ai,
                                    dqkv + halfM * 3 * d, 3 * d, b + bw.dA, Bp, D.T, D.H, D.hd, b + bw.attn, stream));
      HIG_TRY(hig_linattn_ctx_bwd(b + bw.dA, lb + w.Ai, lb + w.iqkv + d, lb + w.iqkv + 2 * d, 3 * d, lb + w.ksti, len_partner,
                                  dqkv + d, dqkv + 2 * d, 3 * d, D.B, D.T, D.H, D.hd, b + bw.attn, stream));
      HIG_TRY(wgrad_act(dqkv, 3 * d, lb + w.xn3, d, GL(grads, l, HIG_L_INT_QKV_W), M, nullptr, nullptr, nullptr,
                        GL(grads, l, HIG_L_INT_QKV_B)));
      HIG_TRY(hig_gemm_launch(G(dqkv, 3 * d, 0, wT + o_iqkv, 3 * d, 0, b + bw.t2, d, M, d, 3 * d).prec(D.prec).g, 1,
                              nullptr, st));
      HIG_TRY(hig_ln_bwd(b + bw.t2, d, lb + w.h2, d, lb + w.st6, PL(params, l, HIG_L_INT_NORM_W),
                         PL(params, l, HIG_L_INT_NORM_B), nullptr, 0, 0, 0, dh, d, dh_alt, d, M, d, D.T,
                         GL(grads, l, HIG_L_INT_NORM_W), GL(grads, l, HIG_L_INT_NORM_B), nullptr, 0, lnp, stream));
      { float* tmp = dh; dh = dh_alt; dh_alt = tmp; }  // dh = d(h2)
    }
    // ---- cross attention ---------------------------------------------------------------
    HIG_TRY(sty_bwd(l, D.nsty * l + 1, dh, lb + w.y2, lb + w.a2, lb + w.st4, HIG_L_CA_STY_NORM_W, HIG_L_CA_STY_NORM_B,
                    HIG_L_CA_STY_OUT_W, HIG_L_CA_STY_OUT_B, o_sty2, b + bw.t2));
    const float* Ac = tc + tl.layer0 + tl.lstride * l + tl.Ac;
    const float* kstc = tc + tl.layer0 + tl.lstride * l + tl.kstc;
    const float* kv = tc + tl.kv + tl.kv_stride * l;
    if (D.full)
      HIG_TRY(hig_fullattn_bwd(b + bw.t2, d, lb + w.y2, d, lb + w.qc, d, kv, kv + d, 2 * d, D.B, D.T, D.N, D.H, D.hd,
                               nullptr, lb + w.lse2, b + bw.delta, b + bw.t1, d, b + bw.dkv, b + bw.dkv + d, 2 * d,
                               stream));
    else
      HIG_TRY(hig_linattn_apply_bwd(b + bw.t2, d, lb + w.qc, d, Ac, b + bw.t1, d, b + bw.dA, D.B, D.T, D.H, D.hd,
                                    b + bw.attn, stream));
    const float* dqc = b + bw.t1;
    HIG_TRY(wgrad_act(dqc, d, lb + w.xn2, d, GL(grads, l, HIG_L_CA_Q_W), M, nullptr, nullptr, nullptr,
                      GL(grads, l, HIG_L_CA_Q_B)));
    HIG_TRY(hig_gemm_launch(G(dqc, d, 0, wT + o_caq, d, 0, b + bw.t2, d, M, d, d).prec(D.prec).g, 1, nullptr, st));
    HIG_TRY(hig_ln_bwd(b + bw.t2, d, lb + w.h1, d, lb + w.st3, PL(params, l, HIG_L_CA_NORM_W),
                       PL(params, l, HIG_L_CA_NORM_B), nullptr, 0, 0, 0, dh, d, dh_alt, d, M, d, D.T,
                       GL(grads, l, HIG_L_CA_NORM_W), GL(grads, l, HIG_L_CA_NORM_B), nullptr, 0, lnp, stream));
    { float* tmp = dh; dh = dh_alt; dh_alt = tmp; }  // dh = d(h1)
    // text side of this layer: d(A_c) -> d(key,value) -> text_norm -> d(xf_out)
    if (!D.full)
      HIG_TRY(hig_linattn_ctx_bwd(b + bw.dA, Ac, kv, kv + d, 2 * d, kstc, nullptr, b + bw.dkv, b + bw.dkv + d, 2 * d, D.B,
                                  D.N, D.H, D.hd, b + bw.attn, stream));
    HIG_TRY(wgrad_act(b + bw.dkv, 2 * d, xf_out, Lt, GL(grads, l, HIG_L_CA_KV_W), Mt, tc + tl.stt,
                      PL(params, l, HIG_L_CA_TNORM_W), PL(params, l, HIG_L_CA_TNORM_B), GL(grads, l, HIG_L_CA_KV_B)));
    HIG_TRY(hig_gemm_launch(G(b + bw.dkv, 2 * d, 0, wT + o_kv, 2 * d, 0, b + bw.dxfn, Lt, Mt, Lt, 2 * d).prec(D.prec).g,
                            1, nullptr, st));
    HIG_TRY(hig_ln_bwd(b + bw.dxfn, Lt, xf_out, Lt, tc + tl.stt, PL(params, l, HIG_L_CA_TNORM_W),
                       PL(params, l, HIG_L_CA_TNORM_B), nullptr, 0, 0, 0, l == D.L - 1 ? nullptr : dxf_out, Lt,
                       dxf_out, Lt, Mt, Lt, D.N, GL(grads, l, HIG_L_CA_TNORM_W), GL(grads, l, HIG_L_CA_TNORM_B),
                       nullptr, 0, lnp, stream));
    // ---- self attention ----------------------------------------------------------------
    HIG_TRY(sty_bwd(l, D.nsty * l, dh, lb + w.y1, lb + w.a1, lb + w.st2, HIG_L_SA_STY_NORM_W, HIG_L_SA_STY_NORM_B,
                    HIG_L_SA_STY_OUT_W, HIG_L_SA_STY_OUT_B, o_sty1, b + bw.t2));
    float* dqkv = b + bw.dqkv;
    if (D.full) {
      HIG_TRY(hig_fullattn_bwd(b + bw.t2, d, lb + w.y1, d, lb + w.qkv, 3 * d, lb + w.qkv + d, lb + w.qkv + 2 * d, 3 * d,
                               D.B, D.T, D.T, D.H, D.hd, length, lb + w.lse1, b + bw.delta, dqkv, 3 * d, dqkv + d,
                               dqkv + 2 * d, 3 * d, stream));
    } else {
      HIG_TRY(hig_linattn_apply_bwd(b + bw.t2, d, lb + w.qkv, 3 * d, lb + w.A1, dqkv, 3 * d, b + bw.dA, D.B, D.T, D.H,
                                    D.hd, b + bw.attn, stream));
      HIG_TRY(hig_linattn_ctx_bwd(b + bw.dA, lb + w.A1, lb + w.qkv + d, lb + w.qkv + 2 * d, 3 * d, lb + w.kst1, length, dqkv + d,
                                  dqkv + 2 * d, 3 * d, D.B, D.T, D.H, D.hd, b + bw.attn, stream));
    }
    HIG_TRY(wgrad_act(dqkv, 3 * d, lb + w.xn1, d, GL(grads, l, HIG_L_SA_QKV_W), M, nullptr, nullptr, nullptr,
                      GL(grads, l, HIG_L_SA_QKV_B)));
    HIG_TRY(hig_gemm_launch(G(dqkv, 3 * d, 0, wT + o_qkv, 3 * d, 0, b + bw.t2, d, M, d, 3 * d).prec(D.prec).g, 1,
                            nullptr, st));
    HIG_TRY(hig_ln_bwd(b + bw.t2, d, hin, d, lb + w.st1, PL(params, l, HIG_L_SA_NORM_W), PL(params, l, HIG_L_SA_NORM_B),
                       nullptr, 0, 0, 0, dh, d, dh_alt, d, M, d, D.T, GL(grads, l, HIG_L_SA_NORM_W),
                       GL(grads, l, HIG_L_SA_NORM_B), nullptr, 0, lnp, stream));
    { float* tmp = dh; dh = dh_alt; dh_alt = tmp; }  // dh = d(h_in of this layer)
    if (hook) {
      // Every parameter gradient of layer l exists once this layer's stylization `emb_layers.1.weight` rows are
      // done too (they only need this layer's columns of dss): d(W_emb)[l] = dss[:, l]^T . silu(emb), one more request on
      // the weight-gradient stream.  Then `comm_stream` is made to wait for both streams and the caller is told: it
      // can start exchanging layer l's gradients while layers l-1 ... 0 are still in backward.
      const int64_t rows_l = (int64_t)D.nsty * 2 * d;
      HIG_TRY(wgrad(G(dss + (int64_t)l * rows_l, ss_ld, 1, ws + w.emb, E, 1, GP(grads, HIG_P_STY_EMB_W) + (int64_t)l * rows_l * E, E,
                      rows_l, E, D.B).silu(1)));
      if (comm_stream) {
        hipStream_t cs = hig_stream(comm_stream);
        hipEvent_t& ev_layer = layer_event();
        if (!ev_layer && hipEventCreateWithFlags(&ev_layer, hipEventDisableTiming) != hipSuccess)
          return hig_set_error(HIG_EHIP, "hipEventCreate failed");
        if (hipEventRecord(ev_layer, st) != hipSuccess || hipStreamWaitEvent(cs, ev_layer, 0) != hipSuccess)
          return hig_set_error(HIG_EHIP, "layer hook: event on the caller's stream failed");
        HIG_TRY(fork.wait_on(cs));       // (and for everything on the weight-gradient stream)
      }
      hook(hook_user, l);
    }
  }

  // ---- joint_embed + sequence_embedding ------------------------------------------------
  if (D.two) {
    // init-pose rows came from joint_embed2 (no positional term): move them aside, leaving zeros so the
    // joint_embed / sequence_embedding adjoints below see only the motion rows
    hipLaunchKernelGGL(tok0_kernel, dim3((D.B * d + 255) / 256), dim3(256), 0, st, dh, (int64_t)D.T * d, D.B, d,
                       b + bw.tok0, 1);
    HIG_CHECK_LAUNCH();
    HIG_TRY(colsum(b + bw.tok0, d, D.B, d, GP(grads, HIG_P_JOINT2_B)));
    HIG_TRY(wgrad(G(b + bw.tok0, d, 1, x, (int64_t)D.T * F, 1, GP(grads, HIG_P_JOINT2_W), 4, d, 4, D.B)));
  }
  HIG_TRY(colsum(dh, d, M, d, GP(grads, HIG_P_JOINT_B)));
  HIG_TRY(wgrad(G(dh, d, 1, x, F, 1, GP(grads, HIG_P_JOINT_W), F, d, F, M)));
  // d(sequence_embedding)[t] = sum_b dh[b, t, :]  == column sum of dh viewed as (B, T*d)
  const int Tpos = D.two ? D.T - 1 : D.T;  // rows of sequence_embedding that were used
  if (D.two) {  // frame t used row t-1
    HIG_TRY(colsum(dh, (int64_t)D.T * d, D.B, D.T * d, b + bw.postmp));
    HIG_TRY(hig_copy_async(GP(grads, HIG_P_SEQ_EMB), b + bw.postmp + d, (size_t)Tpos * d * 4, st));
  } else {
    HIG_TRY(colsum(dh, (int64_t)D.T * d, D.B, D.T * d, GP(grads, HIG_P_SEQ_EMB)));
  }
  if (D.nf > Tpos)
    HIG_TRY(hig_zero_async(GP(grads, HIG_P_SEQ_EMB) + (int64_t)Tpos * d, (size_t)(D.nf - Tpos) * d * 4, st));
  if (dx) {
    HIG_TRY(hig_gemm_launch(G(dh, d, 0, P(params, HIG_P_JOINT_W), F, 1, dx, F, M, F, d).g, 1, nullptr, st));
    if (D.two)  // d(x[:, 0, :4]) through joint_embed2; the other features of the init-pose row are unused
      HIG_TRY(hig_gemm_launch(G(b + bw.tok0, d, 0, P(params, HIG_P_JOINT2_W), 4, 1, dx, (int64_t)D.T * F, D.B, 4, d).g, 1,
                              nullptr, st));
  }

  // ---- time / text embedding path ---------------------------------------------------------
  HIG_TRY(fork.join());   // every weight gradient is done (and the slabs are free) from here on
  const float* emb = ws + w.emb;
  HIG_TRY(colsum(dss, ss_ld, D.B, (int)ss_ld, GP(grads, HIG_P_STY_EMB_B)));
  if (!hook)   // (with a layer hook the rows of each layer were produced inside that layer's backward)
    HIG_TRY(hig_gemm_launch(G(dss, ss_ld, 1, emb, E, 1, GP(grads, HIG_P_STY_EMB_W), E, ss_ld, E, D.B).silu(1).g, 1, nullptr, st));
  {
    G gd(dss, ss_ld, 0, P(params, HIG_P_STY_EMB_W), E, 1, b + bw.dtmp, E, D.B, E, ss_ld);
    int s = (int)(ss_ld / 1024);
    if (s > 16) s = 16;
    HIG_TRY(hig_gemm_launch(gd.g, s < 1 ? 1 : s, slabs, st));
  }
  const int64_t nBE = (int64_t)D.B * E;
  const int eb = (int)((nBE + 255) / 256);
  hipLaunchKernelGGL(mul_dsilu_kernel, dim3(eb), dim3(256), 0, st, b + bw.dtmp, emb, nBE, b + bw.demb);
  HIG_CHECK_LAUNCH();
  HIG_TRY(hig_copy_async(dxf_proj, b + bw.demb, (size_t)nBE * 4, st));
  HIG_TRY(colsum(b + bw.demb, E, D.B, E, GP(grads, HIG_P_TE2_B)));
  HIG_TRY(hig_gemm_launch(G(b + bw.demb, E, 1, ws + w.te_h, E, 1, GP(grads, HIG_P_TE2_W), E, E, E, D.B).silu(1).g, 1, nullptr, st));
  HIG_TRY(hig_gemm_launch(G(b + bw.demb, E, 0, P(params, HIG_P_TE2_W), E, 1, b + bw.dtmp, E, D.B, E, E).g, 1, nullptr, st));
  hipLaunchKernelGGL(mul_dsilu_kernel, dim3(eb), dim3(256), 0, st, b + bw.dtmp, ws + w.te_h, nBE, b + bw.dte_h);
  HIG_CHECK_LAUNCH();
  HIG_TRY(colsum(b + bw.dte_h, E, D.B, E, GP(grads, HIG_P_TE0_B)));
  HIG_TRY(hig_gemm_launch(G(b + bw.dte_h, E, 1, ws + w.te, d, 1, GP(grads, HIG_P_TE0_W), d, E, d, D.B).g, 1, nullptr, st));
  return HIG_OK;
}


// ==========================================================================================================
// bf16-storage TRAINING step (include/hig.h: hig_denoiser_fwd_bf16_train / hig_denoiser_bwd_bf16).
// Forward = the launch sequence of hig_denoiser_fwd_bf16 without the inference-only fusions (no in-place residual stream, no
// LayerNorm fold, no fused stylization blocks) and with every layer's activations kept; backward = the adjoint of
// hig_denoiser_bwd over bf16 rows.  Byte layouts, 256-byte granules.
// ==========================================================================================================
namespace {

inline int64_t rup64(int64_t v) { return (v + 63) & ~(int64_t)63; }

struct Fwd16TLayout {
  int64_t te, te_h, emb, ss, few, few_bytes, h0, cscr, layer0, lstride;
  int64_t xn1, qkv, A1, At1, kst1, y1, a1, h1, xn2, qc, y2, a2, h2, z1, f1, y3, a3, h3;
  int64_t xn3, iqkv, Ai, Ati, ksti, y4, a4, h2b;   // two-person: person <-> person attention block (interaction_transformer.py:167-207)
  int64_t tok0, lenp;                               // two-person: init-pose rows (fp32, B x d), partner lengths (int64)
  int64_t total;
};
Fwd16TLayout fwd16t_layout(const Dims& D) {
  Fwd16TLayout w;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += alb(n); return r; };
  w.te = take((int64_t)D.B * D.d * 4);       // fp32: the per-sample embedding chain runs on the fp32 few-row kernels
  w.te_h = take((int64_t)D.B * D.E * 4);
  w.emb = take((int64_t)D.B * D.E * 4);
  w.ss = take((int64_t)D.B * D.nsty * D.L * 2 * D.d * 4);
  w.few_bytes = (int64_t)64 * 1024 * 1024;   // split-R scratch of hig_gemm_few_rows
  w.few = take(w.few_bytes);
  w.h0 = take(D.M * D.d * 2);
  w.cscr = take(hig_linattn_ctx_scratch_floats(D.B, D.T, D.H, D.hd) * 4);
  w.tok0 = take(D.two ? (int64_t)D.B * D.d * 4 : 0);
  w.lenp = take(D.two ? (int64_t)D.B * 8 : 0);
  w.layer0 = o;
  o = 0;
  w.xn1 = take(D.M * D.d * 2);
  w.qkv = take(D.M * 3 * D.d * 2);
  w.A1 = take((int64_t)D.B * D.H * D.hd * D.hd * 4);
  w.At1 = take((int64_t)D.B * D.H * D.hd * D.hd * 2);
  w.kst1 = take((int64_t)D.B * D.d * 2 * 4);
  w.y1 = take(D.M * D.d * 2);
  w.a1 = take(D.M * D.d * 2);
  w.h1 = take(D.M * D.d * 2);
  w.xn2 = take(D.M * D.d * 2);
  w.qc = take(D.M * D.d * 2);
  w.y2 = take(D.M * D.d * 2);
  w.a2 = take(D.M * D.d * 2);
  w.h2 = take(D.M * D.d * 2);
  w.z1 = take(D.M * D.ff * 2);
  w.f1 = take(D.M * D.ff * 2);
  w.y3 = take(D.M * D.d * 2);
  w.a3 = take(D.M * D.d * 2);
  w.h3 = take(D.M * D.d * 2);
  w.xn3 = w.iqkv = w.Ai = w.Ati = w.ksti = w.y4 = w.a4 = w.h2b = 0;
  if (D.two == 1) {
    w.xn3 = take(D.M * D.d * 2);
    w.iqkv = take(D.M * 3 * D.d * 2);
    w.Ai = take((int64_t)D.B * D.H * D.hd * D.hd * 4);
    w.Ati = take((int64_t)D.B * D.H * D.hd * D.hd * 2);
    w.ksti = take((int64_t)D.B * D.d * 2 * 4);
    w.y4 = take(D.M * D.d * 2);
    w.a4 = take(D.M * D.d * 2);
    w.h2b = take(D.M * D.d * 2);
  }
  w.lstride = o;
  w.total = w.layer0 + w.lstride * D.L;
  return w;
}

// text side, training: per layer the normalised text rows (operand of the key/value weight gradient), key/value (operands of
// the context backward), the context matrices and their column-softmax statistics
struct Text16TLayout {
  int64_t cscr, layer0, lstride, xfn, kv, Ac, Atc, kstc, total;
};
Text16TLayout text16t_layout(const Dims& D) {
  Text16TLayout t;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += alb(n); return r; };
  t.cscr = take(hig_linattn_ctx_scratch_floats(D.B, D.N, D.H, D.hd) * 4);
  t.layer0 = o;
  o = 0;
  t.xfn = take(D.Mt * D.Lt * 2);
  t.kv = take(D.Mt * 2 * D.d * 2);
  t.Ac = take((int64_t)D.B * D.H * D.hd * D.hd * 4);
  t.Atc = take((int64_t)D.B * D.H * D.hd * D.hd * 2);
  t.kstc = take((int64_t)D.B * D.d * 2 * 4);
  t.lstride = o;
  t.total = t.layer0 + t.lstride * D.L;
  return t;
}

struct Bwd16Layout {
  int64_t dhA, dhB, t1, t2, tff, dqkv, dA, dkv, dxfn, dss, demb, dtmp, dte_h, slabs, slab_floats, colpart, colpart_w, lnpart, lnpart_slot,
      wT, tA, tB, attn, f32a, f32b, Mp, Mtp, total;
  int64_t tok0, hl0, postmp;   // two-person (fp32): init-pose rows of a gradient (B x max(d, F)), of h_L (B x d); d(sequence_embedding) before its shift (T x d)
};
Bwd16Layout bwd16_layout(const Dims& D) {
  Bwd16Layout w;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += alb(n); return r; };
  w.Mp = rup64(D.M);      // reduce extents of the weight-gradient GEMMs, padded to whole 64-deep k-tiles (zeros behind M)
  w.Mtp = rup64(D.Mt);
  w.dhA = take(D.M * D.d * 2);
  w.dhB = take(D.M * D.d * 2);
  w.t1 = take(D.M * D.d * 2);
  w.t2 = take(D.M * D.d * 2);
  w.tff = take(D.M * D.ff * 2);
  w.dqkv = take(D.M * 3 * D.d * 2);
  w.dA = take((int64_t)D.B * D.H * D.hd * D.hd * 4);
  w.dkv = take(D.Mt * 2 * D.d * 2);
  w.dxfn = take(D.Mt * D.Lt * 2);
  w.dss = take((int64_t)D.B * D.nsty * D.L * 2 * D.d * 4);
  w.demb = take((int64_t)D.B * D.E * 4);
  w.dtmp = take((int64_t)D.B * D.E * 4);
  w.dte_h = take((int64_t)D.B * D.E * 4);
  int64_t biggest = 0;
  const int64_t outs[] = {(int64_t)3 * D.d * D.d, (int64_t)D.d * D.d, (int64_t)D.ff * D.d, (int64_t)2 * D.d * D.Lt,
                          (int64_t)D.F * D.d, (int64_t)D.B * D.E, (int64_t)D.E * D.E, (int64_t)D.E * D.d};
  for (int64_t v : outs) biggest = v > biggest ? v : biggest;
  w.slab_floats = biggest * 16 > (int64_t)1536 * 128 * 128 ? biggest * 16 : (int64_t)1536 * 128 * 128;
  w.slabs = take(w.slab_floats * 4);
  int64_t colp = 0;
  const int64_t uses[][2] = {{D.M, 3 * D.d}, {D.M, D.ff}, {D.M, D.F}, {D.Mt, 2 * D.d}, {D.B, D.E},
                             {D.B, (int64_t)D.nsty * D.L * 2 * D.d}, {D.B, (int64_t)D.T * D.d}};
  for (auto& u : uses) {
    const int64_t v = (int64_t)hig_colsum_chunks(u[0]) * u[1];
    colp = v > colp ? v : colp;
  }
  w.colpart = take(colp * 4);
  w.colpart_w = take(colp * 4);
  const int64_t lp = hig_ln_bwd_partial_floats(D.M, D.d, D.T), lpt = hig_ln_bwd_partial_floats(D.Mt, D.Lt, D.N);
  w.lnpart_slot = (lp > lpt ? lp : lpt) * 4;      // one partial table per pending LayerNorm backward of a layer (ln_flush)
  w.lnpart = take(w.lnpart_slot * HIG_LN_RB_MAX);
  w.wT = take(((int64_t)5 * D.d * D.d + (int64_t)3 * D.d * D.d + (int64_t)2 * D.d * D.ff + (int64_t)2 * D.d * D.Lt +
               (D.two == 1 ? (int64_t)4 * D.d * D.d : 0)) * 2);
  const int64_t mrows = w.Mp > w.Mtp ? w.Mp : w.Mtp;
  const int64_t wide = 3 * D.d > D.ff ? 3 * D.d : D.ff;
  w.tA = take(wide * mrows * 2);
  w.tB = take((int64_t)(D.ff > D.d ? (D.ff > D.Lt ? D.ff : D.Lt) : (D.d > D.Lt ? D.d : D.Lt)) * mrows * 2);
  const int64_t as1 = hig_linattn_bwd_scratch_floats(D.B, D.T, D.H, D.hd), as2 = hig_linattn_bwd_scratch_floats(D.B, D.N, D.H, D.hd);
  w.attn = take((as1 > as2 ? as1 : as2) * 4);
  w.f32a = take(D.M * D.d * 4);     // fp32 copies at the F-wide edges (input / output projection run on the fp32 kernels)
  w.f32b = take(D.M * D.d * 4);
  w.tok0 = take(D.two ? (int64_t)D.B * (D.d > D.F ? D.d : D.F) * 4 : 0);
  w.hl0 = take(D.two ? (int64_t)D.B * D.d * 4 : 0);
  w.postmp = take(D.two ? (int64_t)D.T * D.d * 4 : 0);
  w.total = o;
  return w;
}

int64_t fwd16t_total(const Dims& D) { return fwd16t_layout(D).total; }
int64_t text16t_total(const Dims& D) { return text16t_layout(D).total; }
int64_t bwd16_total(const Dims& D) { return bwd16_layout(D).total; }

int require_bf16_train(const Dims& D, const char* who) {
  if (!D.bf16) return hig_set_error(HIG_EINVAL, "%s: dims->storage must be HIG_STORE_BF16", who);
  if (bf16_train_unsupported(D, 1)) return HIG_EUNSUPPORTED;
  return HIG_OK;
}

}  // namespace

extern "C" int hig_text_context_bf16_train(const hig_dims* dims, const void* const* params, const void* const* params16,
                                           const float* xf_out, void* textctx, hig_stream_t stream) {
  Dims D;
  HIG_TRY(check_dims(dims, D));
  HIG_TRY(require_bf16_train(D, "hig_text_context_bf16_train"));
  HIG_REQUIRE(params && params16 && xf_out && textctx, "hig_text_context_bf16_train: null argument");
  const Text16TLayout tl = text16t_layout(D);
  char* base = static_cast<char*>(textctx);
  hipStream_t st = hig_stream(stream);
  for (int l = 0; l < D.L; ++l) {
    char* lb = base + tl.layer0 + tl.lstride * l;
    // text_norm of THIS layer, [key; value] projection, softmax over the N tokens, A_c = k^T v   (transformer.py:146-152)
    HIG_TRY(hig_ln_bf16(xf_out, 1, D.Lt, D.Mt, D.Lt, PL(params, l, HIG_L_CA_TNORM_W), PL(params, l, HIG_L_CA_TNORM_B), nullptr, 0, 0, 0,
                        lb + tl.xfn, D.Lt, stream));
    HIG_TRY(hig_gemm16_launch(G16(lb + tl.xfn, D.Lt, PL16(params16, l, HIG_L_CA_KV_W), D.Lt, lb + tl.kv, 2 * D.d, D.Mt, 2 * D.d, D.Lt)
                                  .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_CA_KV_B)).g, st));
    HIG_TRY(ctx16(D, lb + tl.kv, lb + tl.kv + (int64_t)D.d * 2, 2 * D.d, D.B, D.N, nullptr, reinterpret_cast<float*>(lb + tl.Ac),
                  reinterpret_cast<float*>(lb + tl.kstc), reinterpret_cast<float*>(base + tl.cscr), lb + tl.Atc, stream));
  }
  return HIG_OK;
}

extern "C" int hig_denoiser_fwd_bf16_train(const hig_dims* dims, const void* const* params, const void* const* params16,
                                           const float* x, const int64_t* t, const int64_t* length, const float* xf_proj,
                                           const void* textctx, float* out, void* workspace, hig_stream_t stream) {
  Dims D;
  HIG_TRY(check_dims(dims, D));
  HIG_TRY(require_bf16_train(D, "hig_denoiser_fwd_bf16_train"));
  HIG_REQUIRE(params && params16 && x && t && xf_proj && textctx && out && workspace, "hig_denoiser_fwd_bf16_train: null argument");
  const Fwd16TLayout w = fwd16t_layout(D);
  const Text16TLayout tl = text16t_layout(D);
  char* ws = static_cast<char*>(workspace);
  const char* tc = static_cast<const char*>(textctx);
  hipStream_t st = hig_stream(stream);
  const int d = D.d, E = D.E;
  const int64_t M = D.M;
  const int64_t ss_ld = (int64_t)D.nsty * D.L * 2 * d;
  float* te = reinterpret_cast<float*>(ws + w.te);
  float* te_h = reinterpret_cast<float*>(ws + w.te_h);
  float* emb = reinterpret_cast<float*>(ws + w.emb);
  float* ss = reinterpret_cast<float*>(ws + w.ss);
  float* few = reinterpret_cast<float*>(ws + w.few);
  const int64_t few_floats = w.few_bytes / 4;
  // K0: the per-sample embedding chain in fp32 on the master weights (B rows: weight-bandwidth bound; its backward needs the
  //     pre-activations te_h and emb in fp32)                                          (transformer.py:345-349,415,81-83)
  HIG_TRY(hig_timestep_embedding(t, D.B, d, te, stream));
  HIG_TRY(hig_gemm_few_rows(G(te, d, 0, P(params, HIG_P_TE0_W), d, 0, te_h, E, D.B, E, d).epi(HIG_EPI_BIAS, P(params, HIG_P_TE0_B)).g,
                            few, few_floats, st));
  HIG_TRY(hig_gemm_few_rows(G(te_h, E, 0, P(params, HIG_P_TE2_W), E, 0, emb, E, D.B, E, E).silu(0)
                                .epi(HIG_EPI_BIAS_RES, P(params, HIG_P_TE2_B)).res(xf_proj, E).g, few, few_floats, st));
  HIG_TRY(hig_gemm_few_rows(G(emb, E, 0, P(params, HIG_P_STY_EMB_W), E, 0, ss, ss_ld, D.B, ss_ld, E).silu(0)
                                .epi(HIG_EPI_BIAS, P(params, HIG_P_STY_EMB_B)).g, few, few_floats, st));
  // K1: h0 = joint_embed(x) + sequence_embedding[:T], rounded once into the bf16 residual stream
  if (d % 128 == 0 && D.F <= 512) {
    HIG_TRY(hig_joint_embed_bf16(x, M, D.F, P(params, HIG_P_JOINT_W), P(params, HIG_P_JOINT_B), P(params, HIG_P_SEQ_EMB), d, D.T,
                                 D.two ? 1 : 0, ws + w.h0, d, d, few, stream));   // two-person: frame t >= 1 gets sequence_embedding[t - 1]
  } else {
    float* h32 = few;
    HIG_REQUIRE(M * d * 4 <= w.few_bytes, "hig_denoiser_fwd_bf16_train: d %% 128 != 0 needs M d floats of scratch");
    G ge(x, D.F, 0, P(params, HIG_P_JOINT_W), D.F, 0, h32, d, M, d, D.F);
    ge.epi(HIG_EPI_BIAS_POS, P(params, HIG_P_JOINT_B)).pos(P(params, HIG_P_SEQ_EMB), d, D.T);
    ge.g.pos_shift = D.two ? 1 : 0;
    HIG_TRY(hig_gemm_launch(ge.g, 1, nullptr, st));
    HIG_TRY(hig_cast_bf16(h32, ws + w.h0, M * d, stream));
  }
  const int64_t* len_partner = nullptr;
  const int Bp = D.B / 2;
  if (D.two) {
    // token 0 is the init-pose row: joint_embed2 on its first 4 features, no positional term (interaction_transformer.py:596);
    // fp32 GEMM over the B rows, then into the bf16 residual stream.  The partner's lengths mask the person <-> person keys.
    float* tok0 = reinterpret_cast<float*>(ws + w.tok0);
    HIG_TRY(hig_gemm_launch(G(x, (int64_t)D.T * D.F, 0, P(params, HIG_P_JOINT2_W), 4, 0, tok0, d, D.B, d, 4)
                                .epi(HIG_EPI_BIAS, P(params, HIG_P_JOINT2_B)).g, 1, nullptr, st));
    hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3((unsigned)(((int64_t)D.B * d + 255) / 256)), dim3(256), 0, st, tok0, (int64_t)d,
                       reinterpret_cast<__bf16*>(ws + w.h0), (int64_t)D.T * d, D.B, d);
    HIG_CHECK_LAUNCH();
    int64_t* lp = reinterpret_cast<int64_t*>(ws + w.lenp);
    hipLaunchKernelGGL(swap_halves_i64_kernel, dim3((D.B + 255) / 256), dim3(256), 0, st, length, D.B, (int64_t)D.T, lp);
    HIG_CHECK_LAUNCH();
    len_partner = lp;
  }
  const void* hin = ws + w.h0;
  for (int l = 0; l < D.L; ++l) {
    char* lb = ws + w.layer0 + w.lstride * l;
    const float* ssl = ss + (int64_t)(D.nsty * l) * 2 * d;
    static const int fuse_env = getenv("HIG_FUSE_APPLY") ? atoi(getenv("HIG_FUSE_APPLY")) : 2;   // tuning knob (as the inference forward)
    const bool fuse_front = fuse_env == 2 && (D.hd == 64 || D.hd == 128) && (D.H == 4 || D.H == 8);
    // one stylization block: h_out = h_in + Lin_out( silu( LN(y) (1 + scale) + shift ) )     (transformer.py:81-86)
    auto sty_out = [&](const void* a, const void* h_in, void* h_out, int out_w, int out_b) -> int {
      return hig_gemm16_launch(G16(a, d, PL16(params16, l, out_w), d, h_out, d, M, d, d).epi(HIG_EPI_BIAS_RES, PL(params, l, out_b)).res16(h_in, d).g, st);
    };
    auto stylize = [&](int slot, const void* y, void* a, const void* h_in, void* h_out, int norm_w, int norm_b, int out_w, int out_b) -> int {
      HIG_TRY(hig_ln_bf16(y, 0, d, M, d, PL(params, l, norm_w), PL(params, l, norm_b), ssl + (int64_t)slot * 2 * d, ss_ld, d, D.T, a, d, stream));
      return sty_out(a, h_in, h_out, out_w, out_b);
    };
    // attention output -> stylization front: y = softmax(q) . A, LayerNorm, modulation, SiLU as ONE kernel that also writes y (the
    // backward needs it); Bq samples starting at sample b0 (the person <-> person block runs the two halves against swapped contexts)
    auto attn_front = [&](int slot, const void* q, int64_t ldq, const float* A32, const void* At16_, char* y, char* a, int norm_w, int norm_b,
                          int b0, int Bq) -> int {
      const int64_t r0 = (int64_t)b0 * D.T;
      const float* ssb = ssl + (int64_t)slot * 2 * d + (int64_t)b0 * ss_ld;
      if (fuse_front)
        return hig_linattn_apply_sty_mm16_y(static_cast<const char*>(q) + r0 * ldq * 2, ldq, At16_, PL(params, l, norm_w), PL(params, l, norm_b), ssb,
                                            ss_ld, d, a + r0 * d * 2, d, y + r0 * d * 2, d, Bq, D.T, D.H, D.hd, stream);
      HIG_TRY(hig_linattn_apply_bf16(static_cast<const char*>(q) + r0 * ldq * 2, ldq, A32, y + r0 * d * 2, d, Bq, D.T, D.H, D.hd, stream));
      return hig_ln_bf16(y + r0 * d * 2, 0, d, (int64_t)Bq * D.T, d, PL(params, l, norm_w), PL(params, l, norm_b), ssb, ss_ld, d, D.T, a + r0 * d * 2, d, stream);
    };
    // ---- self attention (transformer.py:101-119) ----
    HIG_TRY(hig_ln_bf16(hin, 0, d, M, d, PL(params, l, HIG_L_SA_NORM_W), PL(params, l, HIG_L_SA_NORM_B), nullptr, 0, 0, 0, lb + w.xn1, d, stream));
    HIG_TRY(hig_gemm16_launch(G16(lb + w.xn1, d, PL16(params16, l, HIG_L_SA_QKV_W), d, lb + w.qkv, 3 * d, M, 3 * d, d)
                                  .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_SA_QKV_B)).g, st));
    char* qkv = lb + w.qkv;
    HIG_TRY(ctx16(D, qkv + (int64_t)d * 2, qkv + (int64_t)2 * d * 2, 3 * d, D.B, D.T, length, reinterpret_cast<float*>(lb + w.A1),
                  reinterpret_cast<float*>(lb + w.kst1), reinterpret_cast<float*>(ws + w.cscr), lb + w.At1, stream));
    HIG_TRY(attn_front(0, qkv, 3 * d, reinterpret_cast<const float*>(lb + w.A1), lb + w.At1, lb + w.y1, lb + w.a1, HIG_L_SA_STY_NORM_W, HIG_L_SA_STY_NORM_B, 0, D.B));
    HIG_TRY(sty_out(lb + w.a1, hin, lb + w.h1, HIG_L_SA_STY_OUT_W, HIG_L_SA_STY_OUT_B));
    // ---- cross attention to the text context (transformer.py:135-155) ----
    HIG_TRY(hig_ln_bf16(lb + w.h1, 0, d, M, d, PL(params, l, HIG_L_CA_NORM_W), PL(params, l, HIG_L_CA_NORM_B), nullptr, 0, 0, 0, lb + w.xn2, d, stream));
    HIG_TRY(hig_gemm16_launch(G16(lb + w.xn2, d, PL16(params16, l, HIG_L_CA_Q_W), d, lb + w.qc, d, M, d, d)
                                  .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_CA_Q_B)).g, st));
    HIG_TRY(attn_front(1, lb + w.qc, d, reinterpret_cast<const float*>(tc + tl.layer0 + tl.lstride * l + tl.Ac), tc + tl.layer0 + tl.lstride * l + tl.Atc,
                       lb + w.y2, lb + w.a2, HIG_L_CA_STY_NORM_W, HIG_L_CA_STY_NORM_B, 0, D.B));
    HIG_TRY(sty_out(lb + w.a2, lb + w.h1, lb + w.h2, HIG_L_CA_STY_OUT_W, HIG_L_CA_STY_OUT_B));
    const void* hffn = lb + w.h2;
    if (D.two == 1) {
      // ---- person <-> person linear cross attention (interaction_transformer.py:181-205): queries from the own stream, key /
      // value from the partner's (same LayerNorm on both), key softmax masked with the consumer's length ----
      HIG_TRY(hig_ln_bf16(lb + w.h2, 0, d, M, d, PL(params, l, HIG_L_INT_NORM_W), PL(params, l, HIG_L_INT_NORM_B), nullptr, 0, 0, 0, lb + w.xn3, d, stream));
      HIG_TRY(hig_gemm16_launch(G16(lb + w.xn3, d, PL16(params16, l, HIG_L_INT_QKV_W), d, lb + w.iqkv, 3 * d, M, 3 * d, d)
                                    .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_INT_QKV_B)).g, st));
      char* iqkv = lb + w.iqkv;
      float* Ai = reinterpret_cast<float*>(lb + w.Ai);
      HIG_TRY(ctx16(D, iqkv + (int64_t)d * 2, iqkv + (int64_t)2 * d * 2, 3 * d, D.B, D.T, len_partner, Ai, reinterpret_cast<float*>(lb + w.ksti),
                    reinterpret_cast<float*>(ws + w.cscr), lb + w.Ati, stream));
      const int64_t halfA = (int64_t)Bp * D.H * D.hd * D.hd;
      // (person 1's queries against person 2's context and vice versa: the halves of A / At swapped)
      HIG_TRY(attn_front(2, iqkv, 3 * d, Ai + halfA, lb + w.Ati + halfA * 2, lb + w.y4, lb + w.a4, HIG_L_INT_STY_NORM_W, HIG_L_INT_STY_NORM_B, 0, Bp));
      HIG_TRY(attn_front(2, iqkv, 3 * d, Ai, lb + w.Ati, lb + w.y4, lb + w.a4, HIG_L_INT_STY_NORM_W, HIG_L_INT_STY_NORM_B, Bp, Bp));
      HIG_TRY(sty_out(lb + w.a4, lb + w.h2, lb + w.h2b, HIG_L_INT_STY_OUT_W, HIG_L_INT_STY_OUT_B));
      hffn = lb + w.h2b;
    }
    // ---- FFN (transformer.py:167-170): z = Lin1(h2) kept for gelu'(z), f = gelu(z) is linear2's operand ----
    // (one launch where the specialised-wave kernel serves the shape: it writes f and, as its second output, z -- both from the
    // fp32 pre-activation, f exactly as the inference forward computes it; else linear1 and a GELU pass over z)
    {
      G16 g1(hffn, d, PL16(params16, l, HIG_L_FFN_W1), d, lb + w.f1, D.ff, M, D.ff, d);
      g1.epi(HIG_EPI_BIAS_GELU, PL(params, l, HIG_L_FFN_B1));
      g1.g.aux = lb + w.z1; g1.g.ldaux = D.ff;
      const int rc = hig_gemm_wsp16_try(g1.g, st);
      if (rc < 0) return rc;
      if (rc == 1) {
        HIG_TRY(hig_gemm16_launch(G16(hffn, d, PL16(params16, l, HIG_L_FFN_W1), d, lb + w.z1, D.ff, M, D.ff, d)
                                      .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_FFN_B1)).g, st));
        HIG_TRY(hig_gelu_bf16(lb + w.z1, lb + w.f1, M * D.ff, stream));
      }
    }
    HIG_TRY(hig_gemm16_launch(G16(lb + w.f1, D.ff, PL16(params16, l, HIG_L_FFN_W2), D.ff, lb + w.y3, d, M, d, D.ff)
                                  .epi(HIG_EPI_BIAS, PL(params, l, HIG_L_FFN_B2)).g, st));
    HIG_TRY(stylize(D.nsty - 1, lb + w.y3, lb + w.a3, hffn, lb + w.h3, HIG_L_FFN_STY_NORM_W, HIG_L_FFN_STY_NORM_B, HIG_L_FFN_STY_OUT_W, HIG_L_FFN_STY_OUT_B));
    hin = lb + w.h3;
  }
  // K6: out = Linear(d, F)(h_L), fp32
  HIG_TRY(hig_gemm16_launch(G16(hin, d, P16(params16, HIG_P_OUT_W), d, out, D.F, M, D.F, d).epi(HIG_EPI_BIAS, P(params, HIG_P_OUT_B)).out32().g, st));
  if (D.two)  // init-pose rows go through out2 instead (interaction_transformer.py:613-614)
    HIG_TRY(hig_gemm16_launch(G16(hin, (int64_t)D.T * d, P16(params16, HIG_P_OUT2_W), d, out, (int64_t)D.T * D.F, D.B, D.F, d)
                                  .epi(HIG_EPI_BIAS, P(params, HIG_P_OUT2_B)).out32().g, st));
  return HIG_OK;
}

static int denoiser_bwd_bf16_impl(const hig_dims* dims, const void* const* params, const void* const* params16, const float* x,
                                  const int64_t* t, const int64_t* length, const float* xf_out, const void* textctx,
                                  const void* workspace, const float* dout, void* const* grads, float* dx, float* dxf_proj,
                                  float* dxf_out, void* bwd_workspace, hig_stream_t stream, hig_layer_hook hook, void* hook_user,
                                  hig_stream_t comm_stream);
extern "C" int hig_denoiser_bwd_bf16(const hig_dims* dims, const void* const* params, const void* const* params16, const float* x,
                                     const int64_t* t, const int64_t* length, const float* xf_out, const void* textctx,
                                     const void* workspace, const float* dout, void* const* grads, float* dx, float* dxf_proj,
                                     float* dxf_out, void* bwd_workspace, hig_stream_t stream) {
  return denoiser_bwd_bf16_impl(dims, params, params16, x, t, length, xf_out, textctx, workspace, dout, grads, dx, dxf_proj, dxf_out,
                                bwd_workspace, stream, nullptr, nullptr, nullptr);
}
// hig_denoiser_bwd_bf16 with the per-layer hook of hig_denoiser_bwd_hooked (include/hig.h): hook(user, l) is called on the host once
// every parameter gradient of decoder layer l -- its rows of the stacked stylization matrix included -- is enqueued, after
// `comm_stream` has been made to wait for both of the backward's streams: a data-parallel caller starts the all-reduce of
// layer l's gradients there, while layers l - 1 ... 0 are still in backward.  Events only, so the call (and the collectives
// its hook enqueues) can be captured into a hipGraph.
extern "C" int hig_denoiser_bwd_bf16_hooked(const hig_dims* dims, const void* const* params, const void* const* params16, const float* x,
                                            const int64_t* t, const int64_t* length, const float* xf_out, const void* textctx,
                                            const void* workspace, const float* dout, void* const* grads, float* dx, float* dxf_proj,
                                            float* dxf_out, void* bwd_workspace, hig_stream_t stream, hig_layer_hook hook,
                                            void* hook_user, hig_stream_t comm_stream) {
  HIG_REQUIRE(hook, "hig_denoiser_bwd_bf16_hooked: null hook");
  return denoiser_bwd_bf16_impl(dims, params, params16, x, t, length, xf_out, textctx, workspace, dout, grads, dx, dxf_proj, dxf_out,
                                bwd_workspace, stream, hook, hook_user, comm_stream);
}
static int denoiser_bwd_bf16_impl(const hig_dims* dims, const void* const* params, const void* const* params16, const float* x,
                                  const int64_t* t, const int64_t* length, const float* xf_out, const void* textctx,
                                  const void* workspace, const float* dout, void* const* grads, float* dx, float* dxf_proj,
                                  float* dxf_out, void* bwd_workspace, hig_stream_t stream, hig_layer_hook hook, void* hook_user,
                                  hig_stream_t comm_stream) {
  (void)t;
  Dims D;
  HIG_TRY(check_dims(dims, D));
  HIG_TRY(require_bf16_train(D, "hig_denoiser_bwd_bf16"));
  HIG_REQUIRE(params && params16 && x && xf_out && textctx && workspace && dout && grads && dxf_proj && dxf_out && bwd_workspace,
              "hig_denoiser_bwd_bf16: null argument");
  const Fwd16TLayout w = fwd16t_layout(D);
  const Text16TLayout tl = text16t_layout(D);
  const Bwd16Layout bw = bwd16_layout(D);
  const char* ws = static_cast<const char*>(workspace);
  const char* tc = static_cast<const char*>(textctx);
  char* b = static_cast<char*>(bwd_workspace);
  hipStream_t st = hig_stream(stream);
  const int d = D.d, E = D.E, ff = D.ff, F = D.F, Lt = D.Lt;
  const int64_t M = D.M, Mt = D.Mt;
  const int64_t ss_ld = (int64_t)D.nsty * D.L * 2 * d;
  float* slabs = reinterpret_cast<float*>(b + bw.slabs);
  float* colp = reinterpret_cast<float*>(b + bw.colpart);
  float* lnp = reinterpret_cast<float*>(b + bw.lnpart);
  float* dss = reinterpret_cast<float*>(b + bw.dss);
  float* dA = reinterpret_cast<float*>(b + bw.dA);
  float* attn = reinterpret_cast<float*>(b + bw.attn);
  float* f32a = reinterpret_cast<float*>(b + bw.f32a);
  float* f32b = reinterpret_cast<float*>(b + bw.f32b);
  const float* ssf = reinterpret_cast<const float*>(ws + w.ss);
  char* tA = b + bw.tA;

  // bf16 storage: the weight gradients stay on the caller's stream unless HIG_BWD_OVERLAP=1 asks for the fork.  The kernels of
  // this mode are one-workgroup-per-CU designs (gemm_wsp16: 159 KB of LDS, wgrad16x: 128 KB + twelve waves): two of them cannot
  // share a CU, so a weight gradient on the second stream delays the workgroups of the data-gradient GEMM CU by CU instead of
  // filling idle slots -- config 2, captured step: 7.27 ms on one stream, 7.40 forked (7.72 with the LayerNorm reductions
  // forked as well).  The fp32 step keeps the fork (tiled kernels, several workgroups per CU: 20.4 vs 21.4 ms eager).
  const int fork16 = bwd_overlap_env() == 1;
  WgradFork fork(fork16 ? side_stream_for_current_device(st) : nullptr, st);
  // dW[n][k] = sum_m dC[m][n] act[m][k] (+ the bias gradient = column sums of dC): both operands transposed to
  // reduce-contiguous bf16 (n_out x Mp), (k_in x Mp), then the split-R bf16 GEMM into the fp32 gradient.  Everything on the
  // weight-gradient stream (protocol: WgradFork).
  // The same batching for the LayerNorm / stylization backward: each call of a layer writes its own partial table and
  // ln_flush() reduces them all (dgamma, dbeta, d(scale, shift)) in one launch at the end of the layer.
  hig_ln_reduce ln_pending[HIG_LN_RB_MAX];
  int ln_n = 0;
  auto ln_flush = [&]() -> int {
    if (ln_n == 0) return HIG_OK;
    HIG_TRY(hig_ln_bwd16_reduce_batch(ln_pending, ln_n, st));
    ln_n = 0;
    return HIG_OK;
  };
  auto ln_bwd16 = [&](const void* da, int64_t ldda, const void* x_, int32_t x_f32, int64_t ldx_, const float* gamma, const float* beta,
                      const float* ss_, int64_t ss_ld_, int32_t shift_off, int32_t mod, const void* res, int64_t ldr, void* dx_, int32_t dx_f32,
                      int64_t lddx, int64_t rows, int32_t n, int32_t rps, float* dgamma, float* dbeta, float* dss_, int64_t dss_ld_) -> int {
    if (ln_n == HIG_LN_RB_MAX) HIG_TRY(ln_flush());
    float* part = lnp + (int64_t)ln_n * (bw.lnpart_slot / 4);
    HIG_TRY(hig_ln_bwd16_launch(da, ldda, x_, x_f32, ldx_, gamma, beta, ss_, ss_ld_, shift_off, mod, res, ldr, dx_, dx_f32, lddx, rows, n, rps,
                                dgamma, dbeta, dss_, dss_ld_, part, stream, &ln_pending[ln_n]));
    if (ln_pending[ln_n].nsplit > 0) ++ln_n;
    return HIG_OK;
  };
  // The slab reductions of the weight gradients are batched: each gradient keeps its own range of the slab scratch and
  // wg_flush() sums everything pending in one launch -- at the end of every decoder layer (before the layer hook), before
  // another user of the slab scratch, when the scratch or the batch is full.  (Inside the captured step a launch costs
  // 2-3 us of dependency latency whatever it does: 66 reductions of ~5 us were 0.32 ms of the 7.0 ms step.)
  hig_wg_reduce wg_pending[HIG_WG_RB_MAX];
  int wg_n = 0;
  int64_t wg_used = 0;
  // ... and the gradients themselves are grouped: wgrad_act() only queues a problem; wg_launch() puts the two to four whose dC
  // operands exist at the same time into ONE launch of the kernel (FFN: stylization out + linear2 + linear1; cross attention:
  // stylization out + query + text key/value; self attention: stylization out + q/k/v; person <-> person the same) -- a third of
  // the slices per gradient, i.e. a third of the slab traffic, and a workgroup's prologue / tail spread over three times the
  // chunks.  It is called where the last operand of a group has been produced and before any of them is overwritten (walk of
  // the layer loop below).  With the weight-gradient fork (HIG_BWD_OVERLAP=1) every gradient is launched at once, as before: the
  // fork's buffer rule counts requests.
  hig_wg_problem wg_q[HIG_WG_GROUP_MAX];
  int wg_qn = 0;
  auto wg_reduce = [&]() -> int {
    if (wg_n == 0) { wg_used = 0; return HIG_OK; }
    HIG_TRY(fork.begin());
    HIG_TRY(hig_wgrad16_reduce_batch(wg_pending, wg_n, fork.stream()));
    wg_n = 0; wg_used = 0;
    return fork.end();
  };
  auto wg_launch = [&]() -> int {
    if (wg_qn == 0) return HIG_OK;
    if (wg_n + wg_qn > HIG_WG_RB_MAX || wg_used > bw.slab_floats / 2) HIG_TRY(wg_reduce());
    int64_t used = 0;
    HIG_TRY(fork.begin());
    HIG_TRY(hig_wgrad16_launch_group(wg_q, wg_qn, slabs + wg_used, bw.slab_floats - wg_used, fork.stream(), &wg_pending[wg_n], &used));
    wg_n += wg_qn;
    wg_used += (used + 63) / 64 * 64;          // (16-byte aligned ranges)
    wg_qn = 0;
    return fork.end();
  };
  auto wg_flush = [&]() -> int {
    HIG_TRY(wg_launch());
    return wg_reduce();
  };
  auto wgrad_act = [&](const void* dC, int n_out, const void* act, int k_in, float* out, int64_t rows, [[maybe_unused]] int64_t rows_p, float* dbias) -> int {
    // straight from the row-major operands (transpose reads), bias gradient in the same pass (wgrad16.hip)
    wg_q[wg_qn++] = hig_wg_problem{dC, n_out, act, k_in, rows, n_out, k_in, out, dbias, 0};
    if (wg_qn == HIG_WG_GROUP_MAX || fork.side) return wg_launch();
    return HIG_OK;
  };
  // fp32 weight gradients at the F-wide edges (operands fp32, reduce-slow): the fp32 kernel, split over the rows
  auto wgrad32 = [&](G gd) -> int {
    HIG_TRY(wg_flush());
    HIG_TRY(fork.begin());
    const int s = wgrad_splits(gd.g.I, gd.g.J, gd.g.R, bw.slab_floats, gd.g.prec);
    HIG_TRY(hig_gemm_launch(gd.g, s, slabs, fork.stream()));
    return fork.end();
  };
  // one layer's transposed weights (bf16): operands of the data-gradient GEMMs dX = dC . W = dC . (W^T)^T
  char* wT = b + bw.wT;
  const int64_t o_sty3 = 0, o_w2t = o_sty3 + (int64_t)d * d * 2, o_w1t = o_w2t + (int64_t)d * ff * 2, o_sty2 = o_w1t + (int64_t)d * ff * 2,
                o_caq = o_sty2 + (int64_t)d * d * 2, o_kv = o_caq + (int64_t)d * d * 2, o_sty1 = o_kv + (int64_t)2 * d * Lt * 2,
                o_qkv = o_sty1 + (int64_t)d * d * 2, o_isty = o_qkv + (int64_t)3 * d * d * 2, o_iqkv = o_isty + (int64_t)d * d * 2;
  const int Bp = D.B / 2;
  auto dgrad = [&](const void* dC, int n_out, int64_t wt_off, void* dst, int k_in, int64_t rows, int epi, const void* res, int64_t ldr) -> int {
    G16 g(dC, n_out, wT + wt_off, n_out, dst, k_in, rows, k_in, n_out);
    g.g.epi = epi;
    if (res) g.res16(res, ldr);
    return hig_gemm16_launch(g.g, st);
  };
  auto sty_bwd = [&](int l, int s, const void* dh, const void* y, const void* a_saved, int norm_w, int norm_b, int out_w, int out_b,
                     int64_t wt_off, void* dy_out) -> int {
    HIG_TRY(wgrad_act(dh, d, a_saved, d, GL(grads, l, out_w), M, bw.Mp, GL(grads, l, out_b)));
    HIG_TRY(dgrad(dh, d, wt_off, b + bw.t1, d, M, HIG_EPI_NONE, nullptr, 0));
    return ln_bwd16(b + bw.t1, d, y, 0, d, PL(params, l, norm_w), PL(params, l, norm_b), ssf + (int64_t)s * 2 * d, ss_ld, d, 1, nullptr, 0,
                           dy_out, 0, d, M, d, D.T, GL(grads, l, norm_w), GL(grads, l, norm_b), dss + (int64_t)s * 2 * d, ss_ld);
  };

  // ---- output projection ---------------------------------------------------------------------------------------------------
  // The F-wide edges on the bf16 matrix kernels too (HIG_EDGE16=0: the fp32 kernels on fp32 copies, as before): d(out) and x
  // are rounded to bf16 rows padded to Fp = F rounded up to 32 (hig_cast_pad_bf16), the data gradient d(h_L) = d(out) W_out is a
  // bf16 GEMM over Fp (W_out^T padded with zero columns), the two weight gradients run on wgrad16 into padded fp32 scratch
  // and are copied into place.  Scratch: the first of the two (M, d) fp32 buffers the fp32 path needs.
  const char* hL = ws + w.layer0 + w.lstride * (D.L - 1) + w.h3;
  char* dh = b + bw.dhA;
  char* dh_alt = b + bw.dhB;
  static const int edge16_env = getenv("HIG_EDGE16") ? atoi(getenv("HIG_EDGE16")) : 1;   // tuning knob
  const int Fp = (F + 31) / 32 * 32;
  auto up256 = [](int64_t v) { return (v + 255) / 256 * 256; };
  const int64_t e_dout = 0, e_x = e_dout + up256(M * Fp * 2), e_wot = e_x + up256(M * Fp * 2), e_dwo = e_wot + up256((int64_t)d * Fp * 2),
                e_dwj = e_dwo + up256((int64_t)Fp * d * 4), e_dbo = e_dwj + up256((int64_t)d * Fp * 4), e_end = e_dbo + up256((int64_t)Fp * 4);
  const bool edge16 = edge16_env && d % 8 == 0 && e_end <= M * d * 4 && M * (int64_t)Fp < (1ll << 30);
  char* edgeb = reinterpret_cast<char*>(f32a);
  // weight gradient on wgrad16 into a padded scratch, then the real rows / columns into the gradient (weight-gradient stream)
  auto wgrad_edge = [&](const void* dC, int n_out, const void* act, int k_in, float* scratch, float* sbias, float* out, int out_rows, int out_cols,
                        float* out_bias, int nbias) -> int {
    HIG_TRY(wg_flush());
    HIG_TRY(fork.begin());
    HIG_TRY(hig_wgrad16_launch(dC, n_out, act, k_in, M, n_out, k_in, scratch, sbias, 0, slabs, bw.slab_floats, fork.stream()));
    // (kernels, not hipMemcpy2DAsync / hipMemcpyAsync nodes: see the init-pose rows below)
    hipLaunchKernelGGL(copy2d_f32_kernel, dim3((unsigned)(((int64_t)out_rows * out_cols + 255) / 256)), dim3(256), 0, fork.stream(), scratch, k_in, out,
                       out_cols, out_rows, out_cols);
    HIG_CHECK_LAUNCH();
    if (out_bias && sbias != out_bias) {
      hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((nbias + 255) / 256)), dim3(256), 0, fork.stream(), sbias, out_bias, (int64_t)nbias);
      HIG_CHECK_LAUNCH();
    }
    return fork.end();
  };
  float* tok0 = reinterpret_cast<float*>(b + bw.tok0);
  if (D.two) {
    // init-pose rows went through out2 (interaction_transformer.py:613): B rows -- the fp32 kernels on fp32 copies.
    // d(b_out2) = column sums of d(out)[:, 0, :], d(W_out2) = d(out)[:, 0, :]^T h_L[:, 0, :]
    float* hl0 = reinterpret_cast<float*>(b + bw.hl0);
    hipLaunchKernelGGL(tok0_bf16_kernel, dim3((D.B * d + 255) / 256), dim3(256), 0, st, reinterpret_cast<__bf16*>(const_cast<char*>(hL)),
                       (int64_t)D.T * d, D.B, d, hl0, 0);
    HIG_CHECK_LAUNCH();
    HIG_TRY(hig_colsum(dout, (int64_t)D.T * F, D.B, F, GP(grads, HIG_P_OUT2_B), colp, stream));
    HIG_TRY(wgrad32(G(dout, (int64_t)D.T * F, 1, hl0, d, 1, GP(grads, HIG_P_OUT2_W), d, F, d, D.B)));
  }
  // dh[:, 0, :] = d(out)[:, 0, :] W_out2 over the exact zeros the `out` adjoint leaves there (two-person)
  auto out2_rows = [&](char* dh_bf16) -> int {
    HIG_TRY(hig_gemm_launch(G(dout, (int64_t)D.T * F, 0, P(params, HIG_P_OUT2_W), d, 1, tok0, d, D.B, d, F).g, 1, nullptr, st));
    hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3((unsigned)(((int64_t)D.B * d + 255) / 256)), dim3(256), 0, st, tok0, (int64_t)d,
                       reinterpret_cast<__bf16*>(dh_bf16), (int64_t)D.T * d, D.B, d);
    HIG_CHECK_LAUNCH();
    return HIG_OK;
  };
  if (edge16) {
    HIG_TRY(hig_cast_pad_bf16(dout, F, M, F, edgeb + e_dout, Fp, stream));
    if (D.two) {   // the `out` adjoints must not see the init-pose rows: zero them in the rounded copy
      // (a kernel, not hipMemset2DAsync: as a node of the captured step the 2-D memset ran out of order with its neighbours
      // whenever the graph was launched onto an idle GPU -- tools/sync_pattern_probe.py, profiles/r05_notes.md section 8)
      hipLaunchKernelGGL(tok0_bf16_kernel, dim3((D.B * Fp + 255) / 256), dim3(256), 0, st, reinterpret_cast<__bf16*>(edgeb + e_dout), (int64_t)D.T * Fp,
                         D.B, Fp, static_cast<float*>(nullptr), 1);
      HIG_CHECK_LAUNCH();
    }
    HIG_TRY(hig_cast_pad_bf16(x, F, M, F, edgeb + e_x, Fp, stream));
    hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)(((int64_t)d * Fp / 2 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(edgeb + e_wot), (int64_t)d * Fp / 2);
    HIG_CHECK_LAUNCH();
    HIG_TRY(hig_transpose_bf16(P16(params16, HIG_P_OUT_W), d, F, d, edgeb + e_wot, Fp, stream));   // (F, d) -> (d, Fp), pad columns zero
    // d(W_out) (F, d) and d(b_out) (F) = d(out)^T h_L and its column sums
    HIG_TRY(wgrad_edge(edgeb + e_dout, Fp, hL, d, reinterpret_cast<float*>(edgeb + e_dwo), reinterpret_cast<float*>(edgeb + e_dbo), GP(grads, HIG_P_OUT_W), F, d,
                       GP(grads, HIG_P_OUT_B), F));
    // d(h_L) = d(out) W_out
    HIG_TRY(hig_gemm16_launch(G16(edgeb + e_dout, Fp, edgeb + e_wot, Fp, dh, d, M, d, Fp).g, st));
    if (D.two) HIG_TRY(out2_rows(dh));
  } else {
    const float* dout_m = dout;   // rows that went through `out`
    if (D.two) {                  // (the fp32 edge: a copy of d(out) with the init-pose rows zeroed, in the second fp32 scratch)
      hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((M * F + 255) / 256)), dim3(256), 0, st, dout, f32b, M * F);
      HIG_CHECK_LAUNCH();
      hipLaunchKernelGGL(tok0_kernel, dim3((D.B * F + 255) / 256), dim3(256), 0, st, f32b, (int64_t)D.T * F, D.B, F, (float*)nullptr, 1);
      HIG_CHECK_LAUNCH();
      dout_m = f32b;
    }
    HIG_TRY(hig_colsum(dout_m, F, M, F, GP(grads, HIG_P_OUT_B), colp, stream));
    HIG_TRY(hig_cast_f32(hL, f32a, M * d, stream));
    HIG_TRY(wgrad32(G(dout_m, F, 1, f32a, d, 1, GP(grads, HIG_P_OUT_W), d, F, d, M)));
    float* dh32 = D.two ? reinterpret_cast<float*>(tA) : f32b;   // (two-person: f32b holds d(out) with zeroed rows; the transpose scratch is idle)
    HIG_TRY(hig_gemm_launch(G(dout_m, F, 0, P(params, HIG_P_OUT_W), d, 1, dh32, d, M, d, F).g, 1, nullptr, st));
    HIG_TRY(hig_cast_bf16(dh32, dh, M * d, stream));
    if (D.two) HIG_TRY(out2_rows(dh));
  }

  for (int l = D.L - 1; l >= 0; --l) {
    const char* lb = ws + w.layer0 + w.lstride * l;
    const char* tlb = tc + tl.layer0 + tl.lstride * l;
    const char* hin = l == 0 ? ws + w.h0 : ws + w.layer0 + w.lstride * (l - 1) + w.h3;
    {  // all W -> W^T copies of this layer (from the bf16 shadow) in one launch
      const void* srcs[10];
      void* dsts[10];
      int64_t lds_[10], ldd[10];
      int32_t rws[10], cls[10];
      int n = 0;
      auto add = [&](int idx, int out_f, int in_f, int64_t off) {
        srcs[n] = PL16(params16, l, idx); dsts[n] = wT + off; lds_[n] = in_f; ldd[n] = out_f; rws[n] = out_f; cls[n] = in_f; ++n;
      };
      add(HIG_L_FFN_STY_OUT_W, d, d, o_sty3);
      add(HIG_L_FFN_W2, d, ff, o_w2t);
      add(HIG_L_FFN_W1, ff, d, o_w1t);
      add(HIG_L_CA_STY_OUT_W, d, d, o_sty2);
      add(HIG_L_CA_Q_W, d, d, o_caq);
      add(HIG_L_CA_KV_W, 2 * d, Lt, o_kv);
      add(HIG_L_SA_STY_OUT_W, d, d, o_sty1);
      add(HIG_L_SA_QKV_W, 3 * d, d, o_qkv);
      if (D.two == 1) {
        add(HIG_L_INT_STY_OUT_W, d, d, o_isty);
        add(HIG_L_INT_QKV_W, 3 * d, d, o_iqkv);
      }
      HIG_TRY(hig_transpose_bf16_batch(n, srcs, lds_, dsts, ldd, rws, cls, stream));
    }
    const char* hffn = D.two == 1 ? lb + w.h2b : lb + w.h2;
    // ---- FFN --------------------------------------------------------------------------
    HIG_TRY(sty_bwd(l, D.nsty * l + D.nsty - 1, dh, lb + w.y3, lb + w.a3, HIG_L_FFN_STY_NORM_W, HIG_L_FFN_STY_NORM_B, HIG_L_FFN_STY_OUT_W,
                    HIG_L_FFN_STY_OUT_B, o_sty3, b + bw.t2));
    const char* dy3 = b + bw.t2;
    HIG_TRY(wgrad_act(dy3, d, lb + w.f1, ff, GL(grads, l, HIG_L_FFN_W2), M, bw.Mp, GL(grads, l, HIG_L_FFN_B2)));
    HIG_TRY(dgrad(dy3, d, o_w2t, b + bw.tff, ff, M, HIG_EPI_DGELU, lb + w.z1, ff));     // dz = (dy3 . W2) gelu'(z)
    const char* dz1 = b + bw.tff;
    HIG_TRY(wgrad_act(dz1, ff, hffn, d, GL(grads, l, HIG_L_FFN_W1), M, bw.Mp, GL(grads, l, HIG_L_FFN_B1)));
    HIG_TRY(wg_launch());     // FFN group: d(h3) (dh), dy3 (t2) and dz (tff) all exist and none has been overwritten yet
    HIG_TRY(dgrad(dz1, ff, o_w1t, dh_alt, d, M, HIG_EPI_RES, dh, d));                  // d(h2) = d(h3) + dz . W1
    { char* tmp = dh; dh = dh_alt; dh_alt = tmp; }  // dh = d(h2), or d(h2b) in the interaction model
    if (D.two == 1) {
      // ---- person <-> person cross attention (adjoint of interaction_transformer.py:181-205) ------------------------------------
      HIG_TRY(sty_bwd(l, D.nsty * l + 2, dh, lb + w.y4, lb + w.a4, HIG_L_INT_STY_NORM_W, HIG_L_INT_STY_NORM_B, HIG_L_INT_STY_OUT_W,
                      HIG_L_INT_STY_OUT_B, o_isty, b + bw.t2));
      const int64_t* len_partner = reinterpret_cast<const int64_t*>(ws + w.lenp);
      const int64_t halfA = (int64_t)Bp * D.H * D.hd * D.hd, halfM = (int64_t)Bp * D.T;
      char* dqkv = b + bw.dqkv;
      const float* Ai = reinterpret_cast<const float*>(lb + w.Ai);
      const char* iqkv = lb + w.iqkv;
      // consumer sample s read the context of producer (s + B/2) % B: route d(A) back the same way
      HIG_TRY(hig_linattn_apply_bwd_bf16(b + bw.t2, d, iqkv, 3 * d, Ai + halfA, dqkv, 3 * d, dA + halfA, Bp, D.T, D.H, D.hd, attn, stream));
      HIG_TRY(hig_linattn_apply_bwd_bf16(b + bw.t2 + halfM * d * 2, d, iqkv + halfM * 3 * d * 2, 3 * d, Ai, dqkv + halfM * 3 * d * 2, 3 * d, dA, Bp,
                                         D.T, D.H, D.hd, attn, stream));
      HIG_TRY(hig_linattn_ctx_bwd_bf16(dA, Ai, iqkv + (int64_t)d * 2, iqkv + (int64_t)2 * d * 2, 3 * d, reinterpret_cast<const float*>(lb + w.ksti),
                                       len_partner, dqkv + (int64_t)d * 2, dqkv + (int64_t)2 * d * 2, 3 * d, D.B, D.T, D.H, D.hd, stream));
      HIG_TRY(wgrad_act(dqkv, 3 * d, lb + w.xn3, d, GL(grads, l, HIG_L_INT_QKV_W), M, bw.Mp, GL(grads, l, HIG_L_INT_QKV_B)));
      HIG_TRY(wg_launch());   // person <-> person group: d(h2b) (dh) and dqkv
      HIG_TRY(dgrad(dqkv, 3 * d, o_iqkv, b + bw.t2, d, M, HIG_EPI_NONE, nullptr, 0));
      HIG_TRY(ln_bwd16(b + bw.t2, d, lb + w.h2, 0, d, PL(params, l, HIG_L_INT_NORM_W), PL(params, l, HIG_L_INT_NORM_B), nullptr, 0, 0, 0, dh, d,
                              dh_alt, 0, d, M, d, D.T, GL(grads, l, HIG_L_INT_NORM_W), GL(grads, l, HIG_L_INT_NORM_B), nullptr, 0));
      { char* tmp = dh; dh = dh_alt; dh_alt = tmp; }  // dh = d(h2)
    }
    // ---- cross attention ---------------------------------------------------------------
    HIG_TRY(sty_bwd(l, D.nsty * l + 1, dh, lb + w.y2, lb + w.a2, HIG_L_CA_STY_NORM_W, HIG_L_CA_STY_NORM_B, HIG_L_CA_STY_OUT_W,
                    HIG_L_CA_STY_OUT_B, o_sty2, b + bw.t2));
    const float* Ac = reinterpret_cast<const float*>(tlb + tl.Ac);
    HIG_TRY(hig_linattn_apply_bwd_bf16(b + bw.t2, d, lb + w.qc, d, Ac, b + bw.t1, d, dA, D.B, D.T, D.H, D.hd, attn, stream));
    const char* dqc = b + bw.t1;
    HIG_TRY(wgrad_act(dqc, d, lb + w.xn2, d, GL(grads, l, HIG_L_CA_Q_W), M, bw.Mp, GL(grads, l, HIG_L_CA_Q_B)));
    HIG_TRY(dgrad(dqc, d, o_caq, b + bw.t2, d, M, HIG_EPI_NONE, nullptr, 0));
    HIG_TRY(ln_bwd16(b + bw.t2, d, lb + w.h1, 0, d, PL(params, l, HIG_L_CA_NORM_W), PL(params, l, HIG_L_CA_NORM_B), nullptr, 0, 0, 0, dh, d,
                            dh_alt, 0, d, M, d, D.T, GL(grads, l, HIG_L_CA_NORM_W), GL(grads, l, HIG_L_CA_NORM_B), nullptr, 0));
    { char* tmp = dh; dh = dh_alt; dh_alt = tmp; }  // dh = d(h1)
    // text side of this layer: d(A_c) -> d(key, value) -> text_norm -> d(xf_out)
    HIG_TRY(hig_linattn_ctx_bwd_bf16(dA, Ac, tlb + tl.kv, tlb + tl.kv + (int64_t)d * 2, 2 * d, reinterpret_cast<const float*>(tlb + tl.kstc), nullptr,
                                     b + bw.dkv, b + bw.dkv + (int64_t)d * 2, 2 * d, D.B, D.N, D.H, D.hd, stream));
    HIG_TRY(wgrad_act(b + bw.dkv, 2 * d, tlb + tl.xfn, Lt, GL(grads, l, HIG_L_CA_KV_W), Mt, bw.Mtp, GL(grads, l, HIG_L_CA_KV_B)));
    HIG_TRY(wg_launch());     // cross-attention group: d(h2) (the buffer dh pointed to when it was queued), dqc (t1), dkv
    HIG_TRY(dgrad(b + bw.dkv, 2 * d, o_kv, b + bw.dxfn, Lt, Mt, HIG_EPI_NONE, nullptr, 0));
    HIG_TRY(ln_bwd16(b + bw.dxfn, Lt, xf_out, 1, Lt, PL(params, l, HIG_L_CA_TNORM_W), PL(params, l, HIG_L_CA_TNORM_B), nullptr, 0, 0, 0,
                            l == D.L - 1 ? nullptr : dxf_out, Lt, dxf_out, 1, Lt, Mt, Lt, D.N, GL(grads, l, HIG_L_CA_TNORM_W),
                            GL(grads, l, HIG_L_CA_TNORM_B), nullptr, 0));
    // ---- self attention ----------------------------------------------------------------
    HIG_TRY(sty_bwd(l, D.nsty * l, dh, lb + w.y1, lb + w.a1, HIG_L_SA_STY_NORM_W, HIG_L_SA_STY_NORM_B, HIG_L_SA_STY_OUT_W,
                    HIG_L_SA_STY_OUT_B, o_sty1, b + bw.t2));
    char* dqkv = b + bw.dqkv;
    const float* A1 = reinterpret_cast<const float*>(lb + w.A1);
    HIG_TRY(hig_linattn_apply_bwd_bf16(b + bw.t2, d, lb + w.qkv, 3 * d, A1, dqkv, 3 * d, dA, D.B, D.T, D.H, D.hd, attn, stream));
    HIG_TRY(hig_linattn_ctx_bwd_bf16(dA, A1, lb + w.qkv + (int64_t)d * 2, lb + w.qkv + (int64_t)2 * d * 2, 3 * d,
                                     reinterpret_cast<const float*>(lb + w.kst1), length, dqkv + (int64_t)d * 2, dqkv + (int64_t)2 * d * 2, 3 * d,
                                     D.B, D.T, D.H, D.hd, stream));
    HIG_TRY(wgrad_act(dqkv, 3 * d, lb + w.xn1, d, GL(grads, l, HIG_L_SA_QKV_W), M, bw.Mp, GL(grads, l, HIG_L_SA_QKV_B)));
    HIG_TRY(wg_launch());     // self-attention group: d(h1) (dh) and dqkv
    HIG_TRY(dgrad(dqkv, 3 * d, o_qkv, b + bw.t2, d, M, HIG_EPI_NONE, nullptr, 0));
    HIG_TRY(ln_bwd16(b + bw.t2, d, hin, 0, d, PL(params, l, HIG_L_SA_NORM_W), PL(params, l, HIG_L_SA_NORM_B), nullptr, 0, 0, 0, dh, d, dh_alt,
                            0, d, M, d, D.T, GL(grads, l, HIG_L_SA_NORM_W), GL(grads, l, HIG_L_SA_NORM_B), nullptr, 0));
    { char* tmp = dh; dh = dh_alt; dh_alt = tmp; }  // dh = d(h_in of this layer)
    HIG_TRY(wg_flush());      // the layer's weight gradients are final from here on
    HIG_TRY(ln_flush());
    if (hook) {
      // Every parameter gradient of layer l exists once this layer's rows of the stacked stylization matrix are done too (they
      // only need this layer's columns of dss): d(W_emb)[l] = dss[:, l]^T . silu(emb), one more request on the weight-gradient
      // stream (hig_denoiser_bwd_hooked does the same).  Then `comm_stream` waits for both streams and the caller is told.
      const int64_t rows_l = (int64_t)D.nsty * 2 * d;
      HIG_TRY(wgrad32(G(dss + (int64_t)l * rows_l, ss_ld, 1, reinterpret_cast<const float*>(ws + w.emb), E, 1,
                        GP(grads, HIG_P_STY_EMB_W) + (int64_t)l * rows_l * E, E, rows_l, E, D.B).silu(1)));
      if (comm_stream) {
        hipStream_t cs = hig_stream(comm_stream);
        hipEvent_t& ev_layer = layer_event();
        if (!ev_layer && hipEventCreateWithFlags(&ev_layer, hipEventDisableTiming) != hipSuccess)
          return hig_set_error(HIG_EHIP, "hipEventCreate failed");
        if (hipEventRecord(ev_layer, st) != hipSuccess || hipStreamWaitEvent(cs, ev_layer, 0) != hipSuccess)
          return hig_set_error(HIG_EHIP, "layer hook: event on the caller's stream failed");
        HIG_TRY(fork.wait_on(cs));       // (and for everything on the weight-gradient stream)
      }
      hook(hook_user, l);
    }
  }

  // ---- joint_embed + sequence_embedding (F-wide: fp32 kernels on an fp32 copy of d(h0)) ---------------------------------------
  if (D.two) {
    // init-pose rows came from joint_embed2 (no positional term): move them aside (fp32, B rows), leaving zeros so that the
    // joint_embed / sequence_embedding adjoints below see only the motion rows                 (interaction_transformer.py:596)
    hipLaunchKernelGGL(tok0_bf16_kernel, dim3((D.B * d + 255) / 256), dim3(256), 0, st, reinterpret_cast<__bf16*>(dh), (int64_t)D.T * d, D.B, d, tok0, 1);
    HIG_CHECK_LAUNCH();
    HIG_TRY(hig_colsum(tok0, d, D.B, d, GP(grads, HIG_P_JOINT2_B), colp, stream));
    HIG_TRY(wgrad32(G(tok0, d, 1, x, (int64_t)D.T * F, 1, GP(grads, HIG_P_JOINT2_W), 4, d, 4, D.B)));
  }
  const int Tpos = D.two ? D.T - 1 : D.T;        // rows of sequence_embedding that were used (two-person: frame t used row t - 1)
  float* dpos = D.two ? reinterpret_cast<float*>(b + bw.postmp) : GP(grads, HIG_P_SEQ_EMB);
  if (edge16) {
    // d(W_joint) (d, F) and d(b_joint) (d) = d(h_0)^T x and its column sums; the position table's gradient = d(h_0) summed over samples
    HIG_TRY(wgrad_edge(dh, d, edgeb + e_x, Fp, reinterpret_cast<float*>(edgeb + e_dwj), GP(grads, HIG_P_JOINT_B), GP(grads, HIG_P_JOINT_W), d, F,
                       GP(grads, HIG_P_JOINT_B), d));
    HIG_TRY(hig_colsum_bf16(dh, (int64_t)D.T * d, D.B, D.T * d, dpos, colp, stream));
  } else {
    HIG_TRY(hig_colsum_bf16(dh, d, M, d, GP(grads, HIG_P_JOINT_B), colp, stream));
    HIG_TRY(hig_cast_f32(dh, f32b, M * d, stream));
    HIG_TRY(wgrad32(G(f32b, d, 1, x, F, 1, GP(grads, HIG_P_JOINT_W), F, d, F, M)));
    HIG_TRY(hig_colsum(f32b, (int64_t)D.T * d, D.B, D.T * d, dpos, colp, stream));
  }
  if (D.two) {   // (frame t used row t - 1 of the table: shift by one row; a kernel for the same reason as the init-pose rows above)
    const int64_t nsh = (int64_t)Tpos * d;
    hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((nsh + 255) / 256)), dim3(256), 0, st, dpos + d, GP(grads, HIG_P_SEQ_EMB), nsh);
    HIG_CHECK_LAUNCH();
  }
  if (D.nf > Tpos) {
    const int64_t nz = (int64_t)(D.nf - Tpos) * d;
    hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(GP(grads, HIG_P_SEQ_EMB) + (int64_t)Tpos * d), nz);
    HIG_CHECK_LAUNCH();
  }
  if (dx) {   // (input gradient: asked for by tests only -- the fp32 kernel on an fp32 copy of d(h_0))
    if (edge16) HIG_TRY(hig_cast_f32(dh, f32b, M * d, stream));
    HIG_TRY(hig_gemm_launch(G(f32b, d, 0, P(params, HIG_P_JOINT_W), F, 1, dx, F, M, F, d).g, 1, nullptr, st));
    if (D.two)  // d(x[:, 0, :4]) through joint_embed2; the other features of the init-pose row are unused
      HIG_TRY(hig_gemm_launch(G(tok0, d, 0, P(params, HIG_P_JOINT2_W), 4, 1, dx, (int64_t)D.T * F, D.B, 4, d).g, 1, nullptr, st));
  }

  // ---- time / text embedding path (fp32, as hig_denoiser_bwd) --------------------------------------------------------------
  HIG_TRY(wg_flush());
  HIG_TRY(ln_flush());
  HIG_TRY(fork.join());
  const float* emb = reinterpret_cast<const float*>(ws + w.emb);
  const float* te_h = reinterpret_cast<const float*>(ws + w.te_h);
  const float* te = reinterpret_cast<const float*>(ws + w.te);
  float* demb = reinterpret_cast<float*>(b + bw.demb);
  float* dtmp = reinterpret_cast<float*>(b + bw.dtmp);
  float* dte_h = reinterpret_cast<float*>(b + bw.dte_h);
  HIG_TRY(hig_colsum(dss, ss_ld, D.B, (int)ss_ld, GP(grads, HIG_P_STY_EMB_B), colp, stream));
  if (!hook)   // (with a layer hook the rows of each layer were produced inside that layer's backward)
    HIG_TRY(hig_gemm_launch(G(dss, ss_ld, 1, emb, E, 1, GP(grads, HIG_P_STY_EMB_W), E, ss_ld, E, D.B).silu(1).g, 1, nullptr, st));
  {
    G gd(dss, ss_ld, 0, P(params, HIG_P_STY_EMB_W), E, 1, dtmp, E, D.B, E, ss_ld);
    int s = (int)(ss_ld / 1024);
    if (s > 16) s = 16;
    HIG_TRY(hig_gemm_launch(gd.g, s < 1 ? 1 : s, slabs, st));
  }
  const int64_t nBE = (int64_t)D.B * E;
  const int eb = (int)((nBE + 255) / 256);
  hipLaunchKernelGGL(mul_dsilu_kernel, dim3(eb), dim3(256), 0, st, dtmp, emb, nBE, demb);
  HIG_CHECK_LAUNCH();
  hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((nBE + 255) / 256)), dim3(256), 0, st, demb, dxf_proj, nBE);
  HIG_CHECK_LAUNCH();
  HIG_TRY(hig_colsum(demb, E, D.B, E, GP(grads, HIG_P_TE2_B), colp, stream));
  HIG_TRY(hig_gemm_launch(G(demb, E, 1, te_h, E, 1, GP(grads, HIG_P_TE2_W), E, E, E, D.B).silu(1).g, 1, nullptr, st));
  HIG_TRY(hig_gemm_launch(G(demb, E, 0, P(params, HIG_P_TE2_W), E, 1, dtmp, E, D.B, E, E).g, 1, nullptr, st));
  hipLaunchKernelGGL(mul_dsilu_kernel, dim3(eb), dim3(256), 0, st, dtmp, te_h, nBE, dte_h);
  HIG_CHECK_LAUNCH();
  HIG_TRY(hig_colsum(dte_h, E, D.B, E, GP(grads, HIG_P_TE0_B), colp, stream));
  HIG_TRY(hig_gemm_launch(G(dte_h, E, 1, te, d, 1, GP(grads, HIG_P_TE0_W), d, E, d, D.B).g, 1, nullptr, st));
  return HIG_OK;
}
