// Launch sequences of the text head: text_pre_proj -> post-norm TransformerEncoder
// -> text_ln -> EOT gather -> text_proj.  Reference: MotionTransformer.__init__ / encode_text
// (codes/models/transformer.py:324-340, 389-397) over torch's nn.TransformerEncoderLayer
// (norm_first=False, activation "gelu", dropout 0):
//     x = norm1(x + out_proj(softmax(q k^T / sqrt(hd)) v));   x = norm2(x + linear2(gelu(linear1(x))))
// Host code only (kernels live in gemm / fullattn / rowops); hipGraph-capturable like the denoiser.
#include "hig_common.h"
#include "hig_host.h"

namespace {

struct TDims {
  int B, N, W, Lt, H, ff, L, E, hd, prec, pre;
  int64_t M;
};

int check_tdims(const hig_text_dims* p, const void* const* params, TDims& D) {
  HIG_REQUIRE(p, "null text dims");
  D.B = p->B; D.N = p->N; D.W = p->W; D.Lt = p->Lt; D.H = p->H; D.ff = p->ff; D.L = p->L; D.E = p->E;
  HIG_REQUIRE(D.B > 0 && D.N > 0 && D.W > 0 && D.Lt > 0 && D.H > 0 && D.ff > 0 && D.L > 0 && D.E > 0,
              "hig_text_dims: every extent must be positive");
  HIG_REQUIRE(D.Lt % D.H == 0, "hig_text_dims: Lt=%d not divisible by H=%d", D.Lt, D.H);
  D.hd = D.Lt / D.H;
  if (!(D.hd == 8 || D.hd == 16 || D.hd == 32 || D.hd == 64 || D.hd == 128))
    return hig_set_error(HIG_EUNSUPPORTED, "hig text head: head dim %d not in {8,16,32,64,128}", D.hd);
  HIG_REQUIRE(D.Lt % 4 == 0 && D.W % 4 == 0 && D.ff % 4 == 0 && D.E % 4 == 0 && D.Lt <= 1024,
              "hig_text_dims: W, Lt, ff, E must be multiples of 4 and Lt <= 1024");
  if (p->prec != HIG_PREC_F32 && p->prec != HIG_PREC_BF16X3 && p->prec != HIG_PREC_BF16)
    return hig_set_error(HIG_EINVAL, "hig: unknown prec=%d", p->prec);
  D.prec = p->prec;
  D.pre = 1;
  if (params) {
    D.pre = params[HIG_T_PRE_W] != nullptr;
    HIG_REQUIRE(D.pre || D.W == D.Lt, "hig text head: identity text_pre_proj needs W == Lt");
  }
  D.M = (int64_t)D.B * D.N;
  return HIG_OK;
}

// Forward workspace (floats); the per-layer block is repeated L times when training.
struct TFwd {
  int64_t x0, gath, stf, layer0, lstride;
  int64_t qkv, lse, att, r1, st1, x1, z, f, r2, st2, x2;
  int64_t total;
};
TFwd tfwd_layout(const TDims& D, int training) {
  TFwd w;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += al(n); return r; };
  w.x0 = take(D.M * D.Lt);
  w.gath = take((int64_t)D.B * D.Lt);
  w.stf = take(D.M * 2);
  w.layer0 = o;
  o = 0;
  w.qkv = take(D.M * 3 * D.Lt);
  w.lse = take((int64_t)D.B * D.H * D.N);
  w.att = take(D.M * D.Lt);
  w.r1 = take(D.M * D.Lt);
  w.st1 = take(D.M * 2);
  w.x1 = take(D.M * D.Lt);
  w.z = take(D.M * D.ff);
  w.f = take(D.M * D.ff);
  w.r2 = take(D.M * D.Lt);
  w.st2 = take(D.M * 2);
  w.x2 = take(D.M * D.Lt);
  w.lstride = training ? o : 0;
  // inference ping-pongs x2 between two blocks so a layer never overwrites its own input
  w.total = w.layer0 + (training ? o * D.L : 2 * o);
  return w;
}

struct TBwd {
  int64_t dA, dB, dC, dff, dqkv, delta, dgath, wT, tA, tB, slabs, slab_floats, colpart, lnpart, total;
};
TBwd tbwd_layout(const TDims& D) {
  TBwd w;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += al(n); return r; };
  w.dA = take(D.M * D.Lt);
  w.dB = take(D.M * D.Lt);
  w.dC = take(D.M * D.Lt);
  w.dff = take(D.M * D.ff);
  w.dqkv = take(D.M * 3 * D.Lt);
  w.delta = take((int64_t)D.B * D.H * D.N);
  w.dgath = take((int64_t)D.B * D.Lt);
  int64_t big = (int64_t)3 * D.Lt * D.Lt;
  const int64_t outs[] = {(int64_t)D.ff * D.Lt, (int64_t)D.Lt * D.W, (int64_t)D.E * D.Lt};
  for (int64_t v : outs) big = v > big ? v : big;
  w.wT = take(big);
  const int64_t wide = 3 * D.Lt > D.ff ? 3 * D.Lt : D.ff;
  w.tA = take(wide * D.M);
  w.tB = take((int64_t)(D.ff > D.W ? D.ff : D.W) * D.M);
  w.slab_floats = big * 16 > (int64_t)1536 * 128 * 128 ? big * 16 : (int64_t)1536 * 128 * 128;
  w.slabs = take(w.slab_floats);
  int64_t colp = 0;
  const int64_t uses[][2] = {{D.M, 3 * D.Lt}, {D.M, D.ff}, {D.B, D.E}};
  for (auto& u : uses) {
    const int64_t v = (int64_t)hig_colsum_chunks(u[0]) * u[1];
    colp = v > colp ? v : colp;
  }
  w.colpart = take(colp);
  w.lnpart = take(hig_ln_bwd_partial_floats(D.M, D.Lt, D.N));
  w.total = o;
  return w;
}

inline const float* TP(const void* const* t, int idx) { return static_cast<const float*>(t[idx]); }
inline const float* TPL(const void* const* t, int l, int idx) {
  return static_cast<const float*>(t[HIG_T_NGLOBAL + l * HIG_TL_NLAYER + idx]);
}
inline float* TG(void* const* t, int idx) { return static_cast<float*>(t[idx]); }
inline float* TGL(void* const* t, int l, int idx) {
  return static_cast<float*>(t[HIG_T_NGLOBAL + l * HIG_TL_NLAYER + idx]);
}

}  // namespace

extern "C" int64_t hig_text_head_workspace_bytes(const hig_text_dims* dims, int training) {
  TDims D;
  if (check_tdims(dims, nullptr, D) != HIG_OK) return -1;
  return tfwd_layout(D, training).total * 4;
}
extern "C" int64_t hig_text_head_bwd_workspace_bytes(const hig_text_dims* dims) {
  TDims D;
  if (check_tdims(dims, nullptr, D) != HIG_OK) return -1;
  return tbwd_layout(D).total * 4;
}

extern "C" int hig_text_head_fwd(const hig_text_dims* dims, const void* const* params, const float* clip_out,
                                 const int64_t* eot, float* xf_out, float* xf_proj, void* workspace,
                                 int training, hig_stream_t stream) {
  TDims D;
  HIG_REQUIRE(params, "hig_text_head_fwd: null params");
  HIG_TRY(check_tdims(dims, params, D));
  HIG_REQUIRE(clip_out && eot && xf_out && xf_proj && workspace, "hig_text_head_fwd: null argument");
  const TFwd w = tfwd_layout(D, training);
  float* ws = static_cast<float*>(workspace);
  hipStream_t st = hig_stream(stream);
  const int Lt = D.Lt, ff = D.ff;
  const int64_t M = D.M;
  const int64_t blk = w.total - w.layer0;  // inference: two half blocks
  const float* xin = clip_out;
  if (D.pre) {
    HIG_TRY(hig_gemm_launch(G(clip_out, D.W, 0, TP(params, HIG_T_PRE_W), D.W, 0, ws + w.x0, Lt, M, Lt, D.W)
                                .epi(HIG_EPI_BIAS, TP(params, HIG_T_PRE_B)).prec(D.prec).g, 1, nullptr, st));
    xin = ws + w.x0;
  }
  for (int l = 0; l < D.L; ++l) {
    float* lb = ws + w.layer0 + (training ? w.lstride * l : (l & 1) * (blk / 2));
    HIG_TRY(hig_gemm_launch(G(xin, Lt, 0, TPL(params, l, HIG_TL_IN_W), Lt, 0, lb + w.qkv, 3 * Lt, M, 3 * Lt, Lt)
                                .epi(HIG_EPI_BIAS, TPL(params, l, HIG_TL_IN_B)).prec(D.prec).g, 1, nullptr, st));
    HIG_TRY(hig_fullattn_fwd(lb + w.qkv, 3 * Lt, lb + w.qkv + Lt, lb + w.qkv + 2 * Lt, 3 * Lt, D.B, D.N, D.N, D.H, D.hd,
                             nullptr, lb + w.att, Lt, lb + w.lse, stream));
    HIG_TRY(hig_gemm_launch(G(lb + w.att, Lt, 0, TPL(params, l, HIG_TL_OUT_W), Lt, 0, lb + w.r1, Lt, M, Lt, Lt)
                                .epi(HIG_EPI_BIAS_RES, TPL(params, l, HIG_TL_OUT_B)).res(xin, Lt).prec(D.prec).g,
                            1, nullptr, st));
    HIG_TRY(hig_layernorm(lb + w.r1, Lt, M, Lt, TPL(params, l, HIG_TL_N1_W), TPL(params, l, HIG_TL_N1_B), lb + w.x1, Lt,
                          lb + w.st1, stream));
    HIG_TRY(hig_gemm_launch(G(lb + w.x1, Lt, 0, TPL(params, l, HIG_TL_FF1_W), Lt, 0, lb + w.f, ff, M, ff, Lt)
                                .epi(HIG_EPI_BIAS_GELU, TPL(params, l, HIG_TL_FF1_B))
                                .aux(training ? lb + w.z : nullptr, ff).prec(D.prec).g, 1, nullptr, st));
    HIG_TRY(hig_gemm_launch(G(lb + w.f, ff, 0, TPL(params, l, HIG_TL_FF2_W), ff, 0, lb + w.r2, Lt, M, Lt, ff)
                                .epi(HIG_EPI_BIAS_RES, TPL(params, l, HIG_TL_FF2_B)).res(lb + w.x1, Lt).prec(D.prec).g,
                            1, nullptr, st));
    HIG_TRY(hig_layernorm(lb + w.r2, Lt, M, Lt, TPL(params, l, HIG_TL_N2_W), TPL(params, l, HIG_TL_N2_B), lb + w.x2, Lt,
                          lb + w.st2, stream));
    xin = lb + w.x2;
  }
  HIG_TRY(hig_layernorm(xin, Lt, M, Lt, TP(params, HIG_T_LN_W), TP(params, HIG_T_LN_B), xf_out, Lt, ws + w.stf, stream));
  HIG_TRY(hig_gather_rows(xf_out, Lt, D.B, D.N, eot, Lt, ws + w.gath, Lt, stream));
  HIG_TRY(hig_gemm_launch(G(ws + w.gath, Lt, 0, TP(params, HIG_T_PROJ_W), Lt, 0, xf_proj, D.E, D.B, D.E, Lt)
                              .epi(HIG_EPI_BIAS, TP(params, HIG_T_PROJ_B)).g, 1, nullptr, st));
  return HIG_OK;
}

extern "C" int hig_text_head_bwd(const hig_text_dims* dims, const void* const* params, const float* clip_out,
                                 const int64_t* eot, const float* xf_out, const void* workspace,
                                 const float* dxf_out, const float* dxf_proj, void* const* grads, float* dclip,
                                 void* bwd_workspace, hig_stream_t stream) {
  (void)xf_out;
  TDims D;
  HIG_REQUIRE(params, "hig_text_head_bwd: null params");
  HIG_TRY(check_tdims(dims, params, D));
  HIG_REQUIRE(clip_out && eot && workspace && grads && bwd_workspace, "hig_text_head_bwd: null argument");
  const TFwd w = tfwd_layout(D, 1);
  const TBwd bw = tbwd_layout(D);
  const float* ws = static_cast<const float*>(workspace);
  float* b = static_cast<float*>(bwd_workspace);
  hipStream_t st = hig_stream(stream);
  const int Lt = D.Lt, ff = D.ff, E = D.E;
  const int64_t M = D.M;
  float* slabs = b + bw.slabs;
  float* colp = b + bw.colpart;
  float* lnp = b + bw.lnpart;
  float* wT = b + bw.wT;

  auto wgrad = [&](G gd) -> int {
    const int s = wgrad_splits(gd.g.I, gd.g.J, gd.g.R, bw.slab_floats, gd.g.prec);
    return hig_gemm_launch(gd.g, s, slabs, st);
  };
  // dW[n][k] = sum_m dC[m][n] * act[m][k]; bf16 product modes transpose both operands first
  // dbias = column sums of dC (the bias gradient): from the wgrad GEMM itself in the exact-fp32 path
  auto wgrad_act = [&](const float* dC, int n_out, const float* act, int k_in, float* out, int64_t rows,
                       float* dbias) -> int {
    if (D.prec != HIG_PREC_F32 && rows % 32 == 0) {
      float* ta = b + bw.tA;
      float* tb = b + bw.tB;
      HIG_TRY(hig_colsum(dC, n_out, rows, n_out, dbias, colp, stream));
      HIG_TRY(hig_transpose(dC, n_out, (int)rows, n_out, ta, rows, nullptr, nullptr, nullptr, stream));
      HIG_TRY(hig_transpose(act, k_in, (int)rows, k_in, tb, rows, nullptr, nullptr, nullptr, stream));
      return wgrad(G(ta, rows, 0, tb, rows, 0, out, k_in, n_out, k_in, rows).prec(D.prec));
    }
    G gd(dC, n_out, 1, act, k_in, 1, out, k_in, n_out, k_in, rows);
    if (n_out % 4 == 0) gd.xsum(dbias);
    else HIG_TRY(hig_colsum(dC, n_out, rows, n_out, dbias, colp, stream));
    return wgrad(gd);
  };
  // dX = dC . W with W (out_f, in_f) transposed first, so both operands are reduce-contiguous
  auto dgrad = [&](const float* dC, const float* W, int out_f, int in_f, int64_t rows, float* dX, int epi,
                   const float* res, float* aux) -> int {
    HIG_TRY(hig_transpose(W, in_f, out_f, in_f, wT, out_f, nullptr, nullptr, nullptr, stream));
    G gd(dC, out_f, 0, wT, out_f, 0, dX, in_f, rows, in_f, out_f);
    gd.prec(D.prec);
    if (epi == HIG_EPI_RES) gd.epi(HIG_EPI_RES).res(res, in_f);
    if (epi == HIG_EPI_DGELU) gd.epi(HIG_EPI_DGELU).aux(aux, in_f);
    return hig_gemm_launch(gd.g, 1, nullptr, st);
  };
  auto colsum = [&](const float* src, int64_t ld, int64_t rows, int n, float* dst) -> int {
    return hig_colsum(src, ld, rows, n, dst, colp, stream);
  };

  float* dA = b + bw.dA;
  float* dB = b + bw.dB;
  float* dC = b + bw.dC;
  // ---- d(xf_out) total = upstream + scatter of d(text_proj input) at the EOT rows -------------
  if (dxf_out) {
    HIG_TRY(hig_copy_async(dA, dxf_out, (size_t)M * Lt * 4, st));
  } else {
    HIG_TRY(hig_zero_async(dA, (size_t)M * Lt * 4, st));
  }
  if (dxf_proj) {
    HIG_TRY(colsum(dxf_proj, E, D.B, E, TG(grads, HIG_T_PROJ_B)));
    HIG_TRY(hig_gemm_launch(G(dxf_proj, E, 1, ws + w.gath, Lt, 1, TG(grads, HIG_T_PROJ_W), Lt, E, Lt, D.B).g, 1, nullptr, st));
    HIG_TRY(hig_gemm_launch(G(dxf_proj, E, 0, TP(params, HIG_T_PROJ_W), Lt, 1, b + bw.dgath, Lt, D.B, Lt, E).g, 1, nullptr, st));
    HIG_TRY(hig_scatter_add_rows(b + bw.dgath, Lt, D.B, D.N, eot, Lt, dA, Lt, stream));
  } else {
    HIG_TRY(hig_zero_async(TG(grads, HIG_T_PROJ_B), (size_t)E * 4, st));
    HIG_TRY(hig_zero_async(TG(grads, HIG_T_PROJ_W), (size_t)E * Lt * 4, st));
  }
  // ---- text_ln ---------------------------------------------------------------------------
  const float* xL = ws + w.layer0 + w.lstride * (D.L - 1) + w.x2;
  HIG_TRY(hig_ln_bwd(dA, Lt, xL, Lt, ws + w.stf, TP(params, HIG_T_LN_W), TP(params, HIG_T_LN_B), nullptr, 0, 0, 0,
                     nullptr, 0, dB, Lt, M, Lt, D.N, TG(grads, HIG_T_LN_W), TG(grads, HIG_T_LN_B), nullptr, 0, lnp, stream));
  float* d = dB;      // d(x2 of layer l)
  float* t1 = dA;     // scratch (M, Lt)
  float* t2 = dC;
  for (int l = D.L - 1; l >= 0; --l) {
    const float* lb = ws + w.layer0 + w.lstride * l;
    const float* xin = l == 0 ? (D.pre ? ws + w.x0 : clip_out) : ws + w.layer0 + w.lstride * (l - 1) + w.x2;
    // norm2: x2 = LN(r2)
    HIG_TRY(hig_ln_bwd(d, Lt, lb + w.r2, Lt, lb + w.st2, TPL(params, l, HIG_TL_N2_W), TPL(params, l, HIG_TL_N2_B), nullptr,
                       0, 0, 0, nullptr, 0, t1, Lt, M, Lt, D.N, TGL(grads, l, HIG_TL_N2_W), TGL(grads, l, HIG_TL_N2_B),
                       nullptr, 0, lnp, stream));
    const float* dr2 = t1;
    // r2 = x1 + linear2(gelu(z)),  z = linear1(x1)
    HIG_TRY(wgrad_act(dr2, Lt, lb + w.f, ff, TGL(grads, l, HIG_TL_FF2_W), M, TGL(grads, l, HIG_TL_FF2_B)));
    HIG_TRY(dgrad(dr2, TPL(params, l, HIG_TL_FF2_W), Lt, ff, M, b + bw.dff, HIG_EPI_DGELU, nullptr,
                  const_cast<float*>(lb + w.z)));
    const float* dz = b + bw.dff;
    HIG_TRY(wgrad_act(dz, ff, lb + w.x1, Lt, TGL(grads, l, HIG_TL_FF1_W), M, TGL(grads, l, HIG_TL_FF1_B)));
    HIG_TRY(dgrad(dz, TPL(params, l, HIG_TL_FF1_W), ff, Lt, M, t2, HIG_EPI_RES, dr2, nullptr));  // t2 = d(x1)
    // norm1: x1 = LN(r1)
    HIG_TRY(hig_ln_bwd(t2, Lt, lb + w.r1, Lt, lb + w.st1, TPL(params, l, HIG_TL_N1_W), TPL(params, l, HIG_TL_N1_B), nullptr,
                       0, 0, 0, nullptr, 0, t1, Lt, M, Lt, D.N, TGL(grads, l, HIG_TL_N1_W), TGL(grads, l, HIG_TL_N1_B),
                       nullptr, 0, lnp, stream));
    const float* dr1 = t1;
    // r1 = xin + out_proj(att)
    HIG_TRY(wgrad_act(dr1, Lt, lb + w.att, Lt, TGL(grads, l, HIG_TL_OUT_W), M, TGL(grads, l, HIG_TL_OUT_B)));
    HIG_TRY(dgrad(dr1, TPL(params, l, HIG_TL_OUT_W), Lt, Lt, M, t2, HIG_EPI_NONE, nullptr, nullptr));  // t2 = d(att)
    float* dqkv = b + bw.dqkv;
    HIG_TRY(hig_fullattn_bwd(t2, Lt, lb + w.att, Lt, lb + w.qkv, 3 * Lt, lb + w.qkv + Lt, lb + w.qkv + 2 * Lt, 3 * Lt, D.B,
                             D.N, D.N, D.H, D.hd, nullptr, lb + w.lse, b + bw.delta, dqkv, 3 * Lt, dqkv + Lt,
                             dqkv + 2 * Lt, 3 * Lt, stream));
    HIG_TRY(wgrad_act(dqkv, 3 * Lt, xin, Lt, TGL(grads, l, HIG_TL_IN_W), M, TGL(grads, l, HIG_TL_IN_B)));
    HIG_TRY(dgrad(dqkv, TPL(params, l, HIG_TL_IN_W), 3 * Lt, Lt, M, d, HIG_EPI_RES, dr1, nullptr));   // d = d(xin)
  }
  // ---- text_pre_proj -----------------------------------------------------------------------
  if (D.pre) {
    HIG_TRY(wgrad_act(d, Lt, clip_out, D.W, TG(grads, HIG_T_PRE_W), M, TG(grads, HIG_T_PRE_B)));
    if (dclip) HIG_TRY(dgrad(d, TP(params, HIG_T_PRE_W), Lt, D.W, M, dclip, HIG_EPI_NONE, nullptr, nullptr));
  } else if (dclip) {
    HIG_TRY(hig_copy_async(dclip, d, (size_t)M * Lt * 4, st));
  }
  return HIG_OK;
}
