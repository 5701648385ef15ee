// Evaluator feature extraction: forward of the reference's two scoring classifiers,
// MotionEncoder (codes/models/interaction_transformer.py:641-741) and MotionConsistencyEvalModel
// (:743-829), as they are called by EvaluatorModelWrapper.get_motion_embeddings
// (codes/datasets/evaluator.py:479-493) -- inference only.
//
//   h[b] = [cls_input?] ++ embed(x1[b]) ++ embed(x2[b])            (S = cls + 2T tokens)
//   embed(x)[0] = joint_embed2(x[0, :4]);  embed(x)[t>=1] = joint_embed1(x[t]) + sequence_embedding[t-1]
//   L x post-norm nn.TransformerEncoderLayer (gelu, dropout 0) with key padding mask
//        pad[b][cls + p*T + t] = (t >= length[b])
//   MotionEncoder: o = out2(h[token 0 of a person]) / out1(h[other tokens]);
//                  feature = sum_valid o / #valid;  logits = fin_proj(feature)
//   Consistency:   logits = cls_output(h[cls token])
// Host code + two small row kernels; the GEMMs, LayerNorm and attention are the shared ones.
#include "hig_common.h"
#include "hig_host.h"

namespace {

struct EDims {
  int B, T, F, d, H, ff, L, C, cls, hd, prec, S;
  int64_t M;
};

int check_edims(const hig_eval_dims* p, EDims& D) {
  HIG_REQUIRE(p, "null eval dims");
  D.B = p->B; D.T = p->T; D.F = p->F; D.d = p->d; D.H = p->H; D.ff = p->ff; D.L = p->L; D.C = p->C;
  D.cls = p->cls ? 1 : 0;
  HIG_REQUIRE(D.B > 0 && D.T > 1 && D.F >= 4 && D.d > 0 && D.H > 0 && D.ff > 0 && D.L > 0 && D.C > 0,
              "hig_eval_dims: every extent must be positive (T >= 2, F >= 4)");
  HIG_REQUIRE(D.d % D.H == 0, "hig_eval_dims: d=%d not divisible by H=%d", D.d, D.H);
  D.hd = D.d / D.H;
  if (!(D.hd == 8 || D.hd == 16 || D.hd == 32 || D.hd == 64 || D.hd == 128))
    return hig_set_error(HIG_EUNSUPPORTED, "hig eval encoder: head dim %d not in {8,16,32,64,128}", D.hd);
  HIG_REQUIRE(D.d % 4 == 0 && D.ff % 4 == 0 && D.d <= 1024, "hig_eval_dims: d, ff must be multiples of 4 and d <= 1024");
  if (p->prec != HIG_PREC_F32 && p->prec != HIG_PREC_BF16X3 && p->prec != HIG_PREC_BF16)
    return hig_set_error(HIG_EINVAL, "hig: unknown prec=%d", p->prec);
  D.prec = p->prec;
  D.S = D.cls + 2 * D.T;
  D.M = (int64_t)D.B * D.S;
  return HIG_OK;
}

struct EWs {
  int64_t emb, h, kpad, qkv, lse, att, r1, st, x1, f, r2, xa, xb, feat, total;
};
EWs ews_layout(const EDims& D) {
  EWs w;
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += al(n); return r; };
  w.emb = take((int64_t)2 * D.B * D.T * D.d);   // [person][b][t][d]
  w.h = take(D.M * D.d);
  w.kpad = take((D.M + 3) / 4);                 // bytes
  w.qkv = take(D.M * 3 * D.d);
  w.lse = take((int64_t)D.B * D.H * D.S);
  w.att = take(D.M * D.d);
  w.r1 = take(D.M * D.d);
  w.st = take(D.M * 2);
  w.x1 = take(D.M * D.d);
  w.f = take(D.M * D.ff);
  w.r2 = take(D.M * D.d);
  w.xa = take(D.M * D.d);
  w.xb = take(D.M * D.d);
  w.feat = take((int64_t)D.B * D.d);
  w.total = o;
  return w;
}

inline const float* EP(const void* const* t, int idx) { return static_cast<const float*>(t[idx]); }
inline const float* EPL(const void* const* t, int l, int idx) {
  return static_cast<const float*>(t[HIG_EV_NGLOBAL + l * HIG_TL_NLAYER + idx]);
}

// Row (b, s) of h: the [cls] vector, or row (p, b, t) of the per-person embeddings; also the key padding byte.
__global__ __launch_bounds__(128) void assemble_kernel(const float* __restrict__ emb, const float* __restrict__ cls_in,
                                                       const int64_t* __restrict__ length, float* __restrict__ h,
                                                       uint8_t* __restrict__ kpad, int B, int T, int d, int cls) {
  const int S = cls + 2 * T;
  const int64_t row = blockIdx.x;
  const int b = (int)(row / S), s = (int)(row % S);
  const float* src;
  bool pad = false;
  if (s < cls) {
    src = cls_in;
  } else {
    const int p = (s - cls) / T, t = (s - cls) % T;
    src = emb + (((int64_t)p * B + b) * T + t) * d;
    pad = t >= length[b];
  }
  float* dst = h + row * d;
  for (int c = 4 * threadIdx.x; c < d; c += 4 * 128)
    *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
  if (threadIdx.x == 0) kpad[row] = pad ? 1 : 0;
}

// feature[b][c] = sum over both persons' tokens t < length[b] of o[b][cls + p*T + t][c], divided by their count
// (interaction_transformer.py:739-740: (cat([output1, output2]) * src_mask).sum(1) / src_mask.sum(1)).
__global__ __launch_bounds__(256) void masked_mean_kernel(const float* __restrict__ o, const int64_t* __restrict__ length,
                                                          float* __restrict__ feat, int T, int cls, int d) {
  const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= d) return;
  const int S = cls + 2 * T;
  int64_t len = length[b];
  if (len > T) len = T;
  if (len < 0) len = 0;
  const float* ob = o + ((int64_t)b * S + cls) * d + c;
  float acc = 0.f;
  for (int p = 0; p < 2; ++p)
    for (int t = 0; t < (int)len; ++t) acc += ob[((int64_t)p * T + t) * d];
  feat[(int64_t)b * d + c] = acc / (float)(2 * len);
}

}  // namespace

extern "C" int64_t hig_eval_encoder_workspace_bytes(const hig_eval_dims* dims) {
  EDims D;
  if (check_edims(dims, D) != HIG_OK) return -1;
  return ews_layout(D).total * 4;
}

extern "C" int hig_eval_encoder_fwd(const hig_eval_dims* dims, const void* const* params, const float* x1,
                                    const float* x2, const int64_t* length, float* logits, float* feature,
                                    void* workspace, hig_stream_t stream) {
  EDims D;
  HIG_TRY(check_edims(dims, D));
  HIG_REQUIRE(params && x1 && x2 && length && logits && workspace, "hig_eval_encoder_fwd: null argument");
  HIG_REQUIRE(EP(params, HIG_EV_SEQ_EMB) && EP(params, HIG_EV_JOINT1_W) && EP(params, HIG_EV_JOINT2_W) &&
                  EP(params, HIG_EV_HEAD_W),
              "hig_eval_encoder_fwd: missing embedding / head parameters");
  if (D.cls)
    HIG_REQUIRE(EP(params, HIG_EV_CLS_IN), "hig_eval_encoder_fwd: consistency model needs cls_input");
  else
    HIG_REQUIRE(EP(params, HIG_EV_OUT1_W) && EP(params, HIG_EV_OUT2_W), "hig_eval_encoder_fwd: MotionEncoder needs out1 / out2");
  const EWs w = ews_layout(D);
  float* ws = static_cast<float*>(workspace);
  hipStream_t st = hig_stream(stream);
  const int d = D.d, ff = D.ff, T = D.T, S = D.S;
  const int64_t M = D.M, BT = (int64_t)D.B * T;
  uint8_t* kpad = reinterpret_cast<uint8_t*>(ws + w.kpad);

  // per-person embeddings: every row through joint_embed1 (+ sequence_embedding[t-1]), then the init-pose rows
  // (t = 0) overwritten with joint_embed2 of their first 4 features (:717-720 / :814-817)
  const float* xs[2] = {x1, x2};
  for (int p = 0; p < 2; ++p) {
    float* e = ws + w.emb + p * BT * d;
    G ge(xs[p], D.F, 0, EP(params, HIG_EV_JOINT1_W), D.F, 0, e, d, BT, d, D.F);
    ge.epi(HIG_EPI_BIAS_POS, EP(params, HIG_EV_JOINT1_B)).pos(EP(params, HIG_EV_SEQ_EMB), d, T).prec(D.prec);
    ge.g.pos_shift = 1;
    HIG_TRY(hig_gemm_launch(ge.g, 1, nullptr, st));
    HIG_TRY(hig_gemm_launch(G(xs[p], (int64_t)T * D.F, 0, EP(params, HIG_EV_JOINT2_W), 4, 0, e, (int64_t)T * d, D.B, d, 4)
                                .epi(HIG_EPI_BIAS, EP(params, HIG_EV_JOINT2_B)).g, 1, nullptr, st));
  }
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)M), dim3(128), 0, st, ws + w.emb, EP(params, HIG_EV_CLS_IN), length,
                     ws + w.h, kpad, D.B, T, d, D.cls);
  HIG_CHECK_LAUNCH();

  const float* xin = ws + w.h;
  for (int l = 0; l < D.L; ++l) {
    float* xout = ws + ((l & 1) ? w.xb : w.xa);
    HIG_TRY(hig_gemm_launch(G(xin, d, 0, EPL(params, l, HIG_TL_IN_W), d, 0, ws + w.qkv, 3 * d, M, 3 * d, d)
                                .epi(HIG_EPI_BIAS, EPL(params, l, HIG_TL_IN_B)).prec(D.prec).g, 1, nullptr, st));
    HIG_TRY(hig_fullattn_fwd_kpad(ws + w.qkv, 3 * d, ws + w.qkv + d, ws + w.qkv + 2 * d, 3 * d, D.B, S, S, D.H, D.hd,
                                  nullptr, kpad, ws + w.att, d, ws + w.lse, stream));
    HIG_TRY(hig_gemm_launch(G(ws + w.att, d, 0, EPL(params, l, HIG_TL_OUT_W), d, 0, ws + w.r1, d, M, d, d)
                                .epi(HIG_EPI_BIAS_RES, EPL(params, l, HIG_TL_OUT_B)).res(xin, d).prec(D.prec).g,
                            1, nullptr, st));
    HIG_TRY(hig_layernorm(ws + w.r1, d, M, d, EPL(params, l, HIG_TL_N1_W), EPL(params, l, HIG_TL_N1_B), ws + w.x1, d,
                          ws + w.st, stream));
    HIG_TRY(hig_gemm_launch(G(ws + w.x1, d, 0, EPL(params, l, HIG_TL_FF1_W), d, 0, ws + w.f, ff, M, ff, d)
                                .epi(HIG_EPI_BIAS_GELU, EPL(params, l, HIG_TL_FF1_B)).prec(D.prec).g, 1, nullptr, st));
    HIG_TRY(hig_gemm_launch(G(ws + w.f, ff, 0, EPL(params, l, HIG_TL_FF2_W), ff, 0, ws + w.r2, d, M, d, ff)
                                .epi(HIG_EPI_BIAS_RES, EPL(params, l, HIG_TL_FF2_B)).res(ws + w.x1, d).prec(D.prec).g,
                            1, nullptr, st));
    HIG_TRY(hig_layernorm(ws + w.r2, d, M, d, EPL(params, l, HIG_TL_N2_W), EPL(params, l, HIG_TL_N2_B), xout, d,
                          ws + w.st, stream));
    xin = xout;
  }

  if (D.cls) {  // logits from the [cls] token's row of each pair
    HIG_TRY(hig_gemm_launch(G(xin, (int64_t)S * d, 0, EP(params, HIG_EV_HEAD_W), d, 0, logits, D.C, D.B, D.C, d)
                                .epi(HIG_EPI_BIAS, EP(params, HIG_EV_HEAD_B)).g, 1, nullptr, st));
    return HIG_OK;
  }
  float* o = ws + w.r1;
  HIG_TRY(hig_gemm_launch(G(xin, d, 0, EP(params, HIG_EV_OUT1_W), d, 0, o, d, M, d, d)
                              .epi(HIG_EPI_BIAS, EP(params, HIG_EV_OUT1_B)).prec(D.prec).g, 1, nullptr, st));
  for (int p = 0; p < 2; ++p)
    HIG_TRY(hig_gemm_launch(G(xin + (int64_t)p * T * d, (int64_t)S * d, 0, EP(params, HIG_EV_OUT2_W), d, 0,
                              o + (int64_t)p * T * d, (int64_t)S * d, D.B, d, d)
                                .epi(HIG_EPI_BIAS, EP(params, HIG_EV_OUT2_B)).g, 1, nullptr, st));
  float* feat = feature ? feature : ws + w.feat;
  hipLaunchKernelGGL(masked_mean_kernel, dim3(D.B, (d + 255) / 256), dim3(256), 0, st, o, length, feat, T, D.cls, d);
  HIG_CHECK_LAUNCH();
  HIG_TRY(hig_gemm_launch(G(feat, d, 0, EP(params, HIG_EV_HEAD_W), d, 0, logits, D.C, D.B, D.C, d)
                              .epi(HIG_EPI_BIAS, EP(params, HIG_EV_HEAD_B)).g, 1, nullptr, st));
  return HIG_OK;
}
