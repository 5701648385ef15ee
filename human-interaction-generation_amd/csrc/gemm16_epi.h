// Epilogue arithmetic shared by the bf16-storage GEMM kernels (gemm_bf16.hip, gemm_ws16.hip).
#pragma once
#include "hig_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Exact-erf GELU (nn.GELU(), transformer.py:160) for a result that is rounded to bf16 (2^-9) right away:
//     gelu(x) = x Phi(x) = max(x, 0) - |x| Q(|x|),      Q(t) = 1 - Phi(t) = erfc(t / sqrt 2) / 2 = exp2(p(t))
// with p a degree-6 fit of log2 Q on [0, 6.5] (|error| <= 2.5e-4, i.e. Q to 1.7e-4 RELATIVE -- also in the tails, where an
// erf with an absolute error loses the small negative outputs; beyond 6.5 Q < 5e-11 and t is clamped).  Ten vector
// instructions, ONE of them quarter-rate (v_exp_f32): 52 cycles of the SIMD's vector ALU per 64 outputs against 76 for the
// Abramowitz-Stegun 7.1.26 form used until round 5 (rcp + exp + 11 others) -- the GELU epilogue of the weight-stationary GEMMs
// is bound by exactly that ALU (profiles/r05_notes.md section 7).  Against gelu evaluated in double and rounded to bf16, on a
// uniform grid over [-8, 8]: 9.7 % of the rounded outputs differ by one ulp (the old form: 15.5 %).
constexpr float HIG_GELU_Q0 = -1.00025339f, HIG_GELU_Q1 = -1.14901738f, HIG_GELU_Q2 = -0.463114774f, HIG_GELU_Q3 = -0.0499779109f,
                HIG_GELU_Q4 = 0.00682028804f, HIG_GELU_Q5 = -0.000556051072f, HIG_GELU_Q6 = 1.99234738e-05f, HIG_GELU_TMAX = 6.5f;
__device__ __forceinline__ float gelu_bf16(float x) {
  const float t = fminf(fabsf(x), HIG_GELU_TMAX);
  float p = fmaf(t, HIG_GELU_Q6, HIG_GELU_Q5);
  p = fmaf(t, p, HIG_GELU_Q4);
  p = fmaf(t, p, HIG_GELU_Q3);
  p = fmaf(t, p, HIG_GELU_Q2);
  p = fmaf(t, p, HIG_GELU_Q1);
  p = fmaf(t, p, HIG_GELU_Q0);
  return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}

// The same on a pair of outputs (gemm_ws16.hip: packed fp32 arithmetic, the exponential per element): the same operations in
// the same order, so the same bits as gelu_bf16.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_bf16_pk(f32x2 x) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 t = {fminf(ax[0], HIG_GELU_TMAX), fminf(ax[1], HIG_GELU_TMAX)};
  f32x2 p = __builtin_elementwise_fma(t, f32x2{HIG_GELU_Q6, HIG_GELU_Q6}, f32x2{HIG_GELU_Q5, HIG_GELU_Q5});
  p = __builtin_elementwise_fma(t, p, f32x2{HIG_GELU_Q4, HIG_GELU_Q4});
  p = __builtin_elementwise_fma(t, p, f32x2{HIG_GELU_Q3, HIG_GELU_Q3});
  p = __builtin_elementwise_fma(t, p, f32x2{HIG_GELU_Q2, HIG_GELU_Q2});
  p = __builtin_elementwise_fma(t, p, f32x2{HIG_GELU_Q1, HIG_GELU_Q1});
  p = __builtin_elementwise_fma(t, p, f32x2{HIG_GELU_Q0, HIG_GELU_Q0});
  const f32x2 q = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
  return __builtin_elementwise_fma(-ax, q, f32x2{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)});
}
__device__ __forceinline__ f32x2 silu_fast_pk(f32x2 x) {
  const f32x2 w = x * -1.4426950408889634f;
  const f32x2 den = f32x2{__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])} + 1.0f;
  return x * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
}
// Which operands an epilogue takes.  HIG_EPI_RES (out = res + acc) and HIG_EPI_DGELU (out = acc * gelu'(res): `res` holds
// the pre-activation z of FFN linear1) are the data-gradient epilogues of the bf16-storage training step.
__host__ __device__ constexpr bool epi_has_bias(int e) {
  return e == HIG_EPI_BIAS || e == HIG_EPI_BIAS_GELU || e == HIG_EPI_BIAS_RES || e == HIG_EPI_BIAS_SILU || e == HIG_EPI_BIAS_RES_SILU;
}
__host__ __device__ constexpr bool epi_has_res(int e) {
  return e == HIG_EPI_BIAS_RES || e == HIG_EPI_BIAS_RES_SILU || e == HIG_EPI_RES || e == HIG_EPI_DGELU;
}
// gelu'(z) = Phi(z) + z phi(z) with the same Abramowitz-Stegun erf as gelu_bf16 (its exp(-z^2 / 2) is phi's, too)
__device__ __forceinline__ float dgelu_bf16(float z) {
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(az, 0.3275911f * 0.70710678118654752440f, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float zz = az * (0.70710678118654752440f * 1.2011224087864498f);
  const float ex = __builtin_amdgcn_exp2f(-(zz * zz));             // exp(-z^2 / 2)
  const float erf_abs = fmaf(-p, ex, 1.0f);                        // erf(|z| / sqrt 2)
  return fmaf(z * 0.39894228040143267794f, ex, 0.5f + copysignf(0.5f * erf_abs, z));
}

template <int EPI>
__device__ __forceinline__ f32x2 epi_act_pk(f32x2 v) {
  if (EPI == HIG_EPI_BIAS_GELU) return gelu_bf16_pk(v);
  if (EPI == HIG_EPI_BIAS_SILU || EPI == HIG_EPI_BIAS_RES_SILU) return silu_fast_pk(v);
  return v;
}

template <int EPI>
__device__ __forceinline__ float epi_act(float v) {
  if (EPI == HIG_EPI_BIAS_GELU) return gelu_bf16(v);
  if (EPI == HIG_EPI_BIAS_SILU || EPI == HIG_EPI_BIAS_RES_SILU) return hig_silu_fast(v);
  return v;
}

// LayerNorm fold, producer side (gemm_ws16.hip, linattn16.hip): the statistics of one 128-column panel of an output row, from
// the 8 ROUNDED outputs each of 16 consecutive lanes holds (what the consumer will read):
//   s1 = sum x,   m2 = sum (x - s1 / 128)^2      -- two passes over the registers, so no E[x^2] - mean^2 cancellation: rows
// whose |mean| is hundreds of times their spread keep their variance (tests/test_gpu_bf16_storage.py).  The consumer merges the
// four panels of a row with hig_ln_merge4 below (pairwise update of Chan, Golub & LeVeque).  Every lane of the group
// returns the group's totals.
__device__ __forceinline__ void hig_panel_stats16(const bf16x8& v, float& s1, float& m2) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) s = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{v[2 * k], v[2 * k + 1]}, ones, s, false);
  s = row16_sum(s);
  const float mean = s * (1.0f / 128.0f);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float t = (float)v[k] - mean;
    q = fmaf(t, t, q);
  }
  q = row16_sum(q);
  s1 = s;
  m2 = q;
}
// consumer side: (mean, variance) of a K = 512 row from its four panel statistics p = (s1_0, m2_0, s1_1, m2_1), (s1_2, ...)
__device__ __forceinline__ void hig_ln_merge4(const f32x4& p0, const f32x4& p1, float& mean, float& var) {
  mean = (p0.x + p0.z + p1.x + p1.z) * (1.0f / 512.0f);
  const float d0 = fmaf(p0.x, 1.0f / 128.0f, -mean), d1 = fmaf(p0.z, 1.0f / 128.0f, -mean);
  const float d2 = fmaf(p1.x, 1.0f / 128.0f, -mean), d3 = fmaf(p1.z, 1.0f / 128.0f, -mean);
  const float between = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3)));
  var = fmaf(between, 128.0f, p0.y + p0.w + p1.y + p1.w) * (1.0f / 512.0f);
}
