// Epilogue arithmetic shared by the bf16-storage GEMM kernels (gemm_bf16.hip, gemm_ws16.hip).
#pragma once
#include "hig_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Exact-erf GELU (nn.GELU(), transformer.py:160) for a result that is rounded to bf16 (2^-9) right away: erf by
// Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7: one exp, one rcp, five FMAs) instead of the 30-instruction libm erff --
// the epilogue's VALU work was 10 of the 43 us of the FFN linear1 launch (profiles/r02_notes.md).
__device__ __forceinline__ float gelu_bf16(float x) {
  // 13 vector instructions (|.| and negation are operand modifiers): the epilogue of FFN linear1 is bound by the vector
  // unit, not by the matrix pipe (4 outputs per clock and CU leave 16 lane-operations per output)
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float zz = ax * (0.70710678118654752440f * 1.2011224087864498f);   // z sqrt(log2 e): exp(-z^2) = exp2(-zz^2)
  const float erf_abs = fmaf(-p, __builtin_amdgcn_exp2f(-(zz * zz)), 1.0f);
  const float h = 0.5f * x;
  return fmaf(fabsf(h), erf_abs, h);             // 0.5 x (1 + sign(x) erf|z|)
}

// The same on a pair of outputs with packed fp32 arithmetic (v_pk_fma / v_pk_mul_f32: two elements per issue slot): the
// epilogues of the weight-stationary kernel are bound by instruction issue, and of the 13 instructions above only the
// reciprocal and the exponential have no packed form -- 7 packed + 2 x 2 transcendental + 2 (|x|) per pair instead of 26.
// (exp(-z^2) from x^2 directly: no |x| needed there.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_bf16_pk(f32x2 x) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 den = __builtin_elementwise_fma(ax, f32x2{0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f}, f32x2{1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2 p = __builtin_elementwise_fma(t, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(t, p, f32x2{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(t, p, f32x2{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(t, p, f32x2{0.254829592f, 0.254829592f});
  p *= t;
  constexpr float K2 = -(0.70710678118654752440f * 1.2011224087864498f) * (0.70710678118654752440f * 1.2011224087864498f);   // -(z^2 log2 e) / x^2
  const f32x2 arg = (x * x) * K2;
  const f32x2 ex = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  const f32x2 erf_abs = __builtin_elementwise_fma(-p, ex, f32x2{1.0f, 1.0f});
  return __builtin_elementwise_fma(ax * 0.5f, erf_abs, x * 0.5f);      // 0.5 x (1 + sign(x) erf|z|)
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
