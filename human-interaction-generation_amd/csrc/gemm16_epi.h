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
