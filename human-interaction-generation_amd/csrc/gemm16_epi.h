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

template <int EPI>
__device__ __forceinline__ float epi_act(float v) {
  if (EPI == HIG_EPI_BIAS_GELU) return gelu_bf16(v);
  if (EPI == HIG_EPI_BIAS_SILU || EPI == HIG_EPI_BIAS_RES_SILU) return hig_silu_fast(v);
  return v;
}
