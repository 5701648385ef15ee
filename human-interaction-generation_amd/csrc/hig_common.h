// Internal helpers shared by the HIP translation units of libhig.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hig.h"

int hig_set_error(int code, const char* fmt, ...);

#define HIG_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return hig_set_error(HIG_EINVAL, __VA_ARGS__);  \
  } while (0)

#define HIG_CHECK_LAUNCH()                                                                   \
  do {                                                                                       \
    hipError_t e__ = hipGetLastError();                                                      \
    if (e__ != hipSuccess)                                                                   \
      return hig_set_error(HIG_EHIP, "%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
  } while (0)

#define HIG_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != HIG_OK) return rc__; \
  } while (0)

static inline hipStream_t hig_stream(hig_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float hig_silu(float x) { return x / (1.0f + __expf(-x)); }
// d/dx [x * sigmoid(x)] = s * (1 + x * (1 - s))
__device__ __forceinline__ float hig_dsilu(float x) {
  float s = 1.0f / (1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}
// bf16-storage kernels: the result is rounded to 8 significant bits right away, so the 1-ulp hardware reciprocal
// (v_rcp_f32) replaces the correctly rounded division (v_div_scale / v_div_fmas / v_div_fixup: ~10 instructions).
__device__ __forceinline__ float hig_silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// Element order of the transposed bf16 context matrices At16 (At[l][c] = bf16(A[c][l]); written by the context kernels,
// read by apply_sty16_kernel, linattn16.hip): "fragment-major" -- the 32 (l) x 16 (c) block that ONE lane-wise 16-byte
// load of a v_mfma_f32_32x32x16_bf16 row operand needs is one contiguous 1-KiB run, lane (l % 32) + 32 ((c % 16) / 8)
// holding c % 8 = 0 .. 7: the consumer fetches its operands global -> registers in whole coalesced KiB, no LDS staging.
__host__ __device__ __forceinline__ int hig_at16_offset(int hd, int l, int c) {
  return ((((l >> 5) * (hd >> 4) + (c >> 4)) * 64 + (l & 31) + 32 * ((c >> 3) & 1)) << 3) + (c & 7);
}
__device__ __forceinline__ float hig_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float hig_dgelu(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) +
         x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
