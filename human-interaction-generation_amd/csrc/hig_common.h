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

// Cross-lane reductions by DPP (data-parallel primitives: the operand of a VALU instruction comes from another lane of the
// same 16-lane row, ~one instruction's latency) instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32 -- an LDS-pipe
// round trip of ~100 cycles per step, six dependent steps per 64-lane sum: the row kernels (one wave per row, two to four
// dependent reductions per row) spent most of their time there.
//   row16: butterfly inside each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror): every lane of the row
//          ends with the row's total;
//   wave : then the four row totals by row_bcast:15 (rows 1, 3 += lane 15 of the row before) and row_bcast:31 (rows 2, 3 +=
//          lane 31): lane 63 holds the total, v_readlane_b32 hands it to every lane as a scalar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float hig_dpp(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += hig_dpp<0xB1, 0xf>(0.f, v);     // quad_perm [1,0,3,2]
  v += hig_dpp<0x4E, 0xf>(0.f, v);     // quad_perm [2,3,0,1]
  v += hig_dpp<0x141, 0xf>(0.f, v);    // row_half_mirror
  v += hig_dpp<0x140, 0xf>(0.f, v);    // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, hig_dpp<0xB1, 0xf>(v, v));
  v = fmaxf(v, hig_dpp<0x4E, 0xf>(v, v));
  v = fmaxf(v, hig_dpp<0x141, 0xf>(v, v));
  v = fmaxf(v, hig_dpp<0x140, 0xf>(v, v));
  return v;
}
// PRECONDITION of wave_sum / wave_max: all 64 lanes of the wave are active (the row_bcast steps and the v_readlane of lane 63
// read lanes that an EXEC mask would have left unwritten; the __shfl_xor form they replaced saw zeros there).  Every caller
// runs whole waves; a kernel that calls them under a divergent branch or with a block size that is not a multiple of 64 is wrong.
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += hig_dpp<0x142, 0xa>(0.f, v);    // row_bcast:15 into rows 1 and 3
  v += hig_dpp<0x143, 0xc>(0.f, v);    // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, hig_dpp<0x142, 0xa>(v, v));
  v = fmaxf(v, hig_dpp<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
