// Row kernels of the bf16-storage forward (hig_dims.storage == HIG_STORE_BF16): LayerNorm and the stylization front
//   out[m] = LN(x[m]) * gamma + beta                                           (transformer.py:108,144,146)
//   out[m] = silu( (LN(x[m]) * gamma + beta) * (1 + scale[b]) + shift[b] )     (transformer.py:81-85)
// with x bf16 (or fp32: the text embeddings arrive in fp32) and out bf16; statistics and arithmetic in fp32.
// One 64-lane wave per row (four consecutive rows per wave), 8 elements (16 bytes of bf16) per lane per sweep, the rows stay in
// registers between the statistics and the transform: x is read once, out written once -- HBM-bound at 2 x rows x n x 2 bytes.
#include "hig_common.h"
#include "gemm16_epi.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int WAVES = 4;

__device__ __forceinline__ void ld8(const __bf16* p, float (&v)[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// RPW consecutive rows per wave: the per-column vectors (gamma, beta, and with MOD the sample's scale / shift -- 8 KB of
// L2 reads per row when fetched per row, against 1 KB of row data) are loaded ONCE per wave and kept in registers, the
// RPW row loads are all in flight together.  With MOD a workgroup stays inside one sample (blockIdx.y), so the modulation
// vectors are wave-invariant; without it the rows are one run (gridDim.y == 1).  Arithmetic per element unchanged.
template <int NIT, bool MOD, typename TIN, int RPW>
__global__ __launch_bounds__(256) void ln16_kernel(const TIN* __restrict__ x, int64_t ldx, int64_t rows, int n,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ ss, int64_t ss_ld, int shift_off,
                                                   int rows_per_sample, __bf16* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t group0 = MOD ? (int64_t)blockIdx.y * rows_per_sample : 0;
  const int64_t group1 = MOD ? min(group0 + rows_per_sample, rows) : rows;     // this group's rows: [group0, group1)
  const int64_t row0 = group0 + ((int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * RPW;
  if (row0 >= group1) return;
  float v[RPW][NIT][8];
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const TIN* xr = x + min(row0 + k, group1 - 1) * ldx;        // (rows beyond the group: a valid row, result not stored)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 8 * lane + 512 * it;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[k][it][e] = 0.f;
      if (c < n) ld8(xr + c, v[k][it]);
    }
  }
  float g[NIT][8], b[NIT][8], sc[MOD ? NIT : 1][8], sh[MOD ? NIT : 1][8];
  const float* ssrow = MOD ? ss + (int64_t)blockIdx.y * ss_ld : nullptr;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = 8 * lane + 512 * it;
    if (c < n) {
      ld8(gamma + c, g[it]);
      ld8(beta + c, b[it]);
      if (MOD) {
        ld8(ssrow + c, sc[it]);
        ld8(ssrow + shift_off + c, sh[it]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      s += ((v[k][it][0] + v[k][it][1]) + (v[k][it][2] + v[k][it][3])) + ((v[k][it][4] + v[k][it][5]) + (v[k][it][6] + v[k][it][7]));
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (8 * lane + 512 * it < n) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[k][it][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)n + 1e-5f);
    if (row0 + k >= group1) continue;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 8 * lane + 512 * it;
      if (c < n) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[k][it][e] - mean) * rstd * g[it][e] + b[it][e];
        if (MOD) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = hig_silu_fast(o[e] * (1.0f + sc[it][e]) + sh[it][e]);
        }
        *reinterpret_cast<bf16x8*>(out + (row0 + k) * ldo + c) =
            bf16x8{(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3], (__bf16)o[4], (__bf16)o[5], (__bf16)o[6], (__bf16)o[7]};
      }
    }
  }
}

template <bool MOD, typename TIN, int RPW>
int launch_ln16_rpw(const TIN* x, int64_t ldx, int64_t rows, int n, const float* gamma, const float* beta, const float* ss,
                    int64_t ss_ld, int shift_off, int rows_per_sample, __bf16* out, int64_t ldo, hipStream_t st) {
  const int64_t group = MOD ? rows_per_sample : rows;            // rows that share the per-column vectors
  const int64_t ngroups = MOD ? (rows + rows_per_sample - 1) / rows_per_sample : 1;
  if (ngroups > 65535) return hig_set_error(HIG_EUNSUPPORTED, "hig_ln_bf16: more than 65535 samples in one launch");
  const dim3 grid((unsigned)((group + WAVES * RPW - 1) / (WAVES * RPW)), (unsigned)ngroups);
  if (n <= 512)
    hipLaunchKernelGGL((ln16_kernel<1, MOD, TIN, RPW>), grid, dim3(256), 0, st, x, ldx, rows, n, gamma, beta, ss, ss_ld, shift_off,
                       rows_per_sample, out, ldo);
  else
    hipLaunchKernelGGL((ln16_kernel<2, MOD, TIN, RPW>), grid, dim3(256), 0, st, x, ldx, rows, n, gamma, beta, ss, ss_ld, shift_off,
                       rows_per_sample, out, ldo);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// Rows per wave by size (measured, stylization front at d = 512): four rows per wave keep the per-column vectors in
// registers -- 55 -> ~40 us at 100 352 rows -- but leave a 6 272-row launch with too few waves (5.9 -> 7.4 us): one row per
// wave below 32 768 rows.
template <bool MOD, typename TIN>
int launch_ln16(const TIN* x, int64_t ldx, int64_t rows, int n, const float* gamma, const float* beta, const float* ss,
                int64_t ss_ld, int shift_off, int rows_per_sample, __bf16* out, int64_t ldo, hipStream_t st) {
  if (rows >= 32768) return launch_ln16_rpw<MOD, TIN, 4>(x, ldx, rows, n, gamma, beta, ss, ss_ld, shift_off, rows_per_sample, out, ldo, st);
  return launch_ln16_rpw<MOD, TIN, 1>(x, ldx, rows, n, gamma, beta, ss, ss_ld, shift_off, rows_per_sample, out, ldo, st);
}

}  // namespace

extern "C" int hig_ln_bf16(const void* x, int32_t x_f32, int64_t ldx, int64_t rows, int32_t n, const float* gamma,
                           const float* beta, const float* ss, int64_t ss_ld, int32_t ss_shift_off,
                           int32_t rows_per_sample, void* out, int64_t ldo, hig_stream_t stream) {
  HIG_REQUIRE(x && gamma && beta && out && rows >= 0 && n > 0, "hig_ln_bf16: bad arguments");
  if (rows == 0) return HIG_OK;
  HIG_REQUIRE(n % 8 == 0 && n <= 1024 && ldx % 8 == 0 && ldo % 8 == 0,
              "hig_ln_bf16: n, ldx, ldo must be multiples of 8, n <= 1024");
  HIG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gamma) |
                reinterpret_cast<uintptr_t>(beta)) & 15) == 0,
              "hig_ln_bf16: buffers must be 16-byte aligned");
  if (ss)
    HIG_REQUIRE(rows_per_sample > 0 && ss_ld % 4 == 0 && ss_shift_off % 4 == 0 && (reinterpret_cast<uintptr_t>(ss) & 15) == 0,
                "hig_ln_bf16: bad modulation arguments");
  hipStream_t st = hig_stream(stream);
  __bf16* o = static_cast<__bf16*>(out);
  if (x_f32) {
    const float* xf = static_cast<const float*>(x);
    return ss ? launch_ln16<true, float>(xf, ldx, rows, n, gamma, beta, ss, ss_ld, ss_shift_off, rows_per_sample, o, ldo, st)
              : launch_ln16<false, float>(xf, ldx, rows, n, gamma, beta, ss, ss_ld, ss_shift_off, rows_per_sample, o, ldo, st);
  }
  const __bf16* xb = static_cast<const __bf16*>(x);
  return ss ? launch_ln16<true, __bf16>(xb, ldx, rows, n, gamma, beta, ss, ss_ld, ss_shift_off, rows_per_sample, o, ldo, st)
            : launch_ln16<false, __bf16>(xb, ldx, rows, n, gamma, beta, ss, ss_ld, ss_shift_off, rows_per_sample, o, ldo, st);
}

// ---- bf16-storage training step: elementwise helpers ---------------------------------------------------------------------
namespace {
// erf GELU on bf16 rows (nn.GELU(), transformer.py:160,168; gelu_bf16 of gemm16_epi.h: the arithmetic of the GEMM epilogues): f = gelu(z).  The training forward keeps BOTH z (the
// backward's gelu'(z)) and f (operand of linear2); the weight-stationary GEMM writes z, this pass derives f.
__global__ __launch_bounds__(256) void gelu16_kernel(const __bf16* __restrict__ z, __bf16* __restrict__ f, int64_t n8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(z)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)gelu_bf16((float)v[e]);
    reinterpret_cast<bf16x8*>(f)[i] = o;
  }
}
__global__ __launch_bounds__(256) void cast_f32_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, int64_t n8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(src)[i];
    reinterpret_cast<f32x4*>(dst)[2 * i] = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    reinterpret_cast<f32x4*>(dst)[2 * i + 1] = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
  }
}
}  // namespace

extern "C" int hig_gelu_bf16(const void* z, void* f, int64_t n, hig_stream_t stream) {
  HIG_REQUIRE(z && f && n >= 0 && n % 8 == 0 && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(f)) & 15) == 0,
              "hig_gelu_bf16: n %% 8 == 0, 16-byte aligned buffers");
  if (n == 0) return HIG_OK;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gelu16_kernel, dim3((unsigned)blocks), dim3(256), 0, hig_stream(stream), static_cast<const __bf16*>(z),
                     static_cast<__bf16*>(f), n / 8);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
// dst[r][c] = bf16(src[r][c]) for c < cols, 0 for cols <= c < ldd: fp32 rows -> bf16 rows padded to a width the bf16 matrix
// kernels take (the F = 150-wide motion features / their gradients at the edges of the bf16-storage training step).
// One thread per 8 output columns.
__global__ __launch_bounds__(256) void cast_pad16_kernel(const float* __restrict__ src, int64_t lds, int64_t rows, int cols,
                                                         __bf16* __restrict__ dst, int64_t ldd) {
  const int64_t p8 = ldd / 8, n = rows * p8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / p8;
    const int c0 = (int)(i % p8) * 8;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(c0 + e < cols ? src[r * lds + c0 + e] : 0.f);
    *reinterpret_cast<bf16x8*>(dst + r * ldd + c0) = o;
  }
}
extern "C" int hig_cast_pad_bf16(const float* src, int64_t ld_src, int64_t rows, int32_t cols, void* dst, int64_t ld_dst,
                                 hig_stream_t stream) {
  HIG_REQUIRE(src && dst && rows >= 0 && cols > 0 && ld_src >= cols && ld_dst >= cols && ld_dst % 8 == 0 &&
                  (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
              "hig_cast_pad_bf16: ld_dst must be a multiple of 8 and >= cols, dst 16-byte aligned");
  if (rows == 0) return HIG_OK;
  int64_t blocks = (rows * (ld_dst / 8) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_pad16_kernel, dim3((unsigned)blocks), dim3(256), 0, hig_stream(stream), src, ld_src, rows, cols,
                     static_cast<__bf16*>(dst), ld_dst);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_cast_f32(const void* src, float* dst, int64_t n, hig_stream_t stream) {
  HIG_REQUIRE(src && dst && n >= 0 && n % 8 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0,
              "hig_cast_f32: n %% 8 == 0, 16-byte aligned buffers");
  if (n == 0) return HIG_OK;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cast_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, hig_stream(stream), static_cast<const __bf16*>(src), dst, n / 8);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
