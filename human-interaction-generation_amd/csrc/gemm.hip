// Fused MFMA GEMM for gfx950 (MI355X):  C[i][j] = epi( sum_r xf(X)[i][r] * Y[j][r] ).
//
// One kernel template serves every dense contraction of the denoiser (reference:
// nn.Linear calls in codes/models/transformer.py:81-85,108-114,144-150,168,418,425 and their
// autograd backward):
//   forward   X = activations (reduce-contiguous), Y = nn.Linear weight (out,in)      [RC,RC]
//   dgrad     X = dC, Y = weight read reduce-slow                                      [RC,RS]
//   wgrad     X = dC^T, Y = activations^T, both reduce-slow, split over the M rows     [RS,RS]
// with the LayerNorm / stylization-modulation / SiLU of the reference fused into the operand
// staging (global -> registers -> transform -> LDS) and bias / GELU / residual / positional
// embedding fused into the epilogue.
//
// CDNA4 mapping: 256 threads = 4 waves (2x2), each wave owns (BI/2 x BJ/2) of the block tile as
// 32x32 MFMA accumulators.  The MFMA "row" operand is Y (output column j), so each lane ends
// up holding 4 CONSECUTIVE output columns per accumulator quad -> 16-byte epilogue loads and
// stores.  Exact-fp32 path: v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, the fp32 matrix peak of
// 157 TFLOP/s).  Reduce-contiguous tiles sit in LDS as [row][32+4] (pad = one b128 access, so
// the 16-lane ds_read_b128 groups are conflict-free); reduce-slow tiles as [32][rows] read with
// conflict-free ds_read_b32.  Inside a 32-deep K tile the k order is permuted identically for
// both operands: MFMA step j of group ks uses k = 8*ks + 4*(lane>>5) + j.
#include <stdlib.h>

#include "hig_common.h"
#include "hig_host.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef HIG_GEMM_NT
#define HIG_GEMM_NT 0
#endif
// Wave priority raised over the global fetch + MFMA block of each k-tile (0 = off): with 2-4
// workgroups per CU in different phases, the arbiter then favours the wave that can feed the
// matrix pipe over the ones doing the transform / LDS writes of their next tile (fwd -2.3%,
// fwd+bwd -1.3%, same-box A/B in profiles/r01_notes.md; raising it before the LDS writes too, or
// only around the MFMAs, is worse).
#ifndef HIG_GEMM_SETPRIO
#define HIG_GEMM_SETPRIO 1
#endif
#ifndef HIG_GEMM_LDS_EPI
#define HIG_GEMM_LDS_EPI 1
#endif

namespace {

constexpr int BK = 32;
constexpr int NTHREADS = 256;
constexpr int RC_LD = BK + 4;
constexpr int BF_LD = BK + 8;  // bf16 elements per LDS row in the bf16 modes (80 bytes)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct KArgs {
  hig_gemm_desc g;
  int nbj;         // tiles along J
  int ntiles;      // tiles per split
  int r_chunk;     // reduce elements per split (multiple of BK)
  int64_t slab;    // floats between split outputs
  int vecx, vecy, vecc;  // 16-byte access allowed for X / Y / (C,res,aux)
  float* xsum;           // reduce-slow X only: xsum[split * xsum_stride + i] = sum_r X[r][i] over this split, or null
  int64_t xsum_stride;   // (the per-split sums sit right behind each split's slab, so one reduction pass folds both)
  int epi_flags;         // bit 0: LDS-only barriers in the LDS-staged epilogue; bit 1: polynomial erf in the GELU epilogue
  int store_policy;      // staged epilogue's output stores: 0 plain, 1 `sc1` (write-through: the output leaves during the
                         // kernel instead of in the end-of-kernel L2 write-back), 2 `nt`; C must span < 2 GiB for 1 / 2
  // Split tail (tail_s > 1): the first `nmain` tiles (a multiple of 256: whole rounds of the chip) are computed as
  // usual; each of the `tail_rem` tiles behind them is cut into tail_s slices of the reduce range (tail_chunk
  // elements each), so that the last, partly filled round costs 1/tail_s of a tile instead of a whole one.  `ntiles`
  // then counts UNITS: nmain + tail_rem * tail_s.  A slice parks its accumulators in tail_ws, takes a ticket from
  // tail_cnt[tile]; the workgroup that draws the last ticket adds the slices in slice order (deterministic) and runs
  // the epilogue.  No workgroup ever waits for another one.
  int tail_s, nmain, tail_rem, tail_chunk;
  float* tail_ws;          // [tail_rem * tail_s][accumulator floats per thread][NTHREADS]
  unsigned* tail_cnt;      // [tail_rem], zero before the launch, left zero by it
  unsigned long long* stamps;   // diagnostic (hig_gemm_debug_stamps): s_memtime of the phases of each workgroup's first 2 tiles
};

// LDS-only workgroup barrier: orders the LDS traffic of the epilogue without the vmcnt(0) a __syncthreads() carries,
// so the tile's output stores (and the next tile's operand prefetch) stay in flight across it.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. below fp32 GELU rounding for |x| >~ 0.1 and at most
// 1e-7 absolute elsewhere): one exp, one rcp, five FMAs instead of libm erff's ~35 instructions.
__device__ __forceinline__ float gelu_poly(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

template <int XF>
__device__ __forceinline__ f32x4 xf_apply(f32x4 v, float mean, float rstd, const float* gamma,
                                          const float* beta, const float* ssrow, int shift_off,
                                          int f) {
  if (XF == HIG_XF_NONE) return v;
  if (XF == HIG_XF_SILU) {
    v.x = hig_silu(v.x); v.y = hig_silu(v.y); v.z = hig_silu(v.z); v.w = hig_silu(v.w);
    return v;
  }
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + f);
  const float4 b4 = *reinterpret_cast<const float4*>(beta + f);
  v.x = (v.x - mean) * rstd * g4.x + b4.x;
  v.y = (v.y - mean) * rstd * g4.y + b4.y;
  v.z = (v.z - mean) * rstd * g4.z + b4.z;
  v.w = (v.w - mean) * rstd * g4.w + b4.w;
  if (XF == HIG_XF_LN_MOD_SILU) {
    const float4 sc = *reinterpret_cast<const float4*>(ssrow + f);
    const float4 sh = *reinterpret_cast<const float4*>(ssrow + shift_off + f);
    v.x = hig_silu(v.x * (1.0f + sc.x) + sh.x);
    v.y = hig_silu(v.y * (1.0f + sc.y) + sh.y);
    v.z = hig_silu(v.z * (1.0f + sc.z) + sh.z);
    v.w = hig_silu(v.w * (1.0f + sc.w) + sh.w);
  }
  return v;
}

// General (unaligned / ragged) operands -- e.g. the F = 150 pose tensor, whose 600-byte rows rule out
// 16-byte loads: four 4-byte loads per quad at CLAMPED addresses (always in bounds, no control flow,
// lanes still sweep each row contiguously), the out-of-range reduce indices are zeroed later, when
// the registers are written to LDS (`mask_tiles`), so nothing waits on these loads early either.
__device__ __forceinline__ float4 ld_rc(const float* base, int64_t ld, int row, int nrows, int k,
                                        int kend) {
  const float* p = base + (int64_t)min(row, nrows - 1) * ld;
  const int kl = kend - 1;
  return make_float4(p[min(k, kl)], p[min(k + 1, kl)], p[min(k + 2, kl)], p[min(k + 3, kl)]);
}
__device__ __forceinline__ float4 ld_rs(const float* base, int64_t ld, int i, int nrows, int k,
                                        int kend) {
  const float* p = base + (int64_t)min(k, kend - 1) * ld;
  const int il = nrows - 1;
  return make_float4(p[min(i, il)], p[min(i + 1, il)], p[min(i + 2, il)], p[min(i + 3, il)]);
}

// Branch-free loads for the common case (16-byte aligned operand, k-tile entirely inside the
// reduce range): out-of-range ROWS are clamped to the last valid row instead of being zeroed --
// they only feed accumulator rows/columns the epilogue never stores -- so a tile's loads issue
// back to back with no control flow and no use before the MFMA block.  (A guarded load makes
// hipcc branch around it and wait vmcnt(0) each time, serialising the whole tile fetch.)
__device__ __forceinline__ float4 ld_rc_fast(const float* base, int64_t ld, int row, int nrows, int k) {
  return *reinterpret_cast<const float4*>(base + (int64_t)min(row, nrows - 1) * ld + k);
}
__device__ __forceinline__ float4 ld_rs_fast(const float* base, int64_t ld, int i, int nrows, int k) {
  return *reinterpret_cast<const float4*>(base + (int64_t)k * ld + min(i, nrows - 4));
}

// Output tile store.  -DHIG_GEMM_NT=1 makes it non-temporal: that keeps the tile from evicting the X row
// panels / W the other column tiles of the XCD re-read (FFN1 fetch 61 -> 43 MB), but the NEXT kernel then
// finds its input in the Infinity Cache instead of L2, and the whole path gets slower (forward 7.26 ->
// 7.38 ms, bf16 products 3.9 -> 5.0 ms; profiles/r01_notes.md) -- so plain stores are the default.
__device__ __forceinline__ void st_stream(float* p, const float (&v)[4]) {
  const f32x4 x = {v[0], v[1], v[2], v[3]};
#if HIG_GEMM_NT
  __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<f32x4*>(p) = x;
#endif
}

// FAST = every operand 16-byte aligned and the reduce extent a multiple of BK: the tile fetch
// is a straight run of clamped float4 loads (chosen on the host; the general kernel keeps the
// guarded loads for odd shapes such as F = 150).
//
// PREC selects the product arithmetic (reduce-contiguous operands only for PREC != F32):
//   HIG_PREC_F32     v_mfma_f32_32x32x2_f32, exact fp32 products
//   HIG_PREC_BF16X3  each fp32 operand is split x = hi + lo (bf16 each, lo = bf16(x - hi)) while it
//                    is staged into LDS; a*b ~= ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_bf16
//                    (3 MFMAs at 16x the fp32 MFMA rate, relative error ~2^-16 per product,
//                    fp32 accumulate)
//   HIG_PREC_BF16    ah*bh only
// LDS image for the bf16 modes: per operand a hi plane (and a lo plane) of [rows][32+8] bf16 --
// 80-byte rows keep the 16-lane ds_read_b128 groups conflict-free; one 16-byte read is one MFMA
// operand (8 consecutive k of one row).
template <int BI, int BJ, bool X_RS, bool Y_RS, int XF, bool XF_ON_Y, int EPI, bool FAST, int PREC>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_f32_kernel(const KArgs a) {
  static_assert(PREC == HIG_PREC_F32 || (!X_RS && !Y_RS), "bf16 modes need reduce-contiguous operands");
  constexpr int TI = BI / 64, TJ = BJ / 64;
  constexpr int XP = BI / 32, YP = BJ / 32;
  constexpr int NPLANE = PREC == HIG_PREC_BF16X3 ? 2 : 1;
  constexpr int X_TILE = PREC != HIG_PREC_F32 ? BI * (BF_LD / 2) * NPLANE : (X_RS ? BK * BI : BI * RC_LD);
  constexpr int Y_TILE = PREC != HIG_PREC_F32 ? BJ * (BF_LD / 2) * NPLANE : (Y_RS ? BK * BJ : BJ * RC_LD);
  constexpr int STAGE = X_TILE + Y_TILE;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const hig_gemm_desc& g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
  // contiguous run of tiles so neighbours re-use the same X row panel.
  // The kernel is PERSISTENT over tiles: block b walks linear ids b, b + gridDim.x, ... (gridDim.x
  // is a multiple of 8, so a block keeps its XCD), and fetches the first k-tile of its next tile
  // before the epilogue of the current one -- no per-tile launch gap, no exposed prologue latency.
  int i0 = 0, j0 = 0;
  const int split = blockIdx.y;
  int rbeg = split * a.r_chunk;
  int rend = min(g.R, rbeg + a.r_chunk);
  int nk = (rend - rbeg + BK - 1) / BK;
  constexpr bool TAIL = PREC == HIG_PREC_F32 && !X_RS && FAST && BI == 64 && BJ == 64;   // kernels the host rule picks
  int u_slice = -1, u_tile = 0;     // split tail: slice of the reduce range / tail tile this unit belongs to
  auto tile_coords = [&](int lin) {
    int tile;
    if (TAIL && a.tail_s > 1 && lin >= a.nmain) {
      const int lp = lin - a.nmain;
      if ((a.tail_rem & 7) == 0) {   // the slices of one tile stay on one XCD (lin & 7): its L2 holds the partial sums
        u_tile = (lp & 7) + 8 * ((lp >> 3) / a.tail_s);
        u_slice = (lp >> 3) % a.tail_s;
      } else {
        u_tile = lp / a.tail_s;
        u_slice = lp % a.tail_s;
      }
      tile = a.nmain + u_tile;
      rbeg = u_slice * a.tail_chunk;
      rend = rbeg + a.tail_chunk;
    } else {
      const int nt = (TAIL && a.tail_s > 1) ? a.nmain : a.ntiles;
      const int q = nt >> 3, r = nt & 7, xcd = lin & 7;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
      u_slice = -1;
    }
    nk = (rend - rbeg + BK - 1) / BK;
    i0 = (tile / a.nbj) * BI;
    j0 = (tile % a.nbj) * BJ;
  };
  float* __restrict__ C = g.C + (int64_t)split * a.slab;
  typedef int gi32x4 [[maybe_unused]] __attribute__((ext_vector_type(4)));
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C, 0, a.store_policy ? (int)((((int64_t)g.I - 1) * g.ldc + g.J) * 4) : 0, 0x00020000);
  __shared__ unsigned s_ticket;
  int nth = 0;    // tiles this workgroup has started
  auto stamp = [&](int k) {
    if (a.stamps && nth == 2 && threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {   // the SECOND tile: steady state
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      a.stamps[(size_t)blockIdx.x * 8 + k] = t;
    }
  };

  // ---- per-thread staging coordinates ------------------------------------------------
  // RC tile [ROWS][32]: c4 = tid&7, row = (tid>>3) + 32p.   RS tile [32][ROWS]: ROWS/4 float4
  // per k-row.
  constexpr int XQ = BI / 4, YQ = BJ / 4;
  const int x_c4 = X_RS ? (tid % XQ) : (tid & 7);
  const int x_r = X_RS ? (tid / XQ) : (tid >> 3);
  constexpr int X_RSTEP = X_RS ? (NTHREADS / XQ) : 32;
  const int y_c4 = Y_RS ? (tid % YQ) : (tid & 7);
  const int y_r = Y_RS ? (tid / YQ) : (tid >> 3);
  constexpr int Y_RSTEP = Y_RS ? (NTHREADS / YQ) : 32;

  // Hoisted per-row transform state for a reduce-contiguous activation operand on X.
  float xmean[XP], xrstd[XP];
  const float* xss[XP];
  auto setup_row_state = [&]() {
    if (XF != HIG_XF_NONE && XF != HIG_XF_SILU && !XF_ON_Y && !X_RS) {
#pragma unroll
      for (int p = 0; p < XP; ++p) {
        const int m = min(i0 + x_r + 32 * p, g.I - 1);
        xmean[p] = g.stats[2 * (int64_t)m];
        xrstd[p] = g.stats[2 * (int64_t)m + 1];
        xss[p] = (XF == HIG_XF_LN_MOD_SILU) ? g.ss + (int64_t)(m / g.rows_per_sample) * g.ss_ld : nullptr;
      }
    }
  };

  // Operand staging registers (global -> registers -> [transform] -> LDS); see `iteration`.
  // native vector type (not HIP's float4 struct): keeps the arrays in registers (SROA)
  f32x4 xr[XP], yr[YP];
  // FAST path: per-thread source pointers live in registers for the whole kernel and advance by
  // one k-tile per fetch (no per-load address temporaries for the compiler to alias with the
  // staging registers).  Out-of-range rows are clamped (see ld_*_fast).
  const float* xptr[XP];
  const float* yptr[YP];
  auto setup_pointers = [&]() {
    if (FAST) {
#pragma unroll
      for (int p = 0; p < XP; ++p)
        xptr[p] = X_RS ? g.X + (int64_t)(rbeg + x_r + X_RSTEP * p) * g.ldx + min(i0 + 4 * x_c4, g.I - 4)
                       : g.X + (int64_t)min(i0 + x_r + 32 * p, g.I - 1) * g.ldx + rbeg + 4 * x_c4;
#pragma unroll
      for (int p = 0; p < YP; ++p)
        yptr[p] = Y_RS ? g.Y + (int64_t)(rbeg + y_r + Y_RSTEP * p) * g.ldy + min(j0 + 4 * y_c4, g.J - 4)
                       : g.Y + (int64_t)min(j0 + y_r + 32 * p, g.J - 1) * g.ldy + rbeg + 4 * y_c4;
    }
  };
  const int64_t xstep = X_RS ? (int64_t)BK * g.ldx : BK;
  const int64_t ystep = Y_RS ? (int64_t)BK * g.ldy : BK;
  auto load_tiles = [&](int k0) {
    if (FAST) {  // tiles are fetched in order: the pointers already sit at k0
#pragma unroll
      for (int p = 0; p < XP; ++p) {
        xr[p] = *reinterpret_cast<const f32x4*>(xptr[p]);
        xptr[p] += xstep;
      }
#pragma unroll
      for (int p = 0; p < YP; ++p) {
        yr[p] = *reinterpret_cast<const f32x4*>(yptr[p]);
        yptr[p] += ystep;
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      float4 t4;
      if (X_RS)
        t4 = ld_rs(g.X, g.ldx, i0 + 4 * x_c4, g.I, k0 + x_r + X_RSTEP * p, rend);
      else
        t4 = ld_rc(g.X, g.ldx, i0 + x_r + 32 * p, g.I, k0 + 4 * x_c4, rend);
      xr[p] = f32x4{t4.x, t4.y, t4.z, t4.w};
    }
#pragma unroll
    for (int p = 0; p < YP; ++p) {
      float4 t4;
      if (Y_RS)
        t4 = ld_rs(g.Y, g.ldy, j0 + 4 * y_c4, g.J, k0 + y_r + Y_RSTEP * p, rend);
      else
        t4 = ld_rc(g.Y, g.ldy, j0 + y_r + 32 * p, g.J, k0 + 4 * y_c4, rend);
      yr[p] = f32x4{t4.x, t4.y, t4.z, t4.w};
    }
  };
  // general path only: zero what the clamped loads fetched beyond the reduce range of tile k0
  auto mask_tiles = [&](int k0) {
    if (FAST) return;
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      if (X_RS) {
        if (k0 + x_r + X_RSTEP * p >= rend) xr[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        const int k = k0 + 4 * x_c4;
        if (k >= rend) xr[p].x = 0.f;
        if (k + 1 >= rend) xr[p].y = 0.f;
        if (k + 2 >= rend) xr[p].z = 0.f;
        if (k + 3 >= rend) xr[p].w = 0.f;
      }
    }
#pragma unroll
    for (int p = 0; p < YP; ++p) {
      if (Y_RS) {
        if (k0 + y_r + Y_RSTEP * p >= rend) yr[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        const int k = k0 + 4 * y_c4;
        if (k >= rend) yr[p].x = 0.f;
        if (k + 1 >= rend) yr[p].y = 0.f;
        if (k + 2 >= rend) yr[p].z = 0.f;
        if (k + 3 >= rend) yr[p].w = 0.f;
      }
    }
  };
  auto transform = [&](int k0) {
    if (XF == HIG_XF_NONE) return;
    if (!XF_ON_Y) {
      // activation = X, reduce-contiguous: row m fixed per p, features k0+4c4..
      const int f = k0 + 4 * x_c4;
      if (f < rend) {  // R % 4 == 0 is required for transformed operands
#pragma unroll
        for (int p = 0; p < XP; ++p) {
          if (XF == HIG_XF_SILU)
            xr[p] = xf_apply<XF>(xr[p], 0.f, 0.f, nullptr, nullptr, nullptr, 0, f);
          else if (i0 + x_r + 32 * p < g.I)
            xr[p] = xf_apply<XF>(xr[p], xmean[p], xrstd[p], g.gamma, g.beta, xss[p],
                                 g.ss_shift_off, f);
        }
      }
    } else {
      // activation = Y, reduce-slow: row m = reduce index, features j0+4c4..
      const int f = j0 + 4 * y_c4;
      if (f < g.J) {  // J % 4 == 0 is required for transformed operands
#pragma unroll
        for (int p = 0; p < YP; ++p) {
          const int m = k0 + y_r + Y_RSTEP * p;
          if (m < rend) {
            if (XF == HIG_XF_SILU) {
              yr[p] = xf_apply<XF>(yr[p], 0.f, 0.f, nullptr, nullptr, nullptr, 0, f);
            } else {
              const float mean = g.stats[2 * (int64_t)m], rstd = g.stats[2 * (int64_t)m + 1];
              const float* ssrow =
                  (XF == HIG_XF_LN_MOD_SILU) ? g.ss + (int64_t)(m / g.rows_per_sample) * g.ss_ld : nullptr;
              yr[p] = xf_apply<XF>(yr[p], mean, rstd, g.gamma, g.beta, ssrow, g.ss_shift_off, f);
            }
          }
        }
      }
    }
  };
  auto split_store = [&](__bf16* plane0, int rows, int row, int c4, const f32x4 v) {
    // hi = bf16(v) (RNE), lo = bf16(v - hi): 8-byte stores into the hi / lo planes
    bf16x4 hi = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<bf16x4*>(plane0 + row * BF_LD + 4 * c4) = hi;
    if (PREC == HIG_PREC_BF16X3) {
      bf16x4 lo = {(__bf16)(v.x - (float)hi[0]), (__bf16)(v.y - (float)hi[1]), (__bf16)(v.z - (float)hi[2]),
                   (__bf16)(v.w - (float)hi[3])};
      *reinterpret_cast<bf16x4*>(plane0 + rows * BF_LD + row * BF_LD + 4 * c4) = lo;
    }
  };
  auto store_tiles = [&](int buf) {
    float* sx = smem + buf * STAGE;
    float* sy = sx + X_TILE;
    if (PREC != HIG_PREC_F32) {
#pragma unroll
      for (int p = 0; p < XP; ++p) split_store(reinterpret_cast<__bf16*>(sx), BI, x_r + 32 * p, x_c4, xr[p]);
#pragma unroll
      for (int p = 0; p < YP; ++p) split_store(reinterpret_cast<__bf16*>(sy), BJ, y_r + 32 * p, y_c4, yr[p]);
      return;
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      if (X_RS)
        *reinterpret_cast<f32x4*>(sx + (x_r + X_RSTEP * p) * BI + 4 * x_c4) = xr[p];
      else
        *reinterpret_cast<f32x4*>(sx + (x_r + 32 * p) * RC_LD + 4 * x_c4) = xr[p];
    }
#pragma unroll
    for (int p = 0; p < YP; ++p) {
      if (Y_RS)
        *reinterpret_cast<f32x4*>(sy + (y_r + Y_RSTEP * p) * BJ + 4 * y_c4) = yr[p];
      else
        *reinterpret_cast<f32x4*>(sy + (y_r + 32 * p) * RC_LD + 4 * y_c4) = yr[p];
    }
  };

  f32x16 acc[TJ][TI];

  const int xrow = wi * (32 * TI) + lr;  // + 32*ti
  const int yrow = wj * (32 * TJ) + lr;  // + 32*tj
  // MFMA over k-groups [ks0, ks1) of the staged tile in LDS buffer `buf`
  auto mfma_part = [&](int buf, int ks0, int ks1) {
    const float* sx = smem + buf * STAGE;
    const float* sy = sx + X_TILE;
    if (PREC != HIG_PREC_F32) {
      // here a "k-group" is one 16-deep bf16 MFMA step; lane: row lr, k = 16*ks + 8*lh .. +7
      const __bf16* bx = reinterpret_cast<const __bf16*>(sx);
      const __bf16* by = reinterpret_cast<const __bf16*>(sy);
#pragma unroll
      for (int ks = ks0; ks < ks1; ++ks) {
        bf16x8 xh[TI], xl[TI], yh[TJ], yl[TJ];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
          const __bf16* pp = bx + (xrow + 32 * ti) * BF_LD + 16 * ks + 8 * lh;
          xh[ti] = *reinterpret_cast<const bf16x8*>(pp);
          if (PREC == HIG_PREC_BF16X3) xl[ti] = *reinterpret_cast<const bf16x8*>(pp + BI * BF_LD);
        }
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
          const __bf16* pp = by + (yrow + 32 * tj) * BF_LD + 16 * ks + 8 * lh;
          yh[tj] = *reinterpret_cast<const bf16x8*>(pp);
          if (PREC == HIG_PREC_BF16X3) yl[tj] = *reinterpret_cast<const bf16x8*>(pp + BJ * BF_LD);
        }
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) {
            if (PREC == HIG_PREC_BF16X3) {  // small cross terms first, then the leading term
              acc[tj][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yl[tj], xh[ti], acc[tj][ti], 0, 0, 0);
              acc[tj][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[tj], xl[ti], acc[tj][ti], 0, 0, 0);
            }
            acc[tj][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[tj], xh[ti], acc[tj][ti], 0, 0, 0);
          }
      }
      return;
    }
#pragma unroll
    for (int ks = ks0; ks < ks1; ++ks) {
      float xf[TI][4], yf[TJ][4];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        if (X_RS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) xf[ti][j] = sx[(ks * 8 + 4 * lh + j) * BI + xrow + 32 * ti];
        } else {
          const float4 t4 = *reinterpret_cast<const float4*>(sx + (xrow + 32 * ti) * RC_LD + ks * 8 + 4 * lh);
          xf[ti][0] = t4.x; xf[ti][1] = t4.y; xf[ti][2] = t4.z; xf[ti][3] = t4.w;
        }
      }
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        if (Y_RS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) yf[tj][j] = sy[(ks * 8 + 4 * lh + j) * BJ + yrow + 32 * tj];
        } else {
          const float4 t4 = *reinterpret_cast<const float4*>(sy + (yrow + 32 * tj) * RC_LD + ks * 8 + 4 * lh);
          yf[tj][0] = t4.x; yf[tj][1] = t4.y; yf[tj][2] = t4.z; yf[tj][3] = t4.w;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
            acc[tj][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(yf[tj][j], xf[ti][j], acc[tj][ti], 0, 0, 0);
    }
  };
  // One k-tile.  On entry the staging registers hold tile kt+1, fetched at the start of the
  // PREVIOUS iteration (a whole MFMA block ago).  Order: (1) transform + write them to the other
  // LDS buffer (free since the barrier that ended iteration kt-1), (2) re-issue the same
  // registers as the fetch of tile kt+2, (3) the MFMA block on the current buffer, (4) barrier.
  // The only global-memory wait is on loads that had a full MFMA block to land, the LDS writes
  // have a full MFMA block before the barrier, and one register set suffices.
  // Weight gradients read dC as the reduce-slow X operand; the tiles of the first output-column block sum what they
  // stage over the reduce rows -- that IS the bias gradient of the same layer (one column sum of dC saved per layer op).
  f32x4 xsum4 = {0.f, 0.f, 0.f, 0.f};
  bool track_xsum = false;
  auto add_xsum = [&]() {
    if constexpr (X_RS && PREC == HIG_PREC_F32) {
      if (track_xsum) {
#pragma unroll
        for (int p = 0; p < XP; ++p) xsum4 += xr[p];
      }
    }
  };
  auto iteration = [&](int kt, int nk, int buf) {
    constexpr int NG = PREC != HIG_PREC_F32 ? BK / 16 : BK / 8;  // k-groups per tile
    if (kt + 1 < nk) {
      mask_tiles(rbeg + (kt + 1) * BK);
      add_xsum();
      transform(rbeg + (kt + 1) * BK);
      store_tiles(buf ^ 1);
    }
#if HIG_GEMM_SETPRIO
    __builtin_amdgcn_s_setprio(HIG_GEMM_SETPRIO);
#endif
    if (kt + 2 < nk) load_tiles(rbeg + (kt + 2) * BK);
    mfma_part(buf, 0, NG);
#if HIG_GEMM_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();
  };

  bool prefetched = false;
  for (int lin = blockIdx.x; lin < a.ntiles; lin += gridDim.x) {
    ++nth;
    stamp(0);
    if (!prefetched) {  // first tile of this block: nothing in flight yet
      tile_coords(lin);
      setup_pointers();
      if (nk > 0) load_tiles(rbeg);
    }
    setup_row_state();
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tj][ti][e] = 0.f;
    if constexpr (X_RS && PREC == HIG_PREC_F32) {
      track_xsum = a.xsum != nullptr && j0 == 0;
      xsum4 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (nk > 0) {
      mask_tiles(rbeg);
      add_xsum();
      transform(rbeg);
      store_tiles(0);
      if (nk > 1) load_tiles(rbeg + BK);
    }
    __syncthreads();
    stamp(1);
    for (int kt = 0; kt < nk; kt += 2) {
      iteration(kt, nk, 0);
      if (kt + 1 < nk) iteration(kt + 1, nk, 1);
    }
    stamp(2);
    const int cur_slice = u_slice, cur_tile = u_tile;
    if constexpr (X_RS && PREC == HIG_PREC_F32) {
      if (track_xsum) {   // fold the NTHREADS / XQ thread rows that share a column quad, through the idle staging LDS
        float* sB = smem;
        *reinterpret_cast<f32x4*>(sB + x_r * BI + 4 * x_c4) = xsum4;
        __syncthreads();
        if (tid < BI && i0 + tid < g.I) {
          float t = 0.f;
#pragma unroll
          for (int r = 0; r < NTHREADS / XQ; ++r) t += sB[r * BI + tid];
          a.xsum[(int64_t)split * a.xsum_stride + i0 + tid] = t;
        }
        __syncthreads();
      }
    }
    // epilogue coordinates of THIS tile, then start fetching the next one (the staging registers
    // and both LDS buffers are free: the main loop ended on a barrier)
    const int ei0 = i0, ej0 = j0;
    prefetched = lin + (int)gridDim.x < a.ntiles;
    if (prefetched) {
      tile_coords(lin + gridDim.x);
      setup_pointers();
      if (nk > 0) load_tiles(rbeg);
    }
    stamp(3);
    if (TAIL && cur_slice >= 0) {   // split tail: park the partial sums, draw a ticket; the last slice to arrive finishes the tile
      // Every access to the parked sums and the tickets is a device-scope atomic (sc1: performed at the memory side,
      // past the per-XCD L2s), ordered by the vmcnt(0) of the barrier between them.  No device-scope FENCE: its
      // acquire half invalidates the whole L2 of the XCD under the workgroups still streaming operands (measured:
      // +32 us per GEMM).
      constexpr int AF = TI * TJ * 16;
      float* mine = a.tail_ws + ((int64_t)(cur_tile * a.tail_s + cur_slice) * AF) * NTHREADS + tid;
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            __hip_atomic_store(mine + ((tj * TI + ti) * 16 + e) * NTHREADS, acc[tj][ti][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // Every storing wave drains its own stores BEFORE the barrier: gfx950's s_barrier does not wait on vmcnt and
      // hipcc emits no wait for a workgroup-scope barrier here, so without this the ticket could be counted while
      // another wave's partial sums are still in flight (the last arriver would then sum stale scratch).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) s_ticket = __hip_atomic_fetch_add(a.tail_cnt + cur_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const bool last = s_ticket == (unsigned)(a.tail_s - 1);
      __syncthreads();     // (s_ticket is rewritten by this workgroup's next unit)
      if (!last) continue;
      if (tid == 0) __hip_atomic_store(a.tail_cnt + cur_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[tj][ti][e] = 0.f;
      for (int sl = 0; sl < a.tail_s; ++sl) {   // slice order: the sum does not depend on who arrived when
        const float* part = a.tail_ws + ((int64_t)(cur_tile * a.tail_s + sl) * AF) * NTHREADS + tid;
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              acc[tj][ti][e] += __hip_atomic_load(part + ((tj * TI + ti) * 16 + e) * NTHREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

#if HIG_GEMM_LDS_EPI
  // ---- epilogue through LDS: the accumulator layout (a lane = one row x 4 columns per quad) would write
  // 32-byte pieces of 32 different rows per store instruction; staging the tile in LDS (free here: the main
  // loop ended on a barrier and the next tile's first k-tile waits in registers) lets every instruction
  // read `res` / `aux` and write C as whole 16-byte-per-lane rows (BJ * 4 contiguous bytes per row).
  constexpr int CLD = BJ + 4;
  if constexpr (2 * STAGE >= BI * CLD)      // (tiles whose staging buffers are smaller keep the direct epilogue)
  if (a.vecc && ej0 + BJ <= g.J) {
    float* sC = smem;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(sC + (wi * (32 * TI) + 32 * ti + lr) * CLD + wj * (32 * TJ) + 32 * tj + 8 * q + 4 * lh) =
              f32x4{acc[tj][ti][4 * q], acc[tj][ti][4 * q + 1], acc[tj][ti][4 * q + 2], acc[tj][ti][4 * q + 3]};
    [[maybe_unused]] float* sStat = sC + BI * CLD;      // LayerNorm fold, consumer: (rstd, -mean rstd) of the tile's rows
    if constexpr (EPI == HIG_EPI_BIAS && BJ == 64) {
      if (g.row_stats_in && tid < BI) {
        // mean / variance of row ei0 + tid from its R / 64 panel statistics (sum, centred sum of squares): pairwise merge,
        // no E[x^2] - mean^2 cancellation.  Once per row and tile, by one thread: the epilogue below reads two floats.
        const int i = min(ei0 + tid, g.I - 1);
        const int np = g.R >> 6;
        const float* sp = g.row_stats_in + (int64_t)i * np * 2;
        float tot = 0.f, m2 = 0.f;
        for (int p2 = 0; p2 < np; p2 += 2) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(sp + 2 * p2);
          tot += t4.x + t4.z;
          m2 += t4.y + t4.w;
        }
        const float mean = tot / (float)g.R;
        float between = 0.f;
        for (int p2 = 0; p2 < np; p2 += 2) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(sp + 2 * p2);
          const float d0 = t4.x * (1.0f / 64.0f) - mean, d1 = t4.z * (1.0f / 64.0f) - mean;
          between += d0 * d0 + d1 * d1;
        }
        const float rstd = rsqrtf((m2 + 64.0f * between) / (float)g.R + 1e-5f);
        sStat[2 * tid] = rstd;
        sStat[2 * tid + 1] = -mean * rstd;
      }
    }
    if (a.epi_flags & 1) lds_barrier(); else __syncthreads();
    stamp(4);
    constexpr int Q4 = BJ / 4, RPP = NTHREADS / Q4;     // float4 per row, rows per pass
    const int c4 = tid % Q4, rr0 = tid / Q4;
    const int j = ej0 + 4 * c4;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (EPI == HIG_EPI_BIAS || EPI == HIG_EPI_BIAS_GELU || EPI == HIG_EPI_BIAS_RES || EPI == HIG_EPI_BIAS_POS)
      b4 = *reinterpret_cast<const f32x4*>(g.bias + j);
    // LayerNorm folded into this GEMM (hig_gemm_desc.row_stats_in / ln_colsum; 64-column tiles only): X holds UN-normalised
    // rows whose statistics per 64-column panel were written by the GEMM that produced them, Y is W' = gamma (.) W:
    //   C = rstd (X W'^T) - rstd mean colsum + bias'   ==   LayerNorm(X) W^T + bias       (transformer.py:108-110,144)
    [[maybe_unused]] f32x4 cs4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == HIG_EPI_BIAS && BJ == 64) {
      if (g.row_stats_in) cs4 = *reinterpret_cast<const f32x4*>(g.ln_colsum + j);
    }
#pragma unroll 4
    for (int rr = rr0; rr < BI; rr += RPP) {
      const int i = ei0 + rr;
      if (i >= g.I) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(sC + rr * CLD + 4 * c4);
      if constexpr (EPI == HIG_EPI_BIAS && BJ == 64) {
        if (g.row_stats_in) {
          const float rstd = sStat[2 * rr], mr = sStat[2 * rr + 1];
          v = f32x4{v.x * rstd + mr * cs4.x, v.y * rstd + mr * cs4.y, v.z * rstd + mr * cs4.z, v.w * rstd + mr * cs4.w};
        }
      }
      v += b4;
      if (EPI == HIG_EPI_BIAS_POS) {
        const int tp = (i % g.T) - g.pos_shift;
        if (tp >= 0) v += *reinterpret_cast<const f32x4*>(g.pos + (int64_t)tp * g.ldpos + j);
      }
      if (EPI == HIG_EPI_BIAS_RES || EPI == HIG_EPI_RES) v += *reinterpret_cast<const f32x4*>(g.res + (int64_t)i * g.ldr + j);
      if (EPI == HIG_EPI_BIAS_GELU) {
        if (g.aux) *reinterpret_cast<f32x4*>(g.aux + (int64_t)i * g.ldaux + j) = v;
        if (a.epi_flags & 2) v = f32x4{gelu_poly(v.x), gelu_poly(v.y), gelu_poly(v.z), gelu_poly(v.w)};
        else v = f32x4{hig_gelu(v.x), hig_gelu(v.y), hig_gelu(v.z), hig_gelu(v.w)};
      }
      if (EPI == HIG_EPI_DGELU) {
        const f32x4 z = *reinterpret_cast<const f32x4*>(g.aux + (int64_t)i * g.ldaux + j);
        v = f32x4{v.x * hig_dgelu(z.x), v.y * hig_dgelu(z.y), v.z * hig_dgelu(z.z), v.w * hig_dgelu(z.w)};
      }
      if constexpr (EPI == HIG_EPI_BIAS_RES && BJ == 64) {
        if (g.row_stats_out) {
          // LayerNorm fold, producer side: (sum, sum of squared deviations from the panel mean) of this row's 64 outputs of
          // THIS tile -- the row's 16 lanes are one DPP row
          const float sm = row16_sum((v.x + v.y) + (v.z + v.w));
          const float pm = sm * (1.0f / 64.0f);
          const float a0 = v.x - pm, a1 = v.y - pm, a2 = v.z - pm, a3 = v.w - pm;
          const float qq = row16_sum((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3));
          if (c4 == 0) *reinterpret_cast<float2*>(g.row_stats_out + ((int64_t)i * (g.J >> 6) + (ej0 >> 6)) * 2) = make_float2(sm, qq);
        }
      }
#if defined(__HIP_DEVICE_COMPILE__)
      if (a.store_policy == 1)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gi32x4, v), rsC, (int)(((int64_t)i * g.ldc + j) * 4), 0, 16);
      else if (a.store_policy == 2)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gi32x4, v), rsC, (int)(((int64_t)i * g.ldc + j) * 4), 0, 2);
      else
#endif
        *reinterpret_cast<f32x4*>(C + (int64_t)i * g.ldc + j) = v;
    }
    stamp(5);
    if (a.epi_flags & 1) lds_barrier(); else __syncthreads();   // the next tile's store_tiles() reuses this LDS
    stamp(6);
    continue;
  }
  // tiles too large for one pass (128-row tiles of the bf16 modes, whose staging buffers are small): the two
  // 64-row halves -- one wave row each -- go through LDS in turn
  if constexpr (2 * STAGE < BI * CLD && TI == 2 && 2 * STAGE >= (BI / 2) * CLD)
  if (a.vecc && ej0 + BJ <= g.J) {
    float* sC = smem;
    constexpr int Q4 = BJ / 4, RPP = NTHREADS / Q4, PROWS = BI / 2;
    const int c4 = tid % Q4, rr0 = tid / Q4;
    const int j = ej0 + 4 * c4;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (EPI == HIG_EPI_BIAS || EPI == HIG_EPI_BIAS_GELU || EPI == HIG_EPI_BIAS_RES || EPI == HIG_EPI_BIAS_POS)
      b4 = *reinterpret_cast<const f32x4*>(g.bias + j);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      if (wi == ps) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<f32x4*>(sC + (32 * ti + lr) * CLD + wj * (32 * TJ) + 32 * tj + 8 * q + 4 * lh) =
                  f32x4{acc[tj][ti][4 * q], acc[tj][ti][4 * q + 1], acc[tj][ti][4 * q + 2], acc[tj][ti][4 * q + 3]};
      }
      __syncthreads();
#pragma unroll 4
      for (int rr = rr0; rr < PROWS; rr += RPP) {
        const int i = ei0 + ps * PROWS + rr;
        if (i >= g.I) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(sC + rr * CLD + 4 * c4) + b4;
        if (EPI == HIG_EPI_BIAS_POS) {
          const int tp = (i % g.T) - g.pos_shift;
          if (tp >= 0) v += *reinterpret_cast<const f32x4*>(g.pos + (int64_t)tp * g.ldpos + j);
        }
        if (EPI == HIG_EPI_BIAS_RES || EPI == HIG_EPI_RES) v += *reinterpret_cast<const f32x4*>(g.res + (int64_t)i * g.ldr + j);
        if (EPI == HIG_EPI_BIAS_GELU) {
          if (g.aux) *reinterpret_cast<f32x4*>(g.aux + (int64_t)i * g.ldaux + j) = v;
          v = f32x4{hig_gelu(v.x), hig_gelu(v.y), hig_gelu(v.z), hig_gelu(v.w)};
        }
        if (EPI == HIG_EPI_DGELU) {
          const f32x4 z = *reinterpret_cast<const f32x4*>(g.aux + (int64_t)i * g.ldaux + j);
          v = f32x4{v.x * hig_dgelu(z.x), v.y * hig_dgelu(z.y), v.z * hig_dgelu(z.z), v.w * hig_dgelu(z.w)};
        }
        *reinterpret_cast<f32x4*>(C + (int64_t)i * g.ldc + j) = v;
      }
      __syncthreads();
    }
    continue;
  }
#endif
  // ---- epilogue: lane holds, per accumulator quad q, columns j..j+3 of row i ----------
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    const int i = ei0 + wi * (32 * TI) + 32 * ti + lr;
    if (i >= g.I) continue;
    const float* posrow = nullptr;
    if (EPI == HIG_EPI_BIAS_POS) {
      const int tp = (i % g.T) - g.pos_shift;
      if (tp >= 0) posrow = g.pos + (int64_t)tp * g.ldpos;
    }
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = ej0 + wj * (32 * TJ) + 32 * tj + 8 * q + 4 * lh;
        if (j >= g.J) continue;
        float v[4] = {acc[tj][ti][4 * q], acc[tj][ti][4 * q + 1], acc[tj][ti][4 * q + 2],
                      acc[tj][ti][4 * q + 3]};
        const bool full = a.vecc && (j + 3 < g.J);
        const int nv = min(4, g.J - j);
        if (EPI == HIG_EPI_BIAS || EPI == HIG_EPI_BIAS_GELU || EPI == HIG_EPI_BIAS_RES ||
            EPI == HIG_EPI_BIAS_POS) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nv) v[e] += g.bias[j + e];
        }
        if (EPI == HIG_EPI_BIAS_POS && posrow) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nv) v[e] += posrow[j + e];
        }
        if (EPI == HIG_EPI_BIAS_RES || EPI == HIG_EPI_RES) {
          const float* rp = g.res + (int64_t)i * g.ldr + j;
          if (full) {
            const float4 r4 = *reinterpret_cast<const float4*>(rp);
            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < nv) v[e] += rp[e];
          }
        }
        if (EPI == HIG_EPI_BIAS_GELU) {
          if (g.aux) {
            float* ap = g.aux + (int64_t)i * g.ldaux + j;
            if (full) {
              st_stream(ap, v);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (e < nv) ap[e] = v[e];
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = hig_gelu(v[e]);
        }
        if (EPI == HIG_EPI_DGELU) {
          const float* ap = g.aux + (int64_t)i * g.ldaux + j;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nv) v[e] *= hig_dgelu(ap[e]);
        }
        float* cp = C + (int64_t)i * g.ldc + j;
        if (full) {
          st_stream(cp, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nv) cp[e] = v[e];
        }
      }
    }
  }
  }  // persistent tile loop
}

// out[e] = sum_s slabs[s][e] for the first n4 float4 of a slab; the float4 behind them (n4 <= e < n4_all: the
// per-split column sums of X) go to out2
// (few-row forward GEMMs: + bias[column] and + res[row][column], the epilogue the split launch could not apply)
struct SplitEpilogue {
  const float* bias;   // [J] or null
  const float* res;    // rows of ldr floats, or null
  int64_t ldr;
  int J4;              // J / 4
};
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nsplit, int64_t slab,
                                    int64_t n4, float* __restrict__ out, int64_t n4_all, float* __restrict__ out2,
                                    SplitEpilogue se) {
  const int64_t total = n4_all > n4 ? n4_all : n4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    // (eight slabs requested before the first is added -- same order of additions, so the same bits: with one load per trip of
    //  a loop whose count the compiler does not know, every split paid a full L2 latency: 13-17 us per reduction of 16-28 MB)
    // (the last, partial batch as well: loads clamped to the last slab, the surplus not added)
    float4 s = reinterpret_cast<const float4*>(slabs)[e];
    for (int k = 1; k < nsplit; k += 8) {
      float4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = reinterpret_cast<const float4*>(slabs + (int64_t)min(k + j, nsplit - 1) * slab)[e];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k + j < nsplit) { s.x += t[j].x; s.y += t[j].y; s.z += t[j].z; s.w += t[j].w; }
    }
    if (e < n4) {
      if (se.bias) {
        const int64_t i = e / se.J4, j4 = e - i * se.J4;
        const float4 b = reinterpret_cast<const float4*>(se.bias)[j4];
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        if (se.res) {
          const float4 r = *reinterpret_cast<const float4*>(se.res + i * se.ldr + 4 * j4);
          s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
        }
      }
      reinterpret_cast<float4*>(out)[e] = s;
    } else {
      reinterpret_cast<float4*>(out2)[e - n4] = s;
    }
  }
}

// Scratch of the split tail, owned by the caller of the launches that follow on this thread (hig_gemm_set_tail_scratch):
// HIG_GEMM_TAIL_CNT_BYTES of tickets (zero on first use) followed by the parked partial sums.
unsigned long long* g_gemm_stamps = nullptr;   // diagnostic only (hig_gemm_debug_stamps)
struct TailScratch { unsigned* cnt; float* ws; int64_t ws_bytes; };
thread_local TailScratch g_tail = {nullptr, nullptr, 0};

template <int BI, int BJ, bool X_RS, bool Y_RS, int XF, bool XF_ON_Y, int EPI>
int launch(const hig_gemm_desc& g, int splits, float* slabs, int64_t slab, hipStream_t st, const SplitEpilogue& se) {
  KArgs a;
  a.g = g;
  a.tail_s = 1; a.nmain = 0; a.tail_rem = 0; a.tail_chunk = 0; a.tail_ws = nullptr; a.tail_cnt = nullptr;
  a.stamps = g_gemm_stamps;
  a.xsum = nullptr;
  a.xsum_stride = 0;
  constexpr int epi_flags = 0;   // (a former tuning knob, fixed at the value that won its A/B)
  a.epi_flags = epi_flags;
  constexpr int store_policy = 0;   // (a former tuning knob, fixed at the value that won its A/B)
  a.store_policy = (store_policy && splits == 1 && (((int64_t)g.I - 1) * g.ldc + g.J) * 4 < (1ll << 31) && g.res != g.C) ? store_policy : 0;
  const int nbi = (g.I + BI - 1) / BI;
  a.nbj = (g.J + BJ - 1) / BJ;
  a.ntiles = nbi * a.nbj;
  if (splits <= 1) {
    splits = 1;
    a.r_chunk = ((g.R + BK - 1) / BK) * BK;
    a.slab = 0;
  } else {
    const int per = (g.R + splits - 1) / splits;
    a.r_chunk = ((per + BK - 1) / BK) * BK;
    splits = (g.R + a.r_chunk - 1) / a.r_chunk;
    a.slab = slab;
    a.g.C = slabs;
    a.g.ldc = g.J;  // slabs are dense [I][J]
  }
  if (g.xcolsum) {
    if constexpr (X_RS) {
      if (splits > 1) {   // each split's slab is followed by its I column sums
        a.slab = slab + g.I;
        a.xsum = slabs + slab;
        a.xsum_stride = a.slab;
      } else {
        a.xsum = g.xcolsum;
      }
    }
  }
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vecx = (g.ldx % 4 == 0) && al16(g.X);
  a.vecy = (g.ldy % 4 == 0) && al16(g.Y);
  a.vecc = (a.g.ldc % 4 == 0) && al16(a.g.C) && (a.slab % 4 == 0) &&
           (!g.res || ((g.ldr % 4 == 0) && al16(g.res))) &&
           (!g.aux || ((g.ldaux % 4 == 0) && al16(g.aux)));
  // persistent grid: as many workgroups as fit on the chip at once (LDS-limited), rounded to the 8 XCDs
  constexpr int kStage = ((X_RS ? BK * BI : BI * RC_LD) + (Y_RS ? BK * BJ : BJ * RC_LD));
  constexpr int kLdsF32 = 2 * kStage * 4;
  constexpr int kLdsBf3 = 2 * (BI + BJ) * (BF_LD / 2) * 2 * 4;
  const int lds_bytes = (g.prec == HIG_PREC_BF16X3 && !X_RS && !Y_RS) ? kLdsBf3 : kLdsF32;
  int per_cu = 160 * 1024 / lds_bytes;
  per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
  const int max_vgpr_blocks = (BI * BJ >= 128 * 128) ? 2 : 4;   // 128x128: <=256 VGPRs -> 2 waves/SIMD
  if (per_cu > max_vgpr_blocks) per_cu = max_vgpr_blocks;
  constexpr int forced_per_cu = -1;  // (a former tuning knob, fixed at the value that won its A/B)
  if (forced_per_cu > 0) per_cu = forced_per_cu;
  const bool fast = a.vecx && a.vecy && (g.R % BK == 0) && g.R > 0 &&
                    (!X_RS || (g.I % 4 == 0 && g.I >= 4)) && (!Y_RS || (g.J % 4 == 0 && g.J >= 4));
  // Split tail.  M = 12 544 leaves every GEMM of the model a last round that fills 12-37 % of the chip (N = 512:
  // 1568 64x64 tiles = 6 x 256 + 32): cut those remainder tiles along the reduce range so that the last round costs
  // 1/s of a tile (see KArgs).  Exact-fp32 products, whole rounds in front, reduce slices of >= 2 k-tiles.
  static const int tail_on = getenv("HIG_GEMM_TAIL") ? atoi(getenv("HIG_GEMM_TAIL")) : 1;   // tuning knob
  if constexpr (!X_RS) {
    const int ncu = hig_chip_cus();
    if (tail_on && BI == 64 && BJ == 64 && splits == 1 && fast && g.prec == HIG_PREC_F32 && g_tail.ws && a.ntiles > ncu &&
        a.ntiles % ncu != 0) {
      const int rem = a.ntiles % ncu, nkt = g.R / BK;
      int s = 1;
      while (2 * s * rem <= ncu && nkt % (2 * s) == 0 && nkt / (2 * s) >= 2) s *= 2;
      constexpr int64_t unit_bytes = (int64_t)NTHREADS * (BI / 64) * (BJ / 64) * 16 * 4;
      if (s > 1 && rem <= HIG_GEMM_TAIL_CNT_BYTES / 4 && rem * s * unit_bytes <= g_tail.ws_bytes) {
        a.tail_s = s;
        a.tail_rem = rem;
        a.nmain = a.ntiles - rem;
        a.tail_chunk = g.R / s;
        a.tail_ws = g_tail.ws;
        a.tail_cnt = g_tail.cnt;
        a.ntiles = a.nmain + rem * s;
      }
    }
  }
  int gridx = hig_chip_cus() * per_cu;
  if (gridx > a.ntiles || forced_per_cu == 0) gridx = a.ntiles;   // 0: one workgroup per tile
  bool served = false;
  if constexpr (X_RS && Y_RS && XF == HIG_XF_NONE && EPI == HIG_EPI_NONE) {
    // weight gradients of the exact-fp32 step: the output-stationary kernel with specialised waves (wgrad_wsp32.hip) writes the
    // same slabs (and per-split column sums) this function would
    if (splits > 1 && a.g.C == slabs) {
      const int rc = hig_wgrad_wsp32_try(g, splits, slabs, a.slab, a.xsum, a.xsum_stride, st);
      if (rc < 0) return rc;
      served = rc == HIG_OK;
    }
  }
  if (!served && a.ntiles > 0 && g.R >= 0) {
    // the bf16 product modes exist for aligned reduce-contiguous operands; anything else
    // (F = 150 projections, reduce-slow dgrad / wgrad layouts) runs the exact fp32 kernel
    bool launched = false;
    if constexpr (!X_RS && !Y_RS) {
      if (fast && g.prec == HIG_PREC_BF16X3) {
        hipLaunchKernelGGL((gemm_f32_kernel<BI, BJ, X_RS, Y_RS, XF, XF_ON_Y, EPI, true, HIG_PREC_BF16X3>),
                           dim3(gridx, splits), dim3(NTHREADS), 0, st, a);
        launched = true;
      } else if (fast && g.prec == HIG_PREC_BF16) {
        hipLaunchKernelGGL((gemm_f32_kernel<BI, BJ, X_RS, Y_RS, XF, XF_ON_Y, EPI, true, HIG_PREC_BF16>),
                           dim3(gridx, splits), dim3(NTHREADS), 0, st, a);
        launched = true;
      }
    }
    if (!launched) {
      if (fast)
        hipLaunchKernelGGL((gemm_f32_kernel<BI, BJ, X_RS, Y_RS, XF, XF_ON_Y, EPI, true, HIG_PREC_F32>),
                           dim3(gridx, splits), dim3(NTHREADS), 0, st, a);
      else
        hipLaunchKernelGGL((gemm_f32_kernel<BI, BJ, X_RS, Y_RS, XF, XF_ON_Y, EPI, false, HIG_PREC_F32>),
                           dim3(gridx, splits), dim3(NTHREADS), 0, st, a);
    }
  }
  HIG_CHECK_LAUNCH();
  if (splits > 1) {
    const int64_t n4 = (int64_t)g.I * g.J / 4;
    const int64_t want = (n4 + 255) / 256;
    const int blocks = (int)(want > 2048 ? 2048 : want);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, slabs, splits, a.slab, n4, g.C,
                       a.xsum ? n4 + g.I / 4 : (int64_t)0, a.xsum ? g.xcolsum : (float*)nullptr, se);
    HIG_CHECK_LAUNCH();
  }
  return HIG_OK;
}

// Tile choice.  fp32 MFMA is slow enough (64 cycles per 32x32x2) that every tile shape is
// MFMA-bound, so what matters is how evenly ceil(I/BI)*ceil(J/BJ) tiles spread over 256 CUs:
// M = 12544 = 2^8 * 49 gives 3.06 128x128 tiles per CU for N = 1024 (4 rounds, 77 % balance)
// but 12.25 64x64 tiles (13 rounds, 94 %).  Pick the shape with the smallest
// rounds x tile-area x overhead; smaller tiles pay a little more prologue/epilogue/LDS traffic.
template <bool X_RS, bool Y_RS, int XF, bool XF_ON_Y, int EPI>
int launch_sized(const hig_gemm_desc& g, int splits, float* slabs, int64_t slab, hipStream_t st, const SplitEpilogue& se) {
  if (splits > 1 || X_RS) {  // weight gradients: split-R already supplies the parallelism; tile rule in hig_host.h
    if (wgrad_tile(g.I, g.J, g.prec) == 128) return launch<128, 128, X_RS, Y_RS, XF, XF_ON_Y, EPI>(g, splits, slabs, slab, st, se);
    return launch<64, 64, X_RS, Y_RS, XF, XF_ON_Y, EPI>(g, splits, slabs, slab, st, se);
  }
  struct Cand { int bi, bj; double ovh; };
  // measured (tools/gemm_bench.py): with 64-cycle fp32 MFMAs small tiles cost almost nothing;
  // with the 16x faster bf16 MFMAs the extra LDS/L2 traffic of small tiles shows (x1.1-1.2)
  const bool bf = g.prec != HIG_PREC_F32;
  const Cand cands[4] = {{128, 128, 1.00}, {64, 128, bf ? 1.10 : 1.02}, {128, 64, bf ? 1.12 : 1.03},
                         {64, 64, bf ? 1.20 : 1.04}};
  int best = 0;
  double best_cost = 1e300;
  const int ncu = hig_chip_cus();
  static const int forced = getenv("HIG_GEMM_TILE") ? atoi(getenv("HIG_GEMM_TILE")) : -1;  // tuning knob
  for (int c = 0; c < 4 && forced < 0; ++c) {
    const int64_t tiles = (int64_t)((g.I + cands[c].bi - 1) / cands[c].bi) * ((g.J + cands[c].bj - 1) / cands[c].bj);
    const int64_t rounds = (tiles + ncu - 1) / ncu;
    const double cost = (double)rounds * cands[c].bi * cands[c].bj * cands[c].ovh;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  // exact-fp32 products: 64x64 tiles (four co-resident workgroups per CU) were never beaten by a larger tile on the
  // model's shapes -- B = 32 / 64 forwards, tools/fwd_time.py with HIG_GEMM_TILE forced: 4.07 / 6.90 ms against
  // 4.19-4.67 / 7.5-7.6 ms -- the round-counting model above mis-ranks them by a few per cent, so it only
  // breaks ties for shapes with at most one tile per CU
  constexpr int f32_rule = 1;   // (a former tuning knob, fixed at the value that won its A/B)
  if (!bf && forced < 0 && f32_rule) {
    const int64_t t64 = (int64_t)((g.I + 63) / 64) * ((g.J + 63) / 64);
    if (t64 > ncu) best = 3;
  }
  if (bf && forced < 0) {
    // bf16 products: the MFMA part is short, so per-tile latency and the number of workgroups in flight decide.
    // Measured (tools/fwd_time.py, B = 32 / 64): mixed 64x128 / 128x64 tiles are the worst choice, 128x128 wins
    // once it yields enough tiles to occupy the chip, 64x64 below that.
    constexpr int thr = 300;   // (a former tuning knob, fixed at the value that won its A/B)
    const int64_t t128 = (int64_t)((g.I + 127) / 128) * ((g.J + 127) / 128);
    best = t128 >= thr ? 0 : 3;
  }
  if (forced >= 0) best = forced;
  if (g.row_stats_out || g.row_stats_in) best = 3;   // the LayerNorm fold lives in the 64 x 64 tile's staged epilogue
  switch (best) {
    case 0: return launch<128, 128, X_RS, Y_RS, XF, XF_ON_Y, EPI>(g, splits, slabs, slab, st, se);
    case 1: return launch<64, 128, X_RS, Y_RS, XF, XF_ON_Y, EPI>(g, splits, slabs, slab, st, se);
    case 2: return launch<128, 64, X_RS, Y_RS, XF, XF_ON_Y, EPI>(g, splits, slabs, slab, st, se);
    default: return launch<64, 64, X_RS, Y_RS, XF, XF_ON_Y, EPI>(g, splits, slabs, slab, st, se);
  }
}

}  // namespace

// Internal entry (used by denoiser.hip): splits > 1 routes partial sums over the reduce range
// through `slabs` (splits x I x J floats) and a deterministic slab reduction.
namespace {
int gemm_dispatch(const hig_gemm_desc& g, int splits, float* slabs, hipStream_t st, const SplitEpilogue& se);
}
// hig_host.h: compute units of the current device (cached per device; 256 on an MI355X in SPX mode)
int hig_chip_cus() {
  static int cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = __atomic_load_n(&cache[dev], __ATOMIC_RELAXED);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    static const int forced = getenv("HIG_CHIP_CUS") ? atoi(getenv("HIG_CHIP_CUS")) : 0;   // testing knob: pretend another chip
    if (forced > 0) v = forced;
    __atomic_store_n(&cache[dev], v, __ATOMIC_RELAXED);
  }
  return v;
}

int hig_reduce_slabs(const float* slabs, int splits, int64_t slab, int64_t n, float* out, hipStream_t st) {
  const int64_t n4 = n / 4;
  const int64_t want = (n4 + 255) / 256;
  const int blocks = (int)(want > 2048 ? 2048 : (want < 1 ? 1 : want));
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, slabs, splits, slab, n4, out, (int64_t)0,
                     (float*)nullptr, SplitEpilogue{nullptr, nullptr, 0, 1});
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

int hig_gemm_launch(const hig_gemm_desc& g, int splits, float* slabs, hipStream_t st) {
  return gemm_dispatch(g, splits, slabs, st, SplitEpilogue{nullptr, nullptr, 0, 1});
}
void hig_gemm_set_tail_scratch(void* ws, int64_t bytes) {
  if (!ws || bytes <= HIG_GEMM_TAIL_CNT_BYTES || (reinterpret_cast<uintptr_t>(ws) & 15)) {
    g_tail = TailScratch{nullptr, nullptr, 0};
    return;
  }
  g_tail.cnt = static_cast<unsigned*>(ws);
  g_tail.ws = reinterpret_cast<float*>(static_cast<char*>(ws) + HIG_GEMM_TAIL_CNT_BYTES);
  g_tail.ws_bytes = bytes - HIG_GEMM_TAIL_CNT_BYTES;
}
namespace {
int gemm_dispatch(const hig_gemm_desc& g, int splits, float* slabs, hipStream_t st, const SplitEpilogue& se) {
  HIG_REQUIRE(g.X && g.Y && g.C, "hig_gemm: null operand");
  HIG_REQUIRE(g.I >= 0 && g.J >= 0 && g.R >= 0, "hig_gemm: negative extent");
  if (g.I == 0 || g.J == 0) return HIG_OK;
  if (g.xf != HIG_XF_NONE) {
    // transformed operand: features must come in whole float4 quads
    const int nfeat = g.xf_on_y ? g.J : g.R;
    HIG_REQUIRE(nfeat % 4 == 0, "hig_gemm: fused transform needs feature count %% 4 == 0 (got %d)", nfeat);
    if (g.xf != HIG_XF_SILU) HIG_REQUIRE(g.stats && g.gamma && g.beta, "hig_gemm: LN transform needs stats/gamma/beta");
    if (g.xf == HIG_XF_LN_MOD_SILU) HIG_REQUIRE(g.ss && g.rows_per_sample > 0, "hig_gemm: modulation needs ss");
  }
  const int64_t slab = (int64_t)g.I * g.J;
  if (g.xcolsum)
    HIG_REQUIRE(g.x_rs == 1 && g.prec == HIG_PREC_F32 && g.I % 4 == 0 && (g.xf == HIG_XF_NONE || g.xf_on_y),
                "hig_gemm: xcolsum needs a reduce-slow X operand, fp32 products, I %% 4 == 0");
  if (splits > 1) HIG_REQUIRE(slabs && g.epi == HIG_EPI_NONE && slab % 4 == 0 && g.ldc == g.J,
                              "hig_gemm: split-R needs slabs, EPI_NONE, dense C");
  if (g.row_stats_out || g.row_stats_in) {   // LayerNorm fold: only the LDS-staged epilogue of the 64-column tiles implements it
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool ok = splits <= 1 && g.x_rs == 0 && g.y_rs == 0 && g.xf == HIG_XF_NONE && g.J % 64 == 0 && g.R % 32 == 0 && g.ldc % 4 == 0 &&
                    g.ldx % 4 == 0 && g.ldy % 4 == 0 && a16(g.X) && a16(g.Y) && a16(g.C) && a16(g.bias) &&
                    (g.row_stats_out ? (g.epi == HIG_EPI_BIAS_RES && !g.row_stats_in && g.res && g.ldr % 4 == 0 && a16(g.res) && (reinterpret_cast<uintptr_t>(g.row_stats_out) & 7) == 0)
                                     : (g.epi == HIG_EPI_BIAS && g.ln_colsum && g.R % 128 == 0 && a16(g.row_stats_in) && a16(g.ln_colsum)));
    if (!ok) return hig_set_error(HIG_EUNSUPPORTED, "hig_gemm: LayerNorm-fold operands on a launch that cannot apply them "
                                                    "(needs reduce-contiguous aligned operands, J %% 64 == 0, EPI_BIAS_RES producer / EPI_BIAS consumer with R %% 128 == 0)");
  }
  // exact-fp32 products, K = 512 / 1024, many rows: the weight-stationary kernel with specialised waves (gemm_wsp32.hip)
  if (splits <= 1 && !se.bias) {
    const int rc = hig_gemm_wsp32_try(g, st);
    if (rc != 1) return rc;
    // K = 1536 / 2048 (the data gradient of the stacked q/k/v projection: dqkv (M, 3d) . Wqkv): that kernel's weight panel holds
    // at most K = 1024, so the reduce range goes through it in passes -- C = X[:, :1024] W[:, :1024]^T (+ bias / res), then
    // C += X[:, 1024:] W[:, 1024:]^T with C as its own residual (a lane re-reads exactly the element it stores two tiles later).
    // 153 us for the two passes at M = 12 544 against 197 us on the tiled kernel.
    if (hig_gemm_wsp32_active() && g.prec == HIG_PREC_F32 && !g.x_rs && !g.y_rs && g.xf == HIG_XF_NONE && !g.xcolsum && !g.row_stats_in &&
        !g.row_stats_out && g.R > 1024 && g.R <= 2048 && g.R % 512 == 0 && g.I >= 2048 && g.J % 64 == 0 &&
        (g.epi == HIG_EPI_NONE || g.epi == HIG_EPI_RES || g.epi == HIG_EPI_BIAS || g.epi == HIG_EPI_BIAS_RES)) {
      hig_gemm_desc p1 = g;
      p1.R = 1024;
      const int rc1 = hig_gemm_wsp32_try(p1, st);
      if (rc1 < 0) return rc1;
      if (rc1 == HIG_OK) {
        hig_gemm_desc p2 = g;
        p2.X = g.X + 1024; p2.Y = g.Y + 1024; p2.R = g.R - 1024;
        p2.epi = HIG_EPI_RES; p2.bias = nullptr; p2.res = g.C; p2.ldr = g.ldc;
        const int rc2 = hig_gemm_wsp32_try(p2, st);
        if (rc2 != 1) return rc2;
        return hig_set_error(HIG_EHIP, "hig_gemm: second reduce pass declined after the first was launched");
      }
    }
  }
#define CASE(xrs, yrs, xfv, ony, epiv)                                                   \
  if (g.x_rs == xrs && g.y_rs == yrs && g.xf == xfv && (g.xf == HIG_XF_NONE || g.xf_on_y == ony) && \
      g.epi == epiv)                                                                     \
    return launch_sized<xrs, yrs, xfv, ony, epiv>(g, splits, slabs, slab, st, se);
#ifdef HIG_GEMM_PROBE  // compile-time aid: build a single combination
  CASE(0, 0, HIG_XF_NONE, 0, HIG_EPI_BIAS_GELU)
  return HIG_EUNSUPPORTED;
#endif
  // forward (activations x weight^T)
  CASE(0, 0, HIG_XF_NONE, 0, HIG_EPI_NONE)
  CASE(0, 0, HIG_XF_NONE, 0, HIG_EPI_BIAS)
  CASE(0, 0, HIG_XF_NONE, 0, HIG_EPI_BIAS_GELU)
  CASE(0, 0, HIG_XF_NONE, 0, HIG_EPI_BIAS_POS)
  CASE(0, 0, HIG_XF_NONE, 0, HIG_EPI_BIAS_RES)
  CASE(0, 0, HIG_XF_NONE, 0, HIG_EPI_RES)    // dgrad through a transposed weight copy
  CASE(0, 0, HIG_XF_NONE, 0, HIG_EPI_DGELU)
  CASE(0, 0, HIG_XF_LN, 0, HIG_EPI_BIAS)
  CASE(0, 0, HIG_XF_LN_MOD_SILU, 0, HIG_EPI_BIAS_RES)
  CASE(0, 0, HIG_XF_SILU, 0, HIG_EPI_BIAS)
  CASE(0, 0, HIG_XF_SILU, 0, HIG_EPI_BIAS_RES)
  CASE(0, 0, HIG_XF_SILU, 0, HIG_EPI_NONE)   // split-R partials of the few-row GEMMs
  // dgrad (dC x weight)
  CASE(0, 1, HIG_XF_NONE, 0, HIG_EPI_NONE)
  CASE(0, 1, HIG_XF_NONE, 0, HIG_EPI_RES)
  CASE(0, 1, HIG_XF_NONE, 0, HIG_EPI_DGELU)
  // wgrad (dC^T x activations)
  CASE(1, 1, HIG_XF_NONE, 0, HIG_EPI_NONE)
  CASE(1, 1, HIG_XF_LN, 1, HIG_EPI_NONE)
  CASE(1, 1, HIG_XF_LN_MOD_SILU, 1, HIG_EPI_NONE)
  CASE(1, 1, HIG_XF_SILU, 1, HIG_EPI_NONE)
#undef CASE
  return hig_set_error(HIG_EUNSUPPORTED, "hig_gemm: combination x_rs=%d y_rs=%d xf=%d on_y=%d epi=%d not built",
                       g.x_rs, g.y_rs, g.xf, g.xf_on_y, g.epi);
}
}  // namespace

// Few-row GEMMs (I <= 64: the per-sample time / text embedding MLP and the stacked stylization `emb_layers`):
// one row tile, so the launch has only J / 64 workgroups and each streams its whole K range of the weight alone --
// weight-bandwidth bound at a fraction of the HBM rate (B x 24576 x 2048: 200 MB in 102 us; B x 2048 x 2048 on 32
// workgroups: 60 us).  Split the reduce range so ~1024 workgroups stream the weight together, sum the partial outputs
// from `scratch` (deterministic slab reduction), which also applies bias / residual.
// EPI_BIAS / EPI_BIAS_RES, reduce-contiguous operands, dense C; falls back to the plain launch when that does not apply.
int hig_gemm_few_rows(const hig_gemm_desc& g, float* scratch, int64_t scratch_floats, hipStream_t st) {
  static const int enabled = getenv("HIG_FEW_ROWS_SPLIT") ? atoi(getenv("HIG_FEW_ROWS_SPLIT")) : 1;   // tuning knob
  const int64_t out = (int64_t)g.I * g.J;
  const bool ok = enabled && scratch && g.I > 0 && g.I <= 64 && g.x_rs == 0 && g.y_rs == 0 && g.ldc == g.J && g.J % 4 == 0 &&
                  (g.epi == HIG_EPI_BIAS || g.epi == HIG_EPI_BIAS_RES) && g.bias &&
                  (g.xf == HIG_XF_NONE || g.xf == HIG_XF_SILU) && g.R % 32 == 0 &&
                  (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
                  (g.epi != HIG_EPI_BIAS_RES || (g.res && (reinterpret_cast<uintptr_t>(g.res) & 15) == 0 && g.ldr % 4 == 0));
  int splits = 1;
  if (ok) {
    const int64_t tiles = (g.J + 63) / 64;
    int64_t s = 1024 / tiles;
    if (s > g.R / 64) s = g.R / 64;
    if (s > 16) s = 16;
    if (s * out > scratch_floats) s = scratch_floats / out;
    splits = (int)(s < 1 ? 1 : s);
  }
  if (splits <= 1) return hig_gemm_launch(g, 1, nullptr, st);
  hig_gemm_desc p = g;
  p.epi = HIG_EPI_NONE;
  p.bias = nullptr;
  p.res = nullptr;
  return gemm_dispatch(p, splits, scratch, st,
                       SplitEpilogue{g.bias, g.epi == HIG_EPI_BIAS_RES ? g.res : nullptr, g.ldr, g.J / 4});
}

extern "C" int hig_gemm(const hig_gemm_desc* g, hig_stream_t stream) {
  HIG_REQUIRE(g, "hig_gemm: null descriptor");
  HIG_REQUIRE(g->prec == HIG_PREC_F32 || g->prec == HIG_PREC_BF16X3 || g->prec == HIG_PREC_BF16,
              "hig_gemm: unknown prec %d", g->prec);
  return hig_gemm_launch(*g, 1, nullptr, hig_stream(stream));
}

// Diagnostic: thread 0 of every workgroup (< 4096) writes s_memtime stamps of its SECOND tile to buf[block * 8 + k]
// (k: 0 tile start, 1 first k-tile staged, 2 main loop done, 3 next tile's first fetch issued, 4 tile staged in LDS,
// 5 rows read / epilogue applied / stores issued, 6 closing barrier passed).
// NULL switches it off.  Never part of a timed run.
extern "C" int hig_gemm_debug_stamps(void* buf) {
  g_gemm_stamps = static_cast<unsigned long long*>(buf);
  return HIG_OK;
}

extern "C" int64_t hig_gemm_tail_ws_bytes(void) { return HIG_GEMM_TAIL_BYTES; }

extern "C" int hig_gemm_ws(const hig_gemm_desc* g, void* ws, int64_t ws_bytes, hig_stream_t stream) {
  HIG_REQUIRE(g, "hig_gemm_ws: null descriptor");
  HIG_REQUIRE(g->prec == HIG_PREC_F32 || g->prec == HIG_PREC_BF16X3 || g->prec == HIG_PREC_BF16,
              "hig_gemm_ws: unknown prec %d", g->prec);
  HIG_REQUIRE(!ws || (ws_bytes >= HIG_GEMM_TAIL_BYTES && (reinterpret_cast<uintptr_t>(ws) & 15) == 0),
              "hig_gemm_ws: scratch must be 16-byte aligned and hig_gemm_tail_ws_bytes() long");
  hig_gemm_set_tail_scratch(ws, ws_bytes);
  const int rc = hig_gemm_launch(*g, 1, nullptr, hig_stream(stream));
  hig_gemm_set_tail_scratch(nullptr, 0);
  return rc;
}

// Weight-gradient form of hig_gemm as the backward launches it: the reduce range split over `splits` partial outputs in
// `slabs` (+ the per-split column sums of X behind each slab when g->xcolsum is set) and the deterministic slab
// reduction.  splits == 0 asks for the library's own rule (wgrad_splits: tiles x splits fills the chip).
extern "C" int64_t hig_gemm_split_scratch_floats(const hig_gemm_desc* g, int32_t splits) {
  if (!g || g->I <= 0 || g->J <= 0) return -1;
  const int64_t per = (int64_t)g->I * g->J + g->I;
  const int64_t cap = splits > 0 ? splits : (g->R / 256 > 1 ? g->R / 256 : 1);
  return per * cap;
}
extern "C" int hig_gemm_split(const hig_gemm_desc* g, int32_t splits, float* slabs, int64_t slab_floats,
                              hig_stream_t stream) {
  HIG_REQUIRE(g && slabs && splits >= 0, "hig_gemm_split: bad arguments");
  HIG_REQUIRE(g->prec == HIG_PREC_F32 || g->prec == HIG_PREC_BF16X3 || g->prec == HIG_PREC_BF16,
              "hig_gemm_split: unknown prec %d", g->prec);
  int s = splits > 0 ? splits : wgrad_splits(g->I, g->J, g->R, slab_floats, g->prec);
  HIG_REQUIRE((int64_t)s * ((int64_t)g->I * g->J + g->I) <= slab_floats, "hig_gemm_split: %d splits need %lld scratch floats, got %lld",
              s, (long long)((int64_t)s * ((int64_t)g->I * g->J + g->I)), (long long)slab_floats);
  return hig_gemm_launch(*g, s, slabs, hig_stream(stream));
}
