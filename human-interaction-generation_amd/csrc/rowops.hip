// Row-wise HBM-bound kernels: LayerNorm statistics, LayerNorm / stylization backward with its
// broadcast reductions, column sums (bias gradients), timestep embedding.
// Reference arithmetic: nn.LayerNorm (eps 1e-5, biased variance) as used in
// codes/models/transformer.py:68,94,127-128 and StylizationBlock.forward (:81-85).
//
// Mapping: one 64-lane wave per row, lanes stride the row in float4 (16 B/lane, coalesced 1 KiB
// per wave-instruction); cross-lane sums by DPP/shuffle butterflies; no LDS for the row itself.
#include <stdlib.h>

#include <algorithm>

#include "hig_common.h"
#include "hig_host.h"

namespace {

constexpr int WAVES = 4;

__global__ __launch_bounds__(256) void rowstats_kernel(const float* __restrict__ x, int64_t ldx,
                                                       int64_t rows, int n, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  const bool vec = (n % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  float s = 0.f;
  if (vec) {
    for (int c = 4 * lane; c < n; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(xr + c);
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int c = lane; c < n; c += 64) s += xr[c];
  }
  const float mean = wave_sum(s) / (float)n;
  float q = 0.f;
  if (vec) {
    for (int c = 4 * lane; c < n; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(xr + c);
      const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  } else {
    for (int c = lane; c < n; c += 64) {
      const float a = xr[c] - mean;
      q += a * a;
    }
  }
  const float var = wave_sum(q) / (float)n;
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rsqrtf(var + 1e-5f);
  }
}

// Fused StylizationBlock front half (transformer.py:81-84 + the SiLU of out_layers):
//   a[m] = silu( LN(x[m]) * (1 + scale[b(m)]) + shift[b(m)] ),  stats[m] = (mean, rstd)
// One wave per row, the row stays in registers between the statistics and the transform, so x
// is read once and a written once (the GEMM that consumes `a` is then a plain contraction: the
// per-element exp/rcp of the SiLU is paid once per element instead of once per N-tile column).
template <int NIT, bool MOD>
__global__ __launch_bounds__(256) void ln_mod_silu_kernel(
    const float* __restrict__ x, int64_t ldx, int64_t rows, int n, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ ss, int64_t ss_ld, int shift_off,
    int rows_per_sample, float* __restrict__ a, int64_t lda, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float4 v[NIT];
  // gamma / beta / scale / shift of this lane's columns are requested WITH the row, not behind its statistics: issued where they
  // are used (inside `if (c < n)`, one slice after the other) every wave paid two more memory latencies in a row -- row, then
  // the first slice's parameters, then the second's (disassembly; round 6)
  float4 g4[NIT], b4[NIT], sc4[MOD ? NIT : 1], sh4[MOD ? NIT : 1];
  const float* ssrow = MOD ? ss + (row / rows_per_sample) * ss_ld : nullptr;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = 4 * lane + 256 * it;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    g4[it] = b4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (MOD) sc4[it] = sh4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < n) {
      v[it] = *reinterpret_cast<const float4*>(xr + c);
      g4[it] = *reinterpret_cast<const float4*>(gamma + c);
      b4[it] = *reinterpret_cast<const float4*>(beta + c);
      if constexpr (MOD) {
        sc4[it] = *reinterpret_cast<const float4*>(ssrow + c);
        sh4[it] = *reinterpret_cast<const float4*>(ssrow + shift_off + c);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) s += (v[it].x + v[it].y) + (v[it].z + v[it].w);
  const float mean = wave_sum(s) / (float)n;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = 4 * lane + 256 * it;
    if (c < n) {
      const float d0 = v[it].x - mean, d1 = v[it].y - mean, d2 = v[it].z - mean, d3 = v[it].w - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)n + 1e-5f);
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
  if (!MOD) {  // plain LayerNorm (text head): a = LN(x) * gamma + beta
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * lane + 256 * it;
      if (c < n) {
        float4 o;
        o.x = (v[it].x - mean) * rstd * g4[it].x + b4[it].x;
        o.y = (v[it].y - mean) * rstd * g4[it].y + b4[it].y;
        o.z = (v[it].z - mean) * rstd * g4[it].z + b4[it].z;
        o.w = (v[it].w - mean) * rstd * g4[it].w + b4[it].w;
        *reinterpret_cast<float4*>(a + row * lda + c) = o;
      }
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = 4 * lane + 256 * it;
    if (c < n) {
      const float4 sc = sc4[MOD ? it : 0], sh = sh4[MOD ? it : 0];
      float4 o;
      o.x = hig_silu(((v[it].x - mean) * rstd * g4[it].x + b4[it].x) * (1.0f + sc.x) + sh.x);
      o.y = hig_silu(((v[it].y - mean) * rstd * g4[it].y + b4[it].y) * (1.0f + sc.y) + sh.y);
      o.z = hig_silu(((v[it].z - mean) * rstd * g4[it].z + b4[it].z) * (1.0f + sc.z) + sh.z);
      o.w = hig_silu(((v[it].w - mean) * rstd * g4[it].w + b4[it].w) * (1.0f + sc.w) + sh.w);
      *reinterpret_cast<float4*>(a + row * lda + c) = o;
    }
  }
}

// Backward of a = [silu](LN(x) * (1 + scale) + shift).  grid = (samples, splits); wave w of split
// s owns rows s*4 + w, + 4*splits, ... of its sample.  NIT = ceil(n / 256) float4 per lane.
template <int NIT, bool MOD_SILU>
__global__ __launch_bounds__(256) void ln_bwd_kernel(
    const float* __restrict__ da, int64_t ldda, const float* __restrict__ x, int64_t ldx,
    const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ ss, int64_t ss_ld, int shift_off, const float* __restrict__ res,
    int64_t ldr, float* __restrict__ dx, int64_t lddx, int n, int rows_per_sample,
    float* __restrict__ partial) {
  __shared__ float red[WAVES][4][64 * 4];  // one NIT slice at a time
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const float inv_n = 1.0f / (float)n;
  float4 g4[NIT], b4[NIT], sc4[NIT], sh4[NIT];
  float4 a_dg[NIT], a_db[NIT], a_dsc[NIT], a_dsh[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = 4 * lane + 256 * it;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    g4[it] = b4[it] = sc4[it] = sh4[it] = z;
    a_dg[it] = a_db[it] = a_dsc[it] = a_dsh[it] = z;
    if (c < n) {
      g4[it] = *reinterpret_cast<const float4*>(gamma + c);
      b4[it] = *reinterpret_cast<const float4*>(beta + c);
      if (MOD_SILU) {
        sc4[it] = *reinterpret_cast<const float4*>(ss + (int64_t)b * ss_ld + c);
        sh4[it] = *reinterpret_cast<const float4*>(ss + (int64_t)b * ss_ld + shift_off + c);
      }
    }
  }
  // The NEXT row of this wave is requested before the current one is touched (round 6): a wave walks ~6 rows, and with the loads
  // at the top of each trip every row paid a full memory latency on its own (47 us per launch at config 2 = 1.6 TB/s; 24 launches
  // per training step).  Rows past the end re-read the last row (never used).
  const int rstep = WAVES * nsplit;
  const int rl0 = split * WAVES + wave;
  float4 pxv[NIT], pdav[NIT];
  float pmean = 0.f, prstd = 0.f;
  auto fetch = [&](int rl) {
    const int64_t row = (int64_t)b * rows_per_sample + min(rl, rows_per_sample - 1);
    pmean = stats[2 * row];
    prstd = stats[2 * row + 1];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * lane + 256 * it;
      pxv[it] = pdav[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < n) {
        pxv[it] = *reinterpret_cast<const float4*>(x + row * ldx + c);
        pdav[it] = *reinterpret_cast<const float4*>(da + row * ldda + c);
      }
    }
  };
  if (rl0 < rows_per_sample) fetch(rl0);
  for (int rl = rl0; rl < rows_per_sample; rl += rstep) {
    const int64_t row = (int64_t)b * rows_per_sample + rl;
    const float mean = pmean, rstd = prstd;
    float4 cxv[NIT], cdav[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) { cxv[it] = pxv[it]; cdav[it] = pdav[it]; }
    fetch(rl + rstep);
    float4 xh[NIT], dxh[NIT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * lane + 256 * it;
      xh[it] = dxh[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < n) {
        const float4 xv = cxv[it];
        const float4 dav = cdav[it];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dav.x, dav.y, dav.z, dav.w};
        const float gs[4] = {g4[it].x, g4[it].y, g4[it].z, g4[it].w};
        const float bs[4] = {b4[it].x, b4[it].y, b4[it].z, b4[it].w};
        const float scs[4] = {sc4[it].x, sc4[it].y, sc4[it].z, sc4[it].w};
        const float shs[4] = {sh4[it].x, sh4[it].y, sh4[it].z, sh4[it].w};
        float xho[4], dxo[4], dgo[4], dbo[4], dsco[4], dsho[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xhat = (xs[e] - mean) * rstd;
          const float nrm = xhat * gs[e] + bs[e];
          float dn;
          if (MOD_SILU) {
            const float u = nrm * (1.0f + scs[e]) + shs[e];
            const float du = ds[e] * hig_dsilu(u);
            dsho[e] = du;
            dsco[e] = du * nrm;
            dn = du * (1.0f + scs[e]);
          } else {
            dsho[e] = dsco[e] = 0.f;
            dn = ds[e];
          }
          dgo[e] = dn * xhat;
          dbo[e] = dn;
          xho[e] = xhat;
          dxo[e] = dn * gs[e];
          s1 += dxo[e];
          s2 += dxo[e] * xhat;
        }
        xh[it] = make_float4(xho[0], xho[1], xho[2], xho[3]);
        dxh[it] = make_float4(dxo[0], dxo[1], dxo[2], dxo[3]);
        a_dg[it].x += dgo[0]; a_dg[it].y += dgo[1]; a_dg[it].z += dgo[2]; a_dg[it].w += dgo[3];
        a_db[it].x += dbo[0]; a_db[it].y += dbo[1]; a_db[it].z += dbo[2]; a_db[it].w += dbo[3];
        if (MOD_SILU) {
          a_dsc[it].x += dsco[0]; a_dsc[it].y += dsco[1]; a_dsc[it].z += dsco[2]; a_dsc[it].w += dsco[3];
          a_dsh[it].x += dsho[0]; a_dsh[it].y += dsho[1]; a_dsh[it].z += dsho[2]; a_dsh[it].w += dsho[3];
        }
      }
    }
    s1 = wave_sum(s1) * inv_n;
    s2 = wave_sum(s2) * inv_n;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * lane + 256 * it;
      if (c < n) {
        float4 o;
        o.x = rstd * (dxh[it].x - s1 - xh[it].x * s2);
        o.y = rstd * (dxh[it].y - s1 - xh[it].y * s2);
        o.z = rstd * (dxh[it].z - s1 - xh[it].z * s2);
        o.w = rstd * (dxh[it].w - s1 - xh[it].w * s2);
        if (res) {
          const float4 r4 = *reinterpret_cast<const float4*>(res + row * ldr + c);
          o.x += r4.x; o.y += r4.y; o.z += r4.z; o.w += r4.w;
        }
        *reinterpret_cast<float4*>(dx + row * lddx + c) = o;
      }
    }
  }
  // cross-wave reduction of the column accumulators -> partial[(b*nsplit+split)][4][n]
  float* pout = partial + ((int64_t)b * nsplit + split) * 4 * n;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    __syncthreads();
    *reinterpret_cast<float4*>(&red[wave][0][4 * lane]) = a_dg[it];
    *reinterpret_cast<float4*>(&red[wave][1][4 * lane]) = a_db[it];
    *reinterpret_cast<float4*>(&red[wave][2][4 * lane]) = a_dsc[it];
    *reinterpret_cast<float4*>(&red[wave][3][4 * lane]) = a_dsh[it];
    __syncthreads();
    // 4 quantities x 256 columns = 1024 sums, 4 per thread
    for (int e = threadIdx.x; e < 4 * 256; e += 256) {
      const int qn = e >> 8, cl = e & 255, c = cl + 256 * it;
      if (c < n) {
        float s = red[0][qn][cl];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += red[w][qn][cl];
        pout[(int64_t)qn * n + c] = s;
      }
    }
  }
}

// out[c] = sum_{r<nr} in[(r*rstride) + c]  (column reduction of small partial tables).
// 64 columns x 16 row lanes per block; 4 independent accumulators keep 4 loads in flight.
// Columns [0, n_first) go to `out`, columns [n_first, n) to `out2` (two gradients from one partial table).
__global__ __launch_bounds__(1024) void colreduce_kernel(const float* __restrict__ in, int nr,
                                                         int64_t rstride, int n, float* __restrict__ out,
                                                         int n_first = 1 << 30, float* __restrict__ out2 = nullptr) {
  __shared__ float red[16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < n) {
    int r = ry;
    for (; r + 48 < nr; r += 64) {
      s0 += in[(int64_t)r * rstride + c];
      s1 += in[(int64_t)(r + 16) * rstride + c];
      s2 += in[(int64_t)(r + 32) * rstride + c];
      s3 += in[(int64_t)(r + 48) * rstride + c];
    }
    for (; r < nr; r += 16) s0 += in[(int64_t)r * rstride + c];
  }
  red[ry][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ry == 0 && c < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    if (c < n_first) out[c] = t; else out2[c - n_first] = t;
  }
}
// dss[b][c] = sum_s partial[b][s][2][c];  dss[b][shift_off + c] = sum_s partial[b][s][3][c]
__global__ void dss_reduce_kernel(const float* __restrict__ partial, int nsplit, int n,
                                  int shift_off, float* __restrict__ dss, int64_t dss_ld) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= n) return;
  float s2 = 0.f, s3 = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* p = partial + ((int64_t)b * nsplit + s) * 4 * n;
    s2 += p[2 * (int64_t)n + c];
    s3 += p[3 * (int64_t)n + c];
  }
  dss[(int64_t)b * dss_ld + c] = s2;
  dss[(int64_t)b * dss_ld + shift_off + c] = s3;
}

// Both reductions of a stylization LayerNorm backward in one launch: blocks [0, nb_col) are colreduce_kernel blocks over
// the [dgamma | dbeta] columns of the partial table, the blocks behind them do dss_reduce_kernel's (sample, column) sums.
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int samples, int nsplit,
                                                             int n, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int shift_off, float* __restrict__ dss, int64_t dss_ld,
                                                             int nb_col) {
  if ((int)blockIdx.x < nb_col) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx, nr = samples * nsplit;
    const int64_t rstride = (int64_t)4 * n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < 2 * n) {
      int r = ry;
      for (; r + 48 < nr; r += 64) {
        s0 += partial[(int64_t)r * rstride + c];
        s1 += partial[(int64_t)(r + 16) * rstride + c];
        s2 += partial[(int64_t)(r + 32) * rstride + c];
        s3 += partial[(int64_t)(r + 48) * rstride + c];
      }
      for (; r < nr; r += 16) s0 += partial[(int64_t)r * rstride + c];
    }
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && c < 2 * n) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][cx];
      if (c < n) dgamma[c] = t; else dbeta[c - n] = t;
    }
    return;
  }
  const int64_t idx = (int64_t)(blockIdx.x - nb_col) * 1024 + threadIdx.x;
  const int b = (int)(idx / n), c = (int)(idx % n);
  if (b >= samples) return;
  float s2 = 0.f, s3 = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* p = partial + ((int64_t)b * nsplit + s) * 4 * n;
    s2 += p[2 * (int64_t)n + c];
    s3 += p[3 * (int64_t)n + c];
  }
  dss[(int64_t)b * dss_ld + c] = s2;
  dss[(int64_t)b * dss_ld + shift_off + c] = s3;
}

// ln_bwd_reduce_kernel for up to HIG_LN_RB_MAX LayerNorm backward calls in ONE launch (the bf16 training step: a decoder layer's
// five or six calls keep their partial tables and are reduced together at the end of the layer: inside the captured step every
// launch costs 2-3 us of dependency latency, and these reductions are ~1 us of work each).  Entries without `dss` (the plain
// LayerNorms) have no blocks of the second kind.
struct LnReduceBatch {
  hig_ln_reduce e[HIG_LN_RB_MAX];
  int block0[HIG_LN_RB_MAX + 1];
  int n;
};
__global__ __launch_bounds__(1024) void ln_bwd_reduce_batch_kernel(const LnReduceBatch bt) {
  int m = 0;
  while (m + 1 < bt.n && (int)blockIdx.x >= bt.block0[m + 1]) ++m;
  const hig_ln_reduce& q = bt.e[m];
  const int blk = blockIdx.x - bt.block0[m];
  const float* __restrict__ partial = q.partial;
  const int n = q.n;
  if (blk < q.nb_col) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blk * 64 + cx, nr = q.samples * q.nsplit;
    const int64_t rstride = (int64_t)4 * n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < 2 * n) {
      int r = ry;
      for (; r + 48 < nr; r += 64) {
        s0 += partial[(int64_t)r * rstride + c];
        s1 += partial[(int64_t)(r + 16) * rstride + c];
        s2 += partial[(int64_t)(r + 32) * rstride + c];
        s3 += partial[(int64_t)(r + 48) * rstride + c];
      }
      for (; r < nr; r += 16) s0 += partial[(int64_t)r * rstride + c];
    }
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && c < 2 * n) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][cx];
      if (c < n) q.dgamma[c] = t; else q.dbeta[c - n] = t;
    }
    return;
  }
  const int64_t idx = (int64_t)(blk - q.nb_col) * 1024 + threadIdx.x;
  const int b = (int)(idx / n), c = (int)(idx % n);
  if (b >= q.samples) return;
  float s2 = 0.f, s3 = 0.f;
  for (int s = 0; s < q.nsplit; ++s) {
    const float* p = partial + ((int64_t)b * q.nsplit + s) * 4 * n;
    s2 += p[2 * (int64_t)n + c];
    s3 += p[3 * (int64_t)n + c];
  }
  q.dss[(int64_t)b * q.dss_ld + c] = s2;
  q.dss[(int64_t)b * q.dss_ld + q.shift_off + c] = s3;
}

// partial[chunk][c] = sum over this chunk's rows of x[row][c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ldx,
                                                     int64_t rows, int n, float* __restrict__ partial) {
  __shared__ float red[WAVES][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256 + 4 * lane;
  const int chunk = blockIdx.y, nchunk = gridDim.y;
  const bool vec = (n % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t r = (int64_t)chunk * WAVES + wave; r < rows; r += (int64_t)WAVES * nchunk) {
    const float* p = x + r * ldx + c0;
    if (vec) {
      if (c0 < n) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    } else {
      if (c0 < n) s.x += p[0];
      if (c0 + 1 < n) s.y += p[1];
      if (c0 + 2 < n) s.z += p[2];
      if (c0 + 3 < n) s.w += p[3];
    }
  }
  *reinterpret_cast<float4*>(&red[wave][4 * lane]) = s;
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < n) {
    float t = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) t += red[w][threadIdx.x];
    partial[(int64_t)chunk * n + c] = t;
  }
}

// dst[c][r] = [LN](src[r][c])   (64x64 tiles through LDS; both sides coalesced).
// Used to give the weight-gradient / data-gradient GEMMs reduce-contiguous operands.
template <bool LN>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int64_t lds_, int rows,
                                                        int cols, float* __restrict__ dst, int64_t ldd,
                                                        const float* __restrict__ stats,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = src[(int64_t)r * lds_ + c];
      if (LN) v = (v - stats[2 * (int64_t)r]) * stats[2 * (int64_t)r + 1] * gamma[c] + beta[c];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(int64_t)c * ldd + r] = tile[tx][i];
  }
}

// Several dense transposes in ONE launch (all the W -> W^T of a decoder layer's backward: eight to ten
// 1-4 MB matrices cost ~10 us each as separate launches, mostly launch + tail).  The descriptors travel
// by value in the kernel arguments; block -> (matrix, 64x64 tile) through a prefix table.
constexpr int TRB_MAX = 12;
struct TrBatch {
  const float* src[TRB_MAX];
  float* dst[TRB_MAX];
  int rows[TRB_MAX], cols[TRB_MAX];
  int tile0[TRB_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void transpose_batch_kernel(const TrBatch b) {
  __shared__ float tile[64][65];
  int m = 0;
  while (m + 1 < b.n && (int)blockIdx.x >= b.tile0[m + 1]) ++m;
  const int rows = b.rows[m], cols = b.cols[m];
  const int lt = blockIdx.x - b.tile0[m], tcn = (cols + 63) / 64;
  const int c0 = (lt % tcn) * 64, r0 = (lt / tcn) * 64;
  const float* __restrict__ src = b.src[m];
  float* __restrict__ dst = b.dst[m];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(int64_t)c * rows + r] = tile[tx][i];
  }
}

template <typename TO>
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, int B, int d,
                                          TO* __restrict__ out) {
  const int half = d / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * d) return;
  const int b = idx / d, c = idx % d;
  float v = 0.f;
  if (c < 2 * half) {
    const int i = c < half ? c : c - half;
    // freqs = exp(-ln(10000) * i / half) in fp32, args = float(t) * freqs  (transformer.py:25-28)
    const float e32 = -9.210340371976184f * (float)i / (float)half;
    const float freq = (float)exp((double)e32);  // correctly rounded fp32 exp
    const float arg = (float)t[b] * freq;
    v = c < half ? cosf(arg) : sinf(arg);
  }
  out[idx] = (TO)v;
}


// ---------------------------------------------------------------------------------------------------------------------
// Row kernels of the bf16-storage TRAINING step (round 4): the same reductions as above over bf16 rows.
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 rbf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4g(const __bf16* p) {
  const rbf16x4 t = *reinterpret_cast<const rbf16x4*>(p);
  return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
}
// (raw quads: what a lane holds between the request and the first use of a row -- bf16 stays packed)
template <typename T> struct RawQuad;
template <> struct RawQuad<float> { typedef float4 type; };
template <> struct RawQuad<__bf16> { typedef uint2 type; };
__device__ __forceinline__ float4 ldraw(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 ldraw(const __bf16* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 cvtraw(const float4& v) { return v; }
__device__ __forceinline__ float4 cvtraw(const uint2& v) {
  return make_float4(__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                     __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u));
}
typedef unsigned int ru32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int ru32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ typename RawQuad<T>::type ldraw_buf(__amdgpu_buffer_rsrc_t rs, int byte_off);
template <> __device__ __forceinline__ uint2 ldraw_buf<__bf16>(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  const ru32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 0);
  return make_uint2(v.x, v.y);
}
template <> __device__ __forceinline__ float4 ldraw_buf<float>(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}
template <typename T> __device__ __forceinline__ void straw_buf(__amdgpu_buffer_rsrc_t rs, int byte_off, const float4& v);
template <> __device__ __forceinline__ void straw_buf<float>(__amdgpu_buffer_rsrc_t rs, int byte_off, const float4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ru32x4, v), rs, byte_off, 0, 0);
}
template <> __device__ __forceinline__ void straw_buf<__bf16>(__amdgpu_buffer_rsrc_t rs, int byte_off, const float4& v) {
  const rbf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ru32x2, o), rs, byte_off, 0, 0);
}
__device__ __forceinline__ void st4g(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4g(__bf16* p, const float4& v) {
  *reinterpret_cast<rbf16x4*>(p) = rbf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}

// ln_bwd_kernel with bf16 upstream gradients, rows in TX (bf16 activations, or the fp32 text embeddings), residual /
// result in TD (bf16: the gradient of the residual stream; fp32: d(xf_out)), and the LayerNorm statistics RECOMPUTED from
// the row the kernel reads anyway (two passes over the registers: the forward of this mode keeps no statistics).
// A wave works on RPI rows at a time (independent dependency chains: load -> mean -> variance -> gradients -> two row sums
// -> store is ~2 us of latency per row, and a workgroup slot holds only a few waves: one row at a time left the kernel at
// 1.6-1.9 TB/s): all loads of the RPI rows are issued before the first reduction.
template <int NIT, bool MOD_SILU, typename TX, typename TD, int RPI, int NWV = WAVES>
__global__ __launch_bounds__(64 * NWV) void ln_bwd16_kernel(
    const __bf16* __restrict__ da, int64_t ldda, const TX* __restrict__ x, int64_t ldx,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ ss, int64_t ss_ld, int shift_off, const TD* __restrict__ res,
    int64_t ldr, TD* __restrict__ dx, int64_t lddx, int n, int rows_per_sample,
    float* __restrict__ partial) {
  __shared__ float red[NWV][4][64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const float inv_n = 1.0f / (float)n;
  float4 g4[NIT], b4[NIT], sc4[NIT], sh4[NIT];
  float4 a_dg[NIT], a_db[NIT], a_dsc[NIT], a_dsh[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = 4 * lane + 256 * it;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    g4[it] = b4[it] = sc4[it] = sh4[it] = z;
    a_dg[it] = a_db[it] = a_dsc[it] = a_dsh[it] = z;
    if (c < n) {
      g4[it] = *reinterpret_cast<const float4*>(gamma + c);
      b4[it] = *reinterpret_cast<const float4*>(beta + c);
      if (MOD_SILU) {
        sc4[it] = *reinterpret_cast<const float4*>(ss + (int64_t)b * ss_ld + c);
        sh4[it] = *reinterpret_cast<const float4*>(ss + (int64_t)b * ss_ld + shift_off + c);
      }
    }
  }
  const int stride = NWV * nsplit;
  // Software pipeline: the rows of iteration i + 1 are requested (raw: bf16 quads stay packed, two registers) before iteration i
  // is worked on -- with two waves per SIMD and no prefetch the memory pipe idled during the ~1.5 us of arithmetic and
  // reductions per iteration and the kernel ran at ~2.4 TB/s.  The loop body is BRANCH-FREE: rows through buffer descriptors of
  // the sample (an offset beyond the range loads zeros / drops the store without touching memory: rows past the sample's
  // end, columns past n, and a null `res` as a descriptor of zero bytes), so every iteration issues the same number of
  // vector-memory instructions and the compiler's s_waitcnt counts are exact -- with branches around the loads it fell back to
  // vmcnt(0) in front of the arithmetic, i.e. waited for the prefetch it had just issued.
  const int64_t sample0 = (int64_t)b * rows_per_sample;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<TX*>(x + sample0 * ldx), 0,
                                                                        (int)(((int64_t)(rows_per_sample - 1) * ldx + n) * sizeof(TX)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(da + sample0 * ldda), 0,
                                                                        (int)(((int64_t)(rows_per_sample - 1) * ldda + n) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<TD*>(res ? res + sample0 * ldr : dx), 0,
                                                                        res ? (int)(((int64_t)(rows_per_sample - 1) * ldr + n) * sizeof(TD)) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(dx + sample0 * lddx, 0,
                                                                        (int)(((int64_t)(rows_per_sample - 1) * lddx + n) * sizeof(TD)), 0x00020000);
  constexpr int OOB = 0x7ffffff0;                     // (beyond every range: returns zeros, writes nothing)
  bool cok[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) cok[it] = 4 * lane + 256 * it < n;
  typename RawQuad<TX>::type rx[RPI][NIT];
  typename RawQuad<__bf16>::type rda[RPI][NIT];
  typename RawQuad<TD>::type rr[RPI][NIT];
  auto request = [&](int rl0) {
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      const int rl = rl0 + u * stride;
      const bool lv = rl < rows_per_sample;             // (wave-uniform)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = 4 * lane + 256 * it;
        const bool ok = lv && cok[it];
        rx[u][it] = ldraw_buf<TX>(rsx, ok ? (int)((rl * (int)ldx + c) * sizeof(TX)) : OOB);
        rda[u][it] = ldraw_buf<__bf16>(rsd, ok ? (rl * (int)ldda + c) * 2 : OOB);
        rr[u][it] = ldraw_buf<TD>(rsr, ok ? (int)((rl * (int)ldr + c) * sizeof(TD)) : OOB);
      }
    }
  };
  const int rl_first = split * NWV + wave;
  request(rl_first);
  for (int rl0 = rl_first; rl0 < rows_per_sample; rl0 += RPI * stride) {
    float4 xv[RPI][NIT], dav[RPI][NIT], rv[RPI][NIT];
    bool live[RPI];
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      live[u] = rl0 + u * stride < rows_per_sample;                 // (wave-uniform)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        xv[u][it] = cvtraw(rx[u][it]);            // (zeros where the request was out of range)
        dav[u][it] = cvtraw(rda[u][it]);
        rv[u][it] = cvtraw(rr[u][it]);
      }
    }
    request(rl0 + RPI * stride);
    __builtin_amdgcn_sched_barrier(0);      // (the scheduler would sink the requests to the end of the body: no prefetch at all)
    float mean[RPI], rstd[RPI];
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      float sx = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it) sx += (xv[u][it].x + xv[u][it].y) + (xv[u][it].z + xv[u][it].w);
      mean[u] = wave_sum(sx) * inv_n;
    }
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      float sq = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const float a0 = xv[u][it].x - mean[u], a1 = xv[u][it].y - mean[u], a2 = xv[u][it].z - mean[u], a3 = xv[u][it].w - mean[u];
        const float q = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        sq += cok[it] ? q : 0.f;
      }
      rstd[u] = rsqrtf(wave_sum(sq) * inv_n + 1e-5f);
    }
    // Packed fp32 arithmetic (v_pk_fma / v_pk_mul / v_pk_add_f32: two elements per issue slot) and the 1-ulp reciprocal in
    // the SiLU derivative: the kernel is bound by vector instruction issue (~350 instructions per row in the stylization form
    // with scalar arithmetic and a correctly rounded division), not by its 38-51 MB of traffic.
    // Dead lanes (columns past n) and dead rows hold zeros in `dav`: every product below that reaches an accumulator or a row
    // sum has the upstream gradient as a factor, so they add exact zeros.
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 xh[RPI][NIT][2], dxh[RPI][NIT][2];
    float s1[RPI], s2[RPI];
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      v2 t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
      const v2 rs2 = {rstd[u], rstd[u]}, nm2 = {-mean[u] * rstd[u], -mean[u] * rstd[u]};
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const v2 xs[2] = {{xv[u][it].x, xv[u][it].y}, {xv[u][it].z, xv[u][it].w}};
        const v2 ds[2] = {{dav[u][it].x, dav[u][it].y}, {dav[u][it].z, dav[u][it].w}};
        const v2 gs[2] = {{g4[it].x, g4[it].y}, {g4[it].z, g4[it].w}};
        const v2 bs[2] = {{b4[it].x, b4[it].y}, {b4[it].z, b4[it].w}};
        const v2 sc1[2] = {{1.0f + sc4[it].x, 1.0f + sc4[it].y}, {1.0f + sc4[it].z, 1.0f + sc4[it].w}};
        const v2 shs[2] = {{sh4[it].x, sh4[it].y}, {sh4[it].z, sh4[it].w}};
        v2 dgo[2], dbo[2], dsco[2], dsho[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const v2 xhat = __builtin_elementwise_fma(xs[h2], rs2, nm2);
          const v2 nrm = __builtin_elementwise_fma(xhat, gs[h2], bs[h2]);
          v2 dn;
          if (MOD_SILU) {
            const v2 uu = __builtin_elementwise_fma(nrm, sc1[h2], shs[h2]);
            const v2 w = uu * -1.4426950408889634f;
            const v2 den = v2{__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])} + 1.0f;
            const v2 sg = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};         // sigmoid(u)
            const v2 dsil = sg * __builtin_elementwise_fma(uu, 1.0f - sg, v2{1.0f, 1.0f});          // silu'(u) = s (1 + u (1 - s))
            const v2 du = ds[h2] * dsil;
            dsho[h2] = du;
            dsco[h2] = du * nrm;
            dn = du * sc1[h2];
          } else {
            dsho[h2] = dsco[h2] = v2{0.f, 0.f};
            dn = ds[h2];
          }
          dgo[h2] = dn * xhat;
          dbo[h2] = dn;
          const v2 dxo = dn * gs[h2];
          xh[u][it][h2] = xhat;
          dxh[u][it][h2] = dxo;
          t1 += dxo;
          t2 = __builtin_elementwise_fma(dxo, xhat, t2);
        }
        a_dg[it].x += dgo[0][0]; a_dg[it].y += dgo[0][1]; a_dg[it].z += dgo[1][0]; a_dg[it].w += dgo[1][1];
        a_db[it].x += dbo[0][0]; a_db[it].y += dbo[0][1]; a_db[it].z += dbo[1][0]; a_db[it].w += dbo[1][1];
        if (MOD_SILU) {
          a_dsc[it].x += dsco[0][0]; a_dsc[it].y += dsco[0][1]; a_dsc[it].z += dsco[1][0]; a_dsc[it].w += dsco[1][1];
          a_dsh[it].x += dsho[0][0]; a_dsh[it].y += dsho[0][1]; a_dsh[it].z += dsho[1][0]; a_dsh[it].w += dsho[1][1];
        }
      }
      s1[u] = t1[0] + t1[1];
      s2[u] = t2[0] + t2[1];
    }
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      s1[u] = wave_sum(s1[u]) * inv_n;
      s2[u] = wave_sum(s2[u]) * inv_n;
    }
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      const int rl = rl0 + u * stride;
      const v2 rs2 = {rstd[u], rstd[u]}, ns1 = {-s1[u], -s1[u]}, ns2 = {-s2[u], -s2[u]};
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = 4 * lane + 256 * it;
        const v2 r0 = {rv[u][it].x, rv[u][it].y}, r1 = {rv[u][it].z, rv[u][it].w};
        // rstd (dxh - s1 - xh s2) + res
        const v2 o0 = __builtin_elementwise_fma(__builtin_elementwise_fma(xh[u][it][0], ns2, dxh[u][it][0] + ns1), rs2, r0);
        const v2 o1 = __builtin_elementwise_fma(__builtin_elementwise_fma(xh[u][it][1], ns2, dxh[u][it][1] + ns1), rs2, r1);
        straw_buf<TD>(rso, (live[u] && cok[it]) ? (int)((rl * (int)lddx + c) * sizeof(TD)) : OOB, make_float4(o0[0], o0[1], o1[0], o1[1]));
      }
    }
  }
  float* pout = partial + ((int64_t)b * nsplit + split) * 4 * n;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    __syncthreads();
    *reinterpret_cast<float4*>(&red[wave][0][4 * lane]) = a_dg[it];
    *reinterpret_cast<float4*>(&red[wave][1][4 * lane]) = a_db[it];
    *reinterpret_cast<float4*>(&red[wave][2][4 * lane]) = a_dsc[it];
    *reinterpret_cast<float4*>(&red[wave][3][4 * lane]) = a_dsh[it];
    __syncthreads();
    for (int e = threadIdx.x; e < 4 * 256; e += 64 * NWV) {
      const int qn = e >> 8, cl = e & 255, c = cl + 256 * it;
      if (c < n) {
        float s = red[0][qn][cl];
#pragma unroll
        for (int w = 1; w < NWV; ++w) s += red[w][qn][cl];
        pout[(int64_t)qn * n + c] = s;
      }
    }
  }
}

// partial[chunk][c] = sum over this chunk's rows of x[row][c], x bf16, 8 columns per lane (n % 8 == 0)
__global__ __launch_bounds__(256) void colsum16_kernel(const __bf16* __restrict__ x, int64_t ldx, int64_t rows, int n,
                                                       float* __restrict__ partial) {
  __shared__ float red[WAVES][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 512 + 8 * lane;
  const int chunk = blockIdx.y, nchunk = gridDim.y;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < n) {
    for (int64_t r = (int64_t)chunk * WAVES + wave; r < rows; r += (int64_t)WAVES * nchunk) {
      const rbf16x8 v = *reinterpret_cast<const rbf16x8*>(x + r * ldx + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave][8 * lane + e] = s[e];
  __syncthreads();
  for (int cl = threadIdx.x; cl < 512; cl += 256) {
    const int c = blockIdx.x * 512 + cl;
    if (c < n) {
      float t = red[0][cl];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) t += red[w][cl];
      partial[(int64_t)chunk * n + c] = t;
    }
  }
}

// dst[c][r] = src[r][c] for bf16 matrices, 64 x 64 tiles: 16-byte loads along the source rows, the tile stored TRANSPOSED
// in LDS (two-byte writes), 16-byte loads from there along the destination rows.  cols % 8 == 0; a destination row holds
// round_up(rows, 8) elements (the last vector of a row carries zeros beyond `rows`).
struct Tr16Batch {
  const __bf16* src[12];
  __bf16* dst[12];
  int64_t lds_[12], ldd[12];
  int rows[12], cols[12];
  int tile0[13];
  int n;
};
__global__ __launch_bounds__(256) void transpose16_kernel(const Tr16Batch b) {
  __shared__ __attribute__((aligned(16))) __bf16 tile[64][72];   // [c][r], rows padded to 144 bytes
  int m = 0;
  while (m + 1 < b.n && (int)blockIdx.x >= b.tile0[m + 1]) ++m;
  const int rows = b.rows[m], cols = b.cols[m];
  const int lt = blockIdx.x - b.tile0[m], tcn = (cols + 63) / 64;
  const int c0 = (lt % tcn) * 64, r0 = (lt / tcn) * 64;
  const __bf16* __restrict__ src = b.src[m];
  __bf16* __restrict__ dst = b.dst[m];
  const int64_t lds_ = b.lds_[m], ldd = b.ldd[m];
  const int t = threadIdx.x;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rr = 32 * pass + (t >> 3), c8 = 8 * (t & 7);
    rbf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
    if (r0 + rr < rows && c0 + c8 < cols) v = *reinterpret_cast<const rbf16x8*>(src + (int64_t)(r0 + rr) * lds_ + c0 + c8);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[c8 + e][rr] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cc = 32 * pass + (t >> 3), r8 = 8 * (t & 7);
    if (c0 + cc < cols && r0 + r8 < rows)
      *reinterpret_cast<rbf16x8*>(dst + (int64_t)(c0 + cc) * ldd + r0 + r8) = *reinterpret_cast<const rbf16x8*>(&tile[cc][r8]);
  }
}

int splits_for(int64_t samples) {
  constexpr int target = 512;   // (a former tuning knob, fixed at the value that won its A/B): workgroups per launch
  int s = 1;
  while (samples * s < target && s < 64) s *= 2;
  return s;
}

}  // namespace

extern "C" int hig_rowstats(const float* x, int64_t ldx, int64_t rows, int32_t n, float* stats,
                            hig_stream_t stream) {
  HIG_REQUIRE(x && stats && n > 0 && rows >= 0, "hig_rowstats: bad arguments");
  if (rows == 0) return HIG_OK;
  hipLaunchKernelGGL(rowstats_kernel, dim3((unsigned)((rows + WAVES - 1) / WAVES)), dim3(256), 0,
                     hig_stream(stream), x, ldx, rows, n, stats);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_ln_mod_silu(const float* x, int64_t ldx, int64_t rows, int32_t n, const float* gamma,
                               const float* beta, const float* ss, int64_t ss_ld, int32_t ss_shift_off,
                               int32_t rows_per_sample, float* a, int64_t lda, float* stats,
                               hig_stream_t stream) {
  HIG_REQUIRE(x && gamma && beta && ss && a && stats && rows >= 0, "hig_ln_mod_silu: null argument");
  HIG_REQUIRE(n > 0 && n % 4 == 0 && n <= 1024 && ldx % 4 == 0 && lda % 4 == 0 && rows_per_sample > 0,
              "hig_ln_mod_silu: n must be a multiple of 4 and <= 1024 (got %d)", n);
  if (rows == 0) return HIG_OK;
  const dim3 grid((unsigned)((rows + WAVES - 1) / WAVES));
  const int nit = (n + 255) / 256;
#define LMS(NITV)                                                                                      \
  hipLaunchKernelGGL((ln_mod_silu_kernel<NITV, true>), grid, dim3(256), 0, hig_stream(stream), x, ldx, rows, n, \
                     gamma, beta, ss, ss_ld, ss_shift_off, rows_per_sample, a, lda, stats)
  if (nit == 1) LMS(1); else if (nit == 2) LMS(2); else LMS(4);
#undef LMS
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_layernorm(const float* x, int64_t ldx, int64_t rows, int32_t n, const float* gamma,
                             const float* beta, float* y, int64_t ldy, float* stats, hig_stream_t stream) {
  HIG_REQUIRE(x && gamma && beta && y && stats && rows >= 0, "hig_layernorm: null argument");
  HIG_REQUIRE(n > 0 && n % 4 == 0 && n <= 1024 && ldx % 4 == 0 && ldy % 4 == 0,
              "hig_layernorm: n must be a multiple of 4 and <= 1024 (got %d)", n);
  if (rows == 0) return HIG_OK;
  const dim3 grid((unsigned)((rows + WAVES - 1) / WAVES));
  const int nit = (n + 255) / 256;
#define LNF(NITV)                                                                                          \
  hipLaunchKernelGGL((ln_mod_silu_kernel<NITV, false>), grid, dim3(256), 0, hig_stream(stream), x, ldx, rows, n, \
                     gamma, beta, (const float*)nullptr, (int64_t)0, 0, 1, y, ldy, stats)
  if (nit == 1) LNF(1); else if (nit == 2) LNF(2); else LNF(4);
#undef LNF
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

namespace {
// dst[b][:] = src[b*rows_per_sample + idx[b]][:]   (EOT-token gather of encode_text, transformer.py:395)
__global__ void gather_rows_kernel(const float* __restrict__ src, int64_t ld, int rows_per_sample,
                                   const int64_t* __restrict__ idx, int n, float* __restrict__ dst, int64_t ldd) {
  const int b = blockIdx.x;
  const int64_t r = (int64_t)b * rows_per_sample + idx[b];
  for (int c = threadIdx.x; c < n; c += blockDim.x) dst[(int64_t)b * ldd + c] = src[r * ld + c];
}
// dst[b*rows_per_sample + idx[b]][:] += src[b][:]   (its adjoint; one row per sample, no collisions)
__global__ void scatter_add_rows_kernel(const float* __restrict__ src, int64_t ld, int rows_per_sample,
                                        const int64_t* __restrict__ idx, int n, float* __restrict__ dst, int64_t ldd) {
  const int b = blockIdx.x;
  const int64_t r = (int64_t)b * rows_per_sample + idx[b];
  for (int c = threadIdx.x; c < n; c += blockDim.x) dst[r * ldd + c] += src[(int64_t)b * ld + c];
}
}  // namespace

extern "C" int hig_gather_rows(const float* src, int64_t ld, int32_t B, int32_t rows_per_sample, const int64_t* idx,
                               int32_t n, float* dst, int64_t ldd, hig_stream_t stream) {
  HIG_REQUIRE(src && idx && dst && B >= 0 && n > 0 && rows_per_sample > 0, "hig_gather_rows: bad argument");
  if (B == 0) return HIG_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, hig_stream(stream), src, ld, rows_per_sample, idx, n,
                     dst, ldd);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
extern "C" int hig_scatter_add_rows(const float* src, int64_t ld, int32_t B, int32_t rows_per_sample,
                                    const int64_t* idx, int32_t n, float* dst, int64_t ldd, hig_stream_t stream) {
  HIG_REQUIRE(src && idx && dst && B >= 0 && n > 0 && rows_per_sample > 0, "hig_scatter_add_rows: bad argument");
  if (B == 0) return HIG_OK;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(B), dim3(256), 0, hig_stream(stream), src, ld, rows_per_sample, idx,
                     n, dst, ldd);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int64_t hig_ln_bwd_partial_floats(int64_t rows, int32_t n, int32_t rows_per_sample) {
  if (rows_per_sample <= 0) return 0;
  const int64_t samples = rows / rows_per_sample;
  return samples * splits_for(samples) * 4 * (int64_t)n;
}

extern "C" int hig_ln_bwd(const float* da, int64_t ldda, const float* x, int64_t ldx,
                          const float* stats, const float* gamma, const float* beta,
                          const float* ss, int64_t ss_ld, int32_t ss_shift_off, int32_t mod_silu,
                          const float* res, int64_t ldr, float* dx, int64_t lddx, int64_t rows,
                          int32_t n, int32_t rows_per_sample, float* dgamma, float* dbeta,
                          float* dss, int64_t dss_ld, float* partial, hig_stream_t stream) {
  HIG_REQUIRE(da && x && stats && gamma && beta && dx && partial, "hig_ln_bwd: null argument");
  HIG_REQUIRE(n % 4 == 0 && n <= 1024 && ldda % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0,
              "hig_ln_bwd: n must be a multiple of 4 and <= 1024 (got %d)", n);
  HIG_REQUIRE(rows_per_sample > 0 && rows % rows_per_sample == 0, "hig_ln_bwd: rows %% rows_per_sample");
  HIG_REQUIRE(!mod_silu || (ss && dss), "hig_ln_bwd: modulation needs ss / dss");
  if (rows == 0) return HIG_OK;
  const int samples = (int)(rows / rows_per_sample);
  const int nsplit = splits_for(samples);
  const int nit = (n + 255) / 256;
  hipStream_t st = hig_stream(stream);
  dim3 grid(samples, nsplit);
#define LNB(NITV, MODV)                                                                          \
  hipLaunchKernelGGL((ln_bwd_kernel<NITV, MODV>), grid, dim3(256), 0, st, da, ldda, x, ldx, stats, \
                     gamma, beta, ss, ss_ld, ss_shift_off, res, ldr, dx, lddx, n, rows_per_sample, partial)
  if (mod_silu) {
    if (nit == 1) LNB(1, true); else if (nit == 2) LNB(2, true); else LNB(4, true);
  } else {
    if (nit == 1) LNB(1, false); else if (nit == 2) LNB(2, false); else LNB(4, false);
  }
#undef LNB
  HIG_CHECK_LAUNCH();
  const int tb = 128;
  if (dgamma && dbeta && mod_silu) {   // the usual stylization case: all four sums in one launch
    const int nb_col = (2 * n + 63) / 64;
    const int nb_dss = (int)(((int64_t)samples * n + 1023) / 1024);
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(nb_col + nb_dss), dim3(1024), 0, st, partial, samples, nsplit, n, dgamma,
                       dbeta, ss_shift_off, dss, dss_ld, nb_col);
    HIG_CHECK_LAUNCH();
    return HIG_OK;
  }
  if (dgamma && dbeta) {   // partial rows are [dgamma n | dbeta n | ...]: both in one launch
    hipLaunchKernelGGL(colreduce_kernel, dim3((2 * n + 63) / 64), dim3(1024), 0, st, partial,
                       samples * nsplit, (int64_t)4 * n, 2 * n, dgamma, n, dbeta);
    HIG_CHECK_LAUNCH();
  } else if (dgamma) {
    hipLaunchKernelGGL(colreduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, partial,
                       samples * nsplit, (int64_t)4 * n, n, dgamma, 1 << 30, (float*)nullptr);
    HIG_CHECK_LAUNCH();
  } else if (dbeta) {
    hipLaunchKernelGGL(colreduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, partial + n,
                       samples * nsplit, (int64_t)4 * n, n, dbeta, 1 << 30, (float*)nullptr);
    HIG_CHECK_LAUNCH();
  }
  if (mod_silu) {
    hipLaunchKernelGGL(dss_reduce_kernel, dim3((n + tb - 1) / tb, samples), dim3(tb), 0, st, partial,
                       nsplit, n, ss_shift_off, dss, dss_ld);
    HIG_CHECK_LAUNCH();
  }
  return HIG_OK;
}

extern "C" int hig_colsum_chunks(int64_t rows) {
  // enough row chunks to fill the chip (each chunk = one workgroup per 256 columns), at least
  // 16 rows per chunk
  int64_t c = rows / 16;
  return (int)(c < 1 ? 1 : (c > HIG_COLSUM_CHUNKS ? HIG_COLSUM_CHUNKS : c));
}

extern "C" int hig_colsum(const float* x, int64_t ldx, int64_t rows, int32_t n, float* out,
                          float* partial, hig_stream_t stream) {
  HIG_REQUIRE(x && out && partial && n > 0, "hig_colsum: bad arguments");
  hipStream_t st = hig_stream(stream);
  const int chunks = hig_colsum_chunks(rows);
  hipLaunchKernelGGL(colsum_kernel, dim3((n + 255) / 256, chunks), dim3(256), 0, st, x, ldx,
                     rows, n, partial);
  HIG_CHECK_LAUNCH();
  hipLaunchKernelGGL(colreduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, partial,
                     chunks, (int64_t)n, n, out);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}


// Device-side fill / copy as KERNELS.  The library does not put hipMemsetAsync / hipMemcpyAsync (1-D or 2-D) into streams that may be
// under capture: as nodes of a replayed hipGraph they were seen to run out of order with the neighbouring kernel nodes when the graph
// was launched onto an idle device (the two-person training steps diverged: profiles/r05_notes.md section 8).
namespace {
__global__ void zero_words_kernel(unsigned* __restrict__ p, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ void copy_words_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void copy_quads_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace
int hig_zero_async(void* p, int64_t bytes, hipStream_t st) {
  HIG_REQUIRE(bytes >= 0 && bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0, "hig_zero_async: whole 4-byte words");
  if (bytes == 0) return HIG_OK;
  const int64_t n = bytes / 4;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<unsigned*>(p), n);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
int hig_copy_async(void* dst, const void* src, int64_t bytes, hipStream_t st) {
  HIG_REQUIRE(bytes >= 0 && bytes % 4 == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 3) == 0,
              "hig_copy_async: whole 4-byte words");
  if (bytes == 0) return HIG_OK;
  const bool quads = bytes % 16 == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
  const int64_t n = quads ? bytes / 16 : bytes / 4;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (quads) hipLaunchKernelGGL(copy_quads_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n);
  else hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const unsigned*>(src), static_cast<unsigned*>(dst), n);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// ---- bf16-storage training step: row kernels with bf16 I/O -------------------------------------------------------------
// deferred (nullable; needs dgamma and dbeta): the reductions of the partial table are NOT launched -- *deferred describes them for
// hig_ln_bwd16_reduce_batch, and `partial` must stay untouched until then.
int hig_ln_bwd16_launch(const void* da, int64_t ldda, const void* x, int32_t x_f32, int64_t ldx, const float* gamma,
                        const float* beta, const float* ss, int64_t ss_ld, int32_t ss_shift_off, int32_t mod_silu,
                        const void* res, int64_t ldr, void* dx, int32_t dx_f32, int64_t lddx, int64_t rows, int32_t n,
                        int32_t rows_per_sample, float* dgamma, float* dbeta, float* dss, int64_t dss_ld,
                        float* partial, hig_stream_t stream, hig_ln_reduce* deferred) {
  if (deferred) deferred->nsplit = 0;
  HIG_REQUIRE(da && x && gamma && beta && dx && partial, "hig_ln_bwd_bf16: null argument");
  HIG_REQUIRE(n % 4 == 0 && n <= 1024 && ldda % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!res || ldr % 4 == 0),
              "hig_ln_bwd_bf16: n and the leading dimensions must be multiples of 4, n <= 1024 (got %d)", n);
  HIG_REQUIRE(((reinterpret_cast<uintptr_t>(da) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
                reinterpret_cast<uintptr_t>(res)) & 7) == 0, "hig_ln_bwd_bf16: 8-byte aligned rows");
  HIG_REQUIRE(rows_per_sample > 0 && rows % rows_per_sample == 0, "hig_ln_bwd_bf16: rows %% rows_per_sample");
  {  // the kernel addresses a sample's rows through buffer descriptors with 32-bit byte offsets
    const int64_t ldmax = std::max(std::max(ldda, ldx), std::max(lddx, res ? ldr : (int64_t)0));
    if ((int64_t)rows_per_sample * ldmax * 4 >= (1ll << 31))
      return hig_set_error(HIG_EUNSUPPORTED, "hig_ln_bwd_bf16: a sample of more than 2 GiB");
  }
  HIG_REQUIRE(!mod_silu || (ss && dss), "hig_ln_bwd_bf16: modulation needs ss / dss");
  // (the parameter gradients are reduced as a pair: one pointer alone would return HIG_OK and leave that gradient unwritten)
  HIG_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "hig_ln_bwd_bf16: dgamma and dbeta must both be given or both be NULL");
  if (mod_silu && (x_f32 || dx_f32))
    return hig_set_error(HIG_EUNSUPPORTED, "hig_ln_bwd_bf16: the stylization form is built for bf16 rows only");
  if (rows == 0) return HIG_OK;
  const int samples = (int)(rows / rows_per_sample);
  // waves per workgroup: 8, in half as many workgroups (the partial sums -- one set per WORKGROUP -- halve, the rows are shared
  // by as many waves as before).  Same call, config 2 (tools/train16_kernels_time.py): 4 waves x 512 workgroups 25.4 / 22.8 us
  // (stylization / plain + residual form), 8 x 256: 24.0 / 21.1, 8 x 512: 28.3 / 19.1, 4 x 256: 32.6 / 30.4
  constexpr int nwv = 8;   // (a former tuning knob, fixed at the value that won its A/B): 4 / 8
  constexpr bool wgs_forced = false;
  const int nit = (n + 255) / 256;
  int nsplit = splits_for(samples);             // (hig_ln_bwd_partial_floats sizes `partial` for this many)
  if (nwv == 8 && nit <= 2 && !wgs_forced && nsplit > 1) nsplit /= 2;
  hipStream_t st = hig_stream(stream);
  dim3 grid(samples, nsplit);
  const __bf16* dab = static_cast<const __bf16*>(da);
#define LNB16_W(NITV, MODV, TXV, TDV, NWVV)                                                                                                     \
  hipLaunchKernelGGL((ln_bwd16_kernel<NITV, MODV, TXV, TDV, (NITV <= 2 ? 2 : 1), NWVV>), grid, dim3(64 * NWVV), 0, st, dab, ldda, static_cast<const TXV*>(x), ldx, \
                     gamma, beta, ss, ss_ld, ss_shift_off, static_cast<const TDV*>(res), ldr, static_cast<TDV*>(dx), lddx, n,    \
                     rows_per_sample, partial)
#define LNB16(NITV, MODV, TXV, TDV)                                              \
  do {                                                                           \
    if (nwv == 8 && NITV <= 2) LNB16_W(NITV, MODV, TXV, TDV, 8);                 \
    else LNB16_W(NITV, MODV, TXV, TDV, 4);                                       \
  } while (0)
#define LNB16_NIT(MODV, TXV, TDV)                                                    \
  do {                                                                               \
    if (nit == 1) LNB16(1, MODV, TXV, TDV);                                          \
    else if (nit == 2) LNB16(2, MODV, TXV, TDV);                                     \
    else LNB16(4, MODV, TXV, TDV);                                                   \
  } while (0)
  if (mod_silu) LNB16_NIT(true, __bf16, __bf16);
  else if (!x_f32 && !dx_f32) LNB16_NIT(false, __bf16, __bf16);
  else if (x_f32 && dx_f32) LNB16_NIT(false, float, float);
  else if (!x_f32 && dx_f32) LNB16_NIT(false, __bf16, float);
  else return hig_set_error(HIG_EUNSUPPORTED, "hig_ln_bwd_bf16: fp32 rows with a bf16 result is not built");
#undef LNB16_NIT
#undef LNB16
#undef LNB16_W
  HIG_CHECK_LAUNCH();
  if (deferred && dgamma && dbeta) {
    deferred->partial = partial; deferred->samples = samples; deferred->nsplit = nsplit; deferred->n = n;
    deferred->dgamma = dgamma; deferred->dbeta = dbeta; deferred->shift_off = ss_shift_off;
    deferred->dss = mod_silu ? dss : nullptr; deferred->dss_ld = dss_ld;
    deferred->nb_col = (2 * n + 63) / 64;
    deferred->nb_dss = mod_silu ? (int)(((int64_t)samples * n + 1023) / 1024) : 0;
    return HIG_OK;
  }
  if (dgamma && dbeta && mod_silu) {
    const int nb_col = (2 * n + 63) / 64;
    const int nb_dss = (int)(((int64_t)samples * n + 1023) / 1024);
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(nb_col + nb_dss), dim3(1024), 0, st, partial, samples, nsplit, n, dgamma,
                       dbeta, ss_shift_off, dss, dss_ld, nb_col);
    HIG_CHECK_LAUNCH();
    return HIG_OK;
  }
  if (dgamma && dbeta) {
    hipLaunchKernelGGL(colreduce_kernel, dim3((2 * n + 63) / 64), dim3(1024), 0, st, partial,
                       samples * nsplit, (int64_t)4 * n, 2 * n, dgamma, n, dbeta);
    HIG_CHECK_LAUNCH();
  }
  if (mod_silu) {
    hipLaunchKernelGGL(dss_reduce_kernel, dim3((n + 127) / 128, samples), dim3(128), 0, st, partial,
                       nsplit, n, ss_shift_off, dss, dss_ld);
    HIG_CHECK_LAUNCH();
  }
  return HIG_OK;
}

extern "C" int hig_ln_bwd_bf16(const void* da, int64_t ldda, const void* x, int32_t x_f32, int64_t ldx, const float* gamma,
                               const float* beta, const float* ss, int64_t ss_ld, int32_t ss_shift_off, int32_t mod_silu,
                               const void* res, int64_t ldr, void* dx, int32_t dx_f32, int64_t lddx, int64_t rows, int32_t n,
                               int32_t rows_per_sample, float* dgamma, float* dbeta, float* dss, int64_t dss_ld,
                               float* partial, hig_stream_t stream) {
  return hig_ln_bwd16_launch(da, ldda, x, x_f32, ldx, gamma, beta, ss, ss_ld, ss_shift_off, mod_silu, res, ldr, dx, dx_f32, lddx, rows, n,
                             rows_per_sample, dgamma, dbeta, dss, dss_ld, partial, stream, nullptr);
}

int hig_ln_bwd16_reduce_batch(const hig_ln_reduce* entries, int n, hipStream_t st) {
  HIG_REQUIRE(n >= 0 && n <= HIG_LN_RB_MAX, "hig_ln_bwd16_reduce_batch: at most %d entries", HIG_LN_RB_MAX);
  LnReduceBatch b;
  b.n = 0;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (entries[i].nsplit <= 0) continue;
    b.e[b.n] = entries[i];
    b.block0[b.n] = blocks;
    blocks += entries[i].nb_col + entries[i].nb_dss;
    ++b.n;
  }
  if (b.n == 0) return HIG_OK;
  b.block0[b.n] = blocks;
  hipLaunchKernelGGL(ln_bwd_reduce_batch_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, b);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_colsum_bf16(const void* x, int64_t ldx, int64_t rows, int32_t n, float* out, float* partial,
                               hig_stream_t stream) {
  HIG_REQUIRE(x && out && partial && n > 0, "hig_colsum_bf16: bad arguments");
  HIG_REQUIRE(n % 8 == 0 && ldx % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "hig_colsum_bf16: n, ldx multiples of 8, 16-byte aligned rows");
  hipStream_t st = hig_stream(stream);
  const int chunks = hig_colsum_chunks(rows);
  hipLaunchKernelGGL(colsum16_kernel, dim3((n + 511) / 512, chunks), dim3(256), 0, st, static_cast<const __bf16*>(x), ldx, rows, n,
                     partial);
  HIG_CHECK_LAUNCH();
  hipLaunchKernelGGL(colreduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, partial, chunks, (int64_t)n, n, out);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_transpose_bf16_batch(int32_t n, const void* const* srcs, const int64_t* lds_, void* const* dsts,
                                        const int64_t* ldd, const int32_t* rows, const int32_t* cols, hig_stream_t stream) {
  HIG_REQUIRE(n >= 0 && n <= 12 && (n == 0 || (srcs && dsts && rows && cols && lds_ && ldd)),
              "hig_transpose_bf16_batch: 0 <= n <= 12 matrices");
  if (n == 0) return HIG_OK;
  Tr16Batch b;
  b.n = n;
  int t = 0;
  for (int m = 0; m < n; ++m) {
    HIG_REQUIRE(srcs[m] && dsts[m] && rows[m] > 0 && cols[m] > 0 && cols[m] % 8 == 0 && lds_[m] % 8 == 0 &&
                    ldd[m] % 8 == 0 && ldd[m] >= (rows[m] + 7) / 8 * 8 && ((reinterpret_cast<uintptr_t>(srcs[m]) | reinterpret_cast<uintptr_t>(dsts[m])) & 15) == 0,
                "hig_transpose_bf16_batch: matrix %d: extents / leading dimensions multiples of 8, 16-byte aligned", m);
    b.src[m] = static_cast<const __bf16*>(srcs[m]); b.dst[m] = static_cast<__bf16*>(dsts[m]);
    b.lds_[m] = lds_[m]; b.ldd[m] = ldd[m]; b.rows[m] = rows[m]; b.cols[m] = cols[m];
    b.tile0[m] = t;
    t += ((rows[m] + 63) / 64) * ((cols[m] + 63) / 64);
  }
  b.tile0[n] = t;
  hipLaunchKernelGGL(transpose16_kernel, dim3(t), dim3(256), 0, hig_stream(stream), b);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
extern "C" int hig_transpose_bf16(const void* src, int64_t ld, int32_t rows, int32_t cols, void* dst, int64_t ldd,
                                  hig_stream_t stream) {
  const void* s1[1] = {src};
  void* d1[1] = {dst};
  const int64_t l1[1] = {ld}, l2[1] = {ldd};
  const int32_t r1[1] = {rows}, c1[1] = {cols};
  return hig_transpose_bf16_batch(1, s1, l1, d1, l2, r1, c1, stream);
}

extern "C" int hig_transpose(const float* src, int64_t ld, int32_t rows, int32_t cols, float* dst, int64_t ldd,
                             const float* stats, const float* gamma, const float* beta, hig_stream_t stream) {
  HIG_REQUIRE(src && dst && rows > 0 && cols > 0, "hig_transpose: bad arguments");
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (stats) {
    HIG_REQUIRE(gamma && beta, "hig_transpose: LayerNorm needs gamma/beta");
    hipLaunchKernelGGL((transpose_kernel<true>), grid, dim3(256), 0, hig_stream(stream), src, ld, rows, cols, dst,
                       ldd, stats, gamma, beta);
  } else {
    hipLaunchKernelGGL((transpose_kernel<false>), grid, dim3(256), 0, hig_stream(stream), src, ld, rows, cols, dst,
                       ldd, stats, gamma, beta);
  }
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_transpose_batch(int32_t n, const float* const* srcs, float* const* dsts, const int32_t* rows,
                                   const int32_t* cols, hig_stream_t stream) {
  HIG_REQUIRE(n >= 0 && n <= TRB_MAX && (n == 0 || (srcs && dsts && rows && cols)),
              "hig_transpose_batch: 0 <= n <= %d matrices", TRB_MAX);
  if (n == 0) return HIG_OK;
  TrBatch b;
  b.n = n;
  int t = 0;
  for (int m = 0; m < n; ++m) {
    HIG_REQUIRE(srcs[m] && dsts[m] && rows[m] > 0 && cols[m] > 0, "hig_transpose_batch: bad matrix %d", m);
    b.src[m] = srcs[m]; b.dst[m] = dsts[m]; b.rows[m] = rows[m]; b.cols[m] = cols[m];
    b.tile0[m] = t;
    t += ((rows[m] + 63) / 64) * ((cols[m] + 63) / 64);
  }
  b.tile0[n] = t;
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(t), dim3(256), 0, hig_stream(stream), b);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_timestep_embedding(const int64_t* t, int32_t B, int32_t d, float* out,
                                      hig_stream_t s) {
  HIG_REQUIRE(t && out && B > 0 && d > 0, "hig_timestep_embedding: bad arguments");
  hipLaunchKernelGGL(timestep_embedding_kernel<float>, dim3((B * d + 255) / 256), dim3(256), 0, hig_stream(s),
                     t, B, d, out);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
// the same values rounded to bf16 (the bf16-storage forward feeds them to a bf16 GEMM: no fp32 copy, no cast launch)
extern "C" int hig_timestep_embedding_bf16(const int64_t* t, int32_t B, int32_t d, void* out16, hig_stream_t s) {
  HIG_REQUIRE(t && out16 && B > 0 && d > 0, "hig_timestep_embedding_bf16: bad arguments");
  hipLaunchKernelGGL(timestep_embedding_kernel<__bf16>, dim3((B * d + 255) / 256), dim3(256), 0, hig_stream(s),
                     t, B, d, static_cast<__bf16*>(out16));
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
