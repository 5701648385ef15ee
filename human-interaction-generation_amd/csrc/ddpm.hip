// DDPM elementwise arithmetic, masked-MSE loss and fused clip+Adam: the HBM-bound tail of the
// training step / sampling step.  Reference: codes/models/gaussian_diffusion.py:399-417 (q_sample),
// :443-544,606-666 (p_mean_variance / p_sample, EPSILON + FIXED_SMALL, clip_denoised=False),
// codes/trainers/ddpm_trainer.py:172-187 (masked loss, clip_grad_norm_(0.5), Adam).
// All kernels are float4 grid-stride streams (16 B/lane), graph-capturable, no host sync.
#include "hig_common.h"

namespace {

enum { T_SQRT_AC = 0, T_SQRT_1M_AC, T_SQRT_RECIP_AC, T_SQRT_RECIPM1_AC, T_COEF1, T_COEF2, T_LOGVAR };

__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                const int64_t* __restrict__ t, const float* __restrict__ tab,
                                int nsteps, int64_t per_sample, int64_t total,
                                float* __restrict__ xt) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int tt = (int)t[i / per_sample];
    xt[i] = tab[T_SQRT_AC * nsteps + tt] * x0[i] + tab[T_SQRT_1M_AC * nsteps + tt] * noise[i];
  }
}

__global__ void p_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                              const float* __restrict__ z, const int64_t* __restrict__ t,
                              const float* __restrict__ tab, int nsteps, int64_t per_sample,
                              int64_t total, float* __restrict__ x_prev,
                              float* __restrict__ pred_xstart) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int tt = (int)t[i / per_sample];
    const float xi = x[i];
    // same operation order as the reference: x0 = a*x - b*eps; mean = c1*x0 + c2*x
    const float x0 = tab[T_SQRT_RECIP_AC * nsteps + tt] * xi - tab[T_SQRT_RECIPM1_AC * nsteps + tt] * eps[i];
    const float mean = tab[T_COEF1 * nsteps + tt] * x0 + tab[T_COEF2 * nsteps + tt] * xi;
    const float nz = tt != 0 ? 1.0f : 0.0f;
    const float sd = expf(0.5f * tab[T_LOGVAR * nsteps + tt]);
    if (pred_xstart) pred_xstart[i] = x0;
    x_prev[i] = mean + nz * sd * z[i];
  }
}

__global__ void dec_t_kernel(int64_t* t, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) t[i] -= 1;
}

// One wave per (b, t) row: row mean of squared error, masked; dpred written in the same pass.
__global__ __launch_bounds__(256) void masked_mse_kernel(const float* __restrict__ pred,
                                                         const float* __restrict__ target,
                                                         const int64_t* __restrict__ length, int B,
                                                         int T, int F, float* __restrict__ dpred,
                                                         float* __restrict__ partial) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // sum(mask) = sum_b clamp(length[b], 0, T)
  float cnt = 0.f;
  for (int b = lane; b < B; b += 64) {
    int64_t l = length ? length[b] : T;
    l = l < 0 ? 0 : (l > T ? T : l);
    cnt += (float)l;
  }
  cnt = wave_sum(cnt);
  const float gscale = 2.0f / ((float)F * cnt);
  float acc = 0.f;
  const int64_t rows = (int64_t)B * T;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int b = (int)(row / T), tt = (int)(row % T);
    const int64_t l = length ? length[b] : T;
    const bool on = tt < l;
    const float* p = pred + row * F;
    const float* q = target + row * F;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) {
      const float dlt = p[f] - q[f];
      s += dlt * dlt;
      if (dpred) dpred[row * F + f] = on ? gscale * dlt : 0.f;
    }
    s = wave_sum(s);
    if (on) acc += s / (float)F;
  }
  if (lane == 0) wsum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  if (blockIdx.x == 0 && threadIdx.x == 0) partial[gridDim.x] = cnt;
}
__global__ void masked_mse_final_kernel(const float* __restrict__ partial, int nblk,
                                        float* __restrict__ loss) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] / partial[nblk];
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n,
                                                    float inv_world, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t n4 = n / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(g)[i];
    v.x *= inv_world; v.y *= inv_world; v.z *= inv_world; v.w *= inv_world;
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[n4 * 4 + threadIdx.x] * inv_world;
    s += v * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        int64_t n, float lr, float b1, float b2,
                                                        float eps, float max_norm, float inv_world,
                                                        const float* __restrict__ partial,
                                                        float* __restrict__ gnorm_out,
                                                        const int32_t* __restrict__ step_dev,
                                                        const float* __restrict__ lr_dev,
                                                        __bf16* __restrict__ shadow, int64_t shadow_n) {
  __shared__ float red[256];
  __shared__ float s_coef, s_step_size, s_inv_sqrt_bc2;
  {
    float s = 0.f;
    for (int i = threadIdx.x; i < HIG_NORM_BLOCKS; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float gnorm = sqrtf(red[0]);
      float coef = max_norm / (gnorm + 1e-6f);  // torch.nn.utils.clip_grad_norm_
      coef = coef > 1.0f ? 1.0f : coef;
      if (max_norm <= 0.f) coef = 1.0f;
      s_coef = coef * inv_world;
      const double t = (double)(*step_dev + 1);
      const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
      s_step_size = (float)((double)(lr_dev ? lr_dev[0] : lr) / bc1);
      s_inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
      if (blockIdx.x == 0 && gnorm_out) gnorm_out[0] = gnorm;
    }
    __syncthreads();
  }
  const float coef = s_coef, step_size = s_step_size, isb2 = s_inv_sqrt_bc2;
  const float ob1 = 1.0f - b1, ob2 = 1.0f - b2;
  const int64_t n4 = n / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float gg[4] = {gv.x * coef, gv.y * coef, gv.z * coef, gv.w * coef};
    float pp[4] = {pv.x, pv.y, pv.z, pv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w},
          v2[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mm[e] = b1 * mm[e] + ob1 * gg[e];
      v2[e] = b2 * v2[e] + ob2 * gg[e] * gg[e];
      pp[e] -= step_size * (mm[e] / (sqrtf(v2[e]) * isb2 + eps));
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    if (shadow && 4 * i + 3 < shadow_n) {   // bf16 shadow of the new parameters (bf16-storage training: no separate cast pass)
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      reinterpret_cast<bf16x4_t*>(shadow)[i] = bf16x4_t{(__bf16)pp[0], (__bf16)pp[1], (__bf16)pp[2], (__bf16)pp[3]};
    }
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(v2[0], v2[1], v2[2], v2[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    const float gg = g[i] * coef;
    const float mm = b1 * m[i] + ob1 * gg, v2 = b2 * v[i] + ob2 * gg * gg;
    m[i] = mm;
    v[i] = v2;
    const float pn = p[i] - step_size * (mm / (sqrtf(v2) * isb2 + eps));
    p[i] = pn;
    if (shadow && i < shadow_n) shadow[i] = (__bf16)pn;
  }
}
__global__ void inc_step_kernel(int32_t* s) { s[0] += 1; }

int stream_blocks(int64_t total) {
  const int64_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// ---- two-person losses (DDPMMulTrainer.backward_G, mul_ddpm_trainer.py:223-247) ------------------------
// Row r of the model batch, token t: l[r][t] = mean_f (pred - target)^2 over all F features, except the
// init-pose token t == 0 which is scored on its first 4 features only.  rowloss[r] = sum_{t < len[r]} l[r][t].
__global__ __launch_bounds__(256) void pair_rowloss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           const int64_t* __restrict__ length, int R, int T, int F,
                                                           float* __restrict__ rowloss) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x;
  int64_t l = length ? length[r] : T;
  l = l < 0 ? 0 : (l > T ? T : l);
  float acc = 0.f;
  for (int t = wave; t < (int)l; t += 4) {
    const int nf = t == 0 ? 4 : F;
    const float* p = pred + ((int64_t)r * T + t) * F;
    const float* q = target + ((int64_t)r * T + t) * F;
    float sq = 0.f;
    for (int f = lane; f < nf; f += 64) {
      const float dlt = p[f] - q[f];
      sq += dlt * dlt;
    }
    sq = wave_sum(sq);
    acc += sq / (float)nf;
  }
  if (lane == 0) wsum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) rowloss[r] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
// One block.  pit == 0: loss = sum_r rowloss / sum(mask), every row active.  pit == 1 (rows = [m1|c1, m1|c2,
// m2|c2, m2|c1], R = 4P): S0[p] = L[p] + L[2P + p], S1[p] = L[P + p] + L[3P + p]; loss = sum_p min(S0, S1) /
// (sum(mask) / 2); only the rows of the cheaper caption assignment of each pair get a gradient (ties: the first).
// rowscale[r] = d loss / d rowloss[r].
__global__ __launch_bounds__(256) void pair_select_kernel(const float* __restrict__ rowloss,
                                                          const int64_t* __restrict__ length, int R, int T, int pit,
                                                          float* __restrict__ loss, float* __restrict__ rowscale) {
  __shared__ float red[256];
  __shared__ float scnt;
  float cnt = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    int64_t l = length ? length[r] : T;
    cnt += (float)(l < 0 ? 0 : (l > T ? T : l));
  }
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) scnt = red[0];
  __syncthreads();
  const float denom = pit ? scnt * 0.5f : scnt;
  float part = 0.f;
  if (pit) {
    const int P = R / 4;
    for (int p = threadIdx.x; p < P; p += 256) {
      const float s0 = rowloss[p] + rowloss[2 * P + p], s1 = rowloss[P + p] + rowloss[3 * P + p];
      const bool first = s0 <= s1;
      part += first ? s0 : s1;
      rowscale[p] = rowscale[2 * P + p] = first ? 1.0f / denom : 0.f;
      rowscale[P + p] = rowscale[3 * P + p] = first ? 0.f : 1.0f / denom;
    }
  } else {
    for (int r = threadIdx.x; r < R; r += 256) {
      part += rowloss[r];
      rowscale[r] = 1.0f / denom;
    }
  }
  __syncthreads();
  red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] / denom;
}
__global__ __launch_bounds__(256) void pair_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        const int64_t* __restrict__ length,
                                                        const float* __restrict__ rowscale, int R, int T, int F,
                                                        float* __restrict__ dpred) {
  const int64_t n = (int64_t)R * T * F;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const int64_t rt = i / F;
    const int t = (int)(rt % T), r = (int)(rt / T);
    const int64_t l = length ? length[r] : T;
    float g = 0.f;
    if (t < l && (t > 0 || f < 4)) g = rowscale[r] * 2.0f * (pred[i] - target[i]) / (float)(t == 0 ? 4 : F);
    dpred[i] = g;
  }
}

}  // namespace

extern "C" int hig_q_sample(const float* x0, const float* noise, const int64_t* t, const float* tab,
                            int32_t nsteps, int32_t B, int64_t per_sample, float* xt, hig_stream_t s) {
  HIG_REQUIRE(x0 && noise && t && tab && xt && B > 0 && per_sample > 0, "hig_q_sample: bad arguments");
  const int64_t total = (int64_t)B * per_sample;
  hipLaunchKernelGGL(q_sample_kernel, dim3(stream_blocks(total)), dim3(256), 0, hig_stream(s), x0, noise, t,
                     tab, nsteps, per_sample, total, xt);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_p_sample_step(const float* x, const float* eps, const float* z, const int64_t* t,
                                 const float* tab, int32_t nsteps, int32_t B, int64_t per_sample,
                                 float* x_prev, float* pred_xstart, hig_stream_t s) {
  HIG_REQUIRE(x && eps && z && t && tab && x_prev && B > 0 && per_sample > 0, "hig_p_sample_step: bad arguments");
  const int64_t total = (int64_t)B * per_sample;
  hipLaunchKernelGGL(p_step_kernel, dim3(stream_blocks(total)), dim3(256), 0, hig_stream(s), x, eps, z, t,
                     tab, nsteps, per_sample, total, x_prev, pred_xstart);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_dec_timesteps(int64_t* t, int32_t B, hig_stream_t s) {
  HIG_REQUIRE(t && B > 0, "hig_dec_timesteps: bad arguments");
  hipLaunchKernelGGL(dec_t_kernel, dim3((B + 255) / 256), dim3(256), 0, hig_stream(s), t, B);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_pair_mse(const float* pred, const float* target, const int64_t* length, int32_t R, int32_t T,
                            int32_t F, int32_t pit, float* loss, float* dpred, float* scratch, hig_stream_t s) {
  HIG_REQUIRE(pred && target && loss && scratch && R > 0 && T > 0 && F >= 4, "hig_pair_mse: bad arguments");
  HIG_REQUIRE(!pit || R % 4 == 0, "hig_pair_mse: PIT needs 4 groups of rows (got %d rows)", R);
  float* rowloss = scratch;          // [R]
  float* rowscale = scratch + R;     // [R]
  hipLaunchKernelGGL(pair_rowloss_kernel, dim3(R), dim3(256), 0, hig_stream(s), pred, target, length, R, T, F, rowloss);
  HIG_CHECK_LAUNCH();
  hipLaunchKernelGGL(pair_select_kernel, dim3(1), dim3(256), 0, hig_stream(s), rowloss, length, R, T, pit, loss,
                     rowscale);
  HIG_CHECK_LAUNCH();
  if (dpred) {
    const int64_t n = (int64_t)R * T * F;
    hipLaunchKernelGGL(pair_grad_kernel, dim3(stream_blocks(n)), dim3(256), 0, hig_stream(s), pred, target, length,
                       rowscale, R, T, F, dpred);
    HIG_CHECK_LAUNCH();
  }
  return HIG_OK;
}

extern "C" int hig_masked_mse(const float* pred, const float* target, const int64_t* length, int32_t B,
                              int32_t T, int32_t F, float* loss, float* dpred, float* scratch,
                              hig_stream_t s) {
  HIG_REQUIRE(pred && target && loss && scratch && B > 0 && T > 0 && F > 0, "hig_masked_mse: bad arguments");
  const int64_t rows = (int64_t)B * T;
  int nblk = (int)((rows + 3) / 4);
  if (nblk > HIG_NORM_BLOCKS - 1) nblk = HIG_NORM_BLOCKS - 1;
  hipLaunchKernelGGL(masked_mse_kernel, dim3(nblk), dim3(256), 0, hig_stream(s), pred, target, length, B, T,
                     F, dpred, scratch);
  HIG_CHECK_LAUNCH();
  hipLaunchKernelGGL(masked_mse_final_kernel, dim3(1), dim3(256), 0, hig_stream(s), scratch, nblk, loss);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_sumsq_partial(const float* g, int64_t n, float inv_world, float* scratch,
                                 hig_stream_t s) {
  HIG_REQUIRE(g && scratch && n > 0, "hig_sumsq_partial: bad arguments");
  HIG_REQUIRE((reinterpret_cast<uintptr_t>(g) & 15) == 0, "hig_sumsq_partial: g must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(HIG_NORM_BLOCKS), dim3(256), 0, hig_stream(s), g, n, inv_world, scratch);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                             float b2, float eps, float max_norm, float inv_world, const float* scratch,
                             float* gnorm_out, int32_t* step_dev, hig_stream_t s) {
  return hig_clip_adam_lrdev(p, g, m, v, n, lr, nullptr, b1, b2, eps, max_norm, inv_world, scratch, gnorm_out, step_dev, s);
}

extern "C" int hig_clip_adam_lrdev(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                   const float* lr_dev, float b1, float b2, float eps, float max_norm,
                                   float inv_world, const float* scratch, float* gnorm_out, int32_t* step_dev,
                                   hig_stream_t s) {
  return hig_clip_adam_shadow(p, g, m, v, n, lr, lr_dev, b1, b2, eps, max_norm, inv_world, scratch, gnorm_out, step_dev, nullptr, 0, s);
}

extern "C" int hig_clip_adam_shadow(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                    const float* lr_dev, float b1, float b2, float eps, float max_norm,
                                    float inv_world, const float* scratch, float* gnorm_out, int32_t* step_dev,
                                    void* shadow16, int64_t shadow_n, hig_stream_t s) {
  HIG_REQUIRE(p && g && m && v && scratch && step_dev && n > 0, "hig_clip_adam: bad arguments");
  HIG_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                reinterpret_cast<uintptr_t>(v)) & 15) == 0 && (reinterpret_cast<uintptr_t>(shadow16) & 7) == 0,
              "hig_clip_adam: buffers must be 16-byte aligned (the bf16 shadow 8-byte)");
  HIG_REQUIRE(!shadow16 || (shadow_n >= 0 && shadow_n <= n && shadow_n % 4 == 0), "hig_clip_adam: shadow_n must be a multiple of 4, <= n");
  hipLaunchKernelGGL(clip_adam_kernel, dim3(stream_blocks(n / 4 + 1)), dim3(256), 0, hig_stream(s), p, g, m, v,
                     n, lr, b1, b2, eps, max_norm, inv_world, scratch, gnorm_out, step_dev, lr_dev, static_cast<__bf16*>(shadow16), shadow_n);
  HIG_CHECK_LAUNCH();
  hipLaunchKernelGGL(inc_step_kernel, dim3(1), dim3(1), 0, hig_stream(s), step_dev);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// Diagnostic marker: a one-thread kernel whose only purpose is to show up BY NAME in a rocprofv3 kernel trace, so that a
// summariser can cut the launches of one region out of the trace (bench.py brackets the roofline microbenchmark with
// markers 1 / 2: tools/summarize_profiles.py then averages the FFN GEMM launches between them, apart from the same kernel's
// launches inside the forward).  Writes nothing.
namespace {
__global__ void hig_marker_kernel(int id) { (void)id; }
}
extern "C" int hig_debug_marker(int32_t id, hig_stream_t s) {
  hipLaunchKernelGGL(hig_marker_kernel, dim3((unsigned)(id > 0 ? id : 1)), dim3(1), 0, hig_stream(s), id);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
