// Kernels either side of the denoiser: batch assembly from the HBM-resident motion
// bank.  Reference arithmetic: Text2MotionMulDataset.__getitem__, codes/datasets/mul_dataset.py:203-209
// (frame gather + Z-normalisation; the init-pose row is normalised on its first 4 features only).
#include "hig_common.h"

namespace {

// One workgroup per (row r, token t): F contiguous floats gathered from frame `frame_ix[r][t]` of the
// sequence starting at bank + seq_off[r].  ST = float or double: the type numpy promoted to when it
// normalised (float64 statistics make numpy compute in double and round once on the store).
template <typename ST>
__global__ __launch_bounds__(256) void gather_frames_kernel(const float* __restrict__ bank,
                                                            const int64_t* __restrict__ seq_off,
                                                            const int32_t* __restrict__ frame_ix,
                                                            const ST* __restrict__ stats, int T, int F,
                                                            float* __restrict__ out) {
  const int r = blockIdx.y, t = blockIdx.x;
  const float* src = bank + seq_off[r] + (int64_t)frame_ix[(int64_t)r * T + t] * F;
  float* dst = out + ((int64_t)r * T + t) * F;
  const ST* mean = stats;
  const ST* sd = stats + F;
  const ST* imean = stats + 2 * F;
  const ST* isd = imean + 4;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    const float x = src[f];
    float y;
    if (t == 0)
      y = f < 4 ? (float)(((ST)x - imean[f]) / isd[f]) : x;
    else
      y = (float)(((ST)x - mean[f]) / sd[f]);
    dst[f] = y;
  }
}

// ---- post-sampling joint recovery ------------------------------------------------------
// Reference: recover_root_rot_pos / recover_from_ric2 (codes/utils/motion_process.py:362-382, 418-462),
// qinv / qrot (codes/utils/quaternion.py:16-20, 54-73) and the de-normalisation in
// tools/visualization.py:146-152.  One workgroup per person-sample; the two prefix sums over time run
// serially in one lane with a double accumulator (what torch's CPU cumsum does for float tensors).
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// v + 2 * (w * (u x v) + u x (u x v)),  q = (w, u)
__device__ __forceinline__ V3 qrot3(float w, V3 u, V3 v) {
  const V3 uv = cross3(u, v);
  const V3 uuv = cross3(u, uv);
  return V3{v.x + 2.f * (w * uv.x + uuv.x), v.y + 2.f * (w * uv.y + uuv.y), v.z + 2.f * (w * uv.z + uuv.z)};
}

__global__ __launch_bounds__(256) void recover_joints_kernel(const float* __restrict__ motion, const float* __restrict__ stats,
                                                             int T, int F, int J, int init_first,
                                                             float* __restrict__ pos) {
  extern __shared__ float sh[];
  float* ang = sh;           // [T] root yaw
  float* px = sh + T;        // [T] root x
  float* pz = sh + 2 * T;    // [T] root z
  float* ry = sh + 3 * T;    // [T] root height
  const int r = blockIdx.x;
  const float* m = motion + (int64_t)r * (T + 1) * F;
  const float* body = m + (init_first ? F : 0);      // T rows of F
  const float* init = m + (init_first ? 0 : (int64_t)T * F);
  const float* mean = stats;
  const float* sd = stats ? stats + F : nullptr;
  auto feat = [&](int t, int f) -> float {
    const float x = body[(int64_t)t * F + f];
    return stats ? x * sd[f] + mean[f] : x;
  };
  auto ifeat = [&](int f) -> float { return stats ? init[f] * stats[2 * F + 4 + f] + stats[2 * F + f] : init[f]; };
  // step 1: yaw[t] = sum_{s<t} rot_vel[s]
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    ang[t] = t > 0 ? feat(t - 1, 0) : 0.f;
    ry[t] = feat(t, 3);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (int t = 0; t < T; ++t) {
      acc += (double)ang[t];
      ang[t] = (float)acc;
    }
  }
  __syncthreads();
  // step 2: root XZ velocity of frame t-1 rotated into the world frame, then summed over time
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    V3 v{0.f, 0.f, 0.f};
    if (t > 0) {
      v.x = feat(t - 1, 1);
      v.z = feat(t - 1, 2);
    }
    const V3 w = qrot3(cosf(ang[t]), V3{-0.f, -sinf(ang[t]), -0.f}, v);
    px[t] = w.x;
    pz[t] = w.z;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ax = 0.0, az = 0.0;
    for (int t = 0; t < T; ++t) {
      ax += (double)px[t];
      az += (double)pz[t];
      px[t] = (float)ax;
      pz[t] = (float)az;
    }
  }
  __syncthreads();
  // step 3: every (frame, joint): local -> root-yaw frame -> + root XZ -> init rotation -> + init XZ
  const float q0w = ifeat(2), q0y = ifeat(3), ix = ifeat(0), iz = ifeat(1);
  float* out = pos + (int64_t)r * T * J * 3;
  for (int e = threadIdx.x; e < T * J; e += blockDim.x) {
    const int t = e / J, j = e % J;
    V3 p;
    if (j == 0) {
      p = V3{px[t], ry[t], pz[t]};
    } else {
      const int f = 4 + 3 * (j - 1);
      p = qrot3(cosf(ang[t]), V3{-0.f, -sinf(ang[t]), -0.f}, V3{feat(t, f), feat(t, f + 1), feat(t, f + 2)});
      p.x += px[t];
      p.z += pz[t];
    }
    p = qrot3(q0w, V3{0.f, q0y, 0.f}, p);
    p.x += ix;
    p.z += iz;
    out[(int64_t)e * 3] = p.x;
    out[(int64_t)e * 3 + 1] = p.y;
    out[(int64_t)e * 3 + 2] = p.z;
  }
}

}  // namespace

extern "C" int hig_recover_joints(const float* motion, const float* stats, int32_t rows, int32_t T, int32_t F,
                                  int32_t joints, int32_t init_first, float* pos, hig_stream_t stream) {
  HIG_REQUIRE(motion && pos, "hig_recover_joints: null argument");
  HIG_REQUIRE(rows >= 0 && T > 0 && T <= 8192 && joints >= 1 && F >= 4 + 3 * (joints - 1),
              "hig_recover_joints: need 0 < T <= 8192 and F >= 4 + 3 * (joints - 1)");
  if (rows == 0) return HIG_OK;
  hipLaunchKernelGGL(recover_joints_kernel, dim3(rows), dim3(256), (size_t)4 * T * sizeof(float), hig_stream(stream),
                     motion, stats, T, F, joints, init_first, pos);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_gather_frames(const float* bank, const int64_t* seq_off, const int32_t* frame_ix,
                                 const void* stats, int32_t stats_f64, int32_t rows, int32_t T, int32_t F,
                                 float* out, hig_stream_t stream) {
  HIG_REQUIRE(bank && seq_off && frame_ix && stats && out, "hig_gather_frames: null argument");
  HIG_REQUIRE(rows >= 0 && T > 0 && F >= 4, "hig_gather_frames: rows >= 0, T > 0, F >= 4 required");
  if (rows == 0) return HIG_OK;
  const dim3 grid(T, rows);
  if (stats_f64)
    hipLaunchKernelGGL(gather_frames_kernel<double>, grid, dim3(256), 0, hig_stream(stream), bank, seq_off, frame_ix,
                       static_cast<const double*>(stats), T, F, out);
  else
    hipLaunchKernelGGL(gather_frames_kernel<float>, grid, dim3(256), 0, hig_stream(stream), bank, seq_off, frame_ix,
                       static_cast<const float*>(stats), T, F, out);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
