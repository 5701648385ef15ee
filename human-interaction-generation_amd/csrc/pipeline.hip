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

}  // namespace

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
