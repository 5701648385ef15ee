// Micro-probe (tuning aid): the matrix-wave instruction stream of gemm_wsp32.hip in isolation -- 128 weight-fragment registers,
// one ds_read_b128 of X fragments per k-step, MFMAs on them -- for the two exact-fp32 MFMA shapes:
//   MF 0: v_mfma_f32_16x16x4_f32, two chains, 8 MFMAs per k-step (16 k-steps per 16-row tile)
//   MF 1: v_mfma_f32_32x32x2_f32, one chain, 4 MFMAs per k-step (32 k-steps per 32-row tile)
// 512-thread workgroups, one per CU: waves 0-3 run the stream for `nt` tiles (barrier per tile when `bar`), waves 4-7 only join
// the barriers (or exit at once when bar = 0).  Reports cycles per MFMA (ideal 32 / 64).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma32_stream_probe tools/mfma32_stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)

template <int MF, int XS, int RP>
__global__ __launch_bounds__(512, 2) void probe(int nt, int bar, int rd, unsigned long long* out, float* sink, const float* src) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16384; i += 512) lds[i] = src[i] ;
  __syncthreads();
  if (wave >= 4) {
    if (bar) for (int t = 0; t < nt; ++t) __builtin_amdgcn_s_barrier();
    return;
  }
  constexpr int NKS = MF == 0 ? 16 : 32;
  f32x4 wf[NKS][MF == 0 ? 2 : 1];
#pragma unroll
  for (int k = 0; k < NKS; ++k)
#pragma unroll
    for (int c = 0; c < (MF == 0 ? 2 : 1); ++c) wf[k][c] = *reinterpret_cast<const f32x4*>(src + ((k * 2 + c) * 64 + lane) * 4);
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + (lane & 15) * 2048 + (lane >> 4) * 16;
  constexpr int RS = XS + RP;
  f32x4 ring[RS];
#pragma unroll
  for (int k = 0; k < RS; ++k) ring[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned long long t0, t1;
  float s = 0;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if constexpr (MF == 0) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    for (int t = 0; t < nt; ++t) {
      if (bar) { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(XS) : "memory"); __builtin_amdgcn_s_barrier(); }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ks][0][j], ring[ks % RS][j], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ks][1][j], ring[ks % RS][j], a1, 0, 0, 0);
          if (RP && j == 0 && rd) ring[(ks + XS) % RS] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(la + 64 * (ks & 15) + 1024 * (ks >> 4));
        }
        if (!RP && rd) ring[ks % RS] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(la + 64 * (ks & 15) + 1024 * (ks >> 4));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int e = 0; e < 4; ++e) s += a0[e] + a1[e];
  } else {
    f32x16 a0;
    for (int e = 0; e < 16; ++e) a0[e] = 0.f;
    for (int t = 0; t < nt; ++t) {
      if (bar) { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(XS) : "memory"); __builtin_amdgcn_s_barrier(); }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[ks][0][j], ring[ks % RS][j], a0, 0, 0, 0);
          if (RP && j == 0 && rd) ring[(ks + XS) % RS] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(la + 64 * (ks & 15) + 1024 * (ks >> 4));
        }
        if (!RP && rd) ring[ks % RS] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(la + 64 * (ks & 15) + 1024 * (ks >> 4));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int e = 0; e < 16; ++e) s += a0[e];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (s == 12345.f) sink[tid] = s;
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MF, int XS, int RP>
void run(unsigned long long* out, float* sink, const float* src) {
  const int nt = 24;
  for (int bar = 0; bar < 2; ++bar)
    for (int rd = 0; rd < 2; ++rd) {
      CK(hipMemset(out, 0, 256 * 16 * 8));
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<MF, XS, RP>), dim3(256), dim3(512), 0, 0, nt, bar, rd, out, sink, src);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(256 * 16);
      CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> mv;
      for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 4; ++w) mv.push_back((double)h[b * 16 + w]);
      std::sort(mv.begin(), mv.end());
      printf("%-10s XS %d ring %d  barrier per tile %d  fragment reads %d : %6.2f cycles per MFMA (median), p90 %6.2f\n", MF == 0 ? "16x16x4 x2" : "32x32x2 x1", XS, XS + RP, bar, rd,
             mv[mv.size() / 2] / (nt * 128.0), mv[mv.size() * 9 / 10] / (nt * 128.0));
    }
}

int main() {
  unsigned long long* out;
  float *sink, *src;
  CK(hipMalloc(&out, 256 * 16 * 8));
  CK(hipMalloc(&sink, 4096));
  std::vector<float> h(65536);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.001f - 0.5f;
  CK(hipMalloc(&src, h.size() * 4));
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  run<0, 2, 0>(out, sink, src);
  run<0, 1, 1>(out, sink, src);
  run<0, 3, 1>(out, sink, src);
  run<0, 2, 2>(out, sink, src);
  run<1, 2, 0>(out, sink, src);
  run<1, 1, 1>(out, sink, src);
  run<1, 3, 1>(out, sink, src);
  run<1, 2, 2>(out, sink, src);
  return 0;
}
