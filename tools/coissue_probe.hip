// Micro-probe (tuning aid): how fast does a VALU-only wave issue while its SIMD partner runs back-to-back
// v_mfma_f32_32x32x16_bf16?  gemm_wsp16.hip pairs a matrix wave (MFMA + ds_read only) with a service wave (epilogue
// arithmetic) on every SIMD; its stamps show ~30 cycles per vector instruction in the service wave.  One workgroup of 512
// threads per CU: waves 0-3 run NM dependent MFMAs (one accumulator chain, or two when `two_acc`), waves 4-7 run NV vector
// instructions of one kind and stamp s_memtime around them.
//   hipcc --offload-arch=gfx950 -O3 -o tools/coissue_probe tools/coissue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)

// KIND: 0 v_add_f32 (independent x8), 1 v_pk_add_f32, 2 v_cvt_pk_bf16_f32, 3 v_exp_f32, 4 v_mul_lo_u32, 5 dependent v_add_f32 chain
template <int KIND>
__global__ __launch_bounds__(512, 2) void probe(int nm, int nv, int mfma_on, int two_acc, int prio, unsigned long long* out, float* sink) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __syncthreads();
  if (wave < 4) {
    if (!mfma_on) return;
    f32x16 a0, a1;
    for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.01f * (lane + e)); y[e] = (__bf16)(0.02f * (lane - e)); }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (two_acc) {
      for (int i = 0; i < nm; i += 2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
      }
    } else {
#pragma unroll 8
      for (int i = 0; i < nm; ++i) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e];
    if (s == 12345.f) sink[tid] = s;
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    return;
  }
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (prio == 3) __builtin_amdgcn_s_setprio(3);
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = 0.5f + lane * 0.01f + k;
  unsigned u = lane + 3;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < nv; i += 8) {
    if constexpr (KIND == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(1.0f));
    } else if constexpr (KIND == 1) {
      f32x2* p = reinterpret_cast<f32x2*>(v);
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k & 3]) : "v"(p[(k + 1) & 3]));
    } else if constexpr (KIND == 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        unsigned r;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v[k]), "v"(v[(k + 1) & 7]));
        u ^= r;
      }
    } else if constexpr (KIND == 3) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
    } else if constexpr (KIND == 4) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u) : "v"(lane | 1));
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(1.0f));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = (float)u;
  for (int k = 0; k < 8; ++k) s += v[k];
  if (s == 12345.f) sink[tid] = s;
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, unsigned long long* out, float* sink) {
  const int nm = 512, nv = 512;
  for (int mf = 0; mf < 2; ++mf)
    for (int two = 0; two < (mf ? 2 : 1); ++two)
      for (int prio : {0, 3}) {
        if (!mf && prio) continue;
        CK(hipMemset(out, 0, 256 * 16 * 8));
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, nm, nv, mf, two, prio, out, sink);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(256 * 16);
        CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> sv, mv;
        for (int b = 0; b < 256; ++b) {
          for (int w = 4; w < 8; ++w) sv.push_back((double)h[b * 16 + w] / nv);
          if (mf) for (int w = 0; w < 4; ++w) mv.push_back((double)h[b * 16 + w] / nm);
        }
        std::sort(sv.begin(), sv.end());
        std::sort(mv.begin(), mv.end());
        printf("%-22s partner %-26s prio %d : %6.1f cycles per vector instruction (median)   %s\n", name,
               !mf ? "idle" : (two ? "MFMA x2 accumulators" : "MFMA one dependent chain"), prio, sv[sv.size() / 2],
               mf ? (std::to_string(mv[mv.size() / 2]).substr(0, 5) + " cycles per MFMA").c_str() : "");
      }
}

int main() {
  unsigned long long* out;
  float* sink;
  CK(hipMalloc(&out, 256 * 16 * 8));
  CK(hipMalloc(&sink, 4096));
  run<0>("v_add_f32 (indep.)", out, sink);
  run<5>("v_add_f32 (chain)", out, sink);
  run<1>("v_pk_add_f32", out, sink);
  run<2>("v_cvt_pk_bf16_f32", out, sink);
  run<3>("v_exp_f32", out, sink);
  run<4>("v_mul_lo_u32", out, sink);
  return 0;
}
