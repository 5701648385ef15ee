// Micro-probe (tuning aid): what does a partner wave's instruction stream cost a wave that issues back-to-back exact-fp32 MFMAs
// on the same SIMD?  gemm_wsp32.hip pairs a matrix wave (v_mfma_f32_16x16x4_f32 + ds_read_b128 only) with a service wave
// (LDS-DMA issue + epilogue arithmetic) on every SIMD.  The fp32 MFMA runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD): if it
// executes on the SIMD's vector lanes, every vector instruction of the partner takes its cycles out of the matrix stream.
// One workgroup of 512 threads per CU: waves 0-3 run NM MFMAs, waves 4-7 run the partner stream for about as long; both
// stamp s_memtime.  Reported: cycles per MFMA (alone: 32 for 16x16x4, 64 for 32x32x2) and the cycles the matrix wave LOSES per
// partner instruction.
//   hipcc --offload-arch=gfx950 -O3 -o tools/coissue32_probe tools/coissue32_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)

// MF: 0 = 16x16x4, two chains; 1 = 32x32x2, one chain; 2 = 32x32x2, two chains
// KIND: 0 none, 1 v_add_f32, 2 v_mov_b32, 3 s_add_u32, 4 v_exp_f32, 5 ds_read_b128, 6 v_fma_f32, 7 s_nop 0, 8 global_load_dwordx4 (L2 hits),
//       9 v_add_f32 one per 4 s_nop
template <int MF, int KIND>
__global__ __launch_bounds__(512, 2) void probe(int nm, int nv, int prio, unsigned long long* out, float* sink, const float* src) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  lds[tid] = (float)tid;
  __syncthreads();
  if (wave < 4) {
    const float x = 0.01f * (lane + 1), y = 0.02f * (lane - 7);
    unsigned long long t0, t1;
    float s = 0;
    if constexpr (MF == 0) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 8
      for (int i = 0; i < nm; i += 2) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      for (int e = 0; e < 4; ++e) s += a0[e] + a1[e];
    } else {
      f32x16 a0, a1;
      for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
      if constexpr (MF == 1) {
#pragma unroll 8
        for (int i = 0; i < nm; ++i) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      } else {
#pragma unroll 8
        for (int i = 0; i < nm; i += 2) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        }
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      for (int e = 0; e < 16; ++e) s += a0[e] + a1[e];
    }
    if (s == 12345.f) sink[tid] = s;
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    return;
  }
  if (KIND == 0) return;
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (prio == 3) __builtin_amdgcn_s_setprio(3);
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = 0.5f + lane * 0.01f + k;
  unsigned su = 3;
  f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + lane * 16;
  const float* gp = src + (size_t)blockIdx.x * 4096 + (wave - 4) * 1024 + lane * 4;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < nv; i += 8) {
    if constexpr (KIND == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(1.0f));
    } else if constexpr (KIND == 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_mov_b32 %0, %1" : "=v"(v[k]) : "v"(v[(k + 1) & 7]));
    } else if constexpr (KIND == 3) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("s_add_u32 %0, %0, 7" : "+s"(su)::"scc");
    } else if constexpr (KIND == 4) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
    } else if constexpr (KIND == 5) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        f32x4 r;
        asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(la));
        asm volatile("s_waitcnt lgkmcnt(4)");
        acc4 = r;
      }
    } else if constexpr (KIND == 6) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k]) : "v"(1.0001f));
    } else if constexpr (KIND == 7) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("s_nop 0");
    } else if constexpr (KIND == 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        f32x4 r;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(gp + (k & 3) * 256));
        asm volatile("s_waitcnt vmcnt(6)");
        acc4 = r;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(1.0f));
        asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = (float)su + acc4.x;
  for (int k = 0; k < 8; ++k) s += v[k];
  if (s == 12345.f) sink[tid] = s;
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

static double g_alone[3];

template <int MF, int KIND>
void run(const char* name, unsigned long long* out, float* sink, const float* src, int nv) {
  const int nm = 1024;
  for (int prio : {0, 1}) {
    if (KIND == 0 && prio) continue;
    CK(hipMemset(out, 0, 256 * 16 * 8));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<MF, KIND>), dim3(256), dim3(512), 0, 0, nm, nv, prio, out, sink, src);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * 16);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> sv, mv;
    for (int b = 0; b < 256; ++b) {
      for (int w = 4; w < 8; ++w) if (KIND) sv.push_back((double)h[b * 16 + w]);
      for (int w = 0; w < 4; ++w) mv.push_back((double)h[b * 16 + w]);
    }
    std::sort(sv.begin(), sv.end());
    std::sort(mv.begin(), mv.end());
    const double m = mv[mv.size() / 2], s = KIND ? sv[sv.size() / 2] : 0.0;
    if (KIND == 0) { g_alone[MF] = m; printf("%-14s partner idle                 : %6.2f cycles per MFMA\n", MF == 0 ? "16x16x4 x2" : (MF == 1 ? "32x32x2 x1" : "32x32x2 x2"), m / nm); continue; }
    // partner instructions issued while the matrix wave ran (the partner stream is sized to outlast it or nearly so)
    const double overlap = s > m ? nv * (m / s) : nv;
    printf("%-14s partner %-20s prio %d : %6.2f cycles per MFMA   partner %6.1f cycles per instruction   matrix wave loses %5.1f cycles per partner instruction\n",
           MF == 0 ? "16x16x4 x2" : (MF == 1 ? "32x32x2 x1" : "32x32x2 x2"), name, prio, m / nm, s / nv, (m - g_alone[MF]) / overlap);
  }
}

template <int MF>
void all(unsigned long long* out, float* sink, const float* src) {
  const int base = MF == 0 ? 1024 * 32 : 1024 * 64;   // matrix cycles
  run<MF, 0>("", out, sink, src, 0);
  run<MF, 1>("v_add_f32", out, sink, src, base / 8);
  run<MF, 6>("v_fma_f32", out, sink, src, base / 8);
  run<MF, 2>("v_mov_b32", out, sink, src, base / 8);
  run<MF, 4>("v_exp_f32", out, sink, src, base / 16);
  run<MF, 3>("s_add_u32", out, sink, src, base / 8);
  run<MF, 7>("s_nop 0", out, sink, src, base / 8);
  run<MF, 5>("ds_read_b128", out, sink, src, base / 16);
  run<MF, 8>("global_load_dwordx4", out, sink, src, base / 64);
  run<MF, 9>("v_add_f32 + 4 s_nop", out, sink, src, base / 24);
}

int main() {
  unsigned long long* out;
  float *sink, *src;
  CK(hipMalloc(&out, 256 * 16 * 8));
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&src, 256 * 4096 * 4 + 65536));
  CK(hipMemset(src, 0, 256 * 4096 * 4 + 65536));
  all<0>(out, sink, src);
  all<1>(out, sink, src);
  all<2>(out, sink, src);
  return 0;
}
