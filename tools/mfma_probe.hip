// Micro-probe (tuning aid): how close can a wave-level loop get to the fp32 MFMA peak when we add,
// one at a time, the ingredients of the GEMM main loop?  hipcc --offload-arch=gfx950 -O3.
//   variant 0: bare v_mfma_f32_32x32x2_f32 chain(s)
//   variant 1: + 2 ds_read_b128 per 4 MFMAs (operands from LDS)
//   variant 2: + one __syncthreads per 16 MFMAs
//   variant 3: + 4 global_load_dwordx4 + 4 ds_write_b128 per 16 MFMAs (full staging traffic)
//   variant 4: variant 2 + the 4 global loads only (consumed by a VALU add, no LDS write)
//   variant 5: variant 2 + the 4 ds_write_b128 only (constant registers, no global load)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT, int NACC>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* __restrict__ out, int iters, int bmask, int rnd) {
  __shared__ __attribute__((aligned(16))) float lds[2][2 * 64 * 36];
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  // rnd != 0: operands with random mantissas / signs (data-dependent power -> clock); else small integers
  for (int i = tid; i < 2 * 2 * 64 * 36; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    (&lds[0][0])[i] = rnd ? ((float)(int)(h & 0xffffff) - 8388608.0f) * (1.0f / 8388608.0f) : (float)(i & 7);
  }
  __syncthreads();
  float4 stage[4];
  float sink = 0.f;
  for (int p = 0; p < 4; ++p) stage[p] = make_float4(tid, 1.f, 2.f, 3.f);
  const float* gp = g + (size_t)(blockIdx.x & bmask) * 4096 + tid * 4;  // bmask small: every block re-reads the same L2-resident rows
  float xa = 1.0f + lane, xb = 0.5f;
  for (int it = 0; it < iters; ++it) {
    const float* sx = lds[it & 1];
    if (VARIANT == 3 || VARIANT == 4) {
#pragma unroll
      for (int p = 0; p < 4; ++p) stage[p] = *reinterpret_cast<const float4*>(gp + ((it * 4 + p) & 63) * 1024);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float xf[4], yf[4];
      if (VARIANT >= 1) {
        const float4 a4 = *reinterpret_cast<const float4*>(sx + (lane & 31) * 36 + ks * 8 + 4 * (lane >> 5));
        const float4 b4 = *reinterpret_cast<const float4*>(sx + 64 * 36 + (lane & 31) * 36 + ks * 8 + 4 * (lane >> 5));
        xf[0] = a4.x; xf[1] = a4.y; xf[2] = a4.z; xf[3] = a4.w;
        yf[0] = b4.x; yf[1] = b4.y; yf[2] = b4.z; yf[3] = b4.w;
      } else {
        xf[0] = xf[1] = xf[2] = xf[3] = xa;
        yf[0] = yf[1] = yf[2] = yf[3] = xb;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(yf[j], xf[j], acc[a], 0, 0, 0);
    }
    if (VARIANT == 3 || VARIANT == 5) {
      float* dst = lds[(it + 1) & 1];
#pragma unroll
      for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(dst + ((tid >> 3) + 32 * p) * 36 + 4 * (tid & 7)) = stage[p];
    }
    if (VARIANT == 4) {
#pragma unroll
      for (int p = 0; p < 4; ++p) sink += stage[p].x + stage[p].w;
    }
    if (VARIANT >= 2) __syncthreads();
  }
  float s = sink;
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}

// BK = 64 shape of the same loop: 8 global_load_dwordx4 + 8 ds_write_b128 + 16 ds_read_b128 + 32 MFMAs
// (one accumulator) per barrier; 69.6 KB of LDS -> 2 workgroups per CU.
__global__ __launch_bounds__(256) void probe_bk64(const float* __restrict__ g, float* __restrict__ out, int iters, int bmask) {
  __shared__ __attribute__((aligned(16))) float lds[2][2 * 64 * 68];
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int i = tid; i < 2 * 2 * 64 * 68; i += 256) (&lds[0][0])[i] = (float)(i & 7);
  __syncthreads();
  float4 stage[8];
  const float* gp = g + (size_t)(blockIdx.x & bmask) * 4096 + tid * 4;
  for (int it = 0; it < iters; ++it) {
    const float* sx = lds[it & 1];
#pragma unroll
    for (int p = 0; p < 8; ++p) stage[p] = *reinterpret_cast<const float4*>(gp + ((it * 8 + p) & 63) * 1024);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float4 a4 = *reinterpret_cast<const float4*>(sx + (lane & 31) * 68 + ks * 8 + 4 * (lane >> 5));
      const float4 b4 = *reinterpret_cast<const float4*>(sx + 64 * 68 + (lane & 31) * 68 + ks * 8 + 4 * (lane >> 5));
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.x, a4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.y, a4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.z, a4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.w, a4.w, acc, 0, 0, 0);
    }
    float* dst = lds[(it + 1) & 1];
#pragma unroll
    for (int p = 0; p < 8; ++p) *reinterpret_cast<float4*>(dst + ((tid >> 4) + 16 * p) * 68 + 4 * (tid & 15)) = stage[p];
    __syncthreads();
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[e];
  out[(size_t)blockIdx.x * 256 + tid] = s;
}
void run_bk64(int blocks, const float* g, float* out, int bmask) {
  const int iters = 1000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe_bk64, dim3(blocks), dim3(256), 0, 0, g, out, 10, bmask);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe_bk64, dim3(blocks), dim3(256), 0, 0, g, out, iters, bmask);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
  printf("%-34s blocks=%5d acc/wave=1  %.3f ms  %.1f TFLOP/s\n", bmask == 15 ? "BK=64 loop, L2-resident" : "BK=64 loop", blocks, ms, flops / ms / 1e9);
}

template <int V, int NACC>
void run(const char* name, int blocks, const float* g, float* out, int bmask = 0xffff, int rnd = 0) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<V, NACC>), dim3(blocks), dim3(256), 0, 0, g, out, 10, bmask, rnd);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<V, NACC>), dim3(blocks), dim3(256), 0, 0, g, out, iters, bmask, rnd);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 /*waves*/ * iters * 16 * NACC * 4096.0;
  printf("%-34s blocks=%5d acc/wave=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, NACC, ms, flops / ms / 1e9);
}

int main() {
  float *g, *out;
  hipMalloc(&g, (size_t)4096 * 4096 * 4);
  hipMemset(g, 0, (size_t)4096 * 4096 * 4);
  hipMalloc(&out, (size_t)4096 * 256 * 4);
  for (int bpc = 1; bpc <= 4; ++bpc) {
    const int blocks = 256 * bpc;
    printf("--- %d block(s) of 4 waves per CU\n", bpc);
    run<0, 1>("bare mfma", blocks, g, out);
    run<0, 4>("bare mfma", blocks, g, out);
    run<1, 1>("+ds_read_b128", blocks, g, out);
    run<1, 4>("+ds_read_b128", blocks, g, out);
    run<2, 1>("+barrier/16 mfma", blocks, g, out);
    run<2, 4>("+barrier/64 mfma", blocks, g, out);
    run<3, 1>("+global load + ds_write", blocks, g, out);
    run<3, 4>("+global load + ds_write", blocks, g, out);
    run<4, 1>("+global load only", blocks, g, out);
    run<5, 1>("+ds_write only", blocks, g, out);
    run<2, 1>("+barrier, RANDOM operands", blocks, g, out, 0xffff, 1);
    run<2, 4>("+barrier, RANDOM operands", blocks, g, out, 0xffff, 1);
    run<4, 1>("+global load only, L2-resident", blocks, g, out, 15);
    run<3, 1>("+global load + ds_write, L2-res.", blocks, g, out, 15);
    if (bpc <= 2) {
      run_bk64(blocks, g, out, 0xffff);
      run_bk64(blocks, g, out, 15);
    }
  }
  return 0;
}
