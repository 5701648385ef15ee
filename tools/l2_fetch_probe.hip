// Micro-probe (tuning aid): what can ONE compute unit pull from its XCD's L2, by which instruction
// and at which depth?  The short-K bf16 GEMMs of this model (K = 512: FFN linear1, q/k/v, stylization out) need
// 450-650 KB of operands per CU and 16K cycles of matrix work -- whether they can ever run at the matrix rate depends
// on this number alone.  Every workgroup (one per CU, 256 workgroups) reads `bytes_per_wg` from an L2-resident region
// shared by all workgroups (default 2 MB: X rows of one XCD's row range + a weight panel), wrapping around.
//   mode 0  global_load_lds_dwordx4     : 1 KiB contiguous per wave-instruction, straight into an LDS ring, nothing read back
//   mode 1  global_load_dwordx4 -> VGPR : 1 KiB contiguous per wave-instruction (8 whole 128-B lines)
//   mode 2  global_load_dwordx4 -> VGPR : 16 rows x 64 B per wave-instruction (the B fragment of v_mfma_f32_16x16x32_bf16, row pitch 1 KiB)
//   mode 3  global_load_dwordx4 -> VGPR : 32 rows x 32 B per wave-instruction (the fragment of v_mfma_f32_32x32x16_bf16)
//   mode 4  buffer_load_dwordx4 ... lds : as mode 0 through a buffer descriptor (the weight-stationary kernel's form)
//   mode 5  mode 1 + ds_write_b128      : register staging, the loaded KiB written to LDS
// U = wave-instructions in flight per wave (x 1 KiB), W = waves per workgroup: bytes in flight per CU = U x W KiB.
// Output: GB/s per CU from HIP events over `reps` launches, and bytes per shader clock from s_memtime (median workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o tools/l2_fetch_probe tools/l2_fetch_probe.hip ; tools/l2_fetch_probe [region_KB] [KB_per_wg]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int U>
__global__ __launch_bounds__(1024) void probe(const char* __restrict__ src, unsigned region, unsigned per_wave_instr, unsigned long long* cyc,
                                              unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  // wave-instruction n of this workgroup reads KiB number (blockIdx * 37 + n) of the region (every CU walks the whole region)
  unsigned lane_off;
  if (MODE == 2) lane_off = (lane & 15) * 1024u + (lane >> 4) * 16u;          // 16 rows x 64 B
  else if (MODE == 3) lane_off = (lane & 31) * 1024u + (lane >> 5) * 16u;     // 32 rows x 32 B
  else lane_off = lane * 16u;
  // modes 2 / 3 cover a 16 KiB / 32 KiB block of rows with 16 / 32 instructions (column step 64 B / 32 B)
  auto addr = [&](unsigned n) -> unsigned {
    unsigned g = blockIdx.x * 37u * 1024u;
    if (MODE == 2) { const unsigned blk = n >> 4, c = n & 15; return (g + blk * 16384u + c * 64u + lane_off) & (region - 1); }
    if (MODE == 3) { const unsigned blk = n >> 5, c = n & 31; return (g + blk * 32768u + c * 32u + lane_off) & (region - 1); }
    return (g + n * 1024u + lane_off) & (region - 1);   // region is a power of two
  };
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  unsigned acc = 0;
  const unsigned n0 = wave, step = nw;
  if constexpr (MODE == 0 || MODE == 4) {
    char* ring = lds + wave * (U * 1024);
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)region, 0x00020000);
    unsigned n = n0;
    for (unsigned i = 0; i < per_wave_instr; i += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned a = addr(n);
        n += step;
        if constexpr (MODE == 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + a), (__attribute__((address_space(3))) void*)(ring + u * 1024), 16, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + u * 1024), 16, (int)a, 0, 0, 0);
        wait_vm<U - 1>();   // never more than U in flight: slot u is only re-issued once instruction (n - U) has landed
      }
    }
    wait_vm<0>();
  } else {
    u32x4 r[U];
    unsigned n = n0;
#pragma unroll
    for (int u = 0; u < U; ++u) { r[u] = *reinterpret_cast<const u32x4*>(src + addr(n)); n += step; }
    char* wr = lds + tid * 16;
    for (unsigned i = U; i < per_wave_instr; i += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (MODE == 5) *reinterpret_cast<u32x4*>(wr + (u & 3) * 16384) = r[u];
        else acc ^= r[u].x ^ r[u].y ^ r[u].z ^ r[u].w;
        r[u] = *reinterpret_cast<const u32x4*>(src + addr(n));
        n += step;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= r[u].x ^ r[u].w;
  }
  __syncthreads();
  if (tid == 0) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    cyc[blockIdx.x] = t1 - t0;
  }
  if (acc == 0x12345u) sink[0] = acc;
}

template <int MODE, int U>
void run(const char* name, int waves, const char* src, unsigned region, unsigned kb_per_wg, unsigned long long* cyc, unsigned* sink, int reps) {
  const unsigned per_wave = (kb_per_wg / waves / U) * U;
  if (per_wave < (unsigned)U) return;
  const size_t ldsb = (MODE == 0 || MODE == 4) ? (size_t)waves * U * 1024 : (MODE == 5 ? 65536 : 0);
  if (ldsb > 160 * 1024) return;
  auto kern = probe<MODE, U>;
  if (ldsb > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), ldsb, 0, src, region, per_wave, cyc, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), ldsb, 0, src, region, per_wave, cyc, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(256);
  CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double bytes = (double)per_wave * waves * 1024.0;
  const double us = ms * 1000.0 / reps;
  printf("%-22s W=%2d U=%2d inflight=%4d KB  %7.2f us/launch  %6.1f GB/s/CU  %5.2f TB/s chip  cycles med %7llu max %7llu  %5.1f B/clk/CU\n", name, waves, U,
         waves * U, us, bytes / us * 1e-3, bytes * 256 / us * 1e-6, h[128], h[255], bytes / (double)h[128]);
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  unsigned region = (argc > 1 ? atoi(argv[1]) : 2048) * 1024u;
  while (region & (region - 1)) region &= region - 1;   // power of two
  const unsigned kb = argc > 2 ? atoi(argv[2]) : 512;
  const int reps = 50;
  char* src;
  unsigned long long* cyc;
  unsigned* sink;
  CK(hipMalloc(&src, region + 65536));
  CK(hipMalloc(&cyc, 256 * 8));
  CK(hipMalloc(&sink, 64));
  std::vector<unsigned> h((region + 65536) / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i * 2654435761u;
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("region %u KB (L2-resident per XCD when <= 4096), %u KB per workgroup, 256 workgroups\n", region / 1024, kb);
#define ROW(MODE, NAME)                                                       \
  for (int w : {4, 8, 16}) {                                                  \
    run<MODE, 2>(NAME, w, src, region, kb, cyc, sink, reps);                  \
    run<MODE, 4>(NAME, w, src, region, kb, cyc, sink, reps);                  \
    run<MODE, 8>(NAME, w, src, region, kb, cyc, sink, reps);                  \
    run<MODE, 16>(NAME, w, src, region, kb, cyc, sink, reps);                 \
  }
  ROW(0, "glds 1KiB rows")
  ROW(4, "buffer..lds 1KiB rows")
  ROW(1, "vgpr 1KiB rows")
  ROW(2, "vgpr 16 rows x 64B")
  ROW(3, "vgpr 32 rows x 32B")
  ROW(5, "vgpr 1KiB + ds_write")
  return 0;
}
