// Micro-probe (tuning aid): ds_read_b128 rate of a CU for the fragment-read address patterns of the bf16 GEMM kernels.
// W waves (one or two per SIMD) each issue N reads with 8 in flight; the pattern decides which 16-byte chunks the 64 lanes touch.
//   pattern 0: lane l reads chunk l of one 1-KiB row                                   (linear: the best case)
//   pattern 1: gemm_wsp16 / 16x16x32: lane (n = l & 15, g = l >> 4) reads row n, chunk (4 q + g) ^ n
//   pattern 2: gemm_ws16 / 32x32x16: lane (r = l & 31, h = l >> 5) reads row r, chunk (2 q + h) ^ (r & 15)
//   pattern 3: pattern 1 without the XOR swizzle (every lane of a column the same chunk: the worst case)
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_read_probe tools/lds_read_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)

template <int PAT, int U, int WIDTH>
__global__ __launch_bounds__(512) void probe(int n, unsigned long long* out, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 1024 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  unsigned base[4];
  for (int q = 0; q < 4; ++q) {
    if (PAT == 0) base[q] = q * 1024 + lane * 16;
    else if (PAT == 1) base[q] = (lane & 15) * 1024 + 16 * ((4 * q + (lane >> 4)) ^ (lane & 15));
    else if (PAT == 2) base[q] = (lane & 31) * 1024 + 16 * ((2 * q + (lane >> 5)) ^ (lane & 15));
    else base[q] = (lane & 15) * 1024 + 16 * (4 * q + (lane >> 4));
  }
  const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  using RT = typename std::conditional<WIDTH == 16, u32x4, u32x2>::type;
  RT r[U];
  unsigned acc = 0;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
  for (int u = 0; u < U; ++u) r[u] = *reinterpret_cast<const __attribute__((address_space(3))) RT*>(lb + base[u & 3] + 256 * ((u >> 2) & 3));
  for (int i = U; i < n; i += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc ^= r[u].x;
      r[u] = *reinterpret_cast<const __attribute__((address_space(3))) RT*>(lb + base[u & 3] + 256 * ((u >> 2) & 3) + ((i & U) ? 16384 : 0));
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) acc ^= r[u].y;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (acc == 0x1234567u) sink[0] = acc;
  if (lane == 0) out[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
}

template <int PAT, int U, int WIDTH>
void run(const char* name, unsigned long long* out, unsigned* sink) {
  const int n = 2048;
  for (int waves : {1, 4, 8}) {
    CK(hipMemset(out, 0, 256 * 16 * 8));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<PAT, U, WIDTH>), dim3(256), dim3(64 * waves), 0, 0, n, out, sink);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * 16);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> v;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < waves; ++w) v.push_back((double)h[b * 16 + w]);
    std::sort(v.begin(), v.end());
    const double cyc = v[v.size() / 2];
    printf("%-44s b%-3d %2d in flight, %d waves: %6.1f cycles per read per wave  ->  %6.1f B/clk/CU\n", name, WIDTH * 8, U, waves, cyc / n, 64.0 * WIDTH * n * waves / cyc);
  }
}

int main() {
  unsigned long long* out;
  unsigned* sink;
  CK(hipMalloc(&out, 256 * 16 * 8));
  CK(hipMalloc(&sink, 64));
  run<0, 8, 16>("linear (lane l: chunk l of one row)", out, sink);
  run<0, 16, 16>("linear (lane l: chunk l of one row)", out, sink);
  run<0, 4, 16>("linear (lane l: chunk l of one row)", out, sink);
  run<0, 8, 8>("linear, 8-byte reads", out, sink);
  run<0, 16, 8>("linear, 8-byte reads", out, sink);
  run<1, 8, 16>("16x16x32 fragment, XOR swizzle (gemm_wsp16)", out, sink);
  run<1, 16, 16>("16x16x32 fragment, XOR swizzle (gemm_wsp16)", out, sink);
  run<2, 8, 16>("32x32x16 fragment, XOR swizzle (gemm_ws16)", out, sink);
  run<3, 8, 16>("16x16x32 fragment, no swizzle", out, sink);
  return 0;
}
