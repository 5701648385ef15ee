// Linear ("efficient") attention of the reference denoiser, forward and backward.
// Reference: LinearTemporalSelfAttention / LinearTemporalCrossAttention.forward,
// codes/models/transformer.py:110-117 and :146-153:
//     q = softmax over the hd channels of each token           (row softmax)
//     k = softmax over the tokens, per channel, with -1e6 mask (column softmax)
//     A[b,h] = k^T v  (hd x hd);   y = q A
// Masked rows get exp(-1e6 - max) == 0 exactly in fp32 and v*mask == 0, so they are skipped.
//
// These contractions are fp32 with hd x hd outputs: on gfx950 the f32 MFMA runs at the VALU
// rate, so they stay on the VALU with LDS-staged tiles; the kernels are bound by LDS/VALU
// issue on tiny tiles and by HBM on the (rows x hd) streams.  One workgroup per (sample, head);
// channel c of head h lives at column h*hd + c of the (rows, d) activation.
#include <stdlib.h>

#include "hig_common.h"

int hig_chip_cus();   // hig_host.h: compute units of the current device

namespace {

constexpr int CH = 64;  // rows staged per step

template <int HD> struct Patch {  // register patch of the hd x hd context per thread
  static constexpr int PC = HD >= 128 ? 8 : HD >= 64 ? 4 : HD >= 32 ? 2 : 1;
  static constexpr int PL = PC;
  static constexpr int TL = HD / PL;  // threads along l
};

// ---------------------------------------------------------------------------------------------
// ctx: A[b,h][c][l] = sum_r softmax_r(K)[r,c] * V[r,l],  kstat[b][h*HD+c] = (max, sum exp)
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void ctx_kernel(const float* __restrict__ K,
                                                  const float* __restrict__ V, int64_t ld, int rows,
                                                  int H, const int64_t* __restrict__ length,
                                                  float* __restrict__ A, float* __restrict__ kstat) {
  constexpr int LDP = HD + 4;
  constexpr int PC = Patch<HD>::PC, PL = Patch<HD>::PL, TL = Patch<HD>::TL;
  constexpr int RG = 256 / HD;
  __shared__ __attribute__((aligned(16))) float sP[CH * LDP];
  __shared__ __attribute__((aligned(16))) float sV[CH * LDP];
  __shared__ float sred[256];
  __shared__ float smax[HD];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  int len = rows;
  if (length) len = (int)min<int64_t>(max<int64_t>(length[b], 0), rows);
  const float* Kb = K + (int64_t)b * rows * ld + h * HD;
  const float* Vb = V + (int64_t)b * rows * ld + h * HD;

  {  // pass 1: column max over the valid rows
    const int c = tid % HD, rg = tid / HD;
    float m = -INFINITY;
    for (int r = rg; r < len; r += RG) m = fmaxf(m, Kb[(int64_t)r * ld + c]);
    sred[tid] = m;
    __syncthreads();
    if (tid < HD) {
      for (int g2 = 1; g2 < RG; ++g2) m = fmaxf(m, sred[g2 * HD + tid]);
      smax[tid] = m;
    }
    __syncthreads();
  }

  const int pc = tid / TL, pl = tid % TL;
  const int c0 = pc * PC, l0 = pl * PL;
  const bool active = c0 < HD;
  float acc[PC][PL], ks[PC];
#pragma unroll
  for (int i = 0; i < PC; ++i) {
    ks[i] = 0.f;
#pragma unroll
    for (int j = 0; j < PL; ++j) acc[i][j] = 0.f;
  }
  for (int r0 = 0; r0 < len; r0 += CH) {
    for (int idx = tid; idx < CH * HD; idx += 256) {
      const int rr = idx / HD, cc = idx % HD, r = r0 + rr;
      float p = 0.f, v = 0.f;
      if (r < len) {
        p = __expf(Kb[(int64_t)r * ld + cc] - smax[cc]);
        v = Vb[(int64_t)r * ld + cc];
      }
      sP[rr * LDP + cc] = p;
      sV[rr * LDP + cc] = v;
    }
    __syncthreads();
    if (active) {
      const int nr = min(CH, len - r0);
      for (int rr = 0; rr < nr; ++rr) {
        float p[PC], v[PL];
#pragma unroll
        for (int i = 0; i < PC; ++i) p[i] = sP[rr * LDP + c0 + i];
#pragma unroll
        for (int j = 0; j < PL; ++j) v[j] = sV[rr * LDP + l0 + j];
#pragma unroll
        for (int i = 0; i < PC; ++i) {
          ks[i] += p[i];
#pragma unroll
          for (int j = 0; j < PL; ++j) acc[i][j] = fmaf(p[i], v[j], acc[i][j]);
        }
      }
    }
    __syncthreads();
  }
  if (active) {
    float* Ab = A + (int64_t)blockIdx.x * HD * HD;
#pragma unroll
    for (int i = 0; i < PC; ++i) {
      const float inv = ks[i] > 0.f ? 1.0f / ks[i] : 0.f;
#pragma unroll
      for (int j = 0; j < PL; ++j) Ab[(c0 + i) * HD + l0 + j] = acc[i][j] * inv;
      if (pl == 0) {
        float* st = kstat + ((int64_t)blockIdx.x * HD + c0 + i) * 2;
        st[0] = len > 0 ? smax[c0 + i] : 0.f;
        st[1] = len > 0 ? ks[i] : 1.f;
      }
    }
  }
}

// LDS column swizzle for [hd][hd+4] context tiles whose rows are read as "row = part*PER + e" by
// the 4 lanes of a token row: rows PER apart are a multiple of 64 banks apart for every legal row
// stride, i.e. a 4-way conflict on each ds_read_b128.  XOR-ing the column with 4*part (a whole
// 16-byte quad) puts the four lanes on four different quads of the bank row.
template <int HD>
__device__ __forceinline__ int swz(int row_group) { return (4 * row_group) & (HD - 1); }

// Stores PER contiguous floats (16-byte vectors when PER allows; callers guarantee alignment).
template <int PER>
__device__ __forceinline__ void store_per(float* __restrict__ p, const float* v) {
  if constexpr (PER % 4 == 0) {
#pragma unroll
    for (int e = 0; e < PER; e += 4)
      *reinterpret_cast<float4*>(p + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
  } else {
#pragma unroll
    for (int e = 0; e < PER; ++e) p[e] = v[e];
  }
}

// Loads a CH x HD tile of `src` (row r0.., columns of head h) into LDS, zero beyond `rows`.
template <int HD>
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int64_t ld, int r0,
                                          int rows, float* __restrict__ dst) {
  constexpr int LDP = HD + 4;
  constexpr int Q = HD / 4;
  for (int idx = threadIdx.x; idx < CH * Q; idx += 256) {
    const int rr = idx / Q, c4 = idx % Q, r = r0 + rr;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) v = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + 4 * c4);
    *reinterpret_cast<float4*>(dst + rr * LDP + 4 * c4) = v;
  }
}

// In-place softmax over the HD channels of each staged row; thread (rl, part) owns HD/4 values.
template <int HD>
__device__ __forceinline__ void row_softmax_tile(float* __restrict__ sQ) {
  constexpr int LDP = HD + 4, PER = HD / 4;
  const int rl = threadIdx.x >> 2, part = threadIdx.x & 3;
  float* p = sQ + rl * LDP + part * PER;
  float x[PER];
  float m = -INFINITY;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    x[e] = p[e];
    m = fmaxf(m, x[e]);
  }
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    x[e] = __expf(x[e] - m);
    s += x[e];
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  const float inv = 1.0f / s;
#pragma unroll
  for (int e = 0; e < PER; ++e) p[e] = x[e] * inv;
}

// Barrier for the chunk-walking kernels: LDS traffic of this wave is done (lgkmcnt), then s_barrier -- but NOT the vmcnt(0) a
// __syncthreads() carries.  These kernels request the next chunk's tiles into registers right after the first barrier of a
// chunk; a __syncthreads() further down the chunk waits for those loads to land (s_memtime stamps of apply_bwd: 7.5K of a
// chunk's 11K cycles went into the first such barrier), so the prefetch hid nothing.  The registers are first read at the top of
// the next chunk, where the compiler places its own counted wait; LDS reuse needs LDS ordering only.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// The same softmax on a tile still in REGISTERS, in the layout the chunk-walking kernels fetch it in: a thread holds 4 consecutive
// channels of NPRE rows, a row's HD channels sit in HD / 4 consecutive lanes (16 = one DPP row for head dim 64).  Row max and row
// sum by DPP (hig_common.h) -- no LDS pass, no barrier.  (s_memtime stamps of apply_bwd at config 2: the LDS form took 7.5K of a
// chunk's 11K cycles: 32 four-way-conflicting scalar LDS accesses per thread and four ds_bpermute round trips.)
template <int HD, int NPRE>
__device__ __forceinline__ void row_softmax_regs(float4 (&v)[NPRE]) {
  static_assert(HD == 64 || HD == 128, "a row in 16 or 32 consecutive lanes");
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    float m = fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w));
    m = row16_max(m);
    if constexpr (HD == 128) m = fmaxf(m, __shfl_xor(m, 16, 64));
    v[i].x = __expf(v[i].x - m); v[i].y = __expf(v[i].y - m); v[i].z = __expf(v[i].z - m); v[i].w = __expf(v[i].w - m);
    float s = (v[i].x + v[i].y) + (v[i].z + v[i].w);
    s = row16_sum(s);
    if constexpr (HD == 128) s += __shfl_xor(s, 16, 64);
    const float inv = 1.0f / s;
    v[i].x *= inv; v[i].y *= inv; v[i].z *= inv; v[i].w *= inv;
  }
}

// ---------------------------------------------------------------------------------------------
// apply: Y[r, h*HD + l] = sum_c softmax_c(Q[r, h*HD + :])[c] * A[b,h][c][l]
// grid = (B*H, ceil(rows / 64))
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ Q, int64_t ldq,
                                                    const float* __restrict__ A,
                                                    float* __restrict__ Y, int64_t ldy, int rows,
                                                    int H) {
  constexpr int LDP = HD + 4, PER = HD / 4;
  __shared__ __attribute__((aligned(16))) float sA[HD * HD];
  __shared__ __attribute__((aligned(16))) float sQ[CH * LDP];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int r0 = blockIdx.y * CH;
  const float* Ab = A + (int64_t)blockIdx.x * HD * HD;
  for (int idx = tid; idx < HD * HD / 4; idx += 256)
    reinterpret_cast<float4*>(sA)[idx] = reinterpret_cast<const float4*>(Ab)[idx];
  load_tile<HD>(Q + (int64_t)b * rows * ldq + h * HD, ldq, r0, rows, sQ);
  __syncthreads();
  row_softmax_tile<HD>(sQ);
  __syncthreads();
  const int rl = tid >> 2, part = tid & 3;
  float acc[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) acc[e] = 0.f;
  const float* qrow = sQ + rl * LDP;
  for (int c = 0; c < HD; ++c) {
    const float qc = qrow[c];
#pragma unroll
    for (int e = 0; e < PER; ++e) acc[e] = fmaf(qc, sA[c * HD + part * PER + e], acc[e]);
  }
  const int r = r0 + rl;
  if (r < rows) {
    store_per<PER>(Y + ((int64_t)b * rows + r) * ldy + h * HD + part * PER, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// hd = 64 / 128 forward kernels on the matrix cores.  The hd x hd contractions are small GEMMs; on
// the VALU they were LDS-read bound (hd = 64: 30 / 42 us per launch at config 2 against ~12 us of HBM
// time; hd = 128: 127 / 110 us at config 5).  v_mfma_f32_32x32x2_f32: 4 waves in a 2 x 2 grid, each
// owning (rows/2) x (cols/2) of the product as 32 x 32 accumulator blocks; operand k order inside an
// 8-group is k = 8*ks + 4*(lane>>5) + j for both operands (as in gemm.hip); the MFMA "row" operand is
// the one indexed by the OUTPUT COLUMN so that a lane ends up with 4 consecutive output columns per
// accumulator quad (16-byte stores): row = base_i + (lane & 31), columns = base_j + 8q + 4*(lane>>5) + e.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

// Activation I/O of the forward MFMA kernels: fp32, or bf16 storage (hig_dims.storage == HIG_STORE_BF16) converted on
// load / store -- the softmax, the context matrices and every accumulation stay fp32.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
  const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
// The chunk prefetch keeps what it loaded RAW (bf16: two dwords) and converts when the chunk is consumed: a conversion next
// to the load is the load's first use -- the compiler waits for the data right there, and the "prefetch" of the next chunk
// became a blocking load in front of the current chunk's products (s_memtime stamps: 6.9K of a chunk's 7K cycles).
typedef unsigned int la_u32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct RawOf;
template <> struct RawOf<float> { typedef float4 type; };
template <> struct RawOf<__bf16> { typedef la_u32x2 type; };
__device__ __forceinline__ float4 ld4raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ la_u32x2 ld4raw(const __bf16* p) { return *reinterpret_cast<const la_u32x2*>(p); }
__device__ __forceinline__ void raw_zero(float4& r) { r = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void raw_zero(la_u32x2& r) { r = la_u32x2{0u, 0u}; }
__device__ __forceinline__ float4 raw_cvt(const float4& r) { return r; }
__device__ __forceinline__ float4 raw_cvt(const la_u32x2& r) {
  return make_float4(__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                     __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(__bf16* p, float4 v) {
  *reinterpret_cast<bf16x4_t*>(p) = bf16x4_t{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int e = 0; e < 16; ++e) a[e] = 0.f;
}
__device__ __forceinline__ f32x16 mfma4(f32x16 acc, float y0, float y1, float y2, float y3, float4 x) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y0, x.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y1, x.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y2, x.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y3, x.w, acc, 0, 0, 0);
  return acc;
}
template <typename T>
__device__ __forceinline__ void store16(T* p, const f32x16& a, float scale = 1.0f) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    st4(p + 8 * q, make_float4(a[4 * q] * scale, a[4 * q + 1] * scale, a[4 * q + 2] * scale, a[4 * q + 3] * scale));
}

// A finished 64-row x HD tile leaves through LDS as WHOLE rows (sT: [64][HD + 4] fp32; HD / VEC lanes of 16 bytes per row,
// 256 / (HD / VEC) rows per pass): the accumulator layout gives a lane 4 consecutive columns of ONE row per quad, which
// written directly is 16 (fp32) or 8 (bf16) bytes into each of 32 rows per store instruction -- every 64-byte sector of the
// output rewritten piecemeal.  Rows at or beyond `rows` are not stored.
template <int HD, typename TIO>
__device__ __forceinline__ void store_tile_rows(const float* sT, TIO* gbase, int64_t ld, int r0, int rows) {
  constexpr int LDP = HD + 4;
  constexpr int VEC = sizeof(TIO) == 2 ? 8 : 4;          // elements per 16-byte store
  constexpr int LPR = HD / VEC, RPP = 256 / LPR;         // lanes per row, rows per pass
  const int c = (threadIdx.x % LPR) * VEC, rr0 = threadIdx.x / LPR;
#pragma unroll
  for (int rr = rr0; rr < CH; rr += RPP) {
    const int r = r0 + rr;
    if (r >= rows) break;
    const float4 a = *reinterpret_cast<const float4*>(sT + rr * LDP + c);
    if constexpr (sizeof(TIO) == 2) {
      const float4 b = *reinterpret_cast<const float4*>(sT + rr * LDP + c + 4);
      typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
      *reinterpret_cast<bf16x8_t*>(gbase + (int64_t)r * ld + c) =
          bf16x8_t{(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
    } else {
      *reinterpret_cast<float4*>(gbase + (int64_t)r * ld + c) = a;
    }
  }
}
template <int HD>
__device__ __forceinline__ void stage16(float* sT, int row, int col, const f32x16& a, float scale = 1.0f) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(sT + row * (HD + 4) + col + 8 * q) =
        make_float4(a[4 * q] * scale, a[4 * q + 1] * scale, a[4 * q + 2] * scale, a[4 * q + 3] * scale);
}

// bf16 PRODUCTS for the bf16-storage backward (TIO = __bf16): the hd x hd contractions of apply_bwd / ctx_bwd run on
// v_mfma_f32_32x32x16_bf16 (1/16 of the fp32 MFMA time -- the fp32 forms of these kernels are bound by the fp32 matrix rate,
// 1.6 GFLOP per launch) with the operands rounded to bf16 while they are read from the fp32 LDS tiles: the rounding the bf16
// storage mode applies to every other matrix operand; accumulation, softmax and the Jacobians stay fp32.  Same accumulator
// layout as v_mfma_f32_32x32x2_f32 (lane (lr, lh): rows 8 q + 4 lh + e, column lr), so everything around the products is shared.
typedef __bf16 la_bf16x8 __attribute__((ext_vector_type(8)));
// 8 consecutive floats of one LDS row -> MFMA operand (k contiguous)
__device__ __forceinline__ la_bf16x8 frag_row8(const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  return la_bf16x8{(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
}
// one column over 8 consecutive LDS rows (stride ld floats) -> MFMA operand (k = row)
__device__ __forceinline__ la_bf16x8 frag_col8(const float* p, int ld) {
  return la_bf16x8{(__bf16)p[0], (__bf16)p[ld], (__bf16)p[2 * ld], (__bf16)p[3 * ld],
                   (__bf16)p[4 * ld], (__bf16)p[5 * ld], (__bf16)p[6 * ld], (__bf16)p[7 * ld]};
}

// Y tile (64 rows) = softmax_c(Q tile) . A[b,h]
template <int HD, typename TIO>
__global__ __launch_bounds__(256) void apply_mfma_kernel(const TIO* __restrict__ Q, int64_t ldq,
                                                         const float* __restrict__ A,
                                                         TIO* __restrict__ Y, int64_t ldy, int rows, int H) {
  constexpr int LDP = HD + 4, TJ = HD / 64, Q4 = HD / 4, NPRE = CH * Q4 / 256;
  __shared__ __attribute__((aligned(16))) float sA[HD * HD];    // [c][l]: reduce index major
  __shared__ __attribute__((aligned(16))) float sQ[CH * LDP];   // [row][c]
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int nchunk = (rows + CH - 1) / CH;
  const TIO* Qb = Q + (int64_t)b * rows * ldq + h * HD;
  // A workgroup walks chunks blockIdx.y, blockIdx.y + gridDim.y, ...: A[b,h] is staged once, and the next Q tile
  // is requested into registers before the softmax / MFMA of the current one (no exposed load per chunk).
  // TWO chunks are in flight (register sets 0 / 1, the chunk loop unrolled by two): with one, a workgroup had 8-16 KB
  // outstanding against a loaded memory latency of several microseconds -- the kernel ran at ~1.3 TB/s on latency alone.
  typedef typename RawOf<TIO>::type raw_t;
  raw_t pre0[NPRE], pre1[NPRE];
  const TIO* qsrc[NPRE];                        // this thread's piece of row (tid + 256 i) / Q4 of chunk 0: a chunk adds CH rows
#pragma unroll
  for (int i = 0; i < NPRE; ++i) qsrc[i] = Qb + (int64_t)((tid + 256 * i) / Q4) * ldq + 4 * ((tid + 256 * i) % Q4);
  const int64_t chunk_q = (int64_t)CH * ldq;
  auto fetch = [&](int chunk, raw_t (&pre)[NPRE]) {
    const int rleft = chunk < nchunk ? rows - chunk * CH : 0;   // rows of this chunk that exist
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      if ((tid + 256 * i) / Q4 < rleft) pre[i] = ld4raw(qsrc[i] + chunk * chunk_q);
      else raw_zero(pre[i]);
    }
  };
  const int gstep = gridDim.y;
  // A[b,h] is requested FIRST (returns are in order: staged into LDS it would otherwise wait behind both chunk prefetches)
  const float* Ab = A + (int64_t)blockIdx.x * HD * HD;
  constexpr int NA4 = HD * HD / 4 / 256;
  float4 areg[NA4];
#pragma unroll
  for (int i = 0; i < NA4; ++i) areg[i] = reinterpret_cast<const float4*>(Ab)[tid + 256 * i];
  fetch(blockIdx.y, pre0);
  fetch(blockIdx.y + gstep, pre1);
#pragma unroll
  for (int i = 0; i < NA4; ++i) reinterpret_cast<float4*>(sA)[tid + 256 * i] = areg[i];
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, lr = lane & 31, lh = lane >> 5;
  const float* qrow = sQ + (wi * 32 + lr) * LDP + 4 * lh;
  const float* acol = sA + (4 * lh) * HD + wj * (HD / 2) + lr;
  // bf16 rows (TIO = __bf16): the product runs on the bf16 matrix cores like every other product of the bf16-storage mode
  // (p and A rounded to bf16 as they are read from LDS); the A operands are chunk-invariant and stay in registers
  [[maybe_unused]] la_bf16x8 afr[TJ][HD / 16];
  if constexpr (sizeof(TIO) == 2) {
    lds_barrier();                               // sA is complete
    const float* ac8 = sA + (8 * lh) * HD + wj * (HD / 2) + lr;
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) afr[tj][ks] = frag_col8(ac8 + 16 * ks * HD + 32 * tj, HD);
  }
  auto do_chunk = [&](int chunk, raw_t (&pre)[NPRE]) {
    const int r0 = chunk * CH;
    float4 qv[NPRE];
#pragma unroll
    for (int i = 0; i < NPRE; ++i) qv[i] = raw_cvt(pre[i]);
    row_softmax_regs<HD, NPRE>(qv);
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<float4*>(sQ + (idx / Q4) * LDP + 4 * (idx % Q4)) = qv[i];
    }
    lds_barrier();
    fetch(chunk + 2 * gstep, pre);               // (this set is free again: two chunks ahead)
    f32x16 acc[TJ];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) zero16(acc[tj]);
    if constexpr (sizeof(TIO) == 2) {
      const float* q8 = sQ + (wi * 32 + lr) * LDP + 8 * lh;
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        const la_bf16x8 pf = frag_row8(q8 + 16 * ks);
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tj][ks], pf, acc[tj], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int ks = 0; ks < HD / 8; ++ks) {
        const float4 q4 = *reinterpret_cast<const float4*>(qrow + 8 * ks);
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
          const float* ap = acol + (8 * ks) * HD + 32 * tj;
          acc[tj] = mfma4(acc[tj], ap[0], ap[HD], ap[2 * HD], ap[3 * HD], q4);
        }
      }
    }
    lds_barrier();   // every wave is done reading sQ: the Y tile is staged there and leaves as whole rows
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) stage16<HD>(sQ, wi * 32 + lr, wj * (HD / 2) + 32 * tj + 4 * lh, acc[tj]);
    lds_barrier();
    store_tile_rows<HD, TIO>(sQ, Y + (int64_t)b * rows * ldy + h * HD, ldy, r0, rows);
    lds_barrier();   // sQ is rewritten by the next chunk
  };
  for (int chunk = blockIdx.y; chunk < nchunk; chunk += 2 * gstep) {
    do_chunk(chunk, pre0);
    if (chunk + gstep < nchunk) do_chunk(chunk + gstep, pre1);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// apply, exact fp32, head dim 64, WAVE-AUTONOMOUS (round 6): Y tile (16 rows) = softmax_c(Q tile) . A[b,h] per wave, no LDS and no
// barrier in the loop.  The chunk-walking kernel above is insensitive to everything one would tune -- LDS traffic, the padded last
// chunk, the number of workgroups, the row stride (profiles/r06_notes.md section 5): its 1 024 workgroups move through
// load / softmax / barrier / multiply / barrier / stage / barrier / store in step, and each phase leaves the other units idle.
// Here a wave owns 16-row tiles from the load to the store, so the waves of a SIMD drift apart and one's MFMAs cover
// another's loads, exponentials and stores:
//   * v_mfma_f32_16x16x4_f32 with the roles chosen so that NOTHING needs a transpose: the A operand's row index is the OUTPUT
//     column l, the B operand's column index the tile ROW r, and the reduce index of k-step s, lane group kq is channel
//     c = 16 kq + s.  Lane (r, kq) therefore needs p[r][16 kq .. 16 kq + 15] -- 64 contiguous bytes of its row, loaded as four
//     float4 straight into the registers the MFMAs read -- and the row softmax is a reduction over the lane's 16 values and the
//     4 lane groups (two cross-lane exchanges).
//   * output row index i = 4 q + e of column block blk stands for column l = 16 q + 4 blk + e: lane (r, q) ends up with the 16
//     CONSECUTIVE columns 16 q .. 16 q + 15 of row r in its four accumulators -- four 16-byte stores, a row's 256 bytes from
//     its four lanes.
//     its four lanes (written through 4 KB of wave-private LDS as whole rows all the same: the stores of one instruction would
//     otherwise be 64 pieces of 16 bytes).
//   * A[b,h] (16 KB) is staged once per workgroup through LDS (in operand order) into 64 registers per lane (the only barrier); four
//     independent accumulator chains of 16 MFMAs per tile; the next tile's Q is requested before the current one's softmax.
// T = 196: 13 tiles of 16 rows (6 % padding) instead of 4 chunks of 64 (31 %).  Measured (tools/attn_time.py, B = 64): 21.2 ->
// 17.0 us warm -- a fifth, not the factor of two the pipes' budgets (6.3 us of MFMA, 10 us of
// HBM) promise: the launch is short enough that every serial microsecond of its start shows (operand staging, above); prefetching all of a wave's tiles, one wave per tile with the operands in LDS (16 waves per workgroup) and
// two workgroups per (sample, head) all measured the same or worse (profiles/r06_notes.md section 5).
// ---------------------------------------------------------------------------------------------------------------------
typedef float la_f32x4 __attribute__((ext_vector_type(4)));
typedef int la_i32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void apply_wave64_kernel(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ A,
                                                           float* __restrict__ Y, int64_t ldy, int rows, int H) {
  constexpr int HD = 64;
  __shared__ __attribute__((aligned(16))) float sA[HD * HD];    // A[b,h] in MFMA-operand order (below)
  __shared__ __attribute__((aligned(16))) float sO[4 * 16 * HD]; // the waves' output staging
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int r = lane & 15, kq = lane >> 4;
  const int ntile = (rows + 15) >> 4;
  const float* qb = Q + (int64_t)b * rows * ldq + h * HD + 16 * kq;
  // The prefetch is written as loads hipcc does not see and a counted wait that names the loaded registers: as C++ loads it sank
  // them to their use (register pressure), copied them early (a use: vmcnt(0) in the middle of the MFMAs) or -- with any of the
  // loads / stores under an `if` -- lost count and waited vmcnt(0) for the write acknowledgements of the tile before as well;
  // each of the three left a tile waiting a full memory latency (found in the disassembly).  Vector-memory operations of a wave
  // retire in issue order (the counted waits of gemm_wsp32.hip rest on the same): behind a tile's four loads come the four
  // stores of the tile in front of it and the four loads of the tile after it.
  la_f32x4 q0[4], q1[4];
  auto fetch = [&](int tile, la_f32x4 (&v)[4]) {
    const int row = min(tile * 16 + r, rows - 1);              // (rows past the end: a copy of the last row, never stored)
    const float* p = qb + (int64_t)row * ldq;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p) : "memory");
  };
  fetch(min(wave, ntile - 1), q0);                             // (in flight while A is staged: the A loads behind it are C++ loads,
                                                               //  waited for by hipcc before their LDS stores -- in order, so q0 has landed too)
  {
    // staged in MFMA-operand order: float4 (c, r) = A[c][16 (r >> 2) + 4 blk + (r & 3)], blk = 0 .. 3 -- the four A operands of a
    // k-step in ONE conflict-free ds_read_b128 per lane (read as 64 scalars from the row-major image, the eight waves of a CU's
    // two workgroups spent ~2 us of its LDS pipe on four-way bank conflicts before their first MFMA: 19.0 -> 17.5 us)
    const float4* Ab = reinterpret_cast<const float4*>(A + (int64_t)blockIdx.x * HD * HD);
    float4 areg[HD * HD / 4 / 256];
#pragma unroll
    for (int i = 0; i < HD * HD / 4 / 256; ++i) areg[i] = Ab[tid + 256 * i];
#pragma unroll
    for (int i = 0; i < HD * HD / 4 / 256; ++i) {
      const int f = tid + 256 * i, c = f >> 4, l = 4 * (f & 15);   // channel c, columns l .. l + 3: row index 4 (l >> 4) + e, block (l >> 2) & 3
      float* dst = sA + ((c * 16) + 4 * (l >> 4)) * 4 + ((l >> 2) & 3);
      // element e of the float4 goes to dst[4 e]; the four channels c of a wave write DIFFERENT elements in the same instruction
      // (e = (j + c) & 3), so one scalar-store instruction touches all 32 banks instead of 8
      const float el[4] = {areg[i].x, areg[i].y, areg[i].z, areg[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = (j + c) & 3;
        dst[4 * e] = e == 0 ? el[0] : e == 1 ? el[1] : e == 2 ? el[2] : el[3];
      }
    }
  }
  __syncthreads();
  float aop[16][4];                                            // [k-step s][column block]: A[16 kq + s][16 (r >> 2) + 4 blk + (r & 3)]
#pragma unroll
  for (int s2 = 0; s2 < 16; ++s2) {
    const float4 a4 = reinterpret_cast<const float4*>(sA)[(16 * kq + s2) * 16 + r];
    aop[s2][0] = a4.x; aop[s2][1] = a4.y; aop[s2][2] = a4.z; aop[s2][3] = a4.w;
  }
  float* const so = sO + wave * (16 * HD);                     // (its own 4 KB per wave: no second barrier before the first tile; 17.5 -> 17.0 us)
  __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(Y + (int64_t)b * rows * ldy, 0, (int)(((int64_t)(rows - 1) * ldy + H * HD) * 4), 0x00020000);
  auto tile_out = [&](int tile, la_f32x4 (&cur)[4], auto younger) {
    constexpr int YOUNGER = decltype(younger)::value;          // vector-memory operations issued behind this tile's loads
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) : "n"(YOUNGER) : "memory");
    // softmax over the row's 64 channels: 16 in this lane, the other 48 in lanes r + 16, r + 32, r + 48
    float m = fmaxf(fmaxf(fmaxf(cur[0].x, cur[0].y), fmaxf(cur[0].z, cur[0].w)), fmaxf(fmaxf(cur[1].x, cur[1].y), fmaxf(cur[1].z, cur[1].w)));
    m = fmaxf(m, fmaxf(fmaxf(fmaxf(cur[2].x, cur[2].y), fmaxf(cur[2].z, cur[2].w)), fmaxf(fmaxf(cur[3].x, cur[3].y), fmaxf(cur[3].z, cur[3].w))));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float p[16];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      p[4 * i] = __expf(cur[i].x - m); p[4 * i + 1] = __expf(cur[i].y - m); p[4 * i + 2] = __expf(cur[i].z - m); p[4 * i + 3] = __expf(cur[i].w - m);
      sum += (p[4 * i] + p[4 * i + 1]) + (p[4 * i + 2] + p[4 * i + 3]);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    la_f32x4 acc[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) acc[blk] = la_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const float ps = p[s2] * inv;
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[s2][blk], ps, acc[blk], 0, 0, 0);
    }
    // out through this wave's 4 KB of LDS as WHOLE rows (straight from the accumulators a store instruction is 64 pieces of 16
    // bytes, 64 bytes apart: four L2 write requests per 64-byte segment through the write-through L1)
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
      *reinterpret_cast<float4*>(so + r * HD + ((16 * kq + 4 * blk) ^ (4 * r))) = make_float4(acc[blk][0], acc[blk][1], acc[blk][2], acc[blk][3]);   // (16-byte chunk q of row r at q ^ r: conflict-free both ways)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (a wave reads back only what it wrote itself)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = 4 * i + kq, cc = 4 * r;                     // lane (r, kq): row 4 i + kq of the tile, columns 4 r .. 4 r + 3
      const float4 v = *reinterpret_cast<const float4*>(so + rr * HD + (cc ^ (4 * rr)));
      // UNCONDITIONAL stores through a buffer descriptor that ends with the sample's last row (rows past it are dropped by the
      // range check), and unconditional (clamped) prefetches below: with either inside an `if`, hipcc cannot count the
      // outstanding vector-memory operations and waits vmcnt(0) -- for the write acknowledgements of this tile AND the rows of
      // the tile after next -- in the middle of the next tile's MFMAs (found in the disassembly: it made the prefetch worthless)
      const int row = tile * 16 + rr;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(la_i32x4, v), rsY, (row * (int)ldy + h * HD + cc) * 4, 0, 0);
#endif
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // two register sets alternate (no copies: a copy of the prefetched rows is a use, and hipcc schedules it -- and its wait -- early)
  using I4 = std::integral_constant<int, 4>;
  using I8 = std::integral_constant<int, 8>;
  if (wave >= ntile) return;                                   // (never with >= 64 rows)
  fetch(min(wave + 4, ntile - 1), q1);
  tile_out(wave, q0, I4{});                                     // (behind q0: only the loads of q1 -- the A loads were waited for above)
  for (int tile = wave + 4; tile < ntile; tile += 8) {
    fetch(min(tile + 4, ntile - 1), q0);
    tile_out(tile, q1, I8{});
    if (tile + 4 >= ntile) break;
    fetch(min(tile + 8, ntile - 1), q1);
    tile_out(tile + 4, q0, I8{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (a clamped, unused prefetch may still be in flight)
}

struct CtxGroups {
  int Hgrp;            // heads per output group
  int64_t a_gs, k_gs;  // group strides of A / kstat in floats
};
// A[b,h][c][l] = sum_r softmax_r(K)[r,c] V[r,l]  (+ kstat): one workgroup per (sample, head) walks the row chunks
// ONCE with a running column max (online softmax): when a chunk raises the max of channel c, the accumulator row c
// and the running sum are rescaled by exp(m_old - m_new).  The next K / V tiles are prefetched into registers.
template <int HD, typename TIO>
__global__ __launch_bounds__(256) void ctx_mfma_kernel(const TIO* __restrict__ K, const TIO* __restrict__ V,
                                                       int64_t ld, int rows, int H,
                                                       const int64_t* __restrict__ length, float* __restrict__ A,
                                                       float* __restrict__ kstat, __bf16* __restrict__ At16, const CtxGroups grp) {
  constexpr int LDP = HD + 4, TB = HD / 64, Q4 = HD / 4, NRG = 256 / Q4, PER = CH / NRG;
  __shared__ __attribute__((aligned(16))) float sP[CH * LDP];   // [r][c] = exp(K - running max)
  __shared__ __attribute__((aligned(16))) float sV[CH * LDP];   // [r][l]
  __shared__ __attribute__((aligned(16))) float sred[NRG * HD]; // per row-group column maxima, at the end column sums
  __shared__ __attribute__((aligned(16))) float smax[HD];       // running column max
  __shared__ __attribute__((aligned(16))) float sscale[HD];     // exp(m_old - m_new) of the current chunk
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  // outputs: head h belongs to group h / Hgrp (the layers of the batched text side); a group's context matrices and column
  // statistics are laid out (B, Hgrp, ...) at group strides -- Hgrp = H: one group, the plain (B, H, ...) layout
  const int64_t oidx = (int64_t)b * grp.Hgrp + h % grp.Hgrp;
  A += (h / grp.Hgrp) * grp.a_gs;
  kstat += (h / grp.Hgrp) * grp.k_gs;
  int len = rows;
  if (length) len = (int)min<int64_t>(max<int64_t>(length[b], 0), rows);
  const TIO* Kb = K + (int64_t)b * rows * ld + h * HD;
  const TIO* Vb = V + (int64_t)b * rows * ld + h * HD;
  const int c4 = tid % Q4, rgrp = tid / Q4;
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, lr = lane & 31, lh = lane >> 5;
  float4 kreg[PER], vreg[PER];
  auto fetch = [&](int r0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int r = r0 + rgrp + NRG * i;
      if (r < len) {
        kreg[i] = ld4(Kb + (int64_t)r * ld + 4 * c4);
        vreg[i] = ld4(Vb + (int64_t)r * ld + 4 * c4);
      } else {
        kreg[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        vreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  fetch(0);
  if (tid < HD) smax[tid] = -INFINITY;
  f32x16 acc[TB][TB];
#pragma unroll
  for (int ti = 0; ti < TB; ++ti)
#pragma unroll
    for (int tj = 0; tj < TB; ++tj) zero16(acc[ti][tj]);
  float4 ks4 = make_float4(0.f, 0.f, 0.f, 0.f);   // this thread's share of sum_r exp(K[r][c] - m[c]) for its 4 channels
  const float* pcol = sP + (4 * lh) * LDP + wi * (HD / 2) + lr;
  const float* vcol = sV + (4 * lh) * LDP + wj * (HD / 2) + lr;
  for (int r0 = 0; r0 < len; r0 += CH) {
    float4 m4 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      m4.x = fmaxf(m4.x, kreg[i].x); m4.y = fmaxf(m4.y, kreg[i].y);
      m4.z = fmaxf(m4.z, kreg[i].z); m4.w = fmaxf(m4.w, kreg[i].w);
    }
    *reinterpret_cast<float4*>(sred + rgrp * HD + 4 * c4) = m4;
    lds_barrier();
    if (tid < HD) {
      float m = smax[tid];
      const float mo = m;
      for (int g2 = 0; g2 < NRG; ++g2) m = fmaxf(m, sred[g2 * HD + tid]);
      sscale[tid] = mo == -INFINITY ? 0.f : __expf(mo - m);   // (nothing accumulated yet while mo == -inf)
      smax[tid] = m;
    }
    lds_barrier();
    const float4 cm = *reinterpret_cast<const float4*>(smax + 4 * c4);
    const float4 sc = *reinterpret_cast<const float4*>(sscale + 4 * c4);
    ks4.x *= sc.x; ks4.y *= sc.y; ks4.z *= sc.z; ks4.w *= sc.w;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int rr = rgrp + NRG * i;
      float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + rr < len)
        pe = make_float4(__expf(kreg[i].x - cm.x), __expf(kreg[i].y - cm.y), __expf(kreg[i].z - cm.z), __expf(kreg[i].w - cm.w));
      ks4.x += pe.x; ks4.y += pe.y; ks4.z += pe.z; ks4.w += pe.w;
      *reinterpret_cast<float4*>(sP + rr * LDP + 4 * c4) = pe;
      *reinterpret_cast<float4*>(sV + rr * LDP + 4 * c4) = vreg[i];
    }
    // rescale the accumulator rows (channel c = row of A) before this chunk is added
#pragma unroll
    for (int ti = 0; ti < TB; ++ti) {
      const float f = sscale[wi * (HD / 2) + 32 * ti + lr];
#pragma unroll
      for (int tj = 0; tj < TB; ++tj)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ti][tj][e] *= f;
    }
    lds_barrier();
    if (r0 + CH < len) fetch(r0 + CH);
    const int nk = (min(CH, len - r0) + 7) / 8;   // 8-row groups holding valid rows (the rest are zeros)
    for (int g = 0; g < nk; ++g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pv[TB], vv[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          pv[t] = pcol[(8 * g + j) * LDP + 32 * t];
          vv[t] = vcol[(8 * g + j) * LDP + 32 * t];
        }
#pragma unroll
        for (int ti = 0; ti < TB; ++ti)
#pragma unroll
          for (int tj = 0; tj < TB; ++tj)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[tj], pv[ti], acc[ti][tj], 0, 0, 0);
      }
    }
    lds_barrier();   // sP / sV / sred / sscale are rewritten by the next chunk
  }
  *reinterpret_cast<float4*>(sred + rgrp * HD + 4 * c4) = ks4;
  lds_barrier();
  if (tid < HD) {
    float t = 0.f;
    for (int g2 = 0; g2 < NRG; ++g2) t += sred[g2 * HD + tid];
    sscale[tid] = t > 0.f ? 1.0f / t : 0.f;
    float* st = kstat + (oidx * HD + tid) * 2;
    st[0] = len > 0 ? smax[tid] : 0.f;
    st[1] = len > 0 ? t : 1.f;
  }
  lds_barrier();
#pragma unroll
  for (int ti = 0; ti < TB; ++ti) {
    const int cc = wi * (HD / 2) + 32 * ti + lr;      // lane's context row (channel c)
    const float inv = sscale[cc];
    float* ap = A + oidx * HD * HD + cc * HD + wj * (HD / 2) + 4 * lh;
#pragma unroll
    for (int tj = 0; tj < TB; ++tj) store16(ap + 32 * tj, acc[ti][tj], inv);
    if (At16) {   // the same matrix transposed and rounded, At[l][c] = bf16(A[c][l]), in the fragment-major order of
                  // hig_at16_offset: the MFMA row operand of linattn16.hip
#pragma unroll
      for (int tj = 0; tj < TB; ++tj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int l = wj * (HD / 2) + 32 * tj + 8 * (e >> 2) + 4 * lh + (e & 3);
          At16[oidx * HD * HD + hig_at16_offset(HD, l, cc)] = (__bf16)(acc[ti][tj][e] * inv);
        }
    }
  }
}

// ctx over row chunks in parallel (online softmax): workgroup (b,h,chunk j) uses the column max m_j of ITS
// 64 rows and writes the unnormalised partial  A_j[c][l] = sum_r exp(K[r][c] - m_j[c]) V[r][l],  m_j and
// s_j[c] = sum_r exp(K[r][c] - m_j[c]);  ctx_combine_kernel merges them:  m = max_j m_j,
// A = sum_j exp(m_j - m) A_j / sum_j exp(m_j - m) s_j.   part layout per (bh, chunk): [HD*HD | m HD | s HD].
template <int HD, typename TIO>
__global__ __launch_bounds__(256) void ctx_part_mfma_kernel(const TIO* __restrict__ K, const TIO* __restrict__ V,
                                                            int64_t ld, int rows, int H,
                                                            const int64_t* __restrict__ length,
                                                            float* __restrict__ part) {
  constexpr int LDP = HD + 4, TB = HD / 64, Q4 = HD / 4, NRG = 256 / Q4, PER = CH / NRG;
  __shared__ __attribute__((aligned(16))) float sP[CH * LDP];
  __shared__ __attribute__((aligned(16))) float sV[CH * LDP];
  __shared__ __attribute__((aligned(16))) float sred[NRG * HD];   // per row-group column maxima, then column sums
  __shared__ float smax[HD];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  int len = rows;
  if (length) len = (int)min<int64_t>(max<int64_t>(length[b], 0), rows);
  const int r0 = blockIdx.y * CH;
  const int nvalid = max(0, min(CH, len - r0));
  const TIO* Kb = K + ((int64_t)b * rows + r0) * ld + h * HD;
  const TIO* Vb = V + ((int64_t)b * rows + r0) * ld + h * HD;
  // one pass over HBM, 16 bytes per lane: thread = 4 channels (c4) of rows rgrp, rgrp + NRG, ...
  const int c4 = tid % Q4, rgrp = tid / Q4;
  float4 kreg[PER], vreg[PER];
  float4 m4 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int rr = rgrp + NRG * i;
    if (rr < nvalid) {
      kreg[i] = ld4(Kb + (int64_t)rr * ld + 4 * c4);
      vreg[i] = ld4(Vb + (int64_t)rr * ld + 4 * c4);
    } else {
      kreg[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      vreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    m4.x = fmaxf(m4.x, kreg[i].x); m4.y = fmaxf(m4.y, kreg[i].y);
    m4.z = fmaxf(m4.z, kreg[i].z); m4.w = fmaxf(m4.w, kreg[i].w);
  }
  *reinterpret_cast<float4*>(sred + rgrp * HD + 4 * c4) = m4;
  __syncthreads();
  if (tid < HD) {
    float m = sred[tid];
    for (int g2 = 1; g2 < NRG; ++g2) m = fmaxf(m, sred[g2 * HD + tid]);
    smax[tid] = m;
  }
  __syncthreads();
  const float4 cm = *reinterpret_cast<const float4*>(smax + 4 * c4);
  float4 ks4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int rr = rgrp + NRG * i;
    float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rr < nvalid)
      pe = make_float4(__expf(kreg[i].x - cm.x), __expf(kreg[i].y - cm.y), __expf(kreg[i].z - cm.z), __expf(kreg[i].w - cm.w));
    ks4.x += pe.x; ks4.y += pe.y; ks4.z += pe.z; ks4.w += pe.w;
    *reinterpret_cast<float4*>(sP + rr * LDP + 4 * c4) = pe;
    *reinterpret_cast<float4*>(sV + rr * LDP + 4 * c4) = vreg[i];
  }
  *reinterpret_cast<float4*>(sred + rgrp * HD + 4 * c4) = ks4;   // (the maxima were consumed before the barrier above)
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, lr = lane & 31, lh = lane >> 5;
  f32x16 acc[TB][TB];
#pragma unroll
  for (int ti = 0; ti < TB; ++ti)
#pragma unroll
    for (int tj = 0; tj < TB; ++tj) zero16(acc[ti][tj]);
  const float* pcol = sP + (4 * lh) * LDP + wi * (HD / 2) + lr;
  const float* vcol = sV + (4 * lh) * LDP + wj * (HD / 2) + lr;
  const int nk = (nvalid + 7) / 8;
  for (int g = 0; g < nk; ++g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float pv[TB], vv[TB];
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        pv[t] = pcol[(8 * g + j) * LDP + 32 * t];
        vv[t] = vcol[(8 * g + j) * LDP + 32 * t];
      }
#pragma unroll
      for (int ti = 0; ti < TB; ++ti)
#pragma unroll
        for (int tj = 0; tj < TB; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[tj], pv[ti], acc[ti][tj], 0, 0, 0);
    }
  }
  float* pb = part + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (HD * HD + 2 * HD);
#pragma unroll
  for (int ti = 0; ti < TB; ++ti)
#pragma unroll
    for (int tj = 0; tj < TB; ++tj)
      store16(pb + (wi * (HD / 2) + 32 * ti + lr) * HD + wj * (HD / 2) + 32 * tj + 4 * lh, acc[ti][tj]);
  if (tid < HD) {
    float t = 0.f;
    for (int g2 = 0; g2 < NRG; ++g2) t += sred[g2 * HD + tid];
    pb[HD * HD + tid] = smax[tid];
    pb[HD * HD + HD + tid] = t;
  }
}

template <int HD>
__global__ __launch_bounds__(256) void ctx_combine_kernel(const float* __restrict__ part, int nchunk,
                                                          float* __restrict__ A, float* __restrict__ kstat,
                                                          __bf16* __restrict__ At16) {
  constexpr int PS = HD * HD + 2 * HD;
  __shared__ float sw[16][HD];        // weight of chunk j for channel c (nchunk <= 16 per pass)
  __shared__ float sinv[HD];
  const int tid = threadIdx.x;
  const float* pb = part + (int64_t)blockIdx.x * nchunk * PS;
  float* Ab = A + (int64_t)blockIdx.x * HD * HD;
  float gm = -INFINITY, gs = 0.f;
  if (tid < HD) {
    for (int j = 0; j < nchunk; ++j) gm = fmaxf(gm, pb[(int64_t)j * PS + HD * HD + tid]);
    for (int j = 0; j < nchunk; ++j) {
      const float mj = pb[(int64_t)j * PS + HD * HD + tid];
      const float wj = mj == -INFINITY ? 0.f : __expf(mj - gm);
      gs += wj * pb[(int64_t)j * PS + HD * HD + HD + tid];
    }
    sinv[tid] = gs > 0.f ? 1.0f / gs : 0.f;
    float* st = kstat + ((int64_t)blockIdx.x * HD + tid) * 2;
    st[0] = gs > 0.f ? gm : 0.f;
    st[1] = gs > 0.f ? gs : 1.f;
  }
  for (int j0 = 0; j0 < nchunk; j0 += 16) {
    const int nj = min(16, nchunk - j0);
    __syncthreads();
    if (tid < HD)
      for (int j = 0; j < nj; ++j) {
        const float mj = pb[(int64_t)(j0 + j) * PS + HD * HD + tid];
        sw[j][tid] = mj == -INFINITY ? 0.f : __expf(mj - gm);
      }
    __syncthreads();
    for (int idx = tid; idx < HD * HD / 4; idx += 256) {
      const int c = (idx * 4) / HD;
      float4 a = j0 == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(Ab)[idx];
      for (int j = 0; j < nj; ++j) {
        const float4 t = reinterpret_cast<const float4*>(pb + (int64_t)(j0 + j) * PS)[idx];
        const float wv = sw[j][c];
        a.x += wv * t.x; a.y += wv * t.y; a.z += wv * t.z; a.w += wv * t.w;
      }
      if (j0 + nj >= nchunk) {
        const float inv = sinv[c];
        a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
        if (At16) {
          const int l0 = (idx * 4) % HD;
          __bf16* at = At16 + (int64_t)blockIdx.x * HD * HD;
          at[hig_at16_offset(HD, l0, c)] = (__bf16)a.x; at[hig_at16_offset(HD, l0 + 1, c)] = (__bf16)a.y;
          at[hig_at16_offset(HD, l0 + 2, c)] = (__bf16)a.z; at[hig_at16_offset(HD, l0 + 3, c)] = (__bf16)a.w;
        }
      }
      reinterpret_cast<float4*>(Ab)[idx] = a;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// apply_bwd: dq = dY A^T, dQ = q * (dq - sum_c q dq);  dA[c][l] = sum_r q[r,c] dY[r,l]
// grid = (B*H, row chunks): each block owns 64 rows and writes its dA contribution to
// dApart[(bh * nchunk + chunk)]; chunk_sum_kernel adds the chunks in a fixed order (no atomics:
// bitwise reproducible)
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void apply_bwd_kernel(const float* __restrict__ dY, int64_t lddy,
                                                        const float* __restrict__ Q, int64_t ldq,
                                                        const float* __restrict__ A,
                                                        float* __restrict__ dQ, int64_t lddq,
                                                        float* __restrict__ dApart, int rows, int H) {
  constexpr int LDP = HD + 4, PER = HD / 4;
  constexpr int PC = Patch<HD>::PC, PL = Patch<HD>::PL, TL = Patch<HD>::TL;
  __shared__ __attribute__((aligned(16))) float sA[HD * LDP];  // padded: read by rows of c
  __shared__ __attribute__((aligned(16))) float sQ[CH * LDP];
  __shared__ __attribute__((aligned(16))) float sD[CH * LDP];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const float* Ab = A + (int64_t)blockIdx.x * HD * HD;
  for (int idx = tid; idx < HD * HD; idx += 256) {
    const int c = idx / HD;
    sA[c * LDP + ((idx % HD) ^ swz<HD>(c / PER))] = Ab[idx];
  }
  const int rl = tid >> 2, part = tid & 3;
  const int pc = tid / TL, pl = tid % TL;
  const int c0 = pc * PC, l0 = pl * PL;
  const bool active = c0 < HD;
  float acc[PC][PL];
#pragma unroll
  for (int i = 0; i < PC; ++i)
#pragma unroll
    for (int j = 0; j < PL; ++j) acc[i][j] = 0.f;

  {
    const int r0 = blockIdx.y * CH;
    __syncthreads();
    load_tile<HD>(Q + (int64_t)b * rows * ldq + h * HD, ldq, r0, rows, sQ);
    load_tile<HD>(dY + (int64_t)b * rows * lddy + h * HD, lddy, r0, rows, sD);
    __syncthreads();
    row_softmax_tile<HD>(sQ);
    __syncthreads();
    {  // dq for (row rl, channels part*PER ..): dot of the dY row with rows of A
      const float* dr = sD + rl * LDP;
      float dq[PER];
#pragma unroll
      for (int e = 0; e < PER; ++e) dq[e] = 0.f;
      const int sw = swz<HD>(part);
      for (int l = 0; l < HD; l += 4) {
        const float4 d4 = *reinterpret_cast<const float4*>(dr + l);  // logical columns l..l+3
#pragma unroll
        for (int e = 0; e < PER; ++e) {
          const float4 a4 = *reinterpret_cast<const float4*>(sA + (part * PER + e) * LDP + (l ^ sw));
          dq[e] += d4.x * a4.x + d4.y * a4.y + d4.z * a4.z + d4.w * a4.w;
        }
      }
      const float* qr = sQ + rl * LDP + part * PER;
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < PER; ++e) s += qr[e] * dq[e];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      const int r = r0 + rl;
      if (r < rows) {
        float o[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) o[e] = qr[e] * (dq[e] - s);
        store_per<PER>(dQ + ((int64_t)b * rows + r) * lddq + h * HD + part * PER, o);
      }
    }
    if (active) {  // dA patch: rows beyond `rows` are zero in sD
      for (int rr = 0; rr < CH; ++rr) {
        float p[PC], v[PL];
#pragma unroll
        for (int i = 0; i < PC; ++i) p[i] = sQ[rr * LDP + c0 + i];
#pragma unroll
        for (int j = 0; j < PL; ++j) v[j] = sD[rr * LDP + l0 + j];
#pragma unroll
        for (int i = 0; i < PC; ++i)
#pragma unroll
          for (int j = 0; j < PL; ++j) acc[i][j] = fmaf(p[i], v[j], acc[i][j]);
      }
    }
  }
  if (active) {
    float* dAb = dApart + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * HD * HD;
#pragma unroll
    for (int i = 0; i < PC; ++i)
#pragma unroll
      for (int j = 0; j < PL; ++j) dAb[(c0 + i) * HD + l0 + j] = acc[i][j];
  }
}

// out[g][e] = sum_{c < nchunk} part[(g * nchunk + c)][e]   (e < n, n % 4 == 0)
__global__ void chunk_sum_kernel(const float* __restrict__ part, int nchunk, int64_t n, int64_t groups,
                                 float* __restrict__ out) {
  const int64_t total4 = groups * n / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = (i * 4) / n, e = (i * 4) % n;
    const float* p = part + (g * nchunk) * n + e;
    float4 s = *reinterpret_cast<const float4*>(p);
    for (int c = 1; c < nchunk; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)c * n);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *reinterpret_cast<float4*>(out + g * n + e) = s;
  }
}

// ---------------------------------------------------------------------------------------------
// ctx_bwd: with k = exp(K - max)/sum on valid rows:
//   dV[r,l] = sum_c k[r,c] dA[c][l];  dk[r,c] = sum_l V[r,l] dA[c][l];
//   dK[r,c] = k[r,c] (dk[r,c] - sum_r' k[r',c] dk[r',c]);  masked rows get 0.
// Two launches, both grid = (B*H, row chunks): pass 1 writes dV, parks raw dk in dK and writes
// the chunk's column sums of k*dk; pass 2 adds the chunks' column sums and finishes dK in place.
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void ctx_bwd_kernel(const float* __restrict__ dA,
                                                      const float* __restrict__ K,
                                                      const float* __restrict__ V, int64_t ld,
                                                      const float* __restrict__ kstat,
                                                      const int64_t* __restrict__ length,
                                                      float* __restrict__ dK, float* __restrict__ dV,
                                                      int64_t ldd, int rows, int H,
                                                      float* __restrict__ colpart) {
  constexpr int LDP = HD + 4, PER = HD / 4;
  __shared__ __attribute__((aligned(16))) float sdA[HD * LDP];   // [c][l]
  __shared__ __attribute__((aligned(16))) float sK[CH * LDP];    // k (normalised)
  __shared__ __attribute__((aligned(16))) float sV[CH * LDP];
  __shared__ float smax[HD], sinv[HD];
  __shared__ float swsum[4][HD];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  int len = rows;
  if (length) len = (int)min<int64_t>(max<int64_t>(length[b], 0), rows);
  const float* dAb = dA + (int64_t)blockIdx.x * HD * HD;
  for (int idx = tid; idx < HD * HD; idx += 256) {
    const int c = idx / HD;
    sdA[c * LDP + ((idx % HD) ^ swz<HD>(c / PER))] = dAb[idx];  // column-swizzled, see swz()
  }
  if (tid < HD) {
    const float* st = kstat + ((int64_t)blockIdx.x * HD + tid) * 2;
    smax[tid] = st[0];
    sinv[tid] = 1.0f / st[1];
  }
  const float* Kb = K + (int64_t)b * rows * ld + h * HD;
  const float* Vb = V + (int64_t)b * rows * ld + h * HD;
  float* dKb = dK + (int64_t)b * rows * ldd + h * HD;
  float* dVb = dV + (int64_t)b * rows * ldd + h * HD;
  const int rl = tid >> 2, part = tid & 3;
  float colsum[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) colsum[e] = 0.f;
  {
    const int r0 = blockIdx.y * CH;
    __syncthreads();
    for (int idx = tid; idx < CH * HD; idx += 256) {
      const int rr = idx / HD, cc = idx % HD, r = r0 + rr;
      float kk = 0.f, v = 0.f;
      if (r < len) {
        kk = __expf(Kb[(int64_t)r * ld + cc] - smax[cc]) * sinv[cc];
        v = Vb[(int64_t)r * ld + cc];
      }
      sK[rr * LDP + cc] = kk;
      sV[rr * LDP + cc] = v;
    }
    __syncthreads();
    const int r = r0 + rl;
    float dk[PER], dv[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) dk[e] = dv[e] = 0.f;
    const float* kr = sK + rl * LDP;
    const float* vr = sV + rl * LDP;
    const int sw = swz<HD>(part);
    for (int x = 0; x < HD; x += 4) {  // dk[c=mine] = sum_l V[r,l] dA[c][l]
      const float4 v4 = *reinterpret_cast<const float4*>(vr + x);  // logical columns x..x+3
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        const float4 a4 = *reinterpret_cast<const float4*>(sdA + (part * PER + e) * LDP + (x ^ sw));
        dk[e] += v4.x * a4.x + v4.y * a4.y + v4.z * a4.z + v4.w * a4.w;
      }
    }
    for (int c = 0; c < HD; ++c) {  // dv[l=mine] = sum_c k[r,c] dA[c][l]
      const float kc = kr[c];
      const float* arow = sdA + c * LDP;
      const int swc = swz<HD>(c / PER);
      if constexpr (PER % 4 == 0) {
#pragma unroll
        for (int e = 0; e < PER; e += 4) {
          const float4 a4 = *reinterpret_cast<const float4*>(arow + ((part * PER + e) ^ swc));
          dv[e] = fmaf(kc, a4.x, dv[e]); dv[e + 1] = fmaf(kc, a4.y, dv[e + 1]);
          dv[e + 2] = fmaf(kc, a4.z, dv[e + 2]); dv[e + 3] = fmaf(kc, a4.w, dv[e + 3]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < PER; ++e) dv[e] = fmaf(kc, arow[(part * PER + e) ^ swc], dv[e]);
      }
    }
    if (r < rows) {
      const bool valid = r < len;
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        colsum[e] += kr[part * PER + e] * dk[e];  // k == 0 on masked rows
        if (!valid) dk[e] = dv[e] = 0.f;
      }
      store_per<PER>(dKb + (int64_t)r * ldd + part * PER, dk);
      store_per<PER>(dVb + (int64_t)r * ldd + part * PER, dv);
    }
  }
  // column sums of this chunk: lanes with equal `part` hold different rows -> butterfly over lane bits 2..5
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    float s = colsum[e];
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if ((tid & 63) < 4) swsum[tid >> 6][part * PER + e] = s;
  }
  __syncthreads();
  if (tid < HD)
    colpart[((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * HD + tid] =
        (swsum[0][tid] + swsum[1][tid]) + (swsum[2][tid] + swsum[3][tid]);
}

// pass 2: dK[r,c] = k[r,c] * (dk[r,c] - sum_chunks colpart[chunk][c]) on valid rows
template <int HD>
__global__ __launch_bounds__(256) void ctx_bwd_finish_kernel(const float* __restrict__ K, int64_t ld,
                                                             const float* __restrict__ kstat,
                                                             const int64_t* __restrict__ length,
                                                             float* __restrict__ dK, int64_t ldd, int rows,
                                                             int H, const float* __restrict__ colpart) {
  __shared__ float smax[HD], sinv[HD], ssum[HD];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  int len = rows;
  if (length) len = (int)min<int64_t>(max<int64_t>(length[b], 0), rows);
  if (tid < HD) {
    const float* st = kstat + ((int64_t)blockIdx.x * HD + tid) * 2;
    smax[tid] = st[0];
    sinv[tid] = 1.0f / st[1];
    float s = 0.f;
    for (int c = 0; c < (int)gridDim.y; ++c) s += colpart[((int64_t)blockIdx.x * gridDim.y + c) * HD + tid];
    ssum[tid] = s;
  }
  __syncthreads();
  const float* Kb = K + (int64_t)b * rows * ld + h * HD;
  float* dKb = dK + (int64_t)b * rows * ldd + h * HD;
  const int r0 = blockIdx.y * CH, r1 = min(r0 + CH, len);
  for (int idx = tid; idx < (r1 - r0) * HD; idx += 256) {
    const int r = r0 + idx / HD, cc = idx % HD;
    const float kk = __expf(Kb[(int64_t)r * ld + cc] - smax[cc]) * sinv[cc];
    float* p = dKb + (int64_t)r * ldd + cc;
    *p = kk * (*p - ssum[cc]);
  }
}

// ---------------------------------------------------------------------------------------------
// hd = 64 / 128 backward kernels on the matrix cores (same grids, outputs and scratch as the VALU
// versions above; lane layout as in apply_mfma_kernel).
// ---------------------------------------------------------------------------------------------
template <int HD, typename TIO = float>
__global__ __launch_bounds__(256) void apply_bwd_mfma_kernel(const TIO* __restrict__ dY, int64_t lddy,
                                                             const TIO* __restrict__ Q, int64_t ldq,
                                                             const float* __restrict__ A,
                                                             TIO* __restrict__ dQ, int64_t lddq,
                                                             float* __restrict__ dApart, int rows, int H) {
  constexpr int LDP = HD + 4, TB = HD / 64, Q4 = HD / 4, NPRE = CH * Q4 / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                    // [c][l]            HD * LDP
  float* sQ = sA + HD * LDP;           // softmax(Q) [r][c] CH * LDP
  float* sD = sQ + CH * LDP;           // dY [r][l]         CH * LDP
  float* srow = sD + CH * LDP;         // [2][CH]
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int nchunk = (rows + CH - 1) / CH;
  const TIO* Qb = Q + (int64_t)b * rows * ldq + h * HD;
  const TIO* Db = dY + (int64_t)b * rows * lddy + h * HD;
  // The workgroup walks row chunks blockIdx.y, blockIdx.y + gridDim.y, ...: A[b,h] is staged once, the dA
  // accumulators live across chunks (one partial per workgroup instead of one per chunk), and the next Q / dY
  // tiles are requested into registers before the products of the current ones.
  // (two chunks in flight, register sets 0 / 1: see apply_mfma_kernel)
  typedef typename RawOf<TIO>::type raw_t;
  raw_t preq0[NPRE], pred0[NPRE], preq1[NPRE], pred1[NPRE];
  const TIO* qsrc[NPRE];                        // this thread's pieces of row (tid + 256 i) / Q4 of chunk 0: a chunk adds CH rows
  const TIO* dsrc[NPRE];
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    qsrc[i] = Qb + (int64_t)((tid + 256 * i) / Q4) * ldq + 4 * ((tid + 256 * i) % Q4);
    dsrc[i] = Db + (int64_t)((tid + 256 * i) / Q4) * lddy + 4 * ((tid + 256 * i) % Q4);
  }
  const int64_t chunk_q = (int64_t)CH * ldq, chunk_d = (int64_t)CH * lddy;
  auto fetch = [&](int chunk, raw_t (&preq)[NPRE], raw_t (&pred)[NPRE]) {
    const int rleft = chunk < nchunk ? rows - chunk * CH : 0;   // rows of this chunk that exist
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      if ((tid + 256 * i) / Q4 < rleft) {
        preq[i] = ld4raw(qsrc[i] + chunk * chunk_q);
        pred[i] = ld4raw(dsrc[i] + chunk * chunk_d);
      } else {
        raw_zero(preq[i]);
        raw_zero(pred[i]);
      }
    }
  };
  const int gstep = gridDim.y;
  // A[b,h] is requested FIRST (returns are in order: staged into LDS it would otherwise wait behind both chunk prefetches)
  const float* Ab = A + (int64_t)blockIdx.x * HD * HD;
  constexpr int NA4 = HD * HD / 4 / 256;
  float4 areg[NA4];
#pragma unroll
  for (int i = 0; i < NA4; ++i) areg[i] = reinterpret_cast<const float4*>(Ab)[tid + 256 * i];
  fetch(blockIdx.y, preq0, pred0);
  fetch(blockIdx.y + gstep, preq1, pred1);
#pragma unroll
  for (int i = 0; i < NA4; ++i) {
    const int idx = tid + 256 * i, c = idx / (HD / 4), l4 = idx % (HD / 4);
    *reinterpret_cast<float4*>(sA + c * LDP + 4 * l4) = areg[i];
  }
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int rl = wi * 32 + lr;
  f32x16 da[TB][TB];
#pragma unroll
  for (int ti = 0; ti < TB; ++ti)
#pragma unroll
    for (int tj = 0; tj < TB; ++tj) zero16(da[ti][tj]);
  // bf16 products: the operands of dq_pre that come from A[b,h] do not change over the chunks -- rounded and kept in registers
  [[maybe_unused]] la_bf16x8 afrag[TB][HD / 16];
  if constexpr (sizeof(TIO) == 2) {
    lds_barrier();                             // sA is complete
    const float* yrow = sA + (wj * (HD / 2) + lr) * LDP + 8 * lh;
#pragma unroll
    for (int tj = 0; tj < TB; ++tj)
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) afrag[tj][ks] = frag_row8(yrow + 32 * tj * LDP + 16 * ks);
  }
  auto do_chunk = [&](int chunk, raw_t (&preq)[NPRE], raw_t (&pred)[NPRE]) {
    const int r0 = chunk * CH;
    float4 qv[NPRE];
#pragma unroll
    for (int i = 0; i < NPRE; ++i) qv[i] = raw_cvt(preq[i]);
    row_softmax_regs<HD, NPRE>(qv);
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<float4*>(sQ + (idx / Q4) * LDP + 4 * (idx % Q4)) = qv[i];
      *reinterpret_cast<float4*>(sD + (idx / Q4) * LDP + 4 * (idx % Q4)) = raw_cvt(pred[i]);
    }
    lds_barrier();
    fetch(chunk + 2 * gstep, preq, pred);        // (this set is free again: two chunks ahead)
    // dq_pre[r][c] = sum_l dY[r][l] A[c][l]   (64 x HD, reduce over HD)
    f32x16 dq[TB];
#pragma unroll
    for (int tj = 0; tj < TB; ++tj) zero16(dq[tj]);
    if constexpr (sizeof(TIO) == 2) {      // bf16 products: k = l, 16 per MFMA, lane (lr, lh) supplies l = 16 ks + 8 lh .. + 7
      const float* xrow = sD + rl * LDP + 8 * lh;
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        const la_bf16x8 xf = frag_row8(xrow + 16 * ks);
#pragma unroll
        for (int tj = 0; tj < TB; ++tj)
          dq[tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[tj][ks], xf, dq[tj], 0, 0, 0);
      }
    } else {
      const float* xrow = sD + rl * LDP + 4 * lh;
      const float* yrow = sA + (wj * (HD / 2) + lr) * LDP + 4 * lh;
#pragma unroll 4
      for (int ks = 0; ks < HD / 8; ++ks) {
        const float4 x4 = *reinterpret_cast<const float4*>(xrow + 8 * ks);
#pragma unroll
        for (int tj = 0; tj < TB; ++tj) {
          const float4 y4 = *reinterpret_cast<const float4*>(yrow + 32 * tj * LDP + 8 * ks);
          dq[tj] = mfma4(dq[tj], y4.x, y4.y, y4.z, y4.w, x4);
        }
      }
    }
    // dA[c][l] += sum_r q[r][c] dY[r][l]   (HD x HD, reduce over the 64 rows; rows past `rows` are zero in sD)
    if constexpr (sizeof(TIO) == 2) {      // bf16 products: k = row, lane (lr, lh) supplies rows 16 ks + 8 lh .. + 7 of its column
      const float* xcol = sQ + (8 * lh) * LDP + wi * (HD / 2) + lr;
      const float* ycol = sD + (8 * lh) * LDP + wj * (HD / 2) + lr;
#pragma unroll
      for (int ks = 0; ks < CH / 16; ++ks) {
        la_bf16x8 xf[TB], yf[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          xf[t] = frag_col8(xcol + 16 * ks * LDP + 32 * t, LDP);
          yf[t] = frag_col8(ycol + 16 * ks * LDP + 32 * t, LDP);
        }
#pragma unroll
        for (int ti = 0; ti < TB; ++ti)
#pragma unroll
          for (int tj = 0; tj < TB; ++tj)
            da[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[tj], xf[ti], da[ti][tj], 0, 0, 0);
      }
    } else {
      const float* xcol = sQ + (4 * lh) * LDP + wi * (HD / 2) + lr;
      const float* ycol = sD + (4 * lh) * LDP + wj * (HD / 2) + lr;
#pragma unroll 2
      for (int ks = 0; ks < CH / 8; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float xv[TB], yv[TB];
#pragma unroll
          for (int t = 0; t < TB; ++t) {
            xv[t] = xcol[(8 * ks + j) * LDP + 32 * t];
            yv[t] = ycol[(8 * ks + j) * LDP + 32 * t];
          }
#pragma unroll
          for (int ti = 0; ti < TB; ++ti)
#pragma unroll
            for (int tj = 0; tj < TB; ++tj)
              da[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(yv[tj], xv[ti], da[ti][tj], 0, 0, 0);
        }
      }
    }
    // softmax Jacobian over the HD channels of a row: the row lives in 2 lanes (lh) x 2 waves (wj)
    float part = 0.f;
#pragma unroll
    for (int tj = 0; tj < TB; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 qv = *reinterpret_cast<const float4*>(sQ + rl * LDP + wj * (HD / 2) + 32 * tj + 8 * q + 4 * lh);
        part += qv.x * dq[tj][4 * q] + qv.y * dq[tj][4 * q + 1] + qv.z * dq[tj][4 * q + 2] + qv.w * dq[tj][4 * q + 3];
      }
    part += __shfl_xor(part, 32, 64);
    if (lh == 0) srow[wj * CH + rl] = part;
    lds_barrier();
    const float sdot = srow[rl] + srow[CH + rl];
    // (the barrier above also ended every wave's reads of sD: the dQ tile is staged there and leaves as whole rows)
#pragma unroll
    for (int tj = 0; tj < TB; ++tj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 qv = *reinterpret_cast<const float4*>(sQ + rl * LDP + wj * (HD / 2) + 32 * tj + 8 * q + 4 * lh);
        *reinterpret_cast<float4*>(sD + rl * LDP + wj * (HD / 2) + 32 * tj + 8 * q + 4 * lh) =
            make_float4(qv.x * (dq[tj][4 * q] - sdot), qv.y * (dq[tj][4 * q + 1] - sdot),
                        qv.z * (dq[tj][4 * q + 2] - sdot), qv.w * (dq[tj][4 * q + 3] - sdot));
      }
    lds_barrier();
    store_tile_rows<HD, TIO>(sD, dQ + (int64_t)b * rows * lddq + h * HD, lddq, r0, rows);
    lds_barrier();   // sQ / sD / srow are rewritten by the next chunk
  };
  for (int chunk = blockIdx.y; chunk < nchunk; chunk += 2 * gstep) {
    do_chunk(chunk, preq0, pred0);
    if (chunk + gstep < nchunk) do_chunk(chunk + gstep, preq1, pred1);
  }
  float* dAb = dApart + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * HD * HD;
#pragma unroll
  for (int ti = 0; ti < TB; ++ti)
#pragma unroll
    for (int tj = 0; tj < TB; ++tj)
      store16(dAb + (wi * (HD / 2) + 32 * ti + lr) * HD + wj * (HD / 2) + 32 * tj + 4 * lh, da[ti][tj]);
}

// Single pass: the column term of the column-softmax Jacobian,  S[c] = sum_r k[r][c] dk[r][c],  needs no pass over
// the rows:  dk[r][c] = sum_l V[r][l] dA[c][l]  gives  S[c] = sum_l dA[c][l] (sum_r k[r][c] V[r][l]) = sum_l dA[c][l] A[c][l],
// a row-wise dot of the two hd x hd matrices the workgroup already stages -- so dK is finished here.
template <int HD, typename TIO = float>
__global__ __launch_bounds__(256) void ctx_bwd_mfma_kernel(const float* __restrict__ dA, const float* __restrict__ A,
                                                           const TIO* __restrict__ K, const TIO* __restrict__ V,
                                                           int64_t ld, const float* __restrict__ kstat,
                                                           const int64_t* __restrict__ length,
                                                           TIO* __restrict__ dK, TIO* __restrict__ dV, int64_t ldd,
                                                           int rows, int H) {
  constexpr int LDP = HD + 4, TB = HD / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sdA = smem;                   // [c][l]                      HD * LDP
  float* sK = sdA + HD * LDP;          // k (normalised) [r][c]       CH * LDP
  float* sV = sK + CH * LDP;           // [r][l]                      CH * LDP
  float* smax = sV + CH * LDP;         // [HD]; after the staging loop: S[c]
  float* sinv = smax + HD;             // [HD]
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  int len = rows;
  if (length) len = (int)min<int64_t>(max<int64_t>(length[b], 0), rows);
  const float* dAb = dA + (int64_t)blockIdx.x * HD * HD;
  const float* Ab = A + (int64_t)blockIdx.x * HD * HD;
  // dA / A are requested first, the chunk prefetches right behind them (below), and only then are they consumed
  constexpr int NA4 = HD * HD / 4 / 256;
  float4 dreg[NA4], areg[NA4];
#pragma unroll
  for (int i = 0; i < NA4; ++i) {
    dreg[i] = reinterpret_cast<const float4*>(dAb)[tid + 256 * i];
    areg[i] = reinterpret_cast<const float4*>(Ab)[tid + 256 * i];
  }
  const TIO* Kb = K + (int64_t)b * rows * ld + h * HD;
  const TIO* Vb = V + (int64_t)b * rows * ld + h * HD;
  const int nchunk = (rows + CH - 1) / CH;
  constexpr int Q4 = HD / 4, NPRE = CH * Q4 / 256;
  // chunk-walking like apply_bwd: dA / S staged once per workgroup, next K / V tiles prefetched into registers
  // (two chunks in flight, raw register images: see apply_mfma_kernel / RawOf)
  typedef typename RawOf<TIO>::type raw_t;
  raw_t prek0[NPRE], prev0[NPRE], prek1[NPRE], prev1[NPRE];
  int64_t kvoff[NPRE];                           // this thread's piece of row (tid + 256 i) / Q4 of chunk 0: a chunk adds CH rows
#pragma unroll
  for (int i = 0; i < NPRE; ++i) kvoff[i] = (int64_t)((tid + 256 * i) / Q4) * ld + 4 * ((tid + 256 * i) % Q4);
  const int64_t chunk_kv = (int64_t)CH * ld;
  auto fetch = [&](int chunk, raw_t (&prek)[NPRE], raw_t (&prev)[NPRE]) {
    const int rleft = chunk < nchunk ? len - chunk * CH : 0;    // valid rows of this chunk
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      if ((tid + 256 * i) / Q4 < rleft) {
        prek[i] = ld4raw(Kb + kvoff[i] + chunk * chunk_kv);
        prev[i] = ld4raw(Vb + kvoff[i] + chunk * chunk_kv);
      } else {
        raw_zero(prek[i]);
        raw_zero(prev[i]);
      }
    }
  };
  const int gstep = gridDim.y;
  fetch(blockIdx.y, prek0, prev0);
  fetch(blockIdx.y + gstep, prek1, prev1);
#pragma unroll
  for (int i = 0; i < NA4; ++i) {
    const int idx = tid + 256 * i;
    const int c = idx / (HD / 4), l4 = idx % (HD / 4);
    const float4 d4 = dreg[i];
    const float4 a4 = areg[i];
    *reinterpret_cast<float4*>(sdA + c * LDP + 4 * l4) = d4;
    const float pr = d4.x * a4.x + d4.y * a4.y + d4.z * a4.z + d4.w * a4.w;
    // HD / 4 consecutive threads (16 or 32 lanes of one wave) hold one row of this sweep: reduce inside the group
    float t = pr;
#pragma unroll
    for (int o = HD / 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (l4 == 0) sK[c] = t;          // sK is free until the staging loop below: park S there
  }
  if (tid < HD) {
    const float* st = kstat + ((int64_t)blockIdx.x * HD + tid) * 2;
    smax[tid] = st[0];
    sinv[tid] = 1.0f / st[1];
  }
  lds_barrier();
  float scol[TB][4][4];   // S[c] for the lane's columns (read before sK is overwritten)
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int rl = wi * 32 + lr;
#pragma unroll
  for (int tj = 0; tj < TB; ++tj)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 s4 = *reinterpret_cast<const float4*>(sK + wj * (HD / 2) + 32 * tj + 8 * q + 4 * lh);
      scol[tj][q][0] = s4.x; scol[tj][q][1] = s4.y; scol[tj][q][2] = s4.z; scol[tj][q][3] = s4.w;
    }
  // bf16 products: the dA operands do not change over the chunks -- rounded once, kept in registers
  [[maybe_unused]] la_bf16x8 dacol[TB][HD / 16], darow[TB][HD / 16];
  if constexpr (sizeof(TIO) == 2) {
    const float* acol = sdA + (8 * lh) * LDP + wj * (HD / 2) + lr;       // dV: dA[c][l], k = row c of the tile
    const float* arow = sdA + (wj * (HD / 2) + lr) * LDP + 8 * lh;       // dk: dA[c][l] by rows of c
#pragma unroll
    for (int tj = 0; tj < TB; ++tj)
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        dacol[tj][ks] = frag_col8(acol + 16 * ks * LDP + 32 * tj, LDP);
        darow[tj][ks] = frag_row8(arow + 32 * tj * LDP + 16 * ks);
      }
  }
  lds_barrier();
  auto do_chunk = [&](int chunk, raw_t (&prek)[NPRE], raw_t (&prev)[NPRE]) {
    const int r0 = chunk * CH;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {   // k = exp(K - max) / sum on valid rows, 0 (with V = 0) beyond `len`
      const int idx = tid + 256 * i, rr = idx / Q4, c4 = idx % Q4;
      float4 kk = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + rr < len) {
        const float4 mx = *reinterpret_cast<const float4*>(smax + 4 * c4);
        const float4 iv = *reinterpret_cast<const float4*>(sinv + 4 * c4);
        const float4 kr = raw_cvt(prek[i]);
        kk = make_float4(__expf(kr.x - mx.x) * iv.x, __expf(kr.y - mx.y) * iv.y, __expf(kr.z - mx.z) * iv.z, __expf(kr.w - mx.w) * iv.w);
      }
      *reinterpret_cast<float4*>(sK + rr * LDP + 4 * c4) = kk;
      *reinterpret_cast<float4*>(sV + rr * LDP + 4 * c4) = raw_cvt(prev[i]);
    }
    lds_barrier();
    fetch(chunk + 2 * gstep, prek, prev);        // (this set is free again: two chunks ahead)
    f32x16 dv[TB], dk[TB];
#pragma unroll
    for (int tj = 0; tj < TB; ++tj) {
      zero16(dv[tj]);
      zero16(dk[tj]);
    }
    if constexpr (sizeof(TIO) == 2) {      // bf16 products (see frag_row8 / frag_col8)
      const float* krow = sK + rl * LDP + 8 * lh;                          // dV: k[r][c], reduce over c (contiguous)
      const float* vrow = sV + rl * LDP + 8 * lh;                          // dk: V[r][l], reduce over l (contiguous)
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        const la_bf16x8 kf = frag_row8(krow + 16 * ks), vf = frag_row8(vrow + 16 * ks);
#pragma unroll
        for (int tj = 0; tj < TB; ++tj) {
          dv[tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dacol[tj][ks], kf, dv[tj], 0, 0, 0);
          dk[tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(darow[tj][ks], vf, dk[tj], 0, 0, 0);
        }
      }
    } else {
      const float* krow = sK + rl * LDP + 4 * lh;                          // X of dV: k[r][c], reduce over c
      const float* acol = sdA + (4 * lh) * LDP + wj * (HD / 2) + lr;       // Y of dV: dA[c][l] by rows of c
      const float* vrow = sV + rl * LDP + 4 * lh;                          // X of dk: V[r][l], reduce over l
      const float* arow = sdA + (wj * (HD / 2) + lr) * LDP + 4 * lh;       // Y of dk: dA[c][l] by rows of c
#pragma unroll 2
      for (int ks = 0; ks < HD / 8; ++ks) {
        const float4 k4 = *reinterpret_cast<const float4*>(krow + 8 * ks);
        const float4 v4 = *reinterpret_cast<const float4*>(vrow + 8 * ks);
#pragma unroll
        for (int tj = 0; tj < TB; ++tj) {
          const float* ap = acol + (8 * ks) * LDP + 32 * tj;
          dv[tj] = mfma4(dv[tj], ap[0], ap[LDP], ap[2 * LDP], ap[3 * LDP], k4);
          const float4 a4 = *reinterpret_cast<const float4*>(arow + 32 * tj * LDP + 8 * ks);
          dk[tj] = mfma4(dk[tj], a4.x, a4.y, a4.z, a4.w, v4);
        }
      }
    }
    // rows in [len, rows) carry k == 0 and V == 0 in LDS: dV == 0 and dK = k * (..) == 0 there.  Both tiles are staged in
    // place (dV over V, dK over k: each lane rewrites exactly the elements it has just read) and leave as whole rows.
    lds_barrier();   // every wave is done reading sK / sV
#pragma unroll
    for (int tj = 0; tj < TB; ++tj) {
      stage16<HD>(sV, rl, wj * (HD / 2) + 32 * tj + 4 * lh, dv[tj]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* kp = sK + rl * LDP + wj * (HD / 2) + 32 * tj + 8 * q + 4 * lh;
        const float4 k4 = *reinterpret_cast<const float4*>(kp);
        *reinterpret_cast<float4*>(kp) =
            make_float4(k4.x * (dk[tj][4 * q] - scol[tj][q][0]), k4.y * (dk[tj][4 * q + 1] - scol[tj][q][1]),
                        k4.z * (dk[tj][4 * q + 2] - scol[tj][q][2]), k4.w * (dk[tj][4 * q + 3] - scol[tj][q][3]));
      }
    }
    lds_barrier();
    store_tile_rows<HD, TIO>(sV, dV + (int64_t)b * rows * ldd + h * HD, ldd, r0, rows);
    store_tile_rows<HD, TIO>(sK, dK + (int64_t)b * rows * ldd + h * HD, ldd, r0, rows);
    lds_barrier();   // sK / sV are rewritten by the next chunk
  };
  for (int chunk = blockIdx.y; chunk < nchunk; chunk += 2 * gstep) {
    do_chunk(chunk, prek0, prev0);
    if (chunk + gstep < nchunk) do_chunk(chunk + gstep, prek1, prev1);
  }
}

// three [.][HD + 4] tiles + 2 * max(CH, HD) floats of small arrays: 52.7 KB at hd = 64, i.e. three
// workgroups per CU within the 1280-byte LDS allocation granule (54.3 KB would leave only two)
template <int HD> constexpr size_t attn_bwd_lds_bytes() {
  return sizeof(float) * ((size_t)HD * (HD + 4) + 2 * (size_t)CH * (HD + 4) + 2 * (HD > CH ? HD : CH));
}
// hd = 128 needs 135 KB of dynamic LDS (of the 160 KB per CU): raise the per-kernel limit once.
int allow_big_lds() {
  static const int rc = [] {
    const int bytes = (int)attn_bwd_lds_bytes<128>();
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&apply_bwd_mfma_kernel<128>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&ctx_bwd_mfma_kernel<128>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&apply_bwd_mfma_kernel<128, __bf16>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(&ctx_bwd_mfma_kernel<128, __bf16>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && e4 == hipSuccess) ? 0 : 1;
  }();
  return rc;
}

bool hd_ok(int hd) { return hd == 8 || hd == 16 || hd == 32 || hd == 64 || hd == 128; }

#define HD_SWITCH(hd, STMT)                  \
  switch (hd) {                              \
    case 8: { constexpr int HDV = 8; STMT; } break;     \
    case 16: { constexpr int HDV = 16; STMT; } break;   \
    case 32: { constexpr int HDV = 32; STMT; } break;   \
    case 64: { constexpr int HDV = 64; STMT; } break;   \
    default: { constexpr int HDV = 128; STMT; } break;  \
  }

}  // namespace

extern "C" int64_t hig_linattn_ctx_scratch_floats(int32_t B, int32_t rows, int32_t H, int32_t hd) {
  const int64_t nchunk = (rows + CH - 1) / CH;
  return (int64_t)B * H * nchunk * ((int64_t)hd * hd + 2 * hd);
}

namespace {
template <typename TIO>
int linattn_ctx_t(const TIO* K, const TIO* V, int64_t ld, int32_t B, int32_t rows, int32_t H, int32_t hd,
                  const int64_t* length, float* A, float* kstat, float* scratch, hipStream_t st, __bf16* At16 = nullptr) {
  const int nchunk = (rows + CH - 1) / CH;
  constexpr int ctx_walk = 1;   // (a former tuning knob, fixed at the value that won its A/B)
  const bool walk = ctx_walk && B * H >= hig_chip_cus();   // enough (sample, head) pairs to fill the chip with walking workgroups
  if (!walk && scratch && nchunk > 1) {
    // row chunks in parallel + a merge: 4-5x the workgroups of the one-per-(sample, head) kernel
    if (hd == 64) {
      hipLaunchKernelGGL((ctx_part_mfma_kernel<64, TIO>), dim3(B * H, nchunk), dim3(256), 0, st, K, V, ld, rows, H, length,
                         scratch);
      hipLaunchKernelGGL(ctx_combine_kernel<64>, dim3(B * H), dim3(256), 0, st, scratch, nchunk, A, kstat, At16);
    } else {
      hipLaunchKernelGGL((ctx_part_mfma_kernel<128, TIO>), dim3(B * H, nchunk), dim3(256), 0, st, K, V, ld, rows, H, length,
                         scratch);
      hipLaunchKernelGGL(ctx_combine_kernel<128>, dim3(B * H), dim3(256), 0, st, scratch, nchunk, A, kstat, At16);
    }
    HIG_CHECK_LAUNCH();
    return HIG_OK;
  }
  if (hd == 64)
    hipLaunchKernelGGL((ctx_mfma_kernel<64, TIO>), dim3(B * H), dim3(256), 0, st, K, V, ld, rows, H, length, A, kstat, At16, CtxGroups{H, 0, 0});
  else
    hipLaunchKernelGGL((ctx_mfma_kernel<128, TIO>), dim3(B * H), dim3(256), 0, st, K, V, ld, rows, H, length, A, kstat, At16, CtxGroups{H, 0, 0});
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

template <typename TIO>
int linattn_apply_t(const TIO* Q, int64_t ldq, const float* A, TIO* Y, int64_t ldy, int32_t B, int32_t rows, int32_t H,
                    int32_t hd, hipStream_t st) {
  // chunk-walking workgroups: enough of them to fill the chip (~4 per CU), each staging A[b,h] once
  const int nchunk_a = (rows + CH - 1) / CH;
  // (hd = 128 keeps 97 KB of LDS per workgroup = one per CU: fewer, longer-lived workgroups; measured with
  // tools/attn_time.py: 61 -> 47 us at config 5, neutral at hd = 64)
  if constexpr (sizeof(TIO) == 4) {
    // exact fp32, head dim 64: the wave-autonomous kernel (16-row tiles per wave, no barrier in the loop)
    static const int wave_env = getenv("HIG_APPLY_WAVE") ? atoi(getenv("HIG_APPLY_WAVE")) : 1;   // tuning knob
    // (short sequences keep the chunk-walking kernel: T = 91 has 6 tiles for 4 waves, 11.3 against 10.1 us at B = 64)
    if (wave_env && hd == 64 && rows >= 128 && ldq % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(Q) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(Y) & 15) == 0) {
      hipLaunchKernelGGL(apply_wave64_kernel, dim3(B * H), dim3(256), 0, st, reinterpret_cast<const float*>(Q), ldq, A,
                         reinterpret_cast<float*>(Y), ldy, rows, H);
      HIG_CHECK_LAUNCH();
      return HIG_OK;
    }
  }
  constexpr int apply_target = 0;   // (a former tuning knob, fixed at the value that won its A/B)
  // (re-swept in round 2, profiles/r02_attn_sweep.md: hd = 128 at 256 / 512 / 1024 workgroups: 42.4 / 48.7 / 58.9 us)
  const int target = apply_target > 0 ? apply_target : (hd == 128 ? 1 : 4) * hig_chip_cus();
  int gy = (target + B * H - 1) / (B * H);
  gy = gy < 1 ? 1 : (gy > nchunk_a ? nchunk_a : gy);
  if (hd == 64)
    hipLaunchKernelGGL((apply_mfma_kernel<64, TIO>), dim3(B * H, gy), dim3(256), 0, st, Q, ldq, A, Y, ldy, rows, H);
  else
    hipLaunchKernelGGL((apply_mfma_kernel<128, TIO>), dim3(B * H, gy), dim3(256), 0, st, Q, ldq, A, Y, ldy, rows, H);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
}  // namespace

extern "C" int hig_linattn_ctx(const float* K, const float* V, int64_t ld, int32_t B, int32_t rows,
                               int32_t H, int32_t hd, const int64_t* length, float* A, float* kstat,
                               float* scratch, hig_stream_t stream) {
  HIG_REQUIRE(K && V && A && kstat && B > 0 && rows > 0 && H > 0, "hig_linattn_ctx: bad arguments");
  HIG_REQUIRE(hd_ok(hd), "hig_linattn: head dim %d not in {8,16,32,64,128}", hd);
  if (hd == 64 || hd == 128) return linattn_ctx_t<float>(K, V, ld, B, rows, H, hd, length, A, kstat, scratch, hig_stream(stream));
  HD_SWITCH(hd, hipLaunchKernelGGL((ctx_kernel<HDV>), dim3(B * H), dim3(256), 0, hig_stream(stream), K, V,
                                   ld, rows, H, length, A, kstat));
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// The context build of G groups of H heads in ONE launch (the L layers of the batched text side, denoiser.hip): K / V hold
// G H heads side by side (head g H + h at column (g H + h) hd), group g's outputs go to A + g a_gs / kstat + g k_gs in the
// plain (B, H, ...) layout.  fp32 rows, head dim 64 / 128, no length mask; returns 1 when the shape is not served (the
// caller loops over the groups instead).
int hig_linattn_ctx_groups(const float* K, const float* V, int64_t ld, int32_t B, int32_t rows, int32_t H, int32_t G, int32_t hd,
                           float* A, int64_t a_gs, float* kstat, int64_t k_gs, hipStream_t st) {
  if (!(hd == 64 || hd == 128) || B <= 0 || rows <= 0 || H <= 0 || G <= 0) return 1;
  if (hd == 64)
    hipLaunchKernelGGL((ctx_mfma_kernel<64, float>), dim3(B * H * G), dim3(256), 0, st, K, V, ld, rows, H * G, nullptr, A, kstat,
                       nullptr, CtxGroups{H, a_gs, k_gs});
  else
    hipLaunchKernelGGL((ctx_mfma_kernel<128, float>), dim3(B * H * G), dim3(256), 0, st, K, V, ld, rows, H * G, nullptr, A, kstat,
                       nullptr, CtxGroups{H, a_gs, k_gs});
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_linattn_apply(const float* Q, int64_t ldq, const float* A, float* Y, int64_t ldy,
                                 int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream) {
  HIG_REQUIRE(Q && A && Y && B > 0 && rows > 0 && H > 0, "hig_linattn_apply: bad arguments");
  HIG_REQUIRE(hd_ok(hd), "hig_linattn: head dim %d not in {8,16,32,64,128}", hd);
  HIG_REQUIRE(ldq % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(Q) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(Y) & 15) == 0,
              "hig_linattn_apply: Q/Y must be 16-byte aligned");
  if (hd == 64 || hd == 128) return linattn_apply_t<float>(Q, ldq, A, Y, ldy, B, rows, H, hd, hig_stream(stream));
  HD_SWITCH(hd, hipLaunchKernelGGL((apply_kernel<HDV>), dim3(B * H, (rows + CH - 1) / CH), dim3(256), 0,
                                   hig_stream(stream), Q, ldq, A, Y, ldy, rows, H));
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

namespace {

// SiLU of the fused stylization front: exact division for an fp32 result (== ln_mod_silu_kernel), v_rcp_f32 when the
// result is rounded to bf16 anyway
template <typename TO> __device__ __forceinline__ float apply_sty_act(float x) { return hig_silu_fast(x); }
template <> __device__ __forceinline__ float apply_sty_act<float>(float x) { return hig_silu(x); }

// ---------------------------------------------------------------------------------------------
// apply + stylization front in ONE kernel (inference: y is not kept):
//     a[m, :] = silu( LN_d( y[m, :] ) * (1 + scale[b]) + shift[b] ),   y[m, h*HD + l] = sum_c softmax_c(Q[m, h*HD + :])[c] A[b,h][c][l]
// (transformer.py:111,116-118 followed by :81-85).  A workgroup owns 32 rows of one sample for ALL heads: wave w keeps
// the heads w*H/4 ... of y in its accumulators (fp32 MFMA 32x32x2), the LayerNorm statistics are folded across lanes /
// waves, and only `a` reaches HBM -- the (rows x d) round trip of y and one launch per attention are gone.
// Per wave: the query tile of a head (32 rows x HD) arrives by coalesced 16-byte loads and is turned into the MFMA
// layout through a wave-private LDS tile; the context slab (16 KiB of A[b,h]) sits in a second wave-private buffer.
// Both are fetched into registers ONE STEP AHEAD (the next slab while the current one is multiplied, the next head's
// queries during the current head), so a wave's MFMAs do not wait behind its own loads.  Wave-private LDS needs no
// barrier: the LDS executes one wave's operations in order.
// TQ / TO: storage type of the queries / of `a` (float or __bf16).
// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec16 {};
template <> struct Vec16<float> { static constexpr int N = 4; };
template <> struct Vec16<__bf16> { static constexpr int N = 8; };
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));   // native vectors: arrays of them stay in registers (SROA)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void put16(float* dst, u32x4_t raw, const float*) {
  *reinterpret_cast<u32x4_t*>(dst) = raw;
}
__device__ __forceinline__ void put16(float* dst, u32x4_t raw, const __bf16*) {   // 8 bf16 -> 8 floats
  *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u),
                                             __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xffff0000u)};
  *reinterpret_cast<f32x4_t*>(dst + 4) = f32x4_t{__uint_as_float(raw.z << 16), __uint_as_float(raw.z & 0xffff0000u),
                                                 __uint_as_float(raw.w << 16), __uint_as_float(raw.w & 0xffff0000u)};
}

template <int HD, typename TQ, typename TO>
__global__ __launch_bounds__(256) void apply_sty_kernel(const TQ* __restrict__ Q, int64_t ldq,
                                                        const float* __restrict__ A, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ ss,
                                                        int64_t ss_ld, int shift_off, TO* __restrict__ Out,
                                                        int64_t ldo, int rows, int H, int nblk) {
  constexpr int TB = HD / 32;            // 32-column blocks per head
  constexpr int KC = 4096 / HD;          // context channels per 16 KiB slab
  constexpr int NSLAB = HD / KC;         // slabs per head
  constexpr int MAXHPW = 2;              // heads per wave (H <= 8)
  constexpr int QLD = HD + 4;            // row stride of the wave's query tile (floats)
  constexpr int EQ = Vec16<TQ>::N;       // query elements per 16-byte load
  constexpr int CPR = HD / EQ;           // 16-byte pieces per query row of one head
  constexpr int NQL = 32 * CPR / 64;     // ... per lane
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  __shared__ float red[2][4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, kk = lane >> 5;
  const int b = blockIdx.x / nblk, r0 = (blockIdx.x % nblk) * 32;
  const int HPW = H / 4;
  const int d = H * HD;
  constexpr int WBUF = 32 * QLD > KC * HD ? 32 * QLD : KC * HD;   // floats per wave
  float* sAw = reinterpret_cast<float*>(smem_dyn) + wave * WBUF;   // [KC][HD] context slab of the current step
  float* sQw = sAw;   // [32][QLD] query tile of a head: lives there only until its rows are back in registers (in MFMA
                      // order), i.e. before the head's first slab is written -- one wave's LDS operations execute in order

  f32x16 acc[MAXHPW][TB];
#pragma unroll
  for (int hh = 0; hh < MAXHPW; ++hh)
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) zero16(acc[hh][tb]);

  u32x4_t pq[NQL];   // next head's query tile, as loaded
  f32x4_t pa[16];    // next context slab
  auto fetch_q = [&](int h) {
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
      const int idx = lane + 64 * i, rr = idx / CPR, c = idx % CPR;
      const int row = min(r0 + rr, rows - 1);
      pq[i] = *reinterpret_cast<const u32x4_t*>(Q + ((int64_t)b * rows + row) * ldq + h * HD + EQ * c);
    }
  };
  auto fetch_a = [&](int h, int c0) {
    const f32x4_t* src = reinterpret_cast<const f32x4_t*>(A + (((int64_t)b * H + h) * HD + c0) * HD);
#pragma unroll
    for (int i = 0; i < 16; ++i) pa[i] = src[lane + 64 * i];
  };
  fetch_q(wave * HPW);
  fetch_a(wave * HPW, 0);

#pragma unroll
  for (int hh = 0; hh < MAXHPW; ++hh) {
    if (hh >= HPW) break;
    const int h = wave * HPW + hh;
    // query tile -> LDS (fp32), then this lane's share of row lr: channels 8ks + 4kk + j (the MFMA k order)
#pragma unroll
    for (int i = 0; i < NQL; ++i) {
      const int idx = lane + 64 * i, rr = idx / CPR, c = idx % CPR;
      put16(sQw + rr * QLD + EQ * c, pq[i], static_cast<const TQ*>(nullptr));
    }
    if (hh + 1 < HPW) fetch_q(h + 1);
    f32x4_t q[HD / 8];
    float m = -INFINITY;
#pragma unroll
    for (int ks = 0; ks < HD / 8; ++ks) {
      q[ks] = *reinterpret_cast<const f32x4_t*>(sQw + lr * QLD + 8 * ks + 4 * kk);
      m = fmaxf(m, fmaxf(fmaxf(q[ks].x, q[ks].y), fmaxf(q[ks].z, q[ks].w)));
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < HD / 8; ++ks) {
      q[ks].x = __expf(q[ks].x - m); q[ks].y = __expf(q[ks].y - m); q[ks].z = __expf(q[ks].z - m); q[ks].w = __expf(q[ks].w - m);
      sum += (q[ks].x + q[ks].y) + (q[ks].z + q[ks].w);
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int ks = 0; ks < HD / 8; ++ks) q[ks] *= inv;
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
      const int c0 = sl * KC;
#pragma unroll
      for (int i = 0; i < 16; ++i) reinterpret_cast<f32x4_t*>(sAw)[lane + 64 * i] = pa[i];
      if (sl + 1 < NSLAB) fetch_a(h, c0 + KC);
      else if (hh + 1 < HPW) fetch_a(h + 1, 0);
#pragma unroll
      for (int ks = c0 / 8; ks < (c0 + KC) / 8; ++ks) {
        const float* ap = sAw + (8 * ks + 4 * kk - c0) * HD + lr;
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
          acc[hh][tb] = mfma4(acc[hh][tb], ap[32 * tb], ap[HD + 32 * tb], ap[2 * HD + 32 * tb], ap[3 * HD + 32 * tb], make_float4(q[ks].x, q[ks].y, q[ks].z, q[ks].w));
      }
    }
  }
  // ---- LayerNorm statistics of row lr over all d columns: lane -> lane pair -> the 4 waves ----
  float s = 0.f;
#pragma unroll
  for (int hh = 0; hh < MAXHPW; ++hh)
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[hh][tb][e];      // (zero for hh >= HPW)
  s += __shfl_xor(s, 32, 64);
  if (kk == 0) red[0][wave][lr] = s;
  __syncthreads();
  const float mean = ((red[0][0][lr] + red[0][1][lr]) + (red[0][2][lr] + red[0][3][lr])) / (float)d;
  float qd = 0.f;
#pragma unroll
  for (int hh = 0; hh < MAXHPW; ++hh) {
    if (hh >= HPW) break;
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float dv = acc[hh][tb][e] - mean;
        qd += dv * dv;
      }
  }
  qd += __shfl_xor(qd, 32, 64);
  if (kk == 0) red[1][wave][lr] = qd;
  __syncthreads();     // (also: every wave is done with its LDS buffers -- they are reused for the output tile)
  const float rstd = rsqrtf(((red[1][0][lr] + red[1][1][lr]) + (red[1][2][lr] + red[1][3][lr])) / (float)d + 1e-5f);
  // ---- modulate + SiLU, staged as a [32][d] tile of TO (padded rows), out as whole rows ----
  TO* sO = reinterpret_cast<TO*>(smem_dyn);
  const int ldso = d + 16 / (int)sizeof(TO);
  const float* ssrow = ss + (int64_t)b * ss_ld;
#pragma unroll
  for (int hh = 0; hh < MAXHPW; ++hh) {
    if (hh >= HPW) break;
    const int h = wave * HPW + hh;
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int col = h * HD + 32 * tb + 8 * qq + 4 * kk;
        const float4 g4 = *reinterpret_cast<const float4*>(gamma + col), b4 = *reinterpret_cast<const float4*>(beta + col);
        const float4 sc = *reinterpret_cast<const float4*>(ssrow + col), sh = *reinterpret_cast<const float4*>(ssrow + shift_off + col);
        float4 o;
        o.x = apply_sty_act<TO>(((acc[hh][tb][4 * qq] - mean) * rstd * g4.x + b4.x) * (1.0f + sc.x) + sh.x);
        o.y = apply_sty_act<TO>(((acc[hh][tb][4 * qq + 1] - mean) * rstd * g4.y + b4.y) * (1.0f + sc.y) + sh.y);
        o.z = apply_sty_act<TO>(((acc[hh][tb][4 * qq + 2] - mean) * rstd * g4.z + b4.z) * (1.0f + sc.z) + sh.z);
        o.w = apply_sty_act<TO>(((acc[hh][tb][4 * qq + 3] - mean) * rstd * g4.w + b4.w) * (1.0f + sc.w) + sh.w);
        st4(sO + lr * ldso + col, o);
      }
  }
  __syncthreads();
  constexpr int EO = 16 / (int)sizeof(TO);     // output elements per 16-byte chunk
  const int c16 = d / EO;                      // 16-byte chunks per row
  for (int idx = tid; idx < 32 * c16; idx += 256) {
    const int rr = idx / c16, ch = idx % c16;
    if (r0 + rr < rows)
      *reinterpret_cast<uint4*>(Out + ((int64_t)b * rows + r0 + rr) * ldo + EO * ch) =
          *reinterpret_cast<const uint4*>(sO + rr * ldso + EO * ch);
  }
}

template <typename TQ, typename TO>
int launch_apply_sty(const TQ* q, int64_t ldq, const float* A, const float* gamma, const float* beta, const float* ss,
                     int64_t ss_ld, int32_t shift_off, TO* o, int64_t ldo, int32_t B, int32_t rows, int32_t H, int32_t hd,
                     hipStream_t st) {
  const int nblk = (rows + 31) / 32;
  const int d = H * hd;
  const size_t wbuf = (size_t)32 * (hd + 4) > 4096 ? (size_t)32 * (hd + 4) : 4096;
  const size_t lds_main = 4 * wbuf * sizeof(float);
  const size_t lds_out = (size_t)32 * (d + 16 / sizeof(TO)) * sizeof(TO);
  const size_t lds = lds_main > lds_out ? lds_main : lds_out;
  static const int big_lds_rc = [] {   // more than the default dynamic-LDS cap
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&apply_sty_kernel<64, TQ, TO>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&apply_sty_kernel<128, TQ, TO>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    return (e1 == hipSuccess && e2 == hipSuccess) ? 0 : 1;
  }();
  if (big_lds_rc != 0 || lds > 156 * 1024)
    return hig_set_error(HIG_EHIP, "hig_linattn_apply_sty: cannot reserve %zu bytes of LDS", lds);
  if (hd == 64)
    hipLaunchKernelGGL((apply_sty_kernel<64, TQ, TO>), dim3(B * nblk), dim3(256), lds, st, q, ldq, A, gamma, beta, ss, ss_ld,
                       shift_off, o, ldo, rows, H, nblk);
  else
    hipLaunchKernelGGL((apply_sty_kernel<128, TQ, TO>), dim3(B * nblk), dim3(256), lds, st, q, ldq, A, gamma, beta, ss, ss_ld,
                       shift_off, o, ldo, rows, H, nblk);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

}  // namespace

// bf16-storage forms (hig_dims.storage == HIG_STORE_BF16): K / V / Q / Y are bf16, the context matrices and the
// softmax statistics stay fp32.  Head dim 64 or 128 (the MFMA kernels).
extern "C" int hig_linattn_ctx_bf16(const void* K, const void* V, int64_t ld, int32_t B, int32_t rows, int32_t H,
                                    int32_t hd, const int64_t* length, float* A, float* kstat, float* scratch,
                                    void* At16, hig_stream_t stream) {
  HIG_REQUIRE(K && V && A && kstat && B > 0 && rows > 0 && H > 0, "hig_linattn_ctx_bf16: bad arguments");
  if (hd != 64 && hd != 128)
    return hig_set_error(HIG_EUNSUPPORTED, "hig_linattn: bf16 storage is built for head dim 64 / 128 (got %d)", hd);
  HIG_REQUIRE(ld % 4 == 0 && (reinterpret_cast<uintptr_t>(K) & 7) == 0 && (reinterpret_cast<uintptr_t>(V) & 7) == 0,
              "hig_linattn_ctx_bf16: K/V must be 8-byte aligned");
  return linattn_ctx_t<__bf16>(static_cast<const __bf16*>(K), static_cast<const __bf16*>(V), ld, B, rows, H, hd, length, A,
                               kstat, scratch, hig_stream(stream), static_cast<__bf16*>(At16));
}
extern "C" int hig_linattn_apply_bf16(const void* Q, int64_t ldq, const float* A, void* Y, int64_t ldy, int32_t B,
                                      int32_t rows, int32_t H, int32_t hd, hig_stream_t stream) {
  HIG_REQUIRE(Q && A && Y && B > 0 && rows > 0 && H > 0, "hig_linattn_apply_bf16: bad arguments");
  if (hd != 64 && hd != 128)
    return hig_set_error(HIG_EUNSUPPORTED, "hig_linattn: bf16 storage is built for head dim 64 / 128 (got %d)", hd);
  HIG_REQUIRE(ldq % 4 == 0 && ldy % 8 == 0 && (reinterpret_cast<uintptr_t>(Q) & 7) == 0 &&
                  (reinterpret_cast<uintptr_t>(Y) & 15) == 0,
              "hig_linattn_apply_bf16: Q rows must be 8-byte aligned, Y rows 16-byte aligned");
  return linattn_apply_t<__bf16>(static_cast<const __bf16*>(Q), ldq, A, static_cast<__bf16*>(Y), ldy, B, rows, H, hd,
                                 hig_stream(stream));
}

extern "C" int64_t hig_linattn_bwd_scratch_floats(int32_t B, int32_t rows, int32_t H, int32_t hd) {
  const int64_t nchunk = (rows + CH - 1) / CH;
  return (int64_t)B * H * nchunk * ((int64_t)hd * hd + hd);
}

extern "C" int hig_linattn_apply_bwd(const float* dY, int64_t lddy, const float* Q, int64_t ldq,
                                     const float* A, float* dQ, int64_t lddq, float* dA, int32_t B,
                                     int32_t rows, int32_t H, int32_t hd, float* scratch,
                                     hig_stream_t stream) {
  HIG_REQUIRE(dY && Q && A && dQ && dA && scratch && B > 0 && rows > 0 && H > 0,
              "hig_linattn_apply_bwd: bad arguments");
  HIG_REQUIRE(hd_ok(hd), "hig_linattn: head dim %d not in {8,16,32,64,128}", hd);
  HIG_REQUIRE(ldq % 4 == 0 && lddy % 4 == 0 && lddq % 4 == 0 && (reinterpret_cast<uintptr_t>(Q) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(dY) & 15) == 0 && (reinterpret_cast<uintptr_t>(dQ) & 15) == 0,
              "hig_linattn_apply_bwd: Q/dY/dQ must be 16-byte aligned");
  const int nchunk = (rows + CH - 1) / CH;
  int nparts = nchunk;   // dA partials per (sample, head) that chunk_sum_kernel adds up
  if (hd == 64 || (hd == 128 && allow_big_lds() == 0)) {
    // chunk-walking workgroups (dA accumulated in registers across a workgroup's chunks): ~3 per CU resident
    constexpr int tgt = 0;   // (a former tuning knob, fixed at the value that won its A/B)
    // measured (tools/attn_time.py): one workgroup per (sample, head) walking all its chunks is fastest once
    // B * H fills the chip (config 2: 49 -> 38 us, config 5: 135 -> 77 us) and needs no partial sums at all
    const int target = tgt > 0 ? tgt : hig_chip_cus();
    nparts = (target + B * H - 1) / (B * H);
    nparts = nparts < 1 ? 1 : (nparts > nchunk ? nchunk : nparts);
    float* part = nparts == 1 ? dA : scratch;
    if (hd == 64)
      hipLaunchKernelGGL(apply_bwd_mfma_kernel<64>, dim3(B * H, nparts), dim3(256), attn_bwd_lds_bytes<64>(),
                         hig_stream(stream), dY, lddy, Q, ldq, A, dQ, lddq, part, rows, H);
    else
      hipLaunchKernelGGL(apply_bwd_mfma_kernel<128>, dim3(B * H, nparts), dim3(256), attn_bwd_lds_bytes<128>(),
                         hig_stream(stream), dY, lddy, Q, ldq, A, dQ, lddq, part, rows, H);
    if (nparts == 1) {
      HIG_CHECK_LAUNCH();
      return HIG_OK;
    }
  } else {
    HD_SWITCH(hd, hipLaunchKernelGGL((apply_bwd_kernel<HDV>), dim3(B * H, nchunk), dim3(256), 0, hig_stream(stream),
                                     dY, lddy, Q, ldq, A, dQ, lddq, scratch, rows, H));
  }
  HIG_CHECK_LAUNCH();
  const int64_t n = (int64_t)hd * hd, groups = (int64_t)B * H;
  const int64_t want = (groups * n / 4 + 255) / 256;
  hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)(want > 2048 ? 2048 : want)), dim3(256), 0, hig_stream(stream),
                     scratch, nparts, n, groups, dA);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_linattn_ctx_bwd(const float* dA, const float* A, const float* K, const float* V, int64_t ld,
                                   const float* kstat, const int64_t* length, float* dK, float* dV,
                                   int64_t ldd, int32_t B, int32_t rows, int32_t H, int32_t hd,
                                   float* scratch, hig_stream_t stream) {
  HIG_REQUIRE(dA && A && K && V && kstat && dK && dV && scratch && B > 0 && rows > 0 && H > 0,
              "hig_linattn_ctx_bwd: bad arguments");
  HIG_REQUIRE(hd_ok(hd), "hig_linattn: head dim %d not in {8,16,32,64,128}", hd);
  HIG_REQUIRE(ldd % 4 == 0 && (reinterpret_cast<uintptr_t>(dK) & 15) == 0 && (reinterpret_cast<uintptr_t>(dV) & 15) == 0,
              "hig_linattn_ctx_bwd: dK/dV must be 16-byte aligned");
  const int nchunk = (rows + CH - 1) / CH;
  if (hd == 64 || (hd == 128 && allow_big_lds() == 0)) {   // single pass (the column term comes from A and dA)
    constexpr int tgt = 0;   // (a former tuning knob, fixed at the value that won its A/B)
    const int target = tgt > 0 ? tgt : hig_chip_cus();
    int gy = (target + B * H - 1) / (B * H);
    gy = gy < 1 ? 1 : (gy > nchunk ? nchunk : gy);
    if (hd == 64)
      hipLaunchKernelGGL(ctx_bwd_mfma_kernel<64>, dim3(B * H, gy), dim3(256), attn_bwd_lds_bytes<64>(),
                         hig_stream(stream), dA, A, K, V, ld, kstat, length, dK, dV, ldd, rows, H);
    else
      hipLaunchKernelGGL(ctx_bwd_mfma_kernel<128>, dim3(B * H, gy), dim3(256), attn_bwd_lds_bytes<128>(),
                         hig_stream(stream), dA, A, K, V, ld, kstat, length, dK, dV, ldd, rows, H);
    HIG_CHECK_LAUNCH();
    return HIG_OK;
  }
  HD_SWITCH(hd, hipLaunchKernelGGL((ctx_bwd_kernel<HDV>), dim3(B * H, nchunk), dim3(256), 0, hig_stream(stream), dA,
                                   K, V, ld, kstat, length, dK, dV, ldd, rows, H, scratch));
  HIG_CHECK_LAUNCH();
  HD_SWITCH(hd, hipLaunchKernelGGL((ctx_bwd_finish_kernel<HDV>), dim3(B * H, nchunk), dim3(256), 0,
                                   hig_stream(stream), K, ld, kstat, length, dK, ldd, rows, H, scratch));
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}


// bf16-storage forms of the two backward kernels (training step with hig_dims.storage == HIG_STORE_BF16): dY, Q, dQ and K, V,
// dK, dV bf16; A, dA, kstat fp32 (the hd x hd matrices and the statistics stay fp32 in this mode); the products run on the
// fp32 matrix cores after the loads (exact fp32 arithmetic on the bf16 values).  Head dim 64 / 128.
extern "C" int hig_linattn_apply_bwd_bf16(const void* dY, int64_t lddy, const void* Q, int64_t ldq, const float* A, void* dQ,
                                          int64_t lddq, float* dA, int32_t B, int32_t rows, int32_t H, int32_t hd, float* scratch,
                                          hig_stream_t stream) {
  HIG_REQUIRE(dY && Q && A && dQ && dA && scratch && B > 0 && rows > 0 && H > 0, "hig_linattn_apply_bwd_bf16: bad arguments");
  if (!(hd == 64 || (hd == 128 && allow_big_lds() == 0)))
    return hig_set_error(HIG_EUNSUPPORTED, "hig_linattn_apply_bwd_bf16: head dim 64 or 128 (got %d)", hd);
  HIG_REQUIRE(ldq % 4 == 0 && lddy % 4 == 0 && lddq % 8 == 0 && ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(dY)) & 7) == 0 &&
                  (reinterpret_cast<uintptr_t>(dQ) & 15) == 0,
              "hig_linattn_apply_bwd_bf16: Q / dY rows must be 8-byte aligned, dQ rows 16-byte aligned");
  const int nchunk = (rows + CH - 1) / CH;
  constexpr int tgt16 = 0;   // (a former tuning knob, fixed at the value that won its A/B)
  const int target16 = tgt16 > 0 ? tgt16 : hig_chip_cus();
  int nparts = (target16 + B * H - 1) / (B * H);
  nparts = nparts < 1 ? 1 : (nparts > nchunk ? nchunk : nparts);
  float* part = nparts == 1 ? dA : scratch;
  const __bf16* dy = static_cast<const __bf16*>(dY);
  const __bf16* q = static_cast<const __bf16*>(Q);
  __bf16* dq = static_cast<__bf16*>(dQ);
  if (hd == 64)
    hipLaunchKernelGGL((apply_bwd_mfma_kernel<64, __bf16>), dim3(B * H, nparts), dim3(256), attn_bwd_lds_bytes<64>(), hig_stream(stream),
                       dy, lddy, q, ldq, A, dq, lddq, part, rows, H);
  else
    hipLaunchKernelGGL((apply_bwd_mfma_kernel<128, __bf16>), dim3(B * H, nparts), dim3(256), attn_bwd_lds_bytes<128>(), hig_stream(stream),
                       dy, lddy, q, ldq, A, dq, lddq, part, rows, H);
  HIG_CHECK_LAUNCH();
  if (nparts == 1) return HIG_OK;
  const int64_t n = (int64_t)hd * hd, groups = (int64_t)B * H;
  const int64_t want = (groups * n / 4 + 255) / 256;
  hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)(want > 2048 ? 2048 : want)), dim3(256), 0, hig_stream(stream), scratch, nparts, n,
                     groups, dA);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
extern "C" int hig_linattn_ctx_bwd_bf16(const float* dA, const float* A, const void* K, const void* V, int64_t ld, const float* kstat,
                                        const int64_t* length, void* dK, void* dV, int64_t ldd, int32_t B, int32_t rows, int32_t H,
                                        int32_t hd, hig_stream_t stream) {
  HIG_REQUIRE(dA && A && K && V && kstat && dK && dV && B > 0 && rows > 0 && H > 0, "hig_linattn_ctx_bwd_bf16: bad arguments");
  if (!(hd == 64 || (hd == 128 && allow_big_lds() == 0)))
    return hig_set_error(HIG_EUNSUPPORTED, "hig_linattn_ctx_bwd_bf16: head dim 64 or 128 (got %d)", hd);
  HIG_REQUIRE(ld % 4 == 0 && ldd % 8 == 0 && ((reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V)) & 7) == 0 &&
                  ((reinterpret_cast<uintptr_t>(dK) | reinterpret_cast<uintptr_t>(dV)) & 15) == 0,
              "hig_linattn_ctx_bwd_bf16: K / V rows must be 8-byte aligned, dK / dV rows 16-byte aligned");
  const int nchunk = (rows + CH - 1) / CH;
  constexpr int tgt16 = 0;   // (a former tuning knob, fixed at the value that won its A/B)
  const int target16 = tgt16 > 0 ? tgt16 : hig_chip_cus();
  int gy = (target16 + B * H - 1) / (B * H);
  gy = gy < 1 ? 1 : (gy > nchunk ? nchunk : gy);
  const __bf16* k = static_cast<const __bf16*>(K);
  const __bf16* v = static_cast<const __bf16*>(V);
  if (hd == 64)
    hipLaunchKernelGGL((ctx_bwd_mfma_kernel<64, __bf16>), dim3(B * H, gy), dim3(256), attn_bwd_lds_bytes<64>(), hig_stream(stream), dA, A, k, v,
                       ld, kstat, length, static_cast<__bf16*>(dK), static_cast<__bf16*>(dV), ldd, rows, H);
  else
    hipLaunchKernelGGL((ctx_bwd_mfma_kernel<128, __bf16>), dim3(B * H, gy), dim3(256), attn_bwd_lds_bytes<128>(), hig_stream(stream), dA, A, k,
                       v, ld, kstat, length, static_cast<__bf16*>(dK), static_cast<__bf16*>(dV), ldd, rows, H);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// Fused apply + stylization front (bf16 storage): Out = silu( LN(softmax_hd(Q) . A) * (1 + scale) + shift ), see
// apply_sty_kernel.  H must be 4 or 8, head dim 64 or 128; gamma / beta / ss fp32, 16-byte aligned.
extern "C" int hig_linattn_apply_sty_bf16(const void* Q, int64_t ldq, const float* A, const float* gamma,
                                          const float* beta, const float* ss, int64_t ss_ld, int32_t ss_shift_off,
                                          void* Out, int64_t ldo, int32_t B, int32_t rows, int32_t H, int32_t hd,
                                          hig_stream_t stream) {
  HIG_REQUIRE(Q && A && gamma && beta && ss && Out && B > 0 && rows > 0, "hig_linattn_apply_sty_bf16: bad arguments");
  if ((hd != 64 && hd != 128) || (H != 4 && H != 8))
    return hig_set_error(HIG_EUNSUPPORTED, "hig_linattn_apply_sty_bf16: built for head dim 64 / 128 and 4 or 8 heads (got %d, %d)", hd, H);
  HIG_REQUIRE(ldq % 8 == 0 && ldo % 8 == 0 && ss_ld % 4 == 0 && ss_shift_off % 4 == 0 &&
                  ((reinterpret_cast<uintptr_t>(Q) & 15) | (reinterpret_cast<uintptr_t>(Out) & 15) |
                   (reinterpret_cast<uintptr_t>(gamma) & 15) | (reinterpret_cast<uintptr_t>(beta) & 15) |
                   (reinterpret_cast<uintptr_t>(ss) & 15)) == 0,
              "hig_linattn_apply_sty_bf16: alignment");
  return launch_apply_sty<__bf16, __bf16>(static_cast<const __bf16*>(Q), ldq, A, gamma, beta, ss, ss_ld, ss_shift_off,
                                          static_cast<__bf16*>(Out), ldo, B, rows, H, hd, hig_stream(stream));
}

// fp32 storage form of the same kernel (inference forward of hig_denoiser_fwd): Q and Out fp32.
extern "C" int hig_linattn_apply_sty(const float* Q, int64_t ldq, const float* A, const float* gamma, const float* beta,
                                     const float* ss, int64_t ss_ld, int32_t ss_shift_off, float* Out, int64_t ldo,
                                     int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream) {
  HIG_REQUIRE(Q && A && gamma && beta && ss && Out && B > 0 && rows > 0, "hig_linattn_apply_sty: bad arguments");
  if ((hd != 64 && hd != 128) || (H != 4 && H != 8))
    return hig_set_error(HIG_EUNSUPPORTED, "hig_linattn_apply_sty: built for head dim 64 / 128 and 4 or 8 heads (got %d, %d)", hd, H);
  HIG_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && ss_ld % 4 == 0 && ss_shift_off % 4 == 0 &&
                  ((reinterpret_cast<uintptr_t>(Q) & 15) | (reinterpret_cast<uintptr_t>(Out) & 15) |
                   (reinterpret_cast<uintptr_t>(gamma) & 15) | (reinterpret_cast<uintptr_t>(beta) & 15) |
                   (reinterpret_cast<uintptr_t>(ss) & 15)) == 0,
              "hig_linattn_apply_sty: alignment");
  return launch_apply_sty<float, float>(Q, ldq, A, gamma, beta, ss, ss_ld, ss_shift_off, Out, ldo, B, rows, H, hd,
                                        hig_stream(stream));
}
