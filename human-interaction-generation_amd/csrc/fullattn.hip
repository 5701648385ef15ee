// Full softmax attention of the `no_eff=True` denoiser, forward and backward.
// Reference: TemporalSelfAttention / TemporalCrossAttention.forward,
// codes/models/transformer.py:208-227 and :242-262:
//     S[n,m] = (q_n . k_m) / sqrt(hd)  (+ (1 - mask[n]) * -100000 in self attention: the mask lands
//              on the QUERY axis -- constant along m, so it only costs fp32 logit precision on padded
//              query rows; keys are never masked.  Reproduced exactly: the constant is added in fp32
//              before the softmax.)
//     W = softmax_m(S);   y_n = sum_m W[n,m] v_m        (v is not masked)
// Flash-style: the T x T score matrix (78.7 MB per layer at B=64, T=196, H=8) is never written;
// forward keeps one log-sum-exp per (b,h,n), backward recomputes the probabilities from it.
//
// Two sets of kernels:
//  * head dim 64 / 128 (every model of the reference, the text head at Lt = 256 / 512, the evaluator encoders): the
//    matrix-core kernels in the second half of this file (v_mfma_f32_32x32x2_f32, exact fp32 products; 1.8-3.4x the
//    VALU kernels' speed, profiles/r02_notes.md section 5);
//  * head dim 8 / 16 / 32 (small test models): the VALU kernels that follow -- one workgroup = (sample, head, 64-row
//    chunk); thread (row = tid>>2, part = tid&3) scores its row against keys {part, part+4, ...} of each staged 64-key
//    chunk, the 4 lanes of a row combine by shuffles, probabilities cross lanes through a 64x64 LDS tile.
#include "hig_common.h"

namespace {

constexpr int CH = 64;

template <int HD>
__device__ __forceinline__ void stage_tile(const float* __restrict__ src, int64_t ld, int r0, int rows,
                                           float* __restrict__ dst) {
  constexpr int LDP = HD + 4, Q = HD / 4;
  for (int idx = threadIdx.x; idx < CH * Q; idx += 256) {
    const int rr = idx / Q, c4 = idx % Q, r = r0 + rr;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) v = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + 4 * c4);
    *reinterpret_cast<float4*>(dst + rr * LDP + 4 * c4) = v;
  }
}
template <int HD>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&r)[HD], bool valid) {
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = *reinterpret_cast<const float4*>(p + c);
    r[c] = v.x; r[c + 1] = v.y; r[c + 2] = v.z; r[c + 3] = v.w;
  }
}
template <int HD>
__device__ __forceinline__ float dot_row(const float (&a)[HD], const float* __restrict__ b) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(b + c);
    s = fmaf(a[c], v.x, s); s = fmaf(a[c + 1], v.y, s); s = fmaf(a[c + 2], v.z, s); s = fmaf(a[c + 3], v.w, s);
  }
  return s;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  return fmaxf(v, __shfl_xor(v, 2, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  return v + __shfl_xor(v, 2, 64);
}
template <int PER>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float* v) {
  if constexpr (PER % 4 == 0) {
#pragma unroll
    for (int e = 0; e < PER; e += 4) *reinterpret_cast<float4*>(p + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
  } else {
#pragma unroll
    for (int e = 0; e < PER; ++e) p[e] = v[e];
  }
}

// query-axis additive constant of the reference's self attention (0 when qlen == nullptr)
__device__ __forceinline__ float query_const(const int64_t* qlen, int b, int n) {
  return (qlen && n >= qlen[b]) ? -100000.0f : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// forward: Y[n, h*HD + :] = softmax_m(S[n, :]) V,  lse[b,h,n] = log sum_m exp(S[n,m])
// grid = (B*H, ceil(Tq / 64))
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void full_fwd_kernel(const float* __restrict__ Q, int64_t ldq,
                                                       const float* __restrict__ K,
                                                       const float* __restrict__ V, int64_t ldk, int Tq,
                                                       int Tk, int H, const int64_t* __restrict__ qlen,
                                                       const uint8_t* __restrict__ kpad,
                                                       float* __restrict__ Y, int64_t ldy,
                                                       float* __restrict__ lse) {
  constexpr int LDP = HD + 4, PER = HD / 4;
  __shared__ __attribute__((aligned(16))) float sK[CH * LDP];
  __shared__ __attribute__((aligned(16))) float sV[CH * LDP];
  __shared__ float sP[CH * (CH + 1)];
  const int tid = threadIdx.x, rl = tid >> 2, part = tid & 3;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int n = blockIdx.y * CH + rl;
  const bool nvalid = n < Tq;
  const float sq = sqrtf((float)HD);
  float q[HD];
  load_row<HD>(Q + ((int64_t)b * Tq + (nvalid ? n : 0)) * ldq + h * HD, q, nvalid);
  const float addc = nvalid ? query_const(qlen, b, n) : 0.f;
  const float* Kb = K + (int64_t)b * Tk * ldk + h * HD;
  const float* Vb = V + (int64_t)b * Tk * ldk + h * HD;
  const uint8_t* pad = kpad ? kpad + (int64_t)b * Tk : nullptr;  // torch's src_key_padding_mask: 1 = not a key
  float m_run = -INFINITY, l_run = 0.f, acc[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) acc[e] = 0.f;
  for (int kc = 0; kc < Tk; kc += CH) {
    __syncthreads();
    stage_tile<HD>(Kb, ldk, kc, Tk, sK);
    stage_tile<HD>(Vb, ldk, kc, Tk, sV);
    __syncthreads();
    float s[16], cmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ml = 4 * j + part;
      s[j] = (kc + ml < Tk && !(pad && pad[kc + ml])) ? (dot_row<HD>(q, sK + ml * LDP) / sq + addc) : -INFINITY;
      cmax = fmaxf(cmax, s[j]);
    }
    cmax = quad_max(cmax);
    const float m_new = fmaxf(m_run, cmax);
    // exp(-inf) = 0 on the first chunk; a chunk of padded keys only (m_new still -inf) changes nothing
    const float alpha = m_new == -INFINITY ? 1.f : __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float p = s[j] == -INFINITY ? 0.f : __expf(s[j] - m_new);
      psum += p;
      sP[rl * (CH + 1) + 4 * j + part] = p;
    }
    l_run = l_run * alpha + quad_sum(psum);
    m_run = m_new;
#pragma unroll
    for (int e = 0; e < PER; ++e) acc[e] *= alpha;
    __syncthreads();
    for (int ml = 0; ml < CH; ++ml) {
      const float p = sP[rl * (CH + 1) + ml];
#pragma unroll
      for (int e = 0; e < PER; ++e) acc[e] = fmaf(p, sV[ml * LDP + part * PER + e], acc[e]);
    }
  }
  if (nvalid) {
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int e = 0; e < PER; ++e) acc[e] *= inv;
    store_vec<PER>(Y + ((int64_t)b * Tq + n) * ldy + h * HD + part * PER, acc);
    if (part == 0) lse[((int64_t)blockIdx.x) * Tq + n] = m_run + __logf(l_run);
  }
}

// ---------------------------------------------------------------------------------------------
// backward, query side: dQ[n] = sum_m dS[n,m] k_m / sqrt(hd),  dS = W * (dY.V^T - delta),
// delta[n] = dY[n].Y[n].   grid = (B*H, ceil(Tq / 64)).  Also writes delta[b,h,n] for the key pass.
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void full_bwd_q_kernel(const float* __restrict__ dY, int64_t lddy,
                                                         const float* __restrict__ Y, int64_t ldy,
                                                         const float* __restrict__ Q, int64_t ldq,
                                                         const float* __restrict__ K,
                                                         const float* __restrict__ V, int64_t ldk, int Tq,
                                                         int Tk, int H, const int64_t* __restrict__ qlen,
                                                         const float* __restrict__ lse,
                                                         float* __restrict__ delta, float* __restrict__ dQ,
                                                         int64_t lddq) {
  constexpr int LDP = HD + 4, PER = HD / 4;
  __shared__ __attribute__((aligned(16))) float sK[CH * LDP];
  __shared__ __attribute__((aligned(16))) float sV[CH * LDP];
  __shared__ float sP[CH * (CH + 1)];
  const int tid = threadIdx.x, rl = tid >> 2, part = tid & 3;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int n = blockIdx.y * CH + rl;
  const bool nvalid = n < Tq;
  const float sq = sqrtf((float)HD);
  const int64_t rowq = (int64_t)b * Tq + (nvalid ? n : 0);
  float q[HD], dy[HD];
  load_row<HD>(Q + rowq * ldq + h * HD, q, nvalid);
  load_row<HD>(dY + rowq * lddy + h * HD, dy, nvalid);
  float dl = 0.f;
  {
    const float* yp = Y + rowq * ldy + h * HD + part * PER;
#pragma unroll
    for (int e = 0; e < PER; ++e) dl += nvalid ? dy[part * PER + e] * yp[e] : 0.f;
    dl = quad_sum(dl);
  }
  const float addc = nvalid ? query_const(qlen, b, n) : 0.f;
  const float my_lse = nvalid ? lse[(int64_t)blockIdx.x * Tq + n] : 0.f;
  if (nvalid && part == 0) delta[(int64_t)blockIdx.x * Tq + n] = dl;
  const float* Kb = K + (int64_t)b * Tk * ldk + h * HD;
  const float* Vb = V + (int64_t)b * Tk * ldk + h * HD;
  float acc[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) acc[e] = 0.f;
  for (int kc = 0; kc < Tk; kc += CH) {
    __syncthreads();
    stage_tile<HD>(Kb, ldk, kc, Tk, sK);
    stage_tile<HD>(Vb, ldk, kc, Tk, sV);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ml = 4 * j + part;
      float ds = 0.f;
      if (kc + ml < Tk && nvalid) {
        const float s = dot_row<HD>(q, sK + ml * LDP) / sq + addc;
        const float p = __expf(s - my_lse);
        ds = p * (dot_row<HD>(dy, sV + ml * LDP) - dl) / sq;
      }
      sP[rl * (CH + 1) + ml] = ds;
    }
    __syncthreads();
    for (int ml = 0; ml < CH; ++ml) {
      const float ds = sP[rl * (CH + 1) + ml];
#pragma unroll
      for (int e = 0; e < PER; ++e) acc[e] = fmaf(ds, sK[ml * LDP + part * PER + e], acc[e]);
    }
  }
  if (nvalid) store_vec<PER>(dQ + ((int64_t)b * Tq + n) * lddq + h * HD + part * PER, acc);
}

// ---------------------------------------------------------------------------------------------
// backward, key side: dV[m] = sum_n W[n,m] dY[n],  dK[m] = sum_n dS[n,m] q_n / sqrt(hd)
// grid = (B*H, ceil(Tk / 64)); loops over 64-query chunks.
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void full_bwd_kv_kernel(const float* __restrict__ dY, int64_t lddy,
                                                          const float* __restrict__ Q, int64_t ldq,
                                                          const float* __restrict__ K,
                                                          const float* __restrict__ V, int64_t ldk, int Tq,
                                                          int Tk, int H, const int64_t* __restrict__ qlen,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta,
                                                          float* __restrict__ dK, float* __restrict__ dV,
                                                          int64_t lddk) {
  constexpr int LDP = HD + 4, PER = HD / 4;
  __shared__ __attribute__((aligned(16))) float sQ[CH * LDP];
  __shared__ __attribute__((aligned(16))) float sD[CH * LDP];
  __shared__ float sP[CH * (CH + 1)];   // [key row][query] probabilities
  __shared__ float sS[CH * (CH + 1)];   // [key row][query] dS
  __shared__ float s_lse[CH], s_delta[CH], s_addc[CH];
  const int tid = threadIdx.x, rl = tid >> 2, part = tid & 3;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int m = blockIdx.y * CH + rl;
  const bool mvalid = m < Tk;
  const float sq = sqrtf((float)HD);
  float k[HD], v[HD];
  const int64_t rowk = (int64_t)b * Tk + (mvalid ? m : 0);
  load_row<HD>(K + rowk * ldk + h * HD, k, mvalid);
  load_row<HD>(V + rowk * ldk + h * HD, v, mvalid);
  const float* Qb = Q + (int64_t)b * Tq * ldq + h * HD;
  const float* Db = dY + (int64_t)b * Tq * lddy + h * HD;
  float dk[PER], dv[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) dk[e] = dv[e] = 0.f;
  for (int qc = 0; qc < Tq; qc += CH) {
    __syncthreads();
    stage_tile<HD>(Qb, ldq, qc, Tq, sQ);
    stage_tile<HD>(Db, lddy, qc, Tq, sD);
    if (tid < CH) {
      const int n = qc + tid;
      s_lse[tid] = n < Tq ? lse[(int64_t)blockIdx.x * Tq + n] : 0.f;
      s_delta[tid] = n < Tq ? delta[(int64_t)blockIdx.x * Tq + n] : 0.f;
      s_addc[tid] = n < Tq ? query_const(qlen, b, n) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int nl = 4 * j + part;
      float p = 0.f, ds = 0.f;
      if (qc + nl < Tq && mvalid) {
        const float s = dot_row<HD>(k, sQ + nl * LDP) / sq + s_addc[nl];
        p = __expf(s - s_lse[nl]);
        ds = p * (dot_row<HD>(v, sD + nl * LDP) - s_delta[nl]) / sq;
      }
      sP[rl * (CH + 1) + nl] = p;
      sS[rl * (CH + 1) + nl] = ds;
    }
    __syncthreads();
    for (int nl = 0; nl < CH; ++nl) {
      const float p = sP[rl * (CH + 1) + nl], ds = sS[rl * (CH + 1) + nl];
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        dv[e] = fmaf(p, sD[nl * LDP + part * PER + e], dv[e]);
        dk[e] = fmaf(ds, sQ[nl * LDP + part * PER + e], dk[e]);
      }
    }
  }
  if (mvalid) {
    store_vec<PER>(dK + ((int64_t)b * Tk + m) * lddk + h * HD + part * PER, dk);
    store_vec<PER>(dV + ((int64_t)b * Tk + m) * lddk + h * HD + part * PER, dv);
  }
}

// ---------------------------------------------------------------------------------------------
// Matrix-core variant for head dim 64 / 128 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
// One wave owns 32 queries (forward, query-side backward) or 32 keys (key-side backward); a workgroup is 4 such
// waves sharing LDS-staged 32-row chunks of the other side.  Everything is computed TRANSPOSED so that per-query
// (per-key) scalars are per-LANE scalars and no probability tile ever crosses lanes:
//     S^T[key][query] = K_chunk . Q^T     A = LDS rows (keys), B = the lane's own query row in registers
//       -> lane (lr, g) holds the 16 logits of query lr for keys 8*(i/4) + 4g + (i%4): the online softmax is 16
//          in-lane values + one xor-32 shuffle;
//     O^T[col][query] += V^T . P^T        A = V[key_i(g)][32cb + lr] (one LDS word), B = the lane's own p[i]
//       -> the k index an MFMA step consumes is key 8*(i/4)+(i%4) in lanes 0-31 and +4 in lanes 32-63, which is
//          exactly where the S^T accumulator left p[i]: no LDS round trip, no permute.
// The reduce order over the head dim is permuted identically on both operands (k = 8j + 4g + e).
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MC = 32;            // rows of the staged side per chunk

// a 32-row chunk of the staged side: fetched into registers one chunk ahead (fetch32), written to LDS at the top of the
// iteration that consumes it (put32) -- the global latency hides behind the previous chunk's MFMAs
// 4 consecutive elements of an fp32 or bf16 row as floats (bf16 storage: hig_fullattn_fwd_bf16)
typedef __bf16 bf16x4_fa __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4f(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4f(const __bf16* p) {
  const bf16x4_fa v = *reinterpret_cast<const bf16x4_fa*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4f(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4f(__bf16* p, float4 v) {
  *reinterpret_cast<bf16x4_fa*>(p) = bf16x4_fa{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}
// NW = waves per workgroup (2 / 4 / 8: 64 / 128 / 256 rows of the register side share one staged chunk)
// The prefetched chunk stays RAW in registers (bf16 rows: two dwords per piece) and is converted by put32: a conversion next
// to the load is the load's first use, i.e. a wait for the data in front of the current chunk's products (linattn.hip, RawOf).
typedef unsigned int fa_u32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct RawFa;
template <> struct RawFa<float> { typedef float4 type; };
template <> struct RawFa<__bf16> { typedef fa_u32x2 type; };
__device__ __forceinline__ void raw_ld(float4& r, const float* p) { r = *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void raw_ld(fa_u32x2& r, const __bf16* p) { r = *reinterpret_cast<const fa_u32x2*>(p); }
__device__ __forceinline__ void raw_clear(float4& r) { r = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void raw_clear(fa_u32x2& r) { r = fa_u32x2{0u, 0u}; }
__device__ __forceinline__ float4 raw_f4(const float4& r) { return r; }
__device__ __forceinline__ float4 raw_f4(const fa_u32x2& r) {
  return make_float4(__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                     __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u));
}
template <int HD, int NW, typename T = float>
struct Chunk32 { typename RawFa<T>::type v[MC * (HD / 4) / (64 * NW)]; };
template <int HD, int NW, typename T>
__device__ __forceinline__ void fetch32(const T* __restrict__ src, int64_t ld, int r0, int rows, Chunk32<HD, NW, T>& c) {
  constexpr int Q4 = HD / 4, NT = 64 * NW;
#pragma unroll
  for (int it = 0; it < MC * Q4 / NT; ++it) {
    const int idx = threadIdx.x + NT * it, rr = idx / Q4, c4 = idx % Q4, r = r0 + rr;
    raw_clear(c.v[it]);
    if (r < rows) raw_ld(c.v[it], src + (int64_t)r * ld + 4 * c4);
  }
}
template <int HD, int LD, int NW, typename T>
__device__ __forceinline__ void put32(const Chunk32<HD, NW, T>& c, float* __restrict__ dst) {
  constexpr int Q4 = HD / 4, NT = 64 * NW;
#pragma unroll
  for (int it = 0; it < MC * Q4 / NT; ++it) {
    const int idx = threadIdx.x + NT * it, rr = idx / Q4, c4 = idx % Q4;
    *reinterpret_cast<float4*>(dst + rr * LD + 4 * c4) = raw_f4(c.v[it]);
  }
}
// the lane's half of its own row in the permuted reduce order: r[j][e] = row[8j + 4g + e]
template <int HD, typename T>
__device__ __forceinline__ void load_half_row(const T* __restrict__ p, float (&r)[HD / 8][4], bool valid, int g) {
#pragma unroll
  for (int j = 0; j < HD / 8; ++j) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = ld4f(p + 8 * j + 4 * g);
    r[j][0] = v.x; r[j][1] = v.y; r[j][2] = v.z; r[j][3] = v.w;
  }
}
// C[staged row 8*(i/4)+4g+(i%4)][owner lr] = sum_k rows[.][k] * own[k]
template <int HD, int LD>
__device__ __forceinline__ f32x16 rows_dot(const float* __restrict__ rows, const float (&own)[HD / 8][4], int lr, int g) {
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int j = 0; j < HD / 8; ++j) {
    const float4 a = *reinterpret_cast<const float4*>(rows + lr * LD + 8 * j + 4 * g);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, own[j][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, own[j][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, own[j][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, own[j][3], acc, 0, 0, 0);
  }
  return acc;
}
// acc[cb][.] (column 32cb + 8*(i/4)+4g+(i%4), owner lr) += sum over the 32 staged rows of rows[r][col] * w[r]
template <int HD, int LD>
__device__ __forceinline__ void cols_acc(const float* __restrict__ rows, const f32x16& w, f32x16 (&acc)[HD / 32], int lr,
                                         int g) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float* rp = rows + (8 * (i / 4) + 4 * g + (i % 4)) * LD + lr;
#pragma unroll
    for (int cb = 0; cb < HD / 32; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(rp[32 * cb], w[i], acc[cb], 0, 0, 0);
  }
}
template <int HD, typename T>
__device__ __forceinline__ void store_cols(T* __restrict__ rowp, const f32x16 (&acc)[HD / 32], float scale, int g) {
#pragma unroll
  for (int cb = 0; cb < HD / 32; ++cb)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      st4f(rowp + 32 * cb + 8 * q + 4 * g,
           make_float4(acc[cb][4 * q] * scale, acc[cb][4 * q + 1] * scale, acc[cb][4 * q + 2] * scale, acc[cb][4 * q + 3] * scale));
}
__device__ __forceinline__ float half_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// forward.  grid = (B*H, ceil(Tq / (32 NW)))
template <int HD, int NW, typename TIO>
__global__ __launch_bounds__(64 * NW) void full_fwd_mfma_kernel(const TIO* __restrict__ Q, int64_t ldq,
                                                               const TIO* __restrict__ K, const TIO* __restrict__ V,
                                                               int64_t ldk, int Tq, int Tk, int H,
                                                               const int64_t* __restrict__ qlen,
                                                               const uint8_t* __restrict__ kpad, TIO* __restrict__ Y,
                                                               int64_t ldy, float* __restrict__ lse) {
  constexpr int LDK = HD + 4, LDV = HD + 8, NCB = HD / 32;
  __shared__ __attribute__((aligned(16))) float sK[MC * LDK];
  __shared__ __attribute__((aligned(16))) float sV[MC * LDV];
  __shared__ __attribute__((aligned(16))) float s_mask[MC];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, g = lane >> 5;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int q0 = blockIdx.y * (32 * NW) + wave * 32, n = q0 + lr;
  const bool wactive = q0 < Tq, nvalid = n < Tq;
  const float isq = HD == 64 ? 0.125f : 0.08838834764831845f;   // 1 / sqrt(HD)
  float qf[HD / 8][4];
  load_half_row<HD>(Q + ((int64_t)b * Tq + (nvalid ? n : 0)) * ldq + h * HD, qf, nvalid, g);
  const float addc = nvalid ? query_const(qlen, b, n) : 0.f;
  const TIO* Kb = K + (int64_t)b * Tk * ldk + h * HD;
  const TIO* Vb = V + (int64_t)b * Tk * ldk + h * HD;
  const uint8_t* pad = kpad ? kpad + (int64_t)b * Tk : nullptr;
  f32x16 o[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[cb][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  Chunk32<HD, NW, TIO> ck, cv;
  fetch32<HD, NW>(Kb, ldk, 0, Tk, ck);
  fetch32<HD, NW>(Vb, ldk, 0, Tk, cv);
  for (int kc = 0; kc < Tk; kc += MC) {
    __syncthreads();
    put32<HD, LDK, NW>(ck, sK);
    put32<HD, LDV, NW>(cv, sV);
    if (tid < MC) s_mask[tid] = (kc + tid < Tk && !(pad && pad[kc + tid])) ? 0.f : -INFINITY;
    __syncthreads();
    if (kc + MC < Tk) {
      fetch32<HD, NW>(Kb, ldk, kc + MC, Tk, ck);
      fetch32<HD, NW>(Vb, ldk, kc + MC, Tk, cv);
    }
    if (!wactive) continue;
    const f32x16 st = rows_dot<HD, LDK>(sK, qf, lr, g);
    f32x16 p;
    float cmax = -INFINITY;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 mk = *reinterpret_cast<const float4*>(s_mask + 8 * q + 4 * g);
      p[4 * q] = st[4 * q] * isq + addc + mk.x;
      p[4 * q + 1] = st[4 * q + 1] * isq + addc + mk.y;
      p[4 * q + 2] = st[4 * q + 2] * isq + addc + mk.z;
      p[4 * q + 3] = st[4 * q + 3] * isq + addc + mk.w;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) cmax = fmaxf(cmax, p[i]);
    cmax = half_max(cmax);
    const float m_new = fmaxf(m_run, cmax);
    // exp(-inf) = 0 on the first chunk; a chunk of padded keys only (m_new still -inf) changes nothing
    const float alpha = m_new == -INFINITY ? 1.f : __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      p[i] = p[i] == -INFINITY ? 0.f : __expf(p[i] - m_new);
      psum += p[i];
    }
    l_run = l_run * alpha + half_sum(psum);
    m_run = m_new;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[cb][e] *= alpha;
    cols_acc<HD, LDV>(sV, p, o, lr, g);
  }
  if (nvalid) {
    store_cols<HD>(Y + ((int64_t)b * Tq + n) * ldy + h * HD, o, 1.0f / l_run, g);
    if (g == 0 && lse) lse[(int64_t)blockIdx.x * Tq + n] = m_run + __logf(l_run);
  }
}

// backward, query side (also writes delta).  grid = (B*H, ceil(Tq / (32 NW)))
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW) void full_bwd_q_mfma_kernel(const float* __restrict__ dY, int64_t lddy,
                                                              const float* __restrict__ Y, int64_t ldy,
                                                              const float* __restrict__ Q, int64_t ldq,
                                                              const float* __restrict__ K, const float* __restrict__ V,
                                                              int64_t ldk, int Tq, int Tk, int H,
                                                              const int64_t* __restrict__ qlen,
                                                              const float* __restrict__ lse, float* __restrict__ delta,
                                                              float* __restrict__ dQ, int64_t lddq) {
  constexpr int LDK = HD + 8, LDV = HD + 4, NCB = HD / 32;
  __shared__ __attribute__((aligned(16))) float sK[MC * LDK];
  __shared__ __attribute__((aligned(16))) float sV[MC * LDV];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, g = lane >> 5;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int q0 = blockIdx.y * (32 * NW) + wave * 32, n = q0 + lr;
  const bool wactive = q0 < Tq, nvalid = n < Tq;
  const float isq = HD == 64 ? 0.125f : 0.08838834764831845f;   // 1 / sqrt(HD)
  const int64_t rowq = (int64_t)b * Tq + (nvalid ? n : 0);
  float qf[HD / 8][4], dyf[HD / 8][4];
  load_half_row<HD>(Q + rowq * ldq + h * HD, qf, nvalid, g);
  load_half_row<HD>(dY + rowq * lddy + h * HD, dyf, nvalid, g);
  float dl = 0.f;
  if (nvalid) {
    const float* yp = Y + rowq * ldy + h * HD + 4 * g;
#pragma unroll
    for (int j = 0; j < HD / 8; ++j) {
      const float4 yv = *reinterpret_cast<const float4*>(yp + 8 * j);
      dl += dyf[j][0] * yv.x + dyf[j][1] * yv.y + dyf[j][2] * yv.z + dyf[j][3] * yv.w;
    }
  }
  dl = half_sum(dl);
  const float addc = nvalid ? query_const(qlen, b, n) : 0.f;
  const float my_lse = nvalid ? lse[(int64_t)blockIdx.x * Tq + n] : INFINITY;   // +inf: p = 0 on rows past Tq
  if (nvalid && g == 0) delta[(int64_t)blockIdx.x * Tq + n] = dl;
  const float* Kb = K + (int64_t)b * Tk * ldk + h * HD;
  const float* Vb = V + (int64_t)b * Tk * ldk + h * HD;
  f32x16 dq[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[cb][e] = 0.f;
  Chunk32<HD, NW> ck, cv;
  fetch32<HD, NW>(Kb, ldk, 0, Tk, ck);
  fetch32<HD, NW>(Vb, ldk, 0, Tk, cv);
  for (int kc = 0; kc < Tk; kc += MC) {
    __syncthreads();
    put32<HD, LDK, NW>(ck, sK);
    put32<HD, LDV, NW>(cv, sV);
    __syncthreads();
    if (kc + MC < Tk) {
      fetch32<HD, NW>(Kb, ldk, kc + MC, Tk, ck);
      fetch32<HD, NW>(Vb, ldk, kc + MC, Tk, cv);
    }
    if (!wactive) continue;
    const f32x16 st = rows_dot<HD, LDK>(sK, qf, lr, g);
    const f32x16 dp = rows_dot<HD, LDV>(sV, dyf, lr, g);
    f32x16 ds;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = kc + 8 * (i / 4) + 4 * g + (i % 4);
      const float pr = key < Tk ? __expf(st[i] * isq + addc - my_lse) : 0.f;
      ds[i] = pr * (dp[i] - dl) * isq;
    }
    cols_acc<HD, LDK>(sK, ds, dq, lr, g);
  }
  if (nvalid) store_cols<HD>(dQ + ((int64_t)b * Tq + n) * lddq + h * HD, dq, 1.0f, g);
}

// backward, key side.  grid = (B*H, ceil(Tk / (32 NW))); the wave's 32 keys stay in registers, queries are staged.
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW) void full_bwd_kv_mfma_kernel(const float* __restrict__ dY, int64_t lddy,
                                                               const float* __restrict__ Q, int64_t ldq,
                                                               const float* __restrict__ K, const float* __restrict__ V,
                                                               int64_t ldk, int Tq, int Tk, int H,
                                                               const int64_t* __restrict__ qlen,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta, float* __restrict__ dK,
                                                               float* __restrict__ dV, int64_t lddk) {
  constexpr int LD = HD + 8, NCB = HD / 32;
  __shared__ __attribute__((aligned(16))) float sQ[MC * LD];
  __shared__ __attribute__((aligned(16))) float sD[MC * LD];
  __shared__ __attribute__((aligned(16))) float s_lse[MC], s_delta[MC], s_addc[MC];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, g = lane >> 5;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int m0 = blockIdx.y * (32 * NW) + wave * 32, m = m0 + lr;
  const bool wactive = m0 < Tk, mvalid = m < Tk;
  const float isq = HD == 64 ? 0.125f : 0.08838834764831845f;   // 1 / sqrt(HD)
  const int64_t rowk = (int64_t)b * Tk + (mvalid ? m : 0);
  float kf[HD / 8][4], vf[HD / 8][4];
  load_half_row<HD>(K + rowk * ldk + h * HD, kf, mvalid, g);
  load_half_row<HD>(V + rowk * ldk + h * HD, vf, mvalid, g);
  const float* Qb = Q + (int64_t)b * Tq * ldq + h * HD;
  const float* Db = dY + (int64_t)b * Tq * lddy + h * HD;
  f32x16 dk[NCB], dv[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[cb][e] = dv[cb][e] = 0.f;
  Chunk32<HD, NW> cq, cd;
  fetch32<HD, NW>(Qb, ldq, 0, Tq, cq);
  fetch32<HD, NW>(Db, lddy, 0, Tq, cd);
  for (int qc = 0; qc < Tq; qc += MC) {
    __syncthreads();
    put32<HD, LD, NW>(cq, sQ);
    put32<HD, LD, NW>(cd, sD);
    if (tid < MC) {
      const int n = qc + tid;
      s_lse[tid] = n < Tq ? lse[(int64_t)blockIdx.x * Tq + n] : INFINITY;   // +inf: p = 0 on rows past Tq
      s_delta[tid] = n < Tq ? delta[(int64_t)blockIdx.x * Tq + n] : 0.f;
      s_addc[tid] = n < Tq ? query_const(qlen, b, n) : 0.f;
    }
    __syncthreads();
    if (qc + MC < Tq) {
      fetch32<HD, NW>(Qb, ldq, qc + MC, Tq, cq);
      fetch32<HD, NW>(Db, lddy, qc + MC, Tq, cd);
    }
    if (!wactive) continue;
    const f32x16 st = rows_dot<HD, LD>(sQ, kf, lr, g);    // S[query 8*(i/4)+4g+(i%4)][key lr]
    const f32x16 dp = rows_dot<HD, LD>(sD, vf, lr, g);
    f32x16 p, ds;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 l4 = *reinterpret_cast<const float4*>(s_lse + 8 * q + 4 * g);
      const float4 d4 = *reinterpret_cast<const float4*>(s_delta + 8 * q + 4 * g);
      const float4 a4 = *reinterpret_cast<const float4*>(s_addc + 8 * q + 4 * g);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w}, av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e;
        p[i] = __expf(st[i] * isq + av[e] - lv[e]);
        ds[i] = p[i] * (dp[i] - dv4[e]) * isq;
      }
    }
    cols_acc<HD, LD>(sD, p, dv, lr, g);
    cols_acc<HD, LD>(sQ, ds, dk, lr, g);
  }
  if (mvalid) {
    store_cols<HD>(dK + ((int64_t)b * Tk + m) * lddk + h * HD, dk, 1.0f, g);
    store_cols<HD>(dV + ((int64_t)b * Tk + m) * lddk + h * HD, dv, 1.0f, g);
  }
}

// HIG_FULLATTN_VALU=1 keeps head dim 64 on the VALU kernels (A/B measurements); head dim 128 is matrix-core only
// Waves per workgroup of the matrix-core kernels (tuning knob HIG_FULLATTN_WAVES = 2 / 4 / 8).  From the sweep in
// profiles/r02_attn_sweep.md: 8 waves (256 rows share each staged 32-row chunk; <= 256 registers per lane, two waves
// per SIMD) win the forward at both head dims and the backward at head dim 64; the head-dim-128 backward needs more
// than 256 registers per lane (it spills at 8) and is fastest at 4.
int mfma_waves(bool backward, int hd) {
  static const int forced = [] { const char* e = getenv("HIG_FULLATTN_WAVES"); const int v = e ? atoi(e) : 0; return (v == 2 || v == 4 || v == 8) ? v : 0; }();
  if (forced) return forced;
  return (backward && hd == 128) ? 4 : 8;
}
#define FNW_SWITCH(BWD, HDIM, ...)                           \
  switch (mfma_waves(BWD, HDIM)) {                           \
    case 2: { constexpr int NWV = 2; __VA_ARGS__; } break;   \
    case 8: { constexpr int NWV = 8; __VA_ARGS__; } break;   \
    default: { constexpr int NWV = 4; __VA_ARGS__; } break;  \
  }
bool use_mfma(int hd) {
  static const bool valu = [] { const char* e = getenv("HIG_FULLATTN_VALU"); return e && atoi(e) != 0; }();
  return hd == 128 || (hd == 64 && !valu);
}

bool full_hd_ok(int hd) { return hd == 8 || hd == 16 || hd == 32 || hd == 64 || hd == 128; }
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#define FHD_SWITCH(hd, STMT)                            \
  switch (hd) {                                         \
    case 8: { constexpr int HDV = 8; STMT; } break;     \
    case 16: { constexpr int HDV = 16; STMT; } break;   \
    case 32: { constexpr int HDV = 32; STMT; } break;   \
    default: { constexpr int HDV = 64; STMT; } break;   \
  }

}  // namespace

extern "C" int hig_fullattn_fwd(const float* Q, int64_t ldq, const float* K, const float* V, int64_t ldk,
                                int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t hd,
                                const int64_t* qlen, float* Y, int64_t ldy, float* lse, hig_stream_t stream) {
  return hig_fullattn_fwd_kpad(Q, ldq, K, V, ldk, B, Tq, Tk, H, hd, qlen, nullptr, Y, ldy, lse, stream);
}

extern "C" int hig_fullattn_fwd_kpad(const float* Q, int64_t ldq, const float* K, const float* V, int64_t ldk,
                                     int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t hd,
                                     const int64_t* qlen, const uint8_t* kpad, float* Y, int64_t ldy, float* lse,
                                     hig_stream_t stream) {
  HIG_REQUIRE(Q && K && V && Y && lse && B > 0 && Tq > 0 && Tk > 0 && H > 0, "hig_fullattn_fwd: bad arguments");
  if (!full_hd_ok(hd))
    return hig_set_error(HIG_EUNSUPPORTED, "hig_fullattn: head dim %d not in {8,16,32,64,128}", hd);
  HIG_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldy % 4 == 0 && al16(Q) && al16(K) && al16(V) && al16(Y),
              "hig_fullattn_fwd: operands must be 16-byte aligned");
  if (use_mfma(hd)) {
    FNW_SWITCH(false, hd, {
      const dim3 grid(B * H, (Tq + 32 * NWV - 1) / (32 * NWV));
      if (hd == 128)
        hipLaunchKernelGGL((full_fwd_mfma_kernel<128, NWV, float>), grid, dim3(64 * NWV), 0, hig_stream(stream), Q, ldq, K, V, ldk, Tq,
                           Tk, H, qlen, kpad, Y, ldy, lse);
      else
        hipLaunchKernelGGL((full_fwd_mfma_kernel<64, NWV, float>), grid, dim3(64 * NWV), 0, hig_stream(stream), Q, ldq, K, V, ldk, Tq,
                           Tk, H, qlen, kpad, Y, ldy, lse);
    });
    HIG_CHECK_LAUNCH();
    return HIG_OK;
  }
  FHD_SWITCH(hd, hipLaunchKernelGGL((full_fwd_kernel<HDV>), dim3(B * H, (Tq + CH - 1) / CH), dim3(256), 0,
                                    hig_stream(stream), Q, ldq, K, V, ldk, Tq, Tk, H, qlen, kpad, Y, ldy, lse));
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// bf16-storage form of the forward (hig_dims.storage == HIG_STORE_BF16, inference): Q / K / V / Y bf16, no log-sum-exp.
// Head dim 64 or 128 (the matrix-core kernel; logits, softmax and accumulation in fp32).
extern "C" int hig_fullattn_fwd_bf16(const void* Q, int64_t ldq, const void* K, const void* V, int64_t ldk, int32_t B,
                                     int32_t Tq, int32_t Tk, int32_t H, int32_t hd, const int64_t* qlen, void* Y, int64_t ldy,
                                     hig_stream_t stream) {
  HIG_REQUIRE(Q && K && V && Y && B > 0 && Tq > 0 && Tk > 0 && H > 0, "hig_fullattn_fwd_bf16: bad arguments");
  if (hd != 64 && hd != 128) return hig_set_error(HIG_EUNSUPPORTED, "hig_fullattn_fwd_bf16: head dim %d not in {64,128}", hd);
  auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
  HIG_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldy % 4 == 0 && al8(Q) && al8(K) && al8(V) && al8(Y),
              "hig_fullattn_fwd_bf16: operands must be 8-byte aligned");
  const __bf16 *q = static_cast<const __bf16*>(Q), *k = static_cast<const __bf16*>(K), *v = static_cast<const __bf16*>(V);
  __bf16* y = static_cast<__bf16*>(Y);
  FNW_SWITCH(false, hd, {
    const dim3 grid(B * H, (Tq + 32 * NWV - 1) / (32 * NWV));
    if (hd == 128)
      hipLaunchKernelGGL((full_fwd_mfma_kernel<128, NWV, __bf16>), grid, dim3(64 * NWV), 0, hig_stream(stream), q, ldq, k, v, ldk,
                         Tq, Tk, H, qlen, (const uint8_t*)nullptr, y, ldy, (float*)nullptr);
    else
      hipLaunchKernelGGL((full_fwd_mfma_kernel<64, NWV, __bf16>), grid, dim3(64 * NWV), 0, hig_stream(stream), q, ldq, k, v, ldk,
                         Tq, Tk, H, qlen, (const uint8_t*)nullptr, y, ldy, (float*)nullptr);
  });
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

extern "C" int hig_fullattn_bwd(const float* dY, int64_t lddy, const float* Y, int64_t ldy, const float* Q,
                                int64_t ldq, const float* K, const float* V, int64_t ldk, int32_t B,
                                int32_t Tq, int32_t Tk, int32_t H, int32_t hd, const int64_t* qlen,
                                const float* lse, float* delta, float* dQ, int64_t lddq, float* dK,
                                float* dV, int64_t lddk, hig_stream_t stream) {
  HIG_REQUIRE(dY && Y && Q && K && V && lse && delta && dQ && dK && dV && B > 0 && Tq > 0 && Tk > 0 && H > 0,
              "hig_fullattn_bwd: bad arguments");
  if (!full_hd_ok(hd))
    return hig_set_error(HIG_EUNSUPPORTED, "hig_fullattn: head dim %d not in {8,16,32,64,128}", hd);
  HIG_REQUIRE(lddy % 4 == 0 && ldy % 4 == 0 && ldq % 4 == 0 && ldk % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 &&
                  al16(dY) && al16(Y) && al16(Q) && al16(K) && al16(V) && al16(dQ) && al16(dK) && al16(dV),
              "hig_fullattn_bwd: operands must be 16-byte aligned");
  if (use_mfma(hd)) {
    hipStream_t st = hig_stream(stream);
    if (hd == 128) {
      // head dim 128: four waves, always (the key-side kernel needs more than 256 registers per lane: the 8- and 2-wave
      // instances spilled and were never selected -- profiles/r02_attn_sweep.md -- so they are not built)
      constexpr int NWV = 4;
      const dim3 gq(B * H, (Tq + 32 * NWV - 1) / (32 * NWV)), gk(B * H, (Tk + 32 * NWV - 1) / (32 * NWV));
      hipLaunchKernelGGL((full_bwd_q_mfma_kernel<128, NWV>), gq, dim3(64 * NWV), 0, st, dY, lddy, Y, ldy, Q, ldq, K, V, ldk, Tq,
                         Tk, H, qlen, lse, delta, dQ, lddq);
      hipLaunchKernelGGL((full_bwd_kv_mfma_kernel<128, NWV>), gk, dim3(64 * NWV), 0, st, dY, lddy, Q, ldq, K, V, ldk, Tq, Tk, H,
                         qlen, lse, delta, dK, dV, lddk);
    } else {
      FNW_SWITCH(true, hd, {
        const dim3 gq(B * H, (Tq + 32 * NWV - 1) / (32 * NWV)), gk(B * H, (Tk + 32 * NWV - 1) / (32 * NWV));
        hipLaunchKernelGGL((full_bwd_q_mfma_kernel<64, NWV>), gq, dim3(64 * NWV), 0, st, dY, lddy, Y, ldy, Q, ldq, K, V, ldk, Tq,
                           Tk, H, qlen, lse, delta, dQ, lddq);
        hipLaunchKernelGGL((full_bwd_kv_mfma_kernel<64, NWV>), gk, dim3(64 * NWV), 0, st, dY, lddy, Q, ldq, K, V, ldk, Tq, Tk, H,
                           qlen, lse, delta, dK, dV, lddk);
      });
    }
    HIG_CHECK_LAUNCH();
    return HIG_OK;
  }
  FHD_SWITCH(hd, hipLaunchKernelGGL((full_bwd_q_kernel<HDV>), dim3(B * H, (Tq + CH - 1) / CH), dim3(256), 0,
                                    hig_stream(stream), dY, lddy, Y, ldy, Q, ldq, K, V, ldk, Tq, Tk, H, qlen, lse,
                                    delta, dQ, lddq));
  HIG_CHECK_LAUNCH();
  FHD_SWITCH(hd, hipLaunchKernelGGL((full_bwd_kv_kernel<HDV>), dim3(B * H, (Tk + CH - 1) / CH), dim3(256), 0,
                                    hig_stream(stream), dY, lddy, Q, ldq, K, V, ldk, Tq, Tk, H, qlen, lse, delta,
                                    dK, dV, lddk));
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
