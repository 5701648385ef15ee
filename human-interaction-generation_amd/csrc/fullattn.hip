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
// fp32 hd-long dot products run at the VALU rate on gfx950 (the f32 MFMA has the same rate), so
// this stays on the VALU: one workgroup = (sample, head, 64-row chunk); thread (row = tid>>2,
// part = tid&3) scores its row against keys {part, part+4, ...} of each staged 64-key chunk,
// the 4 lanes of a row combine by shuffles, probabilities cross lanes through a 64x64 LDS tile.
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

bool full_hd_ok(int hd) { return hd == 8 || hd == 16 || hd == 32 || hd == 64; }
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
    return hig_set_error(HIG_EUNSUPPORTED, "hig_fullattn: head dim %d not in {8,16,32,64}", hd);
  HIG_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldy % 4 == 0 && al16(Q) && al16(K) && al16(V) && al16(Y),
              "hig_fullattn_fwd: operands must be 16-byte aligned");
  FHD_SWITCH(hd, hipLaunchKernelGGL((full_fwd_kernel<HDV>), dim3(B * H, (Tq + CH - 1) / CH), dim3(256), 0,
                                    hig_stream(stream), Q, ldq, K, V, ldk, Tq, Tk, H, qlen, kpad, Y, ldy, lse));
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
    return hig_set_error(HIG_EUNSUPPORTED, "hig_fullattn: head dim %d not in {8,16,32,64}", hd);
  HIG_REQUIRE(lddy % 4 == 0 && ldy % 4 == 0 && ldq % 4 == 0 && ldk % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 &&
                  al16(dY) && al16(Y) && al16(Q) && al16(K) && al16(V) && al16(dQ) && al16(dK) && al16(dV),
              "hig_fullattn_bwd: operands must be 16-byte aligned");
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
