// bf16-storage linear attention, second half, as ONE kernel on the bf16 matrix cores:
//     Out = silu( LN_d( softmax_hd(Q) . A ) * (1 + scale) + shift )                       (bf16 in, bf16 out)
// i.e. `y = q A` of LinearTemporalSelfAttention / LinearTemporalCrossAttention (codes/models/transformer.py:111,116-117 and
// :147,152-153) followed by the front of the StylizationBlock that consumes it (:81-85: LayerNorm, (1 + scale) / shift
// from the time/text embedding, SiLU).  The (rows, d) attention output never reaches memory.
//
// Why a new kernel.  The bf16 forward ran hig_linattn_apply_bf16 (fp32 MFMA, 64 cycles each: 11.0 us at B = 32) and
// then hig_ln_bf16 (5.7 us): sixteen such pairs are 270 us of the 1.29 ms sampling step, each launch 3-5x above its
// bandwidth time (profiles/r03_kernel_stats_bf16_sampling_step.csv).  The fp32-MFMA fused kernel (apply_sty_kernel) was
// not faster than the pair.  Here the hd x hd products run on v_mfma_f32_32x32x16_bf16 (1/16 of the fp32 MFMA time),
// a workgroup owns 32 whole rows (all heads), so the LayerNorm statistics are a register / LDS reduction, and every
// global access is a whole 1-KiB row: the Q tile comes in by LDS-DMA (global_load_lds_dwordx4, XOR swizzle on the
// source address and on the read address), the result leaves through LDS as whole rows.
//
// CDNA4 mapping.  512 threads = 8 waves at 8 heads (one head per wave; 4 waves with two heads each at 4 heads) over the 32
// rows.  MFMA: the context matrix is the row ("weight") operand, transposed -- At[l][c] = A[c][l], bf16, which the context
// kernels write in matrix-core operand order (hig_at16_offset) so that a wave fetches its heads' operands global -> registers,
// one coalesced KiB each -- and softmax(q) the column operand straight from registers: the lane that holds row lr of a q
// fragment (8 channels) shares the row with lane lr + 32, so the row maximum / sum are one cross-lane exchange.  The
// accumulator leaves a lane 4 consecutive output columns per quad of ONE row: LayerNorm statistics = lane sums + one
// exchange + a reduction over the waves through LDS.  softmax(q) and A are rounded to bf16 for the product (fp32
// accumulate): the same rounding the bf16 storage mode applies to every other matrix operand.
//
// Also in this file: ctx16_mfma_kernel (the context build k^T v on the bf16 matrix cores), the fused forms that carry on with
// the stylization block's output projection (hig_attn_out16 / hig_rows_out16, out_gemm_rows) and hig_weight_frag16.
#include <stdlib.h>

#include "gemm16_epi.h"
#include "hig_host.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// HD: head dim (64 / 128); H: heads (4 or 8); NW: waves per workgroup (H % NW == 0); rows of Q / Out are d = H * HD bf16
// wide.  grid = (ceil(T / 32), B).
// Instruction budget.  One or two workgroups per CU means one or two waves per SIMD, and a wave64 VALU instruction takes
// its SIMD 4 cycles (a transcendental 16): the kernel is bound by the NUMBER of vector instructions, 64 elements per lane
// at d = 512 (tools/apply16_stamps.py: 14.6 K cycles per workgroup with the straightforward arithmetic, 2.5 K of them
// waiting for the first tile).  So the element-wise work is written on float pairs (v_pk_fma / v_pk_mul / v_pk_add_f32:
// two elements per issue): softmax = unpack, v_max3, one packed fma into the exp2 argument, exp2, packed sum, packed
// scale, packed convert; LayerNorm + modulation = TWO packed fmas, with gamma' = gamma (1 + scale), beta' = beta (1 +
// scale) + shift combined once per workgroup in LDS; SiLU = packed multiply, exp2, packed add, rcp, packed multiply.
// Context matrices.  At16 is stored fragment-major (hig_at16_offset): a wave reads the operands of ITS heads global ->
// registers, one coalesced KiB per MFMA operand, while the Q tile is still landing -- no LDS staging (at head dim 128 the
// eight matrices are 256 KB; at 64 it leaves LDS for a second workgroup per CU).
// GEMM = true (hig_attn_out16; d = 512, 8 waves): the workgroup does not store the activated tile but goes on with the
// stylization block's output projection and the residual update for ITS 32 rows -- h[rows] += a W^T + b -- : the tile sits
// in LDS in exactly the layout the weight-stationary kernel reads its X tiles from, wave w owns output columns 64 w .. + 63,
// the weight comes in MFMA-operand order (Y_frag layout of hig_weight_frag16) global -> registers, one coalesced KiB per
// operand, eight k-steps ahead; the residual rows were fetched by DMA at the start and are updated in place in LDS; the new
// rows leave as whole KiB, with their (sum, centred sum of squares) per 128-column panel when the next consumer folds its LayerNorm.
// Why: at M = 6 272 a launch costs ~4.4 us before its first instruction and the projection is bound by the CU's L2 fetch
// rate either way (a workgroup streams the whole 512 KB weight here, 128 KB + its X tiles there): the pair apply (9.2 us)
// + GEMM (10.5 us) becomes one launch.  Same products in the same order as the weight-stationary kernel.
struct OutGemmArgs {
  const __bf16* Wf;      // (d, d) weight in operand order
  const float* bias;     // (d)
  __bf16* h;             // residual stream, updated in place
  int64_t ldh;
  float* stats;          // [rows][4][2] or NULL
};

// Output projection + residual update of 32 whole rows, shared by the fused kernels below (8 waves = 512 threads, d = 512):
// sA = the activated tile, sH = the residual rows, both [32][1 KiB] bf16 with 16-byte chunk c of row r at c ^ (r & 15).
// Wave w owns output columns 64 w .. 64 w + 63 (two 32-column blocks) over 32 k-steps; the weight comes in operand order
// global -> registers, PF k-steps ahead; rows r0 .. r0 + 31 of sample b (T rows per sample) are written while r0 + r < T.
__device__ __forceinline__ void out_gemm_rows(const char* sA, char* sH, const OutGemmArgs& og, int b, int r0, int T) {
  constexpr int D_ = 512, ROWB = 1024, BR = 32, NT = 512, PPR = ROWB / 16;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- h[rows] += a W^T + bias: wave w owns columns 64 w .. 64 w + 63 (two 32-column blocks), 32 k-steps ---------------
  constexpr int NK = D_ / 16, PF = 7;          // k-steps; weight operands requested PF k-steps ahead (8: one fragment spilled, with a vmcnt(0) in the loop)
  f32x16 acc2[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(og.bias + 64 * wave + 32 * cb + 8 * q + 4 * lh);
      acc2[cb][4 * q] = b4.x; acc2[cb][4 * q + 1] = b4.y; acc2[cb][4 * q + 2] = b4.z; acc2[cb][4 * q + 3] = b4.w;
    }
  const __bf16* wbase = og.Wf + ((int64_t)(2 * wave) * NK * 64 + lane) * 8;    // block (2 w + cb, ks) at + ((cb NK + ks) 64) 8
  bf16x8 wring[PF][2];
#pragma unroll
  for (int ks = 0; ks < PF; ++ks)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) wring[ks][cb] = *reinterpret_cast<const bf16x8*>(wbase + (int64_t)(cb * NK + ks) * 512);
  const int tsw = (lr & 15) ^ lh;
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sA + lr * ROWB + 16 * ((2 * ks) ^ tsw));
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wring[ks % PF][cb], xf, acc2[cb], 0, 0, 0);
    if (ks + PF < NK) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) wring[ks % PF][cb] = *reinterpret_cast<const bf16x8*>(wbase + (int64_t)(cb * NK + ks + PF) * 512);
    }
  }
  // residual add in fp32, one rounding, IN PLACE in the residual tile (each lane reads the 8 bytes it overwrites)
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      char* hp = sH + lr * ROWB + 16 * ((8 * wave + 4 * cb + q) ^ (lr & 15)) + 8 * lh;
      const u32x2 rq = *reinterpret_cast<const u32x2*>(hp);
      const float v0 = acc2[cb][4 * q] + bf_lo(rq.x), v1 = acc2[cb][4 * q + 1] + bf_hi(rq.x);
      const float v2 = acc2[cb][4 * q + 2] + bf_lo(rq.y), v3 = acc2[cb][4 * q + 3] + bf_hi(rq.y);
      *reinterpret_cast<bf16x4*>(hp) = bf16x4{(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
    }
  __syncthreads();
  // whole rows out (+ the row statistics per 128-column panel for a LayerNorm-folding consumer, as gemm_ws16 writes them)
#pragma unroll
  for (int u = 0; u < BR * PPR / NT; ++u) {
    const int idx = tid + NT * u;
    const int r = idx / PPR, p = idx % PPR;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(sH + r * ROWB + 16 * (p ^ (r & 15)));
    const int64_t row = (int64_t)b * T + r0 + r;
    if (r0 + r < T) *reinterpret_cast<bf16x8*>(og.h + row * og.ldh + 8 * p) = v;
    if (og.stats) {
      float s1, s2;
      hig_panel_stats16(v, s1, s2);   // (sum, centred sum of squares) of the panel: the arithmetic of gemm_ws16's producer
      if ((p & 15) == 0 && r0 + r < T) *reinterpret_cast<float2*>(og.stats + (row * 4 + (p >> 4)) * 2) = make_float2(s1, s2);
    }
  }
}

// YOUT (training forward; not with GEMM): the attention output y = softmax(q) . A (bf16, the input of the LayerNorm) leaves as
// a second output through og.h / og.ldh -- the backward of the stylization block needs it.
template <int HD, int H, int NW, bool GEMM = false, bool YOUT = false>
__global__ __launch_bounds__(64 * NW, (GEMM ? 4 : 1)) void apply_sty16_kernel(const __bf16* __restrict__ Q, int64_t ldq,
                                                              const __bf16* __restrict__ At16, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ ss,
                                                              int64_t ss_ld, int shift_off, __bf16* __restrict__ Out, int64_t ldo,
                                                              int T, unsigned long long* __restrict__ stamps, const OutGemmArgs og) {
  constexpr int D_ = H * HD;                   // model width
  constexpr int ROWB = D_ * 2;                 // bytes of a Q / Out row
  constexpr int BR = 32;                       // rows per workgroup
  constexpr int NT = 64 * NW;                  // threads
  constexpr int HPW = H / NW;                  // heads per wave
  constexpr int NKS = HD / 16, NLB = HD / 32;  // MFMA k-steps / 32-column blocks per head
  constexpr int NPRE = NLB < 2 ? NLB : 2;      // column blocks whose context operands are requested before the softmax
  constexpr int QBYTES = BR * ROWB;
  constexpr float LOG2E = 1.4426950408889634f;
  static_assert(HD == 64 || HD == 128, "head dim 64 or 128");
  static_assert(ROWB == 512 || ROWB == 1024 || ROWB == 2048, "Q rows of 512, 1024 or 2048 bytes");
  static_assert(H % NW == 0, "whole heads per wave");
  static_assert(!GEMM || (D_ == 512 && NW == 8), "fused output projection: d = 512, 8 waves");
  static_assert(!(GEMM && YOUT), "the second output is for the unfused (training) form");
  __shared__ __attribute__((aligned(1024))) char smem[QBYTES + 4 * D_ * 4 + NW * BR * 2 * 4 + ((GEMM || YOUT) ? QBYTES : 0)];
  char* const sQ = smem;                                       // [32][ROWB] bf16, 16-byte chunk c of row r at c ^ (r & 15); later the output tile
  float* const sPar = reinterpret_cast<float*>(smem + QBYTES); // gamma | beta | scale | shift, then gamma' | beta'
  [[maybe_unused]] char* const sH = smem + QBYTES + 4 * D_ * 4 + NW * BR * 2 * 4;   // GEMM: the residual rows, same layout as sQ; YOUT: the y tile
  float* const sRed = sPar + 4 * D_;                           // [NW waves][32 rows][2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, r0 = blockIdx.x * BR;
  const __bf16* Qb = Q + (int64_t)b * T * ldq;
  auto stamp = [&](int k) {                    // diagnostic only (hig_linattn16_debug_stamps); stamps == NULL in every real run
    if (stamps && tid == 0) {
      unsigned long long tm;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm)::"memory");
      stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = tm;
    }
  };
  stamp(0);

  // ---- Q tile by DMA: whole rows, swizzled on the source side ---------------------------------------------------
  {
    constexpr int NDMA = QBYTES / 1024, NQ = NDMA / NW;
    static_assert(NDMA % NW == 0, "Q tile in whole rounds of the waves");
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int n = wave + NW * q;
      const __bf16* src;
      if constexpr (ROWB >= 1024) {
        constexpr int IPR = ROWB / 1024;        // instructions per row
        const int r = n / IPR, part = n % IPR;
        src = Qb + (int64_t)min(r0 + r, T - 1) * ldq + 8 * ((64 * part + lane) ^ (r & 15));
      } else {
        const int r = 2 * n + (lane >> 5);
        src = Qb + (int64_t)min(r0 + r, T - 1) * ldq + 8 * ((lane & 31) ^ (r & 15));
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sQ + n * 1024), 16, 0, 0);
    }
  }
  // ---- gamma | beta | scale | shift (fp32, D_ each) by DMA as well ----------------------------------------------------
  constexpr int NP = 4 * D_ * 4 / 1024;          // 1-KiB instructions for the four vectors
  constexpr int NPW = (NP + NW - 1) / NW;        // ... per wave (the last round may be partly filled: NP = 4 at d = 256)
  {
    static_assert((D_ * 4) % 1024 == 0, "parameter vectors in whole 1-KiB pieces");
    const float* ssb = ss + (int64_t)b * ss_ld;
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      const int n = min(wave + NW * q, NP - 1);   // piece n: vector n / (D_ / 256), part n % (D_ / 256); a surplus wave repeats the last piece
      const int vec = n / (D_ / 256), part = n % (D_ / 256);
      const float* base = vec == 0 ? gamma : vec == 1 ? beta : vec == 2 ? ssb : ssb + shift_off;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + part * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sPar) + n * 1024), 16, 0, 0);
    }
  }
  // ---- GEMM: the 32 residual rows h[b T + r0 ..] by DMA (same swizzle), landed together with the Q tile -----------------
  if constexpr (GEMM) {
#pragma unroll
    for (int q = 0; q < QBYTES / 1024 / NW; ++q) {
      const int n = wave + NW * q;                // row n of the tile (1 KiB rows)
      const __bf16* src = og.h + ((int64_t)b * T + min(r0 + n, T - 1)) * og.ldh + 8 * (lane ^ (n & 15));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sH + n * 1024), 16, 0, 0);
    }
  }
  // ---- context matrices of this wave's heads, At16[b][h] in fragment-major order: operand (lb, ks) = bytes
  // [(lb NKS + ks) KiB, + 1 KiB), 16 per lane.  The first NPRE column blocks are requested now, the rest after the softmax
  // (their registers are the softmax's until then) ------------------------------------------------------------------
  bf16x8 af[HPW][NLB][NKS];
  auto load_at = [&](int hh, int lb) {
    const __bf16* Ab = At16 + ((int64_t)b * H + wave + NW * hh) * HD * HD;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) af[hh][lb][ks] = *reinterpret_cast<const bf16x8*>(Ab + ((lb * NKS + ks) * 64 + lane) * 8);
  };
#pragma unroll
  for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
    for (int lb = 0; lb < NPRE; ++lb) load_at(hh, lb);
  // (requests so far, oldest first: Q tile, LayerNorm / modulation vectors, context operands.  The softmax below needs
  // only the Q tile: it runs while the context operands are still in flight.)
  // (raw s_barrier, not __syncthreads(): its fence would wait for the register loads as well)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HPW * NPRE * NKS) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  stamp(1);

  // ---- gamma' = gamma (1 + scale), beta' = beta (1 + scale) + shift, in place over gamma | beta (each thread its own
  // columns; read after the next barrier) ----------------------------------------------------------------------------
  for (int c = tid; c < D_; c += NT) {
    const float g = sPar[c], be = sPar[D_ + c], sc = 1.0f + sPar[2 * D_ + c], sh = sPar[3 * D_ + c];
    sPar[c] = g * sc;
    sPar[D_ + c] = fmaf(be, sc, sh);
  }

  // ---- per head: softmax over the head's channels (a row is shared by lanes lr and lr + 32), then y = p . A ----------
  bf16x8 pfs[HPW][NKS];
#pragma unroll
  for (int hh = 0; hh < HPW; ++hh) {
    const int h = wave + NW * hh;
    u32x4 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int chunk = (h * HD) / 8 + 2 * ks + lh;
      qf[ks] = *reinterpret_cast<const u32x4*>(sQ + lr * ROWB + 16 * (chunk ^ (lr & 15)));
    }
    f32x2 v[NKS][4];
    float mx = -INFINITY;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[ks][e] = f32x2{bf_lo(qf[ks][e]), bf_hi(qf[ks][e])};
        mx = max3f(mx, v[ks][e][0], v[ks][e][1]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m2 = -mx * LOG2E;
    f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 arg = __builtin_elementwise_fma(v[ks][e], f32x2{LOG2E, LOG2E}, f32x2{m2, m2});
        v[ks][e] = f32x2{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
        sum2 += v[ks][e];
      }
    float sum = sum2[0] + sum2[1];
    sum += __shfl_xor(sum, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 pp = v[ks][e] * inv;
        pfs[hh][ks][2 * e] = (__bf16)pp[0];
        pfs[hh][ks][2 * e + 1] = (__bf16)pp[1];
      }
  }
  stamp(2);
#pragma unroll
  for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
    for (int lb = NPRE; lb < NLB; ++lb) load_at(hh, lb);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // gamma' / beta' are complete
  asm volatile("" ::: "memory");
  stamp(3);
  f32x16 acc[HPW][NLB];
  f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};      // this lane's share of sum(y), sum(y^2) over its row
#pragma unroll
  for (int hh = 0; hh < HPW; ++hh) {
#pragma unroll
    for (int lb = 0; lb < NLB; ++lb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[hh][lb][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        acc[hh][lb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[hh][lb][ks], pfs[hh][ks], acc[hh][lb], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2 y = {acc[hh][lb][e], acc[hh][lb][e + 1]};
        s1 += y;
        s2 = __builtin_elementwise_fma(y, y, s2);
      }
    }
  }
  // ---- LayerNorm statistics of the 32 rows: lane pair, then the waves ------------------------------------------------
  float r1 = s1[0] + s1[1], r2 = s2[0] + s2[1];
  r1 += __shfl_xor(r1, 32, 64);
  r2 += __shfl_xor(r2, 32, 64);
  if (lh == 0) *reinterpret_cast<f32x2*>(sRed + (wave * BR + lr) * 2) = f32x2{r1, r2};
  stamp(4);
  __syncthreads();                              // (also: every wave is done reading sQ -- it becomes the output tile)
  f32x2 t12 = {0.f, 0.f};
#pragma unroll
  for (int w = 0; w < NW; ++w) t12 += *reinterpret_cast<const f32x2*>(sRed + (w * BR + lr) * 2);
  const float mean = t12[0] * (1.0f / D_);
  const float rstd = rsqrtf(fmaxf(t12[1] * (1.0f / D_) - mean * mean, 0.f) + 1e-5f);
  const f32x2 rs2 = {rstd, rstd}, nm2 = {-mean * rstd, -mean * rstd};
  // ---- LN, modulation, SiLU; bf16 tile into LDS (same swizzle as the Q tile) -----------------------------------------
#pragma unroll
  for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
    for (int lb = 0; lb < NLB; ++lb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = (wave + NW * hh) * HD + 32 * lb + 8 * q + 4 * lh;
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(sPar + col), b4 = *reinterpret_cast<const f32x4*>(sPar + D_ + col);
        f32x2 o[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const f32x2 y = {acc[hh][lb][4 * q + 2 * e], acc[hh][lb][4 * q + 2 * e + 1]};
          const f32x2 u = __builtin_elementwise_fma(y, rs2, nm2);
          const f32x2 x = __builtin_elementwise_fma(u, f32x2{g4[2 * e], g4[2 * e + 1]}, f32x2{b4[2 * e], b4[2 * e + 1]});
          const f32x2 w = x * -LOG2E;
          const f32x2 den = f32x2{__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])} + 1.0f;
          o[e] = x * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
        }
        *reinterpret_cast<bf16x4*>(sQ + lr * ROWB + 16 * ((col >> 3) ^ (lr & 15)) + 2 * (col & 7)) =
            bf16x4{(__bf16)o[0][0], (__bf16)o[0][1], (__bf16)o[1][0], (__bf16)o[1][1]};
        if constexpr (YOUT)
          *reinterpret_cast<bf16x4*>(sH + lr * ROWB + 16 * ((col >> 3) ^ (lr & 15)) + 2 * (col & 7)) =
              bf16x4{(__bf16)acc[hh][lb][4 * q], (__bf16)acc[hh][lb][4 * q + 1], (__bf16)acc[hh][lb][4 * q + 2], (__bf16)acc[hh][lb][4 * q + 3]};
      }
  __syncthreads();
  stamp(5);
  constexpr int PPR = ROWB / 16;
  if constexpr (GEMM) {
    out_gemm_rows(sQ, sH, og, b, r0, T);
  } else {
    // ---- whole rows out --------------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < BR * PPR / NT; ++u) {
      const int idx = tid + NT * u;
      const int r = idx / PPR, p = idx % PPR;
      if (r0 + r < T) {
        *reinterpret_cast<bf16x8*>(Out + ((int64_t)b * T + r0 + r) * ldo + 8 * p) =
            *reinterpret_cast<const bf16x8*>(sQ + r * ROWB + 16 * (p ^ (r & 15)));
        if constexpr (YOUT)
          *reinterpret_cast<bf16x8*>(og.h + ((int64_t)b * T + r0 + r) * og.ldh + 8 * p) =
              *reinterpret_cast<const bf16x8*>(sH + r * ROWB + 16 * (p ^ (r & 15)));
      }
    }
  }
  stamp(6);
}


// The stylization block behind the FFN (and any other block whose input rows Y are already in memory) as ONE launch:
//     h[rows] += silu( LN(Y[rows]) (1 + scale) + shift ) . W^T + bias                              (transformer.py:81-86)
// = hig_ln_bf16 (stylization front) + the stylization-out GEMM: 32 whole rows per workgroup, Y tile / residual rows /
// LayerNorm and modulation vectors by DMA, a wave normalises four rows in registers (8 elements per lane) and writes them
// back in place, then out_gemm_rows.  grid = (ceil(T / 32), B), 512 threads.
__global__ __launch_bounds__(512) void rows_out16_kernel(const __bf16* __restrict__ Y, int64_t ldy, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ ss, int64_t ss_ld,
                                                         int shift_off, int T, const OutGemmArgs og) {
  constexpr int D_ = 512, ROWB = 1024, BR = 32, NW = 8, NT = 512, QBYTES = BR * ROWB;
  __shared__ __attribute__((aligned(1024))) char smem[2 * QBYTES + 4 * D_ * 4];
  char* const sA = smem;                                       // Y tile, then the activated tile (same swizzle as everywhere)
  char* const sH = smem + QBYTES;                              // residual rows
  float* const sPar = reinterpret_cast<float*>(smem + 2 * QBYTES);   // gamma | beta | scale | shift, then gamma' | beta'
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, r0 = blockIdx.x * BR;
#pragma unroll
  for (int q = 0; q < QBYTES / 1024 / NW; ++q) {
    const int n = wave + NW * q;                  // row n of the tile
    const int64_t row = (int64_t)b * T + min(r0 + n, T - 1);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Y + row * ldy + 8 * (lane ^ (n & 15))),
                                     (__attribute__((address_space(3))) void*)(sA + n * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(og.h + row * og.ldh + 8 * (lane ^ (n & 15))),
                                     (__attribute__((address_space(3))) void*)(sH + n * 1024), 16, 0, 0);
  }
  {
    const float* ssb = ss + (int64_t)b * ss_ld;
    const int vec = wave >> 1, part = wave & 1;   // 8 pieces of 1 KiB: vector wave / 2, half wave % 2
    const float* base = vec == 0 ? gamma : vec == 1 ? beta : vec == 2 ? ssb : ssb + shift_off;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + part * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(sPar) + wave * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = tid; c < D_; c += NT) {            // gamma' = gamma (1 + scale), beta' = beta (1 + scale) + shift
    const float g = sPar[c], be = sPar[D_ + c], sc = 1.0f + sPar[2 * D_ + c], sh = sPar[3 * D_ + c];
    sPar[c] = g * sc;
    sPar[D_ + c] = fmaf(be, sc, sh);
  }
  __syncthreads();
  {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(sPar + 8 * lane), g1 = *reinterpret_cast<const f32x4*>(sPar + 8 * lane + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sPar + D_ + 8 * lane), b1 = *reinterpret_cast<const f32x4*>(sPar + D_ + 8 * lane + 4);
    const float gp[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bp[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int k = 0; k < BR / NW; ++k) {
      const int r = wave + NW * k;
      char* rp = sA + r * ROWB + 16 * (lane ^ (r & 15));      // this lane's 8 elements of row r: columns 8 lane .. 8 lane + 7
      const u32x4 w = *reinterpret_cast<const u32x4*>(rp);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] = bf_lo(w[e]); v[2 * e + 1] = bf_hi(w[e]); }
      float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      const float mean = wave_sum(s) * (1.0f / D_);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float dlt = v[e] - mean; q = fmaf(dlt, dlt, q); }
      const float rstd = rsqrtf(wave_sum(q) * (1.0f / D_) + 1e-5f);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)hig_silu_fast(fmaf((v[e] - mean) * rstd, gp[e], bp[e]));
      *reinterpret_cast<bf16x8*>(rp) = o;
    }
  }
  __syncthreads();
  out_gemm_rows(sA, sH, og, b, r0, T);
}

// ---------------------------------------------------------------------------------------------------------------------
// Context build of the bf16-storage forward on the bf16 matrix cores:
//     A[b,h][c][l] = sum_r softmax_r(K)[r][c] V[r][l]          (transformer.py:112-116 / :148-152; rows r >= length masked)
// One workgroup per (sample, head) walks the rows in chunks of 64 with a running column maximum (online softmax), like
// ctx_mfma_kernel (linattn.hip) -- whose 32 fp32 MFMAs per wave and chunk (2 048 cycles) are four bf16 MFMAs here.
// Both operands of v_mfma_f32_32x32x16_bf16 need the ROW index r as their k: V^T and P^T.  Neither is ever transposed in
// memory: the V chunk lands row-major in LDS by DMA, P = exp(K - max) is written row-major by the threads that hold K,
// and ds_read_b64_tr_b16 (the transpose read of gfx950) hands each lane four consecutive rows of one column.
// Per chunk: column maxima (wave shuffles + one LDS exchange), P, two barriers, 4 MFMAs + 16 transpose reads per wave.
// ---------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// Output groups: head h belongs to group h / Hgrp (the layers of the batched text side, denoiser.hip); a group's A / kstat / At16
// are laid out (B, Hgrp, ...) at the group strides (elements).  Hgrp = H: one group, the plain (B, H, ...) layout.
struct Ctx16Groups {
  int Hgrp;
  int64_t a_gs, k_gs, at_gs;
};
template <int HD, int NB>
__global__ __launch_bounds__(256) void ctx16_mfma_kernel(const __bf16* __restrict__ K, const __bf16* __restrict__ V, int64_t ld,
                                                         int rows, int H, const int64_t* __restrict__ length,
                                                         float* __restrict__ A, float* __restrict__ kstat,
                                                         __bf16* __restrict__ At16, const Ctx16Groups grp) {
  static_assert(HD == 64 || HD == 128, "head dim 64 or 128");
  constexpr int CHK = 64;                       // rows per chunk
  constexpr int ROWB = HD * 2;                  // bytes per LDS row (128 / 256)
  // NB: K / V chunks in LDS, the DMA runs NB - 1 chunks ahead of the arithmetic.  Head dim 64: three buffers = 65 KB of LDS, two
  // workgroups per CU (four = 83 KB = one, and the B x H = 512 workgroups of a B = 64 launch then ran as two rounds: B = 64
  // forward -3 %); two buffers = 49 KB, three per CU, for launches of many rounds.  Head dim 128: 129 KB, one per CU.
  static_assert(NB == 2 || NB == 3, "ring of two or three chunks");
  constexpr int LPR = HD / 4;                   // lanes per row when a lane holds 4 channels (16 / 32)
  constexpr int RPP = 64 / LPR;                 // rows a wave covers per pass (4 / 2)
  constexpr int NI = 16 / RPP;                  // passes: a wave holds rows 16 wave .. 16 wave + 15 of the chunk (4 / 8)
  constexpr int DPO = CHK * ROWB / 1024 / 4;    // DMA instructions per wave, operand and chunk (2 / 4)
  constexpr int CPR = ROWB / 16;                // 16-byte chunks per row (8 / 16)
  constexpr int NBW = HD / 64;                  // 32 x 32 blocks of A per wave and dimension (1 / 2)
  __shared__ __attribute__((aligned(1024))) char sV[NB][CHK * ROWB];  // [r][l] bf16, 16-byte chunk c of row r at c ^ f(r)
  __shared__ __attribute__((aligned(1024))) char sK[NB][CHK * ROWB];  // [r][c] bf16, same layout (K arrives by DMA too: no VGPR load
                                                                      // for hipcc to guard with a vmcnt(0) while a DMA is in flight)
  __shared__ __attribute__((aligned(1024))) char sP[2][CHK * ROWB];   // [r][c] bf16 = exp(K - running max), same layout
  __shared__ float sWmax[4][HD];                // per-wave column maxima of the chunk; at the end per-wave column sums
  __shared__ float sScale[HD];                  // exp(m_old - m_new) of the chunk; at the end 1 / column sum
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int64_t oidx = (int64_t)b * grp.Hgrp + h % grp.Hgrp;   // index inside the group's (B, Hgrp) outputs
  A += (h / grp.Hgrp) * grp.a_gs;
  kstat += (h / grp.Hgrp) * grp.k_gs;
  if (At16) At16 += (h / grp.Hgrp) * grp.at_gs;
  int len = rows;
  if (length) len = (int)min<int64_t>(max<int64_t>(length[b], 0), rows);
  const __bf16* Kb = K + (int64_t)b * rows * ld + h * HD;
  const __bf16* Vb = V + (int64_t)b * rows * ld + h * HD;
  auto fsw = [](int r) { return ((r >> 1) & 1) << 2; };   // swizzle: rows r, r + 2 of a transpose read use disjoint bank halves
  // K: thread (wave, rl = lane / LPR, c4 = lane % LPR) holds channels 4 c4 .. 4 c4 + 3 of rows 16 wave + RPP i + rl, i = 0 .. NI - 1
  const int rl = lane / LPR, c4 = lane % LPR;
  float kreg[NI][4];
  auto read_k = [&](int r0, int buf) {          // this thread's K values of the chunk, from LDS (rows beyond len: -inf)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rr = 16 * wave + RPP * i + rl;
      const u32x2 w = *reinterpret_cast<const u32x2*>(sK[buf] + rr * ROWB + 16 * ((c4 >> 1) ^ fsw(rr)) + 8 * (c4 & 1));
      const bool ok = r0 + rr < len;
      kreg[i][0] = ok ? bf_lo(w.x) : -INFINITY; kreg[i][1] = ok ? bf_hi(w.x) : -INFINITY;
      kreg[i][2] = ok ? bf_lo(w.y) : -INFINITY; kreg[i][3] = ok ? bf_hi(w.y) : -INFINITY;
    }
  };
  auto dma_chunk = [&](int r0, int buf) {   // K and V rows [r0, r0 + 64) x ROWB bytes: 2 x DPO instructions per wave (rows beyond len:
                                            // any valid row -- they are masked / multiplied by P = 0)
#pragma unroll
    for (int q = 0; q < DPO; ++q) {
      const int n = wave + 4 * q;
      const int row = (1024 / ROWB) * n + lane / CPR, pos = lane % CPR;
      const int64_t off = (int64_t)min(r0 + row, max(len - 1, 0)) * ld + 8 * (pos ^ fsw(row));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kb + off),
                                       (__attribute__((address_space(3))) void*)(sK[buf] + n * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vb + off),
                                       (__attribute__((address_space(3))) void*)(sV[buf] + n * 1024), 16, 0, 0);
    }
  };
  // MFMA roles: wave (wi, wj) owns the NBW x NBW blocks A[32 (NBW wi + bi) ..][32 (NBW wj + bj) ..]; first operand V^T (rows
  // l), second P^T (rows c): accumulator element 4 q + e of lane (lr, lh) is A[c = cbase + lr][l = lbase + 8 q + 4 lh + e]
  const int wi = wave >> 1, wj = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int gi = lane & 15, gg = lane >> 4;     // transpose read: lane gi of 16-lane group gg
  auto tr_addr = [&](int colbase, int rr) {     // byte offset of (row rr, columns colbase + 16 (gg & 1) + 4 (gi & 3) ..) in a chunk image
    const int col = colbase + 16 * (gg & 1) + 4 * (gi & 3);
    return rr * ROWB + 16 * ((col >> 3) ^ fsw(rr)) + 2 * (col & 7);
  };
  f32x16 acc[NBW][NBW];
#pragma unroll
  for (int bi = 0; bi < NBW; ++bi)
#pragma unroll
    for (int bj = 0; bj < NBW; ++bj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[bi][bj][e] = 0.f;
  float mrun[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // running maximum of this thread's 4 channels
  float ksum[4] = {0.f, 0.f, 0.f, 0.f};         // this thread's share of sum_r exp(K - m)
  const int nchunk = (len + CHK - 1) / CHK;
  for (int t = 0; t < NB - 1 && t < nchunk; ++t) dma_chunk(t * CHK, t);
  for (int r0 = 0, it = 0; r0 < len; r0 += CHK, ++it) {
    const int buf = it & 1, kb = it % NB;
    // chunk `it` has landed once only the younger chunks' requests (2 DPO per wave and chunk) are outstanding
    {
      const int younger = min(NB - 2, nchunk - 1 - it);
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DPO) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPO) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();               // #0: everyone's share of the chunk has landed
    asm volatile("" ::: "memory");
    read_k(r0, kb);
    // ---- column maxima of the chunk ----
    float m4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float m = kreg[0][c];
#pragma unroll
      for (int i = 1; i < NI; ++i) m = fmaxf(m, kreg[i][c]);
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      m4[c] = m;
    }
    if (rl == 0) *reinterpret_cast<f32x4*>(&sWmax[wave][4 * c4]) = f32x4{m4[0], m4[1], m4[2], m4[3]};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // #1
    asm volatile("" ::: "memory");
    float sc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float m = mrun[c];
#pragma unroll
      for (int w = 0; w < 4; ++w) m = fmaxf(m, sWmax[w][4 * c4 + c]);
      sc[c] = mrun[c] == -INFINITY ? 0.f : __expf(mrun[c] - m);   // (nothing accumulated yet while the old maximum is -inf)
      mrun[c] = m;
      ksum[c] *= sc[c];
    }
    if (wave == 0 && rl == 0) *reinterpret_cast<f32x4*>(&sScale[4 * c4]) = f32x4{sc[0], sc[1], sc[2], sc[3]};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rr = 16 * wave + RPP * i + rl;
      float pe[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        pe[c] = (r0 + rr < len) ? __expf(kreg[i][c] - mrun[c]) : 0.f;
        ksum[c] += pe[c];
      }
      *reinterpret_cast<bf16x4*>(sP[buf] + rr * ROWB + 16 * ((c4 >> 1) ^ fsw(rr)) + 8 * (c4 & 1)) =
          bf16x4{(__bf16)pe[0], (__bf16)pe[1], (__bf16)pe[2], (__bf16)pe[3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // P is written
    __builtin_amdgcn_s_barrier();               // #2
    asm volatile("" ::: "memory");
    if (it + NB - 1 < nchunk) dma_chunk((it + NB - 1) * CHK, (it + NB - 1) % NB);   // into the buffer of chunk it - 1: everyone is past it
    // ---- rescale the accumulator rows (channel c = cbase + lr), then add this chunk: 4 k-steps of 16 rows ----
#pragma unroll
    for (int bi = 0; bi < NBW; ++bi) {
      const float f = sScale[32 * (NBW * wi + bi) + lr];
#pragma unroll
      for (int bj = 0; bj < NBW; ++bj)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[bi][bj][e] *= f;
    }
#pragma unroll
    for (int ks = 0; ks < CHK / 16; ++ks) {
      if (r0 + 16 * ks >= len) break;           // (rows beyond the length have P = 0: the short tail chunk of T = 196 is one k-step)
      s16x8 vf[NBW], pf[NBW];
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        const int rr = 16 * ks + 8 * (gg >> 1) + 4 * part + (gi >> 2);
#pragma unroll
        for (int bb = 0; bb < NBW; ++bb) {
          const s16x4 v4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV[kb] + tr_addr(32 * (NBW * wj + bb), rr)));
          const s16x4 p4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sP[buf] + tr_addr(32 * (NBW * wi + bb), rr)));
#pragma unroll
          for (int e = 0; e < 4; ++e) { vf[bb][4 * part + e] = v4[e]; pf[bb][4 * part + e] = p4[e]; }
        }
      }
#pragma unroll
      for (int bi = 0; bi < NBW; ++bi)
#pragma unroll
        for (int bj = 0; bj < NBW; ++bj)
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[bj]), __builtin_bit_cast(bf16x8, pf[bi]), acc[bi][bj], 0, 0, 0);
    }
  }
  // ---- column sums, normalisation, outputs ----
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float t4[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float t = ksum[c];
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) t += __shfl_xor(t, off, 64);
    t4[c] = t;
  }
  if (rl == 0) *reinterpret_cast<f32x4*>(&sWmax[wave][4 * c4]) = f32x4{t4[0], t4[1], t4[2], t4[3]};
  __syncthreads();
  if (tid < HD) {
    const float t = sWmax[0][tid] + sWmax[1][tid] + sWmax[2][tid] + sWmax[3][tid];
    sScale[tid] = t > 0.f ? 1.0f / t : 0.f;
  }
  if (wave == 0 && rl == 0) {                   // (this thread's running maxima are the final ones of its 4 channels)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float t = sWmax[0][4 * c4 + c] + sWmax[1][4 * c4 + c] + sWmax[2][4 * c4 + c] + sWmax[3][4 * c4 + c];
      float* st = kstat + (oidx * HD + 4 * c4 + c) * 2;
      st[0] = len > 0 ? mrun[c] : 0.f;
      st[1] = len > 0 ? t : 1.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int bi = 0; bi < NBW; ++bi) {
    const int cc = 32 * (NBW * wi + bi) + lr;
    const float inv = sScale[cc];
#pragma unroll
    for (int bj = 0; bj < NBW; ++bj) {
      const int lbase = 32 * (NBW * wj + bj);
      float* ap = A + oidx * HD * HD + cc * HD + lbase + 4 * lh;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(ap + 8 * q) = f32x4{acc[bi][bj][4 * q] * inv, acc[bi][bj][4 * q + 1] * inv, acc[bi][bj][4 * q + 2] * inv, acc[bi][bj][4 * q + 3] * inv};
      if (At16) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int l = lbase + 8 * (e >> 2) + 4 * lh + (e & 3);
          At16[oidx * HD * HD + hig_at16_offset(HD, l, cc)] = (__bf16)(acc[bi][bj][e] * inv);
        }
      }
    }
  }
}

}  // namespace

// Out = silu( LN( softmax_hd(Q) . A ) * (1 + scale) + shift ), bf16 matrix products (see the kernel).  Q, Out bf16 (B * rows,
// H * hd); At16 bf16 (B, H, hd, hd) [l][c] = the TRANSPOSED context matrices (hig_linattn_ctx_bf16); gamma, beta, ss fp32; scale = ss[b][0 .. d), shift = ss[b][shift_off .. + d).
// Head dim 64 with 4 or 8 heads (other shapes: hig_linattn_apply_sty_bf16 / the unfused pair).
unsigned long long* g_ap_stamps = nullptr;
// Diagnostic: thread 0 of every workgroup of apply_sty16_kernel writes s_memtime stamps to buf[workgroup * 8 + k] (k: 0 start,
// 1 Q tile landed, 2 softmax done, 3 context matrices landed, 4 products + row sums done, 5 output tile in LDS, 6 end).
// NULL switches it off.  Never part of a timed run (tools/apply16_stamps.py).
extern "C" int hig_linattn16_debug_stamps(void* buf) {
  g_ap_stamps = static_cast<unsigned long long*>(buf);
  return HIG_OK;
}

// Yout (nullable): y = softmax(q) . A as bf16 rows as well (the training forward keeps it for the backward of the LayerNorm)
extern "C" int hig_linattn_apply_sty_mm16_y(const void* Q, int64_t ldq, const void* At16, const float* gamma, const float* beta,
                                            const float* ss, int64_t ss_ld, int32_t ss_shift_off, void* Out, int64_t ldo, void* Yout,
                                            int64_t ldy, int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream) {
  HIG_REQUIRE(Q && At16 && gamma && beta && ss && Out && B > 0 && rows > 0, "hig_linattn_apply_sty_mm16: bad arguments");
  if ((hd != 64 && hd != 128) || (H != 4 && H != 8))
    return hig_set_error(HIG_EUNSUPPORTED, "hig_linattn_apply_sty_mm16: built for head dim 64 / 128 and 4 or 8 heads (got %d, %d)", hd, H);
  HIG_REQUIRE(ldq % 8 == 0 && ldo % 8 == 0 && ss_ld % 4 == 0 && ss_shift_off % 4 == 0 && (!Yout || ldy % 8 == 0) &&
                  ((reinterpret_cast<uintptr_t>(Q) & 15) | (reinterpret_cast<uintptr_t>(Out) & 15) | (reinterpret_cast<uintptr_t>(Yout) & 15) |
                   (reinterpret_cast<uintptr_t>(At16) & 15) | (reinterpret_cast<uintptr_t>(gamma) & 15) |
                   (reinterpret_cast<uintptr_t>(beta) & 15) | (reinterpret_cast<uintptr_t>(ss) & 15)) == 0,
              "hig_linattn_apply_sty_mm16: alignment");
  const dim3 grid((rows + 31) / 32, B);
  hipStream_t st = hig_stream(stream);
  const OutGemmArgs og{nullptr, nullptr, static_cast<__bf16*>(Yout), ldy, nullptr};
#define HIG_AP16(HD_, H_, NW_)                                                                                                        \
  do {                                                                                                                                \
    if (Yout)                                                                                                                         \
      hipLaunchKernelGGL((apply_sty16_kernel<HD_, H_, NW_, false, true>), grid, dim3(64 * NW_), 0, st, static_cast<const __bf16*>(Q), \
                         ldq, static_cast<const __bf16*>(At16), gamma, beta, ss, ss_ld, ss_shift_off, static_cast<__bf16*>(Out), ldo, \
                         rows, g_ap_stamps, og);                                                                                      \
    else                                                                                                                              \
      hipLaunchKernelGGL((apply_sty16_kernel<HD_, H_, NW_>), grid, dim3(64 * NW_), 0, st, static_cast<const __bf16*>(Q), ldq,         \
                         static_cast<const __bf16*>(At16), gamma, beta, ss, ss_ld, ss_shift_off, static_cast<__bf16*>(Out), ldo,      \
                         rows, g_ap_stamps, og);                                                                                      \
  } while (0)
  if (hd == 64 && H == 8) HIG_AP16(64, 8, 8);
  else if (hd == 64) HIG_AP16(64, 4, 4);
  else if (H == 8) HIG_AP16(128, 8, 8);
  else HIG_AP16(128, 4, 4);
#undef HIG_AP16
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
extern "C" int hig_linattn_apply_sty_mm16(const void* Q, int64_t ldq, const void* At16, const float* gamma, const float* beta,
                                          const float* ss, int64_t ss_ld, int32_t ss_shift_off, void* Out, int64_t ldo,
                                          int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream) {
  return hig_linattn_apply_sty_mm16_y(Q, ldq, At16, gamma, beta, ss, ss_ld, ss_shift_off, Out, ldo, nullptr, 0, B, rows, H, hd, stream);
}

namespace {
// Y_frag[((j / 32) (R / 16) + r / 16) 512 + (j % 32 + 32 ((r % 16) / 8)) 8 + (r % 8)] = Y[j][r]: one thread per 16-byte piece
__global__ void weight_frag16_kernel(const __bf16* __restrict__ Y, int64_t ldy, int J, int R, __bf16* __restrict__ out) {
  const int64_t n = (int64_t)J * R / 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const int64_t blk = i >> 6;
    const int ks = (int)(blk % (R / 16)), jb = (int)(blk / (R / 16));
    const int j = 32 * jb + (lane & 31), r = 16 * ks + 8 * (lane >> 5);
    reinterpret_cast<bf16x8*>(out)[i] = *reinterpret_cast<const bf16x8*>(Y + (int64_t)j * ldy + r);
  }
}
}  // namespace

extern "C" int hig_weight_frag16(const void* Y, int64_t ldy, int32_t J, int32_t R, void* Y_frag, hig_stream_t stream) {
  HIG_REQUIRE(Y && Y_frag && J > 0 && R > 0, "hig_weight_frag16: bad arguments");
  HIG_REQUIRE(J % 32 == 0 && R % 16 == 0 && ldy % 8 == 0 && ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(Y_frag)) & 15) == 0,
              "hig_weight_frag16: J %% 32, R %% 16, 16-byte aligned rows");
  const int64_t n = (int64_t)J * R / 8;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(weight_frag16_kernel, dim3((unsigned)blocks), dim3(256), 0, hig_stream(stream), static_cast<const __bf16*>(Y), ldy, J, R,
                     static_cast<__bf16*>(Y_frag));
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// hig_linattn_apply_sty_mm16 followed by the stylization block's output projection and residual update, as one kernel:
//     h[rows] += silu( LN( softmax_hd(Q) . A ) * (1 + scale) + shift ) . W^T + bias          (transformer.py:111-118 then :81-86)
// W_frag: the (d, d) weight in operand order (hig_weight_frag16); h bf16 (B * rows, d), updated in place; stats (nullable):
// (sum, centred sum of squares) of the new rows per 128-column panel, [B * rows][4][2] fp32, for a LayerNorm-folding consumer
// (hig_gemm16_desc.row_stats_in).  d = 512 with 8 heads of 64.
extern "C" int hig_attn_out16(const void* Q, int64_t ldq, const void* At16, const float* gamma, const float* beta, const float* ss,
                              int64_t ss_ld, int32_t ss_shift_off, const void* W_frag, const float* bias, void* h, int64_t ldh,
                              float* stats, int32_t B, int32_t rows, int32_t H, int32_t hd, hig_stream_t stream) {
  HIG_REQUIRE(Q && At16 && gamma && beta && ss && W_frag && bias && h && B > 0 && rows > 0, "hig_attn_out16: bad arguments");
  if (hd != 64 || H != 8) return hig_set_error(HIG_EUNSUPPORTED, "hig_attn_out16: built for 8 heads of 64 (got %d x %d)", H, hd);
  HIG_REQUIRE(ldq % 8 == 0 && ldh % 8 == 0 && ss_ld % 4 == 0 && ss_shift_off % 4 == 0 &&
                  ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(At16) |
                    reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(ss) |
                    reinterpret_cast<uintptr_t>(W_frag) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(stats) & 7) == 0,
              "hig_attn_out16: alignment");
  const dim3 grid((rows + 31) / 32, B);
  hipLaunchKernelGGL((apply_sty16_kernel<64, 8, 8, true>), grid, dim3(512), 0, hig_stream(stream), static_cast<const __bf16*>(Q), ldq,
                     static_cast<const __bf16*>(At16), gamma, beta, ss, ss_ld, ss_shift_off, static_cast<__bf16*>(nullptr), (int64_t)0, rows,
                     g_ap_stamps, OutGemmArgs{static_cast<const __bf16*>(W_frag), bias, static_cast<__bf16*>(h), ldh, stats});
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// hig_ln_bf16 (stylization front) followed by the stylization-out projection and the residual update, as one kernel:
//     h[rows] += silu( LN(Y[rows]) (1 + scale) + shift ) . W^T + bias                               (transformer.py:81-86)
// Y, h bf16 (B * rows, 512); W_frag / stats as in hig_attn_out16.
extern "C" int hig_rows_out16(const void* Y, int64_t ldy, const float* gamma, const float* beta, const float* ss, int64_t ss_ld,
                              int32_t ss_shift_off, const void* W_frag, const float* bias, void* h, int64_t ldh, float* stats,
                              int32_t B, int32_t rows, int32_t d, hig_stream_t stream) {
  HIG_REQUIRE(Y && gamma && beta && ss && W_frag && bias && h && B > 0 && rows > 0, "hig_rows_out16: bad arguments");
  if (d != 512) return hig_set_error(HIG_EUNSUPPORTED, "hig_rows_out16: built for d = 512 (got %d)", d);
  HIG_REQUIRE(ldy % 8 == 0 && ldh % 8 == 0 && ss_ld % 4 == 0 && ss_shift_off % 4 == 0 &&
                  ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(gamma) |
                    reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(ss) | reinterpret_cast<uintptr_t>(W_frag) |
                    reinterpret_cast<uintptr_t>(bias)) & 15) == 0 && (reinterpret_cast<uintptr_t>(stats) & 7) == 0,
              "hig_rows_out16: alignment");
  hipLaunchKernelGGL(rows_out16_kernel, dim3((rows + 31) / 32, B), dim3(512), 0, hig_stream(stream), static_cast<const __bf16*>(Y), ldy,
                     gamma, beta, ss, ss_ld, ss_shift_off, rows,
                     OutGemmArgs{static_cast<const __bf16*>(W_frag), bias, static_cast<__bf16*>(h), ldh, stats});
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// A (fp32, [c][l]) + kstat (column max, column sum) + At16 (nullable: transposed bf16 copy) of the linear-attention context,
// K / V bf16, on the bf16 matrix cores (ctx16_mfma_kernel): softmax_r(K) and V are rounded to bf16 for the product, fp32
// accumulate and statistics.  Head dim 64; one workgroup per (sample, head).
extern "C" int hig_linattn_ctx_mm16(const void* K, const void* V, int64_t ld, int32_t B, int32_t rows, int32_t H, int32_t hd,
                                    const int64_t* length, float* A, float* kstat, void* At16, hig_stream_t stream) {
  HIG_REQUIRE(K && V && A && kstat && B > 0 && rows > 0 && H > 0, "hig_linattn_ctx_mm16: bad arguments");
  if (hd != 64 && hd != 128) return hig_set_error(HIG_EUNSUPPORTED, "hig_linattn_ctx_mm16: built for head dim 64 / 128 (got %d)", hd);
  HIG_REQUIRE(ld % 8 == 0 && (reinterpret_cast<uintptr_t>(K) & 15) == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(A) & 15) == 0,
              "hig_linattn_ctx_mm16: K / V / A must be 16-byte aligned with ld %% 8 == 0");
  // (same-call A/B, forward: B = 64 1.546 -> 1.535 ms, B = 128 2.451 -> 2.420, B = 512 8.45 -> 8.31 with two buffers)
  constexpr int nb2_from = 512;   // (a former tuning knob, fixed at the value that won its A/B): workgroups from which the ring has two buffers (0 = never)
  if (hd == 64 && nb2_from > 0 && B * H >= nb2_from)
    hipLaunchKernelGGL((ctx16_mfma_kernel<64, 2>), dim3(B * H), dim3(256), 0, hig_stream(stream), static_cast<const __bf16*>(K),
                       static_cast<const __bf16*>(V), ld, rows, H, length, A, kstat, static_cast<__bf16*>(At16), Ctx16Groups{H, 0, 0, 0});
  else if (hd == 64)
    hipLaunchKernelGGL((ctx16_mfma_kernel<64, 3>), dim3(B * H), dim3(256), 0, hig_stream(stream), static_cast<const __bf16*>(K),
                       static_cast<const __bf16*>(V), ld, rows, H, length, A, kstat, static_cast<__bf16*>(At16), Ctx16Groups{H, 0, 0, 0});
  else
    hipLaunchKernelGGL((ctx16_mfma_kernel<128, 3>), dim3(B * H), dim3(256), 0, hig_stream(stream), static_cast<const __bf16*>(K),
                       static_cast<const __bf16*>(V), ld, rows, H, length, A, kstat, static_cast<__bf16*>(At16), Ctx16Groups{H, 0, 0, 0});
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// The same for G groups of H heads in ONE launch (the L layers of the batched text side): K / V hold G H heads side by side, group
// g's outputs go to A + g a_gs, kstat + g k_gs, At16 + g at_gs in the plain (B, H, ...) layout.  No length mask.
int hig_linattn_ctx16_groups(const void* K, const void* V, int64_t ld, int32_t B, int32_t rows, int32_t H, int32_t G, int32_t hd,
                             float* A, int64_t a_gs, float* kstat, int64_t k_gs, void* At16, int64_t at_gs, hipStream_t st) {
  if (!(hd == 64 || hd == 128) || B <= 0 || rows <= 0 || H <= 0 || G <= 0 || ld % 8 != 0) return 1;
  const Ctx16Groups grp{H, a_gs, k_gs, at_gs};
  const int n = B * H * G;
  if (hd == 64)
    hipLaunchKernelGGL((ctx16_mfma_kernel<64, 2>), dim3(n), dim3(256), 0, st, static_cast<const __bf16*>(K), static_cast<const __bf16*>(V),
                       ld, rows, H * G, nullptr, A, kstat, static_cast<__bf16*>(At16), grp);
  else
    hipLaunchKernelGGL((ctx16_mfma_kernel<128, 3>), dim3(n), dim3(256), 0, st, static_cast<const __bf16*>(K), static_cast<const __bf16*>(V),
                       ld, rows, H * G, nullptr, A, kstat, static_cast<__bf16*>(At16), grp);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
