// Weight gradients of the bf16-storage training step on the bf16 matrix cores, WITHOUT transposed operand copies:
//     dW[j][k] = sum_i dC[i][j] * act[i][k]        dbias[j] = sum_i dC[i][j]
// (autograd of nn.Linear over the M = B T rows; reference: the torch autograd of codes/models/transformer.py:81-85,108-114,
// 144-150,168 inside DDPMTrainer.backward_G, codes/trainers/ddpm_trainer.py:172-178).  The reduce index is the ROW of both
// operands, so both MFMA operands need "k = row": neither dC^T nor act^T is ever built in memory.  Row chunks of dC
// (64 rows x 128 columns of j) and of act (64 rows x 128 columns of k) land row-major in LDS by DMA
// (global_load_lds_dwordx4, XOR swizzle on source and read address), and ds_read_b64_tr_b16 -- the transpose read of gfx950 --
// hands each lane four consecutive rows of one column: the operand shape of v_mfma_f32_32x32x16_bf16.  Same scheme as
// ctx16_mfma_kernel (linattn16.hip), which is this contraction with a column softmax in front of it.
//
// CDNA4 mapping.  256 threads = 4 waves (2 x 2), a workgroup owns a 128 x 128 tile of dW for ONE slice of the rows
// (split-R: few output tiles, M = 12 544 rows -- tiles x splits fills the two resident workgroups per CU), each wave 2 x 2
// blocks of 32 x 32 fp32 accumulators that live across the slice's chunks.  DMA ring of two chunks (64 KB of LDS, two
// workgroups per CU).  Per 16-row k-step and wave: 8 transpose reads, 4 MFMAs.  The partial tiles go to fp32 slabs
// [split][J x K | J], summed in split order by hig_reduce_slabs2 (deterministic, no float atomics).  The bias gradient rides
// along: the waves that hold the dC fragments of a tile column sum them with v_dot2_f32_bf16 (x . (1, 1)).
#include <stdlib.h>

#include "gemm16_epi.h"
#include "hig_host.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));

struct Wg16Args {
  const __bf16* dC; int64_t ldd;     // (rows, J)
  const __bf16* X; int64_t ldx;      // (rows, K)
  float* out;                        // splits == 1: dW (J, K) dense; else slabs: [split][J * K + J]
  float* dbias;                      // splits == 1: dbias (J) or null; else unused (the sums sit behind each slab)
  int64_t slab;                      // floats per split (J * K + J), 0 when splits == 1
  int J, K, rows, rows_per_split, ntk, ntiles, want_bias;
};

// NSG sub-groups of four waves per workgroup (NSG = 2: 512 threads, one workgroup per CU -- the same eight waves per CU as two
// 4-wave workgroups): sub-group g walks the row chunks g, g + NSG, ... of the workgroup's slice with its own DMA ring and its
// own accumulators, and the sub-groups' tiles are summed through LDS before ONE partial tile leaves.  The partial tiles are
// what a small weight gradient costs: tiles x splits x 64 KB = 32 MB written and read back per launch at two workgroups per
// CU -- more than the operands (25.7 MB for a 512 x 512 weight) -- and NSG = 2 halves them.
template <int NB, int CHK = 64, int NSG = 1>
__global__ __launch_bounds__(256 * NSG, (NSG > 1 ? 2 : (NB * CHK <= 128 ? 2 : 1))) void wgrad16_kernel(const Wg16Args a) {
  constexpr int ROWB = 256, CPR = 16, DPO = CHK / 16, NBW = 2;   // DPO: DMA instructions per wave, operand and chunk
  static_assert(CHK == 32 || CHK == 64, "row chunks of 32 or 64");
  static_assert(NSG == 1 || NSG * NB * CHK * ROWB * 2 >= 4 * NBW * NBW * 16 * 64 * 4, "the rings double as the reduction buffer");
  __shared__ __attribute__((aligned(1024))) char sDall[NSG][NB][CHK * ROWB];   // dC chunk [r][j]  bf16, 16-byte chunk c of row r at c ^ f(r)
  __shared__ __attribute__((aligned(1024))) char sXall[NSG][NB][CHK * ROWB];   // act chunk [r][k]
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                   // wave inside its sub-group
  const int sg = NSG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
  char (*sD)[CHK * ROWB] = sDall[sg];
  char (*sX)[CHK * ROWB] = sXall[sg];
  const int split = blockIdx.x / a.ntiles, tile = blockIdx.x - split * a.ntiles;
  const int tj = tile / a.ntk, tk = tile - tj * a.ntk;
  const int j0 = tj * 128, k0 = tk * 128;
  const int rbeg = split * a.rows_per_split;
  const int len = min(a.rows - rbeg, a.rows_per_split);          // rows of this slice (> 0 by construction)
  const __bf16* Db = a.dC + (int64_t)rbeg * a.ldd + j0;
  const __bf16* Xb = a.X + (int64_t)rbeg * a.ldx + k0;
  // valid 16-byte chunks of a row inside this tile's column window (J, K multiples of 8): the DMA clamps to the last one,
  // the columns beyond hold copies whose outputs are never stored
  const int ncj = min(16, (a.J - j0) / 8), nck = min(16, (a.K - k0) / 8);
  auto fsw = [](int r) { return ((r >> 1) & 1) << 2; };
  auto dma_chunk = [&](int r0, int buf) {
#pragma unroll
    for (int q = 0; q < DPO; ++q) {
      const int n = wave + 4 * q;
      const int row = 4 * n + lane / CPR, pos = lane % CPR;
      const int src = pos ^ fsw(row);
      const int64_t r = min(r0 + row, len - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Db + r * a.ldd + 8 * min(src, ncj - 1)),
                                       (__attribute__((address_space(3))) void*)(sD[buf] + n * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xb + r * a.ldx + 8 * min(src, nck - 1)),
                                       (__attribute__((address_space(3))) void*)(sX[buf] + n * 1024), 16, 0, 0);
    }
  };
  const int wi = wave >> 1, wj = wave & 1, lr = lane & 31, lh = lane >> 5;
  const int gi = lane & 15, gg = lane >> 4;
  auto tr_addr = [&](int colbase, int rr) {
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
  float cs[NBW] = {0.f, 0.f};                       // column sums of dC: this lane's 8 rows of column 32 (NBW wi + bi) + lr per k-step
  const bool do_bias = a.want_bias && tk == 0 && wj == 0;
  const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
  const int nchunk_all = (len + CHK - 1) / CHK;                 // chunks of the slice; this sub-group's: sg, sg + NSG, ...
  const int nchunk = (nchunk_all - sg + NSG - 1) / NSG;
  const int nround = (nchunk_all + NSG - 1) / NSG;               // loop trips (uniform over the workgroup: barriers inside)
  for (int t = 0; t < NB - 1 && t < nchunk; ++t) dma_chunk((t * NSG + sg) * CHK, t);
  for (int it = 0; it < nround; ++it) {
    const int r0 = (it * NSG + sg) * CHK, buf = it % NB;
    const bool mine = it < nchunk;                               // (wave-uniform)
    {   // chunk `it` has landed once only the younger chunks' requests (2 DPO per wave and chunk) are outstanding
      const int younger = min(NB - 2, nchunk - 1 - it);
      static_assert(NB <= 6, "ring depth");
      if (younger >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * DPO) : "memory");
      else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * DPO) : "memory");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DPO) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPO) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();               // everyone's share of the chunk has landed; chunk it - 1's buffer is free
    asm volatile("" ::: "memory");
    if (it + NB - 1 < nchunk) dma_chunk(((it + NB - 1) * NSG + sg) * CHK, (it + NB - 1) % NB);
    if (it == nround - 1 && len % CHK != 0) {   // (uniform over the workgroup)
      // last, partly filled chunk: rows beyond the slice hold copies of its last row -- zero them in the dC image (one zero
      // operand is enough), whole 16-byte pieces, then publish
      if (mine && r0 + CHK > len) {
        const int first = len - r0;             // 1 .. CHK - 1
        for (int idx = tid; idx < (CHK - first) * CPR; idx += 256)
          *reinterpret_cast<wu32x4*>(sD[buf] + (first + idx / CPR) * ROWB + 16 * (idx % CPR)) = wu32x4{0u, 0u, 0u, 0u};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (!mine) continue;
#pragma unroll
    for (int ks = 0; ks < CHK / 16; ++ks) {
      if (r0 + 16 * ks >= len) break;
      s16x8 xf[NBW], df[NBW];
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        const int rr = 16 * ks + 8 * (gg >> 1) + 4 * part + (gi >> 2);
#pragma unroll
        for (int bb = 0; bb < NBW; ++bb) {
          const s16x4 x4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX[buf] + tr_addr(32 * (NBW * wj + bb), rr)));
          const s16x4 d4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sD[buf] + tr_addr(32 * (NBW * wi + bb), rr)));
#pragma unroll
          for (int e = 0; e < 4; ++e) { xf[bb][4 * part + e] = x4[e]; df[bb][4 * part + e] = d4[e]; }
        }
      }
      if (do_bias) {
#pragma unroll
        for (int bb = 0; bb < NBW; ++bb) {
          const bf16x8 dv = __builtin_bit_cast(bf16x8, df[bb]);
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[bb] = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{dv[2 * e], dv[2 * e + 1]}, ones, cs[bb], false);
        }
      }
#pragma unroll
      for (int bi = 0; bi < NBW; ++bi)
#pragma unroll
        for (int bj = 0; bj < NBW; ++bj)
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xf[bj]), __builtin_bit_cast(bf16x8, df[bi]), acc[bi][bj], 0, 0, 0);
    }
  }
  if constexpr (NSG > 1) {
    // ---- the sub-groups' tiles summed through LDS (the rings are idle: every DMA was waited for, the barrier ends every read) ----
    static_assert(NSG <= 2, "reduction written for two sub-groups");
    float* red = reinterpret_cast<float*>(&sDall[0][0][0]);      // [wave][bi][bj][16][64 lanes] fp32 = 64 KB, then [4][2][64] column sums
    float* redb = reinterpret_cast<float*>(&sXall[0][0][0]);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (sg == 1) {
#pragma unroll
      for (int bi = 0; bi < NBW; ++bi) {
#pragma unroll
        for (int bj = 0; bj < NBW; ++bj)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[(((wave * NBW + bi) * NBW + bj) * 16 + e) * 64 + lane] = acc[bi][bj][e];
        redb[(wave * NBW + bi) * 64 + lane] = cs[bi];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (sg != 0) return;
#pragma unroll
    for (int bi = 0; bi < NBW; ++bi) {
#pragma unroll
      for (int bj = 0; bj < NBW; ++bj)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[bi][bj][e] += red[(((wave * NBW + bi) * NBW + bj) * 16 + e) * 64 + lane];
      cs[bi] += redb[(wave * NBW + bi) * 64 + lane];
    }
  }
  // ---- partial tile out: accumulator element 4 q + e of lane (lr, lh) is dW[j = jb + lr][k = kb + 8 q + 4 lh + e] ----
  float* outp = a.out + (int64_t)split * a.slab;
#pragma unroll
  for (int bi = 0; bi < NBW; ++bi) {
    const int j = j0 + 32 * (NBW * wi + bi) + lr;
    if (j < a.J) {
#pragma unroll
      for (int bj = 0; bj < NBW; ++bj) {
        const int kb = k0 + 32 * (NBW * wj + bj) + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (kb + 8 * q < a.K)
            *reinterpret_cast<f32x4*>(outp + (int64_t)j * a.K + kb + 8 * q) =
                f32x4{acc[bi][bj][4 * q], acc[bi][bj][4 * q + 1], acc[bi][bj][4 * q + 2], acc[bi][bj][4 * q + 3]};
      }
    }
    if (do_bias) {
      const float s = cs[bi] + __shfl_xor(cs[bi], 32, 64);   // the two 8-row halves of every k-step
      if (lh == 0 && j < a.J) (a.slab ? outp + (int64_t)a.J * a.K : a.dbias)[j] = s;
    }
  }
}

// out[e] = sum_s slabs[s * slab + e] for e < n (dW), and dbias[e - n] for n <= e < n + nb: one pass, split order
__global__ __launch_bounds__(256) void wg16_reduce_kernel(const float* __restrict__ slabs, int nsplit, int64_t slab, int64_t n4,
                                                          float* __restrict__ out, int64_t nb4, float* __restrict__ dbias) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n4 + nb4; e += (int64_t)gridDim.x * blockDim.x) {
    // (eight slabs in flight before the first addition; additions in split order: see reduce_slabs_kernel, gemm.hip)
    f32x4 s = reinterpret_cast<const f32x4*>(slabs)[e];
    int k = 1;
    for (; k + 8 <= nsplit; k += 8) {
      f32x4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = reinterpret_cast<const f32x4*>(slabs + (int64_t)(k + j) * slab)[e];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += t[j];
    }
    for (; k < nsplit; ++k) s += reinterpret_cast<const f32x4*>(slabs + (int64_t)k * slab)[e];
    if (e < n4) reinterpret_cast<f32x4*>(out)[e] = s;
    else reinterpret_cast<f32x4*>(dbias)[e - n4] = s;
  }
}

}  // namespace

// dW (J, K) fp32 dense and dbias (J) fp32 (nullable) from dC (rows, J) and act (rows, K), bf16 row-major.  J, K multiples of 8
// (J K and J multiples of 4 for the slab reduction), 16-byte aligned operands, leading dimensions multiples of 8.
// splits == 0: the library's rule.  slabs: hig_wgrad_bf16_scratch_floats(J, K, splits) floats.
int hig_wgrad16_launch(const void* dC, int64_t ldd, const void* act, int64_t ldx, int64_t rows, int J, int K, float* dW, float* dbias,
                       int splits, float* slabs, int64_t slab_floats, hipStream_t st) {
  HIG_REQUIRE(dC && act && dW && rows > 0 && J > 0 && K > 0, "hig_wgrad_bf16: bad arguments");
  HIG_REQUIRE(J % 8 == 0 && K % 8 == 0 && ldd % 8 == 0 && ldx % 8 == 0 &&
                  ((reinterpret_cast<uintptr_t>(dC) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(dW) |
                    reinterpret_cast<uintptr_t>(dbias) | reinterpret_cast<uintptr_t>(slabs)) & 15) == 0,
              "hig_wgrad_bf16: J, K and the leading dimensions must be multiples of 8, buffers 16-byte aligned");
  HIG_REQUIRE(rows < (1ll << 31), "hig_wgrad_bf16: too many rows");
  // DMA ring (tuning knob HIG_WG16_RING = depth x 100 + chunk rows): 264 = two 64-row chunks (64 KB, two workgroups per CU),
  // 364 / 464 = three / four (one per CU), 432 / 632 = four / six 32-row chunks (64 / 96 KB)
  static const int ring = getenv("HIG_WG16_RING") ? atoi(getenv("HIG_WG16_RING")) : 264;
  static const int percu_knob = getenv("HIG_WG16_PERCU") ? atoi(getenv("HIG_WG16_PERCU")) : 0;
  const int per_cu = percu_knob > 0 ? percu_knob : ((ring == 264 || ring == 432) ? 2 : 1);   // (2642 = 264 with two sub-groups: one)
  const int ntj = (J + 127) / 128, ntk = (K + 127) / 128, ntiles = ntj * ntk;
  const int64_t slab = (int64_t)J * K + J;
  const int nchunks = (int)((rows + 63) / 64);
  if (splits <= 0) {
    // units = tiles x splits fill, without exceeding, the two resident workgroups per CU; at least four 64-row chunks per
    // unit (the DMA ring needs a few to overlap); the slabs must fit
    splits = 1;
    const int target = per_cu * hig_chip_cus();
    for (int s = 2; s <= 64; ++s) {
      const int cps = (nchunks + s - 1) / s;                    // chunks per split
      if (cps < 4 || (int64_t)ntiles * s > target || !slabs || slab * s > slab_floats) break;
      splits = s;
    }
  }
  int cps = (nchunks + splits - 1) / splits;
  splits = (nchunks + cps - 1) / cps;                            // no empty slice
  HIG_REQUIRE(splits == 1 || (slabs && slab * splits <= slab_floats), "hig_wgrad_bf16: slab scratch too small");
  Wg16Args a;
  a.dC = static_cast<const __bf16*>(dC); a.ldd = ldd;
  a.X = static_cast<const __bf16*>(act); a.ldx = ldx;
  a.out = splits == 1 ? dW : slabs;
  a.dbias = dbias;
  a.slab = splits == 1 ? 0 : slab;
  a.J = J; a.K = K; a.rows = (int)rows; a.rows_per_split = cps * 64; a.ntk = ntk; a.ntiles = ntiles; a.want_bias = dbias != nullptr;
  switch (ring) {
    case 364: hipLaunchKernelGGL((wgrad16_kernel<3, 64>), dim3(ntiles * splits), dim3(256), 0, st, a); break;
    case 432: hipLaunchKernelGGL((wgrad16_kernel<4, 32>), dim3(ntiles * splits), dim3(256), 0, st, a); break;
    case 632: hipLaunchKernelGGL((wgrad16_kernel<6, 32>), dim3(ntiles * splits), dim3(256), 0, st, a); break;
    case 464: hipLaunchKernelGGL((wgrad16_kernel<4, 64>), dim3(ntiles * splits), dim3(256), 0, st, a); break;
    case 2642: hipLaunchKernelGGL((wgrad16_kernel<2, 64, 2>), dim3(ntiles * splits), dim3(512), 0, st, a); break;
    default: hipLaunchKernelGGL((wgrad16_kernel<2, 64>), dim3(ntiles * splits), dim3(256), 0, st, a); break;
  }
  HIG_CHECK_LAUNCH();
  if (splits > 1) {
    const int64_t n4 = (int64_t)J * K / 4, nb4 = dbias ? J / 4 : 0;
    HIG_REQUIRE(((int64_t)J * K) % 4 == 0 && J % 4 == 0, "hig_wgrad_bf16: J K and J must be multiples of 4");
    int64_t blocks = (n4 + nb4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wg16_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, slabs, splits, slab, n4, dW, nb4, dbias);
    HIG_CHECK_LAUNCH();
  }
  return HIG_OK;
}

extern "C" int64_t hig_wgrad_bf16_scratch_floats(int32_t J, int32_t K, int32_t splits) {
  if (J <= 0 || K <= 0) return -1;
  return ((int64_t)J * K + J) * (splits > 0 ? splits : 64);
}
extern "C" int hig_wgrad_bf16(const void* dC, int64_t ldd, const void* act, int64_t ldx, int64_t rows, int32_t J, int32_t K, float* dW,
                              float* dbias, int32_t splits, float* slabs, int64_t slab_floats, hig_stream_t stream) {
  return hig_wgrad16_launch(dC, ldd, act, ldx, rows, J, K, dW, dbias, splits, slabs, slab_floats, hig_stream(stream));
}
