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


// ---------------------------------------------------------------------------------------------------------------------------
// Second form (round 5): the same contraction with SPECIALISED waves, one workgroup of twelve waves per CU.
// What bounded the kernel above (profiles/r04_notes.md section 1: MFMA pipe 16 % busy): (1) a wave reads LDS at ~24 bytes per
// clock however the reads are shaped (tools/lds_read_probe.hip), and a 64 x 64 sub-tile needs 4 KB of transposed fragments per
// four MFMAs = 32 bytes per clock from ONE wave per SIMD; (2) each wave issued its own share of the DMA between its MFMAs (eight
// 1-KiB instructions per chunk at 60-185 cycles of issue stall); (3) the transpose reads ran into bank conflicts (half of
// the LDS cycles); (4) 28-32 splits of a 512 x 512 weight = 32 MB of partial tiles per launch.
//   * waves 0-7, the MATRIX waves, two per SIMD: two groups of four waves (2 x 2 sub-tiles of 64 x 64) over the SAME 128 x 128
//     tile; group g takes the 16-row k-steps of parity g.  Each SIMD's matrix pipe alternates between its two waves, so a
//     wave needs its 4 KB of fragments per 256 cycles = 16 bytes per clock; the next k-step's eight transpose reads are in
//     flight while the current four MFMAs run.  The groups' tiles are summed through LDS at the end.
//   * waves 8-11, the LOADER waves (one per SIMD): every LDS-DMA instruction of the workgroup (32 per 64-row chunk), three chunks
//     ahead in a ring of four; rows beyond the slice come back as zeros from the buffer descriptor's range check (no clamping,
//     no zero-fill pass).  One s_barrier per chunk; a chunk is confirmed one barrier early, so the matrix waves' fragment
//     prefetch runs across chunk seams.
//   * LDS image: plain 256-byte rows, 16-byte chunk c of row r at c ^ (((r & 3) << 2) | ((r >> 2) & 3)): conflict-free for the
//     transposed reads of the 32x32x16 operand (cdna_hip_programming.md T10, image (b)).
//   * one workgroup per CU: tiles x splits <= 256, i.e. 16 splits of a 512 x 512 weight (16 MB of partial tiles, half of before).
struct Wg2Args {
  const __bf16* dC; int64_t ldd;     // (rows, J)
  const __bf16* X; int64_t ldx;      // (rows, K)
  float* out;                        // splits == 1: dW (J, K) dense; else slabs: [split][J * K + J]
  float* dbias;                      // splits == 1: dbias (J) or null
  int64_t slab;                      // floats per split (J * K + J), 0 when splits == 1
  int J, K, rows, nsplit, ntk, ntiles, want_bias;   // slice s = 64-row chunks [s nch / nsplit, (s + 1) nch / nsplit) of the rows
  int units;                         // tiles x splits (the grid is rounded up to a multiple of 8)
  unsigned long long* stamps;        // diagnostic (hig_wgrad16_debug_stamps), else NULL
};
unsigned long long* g_wg_stamps = nullptr;

__global__ __launch_bounds__(768, 3) void wgrad16x_kernel(const Wg2Args a) {
  constexpr int CHK = 64, ROWB = 256, NB = 4, OPB = CHK * ROWB;   // 16 KB per operand and chunk
  __shared__ __attribute__((aligned(1024))) char smem[NB * 2 * OPB];   // [slot][dC | act][64 rows][256 bytes]: 128 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every tile of a split streams the SAME rows of dC and act: they sit on one XCD (blocks b, b + 8, ... share one: unit
  // u = (b % 8) (grid / 8) + b / 8 is contiguous per XCD; speed only), so a row is fetched from HBM once and served to the
  // other tiles by that XCD's L2 -- in blockIdx order a 256-byte piece of dC went to four XCDs (3x the HBM traffic)
  const int u = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (u >= a.units) return;
  const int split = u / a.ntiles, tile = u - split * a.ntiles;
  const int tj = tile / a.ntk, tk = tile - tj * a.ntk;
  const int j0 = tj * 128, k0 = tk * 128;
  const int nch_all = (a.rows + CHK - 1) / CHK;
  const int cbeg = (int)((int64_t)split * nch_all / a.nsplit), cend = (int)((int64_t)(split + 1) * nch_all / a.nsplit);
  const int rbeg = cbeg * CHK;
  const int len = min(a.rows, cend * CHK) - rbeg;                // rows of this slice (> 0: nsplit <= chunks)
  const int nchunk = cend - cbeg;
  auto fsw = [](int r) { return ((r & 3) << 2) | ((r >> 2) & 3); };
  auto stamp = [&](int k) {                      // thread 0 (a matrix wave) / thread 512 (a loader wave), diagnostic runs only
    if (a.stamps && (tid == 0 || tid == 512)) {
      unsigned long long tm;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm)::"memory");
      a.stamps[(size_t)(tid ? 4096 : 0) + (size_t)blockIdx.x * 16 + k] = tm;
    }
  };
  stamp(0);

  if (wave >= 8) {
    // ======================================================= LOADER WAVES =======================================================
    const int lw = wave - 8;
    // descriptors over the slice's rows only: a row index >= len lands beyond num_records and reads as zero
    const int64_t endD = rbeg + len == a.rows ? ((int64_t)(len - 1) * a.ldd + a.J) * 2 : (int64_t)len * a.ldd * 2;
    const int64_t endX = rbeg + len == a.rows ? ((int64_t)(len - 1) * a.ldx + a.K) * 2 : (int64_t)len * a.ldx * 2;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.dC + (int64_t)rbeg * a.ldd), 0, (int)endD, 0x00020000);
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.X + (int64_t)rbeg * a.ldx), 0, (int)endX, 0x00020000);
    // instruction n of a chunk and operand covers rows 4 n .. 4 n + 3: lane -> row 4 n + lane / 16, LDS position lane % 16
    // receives the row's chunk position ^ f(row)
    int voD[4], voX[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 4 * (lw + 4 * q) + (lane >> 4);
      const int src = (lane & 15) ^ fsw(row);
      voD[q] = (row * (int)a.ldd + j0 + 8 * src) * 2;
      voX[q] = (row * (int)a.ldx + k0 + 8 * src) * 2;
    }
    auto dma_chunk = [&](int c) {
      [[maybe_unused]] char* const base = smem + (c % NB) * 2 * OPB;
      // (the whole offset goes into the per-lane part: the range check of a raw buffer does not see the scalar offset)
      [[maybe_unused]] const int soD = c * CHK * (int)a.ldd * 2, soX = c * CHK * (int)a.ldx * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        [[maybe_unused]] const int n = lw + 4 * q;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(base + n * 1024), 16, voD[q] + soD, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(base + OPB + n * 1024), 16, voX[q] + soX, 0, 0, 0);
#endif
      }
    };
    for (int c = 0; c < 3 && c < nchunk; ++c) dma_chunk(c);
    // chunks 0 and 1 have landed once only chunk 2's eight requests are outstanding
    if (nchunk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(1);
    for (int c = 0; c < nchunk; ++c) {
      __builtin_amdgcn_s_barrier();              // B_c: chunks <= c + 1 have landed; chunk c - 1 has been consumed
      asm volatile("" ::: "memory");
      if (c == 4) stamp(2);
      if (c + 3 < nchunk) {
        dma_chunk(c + 3);
        if (c == 4) stamp(3);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // chunk c + 2 has landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (c == 4) stamp(4);
    }
    stamp(5);
    __builtin_amdgcn_s_barrier();                // E1: every chunk consumed (the ring becomes the reduction buffer)
    __builtin_amdgcn_s_barrier();                // E2: group 1's tile is in LDS
    return;
  }

  // ========================================================= MATRIX WAVES =========================================================
  const int g = wave >> 2, w4 = wave & 3;
  const int wi = w4 >> 1, wj = w4 & 1, lr = lane & 31, lh = lane >> 5;
  const int gi = lane & 15, gg = lane >> 4;
  // transposed read of a 4-row x 16-column block: lane 4 q + p of a 16-lane group supplies the address of row r0 + q, columns
  // 4 p .. 4 p + 3 (8 bytes); group gg covers columns 16 (gg & 1) .. of the 32-column block, rows 8 (gg >> 1) + 4 part of the k-step
  auto tr_off = [&](int colbase, int rr) {       // rr: row inside the chunk
    const int col = colbase + 16 * (gg & 1) + 4 * (gi & 3);
    return rr * ROWB + 16 * ((col >> 3) ^ fsw(rr)) + 2 * (col & 7);
  };
  // per k-step ks4 (0..3 inside a chunk) and part: the row is 16 ks4 + 8 (gg >> 1) + 4 part + (gi >> 2); f(row) depends on
  // row & 15 only, so the byte offsets of k-step 0 serve every k-step with + 16 ks4 ROWB
  int offD[2][2], offX[2][2];                    // [part][32-column block]
#pragma unroll
  for (int part = 0; part < 2; ++part)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int rr = 8 * (gg >> 1) + 4 * part + (gi >> 2);
      offD[part][bb] = tr_off(32 * (2 * wi + bb), rr);
      offX[part][bb] = OPB + tr_off(32 * (2 * wj + bb), rr);
    }
  f32x16 acc[2][2];
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[bi][bj][e] = 0.f;
  float cs[2] = {0.f, 0.f};                      // column sums of dC (bias gradient): waves wj == 0 of the tiles tk == 0
  const bool do_bias = a.want_bias && tk == 0 && wj == 0;
  const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
  const unsigned sm_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  s16x8 fA[2][2], fB[2][2];                      // [buffer][block]: act (A operand) and dC (B operand) fragments of a k-step
  auto load_frags = [&](s16x8 (&xa)[2], s16x8 (&da)[2], int c, int ks4) {
    const unsigned base = sm_lds + (c % NB) * 2 * OPB + 16 * ks4 * ROWB;
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        s16x4 x4 = {0, 0, 0, 0}, d4 = {0, 0, 0, 0};
#if defined(__HIP_DEVICE_COMPILE__)   // (LDS pointers are 32 bits wide on the device only)
        x4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + offX[part][bb]));
        d4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + offD[part][bb]));
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) { xa[bb][4 * part + e] = x4[e]; da[bb][4 * part + e] = d4[e]; }
      }
  };
  auto compute = [&](s16x8 (&xa)[2], s16x8 (&da)[2]) {
    if (do_bias) {
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const bf16x8 dv = __builtin_bit_cast(bf16x8, da[bb]);
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[bb] = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{dv[2 * e], dv[2 * e + 1]}, ones, cs[bb], false);
      }
    }
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
        acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xa[bj]), __builtin_bit_cast(bf16x8, da[bi]), acc[bi][bj], 0, 0, 0);
  };
  // group g: k-steps g and g + 2 of every chunk.  Buffer 0 holds the first, buffer 1 the second k-step of a chunk.
  __builtin_amdgcn_s_barrier();                  // B_0: chunks 0 and 1 have landed
  asm volatile("" ::: "memory");
  stamp(1);
  load_frags(fA[0], fB[0], 0, g);
  for (int c = 0; c < nchunk; ++c) {
    load_frags(fA[1], fB[1], c, g + 2);
    compute(fA[0], fB[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < nchunk) load_frags(fA[0], fB[0], c + 1, g);      // (chunk c + 1 was confirmed at B_c)
    compute(fA[1], fB[1]);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < nchunk) {
      __builtin_amdgcn_s_barrier();              // B_(c+1): chunk c consumed by this wave (its fragments are in registers or used)
      asm volatile("" ::: "memory");
    }
  }
  // ---- the two groups' tiles summed through LDS (every DMA has landed and every fragment read returned): group g finishes the
  // block row bi = g of every wave's 2 x 2 blocks -- it hands the other block row to its partner (16-byte LDS operations, 32 KB
  // each way) and stores 32 of the wave pair's 64 KB
  f32x4* red = reinterpret_cast<f32x4*>(smem);                   // [group][w4][bj][4 quads][64 lanes] x 16 bytes = 2 x 32 KB
  float* redb = reinterpret_cast<float*>(smem + 65536);          // [group][w4][64 lanes]
  stamp(2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                  // E1
  asm volatile("" ::: "memory");
  {
    const int give = g ^ 1;                      // the block row this wave gives away
#pragma unroll
    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x16& t = give ? acc[1][bj] : acc[0][bj];
        red[(((g * 4 + w4) * 2 + bj) * 4 + q) * 64 + lane] = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
      }
    redb[(g * 4 + w4) * 64 + lane] = give ? cs[1] : cs[0];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                  // E2
  asm volatile("" ::: "memory");
  float* outp = a.out + (int64_t)split * a.slab;
  {
    const int bi = g;                            // the block row this wave finishes: its own + the partner group's
    const int j = j0 + 32 * (2 * wi + bi) + lr;
    float csum = (bi ? cs[1] : cs[0]) + redb[((g ^ 1) * 4 + w4) * 64 + lane];
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      const int kb = k0 + 32 * (2 * wj + bj) + 4 * lh;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x16& t = bi ? acc[1][bj] : acc[0][bj];
        const f32x4 o = red[((((g ^ 1) * 4 + w4) * 2 + bj) * 4 + q) * 64 + lane];
        // accumulator element 4 q + e of lane (lr, lh) is dW[j = jb + lr][k = kb + 8 q + 4 lh + e]
        if (j < a.J && kb + 8 * q < a.K)
          *reinterpret_cast<f32x4*>(outp + (int64_t)j * a.K + kb + 8 * q) = f32x4{t[4 * q] + o.x, t[4 * q + 1] + o.y, t[4 * q + 2] + o.z, t[4 * q + 3] + o.w};
      }
    }
    if (do_bias) {
      const float s2 = csum + __shfl_xor(csum, 32, 64);   // the two 8-row halves of every k-step
      if (lh == 0 && j < a.J) (a.slab ? outp + (int64_t)a.J * a.K : a.dbias)[j] = s2;
    }
  }
  stamp(3);
  if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 16 + 14] = (unsigned long long)nchunk;
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
  static const int form = getenv("HIG_WG16_FORM") ? atoi(getenv("HIG_WG16_FORM")) : 2;   // tuning knob: 1 = the round-4 kernel
  const int ntj = (J + 127) / 128, ntk = (K + 127) / 128, ntiles = ntj * ntk;
  const int64_t slab = (int64_t)J * K + J;
  const int nchunks = (int)((rows + 63) / 64);
  // the buffer descriptors of the loader waves address a slice with 32-bit byte offsets
  const bool form2 = form == 2 && rows * ldd * 2 < (1ll << 31) && rows * ldx * 2 < (1ll << 31);
  const int per_cu = form2 ? 1 : 2;              // (128 KB of LDS, twelve waves: one workgroup per CU; the round-4 kernel: two)
  if (splits <= 0) {
    // units = tiles x splits fill, without exceeding, the resident workgroups; at least four 64-row chunks per unit (the DMA
    // ring needs a few to overlap); the slabs must fit
    splits = 1;
    const int target = per_cu * hig_chip_cus();
    for (int s = 2; s <= 64; ++s) {
      const int cps = (nchunks + s - 1) / s;                    // chunks per split
      if (cps < 4 || (int64_t)ntiles * s > target || !slabs || slab * s > slab_floats) break;
      splits = s;
    }
  }
  int cps = (nchunks + splits - 1) / splits;
  if (form2) { if (splits > nchunks) splits = nchunks; }         // (slices are chunk ranges of near-equal length: none is empty)
  else splits = (nchunks + cps - 1) / cps;                       // no empty slice
  HIG_REQUIRE(splits == 1 || (slabs && slab * splits <= slab_floats), "hig_wgrad_bf16: slab scratch too small");
  if (form2) {
    Wg2Args a;
    a.dC = static_cast<const __bf16*>(dC); a.ldd = ldd;
    a.X = static_cast<const __bf16*>(act); a.ldx = ldx;
    a.out = splits == 1 ? dW : slabs;
    a.dbias = dbias;
    a.slab = splits == 1 ? 0 : slab;
    a.J = J; a.K = K; a.rows = (int)rows; a.nsplit = splits; a.ntk = ntk; a.ntiles = ntiles; a.want_bias = dbias != nullptr;
    a.units = ntiles * splits;
    a.stamps = g_wg_stamps;
    hipLaunchKernelGGL(wgrad16x_kernel, dim3((a.units + 7) / 8 * 8), dim3(768), 0, st, a);
  } else {
    Wg16Args a;
    a.dC = static_cast<const __bf16*>(dC); a.ldd = ldd;
    a.X = static_cast<const __bf16*>(act); a.ldx = ldx;
    a.out = splits == 1 ? dW : slabs;
    a.dbias = dbias;
    a.slab = splits == 1 ? 0 : slab;
    a.J = J; a.K = K; a.rows = (int)rows; a.rows_per_split = cps * 64; a.ntk = ntk; a.ntiles = ntiles; a.want_bias = dbias != nullptr;
    hipLaunchKernelGGL((wgrad16_kernel<2, 64>), dim3(ntiles * splits), dim3(256), 0, st, a);
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

// Diagnostic (tools/wgrad_stamps.py): s_memtime stamps of thread 0 (matrix wave: buf[block * 16 + k], k = 0 start, 1 first chunks
// landed, 2 chunk loop done, 3 end, 14 = chunks) and of thread 512 (loader wave: buf[4096 + block * 16 + k], k = 0 start, 1 first
// chunks landed, 2 / 3 / 4 = iteration 4: behind the barrier / DMA issued / chunk 6 landed, 5 loop done).  buf: 8192 x 8 bytes.
extern "C" int hig_wgrad16_debug_stamps(void* buf) {
  g_wg_stamps = static_cast<unsigned long long*>(buf);
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
