// Weight gradients of the bf16-storage training step on the bf16 matrix cores, WITHOUT transposed operand copies:
//     dW[j][k] = sum_i dC[i][j] * act[i][k]        dbias[j] = sum_i dC[i][j]
// (autograd of nn.Linear over the M = B T rows; reference: the torch autograd of codes/models/transformer.py:81-85,108-114,
// 144-150,168 inside DDPMTrainer.backward_G, codes/trainers/ddpm_trainer.py:172-178).  The reduce index is the ROW of both
// operands, so both MFMA operands need "k = row": neither dC^T nor act^T is ever built in memory.  Row chunks of dC
// (64 rows x 128 columns of j) and of act (64 rows x 128 columns of k) land row-major in LDS by DMA, and ds_read_b64_tr_b16
// -- the transpose read of gfx950 -- hands each lane four consecutive rows of one column: the operand shape of
// v_mfma_f32_32x32x16_bf16.  Same scheme as ctx16_mfma_kernel (linattn16.hip), which is this contraction with a column
// softmax in front of it.  A workgroup owns a 128 x 128 tile of dW for ONE slice of the rows (split-R: few output tiles,
// M = 12 544 rows); the partial tiles go to fp32 slabs [split][J x K | J], summed in split order by wg16_reduce_kernel
// (deterministic, no float atomics).  The bias gradient rides along: the waves that hold the dC fragments of a tile column
// sum them with v_dot2_f32_bf16 (x . (1, 1)).
#include <stdlib.h>

#include "gemm16_epi.h"
#include "hig_host.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));

// CDNA4 mapping (round 5): SPECIALISED waves, one workgroup of twelve waves per CU.
// What bounded the round-4 kernel (four waves that did everything, two workgroups per CU; profiles/r04_notes.md section 1: MFMA
// pipe 16 % busy): (1) a wave reads LDS at ~24 bytes per
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

// Up to HIG_WG_GROUP_MAX weight gradients in ONE launch (the training step's backward: the two or three gradients whose dC
// operands exist at the same time -- stylization out + FFN linear2 + linear1, ... -- share the chip): unit u belongs to problem
// p with unit0[p] <= u < unit0[p + 1].  One launch of 240-256 units instead of three of 256 means a third of the slices per
// gradient: a third of the fp32 slabs, and the prologue / tail of a workgroup amortised over three times as many chunks.
struct WgGroupArgs {
  Wg2Args p[HIG_WG_GROUP_MAX];
  int unit0[HIG_WG_GROUP_MAX + 1];
  int np, units;
};

__global__ __launch_bounds__(768, 3) void wgrad16x_kernel(const WgGroupArgs grp) {
  constexpr int CHK = 64, ROWB = 256, NB = 4, OPB = CHK * ROWB;   // 16 KB per operand and chunk
  __shared__ __attribute__((aligned(1024))) char smem[NB * 2 * OPB];   // [slot][dC | act][64 rows][256 bytes]: 128 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every tile of a split streams the SAME rows of dC and act: they sit on one XCD (blocks b, b + 8, ... share one: unit
  // u = (b % 8) (grid / 8) + b / 8 is contiguous per XCD; speed only), so a row is fetched from HBM once and served to the
  // other tiles by that XCD's L2 -- in blockIdx order a 256-byte piece of dC went to four XCDs (3x the HBM traffic)
  const int ug = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (ug >= grp.units) return;
  int pi = 0;
  while (pi + 1 < grp.np && ug >= grp.unit0[pi + 1]) ++pi;
  const Wg2Args& a = grp.p[pi];                  // (workgroup-uniform: scalar loads from the kernel arguments)
  const int u = ug - grp.unit0[pi];
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
    [[maybe_unused]] const unsigned base = sm_lds + (c % NB) * 2 * OPB + 16 * ks4 * ROWB;
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
    // (eight slabs in flight before the first addition; additions in split order: see reduce_slabs_kernel, gemm.hip.  The last,
    // partial batch too: its loads are clamped to the last slab and the surplus is not added -- a sequential tail was up to
    // seven DEPENDENT loads, 20 us for the 8 slabs of an FFN weight against 7 us for the 16 slabs of a 512 x 512 one)
    f32x4 s = reinterpret_cast<const f32x4*>(slabs)[e];
    for (int k = 1; k < nsplit; k += 8) {
      f32x4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = reinterpret_cast<const f32x4*>(slabs + (int64_t)min(k + j, nsplit - 1) * slab)[e];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k + j < nsplit) s += t[j];
    }
    if (e < n4) reinterpret_cast<f32x4*>(out)[e] = s;
    else reinterpret_cast<f32x4*>(dbias)[e - n4] = s;
  }
}

// The same reduction for up to WG_RB_MAX weight gradients in ONE launch (the training step's backward: a decoder layer's eight
// gradients wait in their own slab ranges and are summed together at the end of the layer -- 66 launches of ~5 us, mostly
// launch latency inside the captured step, become 8).  Block -> (entry, block of the entry) through a prefix table.
struct WgReduceBatch {
  hig_wg_reduce e[HIG_WG_RB_MAX];
  int block0[HIG_WG_RB_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void wg16_reduce_batch_kernel(const WgReduceBatch b) {
  int m = 0;
  while (m + 1 < b.n && (int)blockIdx.x >= b.block0[m + 1]) ++m;
  const hig_wg_reduce& r = b.e[m];
  const float* __restrict__ slabs = r.slabs;
  const int nsplit = r.nsplit;
  const int64_t slab = r.slab, n4 = r.n4, total = r.n4 + r.nb4;
  const int nblk = b.block0[m + 1] - b.block0[m];
  for (int64_t e = (blockIdx.x - b.block0[m]) * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)nblk * blockDim.x) {
    f32x4 s = reinterpret_cast<const f32x4*>(slabs)[e];
    for (int k = 1; k < nsplit; k += 8) {
      f32x4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = reinterpret_cast<const f32x4*>(slabs + (int64_t)min(k + j, nsplit - 1) * slab)[e];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k + j < nsplit) s += t[j];
    }
    if (e < n4) reinterpret_cast<f32x4*>(r.out)[e] = s;
    else reinterpret_cast<f32x4*>(r.dbias)[e - n4] = s;
  }
}

}  // namespace

// dW (J, K) fp32 dense and dbias (J) fp32 (nullable) from dC (rows, J) and act (rows, K), bf16 row-major.  J, K multiples of 8
// (J K and J multiples of 4 for the slab reduction), 16-byte aligned operands, leading dimensions multiples of 8.
// splits == 0: the library's rule.  slabs: hig_wgrad_bf16_scratch_floats(J, K, splits) floats.
// The library's split rule: floats of slab scratch a gradient of this shape uses (splits x (J K + J); J K + J means "one split, no
// slabs").  units = tiles x splits fill, without exceeding, the CUs (128 KB of LDS, twelve waves: one workgroup per CU); at least
// four 64-row chunks per unit (the DMA ring needs a few to overlap); the slabs must fit into `room` floats.
int64_t hig_wgrad16_rule_floats(int64_t rows, int J, int K, int64_t room) {
  const int ntiles = ((J + 127) / 128) * ((K + 127) / 128);
  const int64_t slab = (int64_t)J * K + J;
  const int nchunks = (int)((rows + 63) / 64);
  int splits = 1;
  const int target = hig_chip_cus();
  for (int s = 2; s <= 64; ++s) {
    const int cps = (nchunks + s - 1) / s;                    // chunks per split
    if (cps < 4 || (int64_t)ntiles * s > target || slab * s > room) break;
    splits = s;
  }
  return slab * splits;
}

// A group of weight gradients in one launch.  splits_p (0: the rule below) slices per problem; the slabs of the problems that
// need them are laid out one after the other in `slabs` (16-byte aligned ranges); *used_floats (nullable) receives the floats taken.
// deferred (nullable, n entries): the slab reductions are NOT launched; deferred[p] describes problem p's for
// hig_wgrad16_reduce_batch (nsplit = 0 when it needed no slabs) and the slabs must stay untouched until then.
// Rule: every problem starts with one slice; the slice count of the problem with the most rows per slice grows while the units
// still fit the CUs (128 KB of LDS, twelve waves: one workgroup per CU), at least four 64-row chunks stay in a slice and
// the slabs fit.
int hig_wgrad16_launch_group(const hig_wg_problem* probs, int n, float* slabs, int64_t slab_floats, hipStream_t st,
                             hig_wg_reduce* deferred, int64_t* used_floats) {
  HIG_REQUIRE(probs && n >= 1 && n <= HIG_WG_GROUP_MAX, "hig_wgrad_bf16: 1 .. %d problems per launch", HIG_WG_GROUP_MAX);
  int ntiles[HIG_WG_GROUP_MAX], nchunks[HIG_WG_GROUP_MAX], splits[HIG_WG_GROUP_MAX];
  int64_t slab[HIG_WG_GROUP_MAX];
  int units = 0;
  bool forced = false;
  for (int p = 0; p < n; ++p) {
    const hig_wg_problem& q = probs[p];
    HIG_REQUIRE(q.dC && q.act && q.dW && q.rows > 0 && q.J > 0 && q.K > 0, "hig_wgrad_bf16: bad arguments");
    HIG_REQUIRE(q.J % 8 == 0 && q.K % 8 == 0 && q.ldd % 8 == 0 && q.ldx % 8 == 0 &&
                    ((reinterpret_cast<uintptr_t>(q.dC) | reinterpret_cast<uintptr_t>(q.act) | reinterpret_cast<uintptr_t>(q.dW) |
                      reinterpret_cast<uintptr_t>(q.dbias) | reinterpret_cast<uintptr_t>(slabs)) & 15) == 0,
                "hig_wgrad_bf16: J, K and the leading dimensions must be multiples of 8, buffers 16-byte aligned");
    HIG_REQUIRE(q.rows < (1ll << 31), "hig_wgrad_bf16: too many rows");
    // the buffer descriptors of the loader waves address the rows with 32-bit byte offsets
    if (q.rows * q.ldd * 2 >= (1ll << 31) || q.rows * q.ldx * 2 >= (1ll << 31))
      return hig_set_error(HIG_EUNSUPPORTED, "hig_wgrad_bf16: an operand of more than 2 GiB");
    ntiles[p] = ((q.J + 127) / 128) * ((q.K + 127) / 128);
    nchunks[p] = (int)((q.rows + 63) / 64);
    slab[p] = ((int64_t)q.J * q.K + q.J + 3) / 4 * 4;
    splits[p] = q.splits > 0 ? (q.splits < nchunks[p] ? q.splits : nchunks[p]) : 1;
    forced = forced || q.splits > 0;
    units += ntiles[p] * splits[p];
  }
  auto slab_need = [&]() {
    int64_t f = 0;
    for (int p = 0; p < n; ++p) f += splits[p] > 1 ? slab[p] * splits[p] : 0;
    return f;
  };
  if (!forced) {
    const int target = hig_chip_cus();
    for (;;) {
      int best = -1;
      double most = 0;
      for (int p = 0; p < n; ++p) {
        const int s1 = splits[p] + 1;
        if (s1 > 64 || (nchunks[p] + s1 - 1) / s1 < 4 || units + ntiles[p] > target) continue;
        if (!slabs) continue;
        const double per = (double)nchunks[p] / splits[p];
        if (per > most) { most = per; best = p; }
      }
      if (best < 0) break;
      ++splits[best];
      if (slab_need() > slab_floats) { --splits[best]; break; }
      units += ntiles[best];
    }
  }
  HIG_REQUIRE(slab_need() == 0 || (slabs && slab_need() <= slab_floats), "hig_wgrad_bf16: slab scratch too small");
  WgGroupArgs g;
  g.np = n;
  int64_t off = 0;
  int u0 = 0;
  for (int p = 0; p < n; ++p) {
    const hig_wg_problem& q = probs[p];
    Wg2Args& a = g.p[p];
    a.dC = static_cast<const __bf16*>(q.dC); a.ldd = q.ldd;
    a.X = static_cast<const __bf16*>(q.act); a.ldx = q.ldx;
    a.out = splits[p] == 1 ? q.dW : slabs + off;
    a.dbias = q.dbias;
    a.slab = splits[p] == 1 ? 0 : slab[p];
    a.J = q.J; a.K = q.K; a.rows = (int)q.rows; a.nsplit = splits[p]; a.ntk = (q.K + 127) / 128; a.ntiles = ntiles[p]; a.want_bias = q.dbias != nullptr;
    a.units = ntiles[p] * splits[p];
    a.stamps = g_wg_stamps;
    g.unit0[p] = u0;
    u0 += a.units;
    if (deferred) deferred[p].nsplit = 0;
    if (splits[p] > 1) {
      HIG_REQUIRE(((int64_t)q.J * q.K) % 4 == 0 && q.J % 4 == 0, "hig_wgrad_bf16: J K and J must be multiples of 4");
      if (deferred) {
        deferred[p].slabs = slabs + off; deferred[p].nsplit = splits[p]; deferred[p].slab = slab[p]; deferred[p].n4 = (int64_t)q.J * q.K / 4;
        deferred[p].out = q.dW; deferred[p].nb4 = q.dbias ? q.J / 4 : 0; deferred[p].dbias = q.dbias;
      }
      off += slab[p] * splits[p];
    }
  }
  g.unit0[n] = u0;
  g.units = u0;
  if (used_floats) *used_floats = off;
  hipLaunchKernelGGL(wgrad16x_kernel, dim3((g.units + 7) / 8 * 8), dim3(768), 0, st, g);
  HIG_CHECK_LAUNCH();
  if (!deferred) {
    for (int p = 0; p < n; ++p) {
      if (g.p[p].nsplit <= 1) continue;
      const hig_wg_problem& q = probs[p];
      const int64_t n4 = (int64_t)q.J * q.K / 4, nb4 = q.dbias ? q.J / 4 : 0;
      int64_t blocks = (n4 + nb4 + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(wg16_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g.p[p].out, g.p[p].nsplit, g.p[p].slab, n4, q.dW, nb4, q.dbias);
      HIG_CHECK_LAUNCH();
    }
  }
  return HIG_OK;
}

int hig_wgrad16_launch(const void* dC, int64_t ldd, const void* act, int64_t ldx, int64_t rows, int J, int K, float* dW, float* dbias,
                       int splits, float* slabs, int64_t slab_floats, hipStream_t st, hig_wg_reduce* deferred) {
  const hig_wg_problem q{dC, ldd, act, ldx, rows, J, K, dW, dbias, splits};
  return hig_wgrad16_launch_group(&q, 1, slabs, slab_floats, st, deferred, nullptr);
}

int hig_wgrad16_reduce_batch(const hig_wg_reduce* entries, int n, hipStream_t st) {
  HIG_REQUIRE(n >= 0 && n <= HIG_WG_RB_MAX, "hig_wgrad16_reduce_batch: at most %d entries", HIG_WG_RB_MAX);
  WgReduceBatch b;
  b.n = 0;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (entries[i].nsplit <= 1) continue;
    b.e[b.n] = entries[i];
    b.block0[b.n] = blocks;
    int64_t nb = (entries[i].n4 + entries[i].nb4 + 255) / 256;
    if (nb > 1024) nb = 1024;
    blocks += (int)nb;
    ++b.n;
  }
  if (b.n == 0) return HIG_OK;
  b.block0[b.n] = blocks;
  hipLaunchKernelGGL(wg16_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
  HIG_CHECK_LAUNCH();
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
  return hig_wgrad16_launch(dC, ldd, act, ldx, rows, J, K, dW, dbias, splits, slabs, slab_floats, hig_stream(stream), nullptr);
}
