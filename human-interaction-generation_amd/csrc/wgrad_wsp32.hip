// OUTPUT-STATIONARY exact-fp32 weight-gradient GEMM with SPECIALISED WAVES for gfx950 (MI355X):
//   dW[i][j] = sum_m dC[m][i] * act[m][j]   (+ dbias[i] = sum_m dC[m][i]),   fp32, v_mfma_f32_16x16x4_f32,
// the autograd of every nn.Linear over the frame rows of the fp32 training step (codes/models/transformer.py:81-85, 108-114,
// 144-150, 157-170; codes/trainers/ddpm_trainer.py:172-187): both operands are the ROW-MAJOR buffers the backward already holds
// (dC (M, I), activations (M, J)), the reduce range is the M = B.T rows, split over the workgroups into fp32 slabs that the
// deterministic slab reduction of gemm.hip sums in split order -- the contract of hig_gemm_split / hig_gemm_launch(splits > 1).
//
// Why (profiles/r06_notes.md section 2).  On the tiled kernel of gemm.hip (64 x 64 tiles, 1024 workgroups, every wave staging
// both operands through registers) the weight gradients were 30 % of the fp32 training step's kernel time at 0.33-0.55 of the
// fp32 matrix peak.  Same recipe as gemm_wsp32.hip, with the roles turned: here the OUTPUT tile is what stays in registers.
//
// CDNA4 mapping.  One workgroup of eight waves per CU; tiles x splits <= 256 workgroups.
//   * waves 0-3, the MATRIX waves (2 x 2): each owns 64 x 64 of the 128 x 128 output tile as 16 accumulators of 16 x 16
//     (64 VGPRs) for the whole row range.  Per 32-row stage: 8 k-steps of 4 rows, each 4 + 4 ds_read_b32 (A = dC^T and B = act
//     fragments straight from the row-major LDS tiles: a lane reads one element, 16 consecutive lanes one 64-byte run) and 16
//     MFMAs -- 128 MFMAs = 4096 matrix cycles per stage.  The fragment ring runs across the stage barrier (the last XS k-steps of
//     stage t are multiplied behind the barrier that opens stage t + 1).
//   * waves 4-7, the SERVICE waves: the LDS-DMA of both operand tiles of stage t + 1 (32 KB: 1 KiB = two rows per instruction;
//     rows at or beyond the split's end come back as ZEROS from the buffer descriptor's range check) and, for the tiles of the
//     first output-column block, the column sums of the dC tile (the bias gradient), read from the same LDS tile.
//   * ONE s_barrier per stage.  LDS: 3 stages x 32 KB; the finished tile leaves through the (then idle) ring as whole 256-byte
//     row segments.
// LDS image of a stage tile: [32 rows][128 columns] fp32, 512-byte rows; in ODD rows the two 64-byte halves of every 128-byte
// group are swapped (16-byte chunk c at position c ^ 4), so the two rows a 32-lane ds_read_b32 group touches fall on different
// banks.  Work split: slot w = 32 (block % 8) + block / 8; split = w / tiles, tile = w % tiles: the tiles of one split (which
// stream the same rows of dC and act) sit on one XCD.
#include <stdlib.h>

#include <type_traits>

#include "hig_common.h"
#include "hig_host.h"

namespace {

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
typedef int wg_i32x4 __attribute__((ext_vector_type(4)));

struct Wg32Args {
  const float* dC; int64_t ldd;    // (M, I) row-major
  const float* act; int64_t lda;   // (M, J) row-major
  float* slabs; int64_t slab;      // split s writes slabs + s * slab, dense [I][J]
  float* xsum; int64_t xsum_stride;   // split s writes xsum + s * xsum_stride, [I] (nullable)
  int I, J, M;
  int nti, ntj;                    // 128-wide tiles along I / J
  int splits;
  int nstage;                      // 32-row stages in all
};

constexpr int WG_BM = 32;                  // rows per stage
constexpr int WG_TROW = 512;               // bytes per tile row (128 floats)
constexpr int WG_TILE = WG_BM * WG_TROW;   // 16 KB
constexpr int WG_STAGE = 2 * WG_TILE;      // dC tile + act tile
constexpr int WG_NS = 3;
constexpr int WG_NKS = 8;                  // k-steps (4 rows) per stage
constexpr int WG_XS = 3, WG_RS = 4;        // fragment ring (gemm_wsp32.hip: reads go into the slot released one k-step ago)
constexpr int WG_OLD = 64 + 4;             // output staging: floats per row of a wave's 64 x 64 block
static_assert(WG_NKS % WG_RS == 0, "ring index = k-step modulo RS in every stage");
static_assert(4 * 64 * WG_OLD * 4 <= WG_NS * WG_STAGE, "the output staging fits in the idle ring");

__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(512, 2) void wgrad_wsp32_kernel(const Wg32Args a) {
  __shared__ __attribute__((aligned(1024))) char smem[WG_NS * WG_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = (blockIdx.x & 7) * 32 + (blockIdx.x >> 3);
  const int ntile = a.nti * a.ntj;
  const int split = w / ntile, tile = w % ntile;
  if (split >= a.splits) return;
  const int ti = tile / a.ntj, tj = tile % a.ntj;
  const int i0 = ti * 128, j0 = tj * 128;
  const int sb = (int)((int64_t)split * a.nstage / a.splits), se = (int)((int64_t)(split + 1) * a.nstage / a.splits);
  const int nst = se - sb;
  const int m_end = min(se * WG_BM, a.M);          // rows of this split end here: beyond, the operands read as zero
  const unsigned s_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  if (wave < 4) {
    // =================================================== MATRIX WAVES ===================================================
    const int wi = wave >> 1, wj = wave & 1;
    const int c = lane & 15, mq = lane >> 4;
    // fragment addresses inside a stage: row 4 s + mq, column 64 w + 16 b + c; odd rows: column ^ 16
    const int sw16 = (mq & 1) * 16;
    unsigned aoff[4], boff[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      aoff[b] = mq * WG_TROW + ((64 * wi + 16 * b) ^ sw16) * 4 + c * 4;
      boff[b] = WG_TILE + mq * WG_TROW + ((64 * wj + 16 * b) ^ sw16) * 4 + c * 4;
    }
    wg_f32x4 acc[4][4];
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) acc[bi][bj] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    float fa[WG_RS][4], fb[WG_RS][4];
    // iteration i (behind barrier B_i): reads of stage i's k-steps, MFMAs of the last XS k-steps of stage i - 1 (TAIL, fragments
    // already in registers) and of the first NKS - XS of stage i (MAIN)
    auto step = [&](auto has_tail, auto has_main, int i) {
      constexpr bool TAIL = decltype(has_tail)::value, MAIN = decltype(has_main)::value;
      if constexpr (TAIL && MAIN) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(8 * WG_XS > 15 ? 15 : 8 * WG_XS) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // eight fragment addresses per stage, the k-step (4 rows = 2 KB) as the instruction's immediate offset: opaque to hipcc on
      // purpose (it otherwise keeps base + k-step and spends a v_add per read)
      const unsigned base = s_lds + (i % WG_NS) * WG_STAGE;
      unsigned ab[4], bb[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        ab[b] = base + aoff[b];
        bb[b] = base + boff[b];
        asm volatile("" : "+v"(ab[b]), "+v"(bb[b]));
      }
#pragma unroll
      for (int s = 0; s < WG_NKS; ++s) {
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) {
#pragma unroll
          for (int bj = 0; bj < 4; ++bj) {
            if (s < WG_XS) {
              if constexpr (TAIL)
                acc[bi][bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[(WG_NKS - WG_XS + s) % WG_RS][bi], fb[(WG_NKS - WG_XS + s) % WG_RS][bj], acc[bi][bj], 0, 0, 0);
            } else {
              if constexpr (MAIN)
                acc[bi][bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[(s - WG_XS) % WG_RS][bi], fb[(s - WG_XS) % WG_RS][bj], acc[bi][bj], 0, 0, 0);
            }
          }
          if constexpr (MAIN) {
            if (bi == 0) {   // (behind the first MFMAs of the k-step: the ring slot written was released one k-step ago)
#pragma unroll
              for (int b = 0; b < 4; ++b) {
                fa[s % WG_RS][b] = *reinterpret_cast<const __attribute__((address_space(3))) float*>(ab[b] + s * 4 * WG_TROW);
                fb[s % WG_RS][b] = *reinterpret_cast<const __attribute__((address_space(3))) float*>(bb[b] + s * 4 * WG_TROW);
              }
            }
          }
        }
        // The fragments this k-step multiplied stay "live" to its end: hipcc otherwise hands their registers -- free from the
        // MFMA that read them last, a few instructions up -- to this k-step's address temporaries and fragment reads, and a
        // write into an operand of an MFMA still in flight waits for it (tools/mfma32_stream_probe.hip: +10 % per MFMA).  Held
        // to here, the reads can only land in the slot the PREVIOUS k-step released.
        {
          const int cur = (s < WG_XS ? WG_NKS - WG_XS + s : s - WG_XS) % WG_RS;      // (the loop is fully unrolled: a constant)
          if ((s < WG_XS && TAIL) || (s >= WG_XS && MAIN))
            asm volatile("" ::"v"(fa[cur][0]), "v"(fa[cur][1]), "v"(fa[cur][2]), "v"(fa[cur][3]), "v"(fb[cur][0]), "v"(fb[cur][1]), "v"(fb[cur][2]), "v"(fb[cur][3]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using T = std::true_type;
    using F = std::false_type;
    step(F{}, T{}, 0);
    for (int i = 1; i < nst; ++i) step(T{}, T{}, i);
    step(T{}, F{}, nst);
    wg_barrier();                                // every wave is done with the ring: it becomes the output staging
    // ---- the finished 64 x 64 block leaves as whole rows: lane (c, mq) holds rows 16 bi + 4 mq + e, column 16 bj + c ----
    float* so = reinterpret_cast<float*>(smem) + wave * 64 * WG_OLD;
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int bj = 0; bj < 4; ++bj)
#pragma unroll
        for (int e = 0; e < 4; ++e) so[(16 * bi + 4 * mq + e) * WG_OLD + 16 * bj + c] = acc[bi][bj][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a wave reads back only what it wrote itself)
    float* out = a.slabs + (int64_t)split * a.slab + (int64_t)(i0 + 64 * wi) * a.J + j0 + 64 * wj;
#pragma unroll 4
    for (int r = mq; r < 64; r += 4) {
      const wg_f32x4 v = *reinterpret_cast<const wg_f32x4*>(so + r * WG_OLD + 4 * c);
      *reinterpret_cast<wg_f32x4*>(out + (int64_t)r * a.J + 4 * c) = v;
    }
    return;
  }

  // ===================================================== SERVICE WAVES =====================================================
  const int sw = wave - 4;
  __builtin_amdgcn_s_setprio(1);
  // raw buffer descriptors that END at the split's last row: rows beyond it read as zero
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dC), 0, (int)((int64_t)m_end * a.ldd * 4), 0x00020000);
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.act), 0, (int)((int64_t)m_end * a.lda * 4), 0x00020000);
  // piece n of a tile = rows 2 n, 2 n + 1: lane l -> row 2 n + (l >> 5), LDS position l & 31 receives chunk (l & 31) ^ 4 (l >> 5)
  [[maybe_unused]] const int vD = (lane >> 5) * (int)a.ldd * 4 + (((lane & 31) ^ (4 * (lane >> 5))) * 16) + i0 * 4;
  [[maybe_unused]] const int vA = (lane >> 5) * (int)a.lda * 4 + (((lane & 31) ^ (4 * (lane >> 5))) * 16) + j0 * 4;
  auto dma_stage = [&](int t) {                  // stage t of this split -> ring slot t % NS; this wave: pieces sw, sw + 4, ... of both tiles
    [[maybe_unused]] char* dst = smem + (t % WG_NS) * WG_STAGE;
    [[maybe_unused]] const int m0 = (sb + t) * WG_BM;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      [[maybe_unused]] const int n = sw + 4 * q;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(dst + n * 1024), 16, vD, (m0 + 2 * n) * (int)a.ldd * 4, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + WG_TILE + n * 1024), 16, vA, (m0 + 2 * n) * (int)a.lda * 4, 0, 0);
#endif
    }
  };
  // bias gradient (tiles of the first output-column block only): lane l sums column 32 sw + (l & 31) of the dC tile over the rows
  // 16 (l >> 5) .. + 15 of every stage in a register; the two row halves meet by a lane exchange at the end.  Odd rows hold
  // the column at c2 ^ 16.
  const bool colsum = a.xsum != nullptr && tj == 0;
  const int c2 = 32 * sw + (lane & 31), rh = lane >> 5;
  float csum = 0.f;
  const unsigned cs_even = (16 * rh) * WG_TROW + c2 * 4, cs_odd = (16 * rh + 1) * WG_TROW + (c2 ^ 16) * 4;

  dma_stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = 0; i <= nst; ++i) {
    wg_barrier();                                // B_i
    if (i + 1 < nst) dma_stage(i + 1);           // (slot of stage i - 2: its last fragment read completed before B_(i-1))
    if (colsum && i < nst) {
      const unsigned base = s_lds + (i % WG_NS) * WG_STAGE;
      float v[16];
#pragma unroll
      for (int r = 0; r < 8; ++r) {              // (LDS reads hipcc does not see: see gemm_wsp32.hip, p32_lds16)
        asm volatile("ds_read_b32 %0, %1" : "=v"(v[2 * r]) : "v"(base + cs_even + 2 * r * WG_TROW));
        asm volatile("ds_read_b32 %0, %1" : "=v"(v[2 * r + 1]) : "v"(base + cs_odd + 2 * r * WG_TROW));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                                            "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#pragma unroll
      for (int r = 0; r < 16; ++r) csum += v[r];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage i + 1 has landed before B_(i+1)
  }
  wg_barrier();                                  // (the matrix waves' "ring is free" barrier)
  if (colsum) {
    const float tot = csum + __shfl_xor(csum, 32);
    if (lane < 32) a.xsum[(int64_t)split * a.xsum_stride + i0 + c2] = tot;
  }
}

}  // namespace

// Returns HIG_OK when the launch was made (the caller then runs the slab reduction), 1 when this kernel does not serve the call.
int hig_wgrad_wsp32_try(const hig_gemm_desc& g, int splits, float* slabs, int64_t slab, float* xsum, int64_t xsum_stride, hipStream_t st) {
  if (!hig_gemm_wsp32_active()) return 1;
  if (g.prec != HIG_PREC_F32 || !g.x_rs || !g.y_rs || g.xf != HIG_XF_NONE || g.epi != HIG_EPI_NONE || splits < 2) return 1;
  if (g.I % 128 || g.J % 128 || g.R < 2048) return 1;
  const int nti = g.I / 128, ntj = g.J / 128;
  if ((int64_t)nti * ntj * splits > 256) return 1;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!(g.ldx % 4 == 0 && g.ldy % 4 == 0 && al(g.X) && al(g.Y) && al(slabs) && slab % 4 == 0)) return 1;
  if ((int64_t)g.R * g.ldx >= (1ll << 29) || (int64_t)g.R * g.ldy >= (1ll << 29)) return 1;
  Wg32Args a;
  a.dC = g.X; a.ldd = g.ldx;
  a.act = g.Y; a.lda = g.ldy;
  a.slabs = slabs; a.slab = slab;
  a.xsum = xsum; a.xsum_stride = xsum_stride;
  a.I = g.I; a.J = g.J; a.M = g.R;
  a.nti = nti; a.ntj = ntj;
  a.splits = splits;
  a.nstage = (g.R + WG_BM - 1) / WG_BM;
  if (a.nstage < splits) return 1;
  hipLaunchKernelGGL(wgrad_wsp32_kernel, dim3(256), dim3(512), 0, st, a);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
