// WEIGHT-STATIONARY bf16 GEMM with SPECIALISED WAVES for gfx950 (MI355X), K = 512:
//   C[i][j] = epi( sum_r X[i][r] * W[j][r] ), bf16 in / out, fp32 accumulate on v_mfma_f32_32x32x16_bf16.
// Serves the nn.Linear layers of the bf16-storage denoiser whose reduce extent is d = 512 and whose row count is the
// frame count M = B.T (codes/models/transformer.py:81-85 stylization out, :108-114 q/k/v, :144 cross-attention query,
// :157-170 FFN linear1 + GELU) and the data-gradient products of the training step over the same shapes.
//
// Why a second weight-stationary kernel (profiles/r05_notes.md section 1).  tools/l2_fetch_probe.hip: a CU pulls 55-58 B/clk
// from its XCD's L2 by LDS-DMA as soon as >= 16 KB are in flight -- TWICE what gemm_ws16.hip's launches achieve -- and
// fragment-shaped register loads only 16 B/clk.  gemm_ws16's waves do everything themselves: each issues its share of
// the X-tile DMA between its MFMAs (60-185 cycles of issue stall per 1-KiB instruction) and runs the epilogue's vector
// arithmetic in the MFMA gaps of the same instruction stream; a 32-row tile costs 3.2-4.4 K cycles against 2 K of matrix
// work, and the weight phase (8-10 K cycles) overlaps nothing.
//
// CDNA4 mapping.  One workgroup of EIGHT waves per CU (two per SIMD, 256 registers each), a 128-column weight panel:
//   * waves 0-3, the MATRIX waves (one per SIMD): wave j keeps the MFMA A-fragments of columns [32 j, 32 j + 32) x K = 512
//     in 128 VGPRs for the workgroup's life; per 32-row X tile it issues 32 ds_read_b128 (X fragments, XOR-swizzled image,
//     conflict-free) and 32 MFMAs and nothing else -- no vector-memory instruction, no epilogue arithmetic.  The last XD
//     k-steps of tile t run BEHIND the barrier that opens tile t + 1, on fragments already in registers, while the first
//     fragment reads of tile t + 1 are in flight: the matrix pipe does not drain at the tile seam.  Two accumulator sets
//     alternate; a finished one is handed over as fp32 through LDS (4 ds_write_b128 per lane, 528-byte rows:
//     conflict-free for the writer and the reader).
//   * waves 4-7, the SERVICE waves (the SIMD partners of the matrix waves): they issue every LDS-DMA of the workgroup
//     (X tile t + 2: one 1-KiB row per instruction; residual / LayerNorm-statistics tile t), and run the epilogue of tile
//     t - 2 from the fp32 hand-off (bias / GELU / SiLU / residual / LayerNorm fold in fp32, bf16 pack, 16-byte write-through
//     stores of whole 256-byte row segments).  Their vector work co-issues with the partner's MFMAs (separate pipes).
//   * ONE s_barrier per tile for all eight waves; DMA completion is the issuing wave's counted vmcnt in front of it.
// LDS (157 KB of 160): X ring 3 x 32 KB, hand-off 2 x 16.5 KB, residual ring 3 x 8 KB, statistics ring 3 x 1 KB.  The
// weight panel comes in once through the same buffers (all four 32-KB slices in flight at once).
// Work split: 256 workgroups; slot w = 32 (block % 8) + block / 8 (blocks sharing an XCD are consecutive in w; speed
// only), column panel w % np, row group w / np: the np workgroups that stream the same X rows sit on one XCD, so a tile
// is fetched from HBM once and by the other panels from that XCD's L2.
#include <stdlib.h>

#include <type_traits>

#include "gemm16_epi.h"
#include "hig_host.h"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef int wsp_i32x4 __attribute__((ext_vector_type(4)));

struct WspArgs {
  const __bf16* X; int64_t ldx;
  const __bf16* W; int64_t ldy;
  __bf16* C; int64_t ldc;
  const __bf16* res; int64_t ldr;
  __bf16* aux; int64_t ldaux;      // EPI_BIAS_GELU: the pre-activation acc + bias as well (nullable)
  const float* bias;
  int I, J;
  int np;        // column panels (J / 128)
  int G;         // row groups (256 / np)
  int ntiles;    // 32-row tiles in all
  float* stats_out;           // XT = 1
  const float* stats_in;      // XT = 2
  const float* colsum;        // XT = 2
  unsigned long long* stamps; // diagnostic (hig_gemm_wsp16_debug_stamps), else NULL
  int prio;                   // s_setprio of the service waves (tuning knob HIG_BF16_WSP_PRIO)
  int dbg;                    // timing ablations (HIG_BF16_WSP_DBG; results are wrong): 1 = no X DMA after the first two tiles,
                              // 2 = no epilogue, 4 = epilogue without global stores
};

unsigned long long* g_wsp_stamps = nullptr;

constexpr int KW = 512, BM = 32, BN = 128;
constexpr int ROWB = KW * 2;             // bytes per X row
constexpr int XBUF = BM * ROWB;          // 32 KB
constexpr int NXB = 3;
constexpr int HROW = BN * 4 + 16;        // fp32 hand-off row, padded: 528 bytes
constexpr int HBUF = BM * HROW;          // 16 896
constexpr int RBUF = BM * BN * 2;        // residual tile: 8 KB
constexpr int NRB = 3;
constexpr int LBUF = BM * 32;            // LayerNorm statistics of a tile's rows: [32][4 panels][2] floats
constexpr int OFF_H = NXB * XBUF;
constexpr int OFF_R = OFF_H + 2 * HBUF;
constexpr int OFF_L = OFF_R + NRB * RBUF;
constexpr int OFF_C = OFF_L + NRB * LBUF;   // one counter word: fragment reads of finished tiles (matrix waves -> service waves)
constexpr int SMEM = OFF_C + 16;
static_assert(SMEM <= 160 * 1024, "LDS budget");
static_assert(2 * HBUF >= XBUF, "the fourth weight slice lands in the hand-off buffers");
constexpr int NKS2 = KW / 32;            // 16 k-steps (of 32 reduce elements) per tile
constexpr int XS = 8;                    // k-steps of X fragments read ahead; the last XS k-steps of a tile run behind the next barrier

__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// s_waitcnt vmcnt(n), n wave-uniform and only known at run time (the tail iterations issue fewer requests)
__device__ __forceinline__ void wait_vm_dyn(int n) {
#define WSP_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    WSP_W(0) WSP_W(1) WSP_W(2) WSP_W(3) WSP_W(4) WSP_W(5) WSP_W(6) WSP_W(7) WSP_W(8) WSP_W(9) WSP_W(10) WSP_W(11) WSP_W(12)
    WSP_W(13) WSP_W(14) WSP_W(15) WSP_W(16) WSP_W(17) WSP_W(18) WSP_W(19) WSP_W(20) WSP_W(21) WSP_W(22) WSP_W(23) WSP_W(24)
    default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
  }
#undef WSP_W
}
template <int POL>
__device__ __forceinline__ void wsp_store16(__amdgpu_buffer_rsrc_t rs, int byte_off, const bf16x8& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const wsp_i32x4 d = __builtin_bit_cast(wsp_i32x4, v);
  if constexpr (POL == 1) __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, 16);        // sc1: write-through
  else if constexpr (POL == 2) __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, 2);    // nt
  else __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, 0);
#endif
}

// XT: 0 plain, 1 LayerNorm-fold producer (also writes row statistics of its rounded outputs), 2 consumer (gemm_ws16.hip)
// AUX: EPI_BIAS_GELU only -- also store the pre-activation
// POL: output stores 0 plain / 1 sc1 (write-through) / 2 nt -- a template parameter: as a run-time switch it put three scalar
//      branches in front of every store of the service waves, which are bound by instruction issue
// DIAG: the diagnostic instances (bit 0: s_memtime stamps + the run-time ablations a.dbg, bit 1: no MFMAs, bit 2: no fragment
//       reads); the product instances (DIAG = 0) carry none of it
template <int EPI, int XT, bool AUX, int POL, int DIAG>
__global__ __launch_bounds__(512, 2) void gemm_wsp16_kernel(const WspArgs a) {
  constexpr bool HAS_RES = epi_has_res(EPI);
  static_assert(XT == 0 || (XT == 1 && EPI == HIG_EPI_BIAS_RES) || (XT == 2 && EPI == HIG_EPI_BIAS), "LayerNorm fold: producer = BIAS_RES, consumer = BIAS");
  static_assert(!AUX || EPI == HIG_EPI_BIAS_GELU, "aux output: GELU epilogue only");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  char* const sX = smem;
  char* const sH = smem + OFF_H;
  [[maybe_unused]] char* const sR = smem + OFF_R;
  [[maybe_unused]] char* const sL = smem + OFF_L;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which tiles --------------------------------------------------------------------------------------------------
  const int w = (blockIdx.x & 7) * 32 + (blockIdx.x >> 3);
  const int panel = w % a.np, rg = w / a.np;
  if (rg >= a.G) return;
  const int tb = (int)((int64_t)rg * a.ntiles / a.G), te = (int)((int64_t)(rg + 1) * a.ntiles / a.G);
  const int nt = te - tb;
  if (nt <= 0) return;
  const int j0 = panel * BN;
  // diagnostics (a.stamps != NULL only): thread 0 (matrix wave 0) writes stamps[block * 16 + k], thread 256 (service wave 4)
  // stamps[4096 + block * 16 + k]
  auto stamp = [&]([[maybe_unused]] int k) {
    if constexpr (!DIAG) return;
    if (a.stamps && (tid == 0 || tid == 256)) {
      unsigned long long tm;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm)::"memory");
      a.stamps[(size_t)(tid ? 4096 : 0) + (size_t)blockIdx.x * 16 + k] = tm;
    }
  };
  int* const scnt = reinterpret_cast<int*>(smem + OFF_C);
  if (tid == 0) *scnt = 0;                       // (published by the weight phase's barriers)
  stamp(0);

  if (wave < 4) {
    // =================================================== MATRIX WAVES ===================================================
    // v_mfma_f32_16x16x32_bf16, FOUR accumulator chains per tile (2 blocks of 16 weight columns x 2 blocks of 16 X rows): a
    // single dependent chain of the 32x32x16 form issues every 45-53 cycles next to a partner wave, not every 32
    // (tools/coissue_probe.hip), and that form needs twice the registers per independent chain.
    const int ln = lane & 15, lg = lane >> 4;
    // LDS image of a 32-row x 512-column bf16 tile: row r at r * 1024, its 16-byte chunk c at position c ^ (r & 15).
    // A k-step s (32 reduce elements) reads chunk 4 s + lg of row 16 b + ln (b = 0, 1):
    //   position = 16 (s >> 2) + ((4 (s & 3) + lg) ^ ln)   ->   xo[s & 3] + 256 (s >> 2) bytes into the row
    int xo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xo[q] = ln * ROWB + 16 * ((4 * q + lg) ^ ln);

    wg_barrier();                                // P1: the four weight slices have landed
    bf16x8 wf[NKS2][2];                          // [k-step][block of 16 columns]
    {
      const char* wb = wave < 3 ? sX + wave * XBUF : sH;
#pragma unroll
      for (int ks = 0; ks < NKS2; ++ks)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) wf[ks][cb] = *reinterpret_cast<const bf16x8*>(wb + cb * 16 * ROWB + xo[ks & 3] + 256 * (ks >> 2));
    }
    wg_barrier();                                // P2: every slice is in registers: the buffers are free
    stamp(1);

    f32x4 acc0[2][2], acc1[2][2];                // [column block][row block], two tiles in flight
    bf16x8 ring[XS][2];                          // X fragments of XS k-steps x 2 row blocks
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc0[e >> 1][e & 1] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[e >> 1][e & 1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // hand-off: lane (ln, lg) holds, per (column block cb, row block rb), columns 32 wave + 16 cb + 4 lg + {0..3} of X row 16 rb + ln
    char* const hmine = sH + ln * HROW + (32 * wave + 4 * lg) * 4;
    const unsigned scnt_lds = (unsigned)(size_t)(__attribute__((address_space(3))) int*)scnt;
    const unsigned sx_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sX;
    auto signal_reads_issued = [&]() {
      if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(scnt_lds), "v"(1) : "memory");
    };

    // ONE fragment ring runs through the whole row range: the reads of k-step s are issued XS k-steps (16 MFMAs, 256 cycles)
    // ahead of the MFMAs that consume them, across tile seams too.  Iteration i (behind barrier B_i) issues the 32 fragment
    // reads of tile i; its first XS k-steps of MFMAs are the LAST XS k-steps of tile i - 1 (TAIL: their fragments were read
    // before the barrier), the other NKS2 - XS the first k-steps of tile i (MAIN).
    // aP: accumulators of tile i, aQ: those of tile i - 1 -- two sets alternate, the call sites swap them (no copies).
    static_assert(NKS2 % XS == 0, "the ring index of a k-step is its number modulo XS in every tile");
    auto step = [&](auto has_tail, auto has_main, f32x4(&aP)[2][2], f32x4(&aQ)[2][2], int i) {
      constexpr bool TAIL = decltype(has_tail)::value, MAIN = decltype(has_main)::value;
      // B_i.  The 2 XS fragment reads that close tile i - 1 stay in flight across it (waiting for them here cost 250-290 cycles
      // per tile: the last read is issued right in front of the barrier): everything older -- the hand-off stores of tile
      // i - 2 -- has completed once only the 2 XS youngest LDS operations are outstanding (a wave's LDS operations complete in
      // order).
      if constexpr (TAIL) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * XS < 15 ? 2 * XS : 15) : "memory");   // (the counter has four bits)
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (DIAG && (a.dbg & 8) && i < 9) stamp(2 + i);
      // four LDS addresses per tile (one per chunk class), the row block (16 KB) and the k-step's 256-byte multiples as immediates.
      // Opaque to hipcc on purpose: it otherwise keeps dozens of loop-invariant offsets in registers and spends a v_add3 per read.
      unsigned xa[4];
      const unsigned xb = sx_lds + ((i + 1) % NXB) * XBUF;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xa[q] = xb + xo[q];
        asm volatile("" : "+v"(xa[q]));
      }
      auto dump = [&]() {                        // hand tile i - 1 over, clear its accumulators for tile i + 1
        char* hp = hmine + ((i - 1) & 1) * HBUF;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            *reinterpret_cast<f32x4*>(hp + rb * 16 * HROW + cb * 64) = aQ[cb][rb];
            aQ[cb][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      };
#pragma unroll
      for (int s = 0; s < NKS2; ++s) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            if constexpr ((DIAG & 2) != 0) {      // (timing ablation: no MFMAs -- the operands stay live)
              asm volatile("" : "+v"(aP[cb][rb]), "+v"(aQ[cb][rb]) : "v"(ring[s % XS][rb]), "v"(wf[s][cb]));
              continue;
            }
            if (s < XS) {
              if constexpr (TAIL) aQ[cb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[NKS2 - XS + s][cb], ring[s % XS][rb], aQ[cb][rb], 0, 0, 0);
            } else {
              if constexpr (MAIN) aP[cb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s - XS][cb], ring[s % XS][rb], aP[cb][rb], 0, 0, 0);
            }
          }
        if constexpr (MAIN) {
          if constexpr ((DIAG & 4) == 0) {       // (timing ablation: no fragment reads)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
              ring[s % XS][rb] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(xa[s & 3] + rb * 16 * ROWB + 256 * (s >> 2));
          }
        }
        if constexpr (TAIL) { if (s == (MAIN ? XS + 1 : XS - 1)) dump(); }   // (MAIN: the tail MFMAs finished two k-steps ago)
        if (s % 2 == 1) __builtin_amdgcn_sched_barrier(0);
      }
      // every fragment read of tile i has been handed to the LDS: the add below is executed behind them (in order), so a
      // service wave that reads 4 (i + 1) here knows the X slot of tile i may be overwritten -- without this wave ever
      // waiting for the reads' data in front of a barrier
      if constexpr (MAIN) signal_reads_issued();
    };
    using T = std::true_type;
    using F = std::false_type;
    // iterations 0 .. nt + 1: i = 0 main only; 1 <= i < nt tail + main; i = nt tail only; i = nt + 1 the barrier alone.
    // Even tiles accumulate in set 0, odd tiles in set 1.
    step(F{}, T{}, acc0, acc1, 0);
    int i = 1;
    for (; i + 1 < nt; i += 2) {
      step(T{}, T{}, acc1, acc0, i);
      step(T{}, T{}, acc0, acc1, i + 1);
    }
    if (i < nt) {                                // (nt even: one more odd tile)
      step(T{}, T{}, acc1, acc0, i);
      ++i;
    }
    // i == nt: the tail of tile nt - 1
    if (nt & 1) step(T{}, F{}, acc1, acc0, nt);
    else step(T{}, F{}, acc0, acc1, nt);
    wg_barrier();                                // B_(nt + 1)
    stamp(12);
    return;
  }

  // ===================================================== SERVICE WAVES =====================================================
  const int sw = wave - 4;
  const unsigned scnt_lds_s = (unsigned)(size_t)(__attribute__((address_space(3))) int*)scnt;
  // the service wave's vector instructions win the SIMD's issue arbitration over its (older) matrix partner, which needs one
  // issue slot in four for its MFMA / ds_read stream (MI355X_MICROARCH: priority, then age)
  if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int c8 = lane & 15, rsub = lane >> 4;    // epilogue: this lane owns columns [8 c8, 8 c8 + 8) of row 8 sw + 4 pass + rsub
  // bias (and the LayerNorm-fold column sums) of this lane's eight columns, in registers for the workgroup's life.  Loaded and
  // waited for (with a wait the compiler sees) BEFORE the first DMA goes out: hipcc then never guards their use with a
  // vmcnt(0) of its own further down, where it would drain the DMA ring.
  float bq[8], cq[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bq[k] = epi_has_bias(EPI) ? a.bias[j0 + 8 * c8 + k] : 0.f;
    cq[k] = XT == 2 ? a.colsum[j0 + 8 * c8 + k] : 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0), the other counters at their maximum
  asm volatile("" ::: "memory");

  // raw (stride 0) buffer descriptors over the whole operands; rows are clamped, so nothing is out of range
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.X), 0, (int)(((int64_t)(a.I - 1) * a.ldx + KW) * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.W), 0, (int)(((int64_t)(a.J - 1) * a.ldy + KW) * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(((int64_t)(a.I - 1) * a.ldc + a.J) * 2), 0x00020000);
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(AUX ? a.aux : a.C, 0, (int)(((int64_t)(a.I - 1) * (AUX ? a.ldaux : a.ldc) + a.J) * 2), 0x00020000);
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(HAS_RES ? a.res : a.X), 0, (int)(((int64_t)(a.I - 1) * (HAS_RES ? a.ldr : a.ldx) + (HAS_RES ? a.J : KW)) * 2), 0x00020000);
  // 32 rows [row0, row0 + 32) of a row-major bf16 matrix with K = 512 columns -> an X-tile-shaped buffer: one 1-KiB row per
  // instruction (this wave: rows sw, sw + 4, ...), LDS position p of row r receives the row's 16-byte chunk p ^ (r & 15)
  auto dma_rows = [&]([[maybe_unused]] __amdgpu_buffer_rsrc_t rs, int ld, int row0, int rmax, [[maybe_unused]] char* dst) {
#pragma unroll
    for (int q = 0; q < BM / 4; ++q) {
      const int r = sw + 4 * q;                  // scalar
      [[maybe_unused]] const int voff = 16 * (lane ^ (r & 15));
      [[maybe_unused]] const int soff = min(row0 + r, rmax) * ld * 2;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + r * ROWB), 16, voff, soff, 0, 0);
#endif
    }
  };
  auto dma_x = [&](int t) { dma_rows(rsX, (int)a.ldx, (tb + t) * BM, a.I - 1, sX + ((t + 1) % NXB) * XBUF); };
  // residual tile t -> sR[t % 3]: [32 rows][256 bytes], linear; instruction n covers rows 4 n .. 4 n + 3
  auto dma_res = [&](int t) {
    if constexpr (HAS_RES) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int n = sw + 4 * q;                // scalar
        [[maybe_unused]] const int voff = (min((tb + t) * BM + 4 * n + rsub, a.I - 1) * (int)a.ldr + j0 + 8 * c8) * 2;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsR, (__attribute__((address_space(3))) void*)(sR + (t % NRB) * RBUF + n * 1024), 16, voff, 0, 0, 0);
#endif
      }
    }
  };
  // LayerNorm statistics of tile t's rows -> sL[t % 3]: [32 rows][4 panels][2] floats = 1 KiB, one instruction (wave 4)
  auto dma_stats = [&](int t) {
    if constexpr (XT == 2) {
      if (sw == 0) {
        const int i = min((tb + t) * BM + (lane >> 1), a.I - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.stats_in + (int64_t)i * 8 + 4 * (lane & 1)),
                                         (__attribute__((address_space(3))) void*)(sL + (t % NRB) * LBUF), 16, 0, 0);
      }
    }
  };

  // ---- the weight panel: slices 0..2 into the X ring, slice 3 into the hand-off buffers, all in flight at once -----------
#pragma unroll
  for (int r = 0; r < 3; ++r) dma_rows(rsW, (int)a.ldy, j0 + 32 * r, a.J - 1, sX + r * XBUF);
  dma_rows(rsW, (int)a.ldy, j0 + 96, a.J - 1, sH);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_barrier();                                  // P1
  wg_barrier();                                  // P2: the matrix waves hold their fragments
  dma_x(0);
  if (nt > 1) dma_x(1);
  if (nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // X(0) has landed
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue of tile e from the fp32 hand-off ---------------------------------------------------------------------------
  constexpr int NST = 2 * (1 + (XT == 1 ? 1 : 0) + (AUX ? 1 : 0));   // vector-memory stores per wave and tile
  // (scalar fp32 arithmetic on purpose -- this file is built with -fno-slp-vectorize: next to the partner wave's MFMAs a
  // v_pk_add_f32 issues every 30-40 cycles, a v_add_f32 every 9: tools/coissue_probe.hip)
  auto epilogue = [&](int e) {
    // all four hand-off reads (and the residual / statistics reads) of the two passes first: one LDS round trip per tile
    f32x4 hh[2][2];
    [[maybe_unused]] u32x4_t rr2[2];
    [[maybe_unused]] f32x4 ls[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = 8 * sw + 4 * p + rsub;
      const float* hp = reinterpret_cast<const float*>(sH + (e & 1) * HBUF + row * HROW) + 8 * c8;
      hh[p][0] = *reinterpret_cast<const f32x4*>(hp);
      hh[p][1] = *reinterpret_cast<const f32x4*>(hp + 4);
      if constexpr (HAS_RES) rr2[p] = *reinterpret_cast<const u32x4_t*>(sR + (e % NRB) * RBUF + row * 256 + 16 * c8);
      if constexpr (XT == 2) {
        const char* lp = sL + (e % NRB) * LBUF + row * 32;
        ls[p][0] = *reinterpret_cast<const f32x4*>(lp);
        ls[p][1] = *reinterpret_cast<const f32x4*>(lp + 16);
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = 8 * sw + 4 * p + rsub;
      const f32x4 h0 = hh[p][0], h1 = hh[p][1];
      float v[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
      if constexpr (XT == 2) {
        float mean, var;
        hig_ln_merge4(ls[p][0], ls[p][1], mean, var);
        const float rstd = rsqrtf(var + 1e-5f), mr = -mean * rstd;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], rstd, fmaf(mr, cq[k], bq[k]));
      } else if constexpr (epi_has_bias(EPI)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bq[k];
      }
      const int ig = min((tb + e) * BM + row, a.I - 1);
      if constexpr (AUX) {
        bf16x8 z8;
#pragma unroll
        for (int k = 0; k < 8; ++k) z8[k] = (__bf16)v[k];
        wsp_store16<POL>(rsA, (ig * (int)a.ldaux + j0 + 8 * c8) * 2, z8);
      }
      if constexpr (HAS_RES) {
        const u32x4_t rr = rr2[p];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float r0 = __builtin_bit_cast(float, rr[k] << 16), r1 = __builtin_bit_cast(float, rr[k] & 0xffff0000u);
          if constexpr (EPI == HIG_EPI_DGELU) { v[2 * k] *= dgelu_bf16(r0); v[2 * k + 1] *= dgelu_bf16(r1); }   // `res` = z of FFN linear1
          else { v[2 * k] += r0; v[2 * k + 1] += r1; }
        }
      }
      bf16x8 o8;
#pragma unroll
      for (int k = 0; k < 8; ++k) o8[k] = (__bf16)epi_act<EPI>(v[k]);
      if (DIAG && (a.dbg & 4)) { asm volatile("" ::"v"(o8)); continue; }
      wsp_store16<POL>(rsC, (ig * (int)a.ldc + j0 + 8 * c8) * 2, o8);
      if constexpr (XT == 1) {
        // LayerNorm fold, producer side: (sum, centred sum of squares) of this row's 128 ROUNDED outputs (the 16 lanes of a
        // row are consecutive: hig_panel_stats16); every wave issues the store (lanes c8 != 0 masked), so the counted
        // waits hold
        float s1, s2;
        hig_panel_stats16(o8, s1, s2);
        float* sp = a.stats_out + ((int64_t)ig * a.np + panel) * 2;
        if (c8 == 0) *reinterpret_cast<float2*>(sp) = make_float2(s1, s2);
      }
    }
  };

  // ---- main loop: iteration i (behind barrier B_i) requests X(i + 2), residual / statistics of tile i, finishes tile i - 2 ---
  // requests of iteration i in issue order: X(i + 2) [8], residual(i) [2], statistics(i) [wave 4: 1], stores of tile i - 2 [NST]
  auto n_x = [&](int i) { return (i + 2 < nt && !(DIAG && (a.dbg & 1))) ? BM / 4 : 0; };
  auto n_r = [&](int i) { return (HAS_RES && i < nt) ? 2 : 0; };
  auto n_s = [&](int i) { return (XT == 2 && i < nt && sw == 0) ? 1 : 0; };
  auto n_st = [&](int i) { return (i >= 2 && i < nt + 2 && !(DIAG && (a.dbg & 6))) ? NST : 0; };
  for (int i = 0; i < nt + 2; ++i) {
    wg_barrier();                                // B_i
    if (i == 4) stamp(0);
    if (i + 2 < nt && !(DIAG && (a.dbg & 1))) {
      // X(i + 2) overwrites the slot of tile i - 1: all four matrix waves have issued its last fragment reads (see there)
      int seen;
      do {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(scnt_lds_s) : "memory");
      } while (__builtin_amdgcn_readfirstlane(seen) < 4 * i);
      dma_x(i + 2);
    }
    if (i == 4) stamp(1);
    if (i < nt) { dma_res(i); dma_stats(i); }
    if (i >= 2 && !(DIAG && (a.dbg & 2))) epilogue(i - 2);
    if (i == 4) stamp(2);
    // before B_(i+1): everything requested in iteration i - 1 ahead of its stores has landed -- X(i + 1), residual(i - 1),
    // statistics(i - 1); younger requests (the stores of iteration i - 1, all of iteration i) may stay in flight
    wait_vm_dyn(n_st(i - 1) + n_x(i) + n_r(i) + n_s(i) + n_st(i));
    if (i == 4) stamp(3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI, int XT, bool AUX>
int launch_wsp(const hig_gemm16_desc& g, hipStream_t st) {
  WspArgs a;
  a.X = static_cast<const __bf16*>(g.X); a.ldx = g.ldx;
  a.W = static_cast<const __bf16*>(g.Y); a.ldy = g.ldy;
  a.C = static_cast<__bf16*>(g.C); a.ldc = g.ldc;
  a.res = static_cast<const __bf16*>(g.res); a.ldr = g.ldr;
  a.aux = static_cast<__bf16*>(g.aux); a.ldaux = g.ldaux;
  a.bias = g.bias;
  a.I = g.I; a.J = g.J;
  a.np = g.J / BN;
  a.G = 256 / a.np;
  a.ntiles = (g.I + BM - 1) / BM;
  a.stats_out = g.row_stats_out;
  a.stats_in = g.row_stats_in;
  a.colsum = g.ln_colsum;
  a.stamps = g_wsp_stamps;
  constexpr int prio = 1;   // (a former tuning knob, fixed at the value that won its A/B)
  a.prio = prio;
  static const int dbg = getenv("HIG_BF16_WSP_DBG") ? atoi(getenv("HIG_BF16_WSP_DBG")) : 0;   // timing ablations (never in a product run)
  a.dbg = dbg;
  // output stores write through (`sc1`, as gemm_ws16.hip: the launch's output otherwise sits dirty in the XCDs' L2s until the
  // end-of-kernel write-back); in-place residual updates (C aliases res: the inference forward's residual stream) keep
  // plain stores
  constexpr int store_policy = 1;   // (a former tuning knob, fixed at the value that won its A/B) (gemm_ws16.hip): 0 plain, else sc1
  const bool plain = (g.res && g.res == g.C) || store_policy == 0;
  const dim3 gr(256), bl(512);
  if constexpr (!AUX && XT == 0 && (EPI == HIG_EPI_BIAS || EPI == HIG_EPI_BIAS_GELU || EPI == HIG_EPI_BIAS_RES)) {
    if (a.stamps || dbg) {                       // diagnostic instances (tools/gemm_wsp16_stamps.py)
      // (the no-MFMA / no-fragment-read ablations of profiles/r05_notes.md section 2 were separate instances, DIAG = 3 / 5; they
      // spilled registers and are no longer built)
      hipLaunchKernelGGL((gemm_wsp16_kernel<EPI, XT, AUX, 1, 1>), gr, bl, 0, st, a);
      HIG_CHECK_LAUNCH();
      return HIG_OK;
    }
  }
  if (plain) hipLaunchKernelGGL((gemm_wsp16_kernel<EPI, XT, AUX, 0, 0>), gr, bl, 0, st, a);
  else hipLaunchKernelGGL((gemm_wsp16_kernel<EPI, XT, AUX, 1, 0>), gr, bl, 0, st, a);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

}  // namespace

// Returns HIG_OK when the launch was made, 1 when this kernel does not serve the shape (the caller goes on to
// gemm_ws16.hip / the tiled kernel), a negative HIG_E* code on error.
int hig_gemm_wsp16_try(const hig_gemm16_desc& g, hipStream_t st) {
  static const int on = getenv("HIG_BF16_WSP") ? atoi(getenv("HIG_BF16_WSP")) : 1;   // tuning knob: 0 = this kernel off
  static const int min_rows = getenv("HIG_BF16_WS_ROWS") ? atoi(getenv("HIG_BF16_WS_ROWS")) : 2048;
  static const int forced_nwj = getenv("HIG_BF16_WS_NWJ") ? atoi(getenv("HIG_BF16_WS_NWJ")) : 0;   // a gemm_ws16 variant is forced
  if (!on || forced_nwj || hig_chip_cus() != 256) return 1;
  if (g.R != KW || g.J % BN != 0 || g.J / BN > 256 || g.I < min_rows) return 1;
  if (g.c_f32 || (g.res && g.res_f32)) return 1;
  auto al = [](const void* p, int n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
  if (!(g.ldc % 8 == 0 && al(g.C, 16))) return 1;
  if ((int64_t)g.I * g.ldx >= (1ll << 30) || (int64_t)g.J * g.ldy >= (1ll << 30) || (g.res && (int64_t)g.I * g.ldr >= (1ll << 30)) ||
      (int64_t)g.I * g.ldc >= (1ll << 30) || (g.aux && (int64_t)g.I * g.ldaux >= (1ll << 30)))
    return 1;
  const bool has_res = epi_has_res(g.epi);
  if (has_res && !(g.ldr % 8 == 0 && al(g.res, 16))) return 1;
  if (g.aux && !(g.epi == HIG_EPI_BIAS_GELU && g.ldaux % 8 == 0 && al(g.aux, 16))) return 1;
  if (g.row_stats_out && !(g.epi == HIG_EPI_BIAS_RES && g.J == g.R && !g.row_stats_in)) return 1;
  if (g.row_stats_in && !(g.epi == HIG_EPI_BIAS && g.ln_colsum)) return 1;
  switch (g.epi) {
    case HIG_EPI_NONE: return launch_wsp<HIG_EPI_NONE, 0, false>(g, st);
    case HIG_EPI_BIAS: return g.row_stats_in ? launch_wsp<HIG_EPI_BIAS, 2, false>(g, st) : launch_wsp<HIG_EPI_BIAS, 0, false>(g, st);
    case HIG_EPI_BIAS_GELU: return g.aux ? launch_wsp<HIG_EPI_BIAS_GELU, 0, true>(g, st) : launch_wsp<HIG_EPI_BIAS_GELU, 0, false>(g, st);
    case HIG_EPI_BIAS_RES: return g.row_stats_out ? launch_wsp<HIG_EPI_BIAS_RES, 1, false>(g, st) : launch_wsp<HIG_EPI_BIAS_RES, 0, false>(g, st);
    case HIG_EPI_BIAS_SILU: return launch_wsp<HIG_EPI_BIAS_SILU, 0, false>(g, st);
    case HIG_EPI_BIAS_RES_SILU: return launch_wsp<HIG_EPI_BIAS_RES_SILU, 0, false>(g, st);
    case HIG_EPI_RES: return launch_wsp<HIG_EPI_RES, 0, false>(g, st);
    case HIG_EPI_DGELU: return launch_wsp<HIG_EPI_DGELU, 0, false>(g, st);
    default: return 1;
  }
}

// Diagnostic: thread 0 of every workgroup writes s_memtime stamps to buf[block * 16 + k] (k: 0 start, 1 weights in registers,
// 2 + i barrier of iteration i (i < 9), 12 end of the matrix waves).  buf: 16 x 256 x 8 bytes; NULL switches it off.
extern "C" int hig_gemm_wsp16_debug_stamps(void* buf) {
  g_wsp_stamps = static_cast<unsigned long long*>(buf);
  return HIG_OK;
}
