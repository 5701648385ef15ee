// WEIGHT-STATIONARY bf16-storage GEMM for gfx950 (MI355X):  C[i][j] = epi( sum_r X[i][r] * W[j][r] ), bf16 in / out,
// fp32 accumulate on v_mfma_f32_32x32x16_bf16.  Serves the nn.Linear layers of the bf16-storage denoiser forward
// (codes/models/transformer.py:81-85,108-114,144-150,168) whose reduce extent is short (K = 256 / 512 / 1024) and
// whose row count is large (M = B.T = 6 272 ... 12 544): every one of them is a short-K problem in which a tiled
// kernel (gemm_bf16.hip) re-stages the same 128 x K weight panel for every 128 rows -- 200 MB of L2 -> LDS operand
// traffic for the 39.5 MB FFN linear1 launch, at the ~30 B/clk a CU can pull from its L2 (profiles/r02_notes.md s.7).
//
// CDNA4 mapping.  The register file of a CU (512 KB) is three times its LDS: a workgroup keeps its WEIGHT PANEL in
// VGPRs for its whole life -- wave w holds the MFMA A-fragments of 32 output columns x KW reduce elements (KW / 4
// VGPRs: 128 at KW = 512) -- and only the activation rows stream: 32-row X tiles go global -> LDS by DMA
// (global_load_lds_dwordx4, XOR swizzle on the per-lane SOURCE address and on the read address), double-buffered as
// WHOLE tiles (32 x K), so the k-loop of a tile is barrier-free: per k-step one ds_read_b128 (the X fragment, shared
// by all waves) and one MFMA.  One barrier per tile.  The workgroup is persistent over the row tiles of ONE column
// panel; bias columns live in registers for its whole life.
// The epilogue of tile t-1 (bias / GELU / SiLU / residual in fp32, bf16 pack, swizzled LDS staging) is interleaved
// with the MFMAs of tile t inside each wave (the two waves of a SIMD alternate on the matrix pipe, which leaves
// each ~56 cycles of vector issue per MFMA), and tile t-2 leaves the staging buffer as whole rows (16 bytes per
// lane) right after the barrier of iteration t.
// K = 1024: the reduce range is split over wave pairs (wave (j, kh) holds columns 32 j, k in [512 kh, +512)); the
// partial accumulators are exchanged through LDS: each wave keeps two of its four accumulator quads, parks the other
// two for its partner, and finishes the kept half -- balanced; one more barrier per tile (the parked halves have one
// buffer: X double-buffered is 128 of the 160 KB).
// Work split: block b runs on XCD b % 8 (observed round-robin; speed only).  XCD x owns a contiguous range of row
// tiles for ALL column panels, so an X tile is fetched from HBM once and by the other panels from that XCD's L2.
#include <stdlib.h>

#include <type_traits>

#include "gemm16_epi.h"
#include "hig_host.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct WsArgs {
  const __bf16* X; int64_t ldx;
  const __bf16* W; int64_t ldy;
  __bf16* C; int64_t ldc;
  const __bf16* res; int64_t ldr;
  const float* bias;
  int I, J;
  int np;        // column panels (J / BN)
  int g;         // row groups per XCD: slot s of an XCD works on panel s % np, row tiles rg, rg + g, ... (rg = s / np)
  int ntiles;    // 32-row tiles in all
  // LayerNorm folded into the NEXT GEMM (XT template parameter): a producer (XT = 1) also writes, per output row and column
  // panel, (sum, centred sum of squares) of its bf16-rounded outputs to stats_out[row][np][2]; a consumer (XT = 2) reads
  // stats_in[row][4][2] of its X rows and computes  rstd (x . W'^T) - rstd mean colsum + bias'  with W' = gamma (.) W.
  float* stats_out;
  const float* stats_in;
  const float* colsum;
  unsigned long long* stamps;   // diagnostic (hig_gemm_ws16_debug_stamps): 16 s_memtime stamps per workgroup, else NULL
  int store_slack;              // 1: the stores of the last iteration may still be in flight at the top of an iteration
  int store_policy;             // output stores: 0 plain (lines stay dirty in the XCD's L2 until the end-of-kernel write-back),
                                // 1 `sc1` (write-through: the bytes leave during the kernel), 2 `nt`
};

typedef int ws_i32x4 __attribute__((ext_vector_type(4)));
// 16-byte output store through the C buffer descriptor with a cache policy (a compiler builtin, not inline asm: the hazard
// recogniser and the wait-count pass must see a 128-bit VMEM store)
__device__ __forceinline__ void ws_store16(__amdgpu_buffer_rsrc_t rsC, int byte_off, const bf16x8& v, int policy) {
#if defined(__HIP_DEVICE_COMPILE__)
  const ws_i32x4 d = __builtin_bit_cast(ws_i32x4, v);
  if (policy == 1) __builtin_amdgcn_raw_buffer_store_b128(d, rsC, byte_off, 0, 16);        // sc1: write-through
  else if (policy == 2) __builtin_amdgcn_raw_buffer_store_b128(d, rsC, byte_off, 0, 2);    // nt
  else __builtin_amdgcn_raw_buffer_store_b128(d, rsC, byte_off, 0, 0);
#endif
}

unsigned long long* g_ws_stamps = nullptr;

// s_waitcnt vmcnt(N) as a BUILTIN (hipcc's wait-count pass sees it and retires the loads it tracks; an asm wait it does
// not see) -- gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14, unused counters at their maximum
template <int N>
__device__ __forceinline__ void ws_wait_vmcnt_visible() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
  asm volatile("" ::: "memory");
}

// KW: reduce elements per wave; KSPLIT: wave pairs splitting K = KW * KSPLIT; NWJ: column slices per workgroup, each
// NCB blocks of 32 columns wide (NCB = 2: a wave holds 64 columns x KW = 256 registers of weights -- one wave per SIMD,
// 512 registers each -- and every X fragment it reads from LDS feeds two MFMAs).
// OCC: workgroups per CU the kernel is laid out for (LDS budget 160 KB / OCC, registers 512 / (OCC waves per SIMD)).
template <int KW, int KSPLIT, int NWJ, int NCB, int EPI, int DBG = 0, int OCC = 1, int XT = 0>
__global__ __launch_bounds__(64 * NWJ * KSPLIT, (NWJ * KSPLIT >= 8 ? 2 : OCC)) void gemm_ws16_kernel(const WsArgs a) {
  static_assert(XT == 0 || (NCB == 1 && KW == 512 && (KSPLIT == 1 || OCC == 1) && (OCC == 1 || XT == 1)), "LayerNorm fold: K = 512 / K = 1024 (two K halves) variants");
  constexpr int LDS_MAX = 160 * 1024 / OCC;
  static_assert(OCC == 1 || NWJ * KSPLIT == 4, "two workgroups per CU: 4-wave variants only");
  constexpr int NW = NWJ * KSPLIT, NT = 64 * NW;
  constexpr int K = KW * KSPLIT;
  constexpr int BM = 32, BN = 32 * NWJ * NCB;
  static_assert(NCB == 1 || KSPLIT == 1, "wide slices: single-split variants only");
  constexpr int NKS = KW / 16;                 // MFMA k-steps per wave and tile
  constexpr int ROWB = K * 2;                  // bytes per X row in LDS
  constexpr int XBUF = BM * ROWB;
  constexpr int NDMA = XBUF / 1024, NQ = NDMA / NW;
  static_assert(NDMA % NW == 0, "DMA instructions must split evenly over the waves");
  constexpr int SROWB = BN * 2;                // bytes per staged output row
  constexpr int SBUF = BM * SROWB;
  constexpr int PBUF = KSPLIT > 1 ? NW * 8 * 64 * 4 : 0;   // parked accumulator halves: [wave][8][64] floats (one buffer: 160 KB of LDS are all there is at K = 1024)
  constexpr int PPR = BN / 8;                  // 16-byte pieces per output row
  constexpr int NPC = BM * PPR / NT;           // pieces per thread in the store pass
  static_assert((BM * PPR) % NT == 0, "store pass split");
  constexpr bool HAS_RES = epi_has_res(EPI);
  // 4-wave workgroups (one wave per SIMD, 512 registers each): X fragments are read XD k-steps ahead into a register ring
  // and the k-loop's order is pinned; 8-wave workgroups (256 registers per wave: the ring spills) leave the k-loop to hipcc
  constexpr bool RING = NW < 8 && OCC == 1;   // (two workgroups per CU: 256 registers again, and the other workgroup's waves hide the LDS latency)
  constexpr int XD = OCC > 1 ? 2 : 4;          // (256 registers per wave with two workgroups per CU: a shorter ring)
  constexpr int GK = OCC > 1 ? 2 : 4;          // k-steps per scheduling group
  // X ring: three tiles where LDS allows (a tile requested in iteration t is waited for at the top of iteration t + 2:
  // one tile in flight across every barrier); two at K = 1024 (64 KB tiles)
  // residual tiles come in by DMA too: two buffers, staged-tile sized -- except in the K-split variant, whose 160 KB are
  // all taken: there the residual of tile t lands IN the staging buffer tile t will be staged in (each lane reads the 8
  // bytes it is about to overwrite), requested once the stores of tile t - 2 have read that buffer (one more barrier)
  constexpr bool RES_INPLACE = HAS_RES && (KSPLIT > 1 || OCC > 1);   // (two workgroups per CU: 80 KB each, none to spare either)
  constexpr int RBUF = HAS_RES && !RES_INPLACE ? SBUF : 0;
  // staging buffers: two (tile t - 1 is staged while tile t - 2 leaves) -- ONE in the K-split LayerNorm-fold consumer, whose
  // 160 KB are otherwise all taken (two 64 KB X tiles, staging, parked halves): the tile staged during k-loop t leaves right
  // behind that k-loop's closing barrier, and the 8 KB hold the consumer's statistics / vectors instead
  constexpr int NSB = (XT == 2 && KSPLIT > 1) ? 1 : 2;
  constexpr int NPAN = K / 128;                 // 128-column panels of an X row (LayerNorm-fold statistics per panel)
  constexpr int STATB = BM * NPAN * 8;          // statistics of one tile's rows: [32][NPAN][2] floats
  constexpr int NXB = (3 * XBUF + NSB * SBUF + 2 * RBUF + PBUF <= LDS_MAX) ? 3 : 2;
  static_assert(2 * XBUF + NSB * SBUF + 2 * RBUF + PBUF <= LDS_MAX, "LDS budget");
  constexpr int NRQ = HAS_RES ? SBUF / 1024 / NW : 0;   // residual DMA instructions per wave and tile
  static_assert(!HAS_RES || SBUF % (1024 * NW) == 0, "residual DMA instructions must split evenly over the waves");
  // bias: in LDS where there is room (sixteen registers less per lane), else in registers for the workgroup's life
  constexpr bool BIAS_LDS = NXB * XBUF + NSB * SBUF + 2 * RBUF + PBUF + BN * 4 <= LDS_MAX;
  static_assert(XT != 2 || BIAS_LDS, "the LayerNorm-fold consumer keeps bias' and the column sums in LDS");
  constexpr int XLDS = XT == 2 ? BN * 4 + 2 * STATB : 0;   // (XT = 1 needs none: the statistics are taken in the store pass)
  static_assert(NXB * XBUF + NSB * SBUF + 2 * RBUF + PBUF + (BIAS_LDS ? BN * 4 : 0) + XLDS <= LDS_MAX, "LDS budget (LayerNorm-fold consumer)");
  __shared__ __attribute__((aligned(1024))) char smem[NXB * XBUF + NSB * SBUF + 2 * RBUF + PBUF + (BIAS_LDS ? BN * 4 : 0) + XLDS];
  [[maybe_unused]] float* const sB = reinterpret_cast<float*>(smem + NXB * XBUF + NSB * SBUF + 2 * RBUF + PBUF);
  [[maybe_unused]] char* const sXT = smem + NXB * XBUF + NSB * SBUF + 2 * RBUF + PBUF + (BIAS_LDS ? BN * 4 : 0);
  [[maybe_unused]] char* const sLn = sXT;                                        // XT = 2: [2][32 rows][NPAN panels][2] floats
  [[maybe_unused]] float* const sC = reinterpret_cast<float*>(sXT + 2 * STATB);  // XT = 2: column sums of W' of this panel
  char* const sX = smem;
  char* const sS = smem + NXB * XBUF;
  [[maybe_unused]] char* const sR = smem + NXB * XBUF + NSB * SBUF;
  [[maybe_unused]] char* const sP = smem + NXB * XBUF + NSB * SBUF + 2 * RBUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: row / buffer arithmetic of the DMA stays scalar
  const int wj = wave % NWJ, kh = wave / NWJ;
  const int lr = lane & 31, lh = lane >> 5;

  // ---- which tiles ----------------------------------------------------------------------------------------------
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int panel = slot % a.np, rg = slot / a.np;
  if (rg >= a.g) return;
  const int tl = (int)((int64_t)xcd * a.ntiles >> 3), th = (int)((int64_t)(xcd + 1) * a.ntiles >> 3);
  const int t0 = tl + rg;
  if (t0 >= th) return;
  const int nt = (th - t0 + a.g - 1) / a.g;
  const int j0 = panel * BN;
  auto stamp = [&](int k) {
    if (a.stamps && tid == 0) {
      unsigned long long tm;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm)::"memory");
      a.stamps[(size_t)blockIdx.x * 16 + k] = tm;
    }
  };
  stamp(0);

  // which accumulator quads this wave finishes: all four, or (K split) two of them: kh = 0 -> quads 0, 1; kh = 1 -> 2, 3.
  // A K-split wave with kh = 1 swaps the halves of its accumulator vector when a tile's k-loop ends, so "slot u" below is
  // real quad qf0 + u for both kinds (a select between two register INDICES would send the vector to scratch memory).
  constexpr int NQF = KSPLIT > 1 ? 2 : 4;
  const int qf0 = KSPLIT > 1 ? 2 * kh : 0;
  // bias of the columns this lane finishes: slot u, element e is column 32 wj + 8 (qf0 + u) + 4 lh + e.  Requested
  // FIRST: the (compiler-visible) wait of the first weight round retires these loads, so that hipcc never guards their
  // use with a vmcnt(0) of its own further down, where it would drain the X ring.
  float bq[BIAS_LDS ? 1 : NCB * 4 * NQF];
  if constexpr (BIAS_LDS) {
    static_assert(BN <= NT, "one thread per bias column");
    if (tid < BN) sB[tid] = epi_has_bias(EPI) ? a.bias[j0 + tid] : 0.f;    // (published by the barriers of the weight rounds)
    if constexpr (XT == 2) { if (tid < BN) sC[tid] = a.colsum[j0 + tid]; }
  } else {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int u = 0; u < NQF; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          bq[(cb * NQF + u) * 4 + e] = epi_has_bias(EPI) ? a.bias[j0 + 32 * (NCB * wj + cb) + 8 * (qf0 + u) + 4 * lh + e] : 0.f;
  }

  // raw (stride 0) buffer descriptors over the whole operands; rows are clamped, so nothing is out of range
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.X), 0, (int)(((int64_t)(a.I - 1) * a.ldx + K) * 2), 0x00020000);
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.W), 0, (int)(((int64_t)(a.J - 1) * a.ldy + K) * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(((int64_t)(a.I - 1) * a.ldc + a.J) * 2), 0x00020000);
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(HAS_RES ? a.res : a.X), 0, (int)(((int64_t)(a.I - 1) * (HAS_RES ? a.ldr : a.ldx) + (HAS_RES ? a.J : K)) * 2), 0x00020000);
  // ---- DMA: instruction n of a tile covers bytes [1024 n, 1024 n + 1024) of the [32][K] image; LDS position p of row r
  // receives the source row's 16-byte chunk p ^ (r & 15) ------------------------------------------------------------
  // 32 rows [row0, row0 + 32) of a row-major bf16 matrix (rows clamped to rmax) -> X-tile-shaped buffer `buf`
  // `buffer_load_dwordx4 ... lds` through a buffer descriptor in SGPRs: the per-lane part of the address is one 32-bit
  // byte offset (the swizzled chunk), the row is a scalar -- no 64-bit per-lane address to build and keep in VGPRs.
  auto dma_one = [&]([[maybe_unused]] __amdgpu_buffer_rsrc_t rs, int ld, int row0, int rmax, int buf, int q) {
    const int n = wave + NW * q;                // scalar
    [[maybe_unused]] int voff, soff;
    if constexpr (ROWB >= 1024) {               // one row (or a 1-KiB part of one) per instruction: lane = chunk position
      constexpr int IPR = ROWB / 1024;
      const int r = n / IPR;
      voff = 16 * ((n % IPR) * 64 + (lane ^ (r & 15)));
      soff = min(row0 + r, rmax) * ld * 2;
    } else {                                    // two 512-byte rows per instruction
      const int r = 2 * n + (lane >> 5);
      voff = (min(row0 + r, rmax) * ld + 8 * ((lane & 31) ^ (r & 15))) * 2;
      soff = 0;
    }
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass cannot type-check the builtin)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sX + buf * XBUF + n * 1024), 16, voff, soff, 0, 0);
#endif
  };
  auto dma_rows = [&](__amdgpu_buffer_rsrc_t rs, int ld, int row0, int rmax, int buf) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) dma_one(rs, ld, row0, rmax, buf, q);
  };
  auto dma_tile = [&](int t, int buf) { dma_rows(rsX, (int)a.ldx, (t0 + t * a.g) * BM, a.I - 1, buf); };

  // read-side offsets: k-step ks reads chunk (2 ks + lh) of row lr, stored at chunk ^ (lr & 15)
  int xo[8];
  {
    const int tsw = (lr & 15) ^ lh;
#pragma unroll
    for (int j = 0; j < 8; ++j) xo[j] = lr * ROWB + kh * (KW * 2) + 16 * ((2 * j) ^ tsw);
  }

  // ---- the weight panel of this wave: 32 columns x KW, as MFMA A-fragments, for the life of the workgroup ----------
  // Fetched through LDS like an X tile (round r = the 32 weight rows of column slice r, whole 1-KiB rows by DMA, NXB
  // rounds in flight), then read as fragments by the wave(s) that own the slice.  Loading the fragments straight from
  // global memory (32 rows x 32 bytes per instruction) took 8.3K cycles to ISSUE and 11K more to land at M = 12 544
  // (tools/gemm_ws16_stamps.py): the address path handles such a fragment-shaped load one 128-byte line at a time.
  bf16x8 wf[NCB][NKS];
  constexpr int NWB = NXB;                                   // X-tile-sized buffers the rounds use
  constexpr int NR = NWJ * NCB;                              // rounds: one per block of 32 columns
  constexpr int RLAST0 = ((NR - 1) / NWB) * NWB;             // last round that uses buffer 0: the first X tile goes there afterwards
#pragma unroll
  for (int r = 0; r < NWB && r < NR; ++r) dma_rows(rsW, (int)a.ldy, j0 + 32 * r, a.J - 1, r);
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    // round r has landed once only the younger requests are outstanding: the rounds behind it, the first X tile
    const int younger = (NWB - 1 < NR - 1 - r ? NWB - 1 : NR - 1 - r) + (r > RLAST0 ? 1 : 0);
    if (younger == 0) ws_wait_vmcnt_visible<0>();
    else if (younger == 1) ws_wait_vmcnt_visible<NQ>();
    else if (younger == 2) ws_wait_vmcnt_visible<2 * NQ>();
    else ws_wait_vmcnt_visible<3 * NQ>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wj == r / NCB) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        wf[r % NCB][ks] = *reinterpret_cast<const bf16x8*>(sX + (r % NWB) * XBUF + xo[ks & 7] + 256 * (ks >> 3));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // the buffer may be overwritten
    asm volatile("" ::: "memory");
    if (r + NWB < NR) dma_rows(rsW, (int)a.ldy, j0 + 32 * (r + NWB), a.J - 1, r % NWB);
    if (r == RLAST0) dma_tile(0, 0);
  }
  stamp(1);

  // a tile's accumulators START at the bias of the columns this wave finishes (K split: the half it parks starts at 0;
  // a kh = 1 wave finishes elements 8 .. 15, which it swaps to the front when the k-loop ends)
  f32x16 acc[NCB], old[NCB];
  auto acc_start = [&]() {
    if constexpr (XT == 2) {                     // (bias' enters after the row scaling)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[0][j] = 0.f;
    } else if constexpr (BIAS_LDS) {
      static_assert(!BIAS_LDS || KSPLIT == 1, "bias in LDS: single-split variants only");
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(sB + 32 * (NCB * wj + cb) + 8 * q + 4 * lh);
          acc[cb][4 * q] = b4.x; acc[cb][4 * q + 1] = b4.y; acc[cb][4 * q + 2] = b4.z; acc[cb][4 * q + 3] = b4.w;
        }
    } else if constexpr (KSPLIT > 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc[0][j] = kh ? 0.f : bq[j]; acc[0][8 + j] = kh ? bq[j] : 0.f; }
    } else {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[cb][j] = bq[cb * 16 + j];
    }
  };
  acc_start();
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) old[cb][e] = 0.f;

  // The epilogue of the tile whose accumulators sit in `old`, one PAIR of elements at a time (pair jp = elements 2 jp,
  // 2 jp + 1 of this wave's finished quads, element j = 4 u + e): the k-loop calls epi_pair(jp) between its MFMAs so that the
  // vector work is spread evenly.  Pairs, because the kernel is bound by instruction issue and v_pk_fma / v_pk_mul / v_pk_add
  // _f32 finish two elements per slot (gemm16_epi.h).
  f32x2 ev[2];
  u32x2 rq = u32x2{0u, 0u};                    // residual of the quad in work: four bf16, raw bits
  [[maybe_unused]] float ln_rstd = 0.f, ln_mr = 0.f;        // XT = 2: rstd and -mean * rstd of this lane's row
  [[maybe_unused]] f32x4 ln_b4 = f32x4{0.f, 0.f, 0.f, 0.f}, ln_c4 = f32x4{0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] const char* lnbuf = nullptr;             //         the tile's statistics in LDS (set per tile)
  auto epi_pair = [&](int jp, char* stg, [[maybe_unused]] const float* parked, [[maybe_unused]] const char* rbuf) {
    const int j = 2 * jp;
    const int cb = j / (4 * NQF), jj = j % (4 * NQF);
    const int u = jj >> 2, e = jj & 3;           // e = 0 or 2
    f32x2 v = {old[cb][jj], old[cb][jj + 1]};
    if constexpr (KSPLIT > 1) v += f32x2{parked[jj * 64], parked[(jj + 1) * 64]};   // (the whole product before any row scaling)
    if constexpr (XT == 2) {
      if (j == 0) {                              // LayerNorm statistics of this lane's row from its NPAN panel partials
        float mean, var;
        if constexpr (NPAN == 4) {
          const f32x4 p0 = *reinterpret_cast<const f32x4*>(lnbuf + lr * 32), p1 = *reinterpret_cast<const f32x4*>(lnbuf + lr * 32 + 16);
          hig_ln_merge4(p0, p1, mean, var);
        } else {                                 // the same pairwise merge over NPAN panels
          f32x4 pp[NPAN / 2];
          float ssum = 0.f, m2 = 0.f;
#pragma unroll
          for (int q = 0; q < NPAN / 2; ++q) {
            pp[q] = *reinterpret_cast<const f32x4*>(lnbuf + lr * (NPAN * 8) + 16 * q);
            ssum += pp[q].x + pp[q].z;
            m2 += pp[q].y + pp[q].w;
          }
          mean = ssum * (1.0f / K);
          float between = 0.f;
#pragma unroll
          for (int q = 0; q < NPAN / 2; ++q) {
            const float d0 = fmaf(pp[q].x, 1.0f / 128.0f, -mean), d1 = fmaf(pp[q].z, 1.0f / 128.0f, -mean);
            between = fmaf(d0, d0, fmaf(d1, d1, between));
          }
          var = fmaf(between, 128.0f, m2) * (1.0f / K);
        }
        ln_rstd = rsqrtf(var + 1e-5f);
        ln_mr = -mean * ln_rstd;
      }
      if (e == 0) {
        ln_b4 = *reinterpret_cast<const f32x4*>(sB + 32 * wj + 8 * (qf0 + u) + 4 * lh);
        ln_c4 = *reinterpret_cast<const f32x4*>(sC + 32 * wj + 8 * (qf0 + u) + 4 * lh);
      }
      const f32x2 tt = __builtin_elementwise_fma(f32x2{ln_mr, ln_mr}, f32x2{ln_c4[e], ln_c4[e + 1]}, f32x2{ln_b4[e], ln_b4[e + 1]});
      v = __builtin_elementwise_fma(v, f32x2{ln_rstd, ln_rstd}, tt);
    }
    if constexpr (HAS_RES) {
      if (e == 0) rq = *reinterpret_cast<const u32x2*>(rbuf + lr * SROWB + 16 * ((4 * (NCB * wj + cb) + qf0 + u) ^ (lr & 15)) + 8 * lh);
      const unsigned w = e < 2 ? rq.x : rq.y;
      const f32x2 r2 = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
      if constexpr (EPI == HIG_EPI_DGELU) v = v * f32x2{dgelu_bf16(r2[0]), dgelu_bf16(r2[1])};   // `res` = z of FFN linear1
      else v += r2;
    }
    ev[e >> 1] = (DBG & 2) ? v : epi_act_pk<EPI>(v);
    if (e == 2) {
      const int pos16 = (4 * (NCB * wj + cb) + qf0 + u) ^ (lr & 15);
      const bf16x4 o4 = bf16x4{(__bf16)ev[0][0], (__bf16)ev[0][1], (__bf16)ev[1][0], (__bf16)ev[1][1]};
      *reinterpret_cast<bf16x4*>(stg + lr * SROWB + 16 * pos16 + 8 * lh) = o4;
    }
  };
  // Residual tile t -> LDS by DMA, in the layout of the staged output tile (row r, 16-byte chunk c at position
  // c ^ (r & 15)): each lane later reads the 8 bytes it is about to overwrite in the staging buffer's twin.  No VGPR
  // destination, hence nothing hipcc could guard with a vmcnt(0) of its own, and nothing it could copy before the data
  // has landed (an inline-asm load into registers was tried: hipcc copied the destination registers BEFORE the wait
  // statement that named them).  Requested as the OLDEST requests of iteration t, so the counted wait at the top of
  // iteration t + 1 retires them while the younger X tile and output stores stay in flight.
  auto dma_res = [&](int t, int buf) {
    if constexpr (HAS_RES) {
      const int i0 = (t0 + t * a.g) * BM;
#pragma unroll
      for (int q = 0; q < NRQ; ++q) {
        const int n = wave + NW * q;            // scalar
        const int o = n * 1024 + lane * 16;
        const int r = o / SROWB, p = (o % SROWB) / 16;
        [[maybe_unused]] const int voff = (min(i0 + r, a.I - 1) * (int)a.ldr + j0 + 8 * (p ^ (r & 15))) * 2;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsR, (__attribute__((address_space(3))) void*)((RES_INPLACE ? sS + buf * SBUF : sR + buf * RBUF) + n * 1024), 16, voff, 0, 0, 0);
#endif
      }
    }
  };
  // s_waitcnt vmcnt(N): everything but the N youngest requests of this wave is done
#define WS_WAIT_ALL_BUT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
  // tile t leaves the staging buffer as whole rows
  auto store_tile = [&](int t, const char* stg) {
    const int i0 = (t0 + t * a.g) * BM;
#pragma unroll
    for (int u = 0; u < NPC; ++u) {
      const int idx = tid + NT * u;
      const int r = idx / PPR, p = idx % PPR;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + r * SROWB + 16 * (p ^ (r & 15)));
      // Rows beyond I go to row I - 1: their X rows (and residual rows) were clamped to it too, so they hold bit-identical
      // values.  No predicate: every wave must issue exactly NPC stores per tile -- the counted waits assume it (a wave
      // whose lanes are all masked would skip the instruction and leave an OLDER request, e.g. a DMA, uncounted).
      const int i = min(i0 + r, a.I - 1);
      ws_store16(rsC, (i * (int)a.ldc + j0 + 8 * p) * 2, v, a.store_policy);
      if constexpr (XT == 1) {
        // LayerNorm fold, producer side: (sum, sum of squared deviations from the panel mean) of this row's 128 ROUNDED
        // outputs (what the consumer will read).  The 16 lanes that hold a row's pieces are consecutive: v_dot2c_f32_bf16
        // on the packed pairs for the sum, four shuffle steps, then the centred squares (hig_panel_stats16); lane 0 of the
        // group writes.  Taken here, in the store pass, and not in the epilogue between the MFMAs (which is bound by
        // instruction issue); no LDS staging.
        static_assert(XT != 1 || PPR == 16, "row statistics: 128-column panels");
        float s1, s2;
        hig_panel_stats16(v, s1, s2);
        if (p == 0) *reinterpret_cast<float2*>(a.stats_out + ((int64_t)i * a.np + panel) * 2) = make_float2(s1, s2);
      }
    }
  };

  // Requests of iteration t, in this order: residual of tile t and stores of tile t - 2 (ahead of the k-loop), then the
  // DMA of tile t + NXB - 1 (spread over the k-loop).  At the top of iteration t + 1 the wave needs the residual and
  // X(t + 1); with three buffers X(t + 1) was requested an iteration earlier, so this iteration's DMA (the youngest NQ
  // requests) may stay in flight.
  bool dma_pend = false;
  if (NXB >= 3 && nt > 1) { dma_tile(1, 1); dma_pend = true; }
  stamp(2);
  for (int t = 0; t < nt; ++t) {
    // (what must have landed here: X(t), requested two iterations ago, and -- residual epilogues -- the residual tile requested
    // in the last one.  Without a residual, the output stores of the last iteration need not be done either: they have read
    // their staging buffer (an LDS read, long retired) and nothing waits for their acknowledgement -- a write-through store
    // takes about as long as an iteration to be acknowledged.  The stores of iteration t - 1 are NPC requests issued ahead of
    // its NQ DMA requests.)
    constexpr bool STORE_SLACK = !HAS_RES && XT != 2 && NXB >= 3;
    if (STORE_SLACK && a.store_slack && dma_pend && t >= 3) WS_WAIT_ALL_BUT(NQ + NPC);
    else if (NXB >= 3 && dma_pend) WS_WAIT_ALL_BUT(NQ);
    else if (NSB == 1 && t >= 2) WS_WAIT_ALL_BUT(NPC);   // (one staging buffer: the NPC stores of tile t - 2 went out behind k-loop t - 1,
    else WS_WAIT_ALL_BUT(0);                              //  younger than everything this iteration needs: they stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // everyone's share has landed; k-loop(t-1) is over everywhere: its buffer, staging(t-1), parked(t-1) are complete
    asm volatile("" ::: "memory");
    if (t < 9) stamp(3 + t);
    if constexpr (!RES_INPLACE) dma_res(t, t & 1);
    if constexpr (XT == 2) {
      // the statistics of tile t's rows: STATB / 1024 DMA instructions of 1 KiB (a row's NPAN x (sum, centred squares) floats are
      // PPS = NPAN / 2 pieces of 16 bytes, 64 / PPS rows per instruction), one per wave
      constexpr int PPS = NPAN / 2;
      if (wave < STATB / 1024) {
        const int i = min((t0 + t * a.g) * BM + wave * (64 / PPS) + lane / PPS, a.I - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.stats_in + (int64_t)i * (2 * NPAN) + 4 * (lane % PPS)),
                                         (__attribute__((address_space(3))) void*)(sLn + (t & 1) * STATB + wave * 1024), 16, 0, 0);
      }
      lnbuf = sLn + ((t + 1) & 1) * STATB;       // statistics of tile t - 1, whose epilogue runs in this iteration
    }
    if constexpr (NSB == 2) { if (t >= 2) store_tile(t - 2, sS + (t & 1) * SBUF); }
    if constexpr (RES_INPLACE) {                 // that staging buffer is free now: the residual of tile t goes there
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      dma_res(t, t & 1);
    }
    dma_pend = t + NXB - 1 < nt;                 // its DMA instructions go out between the MFMAs below
    constexpr bool DMA_IN_LOOP = OCC == 1;       // (two workgroups per CU: issued here in one block -- the other workgroup's
                                                 // waves feed the matrix pipe meanwhile, and the k-loop needs fewer registers)
    if (!DMA_IN_LOOP && dma_pend) dma_tile(t + NXB - 1, (t + NXB - 1) % NXB);
    const int drow0 = (t0 + (t + NXB - 1) * a.g) * BM, dbuf = (t + NXB - 1) % NXB;
    // (a wave stalls ~150-300 cycles on every 1-KiB DMA it issues -- the CU takes ~25 bytes per clock from its L2 --
    // and issued in a block ahead of the k-loop that was 600-1200 cycles per tile in which the wave fed no MFMA; the
    // two waves of a SIMD issue theirs at different k-steps, so one of them keeps the matrix pipe busy)
    const char* xb = sX + (t % NXB) * XBUF;
    char* stg = NSB == 1 ? sS : sS + ((t + 1) & 1) * SBUF;   // staging of tile t-1
    [[maybe_unused]] const char* rbuf = RES_INPLACE ? stg : sR + ((t + 1) & 1) * RBUF;   // its residual
    [[maybe_unused]] const float* parked = reinterpret_cast<const float*>(sP) + (wave ^ NWJ) * 512 + lane;
    // The MFMAs of this tile with the epilogue of the previous one (garbage in, nothing stored, at t = 0) riding in their
    // gaps, element by element.  X fragments are read XD k-steps ahead into a register ring: LDS answers a ds_read_b128
    // in ~130-200 cycles when eight waves read at once, an MFMA issues every 32-64, and hipcc on its own keeps only two
    // reads in flight (4 900 cycles per tile against 2 048 of MFMA work, tools/gemm_ws16_stamps.py).  The k-loop is
    // compiled in four copies (this iteration requests an X tile or not; first or second wave of its SIMD) so that it
    // stays ONE basic block: a runtime test per k-step cut it into blocks of eight MFMAs, each starting with a drained
    // LDS queue.
    auto kloop = [&](auto with_dma, auto second_half) {
      constexpr bool DMA = decltype(with_dma)::value;
      constexpr int DPH = decltype(second_half)::value ? (NKS / NQ) / 2 : 0;
      bf16x8 xr[XD];
      if constexpr (RING) {
#pragma unroll
        for (int i = 0; i < XD; ++i) xr[i] = *reinterpret_cast<const bf16x8*>(xb + xo[i & 7] + 256 * (i >> 3));
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if constexpr (RING) {
          // (DBG: timing ablations of the diagnostic instances -- results are wrong by construction.  1 = no MFMA,
          // 2 = no epilogue arithmetic, 4 = no LDS reads of X fragments, 8 = no DMA inside the k-loop)
          if constexpr (!(DBG & 1)) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][ks], xr[ks % XD], acc[cb], 0, 0, 0);
          } else {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) asm volatile("" : "+v"(acc[cb]) : "v"(wf[cb][ks]), "v"(xr[ks % XD]));
          }
          if constexpr (!(DBG & 4)) {
            if (ks + XD < NKS) xr[ks % XD] = *reinterpret_cast<const bf16x8*>(xb + xo[(ks + XD) & 7] + 256 * ((ks + XD) >> 3));
          }
        } else {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xb + xo[ks & 7] + 256 * (ks >> 3));
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][ks], xf, acc[cb], 0, 0, 0);
        }
        if constexpr (DMA && DMA_IN_LOOP && !(DBG & 8)) {
          if (ks % (NKS / NQ) == DPH) dma_one(rsX, (int)a.ldx, drow0, a.I - 1, dbuf, ks / (NKS / NQ));
        }
        constexpr int NEL = 2 * NQF * NCB;       // epilogue element PAIRS per tile and lane, spread over the NKS k-steps
        if constexpr (NEL >= NKS) {
          static_assert(NEL % NKS == 0 || NEL < NKS, "epilogue pairs per k-step");
#pragma unroll
          for (int i = 0; i < NEL / NKS; ++i) epi_pair(ks * (NEL / NKS) + i, stg, parked, rbuf);
        } else {
          constexpr int STEP = NKS / NEL;
          static_assert(NKS % NEL == 0, "k-steps per epilogue pair");
          if (ks % STEP == STEP - 1) epi_pair(ks / STEP, stg, parked, rbuf);
        }
        // nothing moves across the end of a group of GK k-steps; inside a group hipcc interleaves the MFMAs with the
        // group's epilogue elements (several independent chains: one element alone is a chain of ~14 dependent vector
        // instructions at ~8 cycles each, 4 900 cycles per tile when the elements came one by one)
        if constexpr (RING) { if (ks % GK == GK - 1) __builtin_amdgcn_sched_barrier(0); }
      }
    };
    if (!DMA_IN_LOOP) {
      kloop(std::false_type{}, std::false_type{});
    } else if (wave < NW / 2 || NW < 8) {
      if (dma_pend) kloop(std::true_type{}, std::false_type{}); else kloop(std::false_type{}, std::false_type{});
    } else {
      if (dma_pend) kloop(std::true_type{}, std::true_type{}); else kloop(std::false_type{}, std::true_type{});
    }
    if constexpr (KSPLIT > 1) {
      if (kh) acc[0] = __builtin_shufflevector(acc[0], acc[0], 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7);
      // park the two quads the partner finishes (elements 8 .. 15 after the swap): [wave][8][64] floats, element (4 u + e) of lane l at [(4 u + e)][l];
      // one buffer, so everyone must be done reading the previous tile's parked halves first
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (NSB == 1) { if (t >= 1) store_tile(t - 1, sS); }   // (staged by every wave during this k-loop; next written in the next one, behind the top barrier)
      float* mine = reinterpret_cast<float*>(sP) + wave * 512 + lane;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) mine[(4 * u + e) * 64] = acc[0][8 + 4 * u + e];
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) old[cb] = acc[cb];
    acc_start();
  }
  // ---- drain: epilogue of the last tile, then the last two stores -------------------------------------------------
  WS_WAIT_ALL_BUT(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  stamp(12);
  if constexpr (XT == 2) lnbuf = sLn + ((nt + 1) & 1) * STATB;
  if constexpr (NSB == 2) { if (nt >= 2) store_tile(nt - 2, sS + (nt & 1) * SBUF); }
  {
    char* stg = NSB == 1 ? sS : sS + ((nt + 1) & 1) * SBUF;
    [[maybe_unused]] const float* parked = reinterpret_cast<const float*>(sP) + (wave ^ NWJ) * 512 + lane;
    [[maybe_unused]] const char* rbuf = RES_INPLACE ? stg : sR + ((nt + 1) & 1) * RBUF;
#pragma unroll
    for (int jp = 0; jp < 2 * NQF * NCB; ++jp) epi_pair(jp, stg, parked, rbuf);
  }
  __syncthreads();
  store_tile(nt - 1, NSB == 1 ? sS : sS + ((nt + 1) & 1) * SBUF);
  stamp(13);
  if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 16 + 14] = (unsigned long long)nt;
}

template <int KW, int KSPLIT, int NWJ, int NCB, int EPI, int OCC = 1, int XT = 0>
int launch_ws(const hig_gemm16_desc& g, int slots_per_xcd, hipStream_t st) {
  constexpr int BN = 32 * NWJ * NCB;
  WsArgs a;
  a.X = static_cast<const __bf16*>(g.X); a.ldx = g.ldx;
  a.W = static_cast<const __bf16*>(g.Y); a.ldy = g.ldy;
  a.C = static_cast<__bf16*>(g.C); a.ldc = g.ldc;
  a.res = static_cast<const __bf16*>(g.res); a.ldr = g.ldr;
  a.bias = g.bias;
  a.I = g.I; a.J = g.J;
  a.np = g.J / BN;
  a.g = slots_per_xcd / a.np;
  a.ntiles = (g.I + 31) / 32;
  a.stats_out = g.row_stats_out;
  a.stats_in = g.row_stats_in;
  a.colsum = g.ln_colsum;
  a.stamps = g_ws_stamps;
  // output stores write through (`sc1`): a launch's 13-39 MB of output otherwise sit dirty in the XCDs' L2s until the
  // end-of-kernel write-back, during which nothing runs (same-call A/B at M = 12 544: FFN linear1 25.9 -> 23.4 us, q/k/v 28.5
  // -> 26.4, stylization-out 16.2 -> 15.0; forward B = 64 1.570 -> 1.537 ms; `nt` = 2 is mixed: ca-q 12.6 but FFN linear1 26.9)
  constexpr int store_policy = 1;   // (a former tuning knob, fixed at the value that won its A/B)
  // in-place residual updates (C aliases res: the inference forward's residual stream) keep plain stores: a later tile's
  // residual DMA must see this kernel's own earlier stores in the same L2
  a.store_policy = (g.res && g.res == g.C) ? 0 : store_policy;
  constexpr int store_slack = 0;   // (a former tuning knob, fixed at the value that won its A/B)
  a.store_slack = store_slack;
  hipLaunchKernelGGL((gemm_ws16_kernel<KW, KSPLIT, NWJ, NCB, EPI, 0, OCC, XT>), dim3(8 * slots_per_xcd), dim3(64 * NWJ * KSPLIT), 0, st, a);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

template <int EPI>
int launch_ws_sized(const hig_gemm16_desc& g, int nwj, hipStream_t st) {
  // nwj: 8 = eight waves x 32 columns, 4 = four waves x 32 columns, 2 = four waves x 64 columns (256-column panels),
  // 44 = four waves x 32 columns laid out for TWO workgroups per CU (80 KB of LDS, 256 registers; epilogues without a
  // residual tile only)
  constexpr bool has_res = epi_has_res(EPI);
  if constexpr (!has_res) {
    if (nwj == 44 && g.R == 512 && !g.row_stats_in) return launch_ws<512, 1, 4, 1, EPI, 2>(g, 64, st);
    if (nwj == 44 && g.R == 256) return launch_ws<256, 1, 4, 1, EPI, 2>(g, 64, st);
  } else {
    if (nwj == 44 && g.R == 512) {               // (residual tile lands in the staging buffer: 80 KB of LDS per workgroup)
      if constexpr (EPI == HIG_EPI_BIAS_RES) {
        if (g.row_stats_out) return launch_ws<512, 1, 4, 1, EPI, 2, 1>(g, 64, st);
      }
      return launch_ws<512, 1, 4, 1, EPI, 2>(g, 64, st);
    }
  }
  if (nwj == 44) nwj = 4;
  // LayerNorm folded into the next GEMM (K = 512, K = 1024): the producer writes row statistics, the consumer applies them
  if constexpr (EPI == HIG_EPI_BIAS_RES) {
    if (g.row_stats_out && g.R == 512) return launch_ws<512, 1, 4, 1, EPI, 1, 1>(g, 32, st);
    if (g.row_stats_out && g.R == 1024) return launch_ws<512, 2, 4, 1, EPI, 1, 1>(g, 32, st);
  }
  if constexpr (EPI == HIG_EPI_BIAS) {
    if (g.row_stats_in && g.R == 512) return launch_ws<512, 1, 4, 1, EPI, 1, 2>(g, 32, st);   // (the 8-wave consumer spilled registers: not built)
    if (g.row_stats_in && g.R == 1024) return launch_ws<512, 2, 4, 1, EPI, 1, 2>(g, 32, st);
  }
  // (8 waves x 32 columns: bias-only epilogues only -- with GELU / residual epilogues that variant spills at 256 registers per wave)
  if constexpr (EPI == HIG_EPI_NONE || EPI == HIG_EPI_BIAS) {
    if (g.R == 512 && nwj == 8) return launch_ws<512, 1, 8, 1, EPI>(g, 32, st);
    if (g.R == 256 && nwj == 8) return launch_ws<256, 1, 8, 1, EPI>(g, 32, st);
  }
  if (g.R == 512) return nwj == 2 ? launch_ws<512, 1, 4, 2, EPI>(g, 32, st) : launch_ws<512, 1, 4, 1, EPI>(g, 32, st);
  if (g.R == 256) return launch_ws<256, 1, 4, 1, EPI>(g, 32, st);
  return launch_ws<512, 2, 4, 1, EPI>(g, 32, st);   // K = 1024: 8 waves = 4 column slices x 2 halves of the reduce range
}

}  // namespace

static inline bool has_res_epi(int epi) { return epi_has_res(epi); }

// Returns HIG_OK when the launch was made, 1 when this kernel does not serve the shape (the caller falls back to the
// tiled kernel), a negative HIG_E* code on error.
int hig_gemm_ws16_try(const hig_gemm16_desc& g, hipStream_t st) {
  static const int ws_on = getenv("HIG_BF16_WS") ? atoi(getenv("HIG_BF16_WS")) : 1;          // tuning knob: 0 = tiled kernel only
  static const int forced_nwj = getenv("HIG_BF16_WS_NWJ") ? atoi(getenv("HIG_BF16_WS_NWJ")) : 0;   // 4 / 8
  static const int min_rows = getenv("HIG_BF16_WS_ROWS") ? atoi(getenv("HIG_BF16_WS_ROWS")) : 2048;
  const bool fold = g.row_stats_out || g.row_stats_in;
  if (fold) {   // only this kernel implements the LayerNorm fold: the caller checks hig_gemm_ws16_lnfold_ok() first
    // (a producer's rows are the consumer's X rows: J of the one = K of the other, 512 or 1024)
    const bool ok = ws_on && !g.c_f32 && !(g.res && g.res_f32) && (g.R == 512 || g.R == 1024) && g.I >= min_rows &&
                    (g.row_stats_out ? (g.epi == HIG_EPI_BIAS_RES && g.J == g.R && !g.row_stats_in)
                                     : (g.epi == HIG_EPI_BIAS && g.ln_colsum && g.J % 128 == 0));
    if (!ok) return hig_set_error(HIG_EUNSUPPORTED, "hig_gemm_bf16: LayerNorm-fold operands on a shape the weight-stationary kernel does not serve");
  }
  // `decline`: this kernel does not serve the call.  Without fold operands the caller falls back to the tiled / few-row
  // kernels; WITH them it must not (those kernels know nothing of row_stats_* / ln_colsum: a producer would silently skip
  // the statistics, a consumer would multiply un-normalised rows by W'), so every exit below is an error then.
  auto decline = [&](const char* why) -> int {
    return fold ? hig_set_error(HIG_EUNSUPPORTED, "hig_gemm_bf16: LayerNorm-fold operands, but %s", why) : 1;
  };
  if (!ws_on) return decline("the weight-stationary kernel is switched off (HIG_BF16_WS=0)");
  // the kernel's work split is compiled for 8 XCDs x 32 CUs (block b -> XCD b & 7, 32 or 64 slots per XCD): another
  // partitioning of the chip gets the tiled kernel, whose grid follows hig_chip_cus()
  if (hig_chip_cus() != 256) return decline("the device does not report 256 compute units (8 XCDs x 32)");
  if (g.c_f32 || (g.res && g.res_f32)) return decline("fp32 output / residual");
  if (!(g.R == 256 || g.R == 512 || g.R == 1024)) return decline("reduce extent not in {256, 512, 1024}");
  if (g.I < min_rows) return decline("too few rows");
  auto al = [](const void* p, int n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
  if (!(g.ldc % 8 == 0 && al(g.C, 16))) return decline("C not 16-byte aligned / ldc not a multiple of 8");
  if ((int64_t)g.I * g.ldx >= (1ll << 30) || (int64_t)g.J * g.ldy >= (1ll << 30) || (g.res && (int64_t)g.I * g.ldr >= (1ll << 30)) ||
      (int64_t)g.I * g.ldc >= (1ll << 30))
    return decline("operand beyond the 32-bit byte offsets of the DMA descriptors");
  const bool has_res = has_res_epi(g.epi);
  if (has_res && !(g.ldr % 4 == 0 && al(g.res, 8))) return decline("residual not 8-byte aligned / ldr not a multiple of 4");
  // columns per CU: the weight panel is fetched once per workgroup (cols x K x 2 bytes at the CU's ~30 B/clk), the X
  // rows once per panel -- the sum is smallest near cols = sqrt(M N / 256)
  // (measured, tools/gemm16_bench.py: the 8-wave variant wins for the wide bias-only launches -- q/k/v at M = 12 544: 28 us
  // against 38 -- ; with a GELU or residual epilogue it spills registers at 256 per wave and the 4-wave variant wins)
  int nwj = 8;
  if (g.R == 1024) nwj = 4;
  else if ((int64_t)g.I * g.J < (int64_t)256 * 192 * 192) nwj = 4;
  else if (!(g.epi == HIG_EPI_NONE || g.epi == HIG_EPI_BIAS)) nwj = 4;
  // GELU: every wave is bound by instruction issue (the erf arithmetic alone is 13 instructions per output, ~4 cycles
  // each from one wave); two 4-wave workgroups per CU put a second, independent wave on every SIMD: FFN linear1 at
  // M = 12 544 28.5 -> 25.2 us, at M = 6 272 17.2 -> 15.1 us
  if (g.epi == HIG_EPI_BIAS_GELU && g.R != 1024) nwj = 44;
  // residual epilogues at K = 512: two workgroups per CU from 8 192 rows up (same-call A/B, forward: B = 64 1.630 -> 1.613 ms,
  // B = 512 8.72 -> 8.55 ms; B = 32 1.056 -> 1.066: the 3-4 tiles of a workgroup there are too few to share a CU)
  constexpr int res44 = 8192;   // (a former tuning knob, fixed at the value that won its A/B): rows from which ... (0 = never)
  if (has_res_epi(g.epi) && g.R == 512 && res44 > 0 && g.I >= res44) nwj = 44;
  else if (g.row_stats_out) nwj = 4;            // (the statistics are per 128-column panel)
  if (g.row_stats_out) { if (forced_nwj == 44) nwj = 44; }
  else if (forced_nwj == 4 || (forced_nwj == 8 && g.R != 1024) || (forced_nwj == 2 && g.R == 512) || (forced_nwj == 44 && g.R != 1024)) nwj = forced_nwj;
  if (nwj == 8 && !(g.epi == HIG_EPI_NONE || g.epi == HIG_EPI_BIAS)) nwj = 4;   // (a forced 8-wave variant: bias-only epilogues only)
  if (nwj == 8 && g.row_stats_in) nwj = 4;
  const int bn = (nwj == 4 || nwj == 44) ? 128 : 256;
  if (g.J % bn != 0) {
    if (g.J % 128 == 0) nwj = 4; else return decline("J not a multiple of 128");
  }
  if (g.J / ((nwj == 4 || nwj == 44) ? 128 : 256) > 32) return decline("more than 32 column panels");
  switch (g.epi) {
    case HIG_EPI_NONE: return launch_ws_sized<HIG_EPI_NONE>(g, nwj, st);
    case HIG_EPI_BIAS: return launch_ws_sized<HIG_EPI_BIAS>(g, nwj, st);
    case HIG_EPI_BIAS_GELU: return launch_ws_sized<HIG_EPI_BIAS_GELU>(g, nwj, st);
    case HIG_EPI_BIAS_RES: return launch_ws_sized<HIG_EPI_BIAS_RES>(g, nwj, st);
    case HIG_EPI_BIAS_SILU: return launch_ws_sized<HIG_EPI_BIAS_SILU>(g, nwj, st);
    case HIG_EPI_BIAS_RES_SILU: return launch_ws_sized<HIG_EPI_BIAS_RES_SILU>(g, nwj, st);
    case HIG_EPI_RES: return launch_ws_sized<HIG_EPI_RES>(g, nwj, st);
    case HIG_EPI_DGELU: return launch_ws_sized<HIG_EPI_DGELU>(g, nwj, st);
    default: return decline("epilogue not built for this kernel");
  }
}

// Diagnostic: thread 0 of every workgroup of the weight-stationary kernel writes s_memtime stamps to buf[block * 16 + k]
// (k: 0 start, 1 weights / first X tile requested, 2 landed, 3 + t barrier of iteration t (t < 9), 12 drain barrier, 13
// end, 14 = number of tiles).  buf must hold 16 x 256 x 8 bytes; NULL switches it off.  Never part of a timed run.
extern "C" int hig_gemm_ws16_debug_stamps(void* buf) {
  g_ws_stamps = static_cast<unsigned long long*>(buf);
  return HIG_OK;
}

// Can a d-wide LayerNorm in front of a GEMM over `rows` rows be folded into the weight-stationary kernels (the producer of
// the rows writes their statistics, the consumer applies them)?  HIG_LNFOLD=0 switches it off.
bool hig_gemm_ws16_lnfold_ok(int64_t rows, int d) {
  static const int on = getenv("HIG_LNFOLD") ? atoi(getenv("HIG_LNFOLD")) : 1;            // tuning knob
  static const int ws_on = getenv("HIG_BF16_WS") ? atoi(getenv("HIG_BF16_WS")) : 1;
  static const int min_rows = getenv("HIG_BF16_WS_ROWS") ? atoi(getenv("HIG_BF16_WS_ROWS")) : 2048;
  static const int forced_nwj = getenv("HIG_BF16_WS_NWJ") ? atoi(getenv("HIG_BF16_WS_NWJ")) : 0;
  static const int on1024 = getenv("HIG_LNFOLD1024") ? atoi(getenv("HIG_LNFOLD1024")) : 1;   // tuning knob
  // the same predicates as hig_gemm_ws16_try / hig_gemm_wsp16_try, so that a fold is only chosen where one of those kernels will
  // accept it (they turn a decline into an error once fold operands are present): the 8 x 32 chip geometry their work split
  // is compiled for, and the 32-bit byte offsets of their DMA descriptors for the widest consumer (q/k/v: 3 d columns)
  return on && ws_on && !forced_nwj && (d == 512 || (d == 1024 && on1024)) && rows >= min_rows && hig_chip_cus() == 256 &&
         rows * (int64_t)3 * d < (1ll << 30);
}
