// WEIGHT-STATIONARY exact-fp32 GEMM with SPECIALISED WAVES for gfx950 (MI355X), K = 256, 512 or 1024:
//   C[i][j] = epi( sum_r X[i][r] * W[j][r] ), fp32 in / out, v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate).
// Serves the nn.Linear layers of the fp32 denoiser whose reduce extent is d = 512 or ff = 1024 and whose row count is the
// frame count M = B.T (codes/models/transformer.py:81-85 stylization out, :108-114 q/k/v, :144 cross-attention query,
// :157-170 FFN linear1 + GELU / linear2) and the data-gradient products of the fp32 training step over the same shapes
// (autograd of the same lines, through the transposed weight copies the backward already keeps).
//
// Why (profiles/r06_notes.md section 1).  The tiled kernel of gemm.hip (64 x 64 tiles, four waves that each stage, multiply
// and run the epilogue) holds the matrix pipe 67 % busy: every wave fetches BOTH operands through registers into LDS (16 B/clk
// per CU from L2 at full MFMA rate), pays a workgroup barrier per 16 MFMAs, and its epilogue's vector arithmetic waits behind
// three other workgroups' MFMAs.  Every GEMM of this model has a short reduce range and many rows, so the weight panel of a
// block of output columns is small (64 columns x K = 512 x 4 B = 128 KB = the 512 fragment registers of four waves) and only
// the rows need to stream: the structure of gemm_wsp16.hip, with the 16x slower fp32 MFMA leaving every other pipe idle.
//
// CDNA4 mapping.  One workgroup of EIGHT waves per CU (two per SIMD, <= 256 registers each):
//   * waves 0-3, the MATRIX waves (one per SIMD): wave (c, h) keeps the MFMA A-fragments of 32 weight columns x 256 reduce
//     elements in 128 VGPRs for its whole row range -- K = 512: a 64-column panel, c = column half, h = K half; K = 1024: a
//     32-column panel, h = K quarter.  Per 16-row X tile it issues 16 ds_read_b128 (X fragments: XOR-swizzled image,
//     conflict-free) and 128 MFMAs (two independent 64-long chains: v_mfma_f32_16x16x4_f32 issues every 32 cycles, a
//     dependent one after 40) and nothing else.  The last XS k-steps of tile t run BEHIND the barrier that opens tile t + 1,
//     on fragments already in registers, while the first fragment reads of tile t + 1 are in flight.  Two accumulator sets
//     alternate; a finished one is handed over through LDS, one plane per K part (2 ds_write_b128 per lane).
//   * waves 4-7, the SERVICE waves (the SIMD partners): every LDS-DMA of the workgroup (X tiles LOOK tiles ahead: 1 KiB per
//     instruction; residual / LayerNorm-statistics tile t) and the epilogue of tile t - 2 from the hand-off: the K parts
//     summed in a fixed order (bitwise reproducible, independent of the batch split), bias / GELU / residual / gelu' /
//     LayerNorm fold, 16-byte stores of whole 256- (128-)byte row segments.
//   * ONE s_barrier per tile for all eight waves (4096 matrix cycles); DMA completion is the issuing wave's counted vmcnt.
// LDS (all 160 KB): X ring 128 KB (K = 512: 4 x 32 KB; K = 1024: 2 x 64 KB), hand-off 2 x NH planes, residual ring 3 x 4 KB,
// statistics ring 3 x 1 KB.  The weight panel comes in once per segment through the same buffers.
// Work split: 256 workgroups; slot w = 32 (block % 8) + block / 8 (blocks sharing an XCD are consecutive in w; speed only).
// The J / BN column panels are cut into at most three SEGMENTS of 2^k panels (24 = 16 + 8); within a segment of n panels
// workgroup w owns panel w % n and row group w / n of 256 / n: the n workgroups that stream the same X rows sit on one XCD.
// Every workgroup walks every segment (weights reloaded in between): equal work whatever J.
#include <stdlib.h>

#include <type_traits>

#include "hig_common.h"
#include "hig_host.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int p32_i32x4 __attribute__((ext_vector_type(4)));

constexpr int P32_MAXSEG = 3;
struct Wsp32Args {
  const float* X; int64_t ldx;
  const float* W; int64_t ldy;
  float* C; int64_t ldc;
  const float* res; int64_t ldr;   // residual rows (EPI_BIAS_RES / EPI_RES) or the pre-activation z (EPI_DGELU)
  float* aux; int64_t ldaux;       // EPI_BIAS_GELU: the pre-activation acc + bias as a second output (nullable)
  const float* bias;
  int I, J;
  int ntiles;                      // 16-row tiles in all
  int nseg, seg_p0[P32_MAXSEG], seg_np[P32_MAXSEG];
  float* stats_out;                // XT = 1
  const float* stats_in;           // XT = 2
  const float* colsum;             // XT = 2
  unsigned long long* stamps;      // diagnostic (hig_gemm_wsp32_debug_stamps), else NULL
  int prio;                        // s_setprio of the service waves
  int dbg;                         // timing ablations (HIG_F32_WSP_DBG; results are wrong): 1 = no X DMA after the first tiles,
                                   // 2 = no epilogue, 4 = epilogue without global stores
};

unsigned long long* g_p32_stamps = nullptr;
long long g_p32_launches = 0;       // launches made by this file since the library was loaded (hig_gemm_wsp32_launches: tests)

__device__ __forceinline__ void p32_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// s_waitcnt vmcnt(n), n wave-uniform and only known at run time (the first and last iterations issue fewer requests)
__device__ __forceinline__ void p32_wait_vm(int n) {
#define P32_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    P32_W(0) P32_W(1) P32_W(2) P32_W(3) P32_W(4) P32_W(5) P32_W(6) P32_W(7) P32_W(8) P32_W(9) P32_W(10) P32_W(11) P32_W(12)
    P32_W(13) P32_W(14) P32_W(15) P32_W(16) P32_W(17) P32_W(18) P32_W(19) P32_W(20) P32_W(21) P32_W(22) P32_W(23) P32_W(24)
    P32_W(25) P32_W(26) P32_W(27) P32_W(28) P32_W(29) P32_W(30) P32_W(31) P32_W(32) P32_W(33) P32_W(34) P32_W(35) P32_W(36)
    P32_W(37) P32_W(38) P32_W(39) P32_W(40) P32_W(41) P32_W(42) P32_W(43) P32_W(44) P32_W(45) P32_W(46) P32_W(47) P32_W(48)
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
#undef P32_W
}
template <int POL>
__device__ __forceinline__ void p32_store16(__amdgpu_buffer_rsrc_t rs, int byte_off, const f32x4& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const p32_i32x4 d = __builtin_bit_cast(p32_i32x4, v);
  if constexpr (POL == 1) __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, 16);        // sc1: write-through
  else __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, 0);
#endif
}
// LDS reads of the SERVICE waves, opaque to hipcc.  A C++ load from LDS behind a pending `buffer_load ... lds` makes the compiler
// wait vmcnt(0) first (the DMA may alias the read): the epilogue then starts only after the X tile requested in the same iteration
// has landed, and the service wave -- DMA issue, landing, epilogue, one after the other -- is late at every barrier (found in the
// disassembly; profiles/r06_notes.md section 1).  What the epilogue reads was published by a barrier (hand-off) or by the counted
// vmcnt of an earlier iteration (residual, statistics).  The wait names the loaded registers as operands, so no use moves above it.
__device__ __forceinline__ f32x4 p32_lds16(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t p32_lds8(unsigned addr) {
  f32x2_t v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7: below fp32 GELU rounding for |x| >~ 0.1, at most 1e-7 absolute elsewhere):
// one exp, one rcp, ten other vector instructions instead of libm erff's ~40 with a branch -- the service waves issue a vector
// instruction every 8-16 cycles next to their partner's MFMAs (tools/coissue32_probe.hip), so the epilogue's instruction count
// is what makes them late at the tile barrier
__device__ __forceinline__ float p32_erf_abs(float ax, float& ex) {   // erf(ax / sqrt 2), ex = exp(-ax^2 / 2), ax >= 0
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float zz = ax * (0.70710678118654752440f * 1.2011224087864498f);   // ax / sqrt 2 * sqrt(log2 e)
  ex = __builtin_amdgcn_exp2f(-(zz * zz));
  return fmaf(-p, ex, 1.0f);
}
__device__ __forceinline__ float p32_gelu(float x) {
  float ex;
  const float e = p32_erf_abs(fabsf(x), ex);
  return 0.5f * x * (1.0f + copysignf(e, x));
}
__device__ __forceinline__ float p32_dgelu(float z) {   // Phi(z) + z phi(z)
  float ex;
  const float e = p32_erf_abs(fabsf(z), ex);
  return fmaf(z * 0.39894228040143267794f, ex, 0.5f + copysignf(0.5f * e, z));
}
__host__ __device__ constexpr bool p32_has_bias(int e) { return e == HIG_EPI_BIAS || e == HIG_EPI_BIAS_GELU || e == HIG_EPI_BIAS_RES; }
__host__ __device__ constexpr bool p32_has_res(int e) { return e == HIG_EPI_BIAS_RES || e == HIG_EPI_RES || e == HIG_EPI_DGELU; }

// COUNT pieces of 1 KiB (piece n = first + STEP q: part n % PPR of row n / PPR) of the 16 rows [row0, row0 + 16) of a row-major
// fp32 matrix -> an X-tile-shaped LDS buffer: LDS position p of row r receives the row's 16-byte chunk p ^ (r & 15).  Rows are
// clamped to rmax (nothing is out of range).
template <int PPR, int COUNT, int STEP>
__device__ __forceinline__ void p32_dma_rows([[maybe_unused]] __amdgpu_buffer_rsrc_t rs, int ld, int row0, int rmax, [[maybe_unused]] char* dst, int first, int lane) {
#pragma unroll
  for (int q = 0; q < COUNT; ++q) {
    const int n = first + STEP * q;              // scalar
    const int r = n / PPR, part = n % PPR;
    [[maybe_unused]] const int voff = 16 * (lane ^ (r & 15));
    [[maybe_unused]] const int soff = min(row0 + r, rmax) * ld * 4 + part * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + n * 1024), 16, voff, soff, 0, 0);
#endif
  }
}

// KW: reduce extent (512 / 1024).  XT: 0 plain, 1 LayerNorm-fold producer (EPI_BIAS_RES: also writes the statistics of each
// output row's 64-column panel), 2 consumer (EPI_BIAS: rows arrive un-normalised with their panel statistics) -- the formats of
// hig_gemm_desc.row_stats_out / row_stats_in (include/hig.h), KW = 512 only (BN = 64 = one statistics panel).
// AUX: EPI_BIAS_GELU only -- also store the pre-activation.  POL: output stores 0 plain / 1 sc1 (write-through).
// DIAG: the diagnostic instance (s_memtime stamps + the run-time ablations a.dbg); the product instances carry none of it -- a
//       scalar branch between two MFMAs is an instruction-fetch bubble of its own
template <int KW, int EPI, int XT, bool AUX, int POL, bool DIAG>
__global__ __launch_bounds__(512, 2) void gemm_wsp32_kernel(const Wsp32Args a) {
  constexpr int BM = 16;
  constexpr int ROWB = KW * 4;               // bytes per X row
  constexpr int XBUF = BM * ROWB;            // 32 / 64 KB
  constexpr int RING = 128 * 1024;
  constexpr int NXB = RING / XBUF;           // 4 / 2
#ifndef P32_LOOK
#define P32_LOOK 1
#endif
  // X tiles requested ahead.  One is enough (a tile is 4096 matrix cycles) and measured best: with two or three tiles in flight
  // the service waves' vector-memory issue backs up behind the pending LDS-DMA (profiles/r06_notes.md section 1)
  constexpr int LOOK = NXB - 1 < P32_LOOK ? NXB - 1 : P32_LOOK;
  // tile t lives in ring slot (t + NXB - 1) % NXB: X(i + LOOK), requested behind barrier B_i, overwrites the slot of tile
  // i + LOOK - NXB.  When that is tile i - 2 or older, its last fragment read completed before B_(i-1); when it is tile i - 1
  // (K = 1024: two slots), the matrix waves count their issued reads in LDS and the service waves poll that counter.
  constexpr bool NEED_CNT = NXB < LOOK + 2;
  constexpr int NH = KW / 256;               // K parts (matrix waves per column block): 1 / 2 / 4
  constexpr int NC = 4 / NH;                 // blocks of 32 columns
  constexpr int BN = 32 * NC;                // panel width: 128 / 64 / 32
  constexpr int NKS = 16;                    // k-steps (16 reduce elements) per wave and tile
  constexpr int XS = 3;                      // k-steps of X fragments read ahead; the last XS of a tile run behind the next barrier
  constexpr int RS = XS + 1;                 // fragment ring: the read of k-step s goes into the slot k-step s - RS left ONE k-step ago (a read
                                             // into the slot the MFMAs just issued still use waits for them: +10 % per MFMA, tools/mfma32_stream_probe.hip)
  constexpr int HROW = BN * 4 + 16;          // hand-off row, padded: 272 / 144 bytes
  constexpr int HPL = BM * HROW;             // one K part's plane
  constexpr int HBUF = NH * HPL;             // one tile's hand-off
  constexpr int RBUF = 4096, NRB = 3;        // residual tile: 1 KiB per service wave
  constexpr int LBUF = 1024;                 // LayerNorm statistics of a tile's rows: [16][8 panels][2]
  constexpr int OFF_H = RING;
  constexpr int OFF_R = OFF_H + 2 * HBUF;
  constexpr int OFF_L = OFF_R + NRB * RBUF;
  constexpr int SMEM = OFF_L + (XT == 2 ? NRB * LBUF : 0);
  constexpr int OFF_C = OFF_H + BN * 4;      // the counter word sits in the padding of the first hand-off row
  constexpr int NSL = BN / 16;               // weight slices of 16 columns (one X-tile-shaped buffer each): 4 / 2
  constexpr int PPR = ROWB / 1024;           // 1-KiB DMA pieces per row: 2 / 4
  constexpr int NXI = BM * PPR / 4;          // X DMA instructions per service wave and tile: 8 / 16
  constexpr int QPR = BN / 4;                // float4 per output row: 32 / 16 / 8
  constexpr int RPP = QPR > 16 ? 64 / QPR : 4;   // rows of its four a service wave finishes per epilogue pass (K = 256: two passes)
  constexpr int NPASS = 4 / RPP;
  constexpr bool HAS_RES = p32_has_res(EPI);
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(XT == 0 || (KW == 512 && ((XT == 1 && EPI == HIG_EPI_BIAS_RES) || (XT == 2 && EPI == HIG_EPI_BIAS))), "LayerNorm fold: K = 512, producer = BIAS_RES, consumer = BIAS");
  static_assert(!AUX || EPI == HIG_EPI_BIAS_GELU, "aux output: GELU epilogue only");
  static_assert(KW != 256 || (!HAS_RES && XT == 0 && !AUX), "K = 256 (the text-side key/value projection): plain / bias epilogue only");
  static_assert(NKS % RS == 0, "the ring index of a k-step is its number modulo RS in every tile");
  __shared__ __attribute__((aligned(1024))) char smem[160 * 1024];
  char* const sX = smem;
  char* const sH = smem + OFF_H;
  [[maybe_unused]] char* const sR = smem + OFF_R;
  [[maybe_unused]] char* const sL = smem + OFF_L;
  int* const scnt = reinterpret_cast<int*>(smem + OFF_C);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = (blockIdx.x & 7) * 32 + (blockIdx.x >> 3);
  // K = 512: the fourth weight slice lands behind the ring (hand-off + residual + statistics buffers = 32 KB), so that ring
  // slot 3 can take X tile 0 while the weights are still in flight
  constexpr bool W_BEHIND = KW == 512;
  static_assert(W_BEHIND ? (NSL == NXB && OFF_H + XBUF <= 160 * 1024) : NSL <= NXB, "weight slices fit");
  auto wslice = [&](int s) -> char* { return (W_BEHIND && s == NSL - 1) ? sH : sX + s * XBUF; };
  auto stamp = [&]([[maybe_unused]] int k) {
    if constexpr (!DIAG) return;
    if (a.stamps && (tid == 0 || tid == 256)) {
      unsigned long long tm;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm)::"memory");
      a.stamps[(size_t)(tid ? 4096 : 0) + (size_t)blockIdx.x * 16 + k] = tm;
    }
  };
  stamp(0);

  if (wave < 4) {
    // =================================================== MATRIX WAVES ===================================================
    const int c = wave / NH, h = wave % NH;
    const int ln = lane & 15, kq = lane >> 4;
    // LDS image of a 16-row x KW-column fp32 tile: row r at r * ROWB, its 16-byte chunk q at position q ^ (r & 15).
    // k-step G of K part h reads chunk 64 h + 4 G + kq of row ln:
    //   position = 64 h + 16 (G >> 2) + ((4 (G & 3) + kq) ^ ln)   ->   xo[G & 3] + 256 (G >> 2) bytes into the tile
    int xo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xo[q] = ln * ROWB + 1024 * h + 16 * ((4 * q + kq) ^ ln);
    const unsigned scnt_lds = (unsigned)(size_t)(__attribute__((address_space(3))) int*)scnt;
    const unsigned sx_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sX;
    // hand-off: lane (ln, kq) holds, per column block cb, columns 32 c + 16 cb + 4 kq + {0..3} of X row ln
    char* const hmine = sH + h * HPL + ln * HROW + (32 * c + 4 * kq) * 4;

    for (int seg = 0; seg < a.nseg; ++seg) {
      const int nps = a.seg_np[seg], G = 256 / nps;
      const int rg = w / nps;
      const int tb = (int)((int64_t)rg * a.ntiles / G), te = (int)((int64_t)(rg + 1) * a.ntiles / G);
      const int nt = te - tb;
      p32_barrier();                             // S0: the previous segment's hand-off has been read
      p32_barrier();                             // P1: the weight slices have landed
      f32x4 wf[NKS][2];                          // [k-step][block of 16 columns]: 4 consecutive reduce elements each
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const char* wb = wslice(2 * c + cb);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) wf[ks][cb] = *reinterpret_cast<const f32x4*>(wb + xo[ks & 3] + 256 * (ks >> 2));
      }
      p32_barrier();                             // P2: every slice is in registers: the buffers are free
      if (NEED_CNT && tid == 0) *scnt = 0;       // (published by B_0; the weight slice that covered this word is in everybody's registers)
      if (seg == 0) stamp(1);

      f32x4 acc0[2], acc1[2];                    // [column block], two tiles in flight
      f32x4 ring[RS];
#pragma unroll
      for (int e = 0; e < 2; ++e) { acc0[e] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[e] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      auto signal_reads_issued = [&]() {
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(scnt_lds), "v"(1) : "memory");
      };
      // ONE fragment ring runs through the whole row range: the read of k-step s is issued XS k-steps (16 MFMAs, 512 cycles)
      // ahead of the MFMAs that consume it (768 cycles for XS = 3), across tile seams too.  Iteration i (behind barrier B_i) issues the fragment reads
      // of tile i; its first XS k-steps of MFMAs are the LAST XS k-steps of tile i - 1 (TAIL), the other NKS - XS the first
      // k-steps of tile i (MAIN).  aP: accumulators of tile i, aQ: those of tile i - 1.
      auto step = [&](auto has_tail, auto has_main, f32x4(&aP)[2], f32x4(&aQ)[2], int i) {
        constexpr bool TAIL = decltype(has_tail)::value, MAIN = decltype(has_main)::value;
        // B_i.  The XS fragment reads that close tile i - 1 (and the counter add behind them) stay in flight across it;
        // everything older -- the hand-off stores of tile i - 2 -- has completed (a wave's LDS operations complete in order).
        if (DIAG && seg == 0 && i == 5) stamp(11);           // (arrival at B_5: stamp 7 - stamp 11 = the wait for the slowest wave)
        if constexpr (TAIL && MAIN) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NEED_CNT ? XS + 1 : XS) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (DIAG && seg == 0 && i < 9) stamp(2 + i);
        unsigned xa[4];
        const unsigned xb = sx_lds + ((i + NXB - 1) % NXB) * XBUF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xa[q] = xb + xo[q];
          asm volatile("" : "+v"(xa[q]));
        }
        auto dump = [&]() {                      // hand tile i - 1 over, clear its accumulators for tile i + 1
          char* hp = hmine + ((i - 1) & 1) * HBUF;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            *reinterpret_cast<f32x4*>(hp + cb * 64) = aQ[cb];
            aQ[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        };
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              if (s < XS) {
                if constexpr (TAIL) aQ[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[NKS - XS + s][cb][j], ring[(NKS - XS + s) % RS][j], aQ[cb], 0, 0, 0);
              } else {
                if constexpr (MAIN) aP[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s - XS][cb][j], ring[(s - XS) % RS][j], aP[cb], 0, 0, 0);
              }
            }
            if constexpr (MAIN) {
              if (j == 0) ring[s % RS] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(xa[s & 3] + 256 * (s >> 2));
            }
          }
          if constexpr (TAIL) { if (s == (MAIN ? XS + 1 : XS - 1)) dump(); }
          // the fragment this k-step multiplied stays "live" to its end: hipcc otherwise hands its registers -- free from the MFMA
          // that read them last -- to this k-step's fragment read, and a write into an operand of an MFMA still in flight waits
          // for it (see RS above); held to here, the read can only land in the slot the previous k-step released
          {
            const int cur = (s < XS ? NKS - XS + s : s - XS) % RS;      // (the loop is fully unrolled: a constant)
            if ((s < XS && TAIL) || (s >= XS && MAIN)) asm volatile("" ::"v"(ring[cur]));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // every fragment read of tile i has been handed to the LDS: the add below is executed behind them (in order), so a
        // service wave that reads 4 (i + 1) here knows the X slot of tile i may be overwritten
        if constexpr (MAIN && NEED_CNT) signal_reads_issued();
      };
      using T = std::true_type;
      using F = std::false_type;
      // iterations 0 .. nt + 1: i = 0 main only; 1 <= i < nt tail + main; i = nt tail only; i = nt + 1 the barrier alone.
      step(F{}, T{}, acc0, acc1, 0);
      int i = 1;
      for (; i + 1 < nt; i += 2) {
        step(T{}, T{}, acc1, acc0, i);
        step(T{}, T{}, acc0, acc1, i + 1);
      }
      if (i < nt) {
        step(T{}, T{}, acc1, acc0, i);
        ++i;
      }
      if (nt & 1) step(T{}, F{}, acc1, acc0, nt);
      else step(T{}, F{}, acc0, acc1, nt);
      p32_barrier();                             // B_(nt + 1)
    }
    stamp(12);
    return;
  }

  // ===================================================== SERVICE WAVES =====================================================
  const int sw = wave - 4;
  const unsigned scnt_lds_s = (unsigned)(size_t)(__attribute__((address_space(3))) int*)scnt;
  if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
  // epilogue: this lane owns columns [4 cq, 4 cq + 4) of row 4 sw + rsub of the tile (BN = 32: lanes 32-63 shadow lanes 0-31)
  const int cq = lane & (QPR - 1), rsub = (lane / QPR) % RPP;
  const bool act = lane < 4 * QPR;
  // raw (stride 0) buffer descriptors over the whole operands; rows are clamped, so nothing is out of range
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)(((int64_t)(a.I - 1) * a.ldx + KW) * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W), 0, (int)(((int64_t)(a.J - 1) * a.ldy + KW) * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(((int64_t)(a.I - 1) * a.ldc + a.J) * 4), 0x00020000);
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(AUX ? a.aux : a.C, 0, (int)(((int64_t)(a.I - 1) * (AUX ? a.ldaux : a.ldc) + a.J) * 4), 0x00020000);
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_RES ? a.res : a.X), 0, (int)(((int64_t)(a.I - 1) * (HAS_RES ? a.ldr : a.ldx) + (HAS_RES ? a.J : KW)) * 4), 0x00020000);
  for (int seg = 0; seg < a.nseg; ++seg) {
    const int nps = a.seg_np[seg], G = 256 / nps;
    const int panel = a.seg_p0[seg] + w % nps, rg = w / nps;
    const int tb = (int)((int64_t)rg * a.ntiles / G), te = (int)((int64_t)(rg + 1) * a.ntiles / G);
    const int nt = te - tb;
    const int j0 = panel * BN;
    // bias (and the LayerNorm-fold column sums) of this lane's four columns, in registers for the segment: requested BEHIND the
    // weight DMA (in front of it, a cold first load cost the whole workgroup its ~1 us before anything else was in flight) by
    // loads hipcc does not see -- a C++ load pending beside the DMA would make it guard every use with a vmcnt(0) that drains
    // the X ring -- and retired by the counted wait that retires the weight pieces, which names them as operands.
    f32x4 bq = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
    auto dma_x = [&](int t) { p32_dma_rows<PPR, NXI, 4>(rsX, (int)a.ldx, (tb + t) * BM, a.I - 1, sX + ((t + NXB - 1) % NXB) * XBUF, sw, lane); };
    // residual tile t -> sR[t % 3]: this wave's KiB = its own four rows, every lane fetches the four columns it will finish
    auto dma_res = [&](int t) {
      if constexpr (HAS_RES) {
        [[maybe_unused]] const int voff = (min((tb + t) * BM + 4 * sw + rsub, a.I - 1) * (int)a.ldr + j0 + 4 * cq) * 4;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsR, (__attribute__((address_space(3))) void*)(sR + (t % NRB) * RBUF + sw * 1024), 16, voff, 0, 0, 0);
#endif
      }
    };
    // LayerNorm statistics of tile t's rows -> sL[t % 3]: [16 rows][8 panels][2] floats = 1 KiB, one instruction (wave 4)
    auto dma_stats = [&](int t) {
      if constexpr (XT == 2) {
        if (sw == 0) {
          const int i = min((tb + t) * BM + (lane >> 2), a.I - 1);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.stats_in + (int64_t)i * 16 + 4 * (lane & 3)),
                                           (__attribute__((address_space(3))) void*)(sL + (t % NRB) * LBUF), 16, 0, 0);
        }
      }
    };

    p32_barrier();                               // S0
    // ---- the weight panel: all slices in flight at once; K = 512: X tile 0 behind them in ring slot 3 ----------------------
#pragma unroll
    for (int s = 0; s < NSL; ++s) p32_dma_rows<PPR, NXI, 4>(rsW, (int)a.ldy, j0 + 16 * s, a.J - 1, wslice(s), sw, lane);   // (by all eight waves: slower, 9.3 K against 7.5 K cycles)
    if constexpr (p32_has_bias(EPI)) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq) : "v"(a.bias + j0 + 4 * cq) : "memory");
    if constexpr (XT == 2) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cs) : "v"(a.colsum + j0 + 4 * cq) : "memory");
    int pre = 0;                                 // X tiles requested before the weights were read
    if constexpr (W_BEHIND) { dma_x(0); pre = 1; asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bq), "+v"(cs) : "n"(NXI) : "memory"); }   // (this wave's weight pieces, bias: landed)
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq), "+v"(cs) :: "memory");
    p32_barrier();                               // P1
    p32_barrier();                               // P2: the matrix waves hold their fragments
    for (int t = pre; t < LOOK && t < nt; ++t) dma_x(t);
    // X(0) has landed before B_0
    {
      const int later = (max(min(LOOK, nt), 1) - 1) * NXI;
      p32_wait_vm(later);
    }

    // ---- epilogue of tile e from the hand-off -----------------------------------------------------------------------------
    constexpr int NST = NPASS * (1 + (XT == 1 ? 1 : 0) + (AUX ? 1 : 0));   // vector-memory stores per wave and tile
    const unsigned sh_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sH;
    [[maybe_unused]] const unsigned sr_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sR;
    [[maybe_unused]] const unsigned sl_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sL;
    auto epilogue = [&](int e) {
#pragma unroll
     for (int pass = 0; pass < NPASS; ++pass) {
      const int row = 4 * sw + RPP * pass + rsub;
      const unsigned hp = sh_lds + (e & 1) * HBUF + row * HROW + 16 * cq;
      f32x4 hh[NH];
#pragma unroll
      for (int p = 0; p < NH; ++p) hh[p] = p32_lds16(hp + p * HPL);
      [[maybe_unused]] f32x4 rr = {0.f, 0.f, 0.f, 0.f};
      if constexpr (HAS_RES) rr = p32_lds16(sr_lds + (e % NRB) * RBUF + sw * 1024 + lane * 16);
      [[maybe_unused]] f32x2_t ps = {0.f, 0.f};
      if constexpr (XT == 2) ps = p32_lds8(sl_lds + (e % NRB) * LBUF + row * 64 + 8 * (cq & 7));   // panel cq & 7 of the row: (sum, centred sum of squares)
      if constexpr (NH == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hh[0]), "+v"(rr), "+v"(ps));
      else if constexpr (NH == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hh[0]), "+v"(hh[1]), "+v"(rr), "+v"(ps));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hh[0]), "+v"(hh[1]), "+v"(hh[2]), "+v"(hh[3]), "+v"(rr), "+v"(ps));
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {              // K parts in a fixed order
        v[k] = hh[0][k];
#pragma unroll
        for (int p = 1; p < NH; ++p) v[k] += hh[p][k];
      }
      if constexpr (XT == 2) {
        // mean / variance of the row from its 8 panel statistics, the merge of gemm.hip spread over the row's 16 lanes (one DPP
        // row): lanes 0-7 hold one panel each, lanes 8-15 contribute nothing
        const bool mine = cq < 8;
        const float tot = row16_sum(mine ? ps.x : 0.f), m2 = row16_sum(mine ? ps.y : 0.f);
        const float mean = tot * (1.0f / 512.0f);
        const float d0 = fmaf(ps.x, 1.0f / 64.0f, -mean);
        const float between = row16_sum(mine ? d0 * d0 : 0.f);
        const float rstd = rsqrtf(fmaf(64.0f, between, m2) * (1.0f / 512.0f) + 1e-5f), mr = -mean * rstd;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaf(v[k], rstd, mr * cs[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += bq[k];
      const int ig = min((tb + e) * BM + row, a.I - 1);
      if constexpr (AUX) { if (act && !(DIAG && (a.dbg & 4))) p32_store16<POL>(rsA, (ig * (int)a.ldaux + j0 + 4 * cq) * 4, f32x4{v[0], v[1], v[2], v[3]}); }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if constexpr (EPI == HIG_EPI_BIAS_RES || EPI == HIG_EPI_RES) v[k] += rr[k];
        if constexpr (EPI == HIG_EPI_BIAS_GELU) v[k] = p32_gelu(v[k]);
        if constexpr (EPI == HIG_EPI_DGELU) v[k] *= p32_dgelu(rr[k]);
      }
      if (DIAG && (a.dbg & 4)) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); continue; }
      if (act) p32_store16<POL>(rsC, (ig * (int)a.ldc + j0 + 4 * cq) * 4, f32x4{v[0], v[1], v[2], v[3]});
      if constexpr (XT == 1) {
        // LayerNorm fold, producer side: (sum, sum of squared deviations from the panel mean) of this row's 64 outputs; the 16
        // lanes of a row are one DPP row.  Every wave issues the store (lanes cq != 0 masked), so the counted waits hold.
        const float sm = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
        const float pm = sm * (1.0f / 64.0f);
        const float a0 = v[0] - pm, a1 = v[1] - pm, a2 = v[2] - pm, a3 = v[3] - pm;
        const float qq = row16_sum((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3));
        if (cq == 0) *reinterpret_cast<float2*>(a.stats_out + ((int64_t)ig * (a.J >> 6) + panel) * 2) = make_float2(sm, qq);
      }
     }
    };

    // ---- main loop: iteration i (behind barrier B_i) requests residual / statistics of tile i and X(i + LOOK), finishes tile
    // i - 2.  Requests of iteration i in issue order: residual(i) [1], statistics(i) [wave 4: 1], X(i + LOOK) [NXI], stores of
    // tile i - 2 [NST]
    auto dbg = [&](int bit) { return DIAG && (a.dbg & bit) != 0; };
    auto n_x = [&](int i) { return (i >= 0 && i + LOOK < nt && !(dbg(1) && i > 2)) ? NXI : 0; };
    auto n_r = [&](int i) { return (HAS_RES && i >= 0 && i < nt) ? 1 : 0; };
    auto n_s = [&](int i) { return (XT == 2 && i >= 0 && i < nt && sw == 0) ? 1 : 0; };
    auto n_st = [&](int i) { return (i >= 2 && i < nt + 2 && !dbg(6)) ? NST : 0; };
    for (int i = 0; i < nt + 2; ++i) {
      p32_barrier();                             // B_i
      if (DIAG && i == 4 && seg == 0) stamp(0);
      if (i < nt) { dma_res(i); dma_stats(i); }
      if (DIAG && i == 4 && seg == 0) stamp(4);
      if (n_x(i)) {
        if constexpr (NEED_CNT) {
          // X(i + LOOK) overwrites the slot of tile i - 1: all four matrix waves have issued its last fragment reads
          int seen;
          do {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(scnt_lds_s) : "memory");
          } while (__builtin_amdgcn_readfirstlane(seen) < 4 * i);
        }
        dma_x(i + LOOK);
      }
      if (DIAG && i == 4 && seg == 0) stamp(1);
      if (i >= 2 && !dbg(2)) epilogue(i - 2);
      if (DIAG && i == 4 && seg == 0) stamp(2);
      // before B_(i+1): X(i + 1) and residual / statistics (i - 1) have landed; what may stay in flight is everything issued
      // behind the younger of the two.  LOOK == 1: X(i + 1) is this iteration's own request, only its stores follow it.
      // LOOK == 2: X(i + 1) went out in iteration i - 1 behind residual (i - 1).  LOOK >= 3: residual (i - 1) is the younger one.
      if constexpr (LOOK == 1) p32_wait_vm(n_st(i));
      else if constexpr (LOOK == 2) p32_wait_vm(n_st(i - 1) + n_r(i) + n_s(i) + n_x(i) + n_st(i));
      else p32_wait_vm(n_x(i - 1) + n_st(i - 1) + n_r(i) + n_s(i) + n_x(i) + n_st(i));
      if (DIAG && i == 4 && seg == 0) stamp(3);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int KW, int EPI, int XT, bool AUX>
int launch_p32(const hig_gemm_desc& g, hipStream_t st) {
  constexpr int BN = 32 * (4 / (KW / 256));
  Wsp32Args a;
  a.X = g.X; a.ldx = g.ldx;
  a.W = g.Y; a.ldy = g.ldy;
  a.C = g.C; a.ldc = g.ldc;
  a.res = EPI == HIG_EPI_DGELU ? g.aux : g.res; a.ldr = EPI == HIG_EPI_DGELU ? g.ldaux : g.ldr;
  a.aux = g.aux; a.ldaux = g.ldaux;
  a.bias = g.bias;
  a.I = g.I; a.J = g.J;
  a.ntiles = (g.I + 15) / 16;
  int np = g.J / BN, p0 = 0;
  a.nseg = 0;
  for (int s = 0; s < P32_MAXSEG; ++s) { a.seg_p0[s] = 0; a.seg_np[s] = 1; }
  while (np > 0) {
    int n = 256;
    while (n > np) n >>= 1;
    if (a.nseg == P32_MAXSEG) return 1;
    a.seg_p0[a.nseg] = p0; a.seg_np[a.nseg] = n; ++a.nseg;
    p0 += n; np -= n;
  }
  a.stats_out = g.row_stats_out;
  a.stats_in = g.row_stats_in;
  a.colsum = g.ln_colsum;
  a.stamps = g_p32_stamps;
  a.prio = 1;   // (a tuning knob while the kernel was built: 0 -> +2 % on the seven forward shapes, 2 = 1; profiles/r06_notes.md section 1)
  static const int dbg = getenv("HIG_F32_WSP_DBG") ? atoi(getenv("HIG_F32_WSP_DBG")) : 0;   // timing ablations (never in a product run)
  a.dbg = dbg;
  constexpr int pol = 0;   // (a tuning knob while the kernel was built: sc1 write-through output stores changed nothing here)
  const bool wt = pol == 1 && !(g.res && g.res == g.C);
  const dim3 gr(256), bl(512);
  __atomic_fetch_add(&g_p32_launches, 1, __ATOMIC_RELAXED);
  if constexpr (KW == 512 && XT == 0 && !AUX && (EPI == HIG_EPI_NONE || EPI == HIG_EPI_BIAS_GELU)) {   // (with a residual the diagnostic instance would spill)
    if (a.stamps || dbg) {                       // diagnostic instances (tools/gemm_wsp32_stamps.py)
      hipLaunchKernelGGL((gemm_wsp32_kernel<KW, EPI, XT, AUX, 0, true>), gr, bl, 0, st, a);
      HIG_CHECK_LAUNCH();
      return HIG_OK;
    }
  }
  if (wt) hipLaunchKernelGGL((gemm_wsp32_kernel<KW, EPI, XT, AUX, 1, false>), gr, bl, 0, st, a);
  else hipLaunchKernelGGL((gemm_wsp32_kernel<KW, EPI, XT, AUX, 0, false>), gr, bl, 0, st, a);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

template <int KW>
int dispatch_p32(const hig_gemm_desc& g, hipStream_t st) {
  if constexpr (KW == 256) {   // the text side's key/value projection (transformer.py:146,150): plain / bias epilogue only
    if (g.epi == HIG_EPI_NONE) return launch_p32<KW, HIG_EPI_NONE, 0, false>(g, st);
    if (g.epi == HIG_EPI_BIAS && !g.row_stats_in) return launch_p32<KW, HIG_EPI_BIAS, 0, false>(g, st);
    return 1;
  } else {
  switch (g.epi) {
    case HIG_EPI_NONE: return launch_p32<KW, HIG_EPI_NONE, 0, false>(g, st);
    case HIG_EPI_BIAS:
      if constexpr (KW == 512) { if (g.row_stats_in) return launch_p32<KW, HIG_EPI_BIAS, 2, false>(g, st); }
      return launch_p32<KW, HIG_EPI_BIAS, 0, false>(g, st);
    case HIG_EPI_BIAS_GELU: return g.aux ? launch_p32<KW, HIG_EPI_BIAS_GELU, 0, true>(g, st) : launch_p32<KW, HIG_EPI_BIAS_GELU, 0, false>(g, st);
    case HIG_EPI_BIAS_RES:
      if constexpr (KW == 512) { if (g.row_stats_out) return launch_p32<KW, HIG_EPI_BIAS_RES, 1, false>(g, st); }
      return launch_p32<KW, HIG_EPI_BIAS_RES, 0, false>(g, st);
    case HIG_EPI_RES: return launch_p32<KW, HIG_EPI_RES, 0, false>(g, st);
    case HIG_EPI_DGELU: return launch_p32<KW, HIG_EPI_DGELU, 0, false>(g, st);
    default: return 1;
  }
  }
}

}  // namespace

// Returns HIG_OK when the launch was made, 1 when this kernel does not serve the call (the caller goes on to the tiled kernel
// of gemm.hip), a negative HIG_E* code on error.
bool hig_gemm_wsp32_active() {
  static const int on = getenv("HIG_F32_WSP") ? atoi(getenv("HIG_F32_WSP")) : 1;                 // tuning knob: 0 = this kernel off
  return on && hig_chip_cus() == 256;
}
int hig_gemm_wsp32_try(const hig_gemm_desc& g, hipStream_t st) {
  constexpr int min_rows = 2048;   // below that a workgroup has < 4 tiles per segment to pay its weight phase with
  if (!hig_gemm_wsp32_active()) return 1;
  if (g.prec != HIG_PREC_F32 || g.x_rs || g.y_rs || g.xf != HIG_XF_NONE || g.xcolsum) return 1;
  if (!(g.R == 256 || g.R == 512 || g.R == 1024) || g.I < min_rows) return 1;
  const int bn = 32 * (4 / (g.R / 256));
  if (g.J % bn != 0 || g.J / bn > 256) return 1;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!(g.ldx % 4 == 0 && g.ldy % 4 == 0 && g.ldc % 4 == 0 && al(g.X) && al(g.Y) && al(g.C))) return 1;
  const int64_t lim = 1ll << 29;               // byte offsets are 32-bit
  if ((int64_t)g.I * g.ldx >= lim || (int64_t)g.J * g.ldy >= lim || (int64_t)g.I * g.ldc >= lim) return 1;
  const bool has_bias = p32_has_bias(g.epi), has_res = g.epi == HIG_EPI_BIAS_RES || g.epi == HIG_EPI_RES;
  if (has_bias && !(g.bias && al(g.bias))) return 1;
  if (has_res && !(g.res && g.ldr % 4 == 0 && al(g.res) && (int64_t)g.I * g.ldr < lim)) return 1;
  if ((g.epi == HIG_EPI_DGELU || (g.epi == HIG_EPI_BIAS_GELU && g.aux)) && !(g.aux && g.ldaux % 4 == 0 && al(g.aux) && (int64_t)g.I * g.ldaux < lim)) return 1;
  if (g.row_stats_out && !(g.R == 512 && g.epi == HIG_EPI_BIAS_RES && !g.row_stats_in && (reinterpret_cast<uintptr_t>(g.row_stats_out) & 7) == 0)) return 1;
  if (g.row_stats_in && !(g.R == 512 && g.epi == HIG_EPI_BIAS && g.ln_colsum && al(g.row_stats_in) && al(g.ln_colsum))) return 1;
  if (g.R == 256) return (g.row_stats_out || g.row_stats_in || g.aux) ? 1 : dispatch_p32<256>(g, st);
  return g.R == 512 ? dispatch_p32<512>(g, st) : dispatch_p32<1024>(g, st);
}

// Diagnostic: thread 0 (matrix wave 0) of every workgroup writes s_memtime stamps to buf[block * 16 + k] (k: 0 start, 1 weights
// in registers, 2 + i barrier of iteration i (i < 9), 12 end), thread 256 (service wave 4) to buf[4096 + block * 16 + k] (iteration 4 of the first segment: 0 barrier
// passed, 1 requests issued, 2 epilogue done, 3 counted wait over).  buf: 2 x 4096 x 8 bytes; NULL switches it off.
// How many launches went to this kernel since the library was loaded (monotonic; tests read the difference around a call to
// prove which kernel served it -- the product instances write no stamps).
extern "C" int64_t hig_gemm_wsp32_launches(void) { return __atomic_load_n(&g_p32_launches, __ATOMIC_RELAXED); }

extern "C" int hig_gemm_wsp32_debug_stamps(void* buf) {
  g_p32_stamps = static_cast<unsigned long long*>(buf);
  return HIG_OK;
}
