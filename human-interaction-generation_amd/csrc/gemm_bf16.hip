// bf16-STORAGE GEMM for gfx950 (MI355X):  C[i][j] = epi( sum_r X[i][r] * Y[j][r] )
//   X (I, ldx) activations and Y (J, ldy) nn.Linear weight (out, in), both bf16, both reduce-contiguous;
//   fp32 accumulate on v_mfma_f32_32x32x16_bf16; C (and the residual) bf16 or fp32.
// Serves every nn.Linear of the denoiser forward when hig_dims.storage == HIG_STORE_BF16 (reference:
// codes/models/transformer.py:81-85,108-114,144-150,168,345-349,425 -- the reference runs them in fp32; BASELINE
// configs 3 and 5 ask for bf16 storage with fp32 accumulation).
//
// CDNA4 mapping.  256 threads = 4 waves; a wave owns 64 rows x (32 TJ) columns as 32x32 accumulators.  Operand tiles
// go global -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR hop, no conversion): the
// LDS image of a tile is [rows][BK] bf16 (BK = 64: 128-byte rows) written LINEARLY by the DMA, so the bank-conflict
// swizzle sits on the per-lane SOURCE address and on the read address (guide rule 21): 16-byte chunk c of row r is
// stored at chunk position c ^ ((r / RB) & (CHUNKS-1)), RB = rows per 256-byte bank row -- the 16-lane ds_read_b128
// groups {0-3,12-15,20-27}... then touch 16 distinct 16-byte slots (conflict-free).  One MFMA operand = one
// ds_read_b128 (8 consecutive k of one row).  The MFMA row operand is Y (output column), so a lane holds 4 consecutive
// output columns per accumulator quad; the finished tile is staged through LDS (fp32) and leaves as whole rows:
// 16 bytes of bf16 (8 columns) or 32 bytes of fp32 per lane, bias / GELU / SiLU / residual applied in fp32 on the way.
// Two LDS stages: the DMA of k-tile kt+1 is issued right after the barrier that publishes k-tile kt and lands
// during its MFMA block; with 2 workgroups per CU (64 KiB of LDS each) the other workgroup's MFMAs cover the rest.
#include <stdlib.h>

#include "gemm16_epi.h"
#include "hig_host.h"

namespace {

struct K16Args {
  hig_gemm16_desc g;
  int nbj, ntiles;
  int nxy;          // tiles of ONE split (ntiles = nxy * splits)
  int chunk;        // reduce elements per split (g.R when splits == 1): split s covers [s chunk, (s + 1) chunk)
  int64_t slab;     // elements between the outputs of consecutive splits (fp32 slabs, summed by hig_reduce_slabs)
  int vec;          // C / res rows allow 16-byte (bf16) / 32-byte (fp32) row pieces
  unsigned long long* stamps;   // HIG_BF16_DBG & 16: per-workgroup s_memtime stamps of the first tile's phases (diagnostic build of the run)
  int dbg;          // timing ablations (HIG_BF16_DBG; results are wrong): 2 = no epilogue, 8 = epilogue without global
                    // stores / residual loads, 16 = s_memtime stamps (the in-loop ablations 1 = no DMA after the first
                    // k-tile and 4 = no MFMA were removed after use: profiles/r02_notes.md)
};

__device__ __forceinline__ void glds16(const __bf16* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

unsigned long long* g_stamps = nullptr;   // diagnostic only (hig_gemm_bf16_debug_stamps)

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// WM x WN waves, each 64 x (32 TJ):  BM = 64 WM, BN = 32 TJ WN.  NS LDS stages of one BK-deep k-tile each: the DMA runs
// NS-1 k-tiles ahead of the MFMAs (ring of buffers, ONE raw s_barrier per k-tile, counted vmcnt so that the younger
// k-tiles stay in flight across the barrier -- a __syncthreads() here would drain them, guide "Pipelining across
// barriers").
template <int WM, int WN, int TI, int TJ, int BK, int NS, int EPI, bool SRD>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_bf16_kernel(const K16Args a) {
  constexpr int NT = 64 * WM * WN, NW = WM * WN;
  constexpr int BM = 32 * TI * WM, BN = 32 * TJ * WN;
  static_assert(TI % 2 == 0, "a wave row is a whole number of 64-row epilogue passes");
  constexpr int ROWB = BK * 2, CHUNKS = BK / 8, RB = 256 / ROWB, RPI = 1024 / ROWB;   // RPI: rows per DMA instruction
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = BM / RPI, NB = BN / RPI, NQ = (NA + NB) / NW;   // DMA instructions: per tile, per wave
  static_assert((NA + NB) % NW == 0, "DMA instructions must split evenly over the waves");
  constexpr int CLD = BN + 4;
  constexpr int EPI_BYTES = 64 * CLD * 4;
  constexpr int SMEM = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const hig_gemm16_desc& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / WN, wj = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const __bf16* __restrict__ X = static_cast<const __bf16*>(g.X);
  const __bf16* __restrict__ Y = static_cast<const __bf16*>(g.Y);
  const int nk = a.chunk / BK;
  // buffer descriptors (SGPRs): whole operand extents, raw (stride 0) buffers; rows are clamped, so nothing is out of range
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(X), 0, (int)(((int64_t)(g.I - 1) * g.ldx + g.R) * 2), 0x00020000);
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Y), 0, (int)(((int64_t)(g.J - 1) * g.ldy + g.R) * 2), 0x00020000);

  // read-side byte offsets inside a stage (k-step ks adds the XOR-ed chunk)
  int xoff[TI], xsw[TI], yoff[TJ], ysw[TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    const int r = wi * (32 * TI) + 32 * ti + lr;
    xoff[ti] = r * ROWB;
    xsw[ti] = (r / RB) & (CHUNKS - 1);
  }
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    const int r = wj * (32 * TJ) + 32 * tj + lr;
    yoff[tj] = A_BYTES + r * ROWB;
    ysw[tj] = (r / RB) & (CHUNKS - 1);
  }

  auto stamp = [&](int k) {
    if ((a.dbg & 16) && tid == 0 && blockIdx.x < 4096) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      a.stamps[(size_t)blockIdx.x * 8 + k] = t;
    }
  };
  bool first_tile = true;
  for (int lin = blockIdx.x; lin < a.ntiles; lin += gridDim.x) {
    if (first_tile) stamp(0);
    // XCD-aware order: blocks b, b + 8, ... share an XCD (private L2): give each XCD a contiguous run of tiles, column
    // tiles of one row panel next to each other, so the X panel is fetched from HBM once per XCD
    const int q8 = a.ntiles >> 3, r8 = a.ntiles & 7, xcd = lin & 7;
    const int unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    // split-R launches (weight gradients of the bf16-storage training step: reduce extent = B T rows): unit = (split, tile),
    // split-major, so the tiles an XCD walks next to each other share their operand panels
    const int split = unit / a.nxy, tile = unit - split * a.nxy;
    const int i0 = (tile / a.nbj) * BM, j0 = (tile % a.nbj) * BN;
    const int64_t k0 = (int64_t)split * a.chunk;      // first reduce element of this unit

    // per-lane DMA sources: instruction q of this wave covers RPI rows of the A (X) or B (Y) tile.
    // SRD form: `buffer_load_dwordx4 ... lds` through a buffer descriptor in SGPRs -- the per-lane part is a 32-bit byte
    // offset that is constant over the k-loop (clamped row x ld + swizzled chunk), the k-tile offset is ONE scalar: no
    // 64-bit per-lane address to send and none to advance per DMA.
    const __bf16* src[NQ];
    int voff[NQ];
    int dst[NQ];
    [[maybe_unused]] int koff = (int)(k0 * 2);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int n = wave + NW * q;                     // instruction index in [A tile | B tile] order
      const bool isA = n < NA;
      const int r = (isA ? n : n - NA) * RPI + lane / CHUNKS;
      const int c = (lane % CHUNKS) ^ ((r / RB) & (CHUNKS - 1));
      if (SRD) {
        voff[q] = isA ? (min(i0 + r, g.I - 1) * (int)g.ldx + 8 * c) * 2 : (min(j0 + r, g.J - 1) * (int)g.ldy + 8 * c) * 2;
      } else {
        src[q] = isA ? X + (int64_t)min(i0 + r, g.I - 1) * g.ldx + 8 * c + k0
                     : Y + (int64_t)min(j0 + r, g.J - 1) * g.ldy + 8 * c + k0;
      }
      dst[q] = (isA ? 0 : A_BYTES) + (isA ? n : n - NA) * 1024;
    }
    auto issue = [&](int buf, int q) {
      if constexpr (SRD) {
        static_assert(NA % NW == 0, "a wave's q-th DMA must be of one operand");
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass cannot type-check the builtin and would silently drop the kernel's stub)
        if (q < NA / NW)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (__attribute__((address_space(3))) void*)(smem + buf * STAGE + dst[q]), 16,
                                                   voff[q], koff, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (__attribute__((address_space(3))) void*)(smem + buf * STAGE + dst[q]), 16,
                                                   voff[q], koff, 0, 0);
#endif
      } else {
        glds16(src[q], smem + buf * STAGE + dst[q]);
        src[q] += BK;
      }
    };
    auto stage = [&](int buf) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) issue(buf, q);
      koff += BK * 2;
    };
    // the same, spread over the KS k-steps of the MFMA block: a wave needs ~150-200 cycles to issue ONE 1-KiB DMA
    // (s_memtime stamps, profiles/r02_notes.md), so issuing a k-tile's 6-8 back to back held the wave (and, with both
    // waves of a SIMD leaving the barrier together, the matrix pipe) for longer than the whole MFMA block; slice
    // `ks` of the DMAs now goes out behind the MFMAs of k-step ks, which execute while the next slice is issued.
    constexpr int KS = BK / 16;
    auto stage_slice = [&](int buf, int ks) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q * KS / NQ == ks) issue(buf, q);
      if (ks == KS - 1) koff += BK * 2;
    };

    // The epilogue's per-thread bias columns are fetched NOW: every VGPR-returning load of a tile is then retired by
    // the compiler-visible wait of the barrier that ends the main loop.  (A VGPR load still "pending" in hipcc's
    // scoreboard when the next tile starts makes it guard the first reuse of that register with s_waitcnt vmcnt(0) --
    // inside the main loop, where it drains the DMA ring every k-tile.)
    // Epilogue work split: a 64-row pass is 64 x Q8 eight-column pieces, piece idx = tid + NT * sweep.  With Q8 = 16
    // (BN = 128) a thread keeps one column piece for all sweeps; with Q8 = 24 (BN = 192) it cycles through 3.
    constexpr int Q8 = BN / 8, NSW = 64 * Q8 / NT, NBS = NT % Q8 == 0 ? 1 : 3;
    static_assert((64 * Q8) % NT == 0 && (NT % Q8 == 0 || (3 * NT) % Q8 == 0), "epilogue split");
    float b8[NBS][8];
#pragma unroll
    for (int u = 0; u < NBS; ++u) {
      const int ju = j0 + 8 * ((tid + NT * u) % Q8);
#pragma unroll
      for (int e = 0; e < 8; ++e) b8[u][e] = epi_has_bias(EPI) ? g.bias[min(ju + e, g.J - 1)] : 0.f;
    }

    f32x16 acc[TJ][TI];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tj][ti][e] = 0.f;

    if (first_tile) stamp(1);
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
      if (p < nk) stage(p);
    if (first_tile) stamp(2);
    int cur = 0, nxt = NS - 1;                            // ring positions of k-tile kt and of k-tile kt + NS - 1
    for (int kt = 0; kt < nk; ++kt) {
      // this wave's share of k-tile kt has landed once at most the NS-2 younger k-tiles' DMAs are outstanding
      if (kt + NS - 2 < nk) wait_vmcnt<(NS - 2) * NQ>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                       // everyone's share has; and k-tile kt-1's buffer is free
      asm volatile("" ::: "memory");
      const bool refill = kt + NS - 1 < nk;
      const int fill = nxt;
      const char* sb = smem + cur * STAGE;
      cur = cur + 1 == NS ? 0 : cur + 1;
      nxt = nxt + 1 == NS ? 0 : nxt + 1;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 xf[TI], yf[TJ];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
          xf[ti] = *reinterpret_cast<const bf16x8*>(sb + xoff[ti] + 16 * ((2 * ks + lh) ^ xsw[ti]));
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
          yf[tj] = *reinterpret_cast<const bf16x8*>(sb + yoff[tj] + 16 * ((2 * ks + lh) ^ ysw[tj]));
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
            acc[tj][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[tj], xf[ti], acc[tj][ti], 0, 0, 0);
        if (refill) stage_slice(fill, ks);
      }
    }
    __syncthreads();   // all MFMA reads of the staging buffers are done (no DMA in flight): reuse them for the output tile
    if (first_tile) stamp(5);

    // ---- epilogue: one 64-row pass per wave row, through LDS (fp32), out as whole rows -------------------
    if (a.dbg & 2) {
      if (acc[0][0][0] + acc[TJ - 1][1][3] == 123.456f) static_cast<float*>(g.C)[0] = b8[0][0];
      continue;
    }
    float* sC = reinterpret_cast<float*>(smem);
    constexpr bool HAS_RES = epi_has_res(EPI);
    // LDS-only barrier: the output stores of one pass stay in flight across it (a __syncthreads() would wait vmcnt(0)
    // for them twice per tile: 16 of 43 us at the FFN shape, profiles/r02_notes.md)
    auto lds_barrier = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int ps = 0; ps < BM / 64; ++ps) {
      // the bf16 residual rows of this pass are requested first: their latency hides under the LDS staging + barrier
      bf16x8 r16[NSW];
      if (HAS_RES && a.vec && !g.res_f32) {
#pragma unroll
        for (int sw = 0; sw < NSW; ++sw) {
          const int idx = tid + NT * sw;
          const int i = min(i0 + 64 * ps + idx / Q8, g.I - 1), j = min(j0 + 8 * (idx % Q8), g.J - 8);
          r16[sw] = *reinterpret_cast<const bf16x8*>(static_cast<const __bf16*>(g.res) + (int64_t)i * g.ldr + j);
        }
      }
      if (wi == ps / (TI / 2)) {         // the wave row that owns rows [64 ps, 64 ps + 64): two of its TI row blocks
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x16& c = acc[tj][(ps % (TI / 2)) * 2 + t2];
              *reinterpret_cast<f32x4*>(sC + (32 * t2 + lr) * CLD + wj * (32 * TJ) + 32 * tj + 8 * q + 4 * lh) =
                  f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
            }
      }
      lds_barrier();
      if (first_tile) stamp(ps == 0 ? 3 : 7);
#pragma unroll
      for (int sw = 0; sw < NSW; ++sw) {
        const int idx = tid + NT * sw;
        const int rr = idx / Q8, c8 = idx % Q8;
        const int i = i0 + 64 * ps + rr, j = j0 + 8 * c8;
        if (i >= g.I || j >= g.J) continue;
        const bool full = a.vec && j + 8 <= g.J;
        const float (&bb)[8] = b8[sw % NBS];
        float v[8];
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sC + rr * CLD + 8 * c8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sC + rr * CLD + 8 * c8 + 4);
        v[0] = v0.x + bb[0]; v[1] = v0.y + bb[1]; v[2] = v0.z + bb[2]; v[3] = v0.w + bb[3];
        v[4] = v1.x + bb[4]; v[5] = v1.y + bb[5]; v[6] = v1.z + bb[6]; v[7] = v1.w + bb[7];
        if (a.dbg & 8) {
          if (v[0] + v[7] == 123.456f) static_cast<float*>(g.C)[0] = v[3];
          continue;
        }
        if (HAS_RES) {
          float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (g.res_f32) {
            const float* rp = static_cast<const float*>(g.res) + (int64_t)i * g.ldr + j;
            if (full) {
              const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
              rv[0] = r0.x; rv[1] = r0.y; rv[2] = r0.z; rv[3] = r0.w; rv[4] = r1.x; rv[5] = r1.y; rv[6] = r1.z; rv[7] = r1.w;
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (j + e < g.J) rv[e] = rp[e];
            }
          } else if (full) {
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = (float)r16[sw][e];
          } else {
            const __bf16* rp = static_cast<const __bf16*>(g.res) + (int64_t)i * g.ldr + j;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (j + e < g.J) rv[e] = (float)rp[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (EPI == HIG_EPI_DGELU) v[e] *= dgelu_bf16(rv[e]);   // `res` = z of FFN linear1 (data gradient through the GELU)
            else v[e] += rv[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = epi_act<EPI>(v[e]);
        if (g.c_f32) {
          float* cp = static_cast<float*>(g.C) + (int64_t)split * a.slab + (int64_t)i * g.ldc + j;
          if (full) {
            *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (j + e < g.J) cp[e] = v[e];
          }
        } else {
          __bf16* cp = static_cast<__bf16*>(g.C) + (int64_t)i * g.ldc + j;
          if (full) {
            *reinterpret_cast<bf16x8*>(cp) = bf16x8{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3],
                                                    (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (j + e < g.J) cp[e] = (__bf16)v[e];
          }
        }
      }
      lds_barrier();   // sC is rewritten by the next pass / the next tile's DMA
      if (first_tile && ps == 0) stamp(4);
    }
    if (first_tile) stamp(6);
    first_tile = false;
  }
}

template <int WM, int WN, int TI, int TJ, int BK, int NS, int EPI>
int launch16(const hig_gemm16_desc& g, hipStream_t st, int splits = 1, int64_t slab = 0) {
  constexpr int use_srd = 1;   // (a former tuning knob, fixed at the value that won its A/B): 0 = global_load_lds
  constexpr int NT = 64 * WM * WN, BM = 32 * TI * WM, BN = 32 * TJ * WN;
  K16Args a;
  a.g = g;
  const int nbi = (g.I + BM - 1) / BM;
  a.nbj = (g.J + BN - 1) / BN;
  a.nxy = nbi * a.nbj;
  a.ntiles = a.nxy * splits;
  a.chunk = g.R / splits;
  a.slab = slab;
  auto al = [](const void* p, int n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
  a.vec = (g.c_f32 ? (g.ldc % 4 == 0 && al(g.C, 16)) : (g.ldc % 8 == 0 && al(g.C, 16)));
  if (g.res) a.vec = a.vec && (g.res_f32 ? (g.ldr % 4 == 0 && al(g.res, 16)) : (g.ldr % 8 == 0 && al(g.res, 16)));
  constexpr int lds = NS * (BM + BN) * BK * 2;
  int per_cu = 160 * 1024 / (lds > 64 * (BN + 4) * 4 ? lds : 64 * (BN + 4) * 4);
  if (per_cu > 4) per_cu = 4;
  constexpr int forced_per_cu = 0;   // (a former tuning knob, fixed at the value that won its A/B)
  if (forced_per_cu > 0) per_cu = forced_per_cu;
  static const int dbg = getenv("HIG_BF16_DBG") ? atoi(getenv("HIG_BF16_DBG")) : 0;
  a.dbg = g_stamps ? dbg : (dbg & ~16);
  a.stamps = g_stamps;
  int grid = hig_chip_cus() * per_cu;
  if (grid > a.ntiles) grid = a.ntiles;
  // (operands beyond 2 GiB would overflow the 32-bit byte offsets of the descriptor form)
  const bool srd_ok = ((int64_t)g.I * g.ldx < (1ll << 30)) && ((int64_t)g.J * g.ldy < (1ll << 30));
  if (use_srd && srd_ok)
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, TI, TJ, BK, NS, EPI, true>), dim3(grid), dim3(NT), 0, st, a);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, TI, TJ, BK, NS, EPI, false>), dim3(grid), dim3(NT), 0, st, a);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// Tile choice.  Every launch of this model is a short-K problem (K = 256 ... 2048): per-tile fixed costs (first DMA,
// epilogue) and the number of ROUNDS the tiles take on the chip decide, not the MFMA rate.  Candidates: 128 x 128 and
// 128 x 192 (2 resident workgroups per CU = 512 slots), 64 x 128 (3 per CU = 768 slots).  Rule: fewest rounds; among
// equals the larger tile when it still gives every CU a workgroup, else the smaller one (more CUs busy).
template <int EPI>
int launch16_sized(const hig_gemm16_desc& g, hipStream_t st) {
  static const int forced_env = getenv("HIG_BF16_TILE") ? atoi(getenv("HIG_BF16_TILE")) : 0;   // tuning knob: 64 / 128
  const int forced = (forced_env == 64 || forced_env == 128) ? forced_env : 0;
  auto tiles = [&](int bm, int bn) { return (int64_t)((g.I + bm - 1) / bm) * ((g.J + bn - 1) / bn); };
  auto rounds = [](int64_t t, int slots) { return (t + slots - 1) / slots; };
  const int64_t t128 = tiles(128, 128), t64 = tiles(64, 128);
  int pick = 64;
  if (forced) {
    pick = forced;
  } else {
    // estimated time = rounds x relative cost of one tile (fitted to tools/gemm16_bench.py at M = 6272 and 12544,
    // profiles/r02_notes.md: a 64 x 128 tile costs 0.75 of a 128 x 128 one, not the 0.5 of its area -- it re-reads the
    // same weight panel for half the rows)
    const int ncu = hig_chip_cus();   // 2 resident 128-row workgroups per CU, 3 of the 64-row ones
    const double c128 = (double)rounds(t128, 2 * ncu) * 1.0, c64 = (double)rounds(t64, 3 * ncu) * 0.75;
    pick = 128;
    double best = c128;
    if (t128 < ncu && c64 <= best * 1.25) { pick = 64; best = c64; }        // too few big tiles to occupy the chip
    else if (c64 < best) { pick = 64; best = c64; }
    // (a 256 x 256 tile -- 8 waves, one workgroup per CU -- won the wide K = 1024 launches of the d = 1024 model alone, 77 against 89 us,
    // and lost inside the forward, 4.61-4.64 against 4.57 ms: not built any more)
  }
  // (ring shape, re-measured with per-phase stamps at the FFN linear1 shape, profiles/r02_notes.md section 7: the main
  // loops of two co-resident workgroups move 2 x 256 KB in ~19.3K cycles = 27 B/clk per CU, which IS the CU's L2 -> LDS
  // rate (guide: 66-73 GB/s per CU for L2-resident rows, DMA and register staging alike).  4 x BK 32 stages: 18.5K;
  // 3 x BK 64 stages (one workgroup per CU): 10.9K for its single tile but no second workgroup to cover the epilogue,
  // 44.9 against 36.5 us; start-time offsets between the co-resident workgroups: no effect.)
  // 64-row tiles are what the small-M launches get (M = 6272: one partly filled round of 2-3 workgroups per CU): there a
  // k-tile is bound by the LATENCY of its DMA, not by the CU's fetch rate, and a third ring stage (72 KB of LDS, still
  // two workgroups per CU) hides it: stylization-out 14.9 -> 12.8 us, FFN linear2 21.4 -> 18.2 us, config-3 forward
  // 1.515 -> 1.409 ms (B = 64: 2.30 -> 2.28 ms).  Not for K = 256 (4 k-tiles: 8.8 -> 15.4 us) and not for the 128-row
  // tile (96 KB = one workgroup per CU: FFN linear1 20.6 -> 24.2 us).  HIG_BF16_RING3=0 switches it off.
  constexpr int ring3 = 1;   // (a former tuning knob, fixed at the value that won its A/B)
  if (ring3 && pick == 64 && g.R % 64 == 0 && g.R >= 512) return launch16<1, 4, 2, 1, 64, 3, EPI>(g, st);
  // 128 x 128 tiles over several rounds (M >= 8192: every launch of the B = 64 forward): four stages of BK = 32 (the
  // same 64 KB) keep one more k-tile in flight than two of BK = 64: B = 64 forward 2.308 -> 2.261 ms (q/k/v 44.2 ->
  // 41.8 us); at M = 6272, where these launches are single partly filled rounds, it is neutral to slightly slower
  // (1.425 -> 1.435 ms), so the rule is on the row count.  HIG_BF16_RING4_ROWS moves the threshold (0 = never).
  constexpr int ring4_rows = 8192;   // (a former tuning knob, fixed at the value that won its A/B)
  if (ring4_rows > 0 && pick == 128 && g.I >= ring4_rows && g.R >= 512) return launch16<2, 2, 2, 2, 32, 4, EPI>(g, st);
  if (g.R % 64 == 0) {
    if (pick == 128) return launch16<2, 2, 2, 2, 64, 2, EPI>(g, st);
    return launch16<1, 4, 2, 1, 64, 2, EPI>(g, st);
  }
  if (pick != 64) return launch16<2, 2, 2, 2, 32, 3, EPI>(g, st);
  return launch16<1, 4, 2, 1, 32, 3, EPI>(g, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Few rows (I <= 64: the per-sample Linear layers of the time / text embedding chain, transformer.py:345-349,81-83 at the
// sampling batch sizes): the launch is a WEIGHT read -- J x K x 2 bytes against I x J outputs -- and what bounds it is how
// many CUs pull on the weight at once (a CU fetches ~25-30 B/clk).  The tiled kernel above gives such a launch J / 128
// workgroups (time_embed.2 at B = 32: 16 workgroups streaming 512 KB each, 20 us).  Here a workgroup owns 32 output
// columns and all rows; its 4 waves split the reduce range into quarters (contiguous 32-byte pieces of every weight row
// per load, successive loads walk the same cache lines), operands go global -> registers in MFMA fragment shape (no LDS
// staging: nothing is reused), the four partial tiles are summed through LDS and the epilogue runs on whole 32-column row
// pieces.  J / 32 workgroups: 64 for the 2048-wide embedding layers.
// ---------------------------------------------------------------------------------------------------------------------
template <int NRB, int EPI>
__global__ __launch_bounds__(256) void gemm_fewrow16_kernel(const hig_gemm16_desc g) {
  constexpr int PLD = 33;                                  // fp32 elements per row of a partial tile in LDS
  __shared__ float sP[4 * NRB * 32 * PLD];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j0 = blockIdx.x * 32;
  const __bf16* __restrict__ X = static_cast<const __bf16*>(g.X);
  const __bf16* __restrict__ Y = static_cast<const __bf16*>(g.Y);
  const int nks = g.R / 16, per = (nks + 3) / 4;           // k-steps: all, per wave
  const int ks0 = wave * per, ks1 = min(nks, ks0 + per);
  const __bf16* yp = Y + (int64_t)(j0 + lr) * g.ldy + 8 * lh;
  const __bf16* xp[NRB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) xp[rb] = X + (int64_t)min(32 * rb + lr, g.I - 1) * g.ldx + 8 * lh;
  f32x16 acc[NRB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[rb][e] = 0.f;
#pragma unroll 8
  for (int ks = ks0; ks < ks1; ++ks) {
    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(yp + 16 * ks);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp[rb] + 16 * ks);
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[rb], 0, 0, 0);
    }
  }
  // acc[rb][4q + e]: output column j0 + 8q + 4lh + e of row 32 rb + lr
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) sP[((wave * NRB + rb) * 32 + lr) * PLD + 8 * q + 4 * lh + e] = acc[rb][4 * q + e];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NRB * 4; ++u) {
    const int idx = tid + 256 * u, r = idx >> 5, c = idx & 31;      // r: row of the (32 NRB)-row tile; 32 lanes = one row piece
    if (r >= g.I) continue;
    const int rb = r >> 5, rr = r & 31;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += sP[((w * NRB + rb) * 32 + rr) * PLD + c];
    const int j = j0 + c;
    if (EPI != HIG_EPI_NONE) v += g.bias[j];
    if (EPI == HIG_EPI_BIAS_RES || EPI == HIG_EPI_BIAS_RES_SILU)
      v += g.res_f32 ? static_cast<const float*>(g.res)[(int64_t)r * g.ldr + j] : (float)static_cast<const __bf16*>(g.res)[(int64_t)r * g.ldr + j];
    v = epi_act<EPI>(v);
    if (g.c_f32) static_cast<float*>(g.C)[(int64_t)r * g.ldc + j] = v;
    else static_cast<__bf16*>(g.C)[(int64_t)r * g.ldc + j] = (__bf16)v;
  }
}

// The same decomposition with the operands staged through LDS (reduce range a multiple of 256): fragment-shaped global
// loads (32 rows x 32 bytes per instruction) run at a fraction of the CU's fetch rate -- time_embed.2 at 32 rows took 15 us
// with them for 256 KB per workgroup.  Here every wave DMA-s ITS quarter of the reduce range in 64-element chunks (whole
// 128-byte row pieces, 8 rows per instruction, XOR-swizzled on the source address) into a wave-private ring of three (two at 64 rows) chunks:
// no workgroup barrier until the final reduction, counted vmcnt waits only.
template <int NRB, int EPI>
__global__ __launch_bounds__(256) void gemm_fewrow16_lds_kernel(const hig_gemm16_desc g) {
  constexpr int PLD = 33;
  constexpr int NCH = NRB == 1 ? 3 : 2;                    // ring depth in chunks of 64 reduce elements (LDS: 96 KB either way)
  constexpr int CHB = 32 * 128;                            // bytes of one 32-row x 64-element operand block
  constexpr int WCH = (1 + NRB) * CHB;                     // bytes per chunk and wave: W block, then NRB X blocks
  constexpr int DPC = (1 + NRB) * 4;                       // DMA instructions per chunk and wave
  __shared__ __attribute__((aligned(1024))) char sOp[4 * NCH * WCH];
  __shared__ float sP[4 * NRB * 32 * PLD];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j0 = blockIdx.x * 32;
  const __bf16* __restrict__ X = static_cast<const __bf16*>(g.X);
  const __bf16* __restrict__ Y = static_cast<const __bf16*>(g.Y);
  const int KW = g.R / 4, nch = KW / 64, k0 = wave * KW;   // this wave's share of the reduce range
  char* const my = sOp + wave * NCH * WCH;
  // per-lane DMA sources: instruction q of an operand block covers rows 8 q .. 8 q + 7, lane = (row % 8, 16-byte position)
  const int drow = lane >> 3, dpos = lane & 7;
  auto dma_chunk = [&](int c, int buf) {
    const int kc = k0 + 64 * c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 8 * q + drow;
      const __bf16* src = Y + (int64_t)(j0 + r) * g.ldy + kc + 8 * (dpos ^ (r & 7));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(my + buf * WCH + q * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 8 * q + drow;
        const __bf16* src = X + (int64_t)min(32 * rb + r, g.I - 1) * g.ldx + kc + 8 * (dpos ^ (r & 7));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(my + buf * WCH + (1 + rb) * CHB + q * 1024), 16, 0, 0);
      }
  };
  f32x16 acc[NRB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[rb][e] = 0.f;
  for (int c = 0; c < NCH - 1 && c < nch; ++c) dma_chunk(c, c);
  for (int c = 0; c < nch; ++c) {
    if (c + NCH - 1 < nch) dma_chunk(c + NCH - 1, (c + NCH - 1) % NCH);   // (its buffer held chunk c - 1: this wave is done with it)
    // chunk c has landed once only the younger chunks' requests of THIS wave are outstanding
    const int younger = min(NCH - 1, nch - 1 - c);
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPC) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* cb = my + (c % NCH) * WCH;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = lr * 128 + 16 * ((2 * ks + lh) ^ (lr & 7));
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(cb + off);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(cb + (1 + rb) * CHB + off);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[rb], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the fragments are in registers before the buffer is requested again
  }
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) sP[((wave * NRB + rb) * 32 + lr) * PLD + 8 * q + 4 * lh + e] = acc[rb][4 * q + e];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NRB * 4; ++u) {
    const int idx = tid + 256 * u, r = idx >> 5, c = idx & 31;
    if (r >= g.I) continue;
    const int rb = r >> 5, rr = r & 31;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += sP[((w * NRB + rb) * 32 + rr) * PLD + c];
    const int j = j0 + c;
    if (EPI != HIG_EPI_NONE) v += g.bias[j];
    if (EPI == HIG_EPI_BIAS_RES || EPI == HIG_EPI_BIAS_RES_SILU)
      v += g.res_f32 ? static_cast<const float*>(g.res)[(int64_t)r * g.ldr + j] : (float)static_cast<const __bf16*>(g.res)[(int64_t)r * g.ldr + j];
    v = epi_act<EPI>(v);
    if (g.c_f32) static_cast<float*>(g.C)[(int64_t)r * g.ldc + j] = v;
    else static_cast<__bf16*>(g.C)[(int64_t)r * g.ldc + j] = (__bf16)v;
  }
}

template <int EPI>
int launch_fewrow16(const hig_gemm16_desc& g, hipStream_t st) {
  constexpr int lds_on = 1;   // (a former tuning knob, fixed at the value that won its A/B)
  if (lds_on && g.R % 256 == 0) {               // operands through wave-private LDS rings
    if (g.I <= 32) hipLaunchKernelGGL((gemm_fewrow16_lds_kernel<1, EPI>), dim3(g.J / 32), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_fewrow16_lds_kernel<2, EPI>), dim3(g.J / 32), dim3(256), 0, st, g);
  } else if (g.I <= 32) hipLaunchKernelGGL((gemm_fewrow16_kernel<1, EPI>), dim3(g.J / 32), dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gemm_fewrow16_kernel<2, EPI>), dim3(g.J / 32), dim3(256), 0, st, g);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// Serves: I <= 64, J a multiple of 32 with at most 512 column blocks (beyond that the tiled kernel already has every CU
// busy), reduce range a multiple of 64 of at least 256.  Returns 1 when the shape is not served.  HIG_BF16_FEWROW=0: off.
int fewrow16_try(const hig_gemm16_desc& g, hipStream_t st) {
  static const int on = getenv("HIG_BF16_FEWROW") ? atoi(getenv("HIG_BF16_FEWROW")) : 1;   // tuning knob
  if (!on || g.I > 64 || g.J % 32 != 0 || g.J / 32 > 512 || g.R % 64 != 0 || g.R < 256) return 1;
  switch (g.epi) {
    case HIG_EPI_NONE: return launch_fewrow16<HIG_EPI_NONE>(g, st);
    case HIG_EPI_BIAS: return launch_fewrow16<HIG_EPI_BIAS>(g, st);
    case HIG_EPI_BIAS_GELU: return launch_fewrow16<HIG_EPI_BIAS_GELU>(g, st);
    case HIG_EPI_BIAS_RES: return launch_fewrow16<HIG_EPI_BIAS_RES>(g, st);
    case HIG_EPI_BIAS_SILU: return launch_fewrow16<HIG_EPI_BIAS_SILU>(g, st);
    case HIG_EPI_BIAS_RES_SILU: return launch_fewrow16<HIG_EPI_BIAS_RES_SILU>(g, st);
    default: return 1;
  }
}


__global__ void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t n) {
  const int64_t n8 = n / 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i], b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
    reinterpret_cast<bf16x8*>(dst)[i] = bf16x8{(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w,
                                               (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[n8 * 8 + threadIdx.x] = (__bf16)src[n8 * 8 + threadIdx.x];
}


// ---------------------------------------------------------------------------------------------------------------------
// joint_embed of the bf16-storage forward:  h0[m][n] = sum_f x[m][f] Wj[n][f] + bj[n] + pos[(m % T) - shift][n]
// (transformer.py:418-419).  x is the fp32 DDPM state with F = 150 / 263 / ... features per row (rows are not 16-byte
// aligned, F is not a multiple of the MFMA k), the output is the bf16 residual stream.  The general kernel above cannot
// take such operands and the exact-fp32 GEMM it went through before (+ a cast) cost 31 us of a 1.5 ms step for 1 GFLOP.
// Here: (1) the weight is padded to Fp = 32-multiple columns and rounded to bf16 (by the caller next to its weight shadow,
// hig_joint_embed_bf16_w, or per call by pad_cast_rows_kernel, 150 KB); (2) a workgroup takes 64 rows of x, converts them to bf16 in LDS (zero beyond F), and walks the d columns
// of ONE block of 128 (grid.y): weight block -> LDS, v_mfma_f32_32x32x16_bf16 with the weight as the row operand (a lane then
// holds 4 consecutive output columns); the fp32 tile goes through LDS so that bias + positional row are added and the bf16
// result is stored a whole 256-byte row segment at a time.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void pad_cast_rows_kernel(const float* __restrict__ W, int rows, int F, int Fp, __bf16* __restrict__ out) {
  const int64_t n = (int64_t)rows * Fp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / Fp), c = (int)(i % Fp);
    out[i] = c < F ? (__bf16)W[(int64_t)r * F + c] : (__bf16)0.f;
  }
}

__global__ __launch_bounds__(256) void joint_embed16_kernel(const float* __restrict__ x, int F, int Fp,
                                                            const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                                                            const float* __restrict__ pos, int64_t ldpos, int T,
                                                            int pos_shift, __bf16* __restrict__ out, int64_t ldo,
                                                            int64_t M, int d) {
  extern __shared__ __attribute__((aligned(16))) char smem_je[];
  const int LDX = Fp + 8;                                  // bf16 elements per LDS row (16-byte aligned, skewed banks)
  __bf16* sX = reinterpret_cast<__bf16*>(smem_je);         // [64][LDX]
  __bf16* sW = sX + 64 * LDX;                              // [128][LDX]
  constexpr int LDC = 132;                                 // fp32 elements per row of the output tile (over sX / sW after the products)
  float* sC = reinterpret_cast<float*>(smem_je);           // [64][LDC]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int cb = blockIdx.y * 128;
  // one 128-column block per workgroup (blockIdx.y): the x tile is fetched d / 128 times (from L2), in exchange every
  // block of every row tile runs at once instead of as a serial chain of load -> MFMA -> store steps per workgroup.
  // Weight block first (independent 16-byte loads, the loop unrolled so they are all in flight together):
  const int c16 = Fp / 8;                                  // 16-byte chunks per padded weight row
  // (12 at a time: the 10 loads per thread of F = 150 are then ONE round trip to L2 / HBM instead of three.  Row / chunk of
  // a piece are stepped, not divided out per piece: Fp and F are run-time values, and an emulated integer division is ~40
  // vector instructions -- the three per piece this kernel used to do were 1 200 of its 1 330 vector instructions per wave,
  // rocprofv3 SQ_INSTS_VALU, and made it issue-bound: 140 us at M = 100 352 for 163 MB.)
  {
    int rr = tid / c16, ch = tid % c16;
    const int drr = 256 / c16, dch = 256 % c16;
#pragma unroll 12
    for (int idx = tid; idx < 128 * c16; idx += 256) {
      *reinterpret_cast<uint4*>(sW + rr * LDX + 8 * ch) = *reinterpret_cast<const uint4*>(Wp + (int64_t)(cb + rr) * Fp + 8 * ch);
      rr += drr;
      ch += dch;
      if (ch >= c16) { ch -= c16; ++rr; }
    }
  }
  // x tile: the 64 rows are ONE contiguous run of 64 F floats (the pose rows are dense), read as 16-byte vectors whatever
  // F is, rounded to bf16 and scattered to (row, column) in LDS; columns F .. Fp of every row are zeroed (rows beyond M
  // only produce outputs that are never stored)
  {
    const int padc = Fp - F;
    for (int c = F + (tid & 3); c < Fp; c += 4) sX[(tid >> 2) * LDX + c] = (__bf16)0.f;     // (row tid / 4, every fourth pad column)
    (void)padc;
    const float* xt = x + m0 * F;
    const int64_t rows_here = M - m0 < 64 ? M - m0 : 64;
    const int n_el = (int)(rows_here * F);
    const bool vec = (reinterpret_cast<uintptr_t>(xt) & 15) == 0;
    int r0 = (4 * tid) / F, c0 = (4 * tid) % F;             // (row, column) of this thread's first element; stepped by 1024 elements
    const int dr = 1024 / F, dc = 1024 % F;
    if (vec && (F & 1) == 0) {
      // usual case (aligned x, even F): whole 16-byte loads, no per-element guards -- with F even every column index this
      // thread meets at the start of a pair is even, so a pair never straddles a row end and goes out as one 4-byte LDS write
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      const int n4 = n_el >> 2;                                // (n_el = rows x F is a multiple of 2; a trailing pair is done below)
#pragma unroll 12
      for (int i = tid; i < n4; i += 256) {
        const float4 q = *reinterpret_cast<const float4*>(xt + 4 * i);
        *reinterpret_cast<bf16x2_t*>(sX + r0 * LDX + c0) = bf16x2_t{(__bf16)q.x, (__bf16)q.y};
        int r1 = r0, c1 = c0 + 2;
        if (c1 >= F) { c1 = 0; ++r1; }
        *reinterpret_cast<bf16x2_t*>(sX + r1 * LDX + c1) = bf16x2_t{(__bf16)q.z, (__bf16)q.w};
        r0 += dr;
        c0 += dc;
        if (c0 >= F) { c0 -= F; ++r0; }
      }
      if ((n_el & 3) && tid == 0) {                            // the last pair of an n_el = 4 k + 2 tile
        const int e = 4 * n4, r = e / F, c = e % F;
        *reinterpret_cast<bf16x2_t*>(sX + r * LDX + c) = bf16x2_t{(__bf16)xt[e], (__bf16)xt[e + 1]};
      }
    } else {
      for (int e0 = 4 * tid; e0 < n_el; e0 += 1024) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec && e0 + 3 < n_el) {
          const float4 q = *reinterpret_cast<const float4*>(xt + e0);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (e0 + k < n_el) v[k] = xt[e0 + k];
        }
        int r = r0, c = c0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (e0 + k < n_el) sX[r * LDX + c] = (__bf16)v[k];
          if (++c == F) { c = 0; ++r; }
        }
        r0 += dr;
        c0 += dc;
        if (c0 >= F) { c0 -= F; ++r0; }
      }
    }
  }
  // bias and positional rows of the four output pieces this thread will store (piece u: row (tid + 256 u) / 16, columns
  // cb + 8 (tid % 16) ..): requested HERE, ahead of the products, so that the row pass at the end waits for nothing
  const int oc8 = 8 * (tid & 15);
  f32x4 pb[2], pp[4][2];
  pb[0] = *reinterpret_cast<const f32x4*>(bias + cb + oc8);
  pb[1] = *reinterpret_cast<const f32x4*>(bias + cb + oc8 + 4);
  int tmod = (int)(((unsigned)m0 + (unsigned)(tid >> 4)) % (unsigned)T);   // frame index of piece 0's row; pieces are 16 rows apart
  const int tstep = 16 % T;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int tp = tmod - pos_shift;
    tmod += tstep;
    if (tmod >= T) tmod -= T;
    const float* pr = pos + (int64_t)max(tp, 0) * ldpos + cb + oc8;
    pp[u][0] = *reinterpret_cast<const f32x4*>(pr);
    pp[u][1] = *reinterpret_cast<const f32x4*>(pr + 4);
    if (tp < 0) pp[u][0] = pp[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};    // (init-pose row of the two-person model: no positional term)
  }
  __syncthreads();
  f32x16 acc[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[rb][e] = 0.f;
  const __bf16* wrow = sW + (32 * wave + lr) * LDX + 8 * lh;
  for (int ks = 0; ks < Fp / 16; ++ks) {
    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + 16 * ks);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sX + (32 * rb + lr) * LDX + 16 * ks + 8 * lh);
      acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[rb], 0, 0, 0);
    }
  }
  __syncthreads();                                         // the operands are consumed: their LDS becomes the fp32 output tile
  // acc[rb][4q + e]: output column cb + 32 wave + 8q + 4lh + e of row m0 + 32 rb + lr
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<f32x4*>(sC + (32 * rb + lr) * LDC + 32 * wave + 8 * q + 4 * lh) =
          f32x4{acc[rb][4 * q], acc[rb][4 * q + 1], acc[rb][4 * q + 2], acc[rb][4 * q + 3]};
  __syncthreads();
  // rows out: 16 lanes per row, 8 columns each: bias + positional row added in fp32 (coalesced 32-byte reads), one 16-byte
  // bf16 store per lane = 256 contiguous bytes per row
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = (tid + 256 * u) >> 4;
    const int64_t m = m0 + r;
    if (m >= M) continue;
    const f32x4 v0 = (*reinterpret_cast<const f32x4*>(sC + r * LDC + oc8) + pb[0]) + pp[u][0];
    const f32x4 v1 = (*reinterpret_cast<const f32x4*>(sC + r * LDC + oc8 + 4) + pb[1]) + pp[u][1];
    *reinterpret_cast<bf16x8*>(out + m * ldo + cb + oc8) = bf16x8{(__bf16)v0.x, (__bf16)v0.y, (__bf16)v0.z, (__bf16)v0.w,
                                                                  (__bf16)v1.x, (__bf16)v1.y, (__bf16)v1.z, (__bf16)v1.w};
  }
}

}  // namespace

int hig_gemm16_launch(const hig_gemm16_desc& g, hipStream_t st) {
  HIG_REQUIRE(g.X && g.Y && g.C, "hig_gemm_bf16: null operand");
  HIG_REQUIRE(g.I >= 0 && g.J >= 0 && g.R > 0, "hig_gemm_bf16: bad extent");
  if (g.I == 0 || g.J == 0) return HIG_OK;
  if (g.R % 32 != 0)
    return hig_set_error(HIG_EUNSUPPORTED, "hig_gemm_bf16: the reduce extent must be a multiple of 32 (got %d)", g.R);
  HIG_REQUIRE(g.ldx % 8 == 0 && g.ldy % 8 == 0 && (reinterpret_cast<uintptr_t>(g.X) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(g.Y) & 15) == 0,
              "hig_gemm_bf16: operands must be 16-byte aligned with leading dimensions that are multiples of 8");
  if (epi_has_bias(g.epi)) HIG_REQUIRE(g.bias, "hig_gemm_bf16: epilogue %d needs a bias", g.epi);
  if (epi_has_res(g.epi)) HIG_REQUIRE(g.res, "hig_gemm_bf16: epilogue %d needs `res`", g.epi);
  {   // many rows, K = 512: the weight-stationary kernel with specialised waves (gemm_wsp16.hip)
    const int rc = hig_gemm_wsp16_try(g, st);
    if (rc <= 0) return rc;
    if (g.aux) return hig_set_error(HIG_EUNSUPPORTED, "hig_gemm_bf16: `aux` (pre-activation output) on a shape gemm_wsp16 does not serve");
  }
  {   // many rows, short reduce range: the weight-stationary kernel (gemm_ws16.hip) when it serves the shape
    const int rc = hig_gemm_ws16_try(g, st);
    if (rc <= 0) return rc;
  }
  {   // a handful of rows: one workgroup per 32 output columns (gemm_fewrow16_kernel)
    const int rc = fewrow16_try(g, st);
    if (rc <= 0) return rc;
  }
  switch (g.epi) {
    case HIG_EPI_NONE: return launch16_sized<HIG_EPI_NONE>(g, st);
    case HIG_EPI_BIAS: return launch16_sized<HIG_EPI_BIAS>(g, st);
    case HIG_EPI_BIAS_GELU: return launch16_sized<HIG_EPI_BIAS_GELU>(g, st);
    case HIG_EPI_BIAS_RES: return launch16_sized<HIG_EPI_BIAS_RES>(g, st);
    case HIG_EPI_BIAS_SILU: return launch16_sized<HIG_EPI_BIAS_SILU>(g, st);
    case HIG_EPI_BIAS_RES_SILU: return launch16_sized<HIG_EPI_BIAS_RES_SILU>(g, st);
    case HIG_EPI_RES: return launch16_sized<HIG_EPI_RES>(g, st);
    case HIG_EPI_DGELU: return launch16_sized<HIG_EPI_DGELU>(g, st);
    default: return hig_set_error(HIG_EUNSUPPORTED, "hig_gemm_bf16: epilogue %d not built", g.epi);
  }
}

// Split-R form (weight gradients of the bf16-storage training step, dW = dC^T . act over the M = B T rows: few output tiles,
// a long reduce range): the reduce range is cut into `splits` chunks, unit (split, tile) writes its partial tile to
// slabs[split] (fp32, dense J-wide rows), hig_reduce_slabs sums them in split order into C (deterministic, no float
// atomics).  EPI_NONE, fp32 C with ldc == J, R % (64 splits) == 0.  splits == 0: the library's rule (units fill the two
// resident workgroups per CU without exceeding them).
int hig_gemm16_split_splits(const hig_gemm16_desc& g, int64_t slab_floats) {
  const int64_t tiles = (int64_t)((g.I + 127) / 128) * ((g.J + 127) / 128);
  const int nk = g.R / 64;
  int best = 1;
  for (int s = 1; s <= nk && s <= 64; ++s) {
    if (nk % s) continue;
    if (tiles * s > 2 * (int64_t)hig_chip_cus() && s > 1) break;
    if ((int64_t)s * g.I * g.J > slab_floats) break;
    if (nk / s < 4 && s > 1) break;               // at least four k-tiles per unit: the DMA ring needs a few to overlap
    best = s;
  }
  return best;
}
int hig_gemm16_split_launch(const hig_gemm16_desc& g0, int splits, float* slabs, int64_t slab_floats, hipStream_t st) {
  HIG_REQUIRE(g0.X && g0.Y && g0.C && slabs, "hig_gemm_bf16_split: null operand");
  HIG_REQUIRE(g0.epi == HIG_EPI_NONE && g0.c_f32 && g0.ldc == g0.J, "hig_gemm_bf16_split: EPI_NONE, dense fp32 C only");
  HIG_REQUIRE(g0.R > 0 && g0.R % 64 == 0 && g0.ldx % 8 == 0 && g0.ldy % 8 == 0 &&
                  (reinterpret_cast<uintptr_t>(g0.X) & 15) == 0 && (reinterpret_cast<uintptr_t>(g0.Y) & 15) == 0,
              "hig_gemm_bf16_split: 16-byte aligned operands, leading dimensions multiples of 8, R a multiple of 64");
  HIG_REQUIRE(((int64_t)g0.I * g0.J) % 4 == 0 && (reinterpret_cast<uintptr_t>(g0.C) & 15) == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0,
              "hig_gemm_bf16_split: I J must be a multiple of 4, C / slabs 16-byte aligned");
  if (splits <= 0) splits = hig_gemm16_split_splits(g0, slab_floats);
  HIG_REQUIRE((g0.R / 64) % splits == 0, "hig_gemm_bf16_split: splits=%d does not divide the %d k-tiles", splits, g0.R / 64);
  const int64_t slab = (int64_t)g0.I * g0.J;
  HIG_REQUIRE(splits == 1 || slab * splits <= slab_floats, "hig_gemm_bf16_split: slab scratch too small");
  hig_gemm16_desc g = g0;
  if (splits > 1) g.C = slabs;
  // 128 x 128 tiles, two k-tiles of 64 in flight (long reduce range: the ring covers the DMA latency)
  HIG_TRY((launch16<2, 2, 2, 2, 64, 2, HIG_EPI_NONE>(g, st, splits, slab)));
  if (splits > 1) return hig_reduce_slabs(slabs, splits, slab, g0.I * (int64_t)g0.J, static_cast<float*>(g0.C), st);
  return HIG_OK;
}
extern "C" int64_t hig_gemm_bf16_split_scratch_floats(const hig_gemm16_desc* g, int32_t splits) {
  if (!g || g->I <= 0 || g->J <= 0) return -1;
  return (int64_t)(splits > 0 ? splits : 64) * g->I * g->J;
}
extern "C" int hig_gemm_bf16_split(const hig_gemm16_desc* g, int32_t splits, float* slabs, int64_t slab_floats, hig_stream_t stream) {
  HIG_REQUIRE(g, "hig_gemm_bf16_split: null descriptor");
  return hig_gemm16_split_launch(*g, splits, slabs, slab_floats, hig_stream(stream));
}

extern "C" int hig_gemm_bf16(const hig_gemm16_desc* g, hig_stream_t stream) {
  HIG_REQUIRE(g, "hig_gemm_bf16: null descriptor");
  return hig_gemm16_launch(*g, hig_stream(stream));
}

// joint_embed + sequence_embedding of the bf16-storage forward (see joint_embed16_kernel).
// hig_joint_embed_bf16_w: the weight already padded and rounded (d x Fp bf16, Fp = F rounded up to a multiple of 32, zero
// columns beyond F) -- what a caller keeps next to its bf16 weight shadow; hig_joint_embed_bf16: fp32 (d, F) weight, padded
// into `w_scratch` (hig_joint_embed_bf16_scratch_bytes) by every call.  d % 128 == 0, F <= 512, 16-byte aligned bias / pos / out.
extern "C" int64_t hig_joint_embed_bf16_scratch_bytes(int32_t F, int32_t d) {
  if (F <= 0 || d <= 0) return -1;
  return (int64_t)d * ((F + 31) / 32 * 32) * 2;
}
extern "C" int hig_joint_embed_bf16_w(const float* x, int64_t M, int32_t F, const void* w_padded, const float* bias, const float* pos,
                                      int64_t ldpos, int32_t T, int32_t pos_shift, void* out, int64_t ldo, int32_t d,
                                      hig_stream_t stream) {
  HIG_REQUIRE(x && w_padded && bias && pos && out && M >= 0 && M < (1ll << 31) && F > 0 && d > 0 && T > 0, "hig_joint_embed_bf16: bad arguments");
  if (M == 0) return HIG_OK;
  if (d % 128 != 0 || F > 512)
    return hig_set_error(HIG_EUNSUPPORTED, "hig_joint_embed_bf16: needs d %% 128 == 0 and F <= 512 (got %d, %d)", d, F);
  HIG_REQUIRE(ldo % 8 == 0 && ldpos % 4 == 0 &&
                  ((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(w_padded) |
                    reinterpret_cast<uintptr_t>(out)) & 15) == 0,
              "hig_joint_embed_bf16: alignment");
  const int Fp = (F + 31) / 32 * 32;
  size_t lds = (size_t)(64 + 128) * (Fp + 8) * 2;
  if (lds < 64 * 132 * 4) lds = 64 * 132 * 4;             // (the fp32 output tile reuses the operand space)
  static const int big_lds_rc = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_embed16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               156 * 1024) == hipSuccess ? 0 : 1;
  }();
  if (big_lds_rc != 0 || lds > 156 * 1024) return hig_set_error(HIG_EHIP, "hig_joint_embed_bf16: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(joint_embed16_kernel, dim3((unsigned)((M + 63) / 64), d / 128), dim3(256), lds, hig_stream(stream), x, F, Fp,
                     static_cast<const __bf16*>(w_padded), bias, pos, ldpos, T, pos_shift, static_cast<__bf16*>(out), ldo, M, d);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}
extern "C" int hig_joint_embed_bf16(const float* x, int64_t M, int32_t F, const float* W, const float* bias, const float* pos,
                                    int64_t ldpos, int32_t T, int32_t pos_shift, void* out, int64_t ldo, int32_t d,
                                    void* w_scratch, hig_stream_t stream) {
  HIG_REQUIRE(W && w_scratch && F > 0 && d > 0, "hig_joint_embed_bf16: bad arguments");
  if (M == 0) return HIG_OK;
  const int Fp = (F + 31) / 32 * 32;
  const int64_t nw = (int64_t)d * Fp;
  hipLaunchKernelGGL(pad_cast_rows_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, hig_stream(stream), W, d, F, Fp,
                     static_cast<__bf16*>(w_scratch));
  HIG_CHECK_LAUNCH();
  return hig_joint_embed_bf16_w(x, M, F, w_scratch, bias, pos, ldpos, T, pos_shift, out, ldo, d, stream);
}

extern "C" int hig_cast_bf16(const float* src, void* dst, int64_t n, hig_stream_t stream) {
  HIG_REQUIRE(src && dst && n >= 0, "hig_cast_bf16: bad arguments");
  if (n == 0) return HIG_OK;
  HIG_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
              "hig_cast_bf16: buffers must be 16-byte aligned");
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)blocks), dim3(256), 0, hig_stream(stream), src,
                     static_cast<__bf16*>(dst), n);
  HIG_CHECK_LAUNCH();
  return HIG_OK;
}

// Diagnostic: with HIG_BF16_DBG & 16, thread 0 of every workgroup (< 4096) writes s_memtime stamps of its first tile's
// phases to buf[block * 8 + k] (k: 0 tile start, 1 setup done, 2 first DMA issued, 3 first k-tile landed, 4 second,
// 5 main loop done, 6 epilogue done).  NULL switches it off.  Never part of a timed run.
extern "C" int hig_gemm_bf16_debug_stamps(void* buf) {
  g_stamps = static_cast<unsigned long long*>(buf);
  return HIG_OK;
}
