// Host-side helpers shared by the orchestration translation units (denoiser.hip, texthead.hip).
#pragma once
#include <stdlib.h>
#include <string.h>

#include "hig_common.h"

int hig_gemm_launch(const hig_gemm_desc& g, int splits, float* slabs, hipStream_t st);
// I <= 64 rows with EPI_BIAS / EPI_BIAS_RES: split-R over `scratch` so the weight streams through ~1024 workgroups
int hig_gemm_few_rows(const hig_gemm_desc& g, float* scratch, int64_t scratch_floats, hipStream_t st);

// Scratch for the split tail of the exact-fp32 GEMM (gemm.hip, KArgs): the launches that follow on this thread may
// park partial sums in it.  HIG_GEMM_TAIL_BYTES long, 16-byte aligned; its first HIG_GEMM_TAIL_CNT_BYTES (tickets)
// must be zero before the first launch and are left zero by every launch.  One scratch serves one stream at a time.
// nullptr switches the tail off.
constexpr int64_t HIG_GEMM_TAIL_CNT_BYTES = 1024;
constexpr int64_t HIG_GEMM_TAIL_BYTES = HIG_GEMM_TAIL_CNT_BYTES + 256 * 16384;   // 256 slices of a 64x64 tile
void hig_gemm_set_tail_scratch(void* ws, int64_t bytes);

// Chip geometry of the CURRENT device, read once per device with hipDeviceGetAttribute (never a literal in a launch
// rule): compute units, and XCDs = CUs / 32 (a gfx950 XCD has 32 active CUs; an MI355X in SPX mode reports 256 -> 8, a
// CPX partition 32 -> 1).  Round-counting tile rules, persistent grid sizes and the residency limits of the fused
// kernels are derived from these; kernels whose block -> XCD mapping is compiled for 8 XCDs decline on anything else.
int hig_chip_cus();
inline int hig_chip_xcds() { const int c = hig_chip_cus() / 32; return c < 1 ? 1 : c; }

int hig_gemm16_launch(const hig_gemm16_desc& g, hipStream_t st);
// split-R form of the tiled bf16 kernel + deterministic slab reduction (weight gradients); splits == 0: library rule
int hig_gemm16_split_launch(const hig_gemm16_desc& g, int splits, float* slabs, int64_t slab_floats, hipStream_t st);
// dW = dC^T . act (+ dbias = column sums of dC) straight from the row-major bf16 operands (wgrad16.hip: transpose reads)
// (deferred != NULL: the slab reduction is left to hig_wgrad16_reduce_batch, which sums up to HIG_WG_RB_MAX gradients in one launch)
struct hig_wg_reduce {
  const float* slabs; int nsplit; int64_t slab;   // nsplit slabs of `slab` floats: [J x K | J] each
  int64_t n4; float* out;                          // J K / 4 float4 of dW
  int64_t nb4; float* dbias;                       // J / 4 float4 of dbias (0: none)
};
constexpr int HIG_WG_RB_MAX = 12;
int hig_wgrad16_launch(const void* dC, int64_t ldd, const void* act, int64_t ldx, int64_t rows, int J, int K, float* dW, float* dbias,
                       int splits, float* slabs, int64_t slab_floats, hipStream_t st, hig_wg_reduce* deferred = nullptr);
int hig_wgrad16_reduce_batch(const hig_wg_reduce* entries, int n, hipStream_t st);
// up to HIG_WG_GROUP_MAX gradients in one launch of the kernel (wgrad16.hip)
struct hig_wg_problem {
  const void* dC; int64_t ldd; const void* act; int64_t ldx; int64_t rows; int J, K; float* dW; float* dbias; int splits;   // splits 0: the rule
};
constexpr int HIG_WG_GROUP_MAX = 4;
int hig_wgrad16_launch_group(const hig_wg_problem* probs, int n, float* slabs, int64_t slab_floats, hipStream_t st, hig_wg_reduce* deferred,
                             int64_t* used_floats);
int64_t hig_wgrad16_rule_floats(int64_t rows, int J, int K, int64_t room);   // slab floats the split rule takes, given `room`
// out[e] = sum_s slabs[s * slab + e], e < n (n % 4 == 0, 16-byte aligned), in split order (gemm.hip)
int hig_reduce_slabs(const float* slabs, int splits, int64_t slab, int64_t n, float* out, hipStream_t st);
// weight-stationary variant (gemm_ws16.hip): HIG_OK = launched, 1 = shape not served (use the tiled kernel), < 0 = error
int hig_gemm_ws16_try(const hig_gemm16_desc& g, hipStream_t st);
// weight-stationary kernel with specialised matrix / service waves (gemm_wsp16.hip: K = 512, J % 128 == 0, >= 2048 rows); same codes
int hig_gemm_wsp16_try(const hig_gemm16_desc& g, hipStream_t st);
bool hig_gemm_ws16_lnfold_ok(int64_t rows, int d);
// linattn.hip: context build of G groups of H heads in one launch (the batched text side); 1 = shape not served
int hig_linattn_ctx_groups(const float* K, const float* V, int64_t ld, int32_t B, int32_t rows, int32_t H, int32_t G, int32_t hd,
                           float* A, int64_t a_gs, float* kstat, int64_t k_gs, hipStream_t st);
int hig_linattn_ctx16_groups(const void* K, const void* V, int64_t ld, int32_t B, int32_t rows, int32_t H, int32_t G, int32_t hd,
                             float* A, int64_t a_gs, float* kstat, int64_t k_gs, void* At16, int64_t at_gs, hipStream_t st);   // (linattn16.hip, bf16 rows)
// exact-fp32 weight-stationary kernel with specialised waves (gemm_wsp32.hip: K = 256 / 512 / 1024, reduce-contiguous aligned
// operands, >= 2048 rows); same return codes
int hig_gemm_wsp32_try(const hig_gemm_desc& g, hipStream_t st);
// weight gradients dW = dC^T . act over >= 2048 rows, I and J multiples of 128, tiles x splits <= 256 (wgrad_wsp32.hip): writes
// the split-R slabs (+ per-split column sums of dC) of hig_gemm_launch(splits > 1); same return codes
int hig_wgrad_wsp32_try(const hig_gemm_desc& g, int splits, float* slabs, int64_t slab, float* xsum, int64_t xsum_stride, hipStream_t st);
bool hig_gemm_wsp32_active();   // that kernel is switched on and the chip has the 256 CUs its work split is written for

namespace {

inline int64_t al(int64_t floats) { return (floats + 63) & ~(int64_t)63; }  // 256-byte granules

struct G {  // small builder for gemm descriptors
  hig_gemm_desc g;
  G(const float* X, int64_t ldx, int xrs, const float* Y, int64_t ldy, int yrs, float* C, int64_t ldc,
    int64_t I, int64_t J, int64_t R) {
    memset(&g, 0, sizeof(g));
    g.X = X; g.ldx = ldx; g.x_rs = xrs; g.Y = Y; g.ldy = ldy; g.y_rs = yrs; g.C = C; g.ldc = ldc;
    g.I = (int)I; g.J = (int)J; g.R = (int)R;
    g.xf = HIG_XF_NONE; g.epi = HIG_EPI_NONE; g.prec = HIG_PREC_F32;
  }
  G& prec(int p) { g.prec = p; return *this; }  // forward products: HIG_PREC_* of the plan
  G& epi(int e, const float* bias = nullptr) { g.epi = e; g.bias = bias; return *this; }
  G& res(const float* r, int64_t ldr) { g.res = r; g.ldr = ldr; return *this; }
  G& aux(float* a, int64_t lda) { g.aux = a; g.ldaux = lda; return *this; }
  G& pos(const float* p, int64_t ldp, int T) { g.pos = p; g.ldpos = ldp; g.T = T; return *this; }
  G& silu(int on_y) { g.xf = HIG_XF_SILU; g.xf_on_y = on_y; return *this; }
  G& ln(int on_y, const float* stats, const float* gamma, const float* beta) {
    g.xf = HIG_XF_LN; g.xf_on_y = on_y; g.stats = stats; g.gamma = gamma; g.beta = beta; return *this;
  }
  G& xsum(float* out) { g.xcolsum = out; return *this; }   // wgrad: also the bias gradient (column sums of dC)
  G& mod(const float* ss, int64_t ss_ld, int shift_off, int rows_per_sample) {
    g.xf = HIG_XF_LN_MOD_SILU; g.ss = ss; g.ss_ld = ss_ld; g.ss_shift_off = shift_off;
    g.rows_per_sample = rows_per_sample; return *this;
  }
};

// Weight-gradient GEMMs (dW = dC^T . act over the M rows).  Tile: exact-fp32 products run 64x64 tiles (four resident
// workgroups per CU, 4x fewer split-R slabs to write and sum than with 128x128: forward+backward 20.6 -> 20.2 ms),
// the bf16 product modes 128x128 (14.9 vs 15.2 ms bf16x3, 12.9 vs 13.5 ms bf16) -- same-box sweeps in
// profiles/r01_notes.md.  HIG_WGRAD_TILE = 64 / 128 forces one.
inline int wgrad_tile(int64_t I, int64_t J, int prec) {
  constexpr int forced = 0;   // (a former tuning knob, fixed at the value that won its A/B)
  if (!(I > 64 && J > 64)) return 64;
  if (forced == 64 || forced == 128) return forced;
  return prec == HIG_PREC_F32 ? 64 : 128;
}

// Split the reduce range so that tiles x splits fills, but does not exceed, the workgroups that are resident at once
// (128x128: 2 per CU = 512, 64 KB of LDS each; 64x64: 4 per CU = 1024) -- one more would start a second, nearly empty
// round (768 in flight cost the training step 0.6 ms; profiles/r01_notes.md).
inline int wgrad_splits(int64_t I, int64_t J, int64_t R, int64_t slab_floats, int prec) {
  const int bi = wgrad_tile(I, J, prec);
  const int64_t tiles = ((I + bi - 1) / bi) * ((J + bi - 1) / bi);
  constexpr int forced_target = 0;  // (a former tuning knob, fixed at the value that won its A/B)
  const int target = forced_target > 0 ? forced_target : (bi == 128 ? 2 : 4) * hig_chip_cus();
  int64_t s = target / tiles;
  const int64_t maxs = R / 256 > 1 ? R / 256 : 1;
  if (s > maxs) s = maxs;
  while (s > 1 && s * (I * J + I) > slab_floats) --s;   // slabs + the per-split column sums of X behind them
  if ((I * J) % 4 != 0) s = 1;
  return (int)(s < 1 ? 1 : s);
}

}  // namespace

// rowops.hip: hig_ln_bwd_bf16 with the reductions of its partial table left to hig_ln_bwd16_reduce_batch (one launch for up to
// HIG_LN_RB_MAX calls; each call needs its own partial table until then)
struct hig_ln_reduce {
  const float* partial; int samples, nsplit, n;
  float* dgamma; float* dbeta;
  int shift_off; float* dss; int64_t dss_ld;       // dss == NULL: a plain LayerNorm (no modulation gradients)
  int nb_col, nb_dss;
};
constexpr int HIG_LN_RB_MAX = 8;
int hig_ln_bwd16_launch(const void* da, int64_t ldda, const void* x, int32_t x_f32, int64_t ldx, const float* gamma,
                        const float* beta, const float* ss, int64_t ss_ld, int32_t ss_shift_off, int32_t mod_silu,
                        const void* res, int64_t ldr, void* dx, int32_t dx_f32, int64_t lddx, int64_t rows, int32_t n,
                        int32_t rows_per_sample, float* dgamma, float* dbeta, float* dss, int64_t dss_ld,
                        float* partial, hig_stream_t stream, hig_ln_reduce* deferred);
int hig_ln_bwd16_reduce_batch(const hig_ln_reduce* entries, int n, hipStream_t st);

// rowops.hip: fill / copy as kernels (never hipMemsetAsync / hipMemcpyAsync on a stream that may be under capture: see there)
int hig_zero_async(void* p, int64_t bytes, hipStream_t st);
int hig_copy_async(void* dst, const void* src, int64_t bytes, hipStream_t st);
