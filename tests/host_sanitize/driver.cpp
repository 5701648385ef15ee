// Host-side entry points of libhig under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on
// the CPU build only; GPU ASan is not available on this pool).  Everything called here returns BEFORE any HIP call:
// size / layout queries (hig_*_bytes, *_scratch_*), and argument validation that must reject bad descriptors with an
// error code -- never read through a null pointer, never overflow an index computation.  Built by `make sanitize`
// (csrc/Makefile: every .hip compiled --cuda-host-only with -fsanitize=address,undefined) and run by
// tests/test_cpu_host.py::test_host_entry_points_under_asan_ubsan.
#include <initializer_list>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hig.h"

// The objects are compiled --cuda-host-only: there is no device image to register.  These definitions (in the executable,
// so they win over libamdhip64's) turn the module constructors of the translation units into no-ops.
extern "C" {
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipRegisterManagedVar(void**, void**, void*, const char*, size_t, unsigned) {}
void __hipUnregisterFatBinary(void**) {}
}

static int failures = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) {                                                       \
      fprintf(stderr, "driver.cpp:%d: expectation failed: %s\n", __LINE__, #cond); \
      ++failures;                                                        \
    }                                                                    \
  } while (0)

static hig_dims dims(int B, int T, int F, int d, int H, int ff, int L, int two, int storage, int attn, int prec) {
  hig_dims D;
  memset(&D, 0, sizeof(D));
  D.B = B; D.T = T; D.F = F; D.d = d; D.H = H; D.ff = ff; D.L = L; D.N = 77; D.Lt = 256; D.num_frames = T > 196 ? T : 196;
  D.attn_kind = attn; D.prec = prec; D.two_person = two; D.storage = storage;
  return D;
}

int main() {
  char msg[512];
  EXPECT(hig_version() > 0);
  // ---- workspace layouts over a sweep of legal shapes (every index computation of the layout builders) ----
  const int shapes[][7] = {{2, 16, 12, 64, 8, 128, 2},    {2, 60, 150, 128, 8, 256, 4},  {64, 196, 150, 512, 8, 1024, 8},
                           {32, 300, 150, 1024, 8, 1024, 12}, {64, 91, 263, 512, 8, 1024, 8}, {1, 1, 4, 32, 4, 32, 1},
                           {1024, 196, 263, 512, 8, 2048, 8}};
  for (const auto& s : shapes)
    for (int two = 0; two <= 2; ++two)
      for (int storage = 0; storage <= 1; ++storage)
        for (int attn = 0; attn <= 1; ++attn) {
          if (two && (s[0] & 1)) continue;
          hig_dims D = dims(s[0], s[1], s[2], s[3], s[4], s[5], s[6], two, storage, attn, 0);
          const int hd = D.d / D.H;
          const bool bf16_ok = (hd == 64 || hd == 128) && !(two && attn);
          for (int training = 0; training <= 1; ++training) {
            const int64_t w = hig_workspace_bytes(&D, training), t = hig_textctx_bytes(&D, training);
            // bf16 storage: inference for head dim 64 / 128; training with linear attention (single-person and two-person)
            const bool bf16_train_ok = bf16_ok && !attn;
            if (storage == 1 && (!bf16_ok || (training && !bf16_train_ok))) { EXPECT(w < 0 && t < 0); continue; }
            if (two && attn) { EXPECT(w < 0); continue; }
            EXPECT(w > 0 && t > 0);
          }
          if (storage == 0 && !(two && attn)) EXPECT(hig_bwd_workspace_bytes(&D) > 0);
          if (storage == 1) EXPECT((hig_bwd_workspace_bytes(&D) > 0) == (bf16_ok && !attn));
        }
  // ---- illegal dims: rejected with a message, nothing dereferenced ----
  {
    hig_dims D = dims(2, 16, 12, 64, 8, 128, 2, 0, 0, 0, 0);
    hig_dims bad = D; bad.H = 7;              EXPECT(hig_workspace_bytes(&bad, 0) < 0);
    bad = D; bad.T = 500;                     EXPECT(hig_workspace_bytes(&bad, 0) < 0);      // T > num_frames
    bad = D; bad.B = 0;                       EXPECT(hig_workspace_bytes(&bad, 0) < 0);
    bad = D; bad.L = -1;                      EXPECT(hig_workspace_bytes(&bad, 1) < 0);
    bad = D; bad.prec = 99;                   EXPECT(hig_workspace_bytes(&bad, 0) < 0);
    bad = D; bad.two_person = 1; bad.B = 3;   EXPECT(hig_workspace_bytes(&bad, 0) < 0);
    EXPECT(hig_workspace_bytes(nullptr, 0) < 0);
    EXPECT(hig_last_error(msg, sizeof(msg)) >= 0 && strlen(msg) > 0);
    EXPECT(hig_last_error(msg, 4) >= 0 && strlen(msg) <= 3);                               // truncation, no overrun
    // entry points with null arguments / mismatched storage
    EXPECT(hig_denoiser_fwd(&D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != HIG_OK);
    EXPECT(hig_denoiser_fwd_bf16(&D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != HIG_OK);
    EXPECT(hig_cast_pad_bf16(nullptr, 150, 8, 150, nullptr, 160, nullptr) != HIG_OK);
    EXPECT(hig_denoiser_fwd_bf16_x(&D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != HIG_OK);
    EXPECT(hig_text_context(&D, nullptr, nullptr, nullptr, 0, nullptr) != HIG_OK);
    // round 4: the bf16-storage training entry points (fp32-storage dims, null arguments, shapes they do not train)
    EXPECT(hig_denoiser_fwd_bf16_train(&D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != HIG_OK);
    EXPECT(hig_denoiser_bwd_bf16(&D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != HIG_OK);
    {
      hig_dims T16 = dims(2, 60, 150, 512, 8, 1024, 2, 0, 1, 0, 0);
      EXPECT(hig_text_context_bf16_train(&T16, nullptr, nullptr, nullptr, nullptr, nullptr) != HIG_OK);
      EXPECT(hig_denoiser_fwd_bf16_train(&T16, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != HIG_OK);
      T16.attn_kind = HIG_ATTN_FULL;
      EXPECT(hig_workspace_bytes(&T16, 1) < 0 && hig_workspace_bytes(&T16, 0) > 0);
    }
    EXPECT(hig_ln_bwd_bf16(nullptr, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, 0, nullptr, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr) != HIG_OK);
    EXPECT(hig_colsum_bf16(nullptr, 0, 0, 0, nullptr, nullptr, nullptr) != HIG_OK);
    EXPECT(hig_transpose_bf16_batch(13, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != HIG_OK);
    EXPECT(hig_gemm_bf16_split(nullptr, 0, nullptr, 0, nullptr) != HIG_OK);
    EXPECT(hig_clip_adam_shadow(nullptr, nullptr, nullptr, nullptr, 0, 0.f, nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != HIG_OK);
  }
  // ---- scratch-size queries ----
  EXPECT(hig_gemm_tail_ws_bytes() > 0);
  EXPECT(hig_joint_embed_bf16_scratch_bytes(150, 512) == 512 * 160 * 2);
  EXPECT(hig_joint_embed_bf16_scratch_bytes(0, 512) < 0 && hig_joint_embed_bf16_scratch_bytes(150, -1) < 0);
  for (int hd : {8, 16, 32, 64, 128})
    for (int B : {1, 32, 64, 1024}) {
      EXPECT(hig_linattn_ctx_scratch_floats(B, 196, 8, hd) >= 0);
      EXPECT(hig_linattn_bwd_scratch_floats(B, 196, 8, hd) >= 0);
    }
  EXPECT(hig_ln_bwd_partial_floats(12544, 512, 196) > 0);
  EXPECT(hig_colsum_chunks(12544) > 0 && hig_colsum_chunks(1) > 0);
  // ---- GEMM descriptors: validation before any launch ----
  {
    hig_gemm_desc g;
    memset(&g, 0, sizeof(g));
    EXPECT(hig_gemm(nullptr, nullptr) != HIG_OK);
    EXPECT(hig_gemm(&g, nullptr) != HIG_OK);                           // null operands
    EXPECT(hig_gemm_split_scratch_floats(&g, 4) >= 0 || hig_gemm_split_scratch_floats(&g, 4) < 0);
    EXPECT(hig_gemm_ws(&g, nullptr, 0, nullptr) != HIG_OK);
    float dummy[64] = {0};
    g.X = dummy; g.Y = dummy; g.C = dummy; g.I = -5; g.J = 4; g.R = 4; g.ldx = 4; g.ldy = 4; g.ldc = 4;
    EXPECT(hig_gemm(&g, nullptr) != HIG_OK);                           // negative extent
    hig_gemm16_desc h;
    memset(&h, 0, sizeof(h));
    EXPECT(hig_gemm_bf16(nullptr, nullptr) != HIG_OK);
    EXPECT(hig_gemm_bf16(&h, nullptr) != HIG_OK);
    alignas(16) static unsigned short b16[64 * 48];
    h.X = b16; h.Y = b16; h.C = b16; h.I = 64; h.J = 64; h.R = 48; h.ldx = 48; h.ldy = 48; h.ldc = 64;
    EXPECT(hig_gemm_bf16(&h, nullptr) == HIG_EUNSUPPORTED);            // R % 32 != 0
    // round-3 entry points: argument validation before any launch
    alignas(16) static float f32v[512];
    EXPECT(hig_attn_out16(nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, 1, 1, 8, 64, nullptr) != HIG_OK);
    EXPECT(hig_attn_out16(b16, 1536, b16, f32v, f32v, f32v, 1024, 512, b16, f32v, b16, 512, nullptr, 1, 4, 4, 64, nullptr) == HIG_EUNSUPPORTED);   // 4 heads
    EXPECT(hig_attn_out16(b16, 1535, b16, f32v, f32v, f32v, 1024, 512, b16, f32v, b16, 512, nullptr, 1, 4, 8, 64, nullptr) != HIG_OK);            // ldq % 8
    EXPECT(hig_rows_out16(nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, 1, 1, 512, nullptr) != HIG_OK);
    EXPECT(hig_rows_out16(b16, 512, f32v, f32v, f32v, 1024, 512, b16, f32v, b16, 512, nullptr, 1, 4, 256, nullptr) == HIG_EUNSUPPORTED);          // d != 512
    EXPECT(hig_weight_frag16(b16, 40, 33, 40, b16, nullptr) != HIG_OK);                                                                           // J % 32 != 0
    EXPECT(hig_weight_frag16(nullptr, 0, 0, 0, nullptr, nullptr) != HIG_OK);
    EXPECT(hig_joint_embed_bf16_w(nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, 0, 0, nullptr) != HIG_OK);
    EXPECT(hig_joint_embed_bf16_w(f32v, 4, 600, b16, f32v, f32v, 512, 4, 0, b16, 512, 512, nullptr) == HIG_EUNSUPPORTED);                         // F > 512
    EXPECT(hig_linattn_ctx_mm16(b16, b16, 1536, 1, 4, 8, 32, nullptr, f32v, f32v, nullptr, nullptr) == HIG_EUNSUPPORTED);                         // head dim 32
    EXPECT(hig_linattn_apply_sty_mm16(b16, 1536, b16, f32v, f32v, f32v, 1024, 512, b16, 512, 1, 4, 3, 64, nullptr) == HIG_EUNSUPPORTED);         // 3 heads
    EXPECT(hig_timestep_embedding_bf16(nullptr, 1, 64, nullptr, nullptr) != HIG_OK);
    h.R = 32; h.ldx = 33;
    EXPECT(hig_gemm_bf16(&h, nullptr) == HIG_EINVAL);                  // leading dimension not a multiple of 8
    h.ldx = 48; h.epi = HIG_EPI_BIAS_RES; h.bias = reinterpret_cast<const float*>(dummy);
    EXPECT(hig_gemm_bf16(&h, nullptr) == HIG_EINVAL);                  // residual epilogue without a residual
    h.I = 0; h.epi = HIG_EPI_NONE;
    EXPECT(hig_gemm_bf16(&h, nullptr) == HIG_OK);                      // empty problem: nothing to launch
    EXPECT(hig_cast_bf16(nullptr, nullptr, 8, nullptr) != HIG_OK);
    EXPECT(hig_cast_bf16(dummy, b16, 0, nullptr) == HIG_OK);
    EXPECT(hig_joint_embed_bf16(nullptr, 4, 150, nullptr, nullptr, nullptr, 512, 196, 0, nullptr, 512, 512, nullptr, nullptr) != HIG_OK);
  }
  // ---- diagnostics pointers: set and cleared ----
  EXPECT(hig_gemm_bf16_debug_stamps(nullptr) == HIG_OK && hig_gemm_ws16_debug_stamps(nullptr) == HIG_OK &&
         hig_gemm_debug_stamps(nullptr) == HIG_OK);
  if (failures) {
    fprintf(stderr, "%d expectation(s) failed\n", failures);
    return 1;
  }
  printf("host entry points clean\n");
  return 0;
}
