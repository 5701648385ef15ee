"""ctypes binding of libhig.so (the C ABI declared in include/hig.h).

The product path has NO CPU fallback: if the shared library is missing or a tensor is not on a
ROCm device, calls raise.  PyTorch is used only for device memory, streams and autograd plumbing.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhig.so")

NGLOBAL = 15
NLAYER = 36
ATTN_LINEAR, ATTN_FULL = 0, 1
PREC_F32, PREC_BF16X3, PREC_BF16 = 0, 1, 2
XF_NONE, XF_LN, XF_LN_MOD_SILU, XF_SILU = 0, 1, 2, 3
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_BIAS_POS, EPI_RES, EPI_DGELU, EPI_BIAS_SILU, EPI_BIAS_RES_SILU = range(9)
STORE_F32, STORE_BF16 = 0, 1
TAB_ROWS = 7
NORM_BLOCKS = 1024
COLSUM_CHUNKS = 512

# every symbol include/hig.h declares (checked by the CPU test-suite)
SYMBOLS = (
    "hig_version", "hig_last_error", "hig_workspace_bytes", "hig_textctx_bytes",
    "hig_bwd_workspace_bytes", "hig_text_context", "hig_denoiser_fwd", "hig_denoiser_bwd",
    "hig_gemm", "hig_gemm_ws", "hig_gemm_tail_ws_bytes", "hig_gemm_debug_stamps", "hig_rowstats", "hig_ln_mod_silu", "hig_linattn_ctx", "hig_linattn_apply", "hig_linattn_apply_bwd",
    "hig_linattn_ctx_bwd", "hig_linattn_bwd_scratch_floats", "hig_fullattn_fwd", "hig_fullattn_bwd", "hig_ln_bwd", "hig_ln_bwd_partial_floats", "hig_transpose", "hig_colsum", "hig_colsum_chunks",
    "hig_timestep_embedding", "hig_timestep_embedding_bf16", "hig_q_sample", "hig_p_sample_step", "hig_dec_timesteps",
    "hig_masked_mse", "hig_sumsq_partial", "hig_clip_adam",
    "hig_text_head_workspace_bytes", "hig_text_head_bwd_workspace_bytes", "hig_text_head_fwd", "hig_text_head_bwd",
    "hig_layernorm", "hig_gather_rows", "hig_scatter_add_rows", "hig_gather_frames", "hig_recover_joints", "hig_transpose_batch", "hig_linattn_ctx_scratch_floats", "hig_pair_mse",
    "hig_fullattn_fwd_kpad", "hig_eval_encoder_workspace_bytes", "hig_eval_encoder_fwd",
    "hig_clip_adam_lrdev", "hig_shutdown", "hig_gemm_split", "hig_gemm_split_scratch_floats",
    "hig_gemm_bf16", "hig_gemm_bf16_debug_stamps", "hig_gemm_ws16_debug_stamps", "hig_gemm_wsp16_debug_stamps", "hig_gemm_wsp32_debug_stamps", "hig_gemm_wsp32_launches", "hig_wgrad16_debug_stamps", "hig_linattn16_debug_stamps", "hig_cast_bf16", "hig_ln_bf16", "hig_linattn_ctx_bf16", "hig_linattn_apply_bf16",
    "hig_text_context_bf16", "hig_denoiser_fwd_bf16", "hig_linattn_apply_sty_bf16", "hig_linattn_apply_sty_mm16", "hig_linattn_apply_sty_mm16_y", "hig_linattn_ctx_mm16", "hig_linattn_apply_sty", "hig_joint_embed_bf16", "hig_joint_embed_bf16_w", "hig_attn_out16", "hig_rows_out16", "hig_weight_frag16", "hig_joint_embed_bf16_scratch_bytes", "hig_fullattn_fwd_bf16", "hig_denoiser_bwd_hooked", "hig_denoiser_bwd_bf16_hooked",
    # round 4: bf16-storage training step
    "hig_text_context_bf16_train", "hig_denoiser_fwd_bf16_train", "hig_denoiser_bwd_bf16", "hig_ln_bwd_bf16",
    "hig_linattn_apply_bwd_bf16", "hig_linattn_ctx_bwd_bf16", "hig_colsum_bf16", "hig_transpose_bf16_batch", "hig_transpose_bf16",
    "hig_gelu_bf16", "hig_cast_f32", "hig_cast_pad_bf16", "hig_gemm_bf16_split", "hig_gemm_bf16_split_scratch_floats", "hig_clip_adam_shadow",
    "hig_debug_marker", "hig_denoiser_fwd_text", "hig_wgrad_bf16", "hig_wgrad_bf16_scratch_floats", "hig_denoiser_fwd_x", "hig_denoiser_fwd_bf16_x",
)


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "T", "F", "d", "H", "ff", "L", "N", "Lt", "num_frames", "attn_kind", "prec", "two_person",
                 "storage")]


class TextDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "N", "W", "Lt", "H", "ff", "L", "E", "prec")]


T_NGLOBAL, T_NLAYER = 6, 12


class EvalDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "T", "F", "d", "H", "ff", "L", "C", "cls", "prec")]


EV_NGLOBAL = 12


class GemmDesc(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("ldx", C.c_int64), ("x_rs", C.c_int32),
        ("Y", C.c_void_p), ("ldy", C.c_int64), ("y_rs", C.c_int32),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("I", C.c_int32), ("J", C.c_int32), ("R", C.c_int32),
        ("xf", C.c_int32), ("xf_on_y", C.c_int32), ("epi", C.c_int32), ("prec", C.c_int32),
        ("bias", C.c_void_p),
        ("res", C.c_void_p), ("ldr", C.c_int64),
        ("aux", C.c_void_p), ("ldaux", C.c_int64),
        ("stats", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("ss", C.c_void_p), ("ss_ld", C.c_int64), ("ss_shift_off", C.c_int32),
        ("rows_per_sample", C.c_int32),
        ("pos", C.c_void_p), ("ldpos", C.c_int64), ("T", C.c_int32), ("pos_shift", C.c_int32),
        ("xcolsum", C.c_void_p),
        ("row_stats_out", C.c_void_p), ("row_stats_in", C.c_void_p), ("ln_colsum", C.c_void_p),
    ]


class Gemm16Desc(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("ldx", C.c_int64),
        ("Y", C.c_void_p), ("ldy", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("c_f32", C.c_int32),
        ("I", C.c_int32), ("J", C.c_int32), ("R", C.c_int32),
        ("epi", C.c_int32),
        ("bias", C.c_void_p),
        ("res", C.c_void_p), ("ldr", C.c_int64), ("res_f32", C.c_int32),
        ("row_stats_out", C.c_void_p), ("row_stats_in", C.c_void_p), ("ln_colsum", C.c_void_p),
        ("aux", C.c_void_p), ("ldaux", C.c_int64),
    ]


LAYER_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)

_lib = None


def lib():
    """Loads libhig.so once.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libhig.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        L.hig_version.restype = C.c_int
        L.hig_last_error.argtypes = [C.c_char_p, C.c_int]
        for fn in ("hig_workspace_bytes", "hig_textctx_bytes"):
            getattr(L, fn).restype = i64
            getattr(L, fn).argtypes = [C.POINTER(Dims), C.c_int]
        L.hig_bwd_workspace_bytes.restype = i64
        L.hig_bwd_workspace_bytes.argtypes = [C.POINTER(Dims)]
        L.hig_text_context.argtypes = [C.POINTER(Dims), vp, vp, vp, C.c_int, vp]
        L.hig_denoiser_fwd.argtypes = [C.POINTER(Dims), vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
        L.hig_denoiser_bwd.argtypes = [C.POINTER(Dims)] + [vp] * 14
        L.hig_denoiser_bwd_hooked.argtypes = [C.POINTER(Dims)] + [vp] * 14 + [LAYER_HOOK, vp, vp]
        L.hig_denoiser_bwd_bf16_hooked.argtypes = [C.POINTER(Dims)] + [vp] * 15 + [LAYER_HOOK, vp, vp]
        L.hig_gemm.argtypes = [C.POINTER(GemmDesc), vp]
        L.hig_gemm_ws.argtypes = [C.POINTER(GemmDesc), vp, i64, vp]
        L.hig_gemm_tail_ws_bytes.argtypes = []
        L.hig_gemm_tail_ws_bytes.restype = i64
        L.hig_gemm_debug_stamps.argtypes = [vp]
        L.hig_rowstats.argtypes = [vp, i64, i64, i32, vp, vp]
        L.hig_ln_mod_silu.argtypes = [vp, i64, i64, i32, vp, vp, vp, i64, i32, i32, vp, i64, vp, vp]
        L.hig_linattn_ctx.argtypes = [vp, vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp]
        L.hig_linattn_ctx_scratch_floats.restype = i64
        L.hig_linattn_ctx_scratch_floats.argtypes = [i32, i32, i32, i32]
        L.hig_linattn_apply.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, i32, vp]
        L.hig_linattn_bwd_scratch_floats.restype = i64
        L.hig_linattn_bwd_scratch_floats.argtypes = [i32, i32, i32, i32]
        L.hig_linattn_apply_bwd.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, vp, vp]
        L.hig_linattn_ctx_bwd.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp]
        L.hig_fullattn_fwd.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, vp, vp, i64, vp, vp]
        L.hig_fullattn_bwd.argtypes = [vp, i64, vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, vp, vp,
                                       vp, vp, i64, vp, vp, i64, vp]
        L.hig_ln_bwd.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp, i64, i32, i32, vp, i64, vp, i64,
                                 i64, i32, i32, vp, vp, vp, i64, vp, vp]
        L.hig_ln_bwd_partial_floats.restype = i64
        L.hig_ln_bwd_partial_floats.argtypes = [i64, i32, i32]
        L.hig_transpose.argtypes = [vp, i64, i32, i32, vp, i64, vp, vp, vp, vp]
        L.hig_colsum_chunks.argtypes = [i64]
        L.hig_colsum.argtypes = [vp, i64, i64, i32, vp, vp, vp]
        L.hig_timestep_embedding.argtypes = [vp, i32, i32, vp, vp]
        L.hig_timestep_embedding_bf16.argtypes = [vp, i32, i32, vp, vp]
        L.hig_q_sample.argtypes = [vp, vp, vp, vp, i32, i32, i64, vp, vp]
        L.hig_p_sample_step.argtypes = [vp, vp, vp, vp, vp, i32, i32, i64, vp, vp, vp]
        L.hig_dec_timesteps.argtypes = [vp, i32, vp]
        L.hig_masked_mse.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
        L.hig_pair_mse.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
        L.hig_sumsq_partial.argtypes = [vp, i64, f32, vp, vp]
        L.hig_clip_adam.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, vp, vp, vp, vp]
        L.hig_clip_adam_lrdev.argtypes = [vp, vp, vp, vp, i64, f32, vp, f32, f32, f32, f32, f32, vp, vp, vp, vp]
        L.hig_shutdown.argtypes = []
        L.hig_gemm_bf16.argtypes = [C.POINTER(Gemm16Desc), vp]
        L.hig_gemm_bf16_debug_stamps.argtypes = [vp]
        L.hig_gemm_ws16_debug_stamps.argtypes = [vp]
        L.hig_gemm_wsp16_debug_stamps.argtypes = [vp]
        L.hig_gemm_wsp32_debug_stamps.argtypes = [vp]
        L.hig_gemm_wsp32_launches.argtypes = []
        L.hig_gemm_wsp32_launches.restype = i64
        L.hig_wgrad16_debug_stamps.argtypes = [vp]
        L.hig_linattn16_debug_stamps.argtypes = [vp]
        L.hig_cast_bf16.argtypes = [vp, vp, i64, vp]
        L.hig_ln_bf16.argtypes = [vp, i32, i64, i64, i32, vp, vp, vp, i64, i32, i32, vp, i64, vp]
        L.hig_linattn_ctx_bf16.argtypes = [vp, vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
        L.hig_linattn_apply_bf16.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, i32, vp]
        L.hig_linattn_apply_sty_bf16.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, vp, i64, i32, i32, i32, i32, vp]
        L.hig_linattn_ctx_mm16.argtypes = [vp, vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp]
        L.hig_linattn_apply_sty_mm16.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, vp, i64, i32, i32, i32, i32, vp]
        L.hig_linattn_apply_sty_mm16_y.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, vp, i64, vp, i64, i32, i32, i32, i32, vp]
        L.hig_fullattn_fwd_bf16.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, vp, vp, i64, vp]
        L.hig_joint_embed_bf16_scratch_bytes.argtypes = [i32, i32]
        L.hig_joint_embed_bf16_scratch_bytes.restype = i64
        L.hig_joint_embed_bf16.argtypes = [vp, i64, i32, vp, vp, vp, i64, i32, i32, vp, i64, i32, vp, vp]
        L.hig_attn_out16.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, vp, vp, vp, i64, vp, i32, i32, i32, i32, vp]
        L.hig_rows_out16.argtypes = [vp, i64, vp, vp, vp, i64, i32, vp, vp, vp, i64, vp, i32, i32, i32, vp]
        L.hig_weight_frag16.argtypes = [vp, i64, i32, i32, vp, vp]
        L.hig_joint_embed_bf16_w.argtypes = [vp, i64, i32, vp, vp, vp, i64, i32, i32, vp, i64, i32, vp]
        L.hig_linattn_apply_sty.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, vp, i64, i32, i32, i32, i32, vp]
        L.hig_text_context_bf16.argtypes = [C.POINTER(Dims), vp, vp, vp, vp, vp]
        L.hig_denoiser_fwd_bf16.argtypes = [C.POINTER(Dims), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.hig_denoiser_fwd_bf16_x.argtypes = [C.POINTER(Dims), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.hig_gemm_split.argtypes = [C.POINTER(GemmDesc), i32, vp, i64, vp]
        L.hig_gemm_split_scratch_floats.restype = i64
        L.hig_gemm_split_scratch_floats.argtypes = [C.POINTER(GemmDesc), i32]
        L.hig_text_head_workspace_bytes.restype = i64
        L.hig_text_head_workspace_bytes.argtypes = [C.POINTER(TextDims), C.c_int]
        L.hig_text_head_bwd_workspace_bytes.restype = i64
        L.hig_text_head_bwd_workspace_bytes.argtypes = [C.POINTER(TextDims)]
        L.hig_text_head_fwd.argtypes = [C.POINTER(TextDims), vp, vp, vp, vp, vp, vp, C.c_int, vp]
        L.hig_text_head_bwd.argtypes = [C.POINTER(TextDims)] + [vp] * 11
        L.hig_fullattn_fwd_kpad.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, vp, vp, vp, i64, vp, vp]
        L.hig_eval_encoder_workspace_bytes.restype = i64
        L.hig_eval_encoder_workspace_bytes.argtypes = [C.POINTER(EvalDims)]
        L.hig_eval_encoder_fwd.argtypes = [C.POINTER(EvalDims)] + [vp] * 8
        L.hig_layernorm.argtypes = [vp, i64, i64, i32, vp, vp, vp, i64, vp, vp]
        L.hig_gather_rows.argtypes = [vp, i64, i32, i32, vp, i32, vp, i64, vp]
        L.hig_scatter_add_rows.argtypes = [vp, i64, i32, i32, vp, i32, vp, i64, vp]
        L.hig_transpose_batch.argtypes = [i32, vp, vp, vp, vp, vp]
        L.hig_recover_joints.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
        L.hig_gather_frames.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
        L.hig_text_context_bf16_train.argtypes = [C.POINTER(Dims), vp, vp, vp, vp, vp]
        L.hig_denoiser_fwd_bf16_train.argtypes = [C.POINTER(Dims)] + [vp] * 10
        L.hig_denoiser_bwd_bf16.argtypes = [C.POINTER(Dims)] + [vp] * 15
        L.hig_ln_bwd_bf16.argtypes = [vp, i64, vp, i32, i64, vp, vp, vp, i64, i32, i32, vp, i64, vp, i32, i64, i64, i32, i32,
                                      vp, vp, vp, i64, vp, vp]
        L.hig_linattn_apply_bwd_bf16.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, vp, vp]
        L.hig_linattn_ctx_bwd_bf16.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]
        L.hig_colsum_bf16.argtypes = [vp, i64, i64, i32, vp, vp, vp]
        L.hig_transpose_bf16_batch.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp]
        L.hig_transpose_bf16.argtypes = [vp, i64, i32, i32, vp, i64, vp]
        L.hig_gelu_bf16.argtypes = [vp, vp, i64, vp]
        L.hig_cast_f32.argtypes = [vp, vp, i64, vp]
        L.hig_cast_pad_bf16.argtypes = [vp, i64, i64, i32, vp, i64, vp]
        L.hig_gemm_bf16_split.argtypes = [C.POINTER(Gemm16Desc), i32, vp, i64, vp]
        L.hig_gemm_bf16_split_scratch_floats.restype = i64
        L.hig_gemm_bf16_split_scratch_floats.argtypes = [C.POINTER(Gemm16Desc), i32]
        L.hig_clip_adam_shadow.argtypes = [vp, vp, vp, vp, i64, f32, vp, f32, f32, f32, f32, f32, vp, vp, vp, vp, i64, vp]
        L.hig_debug_marker.argtypes = [i32, vp]
        L.hig_denoiser_fwd_x.argtypes = [C.POINTER(Dims), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
        L.hig_wgrad_bf16.argtypes = [vp, i64, vp, i64, i64, i32, i32, vp, vp, i32, vp, i64, vp]
        L.hig_wgrad_bf16_scratch_floats.restype = i64
        L.hig_wgrad_bf16_scratch_floats.argtypes = [i32, i32, i32]
        L.hig_denoiser_fwd_text.argtypes = [C.POINTER(Dims), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    lib().hig_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc):
    if rc != 0:
        raise RuntimeError("libhig error %d: %s" % (rc, last_error()))


def stream_ptr():
    """The hipStream_t torch is currently launching on (so torch.cuda.graph captures us)."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: no CPU path exists."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libhig needs ROCm device tensors (got a %s tensor); there is no CPU fallback"
                           % t.device.type)
    return C.c_void_p(t.data_ptr())
