"""ctypes binding of libldmae_hip.so (include/ldmae_hip.h).

There is NO fallback: if the shared library is missing or a call fails the
caller gets a RuntimeError.  torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libldmae_hip.so")
LIB_PATH = os.environ.get("LDMAE_HIP_LIB", LIB_PATH)      # A/B builds of the same library (tools/); unset = the in-tree build

F32, BF16, F16 = 0, 1, 2
EPI_BIAS, EPI_GATE_RES, EPI_BIAS_POS, EPI_BIAS_GELU, EPI_SWIGLU, EPI_SWIGLU_BWD, EPI_GELU_BWD = 0, 1, 2, 3, 4, 5, 6

_vp, _i, _l, _f, _d = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double

# name -> (restype, argtypes); mirrors include/ldmae_hip.h one to one
SIGNATURES = {
    "ldmae_last_error": (C.c_char_p, []),
    "ldmae_version": (C.c_char_p, []),
    "ldmae_arch": (C.c_char_p, []),
    "ldmae_gemm_nt": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _i, _i, _vp]),
    "ldmae_gemm_nt_qkv_rope_ok": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "ldmae_gemm_nt_qkv_rope": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "ldmae_gemm_tn_splits": (_i, [_i, _i, _i, _i]),
    "ldmae_gemm_tn_workspace_bytes": (_l, [_i, _i, _i, _i]),
    "ldmae_gemm_tn": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _vp, _l, _vp]),
    "ldmae_colsum_workspace_bytes": (_l, [_i, _i]),
    "ldmae_colsum": (_i, [_i, _vp, _i, _i, _i, _vp, _f, _vp, _vp]),
    "ldmae_cast_weight": (_i, [_i, _vp, _vp, _vp, _i, _i, _vp]),
    "ldmae_cast": (_i, [_i, _i, _vp, _vp, _l, _vp]),
    "ldmae_cast_stack": (_i, [_i, _vp, _i, _l, _vp, _vp]),
    "ldmae_multi_add": (_i, [_i, _vp, _vp, _vp, _vp]),
    "ldmae_thin_nt": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ldmae_thin_tn_workspace_bytes": (_l, [_i, _i, _i]),
    "ldmae_thin_tn": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _l, _vp]),
    "ldmae_rmsnorm_modulate_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _f, _vp]),
    "ldmae_rmsnorm_modulate_bwd_workspace_bytes": (_l, [_i, _i, _i]),
    "ldmae_rmsnorm_modulate_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _i, _vp, _f, _i, _i, _i, _vp, _vp]),
    "ldmae_rmsnorm_modulate_bwd_gate_workspace_bytes": (_l, [_i, _i, _i]),
    "ldmae_rmsnorm_modulate_bwd_gate": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _i, _vp, _f, _vp, _vp, _i, _vp, _vp, _i, _vp,
                                             _i, _i, _i, _vp, _vp]),
    "ldmae_qknorm_rope_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ldmae_qknorm_rope_bwd_workspace_bytes": (_l, [_i, _i, _i, _i]),
    "ldmae_qknorm_rope_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "ldmae_rope": (_i, [_i, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    "ldmae_attention_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ldmae_attention_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ldmae_attention_fwd_qkv": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ldmae_attention_bwd_qkv": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ldmae_attention_fwd_pv": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ldmae_k_norm_max": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ldmae_attention_fwd_qkv_bounded": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ldmae_attention_fwd_pv_bounded": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ldmae_qk_score_bound": (_i, [_vp, _vp, _i, _f, _vp, _vp]),
    "ldmae_attention_bwd_pv": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ldmae_attention_bwd_pv_qknorm_workspace_bytes": (_l, [_i, _i, _i, _i]),
    "ldmae_attention_bwd_pv_qknorm": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                           _f, _vp]),
    "ldmae_swiglu_fwd": (_i, [_i, _vp, _vp, _i, _i, _vp]),
    "ldmae_swiglu_bwd": (_i, [_i, _vp, _vp, _vp, _i, _i, _vp]),
    "ldmae_gate_bwd_workspace_bytes": (_l, [_i, _i, _i]),
    "ldmae_gate_bwd": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "ldmae_timestep_embedding": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "ldmae_silu_fwd": (_i, [_i, _vp, _vp, _l, _vp]),
    "ldmae_silu_bwd": (_i, [_vp, _vp, _vp, _l, _vp]),
    "ldmae_label_embed_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ldmae_label_embed_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ldmae_adamw_ema": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _d, _d, _d, _d, _d, _d, _d, _vp]),
    "ldmae_ema_only": (_i, [_vp, _vp, _l, _d, _vp]),
    "ldmae_random_masking": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ldmae_patch_gather": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ldmae_latent_prologue": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp]),
    "ldmae_gather_rows": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ldmae_scatter_rows": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ldmae_restore_tokens": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ldmae_restore_tokens_bwd_workspace_bytes": (_l, [_i, _i, _i]),
    "ldmae_restore_tokens_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ldmae_vmae_encoder_blob_bytes": (_l, [_i]),
    "ldmae_vmae_encoder_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ldmae_vmae_encoder_fwd_tiled_workspace_bytes": (_l, [_i, _i]),
    "ldmae_vmae_encoder_fwd_tiled": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ldmae_vmae_encoder_fwd_tiled_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "ldmae_layernorm_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "ldmae_layernorm_bwd_workspace_bytes": (_l, [_i, _i]),
    "ldmae_layernorm_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp]),
    "ldmae_mae_loss_groups": (_l, [_l]),
    "ldmae_mae_loss_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ldmae_mae_loss_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ldmae_layernorm_bwd_cast": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp]),
    "ldmae_gelu_fwd": (_i, [_i, _vp, _vp, _l, _vp]),
    "ldmae_gelu_bwd": (_i, [_i, _vp, _vp, _vp, _l, _vp]),
    "ldmae_gelu_tanh_fwd": (_i, [_i, _vp, _vp, _l, _vp]),
    "ldmae_gelu_tanh_bwd": (_i, [_i, _vp, _vp, _vp, _l, _vp]),
    "ldmae_layernorm_modulate_fwd": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _f, _vp]),
    "ldmae_layernorm_modulate_bwd": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ldmae_layernorm_modulate_bwd_gate": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _f, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "ldmae_conv3x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ldmae_conv3x3_bwd_workspace_bytes": (_l, [_i]),
    "ldmae_conv3x3_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ldmae_prof_enable": (_i, [_i]),
    "ldmae_prof_collect": (_i, [C.POINTER(_d), C.POINTER(_d), C.POINTER(_l)]),
    "ldmae_launch_counts": (_i, [C.POINTER(_l), _i, _i]),
}
# csrc/probe/ldmae_diag.h: present only in the diagnostic build (LDMAE_HIP_LIB=.../libldmae_hip_diag.so, used by tools/)
DIAG_SIGNATURES = {
    "ldmae_tune": (_i, [_i, _i]),
    "ldmae_tune_query": (_i, [_i]),
    "ldmae_debug_nt_stamps": (None, [_vp]),
}
EPI_TILE_LAUNCH = 0x100
EPI_HALF_LINES = 0x200
EPI_F16_INF = 0x400

_lib = None


def load():
    """dlopen the library (once) and declare every prototype.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ldmae_amd/csrc`).  ldmae_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if hasattr(lib, "ldmae_tune"):          # diagnostic build: LDMAE_TUNE="key=value,..." presets its A/B knobs for profiling runs
        for name, (res, args) in DIAG_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        for kv in filter(None, os.environ.get("LDMAE_TUNE", "").split(",")):
            k, v = kv.split("=")
            lib.ldmae_tune(int(k), int(v))
    elif os.environ.get("LDMAE_TUNE"):
        raise RuntimeError("LDMAE_TUNE is set but the loaded library has no A/B knobs: point LDMAE_HIP_LIB at "
                           "ldmae_amd/libldmae_hip_diag.so (make -C ldmae_amd/csrc diag)")
    _lib = lib
    return lib


def last_error() -> str:
    return load().ldmae_last_error().decode()


def call(name, *args):
    """Invoke an int-returning entry point; raise RuntimeError with the library's message on failure."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {last_error()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be CUDA/HIP and contiguous."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("ldmae_amd ops need tensors on a HIP device (no CPU fallback); got " + str(t.device))
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def dt(dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:          # the TF32-class forward path (LDMAE_F16: forward-only entry points)
        return F16
    raise RuntimeError(f"unsupported activation dtype {dtype} (float32, bfloat16, or float16 on the forward-only TF32-class path)")


COUNT_NAMES = ("nt_bf16", "nt_f32", "tn_bf16", "tn_f32", "attn_bf16", "attn_f32", "nt_f16", "attn_f16", "tn_f16")


def launch_counts(reset: bool = False) -> dict:
    """Launch counts by kernel family since the last reset (ldmae_launch_counts): which arithmetic type the calls were dispatched to."""
    arr = (C.c_long * 9)()
    call("ldmae_launch_counts", arr, 9, 1 if reset else 0)
    return dict(zip(COUNT_NAMES, [int(v) for v in arr]))
