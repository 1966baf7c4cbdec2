"""Thin tensor-level wrappers over the C ABI (one python function per entry point).

Every function allocates its outputs with torch (caching allocator) and enqueues
the kernel on torch's current stream.  No function here computes anything in
PyTorch: if the library is missing, or a tensor is on the CPU, they raise.
"""
from __future__ import annotations

import os
import weakref

import torch

from . import _lib as L
from ._lib import (BF16, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_POS, EPI_F16_INF, EPI_GATE_RES, EPI_GELU_BWD, EPI_HALF_LINES, EPI_SWIGLU, EPI_SWIGLU_BWD, EPI_TILE_LAUNCH, F32, call, dt,
                   ptr, stream)

_ws = {}


def workspace(nbytes: int, device, slot: str = "main") -> torch.Tensor:
    """Grow-only f32 scratch buffer per (device, slot, STREAM): the library's entry points take their scratch from the caller and use it only inside
    the launches they enqueue, so two calls may share a buffer exactly when they are ordered on one stream -- work enqueued on different streams
    (the weight-gradient side stream, a data-parallel reducer, two modules driven from two host threads) gets buffers of its own."""
    key = (device, slot, torch.cuda.current_stream(device).cuda_stream)
    n = max(1, (int(nbytes) + 3) // 4)
    t = _ws.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(int(n * 1.25) + 1024, dtype=torch.float32, device=device)
        _ws[key] = t
    return t


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- side stream for weight gradients
_side = {}


def side_stream(device):
    """One extra HIP stream per device for work that is off the backward critical path (the dW = dY^T X GEMMs: their results are
    only needed by the optimizer).  OPT-IN (LDMAE_TN_STREAM=1): measured 3-5 % SLOWER on MI355X -- the 160-KiB-LDS GEMM workgroups
    cannot share a CU with the persistent NT GEMM or the attention workgroups, so the streams mostly take turns and lose L2 locality."""
    st = _side.get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side[device] = st
    return st


class SideGemms:
    """dW GEMMs of one backward on the side stream: `tn(a, b)` is ordered after everything enqueued so far on the current stream;
    `join()` makes the current stream wait for all of them (call before handing the results to autograd)."""

    def __init__(self, device, enabled=True):
        self.enabled = enabled and os.environ.get("LDMAE_TN_STREAM", "0") == "1"
        self.main = torch.cuda.current_stream(device)
        self.side = side_stream(device) if self.enabled else None

    def tn(self, a, b, out=None, beta=0.0):
        if not self.enabled:
            return gemm_tn(a, b, out=out, beta=beta)
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            return gemm_tn(a, b, out=out, beta=beta, ws_slot="tn_side")

    def join(self):
        if self.enabled:
            self.main.wait_stream(self.side)


# ----------------------------------------------------------------------------- GEMMs
# Launch mode of the bf16 GEMMs, owned by the CALLER (a train driver that knows it shares the chip with RCCL's collective kernels sets
# "tile"; everything else keeps "persistent") and handed to the library PER CALL as a flag in `epi` -- the C ABI keeps no mode.
# Until a driver says otherwise ("auto"): persistent, except in a process that belongs to a torch.distributed world of more than one rank
# (the reference's own train_accum.py under accelerate / DDP through the drop-in: nobody there calls set_gemm_launch_mode) -> tile.
_GEMM_MODE = "auto"
_AUTO_FLAG = None          # resolved once torch.distributed is initialised (the world does not change afterwards)


def set_gemm_launch_mode(mode: str) -> None:
    """"persistent" (one workgroup per CU walks the tiles), "tile" (one 256x256 tile per workgroup; bitwise-equal results) or "auto"."""
    global _GEMM_MODE
    if mode not in ("persistent", "tile", "auto"):
        raise ValueError(f"gemm launch mode {mode!r}: 'persistent', 'tile' or 'auto'")
    _GEMM_MODE = mode


_HALF_LINES = 0            # EPI_HALF_LINES while a test / tool asks for the half-line NT kernel (set_gemm_half_lines)


def set_gemm_half_lines(on: bool) -> None:
    """A/B switch of tests and tools: bf16 NT GEMMs keep the half-line kernel (gemm_nt_persist_kernel) where the whole-line kernel
    (gemm_nt_lines.hip, the default) would run.  Bitwise-equal results; a per-call flag of the C ABI like the launch mode."""
    global _HALF_LINES
    _HALF_LINES = EPI_HALF_LINES if on else 0


def _launch_flag() -> int:
    global _AUTO_FLAG
    if _GEMM_MODE != "auto":
        return (EPI_TILE_LAUNCH if _GEMM_MODE == "tile" else 0) | _HALF_LINES
    if _AUTO_FLAG is not None:
        return _AUTO_FLAG | _HALF_LINES
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return _HALF_LINES
    _AUTO_FLAG = EPI_TILE_LAUNCH if dist.get_world_size() > 1 else 0
    return _AUTO_FLAG | _HALF_LINES


def gemm_launch_mode() -> str:
    return "tile" if (_launch_flag() & EPI_TILE_LAUNCH) else "persistent"


def _grad_flag(dtype, grad) -> int:
    """fp16 GRADIENT outputs overflow to infinity (torch's fp16 autocast semantics: the loss scaler then skips the step); forward fp16 outputs
    saturate at +-65504.  A per-call flag of the C ABI (LDMAE_EPI_F16_INF)."""
    return EPI_F16_INF if (grad and dtype == torch.float16) else 0


def gemm_nt(a, b, bias=None, out_dtype=None, out=None, beta=0.0, grad=False):
    """out[M,N] = a[M,K] @ b[N,K]^T + bias (+ beta*out)."""
    M, K = a.shape
    N = b.shape[0]
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    call("ldmae_gemm_nt", dt(a.dtype), dt(out.dtype), EPI_BIAS | _launch_flag() | _grad_flag(out.dtype, grad), ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0),
         M, N, K, ptr(bias), float(beta), None, None, None, 0, 0, stream())
    return out


def gemm_nt_gate_res(a, b, bias, xin, gate, rows_per_batch, save_y=True, xout=None, y_dtype=None):
    """y = a @ b^T + bias ; xout = xin + gate[batch] * y.  Returns (xout, y or None).  gate: [B, D] view (any row stride).
    The residual adds y ROUNDED to `y_dtype` (default: the activation type -- what the reference's autocast Linear hands to
    `x + gate * branch`, lightningdit.py:248-249); y_dtype=float32 with save_y=False adds the unrounded product."""
    M, K = a.shape
    N = b.shape[0]
    y_dtype = y_dtype or a.dtype
    y = torch.empty(M, N, dtype=y_dtype, device=a.device) if save_y else None
    if xout is None:
        xout = torch.empty_like(xin)
    call("ldmae_gemm_nt", dt(a.dtype), dt(y_dtype), EPI_GATE_RES | _launch_flag(), ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(y), N,
         M, N, K, ptr(bias), 0.0, ptr(xin), ptr(xout), ptr(gate), gate.stride(0) if gate is not None else 0, rows_per_batch, stream())
    return xout, y


def gemm_nt_pos(a, b, bias, pos, rows_per_batch):
    """out = a @ b^T + bias + pos[row % rows_per_batch]   (f32 out; patch embed)."""
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    call("ldmae_gemm_nt", dt(a.dtype), F32, EPI_BIAS_POS | _launch_flag(), ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), N,
         M, N, K, ptr(bias), 0.0, ptr(pos), None, None, 0, rows_per_batch, stream())
    return out


def gemm_nt_gelu(a, b, bias, save_pre=True):
    """(gelu(a @ b^T + bias), pre-activation or None)."""
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    pre = torch.empty_like(out) if save_pre else None
    call("ldmae_gemm_nt", dt(a.dtype), dt(a.dtype), EPI_BIAS_GELU | _launch_flag(), ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), N,
         M, N, K, ptr(bias), 0.0, None, ptr(pre), None, 0, 0, stream())
    return out, pre


def gemm_nt_gelu_bwd(dy, w2t, pre):
    """dpre = gelu_bwd(dy @ w2t^T, pre): the input gradient of fc2 with the GELU backward in the GEMM's epilogue (w2t = [hidden, D] transposed
    copy of fc2's weight; pre = fc1's pre-activation as gemm_nt_gelu saved it).  Same bits as gelu_bwd(gemm_nt(dy, w2t), pre)."""
    M, K = dy.shape
    N = w2t.shape[0]
    out = torch.empty(M, N, dtype=dy.dtype, device=dy.device)
    call("ldmae_gemm_nt", dt(dy.dtype), dt(dy.dtype), EPI_GELU_BWD | _launch_flag() | _grad_flag(dy.dtype, True), ptr(dy), dy.stride(0), ptr(w2t), w2t.stride(0), ptr(out), N,
         M, N, K, None, 0.0, ptr(pre), None, None, 0, 0, stream())
    return out


def gemm_nt_swiglu(a, w12, b12, save_h12=True):
    """(h12, hid): h12 = a @ w12^T + b12 ([x1 | x2]), hid = silu(x1) * x2.  bf16: one GEMM with the SwiGLU epilogue;
    otherwise the GEMM followed by ldmae_swiglu_fwd (same numbers: the epilogue rounds to bf16 before the activation).
    save_h12=False (forward-only; bf16 path): h12, which only the backward pass reads, is not stored and None is returned for it."""
    M, K = a.shape
    N = w12.shape[0]
    if a.dtype == torch.bfloat16 and N % 256 == 0 and K % 64 == 0:
        h12 = torch.empty(M, N, dtype=a.dtype, device=a.device) if save_h12 else None
        hid = torch.empty(M, N // 2, dtype=a.dtype, device=a.device)
        call("ldmae_gemm_nt", BF16, BF16, EPI_SWIGLU | _launch_flag(), ptr(a), a.stride(0), ptr(w12), w12.stride(0), ptr(h12), N, M, N, K, ptr(b12), 0.0,
             None, ptr(hid), None, 0, 0, stream())
        return h12, hid
    h12 = gemm_nt(a, w12, b12)
    return h12, swiglu_fwd(h12)


FUSED_QKV = os.environ.get("LDMAE_FUSED_QKV", "1") != "0"      # module switch for A/B runs (tools/bench_qkv_rope.py)


def gemm_nt_qkv_rope_ok(a, w, B, N, H, hd):
    """Does the fused qkv GEMM (QK-norm / RoPE in the epilogue) cover this call?  bf16, head dim 64, B*N % 256 == 0, N % 128 == 0, rows on 128-B lines.
    LDMAE_FUSED_QKV=0 keeps the GEMM + ldmae_qknorm_rope_fwd pair (A/B runs; bitwise the same q2 / k2)."""
    if a.dtype != torch.bfloat16 or not FUSED_QKV or _HALF_LINES:
        return False
    if a.data_ptr() % 128 or w.data_ptr() % 128 or w.shape[0] != 3 * H * hd:
        return False
    return bool(L.load().ldmae_gemm_nt_qkv_rope_ok(B, N, H, hd, a.shape[1], a.stride(0), w.stride(0)))


def gemm_nt_qkv_rope(a, w, bias, wq, wk, cos, sin, B, N, H, hd, eps=1e-6, store_raw_qk=True):
    """(qkv, q2, k2): the qkv Linear with q_norm / k_norm / RoPE applied in its epilogue (lightningdit.py:68-74 in one kernel).  qkv [B*N, 3*H*hd] as
    gemm_nt writes it (store_raw_qk=False -- forward-only: only the v third is written, the q / k thirds stay uninitialised), q2 / k2 [B, H, N, hd] bitwise
    what qknorm_rope_fwd makes of the stored q / k.  wq = wk = None: RoPE only.  Call gemm_nt_qkv_rope_ok first."""
    M, K = a.shape
    qkv = torch.empty(M, 3 * H * hd, dtype=a.dtype, device=a.device)
    q2 = torch.empty(B, H, N, hd, dtype=a.dtype, device=a.device)
    k2 = torch.empty_like(q2)
    call("ldmae_gemm_nt_qkv_rope", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(qkv), ptr(q2), ptr(k2), ptr(wq), ptr(wk), ptr(cos), ptr(sin),
         B, N, H, hd, K, eps, 1 if store_raw_qk else 0, 1 if (_launch_flag() & EPI_TILE_LAUNCH) else 0, stream())
    return qkv, q2, k2


def gemm_nt_swiglu_bwd(dy, w3t, h12, with_bias=False):
    """dh12 = swiglu_bwd(dy @ w3t^T, h12)   (w3t = [Hs, D] transposed copy of w3).  with_bias: also the column sums of dh12 (the
    bias gradient of w12), formed in the GEMM epilogue as per-128-row partials and summed here."""
    M, K = dy.shape
    Hs = w3t.shape[0]
    if dy.dtype == torch.bfloat16 and Hs % 8 == 0 and K % 64 == 0:
        dh12 = torch.empty_like(h12)
        # every (128-row group, column) of the partial-sum matrix is written by exactly one wave when the tile grid is whole: no 33 MB fill
        whole = M % 128 == 0 and Hs % 64 == 0
        part = (torch.empty if whole else torch.zeros)((M + 127) // 128, 2 * Hs, dtype=torch.float32, device=dy.device) if with_bias else None
        call("ldmae_gemm_nt", BF16, BF16, EPI_SWIGLU_BWD | _launch_flag(), ptr(dy), dy.stride(0), ptr(w3t), w3t.stride(0), ptr(dh12), 2 * Hs, M, Hs, K, None, 0.0,
             ptr(h12), ptr(part), None, 0, 0, stream())
        return (dh12, colsum(part)) if with_bias else dh12
    dh12 = swiglu_bwd(gemm_nt(dy, w3t), h12)
    return (dh12, colsum(dh12)) if with_bias else dh12


def gemm_tn(a, b, out=None, beta=0.0, with_bias=False, ws_slot="tn", dbias_out=None):
    """out[N,K] (f32) = beta*out + a[M,N]^T @ b[M,K]   (weight gradient).  with_bias: also return the column sums of `a`
    (the bias gradient of the same Linear), fused into the same kernel on the bf16 path.  dbias_out (f32 [N], contiguous): the bias gradient
    is accumulated THERE under the same beta as `out` (a .grad slab view: no fresh zero-filled buffer, no add afterwards)."""
    M, N = a.shape
    K = b.shape[1]
    if out is None:
        out = torch.empty(N, K, dtype=torch.float32, device=a.device)
        beta = 0.0
    if dbias_out is not None:
        if not with_bias or dbias_out.dtype != torch.float32 or dbias_out.numel() != N or not dbias_out.is_contiguous():
            raise RuntimeError("gemm_tn: dbias_out must be a contiguous f32 [N] buffer and needs with_bias=True")
        dbias = dbias_out
    else:
        # the C entry applies ONE beta to C and to dbias: a fresh bias-gradient buffer must be zero when the caller accumulates into `out`
        dbias = (torch.zeros if beta != 0.0 else torch.empty)(N, dtype=torch.float32, device=a.device) if with_bias else None
    d = dt(a.dtype)
    nb = max(L.load().ldmae_gemm_tn_workspace_bytes(d, M, N, K), L.load().ldmae_colsum_workspace_bytes(M, N) if with_bias else 0)
    ws = workspace(nb, a.device, ws_slot)
    call("ldmae_gemm_tn", d, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), ptr(dbias), M, N, K, float(beta), ptr(ws), ws.numel() * 4,
         stream())
    return (out, dbias) if with_bias else out


def colsum(x, out=None, beta=0.0):
    M, N = x.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
        beta = 0.0
    ws = workspace(L.load().ldmae_colsum_workspace_bytes(M, N), x.device, "colsum")
    call("ldmae_colsum", dt(x.dtype), ptr(x), x.stride(0), M, N, ptr(out), float(beta), ptr(ws), stream())
    return out


def cast_weight(w, dtype, transposed=True, straight=True):
    """f32 master weight [R,C] -> (copy in `dtype` or None, [C,R] transposed copy or None)."""
    R, C = w.shape
    dst = torch.empty(R, C, dtype=dtype, device=w.device) if straight else None
    dstT = torch.empty(C, R, dtype=dtype, device=w.device) if transposed else None
    call("ldmae_cast_weight", dt(dtype), ptr(w), ptr(dst), ptr(dstT), R, C, stream())
    return dst, dstT


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("ldmae_cast", dt(x.dtype), dt(dtype), ptr(_c(x)), ptr(out), x.numel(), stream())
    return out


_THIN = os.environ.get("LDMAE_THIN_GEMM", "1") != "0"       # 0: the generic f32 MFMA GEMMs for these shapes too (A/B)


def thin_ok(N, K):
    """Shapes the thin (K = 16 / 32) f32 products take: see ldmae_thin_nt / ldmae_thin_tn."""
    return _THIN and K in (16, 32) and N % 4 == 0


def thin_nt(t, w, bias=None, pos=None, rows_per_batch=0, out_dtype=torch.float32):
    """out[M,N] = t[M,K] @ w[N,K]^T + bias (+ pos[row % rows_per_batch]) for K = 16 / 32, f32 inputs; one streaming pass."""
    M, K = t.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=out_dtype, device=t.device)
    call("ldmae_thin_nt", dt(out_dtype), ptr(_c(t)), ptr(_c(w)), ptr(bias), ptr(_c(pos)) if pos is not None else None, ptr(out), M, N, K,
         int(rows_per_batch), stream())
    return out


def thin_tn(g, t, with_bias=True):
    """(dW[N,K] = g[M,N]^T @ t[M,K], column sums of g or None) for K = 16 / 32, f32; one pass over g."""
    M, N = g.shape
    K = t.shape[1]
    dW = torch.empty(N, K, dtype=torch.float32, device=g.device)
    db = torch.empty(N, dtype=torch.float32, device=g.device) if with_bias else None
    ws = workspace(L.load().ldmae_thin_tn_workspace_bytes(M, N, K), g.device)
    call("ldmae_thin_tn", ptr(_c(g)), ptr(_c(t)), ptr(dW), ptr(db), M, N, K, 0.0, ptr(ws), ws.numel() * 4, stream())
    return dW, db


def multi_add_(dsts, srcs):
    """dsts[i] += srcs[i] (contiguous f32 tensors of equal sizes pairwise), one launch."""
    import ctypes
    k = len(dsts)
    srcs = [_c(s_) for s_ in srcs]
    for d_, s_ in zip(dsts, srcs):
        if d_.dtype != torch.float32 or s_.dtype != torch.float32 or d_.numel() != s_.numel() or not d_.is_contiguous():
            raise RuntimeError("multi_add_: contiguous f32 pairs of equal size expected")
    da = (ctypes.c_void_p * k)(*[d_.data_ptr() for d_ in dsts])
    sa = (ctypes.c_void_p * k)(*[s_.data_ptr() for s_ in srcs])
    na = (ctypes.c_long * k)(*[d_.numel() for d_ in dsts])
    call("ldmae_multi_add", k, da, sa, na, stream())


def cast_stack(tensors, dtype):
    """Equally shaped contiguous f32 tensors -> one stacked [len * rows, cols] tensor in `dtype`, one launch."""
    import ctypes
    t0 = tensors[0]
    n_each = t0.numel()
    out = torch.empty((len(tensors) * t0.shape[0],) + tuple(t0.shape[1:]), dtype=dtype, device=t0.device)
    srcs = [_c(t) for t in tensors]
    if any(t.dtype != torch.float32 or t.numel() != n_each for t in srcs):
        raise RuntimeError("cast_stack: f32 tensors of one size expected")
    arr = (ctypes.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
    call("ldmae_cast_stack", dt(dtype), arr, len(srcs), n_each, ptr(out), stream())
    return out


# ----------------------------------------------------------------------------- norms / elementwise
def rmsnorm_modulate_fwd(x, w, shift, scale, rows_per_batch, out_dtype, eps=1e-6):
    """modulate(norm(x), shift, scale) (lightningdit.py:26-30,248-249).  w: the RMSNorm weight; w=None: nn.LayerNorm(elementwise_affine=False)
    -- the blocks built with use_rmsnorm=False (:200-201) -- on the same kernels (the norm of the centred row)."""
    M, D = x.shape
    out = torch.empty(M, D, dtype=out_dtype, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    ld = shift.stride(0) if shift is not None else (scale.stride(0) if scale is not None else 0)
    if w is None:
        call("ldmae_layernorm_modulate_fwd", dt(out_dtype), ptr(x), ptr(shift), ptr(scale), ld, ptr(out), ptr(rstd), M, D, rows_per_batch, eps, stream())
        return out, rstd
    call("ldmae_rmsnorm_modulate_fwd", dt(out_dtype), ptr(x), ptr(w), ptr(shift), ptr(scale), ld, ptr(out), ptr(rstd), M, D,
         rows_per_batch, eps, stream())
    return out, rstd


def rmsnorm_modulate_bwd(dout, x, w, scale, rstd, dx_accum, dshift, dscale, rows_per_batch, accumulate=True):
    """dx_accum += dx (in place; accumulate=False: dx_accum = dx, the buffer may be uninitialised); writes dshift/dscale views ([B,D], any
    row stride); returns dw [D]."""
    M, D = x.shape
    ws = workspace(L.load().ldmae_rmsnorm_modulate_bwd_workspace_bytes(M, D, rows_per_batch), x.device)
    dld = (dshift if dshift is not None else dscale).stride(0) if (dshift is not None or dscale is not None) else 0
    if w is None:                      # LayerNorm without affine parameters (use_rmsnorm=False): no weight gradient
        call("ldmae_layernorm_modulate_bwd", dt(dout.dtype), ptr(dout), ptr(x), ptr(scale), scale.stride(0) if scale is not None else 0, ptr(rstd), ptr(dx_accum),
             1.0 if accumulate else 0.0, ptr(dshift), ptr(dscale), dld, M, D, rows_per_batch, ptr(ws), stream())
        return None
    dw = torch.empty(D, dtype=torch.float32, device=x.device)
    call("ldmae_rmsnorm_modulate_bwd", dt(dout.dtype), ptr(dout), ptr(x), ptr(w), ptr(scale), scale.stride(0) if scale is not None else 0,
         ptr(rstd), ptr(dx_accum), 1.0 if accumulate else 0.0, ptr(dshift), ptr(dscale), (dshift if dshift is not None else dscale).stride(0) if (dshift is not None or dscale is not None) else 0,
         ptr(dw), 0.0, M, D, rows_per_batch, ptr(ws), stream())
    return dw


def rmsnorm_modulate_bwd_gate(dout, x, w, scale, rstd, dx_accum, dshift, dscale, y, gate, dgate, rows_per_batch, act_dtype, accumulate=True):
    """rmsnorm_modulate_bwd followed by gate_bwd(dx_accum, y, gate, dgate, with_bias=True) in one pass over the rows.
    Returns (dw [D], dy [M,D] act dtype, dbias [D])."""
    M, D = x.shape
    dy = torch.empty(M, D, dtype=act_dtype, device=x.device)
    dbias = torch.empty(D, dtype=torch.float32, device=x.device)
    ws = workspace(L.load().ldmae_rmsnorm_modulate_bwd_gate_workspace_bytes(M, D, rows_per_batch), x.device)
    if w is None:                      # LayerNorm without affine parameters (use_rmsnorm=False)
        dld = (dshift if dshift is not None else dscale).stride(0) if (dshift is not None or dscale is not None) else 0
        call("ldmae_layernorm_modulate_bwd_gate", dt(dout.dtype), ptr(dout), ptr(x), ptr(scale), scale.stride(0) if scale is not None else 0, ptr(rstd),
             ptr(dx_accum), 1.0 if accumulate else 0.0, ptr(dshift), ptr(dscale), dld, ptr(y), ptr(gate), gate.stride(0), ptr(dy), ptr(dgate), dgate.stride(0),
             ptr(dbias), M, D, rows_per_batch, ptr(ws), stream())
        return None, dy, dbias
    dw = torch.empty(D, dtype=torch.float32, device=x.device)
    call("ldmae_rmsnorm_modulate_bwd_gate", dt(dout.dtype), ptr(dout), ptr(x), ptr(w), ptr(scale), scale.stride(0) if scale is not None else 0,
         ptr(rstd), ptr(dx_accum), 1.0 if accumulate else 0.0, ptr(dshift), ptr(dscale), (dshift if dshift is not None else dscale).stride(0) if (dshift is not None or dscale is not None) else 0, ptr(dw), 0.0,
         ptr(y), ptr(gate), gate.stride(0), ptr(dy), ptr(dgate), dgate.stride(0), ptr(dbias), M, D, rows_per_batch, ptr(ws), stream())
    return dw, dy, dbias


def qknorm_rope_fwd(qkv, wq, wk, cos, sin, B, N, H, hd, eps=1e-6, copy_v=True):
    """copy_v=False: v gets no head-major copy (returned as None); attention then reads it from the packed qkv (attention_fwd_pv).
    wq = wk = None: RoPE only (the block built with use_qknorm=False: q_norm = k_norm = nn.Identity, lightningdit.py:60-61)."""
    q = torch.empty(B, H, N, hd, dtype=qkv.dtype, device=qkv.device)
    k = torch.empty_like(q)
    v = torch.empty_like(q) if copy_v else None
    call("ldmae_qknorm_rope_fwd", dt(qkv.dtype), ptr(qkv), ptr(wq), ptr(wk), ptr(cos), ptr(sin), ptr(q), ptr(k), ptr(v), B, N, H, hd, eps, stream())
    return q, k, v


def qknorm_rope_bwd(dq, dk, dv, qkv, wq, wk, cos, sin, B, N, H, hd, eps=1e-6, with_bias=False, dqkv=None):
    """Backward of qknorm_rope_fwd: (dqkv [B,N,3,H,hd], dwq, dwk[, dbias [3*H*hd]]).  with_bias: the bias gradient of the qkv Linear
    (column sums of dqkv as stored), formed in the same pass.  dv=None with dqkv given: dv already sits in the v slot of dqkv
    (attention_bwd_pv) and is left there."""
    dqkv = torch.empty_like(qkv) if dqkv is None else dqkv
    dwq = torch.empty(hd, dtype=torch.float32, device=qkv.device) if wq is not None else None      # wq = wk = None: RoPE adjoint only
    dwk = torch.empty_like(dwq) if wq is not None else None
    db = torch.empty(H, 3, hd, dtype=torch.float32, device=qkv.device) if with_bias else None
    ws = workspace(L.load().ldmae_qknorm_rope_bwd_workspace_bytes(B, N, H, hd), qkv.device)
    call("ldmae_qknorm_rope_bwd", dt(qkv.dtype), ptr(dq), ptr(dk), ptr(dv), ptr(qkv), ptr(wq), ptr(wk), ptr(cos), ptr(sin), ptr(dqkv),
         ptr(dwq), ptr(dwk), 0.0, ptr(db), B, N, H, hd, eps, ptr(ws), stream())
    if with_bias:
        return dqkv, dwq, dwk, db.permute(1, 0, 2).reshape(-1)          # (head, q|k|v, d) -> the Linear's (q|k|v, head, d) order
    return dqkv, dwq, dwk


def rope(t, cos, sin, transposed=False):
    """t [..., N, hd] (f32 or bf16, contiguous) -> t*cos + rotate_half(t)*sin with cos/sin [N, hd] f32 (VisionRotaryEmbeddingFast.forward);
    transposed: the adjoint (backward)."""
    N, hd = cos.shape
    if t.shape[-2:] != (N, hd):
        raise RuntimeError(f"rope: the last two dims of t {tuple(t.shape)} must be (N, head_dim) = {(N, hd)}")
    t = _c(t)
    out = torch.empty_like(t)
    call("ldmae_rope", dt(t.dtype), ptr(t), ptr(cos), ptr(sin), ptr(out), t.numel() // hd, N, hd, 1 if transposed else 0, stream())
    return out


_ATTN_HDS = (16, 32, 64, 72, 128)            # head dims the attention kernels are instantiated for (csrc/attention.hip: ATTN_HD_DISPATCH)
_ATTN_HDS_F32 = (16, 32, 64, 72, 80, 96, 128)      # ... and the f32 kernels (ATTN_HD_DISPATCH_F32): their backward's LDS tiles end at head_dim 96


def _attn_hds(dtype):
    return _ATTN_HDS_F32 if dtype == torch.float32 else _ATTN_HDS


def _attn_pad(hd: int, dtype=None) -> int:
    for h in _attn_hds(dtype):
        if h >= hd:
            return h
    raise RuntimeError(f"ldmae_amd attention: head_dim {hd} is above the largest instantiated kernel (128)")


def attention_fwd(q, k, v, scale):
    """softmax(q k^T * scale) v.  q, k, v: [B,H,N,hd]; returns (o [B,N,H*hd], lse [B,H,N] f32).
    bf16 head dims that are not a multiple of 32 (LightningDiT-XL: 72, VMAE: 16) are zero-padded to the next multiple of 32
    INSIDE the kernels (LDS images and register fragments); HBM tensors keep the true head dim."""
    B, H, N, hd = q.shape
    if hd not in _attn_hds(q.dtype):
        # Head dims outside the instantiated set (16, 32, 64, 72, 128) -- e.g. 24 (mae_for_ldmae_f8d16_prev_large), 80 (mae_vit_huge), 8: zero
        # columns add nothing to q . k and produce zero output columns, so the next larger kernel on zero-padded copies is exact (`scale` is
        # the caller's, from the true head dim).  Costs the copies; the shipped archs never come here.
        P = _attn_pad(hd, q.dtype)
        pad = lambda t: torch.nn.functional.pad(t, (0, P - hd))      # noqa: E731
        o, lse = attention_fwd(pad(q), pad(k), pad(v), scale)
        return o.view(B, N, H, P)[..., :hd].reshape(B, N, H * hd), lse
    o = torch.empty(B, N, H * hd, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=q.device)
    call("ldmae_attention_fwd", dt(q.dtype), ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), B, H, N, hd, float(scale), stream())
    return o, lse


def attention_bwd(q, k, v, o, do, lse, scale):
    B, H, N, hd = q.shape
    if hd not in _attn_hds(q.dtype):                          # head dims the kernels are not instantiated for: zero-padded (see attention_fwd)
        P = _attn_pad(hd, q.dtype)
        pad, padt = (lambda t: torch.nn.functional.pad(t, (0, P - hd))), (lambda t: torch.nn.functional.pad(t.reshape(B, N, H, hd), (0, P - hd)).reshape(B, N, H * P))
        dq, dk, dv = attention_bwd(pad(q), pad(k), pad(v), padt(o), padt(_c(do)), lse, scale)
        return dq[..., :hd].contiguous(), dk[..., :hd].contiguous(), dv[..., :hd].contiguous()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    delta = torch.empty(2, B, H, (N + 63) // 64 * 64, dtype=torch.float32, device=q.device)      # rows padded to whole 64-row tiles
    call("ldmae_attention_bwd", dt(q.dtype), ptr(q), ptr(k), ptr(v), ptr(o), ptr(_c(do)), ptr(lse), ptr(dq), ptr(dk), ptr(dv), ptr(delta),
         B, H, N, hd, float(scale), stream())
    return dq, dk, dv


def qk_score_bound(wq, wk, hd, scale):
    """One float on the device: hd * max|wq| * max|wk| * scale * log2(e) * 1.02, an upper bound of every attention score (in the kernels'
    log2 units) of heads that went through QK-RMSNorm with these weights and RoPE -- for attention_fwd_pv(bound=...)."""
    out = torch.empty(1, dtype=torch.float32, device=wq.device)
    call("ldmae_qk_score_bound", ptr(wq), ptr(wk), hd, float(scale), ptr(out), stream())
    return out


def attention_fwd_pv(q, k, qkv, scale, bound=None):
    """q, k head-major [B,H,N,hd]; v read from the packed qkv [B*N, 3*H*hd] (bf16).  bound: a proven upper bound of the scores
    (qk_score_bound): the softmax then runs with a static shift instead of a running maximum (same result, fewer vector instructions)."""
    B, H, N, hd = q.shape
    o = torch.empty(B, N, H * hd, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=q.device)
    if bound is not None:
        call("ldmae_attention_fwd_pv_bounded", dt(q.dtype), ptr(q), ptr(k), ptr(qkv), ptr(o), ptr(lse), ptr(bound), B, H, N, hd, float(scale), stream())
    else:
        call("ldmae_attention_fwd_pv", dt(q.dtype), ptr(q), ptr(k), ptr(qkv), ptr(o), ptr(lse), B, H, N, hd, float(scale), stream())
    return o, lse


def attention_bwd_pv(q, k, qkv, o, do, lse, scale):
    """-> (dq, dk head-major, dqkv with ONLY its v slot written: dv in the packed layout)."""
    B, H, N, hd = q.shape
    dq, dk = torch.empty_like(q), torch.empty_like(q)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(2, B, H, (N + 63) // 64 * 64, dtype=torch.float32, device=q.device)
    call("ldmae_attention_bwd_pv", dt(q.dtype), ptr(q), ptr(k), ptr(qkv), ptr(o), ptr(_c(do)), ptr(lse), ptr(dq), ptr(dk), ptr(dqkv), ptr(delta),
         B, H, N, hd, float(scale), stream())
    return dq, dk, dqkv


def attention_bwd_pv_qknorm(q, k, qkv, o, do, lse, scale, wq, wk, cos, sin, eps=1e-6):
    """attention_bwd_pv + qknorm_rope_bwd(with_bias=True) in one (bf16, head_dim 64 / 128): -> (dqkv [B,N,3,H,hd] complete, dwq, dwk,
    dbias [3*H*hd])."""
    B, H, N, hd = q.shape
    dqkv = torch.empty_like(qkv)
    if wq is not None:
        dw2 = torch.empty(2 * hd, dtype=torch.float32, device=q.device)      # dwq | dwk adjacent: the library reduces straight into them
        dwq, dwk = dw2[:hd], dw2[hd:]
    else:
        dwq = dwk = None                                                      # RoPE adjoint only (use_qknorm=False)
    db = torch.empty(3 * H * hd, dtype=torch.float32, device=q.device)
    ws = workspace(L.load().ldmae_attention_bwd_pv_qknorm_workspace_bytes(B, H, N, hd), q.device)
    call("ldmae_attention_bwd_pv_qknorm", dt(q.dtype), ptr(q), ptr(k), ptr(qkv), ptr(o), ptr(_c(do)), ptr(lse), ptr(wq), ptr(wk), ptr(cos),
         ptr(sin), float(eps), ptr(dqkv), ptr(dwq), ptr(dwk), ptr(db), ptr(ws), B, H, N, hd, float(scale), stream())
    return dqkv, dwq, dwk, db


# attention_fwd_qkv: B*H*N*N from which the extra pass over the k slots pays (256 images x 24 heads x 1024^2: -7 %); LDMAE_BOUNDED_ATTN_MIN overrides
BOUNDED_ATTENTION_MIN_SCORES = int(os.environ.get("LDMAE_BOUNDED_ATTN_MIN", 1 << 31))


def attention_fwd_qkv(qkv, B, N, H, hd, scale):
    """Attention straight on the packed token-major qkv [B*N, 3*H*hd] (bf16; f32 at head_dim 16): no head-major relayout.  -> (o [B,N,H*hd], lse).
    Long sequences of small heads (the 1024-token VMAE decoder: the kernel is bound by vector issue) first take one pass over the k slots
    for max |k|^2 per (image, head): with each query's own norm it bounds the scores, and the softmax runs with that static shift instead of
    a running maximum (same result; ldmae_k_norm_max + ldmae_attention_fwd_qkv_bounded)."""
    if hd not in _attn_hds(qkv.dtype):                        # (see attention_fwd: head-major zero-padded copies)
        return attention_fwd(*heads_split(qkv, B, N, H, hd), scale)
    o = torch.empty(B, N, H * hd, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    if qkv.dtype == torch.bfloat16 and hd <= 32 and N >= 512 and B * H * N * N >= BOUNDED_ATTENTION_MIN_SCORES:
        kmax = torch.empty(B * H, 2, dtype=torch.float32, device=qkv.device)
        call("ldmae_k_norm_max", ptr(qkv), ptr(kmax), B, N, H, hd, stream())
        call("ldmae_attention_fwd_qkv_bounded", dt(qkv.dtype), ptr(qkv), ptr(o), ptr(lse), ptr(kmax), B, H, N, hd, float(scale), stream())
    else:
        call("ldmae_attention_fwd_qkv", dt(qkv.dtype), ptr(qkv), ptr(o), ptr(lse), B, H, N, hd, float(scale), stream())
    return o, lse


def attention_bwd_qkv(qkv, o, do, lse, B, N, H, hd, scale):
    """-> dqkv [B*N, 3*H*hd] (dq / dk / dv written in the packed layout)."""
    if hd not in _attn_hds(qkv.dtype):                        # (see attention_fwd: head-major zero-padded copies)
        q, k, v = heads_split(qkv, B, N, H, hd)
        return heads_merge(*attention_bwd(q, k, v, o, do, lse, scale), B, N, H, hd)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(2, B, H, (N + 63) // 64 * 64, dtype=torch.float32, device=qkv.device)
    call("ldmae_attention_bwd_qkv", dt(qkv.dtype), ptr(qkv), ptr(o), ptr(_c(do)), ptr(lse), ptr(dqkv), ptr(delta), B, H, N, hd, float(scale), stream())
    return dqkv


def swiglu_fwd(h12):
    M, H2 = h12.shape
    hid = torch.empty(M, H2 // 2, dtype=h12.dtype, device=h12.device)
    call("ldmae_swiglu_fwd", dt(h12.dtype), ptr(h12), ptr(hid), M, H2 // 2, stream())
    return hid


def swiglu_bwd(dhid, h12):
    M, H2 = h12.shape
    dh12 = torch.empty_like(h12)
    call("ldmae_swiglu_bwd", dt(h12.dtype), ptr(dhid), ptr(h12), ptr(dh12), M, H2 // 2, stream())
    return dh12


def gate_bwd(dxout, y, gate, dgate, rows_per_batch, act_dtype, with_bias=False):
    """dy = dxout * gate[b] (act dtype);  dgate view [B,D] <- sum_n dxout*y (skipped when dgate is None).
    with_bias: also return the column sums of dy (the bias gradient of the Linear that produced the branch)."""
    M, D = dxout.shape
    dy = torch.empty(M, D, dtype=act_dtype, device=dxout.device)
    dbias = torch.empty(D, dtype=torch.float32, device=dxout.device) if with_bias else None
    need_ws = dgate is not None or with_bias
    ws = workspace(L.load().ldmae_gate_bwd_workspace_bytes(M, D, rows_per_batch), dxout.device) if need_ws else None
    call("ldmae_gate_bwd", dt(act_dtype), ptr(dxout), ptr(y), ptr(gate), gate.stride(0) if gate is not None else 0, ptr(dy), ptr(dgate),
         dgate.stride(0) if dgate is not None else 0, ptr(dbias), M, D, rows_per_batch, ptr(ws), stream())
    return (dy, dbias) if with_bias else dy


def timestep_embedding(t, dim=256, max_period=10000.0):
    out = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
    call("ldmae_timestep_embedding", ptr(_c(t.float())), ptr(out), t.shape[0], dim, float(max_period), stream())
    return out


def silu_fwd(x, out_dtype=torch.float32):
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    call("ldmae_silu_fwd", dt(out_dtype), ptr(x), ptr(out), x.numel(), stream())
    return out


def silu_bwd(dy, x):
    dx = torch.empty_like(x)
    call("ldmae_silu_bwd", ptr(_c(dy)), ptr(x), ptr(dx), x.numel(), stream())
    return dx


def label_embed_fwd(table, y, drop, num_classes):
    B, D = y.shape[0], table.shape[1]
    out = torch.empty(B, D, dtype=torch.float32, device=table.device)
    call("ldmae_label_embed_fwd", ptr(table), ptr(y), ptr(drop), ptr(out), B, D, num_classes, stream())
    return out


def label_embed_bwd(dout, y, drop, num_classes, rows):
    dtable = torch.zeros(rows, dout.shape[1], dtype=torch.float32, device=dout.device)
    call("ldmae_label_embed_bwd", ptr(_c(dout)), ptr(y), ptr(drop), ptr(dtable), dout.shape[0], dout.shape[1], num_classes, rows, stream())
    return dtable


# Bumped by every op that rewrites parameter storage through the C ABI (torch's own version counters do not see those writes): part of the
# key of the forward-only weight-copy cache below.
WEIGHT_EPOCH = 0
_WCACHE: dict = {}


def invalidate_weight_cache() -> None:
    """Call after writing parameter STORAGE behind torch's back -- through the flat slab the parameters are views of (dist.broadcast(
    flat.params), flat.params.copy_(ema), a checkpoint restored into the slab): those writes bump neither the parameters' version
    counters nor WEIGHT_EPOCH, and a cached bf16 copy would silently go stale."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def cached_weight_copy(w, dtype):
    """`dtype` copy of the f32 master weight `w` for FORWARD-ONLY use (sampling / encoding under no_grad runs the same weights hundreds of
    times: 112 cast launches per XL/1 forward).  Valid while the storage pointer, torch's version counter of `w` and WEIGHT_EPOCH are
    unchanged; training never comes here (its weights change every step)."""
    key = (id(w), dtype)
    stamp = (w.data_ptr(), w._version, WEIGHT_EPOCH)
    hit = _WCACHE.get(key)
    if hit is not None and hit[0]() is w and hit[1] == stamp:      # the weak reference guards against a recycled id()
        return hit[2]
    if len(_WCACHE) > 4096:
        _WCACHE.clear()
    c = cast_weight(w, dtype, transposed=False, straight=True)[0]
    _WCACHE[key] = (weakref.ref(w), stamp, c)
    return c


def cached_stack_copy(tensors, dtype):
    """cast_stack for FORWARD-ONLY use (sampling re-runs the same weights hundreds of times): valid while every source's storage pointer and
    version counter and WEIGHT_EPOCH are unchanged.  Keyed by the first tensor (weak reference, like cached_weight_copy)."""
    w = tensors[0]
    key = (id(w), dtype, "stack", len(tensors))
    stamp = (tuple((t.data_ptr(), t._version) for t in tensors), WEIGHT_EPOCH)
    hit = _WCACHE.get(key)
    if hit is not None and hit[0]() is w and hit[1] == stamp:
        return hit[2]
    if len(_WCACHE) > 4096:
        _WCACHE.clear()
    c = cast_stack(tensors, dtype)
    _WCACHE[key] = (weakref.ref(w), stamp, c)
    return c


def adamw_ema(p, g, m, v, ema, step, lr, beta1, beta2, eps, weight_decay, ema_decay, grad_scale=1.0):
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1
    call("ldmae_adamw_ema", ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), p.numel(), int(step), float(lr), float(beta1), float(beta2),
         float(eps), float(weight_decay), float(ema_decay), float(grad_scale), stream())


def ema_only(ema, p, ema_decay):
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1
    call("ldmae_ema_only", ptr(ema), ptr(p), p.numel(), float(ema_decay), stream())


# ----------------------------------------------------------------------------- VMAE
def random_masking(noise, keep):
    N, Lq = noise.shape
    ids_restore = torch.empty(N, Lq, dtype=torch.int64, device=noise.device)
    mask = torch.empty(N, Lq, dtype=torch.float32, device=noise.device)
    ids_keep = torch.empty(N, keep, dtype=torch.int64, device=noise.device)
    call("ldmae_random_masking", ptr(_c(noise)), ptr(ids_restore), ptr(mask), ptr(ids_keep), N, Lq, keep, stream())
    return ids_keep, mask, ids_restore


def patch_embed_kept(img, ids_keep, pos, w2d, bias, patch, dtype):
    """Patch embedding of the kept tokens only: img [N,C,S,S] f32, ids_keep [N,keep] i64, pos [L,D] f32, w2d [D, C*p*p] f32 master weight
    -> [N, keep, D] f32 = conv(img)[kept] + bias + pos[kept].  Same bits as embedding all patches (gemm_nt_pos) and gathering."""
    N, C, S, _ = img.shape
    keep, D = ids_keep.shape[1], w2d.shape[0]
    tok = torch.empty(N * keep, C * patch * patch, dtype=dtype, device=img.device)
    posg = torch.empty(N * keep, D, dtype=torch.float32, device=img.device)
    call("ldmae_patch_gather", dt(dtype), ptr(_c(img.float())), ptr(ids_keep), ptr(pos), ptr(tok), ptr(posg), N, keep, C, S, patch, D, stream())
    wb = cast(_c(w2d), dtype)          # 192 x 192: one tiny launch (w2d is a fresh view per call, so the id-keyed weight cache does not apply)
    out, _ = gemm_nt_gate_res(tok, wb, bias, posg, None, keep, save_y=False, xout=posg, y_dtype=torch.float32)   # unrounded, as gemm_nt_pos
    return out.view(N, keep, D)


def latent_prologue(moments, noise=None, lat_mean=None, lat_std=None, multiplier=1.0, sample=True):
    """Device-side counterpart of ImgLatentDataset.__getitem__ after the shard read: moments [B, 2C, H, W] f32 (sample) or latents
    [B, C, H, W] -> normalised model input [B, C, H, W] f32.  noise [B, C, H, W] (drawn by the caller); lat_mean / lat_std [C]-sized."""
    B, C2, H, W = moments.shape
    C = C2 // 2 if sample else C2
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=moments.device)
    mom = _c(moments.float())
    nz = _c(noise.float()) if noise is not None else None
    mu, sd = (_c(t.float().reshape(-1)) if t is not None else None for t in (lat_mean, lat_std))
    call("ldmae_latent_prologue", ptr(mom), ptr(nz) if nz is not None else None, ptr(mu) if mu is not None else None,
         ptr(sd) if sd is not None else None, float(multiplier), ptr(out), B, C, H * W, 1 if sample else 0, stream())
    return out


def gather_rows(x, ids):
    N, Lq, D = x.shape
    keep = ids.shape[1]
    out = torch.empty(N, keep, D, dtype=torch.float32, device=x.device)
    call("ldmae_gather_rows", ptr(x), ptr(ids), ptr(out), N, Lq, keep, D, stream())
    return out


def scatter_rows(dout, ids, Lq):
    N, keep, D = dout.shape
    dx = torch.zeros(N, Lq, D, dtype=torch.float32, device=dout.device)
    call("ldmae_scatter_rows", ptr(_c(dout)), ptr(ids), ptr(dx), N, Lq, keep, D, stream())
    return dx


def restore_tokens(x, mask_token, pos, ids_restore):
    """Decoder input of the pre-training step (models_mae.py:536-541) in one pass: x [B, keep, D] f32, mask_token [D], pos [L, D], ids_restore
    [B, L] i64 -> [B, L, D] = (kept row or mask token) + pos."""
    B, keep, D = x.shape
    Lq = ids_restore.shape[1]
    out = torch.empty(B, Lq, D, dtype=torch.float32, device=x.device)
    call("ldmae_restore_tokens", ptr(_c(x)), ptr(_c(mask_token)), ptr(_c(pos)), ptr(_c(ids_restore)), ptr(out), B, Lq, keep, D, stream())
    return out


def restore_tokens_bwd(dout, ids_restore, keep, need_mask_grad=True):
    """-> (dx [B, keep, D], dmask_token [D] or None)."""
    B, Lq, D = dout.shape
    dx = torch.empty(B, keep, D, dtype=torch.float32, device=dout.device)
    dm = torch.empty(D, dtype=torch.float32, device=dout.device) if need_mask_grad else None
    ws = workspace(L.load().ldmae_restore_tokens_bwd_workspace_bytes(B, Lq, D), dout.device) if need_mask_grad else None
    call("ldmae_restore_tokens_bwd", ptr(_c(dout)), ptr(_c(ids_restore)), ptr(dx), ptr(dm), B, Lq, keep, D, ptr(ws), stream())
    return dx, dm


def vmae_encoder_fwd(x, blob, nblocks, dim, heads, hidden, eps=1e-6):
    """The whole VMAE encoder stack (blocks + closing LayerNorm) in one launch: x [B, tokens, dim] f32 -> same shape (inference, bf16 MFMA,
    f32 residual stream in registers).  `blob`: weights packed by tokenizer/fused_encoder.py."""
    B, T, D = x.shape
    x = _c(x.float())
    need = L.load().ldmae_vmae_encoder_blob_bytes(nblocks)
    if blob.numel() * blob.element_size() != need:
        raise RuntimeError(f"vmae_encoder_fwd: weight blob has {blob.numel() * blob.element_size()} bytes, the kernel expects {need}")
    out = torch.empty_like(x)
    call("ldmae_vmae_encoder_fwd", ptr(x), ptr(out), ptr(blob), B, T, D, heads, hidden, nblocks, float(eps), stream())
    return out


def vmae_encoder_fwd_tiled(x, blob, nblocks, dim, heads, hidden, eps=1e-6, f16=False):
    """The same stack on sequences of several whole 256-token tiles per image (the docking encoder on all 1024 patches): three launches per
    block -- q|k|v of a tile, flash attention on the packed qkv, proj + MLP of a tile -- from the same weight blob.  f16: the TF32-class form
    (the blob packed in fp16)."""
    B, T, D = x.shape
    x = _c(x.float())
    need = L.load().ldmae_vmae_encoder_blob_bytes(nblocks)
    if blob.numel() * blob.element_size() != need:
        raise RuntimeError(f"vmae_encoder_fwd_tiled: weight blob has {blob.numel() * blob.element_size()} bytes, the kernels expect {need}")
    out = torch.empty_like(x)
    ws = workspace(L.load().ldmae_vmae_encoder_fwd_tiled_workspace_bytes(B, T), x.device)
    call("ldmae_vmae_encoder_fwd_tiled_f16" if f16 else "ldmae_vmae_encoder_fwd_tiled", ptr(x), ptr(out), ptr(blob), ptr(ws), B, T, D, heads, hidden,
         nblocks, float(eps), stream())
    return out


def heads_split(qkv, B, N, H, hd):
    """[B,N,3,H,hd] -> q,k,v [B,H,N,hd] (no norm / rope)."""
    if hd % 8 != 0:
        # head dims off the kernels' 8-element grid (12: the pre-training tree's mae_for_ldmae_f8d16_small, VMAE/models_mae.py:1036-1041): a pure
        # re-layout, done by torch's copy; the attention wrappers then zero-pad the heads to the next instantiated head dim
        t = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        return t[0].contiguous(), t[1].contiguous(), t[2].contiguous()
    q = torch.empty(B, H, N, hd, dtype=qkv.dtype, device=qkv.device)
    k, v = torch.empty_like(q), torch.empty_like(q)
    call("ldmae_qknorm_rope_fwd", dt(qkv.dtype), ptr(qkv), None, None, None, None, ptr(q), ptr(k), ptr(v), B, N, H, hd, 0.0, stream())
    return q, k, v


def heads_merge(dq, dk, dv, B, N, H, hd):
    """inverse of heads_split for the gradients: -> [B*N, 3*H*hd]."""
    if hd % 8 != 0:                                   # (see heads_split)
        return torch.stack((dq, dk, dv), 0).permute(1, 3, 0, 2, 4).reshape(B * N, 3 * H * hd)
    dqkv = torch.empty(B * N, 3 * H * hd, dtype=dq.dtype, device=dq.device)
    call("ldmae_qknorm_rope_bwd", dt(dq.dtype), ptr(dq), ptr(dk), ptr(dv), None, None, None, None, None, ptr(dqkv), None, None, 0.0,
         None, B, N, H, hd, 0.0, None, stream())
    return dqkv


def conv3x3(x, w, b):
    B, C, Hh, Ww = x.shape
    out = torch.empty_like(x)
    call("ldmae_conv3x3", ptr(_c(x)), ptr(_c(w)), ptr(b), ptr(out), B, C, Hh, Ww, stream())
    return out


def conv3x3_bwd(dout, x, w, need_dx=True):
    B, C, Hh, Ww = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty_like(w)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(L.load().ldmae_conv3x3_bwd_workspace_bytes(C), x.device, "conv")
    call("ldmae_conv3x3_bwd", ptr(_c(dout)), ptr(_c(x)), ptr(_c(w)), ptr(dx), ptr(dw), ptr(db), B, C, Hh, Ww, ptr(ws), stream())
    return dx, dw, db


def mae_loss_fwd(pred_img, imgs, mask, p):
    """-> [2] f32 on the device: (sum over masked patches' pixels of (pred - img)^2, the same over visible patches)."""
    B, C, Hh, Ww = imgs.shape
    G = L.load().ldmae_mae_loss_groups(imgs.numel())
    part = torch.empty(G, 2, dtype=torch.float32, device=imgs.device)
    call("ldmae_mae_loss_fwd", ptr(_c(pred_img)), ptr(_c(imgs)), ptr(_c(mask)), ptr(part), B, C, Hh, Ww, int(p), stream())
    return part.sum(0)


def mae_loss_bwd(pred_img, imgs, mask, coef, p):
    B, C, Hh, Ww = imgs.shape
    d = torch.empty_like(pred_img)
    call("ldmae_mae_loss_bwd", ptr(_c(pred_img)), ptr(_c(imgs)), ptr(_c(mask)), ptr(_c(coef)), ptr(d), B, C, Hh, Ww, int(p), stream())
    return d


def layernorm_fwd(x, w, b, out_dtype, eps=1e-6):
    M, D = x.shape
    out = torch.empty(M, D, dtype=out_dtype, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    call("ldmae_layernorm_fwd", dt(out_dtype), ptr(x), ptr(w), ptr(b), ptr(out), ptr(mean), ptr(rstd), M, D, eps, stream())
    return out, mean, rstd


def layernorm_bwd(dout, x, w, mean, rstd, dx_accum, cast=False):
    """dx_accum += dLN/dx; -> (dw, db), with cast=True (16-bit dout) also the updated dx_accum rounded to dout's type."""
    M, D = x.shape
    dw = torch.empty(D, dtype=torch.float32, device=x.device)
    db = torch.empty_like(dw)
    ws = workspace(L.load().ldmae_layernorm_bwd_workspace_bytes(M, D), x.device)
    dxc = torch.empty(M, D, dtype=dout.dtype, device=x.device) if cast else None
    call("ldmae_layernorm_bwd_cast", dt(dout.dtype), ptr(dout), ptr(x), ptr(w), ptr(mean), ptr(rstd), ptr(dx_accum), ptr(dxc), ptr(dw), ptr(db), 0.0,
         M, D, ptr(ws), stream())
    return (dw, db, dxc) if cast else (dw, db)


def gelu_bwd(dout, pre):
    dx = torch.empty_like(pre)
    call("ldmae_gelu_bwd", dt(pre.dtype), ptr(dout), ptr(pre), ptr(dx), pre.numel(), stream())
    return dx


def gelu_tanh_fwd(x):
    """nn.GELU(approximate="tanh") (the timm Mlp of a use_swiglu=False LightningDiT block, lightningdit.py:208,219-224)."""
    out = torch.empty_like(x)
    call("ldmae_gelu_tanh_fwd", dt(x.dtype), ptr(_c(x)), ptr(out), x.numel(), stream())
    return out


def gelu_tanh_bwd(dout, pre):
    dx = torch.empty_like(pre)
    call("ldmae_gelu_tanh_bwd", dt(pre.dtype), ptr(_c(dout)), ptr(pre), ptr(dx), pre.numel(), stream())
    return dx
