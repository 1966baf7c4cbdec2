"""LightningDiT on hand-written gfx950 kernels -- drop-in for the reference module
``models/lightningdit.py`` (same class names, constructor kwargs, attributes,
state-dict keys; reference lines cited per class, relative to
/root/reference/LDMAE/models/lightningdit.py).

Each ``LightningDiTBlock`` runs as ONE ``torch.autograd.Function`` whose forward
and backward are explicit sequences of C-ABI kernel launches
(``ldmae_amd.ops``); there is no PyTorch math on the path and no CPU fallback.

Precision: inside ``torch.autocast(dtype=bfloat16)`` (what ``accelerate
--mixed_precision bf16`` sets up, run_train.sh:9,20) the activations are bf16
with f32 accumulation; otherwise everything is f32 (the parity path).  The
residual stream, modulation vectors, norm statistics and all parameter
gradients are f32 in both modes -- the explicit form of the reference's
autocast behaviour (SURVEY.md §7 "mixed-precision semantics").
"""
from __future__ import annotations

import math

import numpy as np
import os

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from ldmae_amd import ops
from ldmae_amd.tables import sincos_2d
from .pos_embed import VisionRotaryEmbeddingFast
from .rmsnorm import RMSNorm
from .swiglu_ffn import SwiGLUFFN


def _act_dtype(override=None, allow_f16=False):
    """Activation type of a call: the module's `precision` override, else torch's autocast state.  allow_f16: fp16 autocast is accepted (the VMAE
    blocks have an fp16 kernel family -- VMAE/engine_pretrain.py:51-57 trains under it; the LightningDiT kernels are bf16 / f32 only)."""
    if override is not None:
        return override
    if torch.is_autocast_enabled("cuda"):
        d = torch.get_autocast_dtype("cuda")
        if d == torch.float16 and allow_f16:
            return torch.float16
        if d != torch.bfloat16:
            raise RuntimeError(f"ldmae_amd: autocast dtype {d} unsupported here (bfloat16 or no autocast; float16 on the VMAE tokenizer only)")
        return torch.bfloat16
    return torch.float32


def _wcopies(w, dtype, transposed=True):
    """(W in act dtype, W^T [in,out] in act dtype) from the f32 master weight.  transposed=False (forward-only calls): no W^T copy, which
    only the backward pass reads."""
    if dtype == torch.float32:
        return w, (ops.cast_weight(w, dtype, transposed=True, straight=False)[1] if transposed else None)
    if not transposed:
        return ops.cached_weight_copy(w, dtype), None
    return ops.cast_weight(w, dtype, transposed=True, straight=True)


# ----------------------------------------------------------------------------- autograd Functions
class _SiluFn(torch.autograd.Function):
    """SiLU of the conditioning vector (first op of every adaLN_modulation, :233-236, :262-265)."""

    @staticmethod
    def forward(ctx, c):
        c = c.contiguous()
        ctx.save_for_backward(c)
        return ops.silu_fwd(c)

    @staticmethod
    def backward(ctx, g):
        (c,) = ctx.saved_tensors
        return ops.silu_bwd(g.contiguous(), c)


class _LinearF32Fn(torch.autograd.Function):
    """y = act(x) @ W^T + b in f32 for the tiny conditioning GEMMs (t-embedder MLP, :100-104)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        return ops.gemm_nt(x, w, b, out_dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        dx = ops.gemm_nt(g, ops.cast_weight(w, torch.float32, True, False)[1]) if ctx.needs_input_grad[0] else None
        return dx, ops.gemm_tn(g, x), ops.colsum(g)


class _TimestepFreqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, dim):
        return ops.timestep_embedding(t, dim)

    @staticmethod
    def backward(ctx, g):
        return None, None


class _LabelEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, y, drop, num_classes):
        ctx.save_for_backward(y, drop)
        ctx.num_classes, ctx.rows = num_classes, table.shape[0]
        return ops.label_embed_fwd(table, y, drop, num_classes)

    @staticmethod
    def backward(ctx, g):
        y, drop = ctx.saved_tensors
        return ops.label_embed_bwd(g, y, drop, ctx.num_classes, ctx.rows), None, None, None


class _PatchEmbedFn(torch.autograd.Function):
    """tokens @ W^T + b + pos_embed (timm PatchEmbed as used at :309,402 then + pos_embed)."""

    @staticmethod
    def forward(ctx, tok, w2d, b, pos, T, dtype=torch.float32):
        tok = tok.contiguous()
        # under bf16 autocast the conv runs in bf16 (f32 accumulate, f32 output) like the reference's autocast conv2d; the bf16
        # MFMA GEMM needs K = C*p*p to be a multiple of 64 (VMAE 8x8x3 = 192; the DiT's K = 16 stays on the f32 kernel)
        lowp = dtype == torch.bfloat16 and tok.shape[1] % 64 == 0 and tok.shape[0] % 64 == 0
        ctx.lowp = lowp
        if lowp:
            tok = ops.cast(tok, torch.bfloat16)
            ctx.save_for_backward(tok, w2d)
            return ops.gemm_nt_pos(tok, ops.cast(w2d.contiguous(), torch.bfloat16), b, pos, T)
        ctx.save_for_backward(tok, w2d)
        if tok.dtype == torch.float32 and ops.thin_ok(w2d.shape[0], tok.shape[1]):      # K = C*p*p = 16 / 32: one streaming pass (ldmae_thin_nt)
            return ops.thin_nt(tok, w2d.float(), b, pos, T)
        return ops.gemm_nt_pos(tok, w2d, b, pos, T)

    @staticmethod
    def backward(ctx, g):
        tok, w2d = ctx.saved_tensors
        g = g.contiguous()
        dtok = ops.gemm_nt(g, ops.cast_weight(w2d, torch.float32, True, False)[1]) if ctx.needs_input_grad[0] else None
        if not ctx.lowp and g.dtype == torch.float32 and tok.dtype == torch.float32 and ops.thin_ok(g.shape[1], tok.shape[1]):
            dw, db = ops.thin_tn(g, tok)            # weight and bias gradient in ONE pass over the 805-MB gradient
            return dtok, dw, db, None, None, None
        dw = ops.gemm_tn(ops.cast(g, torch.bfloat16), tok) if ctx.lowp else ops.gemm_tn(g, tok)
        return dtok, dw, ops.colsum(g), None, None, None


def _dmod_times_w(dmod, adaw):
    """dSiLU(c) = dmod [B,6D] @ adaW [6D,D] (f32).  The contraction is 6D long and B is small: as an NT GEMM it is 48 workgroups of 288
    K-steps; as a row-split TN GEMM (A = dmod^T) it spreads over the chip (105 -> 36 us at B = 256).  TN needs B % 4 == 0."""
    if dmod.shape[0] % 4 == 0:
        return ops.gemm_tn(dmod.t().contiguous(), adaw)
    return ops.gemm_nt(dmod, ops.cast_weight(adaw, torch.float32, True, False)[1])


def _dw_into_grad(sg, dy, x, param, direct):
    """Weight gradient dy^T x.  `direct` (the training driver opted in: LightningDiT.direct_param_grads) and the parameter already has a
    contiguous f32 `.grad` (a view of the optimizer's gradient slab): the TN GEMM's deterministic slab reduce ADDS into it (beta = 1, what
    AccumulateGrad would do with a separate pass over the 2.4-14 MB tensor) and autograd is handed None; the data-parallel reducer, whose
    post-accumulate hook then does not fire, is told through the callback it left on the parameter.  Otherwise: a new tensor for autograd."""
    g = param.grad if direct else None
    if g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == (dy.shape[1], x.shape[1]):
        sg.tn(dy, x, out=g, beta=1.0)
        ready = getattr(param, "_ldmae_grad_ready", None)
        return None, ready
    return sg.tn(dy, x), None


def _pad_swiglu(w12, b12, w3):
    """SwiGLU hidden sizes off the kernels' grid -- int(2/3 * 4 * hidden) is 2730 for LightningDiT-L (hidden 1024) and 4778 for 1p6B (1792), :213,217 -- are
    ZERO-PADDED to the next multiple of 128: padded units compute silu(0) * 0 = 0 and meet zero columns of w3, so outputs and every gradient of the real
    units are exactly those of the unpadded block.  -> (w12 [2 Hp, D] as [x1 rows, 0 | x2 rows, 0], b12 [2 Hp], w3 [D, Hp], Hs, Hp); identity when Hs % 128 == 0."""
    Hs = w3.shape[1]
    Hp = (Hs + 127) // 128 * 128
    if Hp == Hs:
        return w12, b12, w3, Hs, Hp
    D = w3.shape[0]
    z = w12.new_zeros(Hp - Hs, D)
    w12p = torch.cat([w12[:Hs], z, w12[Hs:], z], 0)
    zb = b12.new_zeros(Hp - Hs)
    b12p = torch.cat([b12[:Hs], zb, b12[Hs:], zb], 0)
    w3p = torch.cat([w3, w3.new_zeros(D, Hp - Hs)], 1)
    return w12p, b12p, w3p, Hs, Hp


def _unpad_swiglu_grads(dW12, db12, dW3, Hs, Hp):
    if Hp == Hs:
        return dW12, db12, dW3
    cut = lambda t_: None if t_ is None else torch.cat([t_[:Hs], t_[Hp:Hp + Hs]], 0)      # noqa: E731
    return cut(dW12), cut(db12), (None if dW3 is None else dW3[:, :Hs].contiguous())


_FUSED_QKN_BWD = os.environ.get("LDMAE_FUSED_QKN_BWD", "1") != "0"      # A/B switch (tools/): 0 = attention_bwd_pv + qknorm_rope_bwd

_QK_LN_EPS = 1e-5      # nn.LayerNorm's default: Attention builds q_norm / k_norm as norm_layer(head_dim) (:60-61)


def _qk_layernorm_fwd(qkv, qnw, qnb, knw, knb, cos, sin, B, N, H, hd, dtype):
    """Attention front end of a block with use_qknorm=True and use_rmsnorm=False: q_norm / k_norm = nn.LayerNorm(head_dim) WITH weight and bias
    (:54-61), then RoPE (:71-73).  No shipped YAML builds it, so it is composed from kernels that exist: head-major split, the affine LayerNorm
    of the VMAE blocks on rows of head_dim (f32 statistics), the standalone rotation.  -> (q, k, v head-major, saved tuple for the backward)."""
    qs, ks, v = ops.heads_split(qkv, B, N, H, hd)
    q32, k32 = ops.cast(qs, torch.float32).view(-1, hd), ops.cast(ks, torch.float32).view(-1, hd)
    qn, muq, rsq = ops.layernorm_fwd(q32, qnw, qnb, dtype, _QK_LN_EPS)
    kn, muk, rsk = ops.layernorm_fwd(k32, knw, knb, dtype, _QK_LN_EPS)
    q, k = ops.rope(qn.view(B, H, N, hd), cos, sin), ops.rope(kn.view(B, H, N, hd), cos, sin)
    return q, k, v, (q32, k32, muq, rsq, muk, rsk)


def _qk_layernorm_bwd(dq, dk, dv, saved, qnw, knw, cos, sin, B, N, H, hd, dtype):
    """-> (dqkv [B*N, 3*H*hd], dqnw, dqnb, dknw, dknb, dbqkv)."""
    q32, k32, muq, rsq, muk, rsk = saved
    dxq, dxk = torch.zeros_like(q32), torch.zeros_like(k32)
    dqw, dqb = ops.layernorm_bwd(ops.rope(dq, cos, sin, transposed=True).view(-1, hd), q32, qnw, muq, rsq, dxq)
    dkw, dkb = ops.layernorm_bwd(ops.rope(dk, cos, sin, transposed=True).view(-1, hd), k32, knw, muk, rsk, dxk)
    dqkv = ops.heads_merge(ops.cast(dxq, dtype).view(B, H, N, hd), ops.cast(dxk, dtype).view(B, H, N, hd), dv, B, N, H, hd)
    return dqkv, dqw, dqb, dkw, dkb, ops.colsum(dqkv)


class _GradChain:
    """Hand-off between the backward passes of consecutive members of LightningDiT.forward's block chain (one object per forward).
    Backward of member j+1 ends with norm1's backward, which finishes the residual-stream gradient dx that member j's backward starts
    from with gate_bwd(dx, y2_j, gate_mlp_j): member j+1 runs both in one pass over the rows (ops.rmsnorm_modulate_bwd_gate, 805 MB
    less traffic per block) and leaves (dy2_j, db3_j, dmod_j with the gate slot filled) here.  `up[j]` = (y2, mod) of member j,
    registered in forward; `pre[j]` = what member j+1's backward prepared, keyed by the data pointer of the dx it belongs to so that
    member j falls back to its own gate_bwd if it is ever handed a different gradient buffer."""

    def __init__(self):
        self.up, self.pre = {}, {}
        self.dmod_all, self.mod_cols = None, 0      # batched adaLN (_AdaLNAllFn): ONE [B, depth * 6D] gradient buffer, a column slice per block

    def dmod(self, j, mod):
        """Gradient buffer of member j's modulation vectors `mod` [B, 6D]: a private tensor, or member j's column slice of the shared
        buffer when the adaLN Linears of all blocks run as one GEMM (their backward then is one GEMM over the whole buffer)."""
        if self.mod_cols == 0:
            return torch.empty(mod.shape, dtype=mod.dtype, device=mod.device)
        if self.dmod_all is None:
            self.dmod_all = torch.empty(mod.shape[0], self.mod_cols, dtype=torch.float32, device=mod.device)
        w = mod.shape[1]
        return self.dmod_all[:, j * w:(j + 1) * w]

    def norm_bwd(self, j, dout, x, w, scale, rstd, dx, dshift, dscale, N, dtype, accumulate=True):
        """norm backward of member j (accumulating into dx); fused with member j-1's gate backward when that member is registered."""
        D = x.shape[1]
        # consumed once: the chain must not keep y2 / mod of every block alive until the whole graph dies (0.4 GB per block at bs 256).
        # On a second pass over a retained graph the entry is gone and both members take their unfused paths.
        prev = self.up.pop(j - 1, None)
        if prev is None:
            return ops.rmsnorm_modulate_bwd(dout, x, w, scale, rstd, dx, dshift, dscale, N, accumulate)
        y2p, modp = prev
        dmodp = self.dmod(j - 1, modp)
        gc = modp.shape[1] - D                      # gate_mlp: the last chunk (:246 six chunks; :242 four with wo_shift)
        dw, dy2p, db3p = ops.rmsnorm_modulate_bwd_gate(dout, x, w, scale, rstd, dx, dshift, dscale, y2p, modp[:, gc:gc + D],
                                                       dmodp[:, gc:gc + D], N, dtype, accumulate)
        self.pre[j - 1] = (dx.data_ptr(), dy2p, db3p, dmodp)
        return dw


class _AdaLNAllFn(torch.autograd.Function):
    """The adaLN_modulation Linears (:233-236) of ALL blocks as one GEMM on the shared SiLU(c): mod_all [B, depth * 6D] = sc @ W_all^T +
    b_all, W_all = the depth weights stacked (one cast launch), bf16 operands with f32 accumulation and f32 output -- what the
    reference's autocast Linear computes (its output is bf16 on top).  Per block these were a 40-us f32 GEMM forward and, backward, two
    46-51-us f32 GEMMs, a transpose, two split reductions, a column sum and three AccumulateGrad adds (2.8 ms per B/1 step in ~190
    launches); batched: one NT GEMM forward, two TN GEMMs + a column sum backward.  The blocks write their modulation gradients into
    column slices of ONE buffer (_GradChain.dmod) and block 0 -- whose backward runs last -- hands that buffer to autograd as the
    gradient of mod_all; every block lists mod_all as an input, so this backward cannot start before all of them are done."""

    @staticmethod
    def forward(ctx, sc, fwd_only, *wb):
        n = len(wb) // 2
        ws, bs = wb[:n], wb[n:]
        scb = ops.cast(sc.contiguous(), torch.bfloat16)
        stack = ops.cached_stack_copy if fwd_only else ops.cast_stack       # sampling re-runs the same weights: keep the stacked copies
        W_all = stack(ws, torch.bfloat16)                           # [n * 6D, D]
        b_all = stack(bs, torch.float32)                            # [n * 6D]
        mod_all = ops.gemm_nt(scb, W_all, b_all, out_dtype=torch.float32)
        ctx.save_for_backward(scb, W_all)
        ctx.n, ctx.rows = n, ws[0].shape[0]
        return mod_all

    @staticmethod
    def backward(ctx, g):
        scb, W_all = ctx.saved_tensors
        n, rows = ctx.n, ctx.rows
        g = g.contiguous()
        gb = ops.cast(g, torch.bfloat16)
        dW_all = ops.gemm_tn(gb, scb)                               # [n * 6D, D] f32: contraction over the batch
        db_all = ops.colsum(g)
        dsc = ops.gemm_tn(gb.t().contiguous(), W_all) if ctx.needs_input_grad[0] else None      # [B, D]: contraction over all n * 6D outputs
        return (dsc, None) + tuple(dW_all[i * rows:(i + 1) * rows] for i in range(n)) + tuple(db_all[i * rows:(i + 1) * rows] for i in range(n))


class _DiTBlockFn(torch.autograd.Function):
    """LightningDiTBlock.forward (:239-250) with RMSNorm, QK-norm, RoPE, SwiGLU, shift."""

    @staticmethod
    def forward(ctx, x, sc, cos, sin, H, eps, dtype, inplace, chain, idx, direct, fwd_only, mod_all, swiglu,
                n1w, qkvw, qkvb, qnw, knw, qnb, knb, pw, pb, n2w, w12, b12, w3, b3, adaw, adab):
        """n1w / n2w None: LayerNorm without affine parameters (use_rmsnorm=False).  qnw / knw None: no QK-norm; with qnb / knb: nn.LayerNorm
        QK-norm (use_qknorm without use_rmsnorm).  swiglu False: w12 / b12 / w3 / b3 are fc1 / fc2 of the timm Mlp with tanh-GELU (use_swiglu=False)."""
        B, N, D = x.shape
        M, hd = B * N, D // H
        x2 = x.contiguous().view(M, D)
        sc = sc.contiguous()
        ctx.batched_ada = mod_all is not None
        nmod = adaw.shape[0] // D                                                        # 6, or 4 with wo_shift (:227-236)
        if mod_all is not None:                                                          # _AdaLNAllFn ran the Linear of every block at once
            mod = mod_all[:, idx * nmod * D:(idx + 1) * nmod * D]
        else:
            mod = ops.gemm_nt(sc, adaw, adab, out_dtype=torch.float32)                   # [B, 6D] f32
        if nmod == 6:
            sh1, s1, g1, sh2, s2, g2 = (mod[:, i * D:(i + 1) * D] for i in range(6))      # :246 chunk order
        else:
            s1, g1, s2, g2 = (mod[:, i * D:(i + 1) * D] for i in range(4))                # :242 (wo_shift: modulate without the shift)
            sh1 = sh2 = None
        # forward-only calls (torch.no_grad sampling: forward_with_cfg) skip everything only the backward pass reads: the transposed
        # weight copies, the pre-gate branch outputs y1 / y2 and h12 = [x1 | x2] of the SwiGLU (1.6 GB per XL/1 block at batch 128).
        # The flag comes from the module (grad mode is always off in here and needs_input_grad ignores torch.no_grad()).
        bwd = not fwd_only and any(ctx.needs_input_grad)
        Wqkv, WqkvT = _wcopies(qkvw, dtype, bwd)
        Wp, WpT = _wcopies(pw, dtype, bwd)
        ctx.hs = None
        if swiglu and w3.shape[1] % 128 != 0:          # LightningDiT-L / 1p6B: hidden 2730 / 4778 (see _pad_swiglu); b12 below is the padded bias
            w12, b12, w3, Hs_, Hp_ = _pad_swiglu(w12, b12, w3)
            ctx.hs = (Hs_, Hp_)
        W12, W12T = _wcopies(w12, dtype, bwd)
        W3, W3T = _wcopies(w3, dtype, bwd)
        # attention branch (:248)
        xm1, rstd1 = ops.rmsnorm_modulate_fwd(x2, n1w, sh1, s1, N, dtype, eps)
        qk_saved = None
        fused_qkv = qnb is None and dtype == torch.bfloat16 and ops.gemm_nt_qkv_rope_ok(xm1, Wqkv, B, N, H, hd)
        if not fused_qkv:
            qkv = ops.gemm_nt(xm1, Wqkv, qkvb)                                           # [M, 3D] == [B,N,3,H,hd]
        if qnb is not None:              # nn.LayerNorm QK-norm: composed path (see _qk_layernorm_fwd)
            q, k, v, qk_saved = _qk_layernorm_fwd(qkv, qnw, qnb, knw, knb, cos, sin, B, N, H, hd, dtype)
            o, lse = ops.attention_fwd(q, k, v, hd ** -0.5)
        elif dtype == torch.bfloat16:    # v is consumed where the qkv Linear wrote it: no head-major copy of v (nor of dv in backward)
            if fused_qkv:                # q_norm / k_norm / RoPE in the qkv GEMM's epilogue (bitwise the pair below; forward-only calls skip the pre-norm q / k)
                qkv, q, k = ops.gemm_nt_qkv_rope(xm1, Wqkv, qkvb, qnw, knw, cos, sin, B, N, H, hd, eps, store_raw_qk=bwd)
                v = None
            else:
                q, k, v = ops.qknorm_rope_fwd(qkv, qnw, knw, cos, sin, B, N, H, hd, eps, copy_v=False)
            # QK-RMSNorm bounds |q|, |k| by max|w| sqrt(hd) and the rotation keeps norms: a proven score bound, so the softmax runs with a
            # static shift (no running maximum in the kernel).  use_qknorm=False (qnw None; q_norm = nn.Identity, :60-61): nothing bounds the
            # scores, the kernel tracks the running maximum.
            o, lse = ops.attention_fwd_pv(q, k, qkv, hd ** -0.5, bound=ops.qk_score_bound(qnw, knw, hd, hd ** -0.5) if qnw is not None else None)      # [B,N,D]
        else:
            q, k, v = ops.qknorm_rope_fwd(qkv, qnw, knw, cos, sin, B, N, H, hd, eps)
            o, lse = ops.attention_fwd(q, k, v, hd ** -0.5)
        xmid, y1 = ops.gemm_nt_gate_res(o.view(M, D), Wp, pb, x2, g1, N, save_y=bwd)
        # MLP branch (:249)
        xm2, rstd2 = ops.rmsnorm_modulate_fwd(xmid, n2w, sh2, s2, N, dtype, eps)
        if swiglu:
            h12, hid = ops.gemm_nt_swiglu(xm2, W12, b12, save_h12=bwd)
        else:                            # timm Mlp (:219-224): fc1 -> GELU(tanh) -> fc2; h12 = fc1's pre-activation, hid = the activation
            h12 = ops.gemm_nt(xm2, W12, b12)
            hid = ops.gelu_tanh_fwd(h12)
        xout, y2 = ops.gemm_nt_gate_res(hid, W3, b3, xmid, g2, N, save_y=bwd)
        if not bwd:
            return xout.view(B, N, D)
        ctx.save_for_backward(x2, sc, cos, sin, mod, rstd1, xm1, qkv, q, k, v, o, lse, y1, xmid, rstd2, xm2, h12, hid, y2,
                              n1w, qnw, knw, n2w, adaw, WqkvT, WpT, W12T, W3T, *(qk_saved or ()))
        ctx.dims = (B, N, D, H, hd, eps, dtype)
        ctx.swiglu, ctx.qk_ln = bool(swiglu), qk_saved is not None
        ctx.nmod = nmod
        ctx.inplace = bool(inplace)
        ctx.chain, ctx.idx = chain, idx
        ctx.direct, ctx.wparams = bool(direct), (qkvw, pw, w12, w3)      # (padded hidden: w12 / w3 are the padded copies -- no .grad, so the direct path declines them)
        ctx.sparams = (n1w, qkvb, qnw, knw, pb, n2w, b12, b3, qnb, knb) if direct else None
        if chain is not None:
            chain.up[idx] = (y2, mod)
        return xout.view(B, N, D)

    @staticmethod
    def backward(ctx, gout):
        (x2, sc, cos, sin, mod, rstd1, xm1, qkv, q, k, v, o, lse, y1, xmid, rstd2, xm2, h12, hid, y2,
         n1w, qnw, knw, n2w, adaw, WqkvT, WpT, W12T, W3T) = ctx.saved_tensors[:29]
        qk_saved = ctx.saved_tensors[29:] if ctx.qk_ln else None
        B, N, D, H, hd, eps, dtype = ctx.dims
        M = B * N
        # f32 residual-stream gradient.  `inplace` (set by LightningDiT.forward for its own block chain, where a block output
        # feeds only the next block / final layer, whose backward allocates the buffer handed to us): accumulate IN PLACE in
        # that buffer -- no 805 MB copy per block.  Otherwise (a block called on its own, or with hooks tapping its output) the
        # incoming gradient may be shared with another consumer and must not be modified: work on a private copy.
        dx = gout.contiguous().view(M, D)
        if not ctx.inplace and dx.data_ptr() == gout.data_ptr():
            dx = dx.clone()
        chain, idx = ctx.chain, ctx.idx
        # column chunk of each modulation vector in mod / dmod: (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp), :246; no shifts with wo_shift, :242
        c_sh1, c_s1, c_g1, c_sh2, c_s2, c_g2 = (0, 1, 2, 3, 4, 5) if ctx.nmod == 6 else (None, 0, 1, None, 2, 3)
        col = lambda t_, c_: None if c_ is None else t_[:, c_ * D:(c_ + 1) * D]      # noqa: E731
        s1, g1, s2, g2 = col(mod, c_s1), col(mod, c_g1), col(mod, c_s2), col(mod, c_g2)
        sg = ops.SideGemms(dx.device, enabled=dtype == torch.bfloat16)      # weight gradients: off the critical path
        # ---- MLP branch
        pre = chain.pre.pop(idx, None) if chain is not None else None
        if pre is not None and pre[0] == dx.data_ptr():      # the next member's backward already gated this dx (see _GradChain)
            _, dy2, db3, dmod = pre
        else:
            dmod = chain.dmod(idx, mod) if chain is not None else torch.empty(mod.shape, dtype=mod.dtype, device=mod.device)
            dy2, db3 = ops.gate_bwd(dx, y2, g2, col(dmod, c_g2), N, dtype, with_bias=True)   # bias grads where dy is produced
        qkvw_p, pw_p, w12_p, w3_p = ctx.wparams
        notify = []
        dW3, r = _dw_into_grad(sg, dy2, hid, w3_p, ctx.direct); notify.append((r, w3_p))
        if ctx.swiglu:
            dh12, db12 = ops.gemm_nt_swiglu_bwd(dy2, W3T, h12, with_bias=True)
        else:                            # timm Mlp: fc2's input gradient through the tanh-GELU backward; fc1's bias gradient = its column sums
            dh12 = ops.gelu_tanh_bwd(ops.gemm_nt(dy2, W3T), h12)
            db12 = ops.colsum(dh12)
        dW12, r = _dw_into_grad(sg, dh12, xm2, w12_p, ctx.direct); notify.append((r, w12_p))
        dxm2 = ops.gemm_nt(dh12, W12T)
        # norm2 backward and the attention branch's gate backward in one pass (the updated dx is consumed from registers)
        dn2, dy1, dbp = ops.rmsnorm_modulate_bwd_gate(dxm2, xmid, n2w, s2, rstd2, dx, col(dmod, c_sh2), col(dmod, c_s2),
                                                      y1, g1, col(dmod, c_g1), N, dtype)
        # ---- attention branch
        dWp, r = _dw_into_grad(sg, dy1, o.view(M, D), pw_p, ctx.direct); notify.append((r, pw_p))
        do = ops.gemm_nt(dy1, WpT)
        dqnb = dknb = None
        if qk_saved is not None:         # nn.LayerNorm QK-norm (composed path)
            dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, hd ** -0.5)
            dqkv, dqn, dqnb, dkn, dknb, dbqkv = _qk_layernorm_bwd(dq, dk, dv, qk_saved, qnw, knw, cos, sin, B, N, H, hd, dtype)
        elif v is None and hd in (64, 128) and N % 64 == 0 and _FUSED_QKN_BWD:      # QK-norm / RoPE backward inside the attention backward's epilogues: no head-major dq / dk
            dqkv, dqn, dkn, dbqkv = ops.attention_bwd_pv_qknorm(q, k, qkv, o, do, lse, hd ** -0.5, qnw, knw, cos, sin, eps)
        elif v is None:
            dq, dk, dqkv = ops.attention_bwd_pv(q, k, qkv, o, do, lse, hd ** -0.5)          # dv lands in the v slot of dqkv
            dqkv, dqn, dkn, dbqkv = ops.qknorm_rope_bwd(dq, dk, None, qkv, qnw, knw, cos, sin, B, N, H, hd, eps, with_bias=True, dqkv=dqkv)
        else:
            dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, hd ** -0.5)
            dqkv, dqn, dkn, dbqkv = ops.qknorm_rope_bwd(dq, dk, dv, qkv, qnw, knw, cos, sin, B, N, H, hd, eps, with_bias=True)
        dqkv = dqkv.view(M, 3 * D)
        dWqkv, r = _dw_into_grad(sg, dqkv, xm1, qkvw_p, ctx.direct); notify.append((r, qkvw_p))
        dxm1 = ops.gemm_nt(dqkv, WqkvT)
        if chain is not None:
            dn1 = chain.norm_bwd(idx, dxm1, x2, n1w, s1, rstd1, dx, col(dmod, c_sh1), col(dmod, c_s1), N, dtype)
        else:
            dn1 = ops.rmsnorm_modulate_bwd(dxm1, x2, n1w, s1, rstd1, dx, col(dmod, c_sh1), col(dmod, c_s1), N)
        # ---- adaLN: per block in f32, or (batched) nothing here -- block 0, the last to run, returns the shared dmod buffer for mod_all
        dmod_all = None
        if ctx.batched_ada:
            dadaw = dadab = dsc = None
            if idx == 0:
                dmod_all = chain.dmod_all
        else:
            dadaw, dadab = ops.gemm_tn(dmod, sc), ops.colsum(dmod)
            dsc = _dmod_times_w(dmod, adaw)
        if ctx.hs is not None:                         # padded SwiGLU hidden: hand autograd the real units' rows / columns
            dW12, db12, dW3 = _unpad_swiglu_grads(dW12, db12, dW3, *ctx.hs)
        sg.join()
        # the eight small gradients of the block (norm weights, biases, QK-norm weights): with `direct`, ONE launch adds them into their .grad
        # views instead of one AccumulateGrad add each
        small = [dn1, dbqkv, dqn, dkn, dbp, dn2, db12, db3, dqnb, dknb]
        if ctx.sparams is not None:
            pairs = [(p_, g_) for p_, g_ in zip(ctx.sparams, small) if p_ is not None]      # use_qknorm=False: no q_norm / k_norm weights
            if all(g_ is not None and p_.grad is not None and p_.grad.dtype == torch.float32 and p_.grad.is_contiguous() and p_.grad.shape == g_.shape
                   for p_, g_ in pairs):
                ops.multi_add_([p_.grad for p_, _ in pairs], [g_ for _, g_ in pairs])
                notify.extend((getattr(p_, "_ldmae_grad_ready", None), p_) for p_, _ in pairs)
                dn1 = dbqkv = dqn = dkn = dbp = dn2 = db12 = db3 = dqnb = dknb = None
        for r, p_ in notify:          # gradients written straight into .grad: tell the reducer (no-op without one)
            if r is not None:
                r(p_)
        return (dx.view(B, N, D), dsc, None, None, None, None, None, None, None, None, None, None, dmod_all, None,
                dn1, dWqkv, dbqkv, dqn, dkn, dqnb, dknb, dWp, dbp, dn2, dW12, db12, dW3, db3, dadaw, dadab)


class _AttentionFn(torch.autograd.Function):
    """Attention.forward (:66-91) on its own: qkv Linear -> QK-RMSNorm -> RoPE -> softmax attention -> proj, the same kernels the block
    uses.  Activations in `dtype` (bf16 under autocast, like the reference's autocast Linear / SDPA; else f32)."""

    @staticmethod
    def forward(ctx, x, cos, sin, H, eps, dtype, qkvw, qkvb, qnw, knw, qnb, knb, pw, pb):
        B, N, D = x.shape
        M, hd = B * N, D // H
        xa = ops.cast(x.contiguous().view(M, D), dtype)
        Wqkv, WqkvT = _wcopies(qkvw, dtype, True)
        Wp, WpT = _wcopies(pw, dtype, True)
        qkv = ops.gemm_nt(xa, Wqkv, qkvb)
        qk_saved = None
        if qnb is not None:              # nn.LayerNorm QK-norm (use_rmsnorm=False): composed path
            q, k, v, qk_saved = _qk_layernorm_fwd(qkv, qnw, qnb, knw, knb, cos, sin, B, N, H, hd, dtype)
            o, lse = ops.attention_fwd(q, k, v, hd ** -0.5)
        elif dtype == torch.bfloat16:
            q, k, v = ops.qknorm_rope_fwd(qkv, qnw, knw, cos, sin, B, N, H, hd, eps, copy_v=False)
            o, lse = ops.attention_fwd_pv(q, k, qkv, hd ** -0.5, bound=ops.qk_score_bound(qnw, knw, hd, hd ** -0.5) if qnw is not None else None)
        else:
            q, k, v = ops.qknorm_rope_fwd(qkv, qnw, knw, cos, sin, B, N, H, hd, eps)
            o, lse = ops.attention_fwd(q, k, v, hd ** -0.5)
        out = ops.gemm_nt(o.view(M, D), Wp, pb)
        ctx.save_for_backward(xa, cos, sin, qkv, q, k, v, o, lse, qnw, knw, WqkvT, WpT, *(qk_saved or ()))
        ctx.dims = (B, N, D, H, hd, eps, x.dtype)
        ctx.qk_ln = qk_saved is not None
        return out.view(B, N, D)

    @staticmethod
    def backward(ctx, g):
        xa, cos, sin, qkv, q, k, v, o, lse, qnw, knw, WqkvT, WpT = ctx.saved_tensors[:13]
        B, N, D, H, hd, eps, xdt = ctx.dims
        M = B * N
        dy = ops.cast(g.contiguous().view(M, D), xa.dtype)
        dWp, dbp = ops.gemm_tn(dy, o.view(M, D), with_bias=True)
        do = ops.gemm_nt(dy, WpT)
        dqnb = dknb = None
        if ctx.qk_ln:
            dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, hd ** -0.5)
            dqkv, dqn, dqnb, dkn, dknb, dbqkv = _qk_layernorm_bwd(dq, dk, dv, ctx.saved_tensors[13:], qnw, knw, cos, sin, B, N, H, hd, xa.dtype)
        elif v is None:
            dq, dk, dqkv = ops.attention_bwd_pv(q, k, qkv, o, do, lse, hd ** -0.5)
            dqkv, dqn, dkn, dbqkv = ops.qknorm_rope_bwd(dq, dk, None, qkv, qnw, knw, cos, sin, B, N, H, hd, eps, with_bias=True, dqkv=dqkv)
        else:
            dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, hd ** -0.5)
            dqkv, dqn, dkn, dbqkv = ops.qknorm_rope_bwd(dq, dk, dv, qkv, qnw, knw, cos, sin, B, N, H, hd, eps, with_bias=True)
        dqkv = dqkv.view(M, 3 * D)
        dWqkv = ops.gemm_tn(dqkv, xa)
        dx = ops.gemm_nt(dqkv, WqkvT).view(B, N, D).to(xdt) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None, None, dWqkv, dbqkv, dqn, dkn, dqnb, dknb, dWp, dbp


class _FinalLayerFn(torch.autograd.Function):
    """FinalLayer.forward (:267-272): adaLN(2) -> RMSNorm -> modulate -> Linear."""

    @staticmethod
    def forward(ctx, x, sc, eps, dtype, chain, idx, nw, lw, lb, adaw, adab):
        B, N, D = x.shape
        M = B * N
        x2 = x.contiguous().view(M, D)
        sc = sc.contiguous()
        ctx.chain, ctx.idx = chain, idx
        mod = ops.gemm_nt(sc, adaw, adab, out_dtype=torch.float32)
        xf, rstd = ops.rmsnorm_modulate_fwd(x2, nw, mod[:, :D], mod[:, D:], N, dtype, eps)
        out = ops.gemm_nt(xf, ops.cast(lw, dtype), lb, out_dtype=torch.float32)
        ctx.save_for_backward(x2, sc, mod, rstd, xf, nw, lw, adaw)
        ctx.dims = (B, N, D, dtype)
        return out.view(B, N, -1)

    @staticmethod
    def backward(ctx, gout):
        x2, sc, mod, rstd, xf, nw, lw, adaw = ctx.saved_tensors
        B, N, D, dtype = ctx.dims
        M = B * N
        g = gout.contiguous().view(M, -1)
        ga = ops.cast(g, dtype)
        dlw, dlb = ops.gemm_tn(ga, xf), ops.colsum(g)
        if g.dtype == torch.float32 and ops.thin_ok(lw.shape[1], lw.shape[0]):                        # K = p*p*C = 16 / 32: ldmae_thin_nt
            dxf = ops.thin_nt(g, lw.float().t().contiguous(), out_dtype=dtype)
        else:
            dxf = ops.gemm_nt(g, ops.cast_weight(lw, torch.float32, True, False)[1], out_dtype=dtype)     # f32 MFMA
        dx = torch.empty(M, D, dtype=torch.float32, device=g.device)     # written, not accumulated into: no 805 MB memset + read
        dmod = torch.empty_like(mod)
        if ctx.chain is not None:
            dnw = ctx.chain.norm_bwd(ctx.idx, dxf, x2, nw, mod[:, D:], rstd, dx, dmod[:, :D], dmod[:, D:], N, dtype, accumulate=False)
        else:
            dnw = ops.rmsnorm_modulate_bwd(dxf, x2, nw, mod[:, D:], rstd, dx, dmod[:, :D], dmod[:, D:], N, accumulate=False)
        dadaw, dadab = ops.gemm_tn(dmod, sc), ops.colsum(dmod)
        dsc = _dmod_times_w(dmod, adaw)
        return dx.view(B, N, D), dsc, None, None, None, None, dnw, dlw, dlb, dadaw, dadab


# ----------------------------------------------------------------------------- modules (reference names / keys)
class PatchEmbed(nn.Module):
    """Stand-in for timm.models.vision_transformer.PatchEmbed with the attributes the reference
    touches (.proj.weight/bias, .patch_size, .num_patches, .grid_size; :309-312,354-356,382)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.norm = nn.Identity()

    def tokens(self, x):
        """[B,C,H,W] -> [B*T, C*p*p] in the conv weight's (c, i, j) order (layout plumbing only)."""
        B, C, Hh, Ww = x.shape
        p = self.patch_size[0]
        if p == 1:
            return x.float().reshape(B, C, Hh * Ww).transpose(1, 2).reshape(B * Hh * Ww, C)
        x = x.float().reshape(B, C, Hh // p, p, Ww // p, p).permute(0, 2, 4, 1, 3, 5)
        return x.reshape(B * (Hh // p) * (Ww // p), C * p * p)

    def forward(self, x, pos=None, dtype=None):
        B = x.shape[0]
        w2d = self.proj.weight.view(self.proj.weight.shape[0], -1)
        if pos is None:
            pos = torch.zeros(self.num_patches, w2d.shape[0], device=x.device)
        tok = self.tokens(x)
        if w2d.shape[1] % 16 != 0:        # C * p * p off the GEMMs' 16-grid (4-channel latents at patch size 1: K = 4): zero columns on both operands -- exact,
            pad = (-w2d.shape[1]) % 16    # and the weight gradient comes back through the pad sliced to the real columns
            tok, w2d = torch.nn.functional.pad(tok, (0, pad)), torch.nn.functional.pad(w2d, (0, pad))
        return _PatchEmbedFn.apply(tok, w2d, self.proj.bias, pos, self.num_patches,
                                   dtype or torch.float32).view(B, self.num_patches, -1)


class TimestepEmbedder(nn.Module):
    """:94-137."""

    def __init__(self, hidden_size: int, frequency_embedding_size: int = 256) -> None:
        super().__init__()
        self.frequency_embedding_size = frequency_embedding_size
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size, bias=True))

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        return _TimestepFreqFn.apply(t, dim)

    def forward(self, t):
        f = self.timestep_embedding(t, self.frequency_embedding_size)
        h = _LinearF32Fn.apply(f, self.mlp[0].weight, self.mlp[0].bias)
        return _LinearF32Fn.apply(_SiluFn.apply(h), self.mlp[2].weight, self.mlp[2].bias)


class LabelEmbedder(nn.Module):
    """:140-169.  The label-drop draw (`torch.rand(B) < p`, :157) happens here on the device RNG,
    exactly where the reference draws it."""

    def __init__(self, num_classes, hidden_size, dropout_prob):
        super().__init__()
        use_cfg_embedding = dropout_prob > 0
        self.embedding_table = nn.Embedding(num_classes + use_cfg_embedding, hidden_size)
        self.num_classes = num_classes
        self.dropout_prob = dropout_prob

    def token_drop(self, labels, force_drop_ids=None):
        """:152-161 (integer work: the dropped labels become the null class)."""
        return torch.where(self.token_drop_ids(labels, force_drop_ids), torch.full_like(labels, self.num_classes), labels)

    def token_drop_ids(self, labels, force_drop_ids=None):
        if force_drop_ids is None:
            return torch.rand(labels.shape[0], device=labels.device) < self.dropout_prob
        return force_drop_ids == 1

    def forward(self, labels, train, force_drop_ids=None):
        drop = None
        if (train and self.dropout_prob > 0) or (force_drop_ids is not None):
            drop = self.token_drop_ids(labels, force_drop_ids).to(torch.uint8).contiguous()
        return _LabelEmbedFn.apply(self.embedding_table.weight, labels.to(torch.int64).contiguous(), drop, self.num_classes)


class Attention(nn.Module):
    """:32-91 (qkv, q_norm, k_norm, proj).  Inside LightningDiTBlock the branch runs as part of _DiTBlockFn; called on its own
    (`block.attn(x, rope)`, the reference's signature) it runs the same kernels through _AttentionFn."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False, use_rmsnorm=False, **_):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'

        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        norm_layer = RMSNorm if use_rmsnorm else nn.LayerNorm                 # :54-61
        self.q_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.proj = nn.Linear(dim, dim)
        self.precision = None

    def norm_args(self):
        """(eps, q_norm weight, k_norm weight, q_norm bias, k_norm bias): weights None with qk_norm=False (q_norm = k_norm = nn.Identity, :60-61);
        biases only for the nn.LayerNorm form (use_rmsnorm=False)."""
        if isinstance(self.q_norm, RMSNorm):
            return self.q_norm.eps, self.q_norm.weight, self.k_norm.weight, None, None
        if isinstance(self.q_norm, nn.LayerNorm):
            return self.q_norm.eps, self.q_norm.weight, self.k_norm.weight, self.q_norm.bias, self.k_norm.bias
        return 1e-6, None, None, None, None

    def forward(self, x, rope=None):
        B, N, C = x.shape
        cos, sin = (rope.freqs_cos, rope.freqs_sin) if rope is not None else _identity_rope(N, self.head_dim, x.device)      # :71
        dtype = _act_dtype(self.precision)
        with torch.autocast(device_type="cuda", enabled=False):
            eps, qnw, knw, qnb, knb = self.norm_args()
            return _AttentionFn.apply(x, cos, sin, self.num_heads, eps, dtype, self.qkv.weight, self.qkv.bias, qnw, knw, qnb, knb, self.proj.weight, self.proj.bias)


_IDENTITY_ROPE: dict = {}


def _identity_rope(N, hd, device):
    """cos = 1, sin = 0 tables: the RoPE kernels then return their input bit for bit (x * 1 - y * 0) -- the call without a rotation
    (use_rope=False: feat_rope = None, :71,324-325) on the same kernels."""
    key = (N, hd, str(device))
    t = _IDENTITY_ROPE.get(key)
    if t is None:
        t = _IDENTITY_ROPE[key] = (torch.ones(N, hd, device=device), torch.zeros(N, hd, device=device))
    return t


class Mlp(nn.Module):
    """timm.models.vision_transformer.Mlp's parameter layout (fc1 / fc2) as the block builds it with use_swiglu=False (:219-224: GELU with the
    tanh approximation, no dropout); computed inside _DiTBlockFn."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=None, drop=0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = nn.GELU(approximate="tanh")
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)


class LightningDiTBlock(nn.Module):
    """:171-250, every constructor flag.  Tuned for the shipped form (RMSNorm + SwiGLU, both YAMLs), with or without QK-norm (configs/imagenet/
    ...yaml:28 true; configs/celeba_hq/...yaml:30 false), the adaLN shift (wo_shift) and RoPE.  use_rmsnorm=False (LayerNorm without affine
    parameters on the same norm kernels; nn.LayerNorm QK-norm composed from the VMAE LayerNorm + rotation kernels) and use_swiglu=False (timm
    Mlp with tanh-GELU: GEMMs + an elementwise pass) run on kernels too, but no shipped YAML builds them and they are not tuned."""

    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, use_qknorm=False, use_swiglu=False, use_rmsnorm=False,
                 wo_shift=False, **block_kwargs):
        super().__init__()
        if use_rmsnorm:
            self.norm1, self.norm2 = RMSNorm(hidden_size), RMSNorm(hidden_size)
        else:                                                                             # :199-201
            self.norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
            self.norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.attn = Attention(hidden_size, num_heads=num_heads, qkv_bias=True, qk_norm=use_qknorm, use_rmsnorm=use_rmsnorm)
        if use_swiglu:
            self.mlp = SwiGLUFFN(hidden_size, int(2 / 3 * int(hidden_size * mlp_ratio)))
        else:                                                                             # :219-224
            self.mlp = Mlp(in_features=hidden_size, hidden_features=int(hidden_size * mlp_ratio))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, (4 if wo_shift else 6) * hidden_size, bias=True))
        self.wo_shift = wo_shift
        self.precision = None

    def forward(self, x, c, feat_rope=None, _silu_c=None, _dtype=None, _inplace_grad=False, _chain=None, _idx=0, _direct=False, _mod_all=None):
        sc = _silu_c if _silu_c is not None else _SiluFn.apply(c.float())
        a, m = self.attn, self.mlp
        cos, sin = (feat_rope.freqs_cos, feat_rope.freqs_sin) if feat_rope is not None else _identity_rope(x.shape[1], a.head_dim, x.device)
        _, qnw, knw, qnb, knb = a.norm_args()
        swiglu = isinstance(m, SwiGLUFFN)
        l1, l2 = (m.w12, m.w3) if swiglu else (m.fc1, m.fc2)
        return _DiTBlockFn.apply(
            x.float(), sc, cos, sin, a.num_heads, self.norm1.eps, _dtype or _act_dtype(self.precision),
            _inplace_grad, _chain, _idx, _direct, not torch.is_grad_enabled(), _mod_all, swiglu,
            self.norm1.weight, a.qkv.weight, a.qkv.bias, qnw, knw, qnb, knb, a.proj.weight, a.proj.bias,
            self.norm2.weight, l1.weight, l1.bias, l2.weight, l2.bias,
            self.adaLN_modulation[1].weight, self.adaLN_modulation[1].bias)


class FinalLayer(nn.Module):
    """:252-272."""

    def __init__(self, hidden_size, patch_size, out_channels, use_rmsnorm=False):
        super().__init__()
        self.norm_final = RMSNorm(hidden_size) if use_rmsnorm else nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)      # :256-259
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))
        self.precision = None

    def forward(self, x, c, _silu_c=None, _dtype=None, _chain=None, _idx=0):
        sc = _silu_c if _silu_c is not None else _SiluFn.apply(c.float())
        lw, lb = self.linear.weight, self.linear.bias
        nout = lw.shape[0]
        if nout % 16 != 0:                # p * p * out_channels off the GEMMs' grid (4- or 3-channel latents at patch size 1): zero rows, sliced off below
            pad = (-nout) % 16
            lw, lb = torch.nn.functional.pad(lw, (0, 0, 0, pad)), torch.nn.functional.pad(lb, (0, pad))
        out = _FinalLayerFn.apply(x.float(), sc, self.norm_final.eps, _dtype or _act_dtype(self.precision), _chain, _idx,
                                  self.norm_final.weight, lw, lb, self.adaLN_modulation[1].weight, self.adaLN_modulation[1].bias)
        return out if out.shape[-1] == nout else out[..., :nout]


class LightningDiT(nn.Module):
    """:275-442 -- same constructor signature, attributes and state-dict keys."""

    def __init__(self, input_size=32, patch_size=2, in_channels=32, hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4.0,
                 class_dropout_prob=0.1, num_classes=1000, learn_sigma=False, use_qknorm=False, use_swiglu=False,
                 use_rope=False, use_rmsnorm=False, wo_shift=False, use_checkpoint=False):
        super().__init__()
        # opt-in of the training driver (which owns a gradient slab and calls plain loss.backward()): the four Linear weight gradients
        # of every block are accumulated straight into param.grad by the GEMM's reduce instead of through autograd's AccumulateGrad
        self.direct_param_grads = False
        # batched adaLN (_AdaLNAllFn).  None = automatic: on, except in a process of a torch.distributed world of more than one rank -- the
        # weight gradients of the batched form complete at the very END of backward, which delays every gradient bucket that holds one of
        # them (the reference's train_accum.py under accelerate / DDP through the drop-in: DDP's buckets interleave them with everything
        # else).  A driver that lays the adaLN weights out FIRST in its gradient slab (optim.adaln_first) sets True.  LDMAE_BATCHED_ADALN=0 / 1
        # overrides (A/B and parity tests).
        env = os.environ.get("LDMAE_BATCHED_ADALN")
        self.batched_adaln = None if env is None else env != "0"
        self.learn_sigma = learn_sigma
        self.in_channels = in_channels
        self.out_channels = in_channels if not learn_sigma else in_channels * 2
        self.patch_size = patch_size
        self.num_heads = num_heads
        self.use_rope = use_rope
        self.use_rmsnorm = use_rmsnorm
        self.depth = depth
        self.hidden_size = hidden_size
        self.use_checkpoint = use_checkpoint
        self.x_embedder = PatchEmbed(input_size, patch_size, in_channels, hidden_size, bias=True)
        self.t_embedder = TimestepEmbedder(hidden_size)
        self.y_embedder = LabelEmbedder(num_classes, hidden_size, class_dropout_prob)
        num_patches = self.x_embedder.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, hidden_size), requires_grad=False)
        self.feat_rope = VisionRotaryEmbeddingFast(dim=hidden_size // num_heads // 2, pt_seq_len=input_size // patch_size) if use_rope else None      # :317-325
        self.blocks = nn.ModuleList([
            LightningDiTBlock(hidden_size, num_heads, mlp_ratio=mlp_ratio, use_qknorm=use_qknorm, use_swiglu=use_swiglu,
                              use_rmsnorm=use_rmsnorm, wo_shift=wo_shift) for _ in range(depth)])
        self.final_layer = FinalLayer(hidden_size, patch_size, self.out_channels, use_rmsnorm=use_rmsnorm)
        self.precision = None          # None: follow torch.autocast; or torch.float32 / torch.bfloat16
        self.initialize_weights()

    def set_precision(self, dtype):
        self.precision = dtype
        for b in self.blocks:
            b.precision = dtype
            b.attn.precision = dtype
        self.final_layer.precision = dtype
        return self

    def initialize_weights(self):
        """:340-374."""
        def _basic_init(module):
            if isinstance(module, nn.Linear):
                torch.nn.init.xavier_uniform_(module.weight)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)
        self.apply(_basic_init)
        pos_embed = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.x_embedder.num_patches ** 0.5))
        self.pos_embed.data.copy_(torch.from_numpy(pos_embed).float().unsqueeze(0))
        w = self.x_embedder.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.constant_(self.x_embedder.proj.bias, 0)
        nn.init.normal_(self.y_embedder.embedding_table.weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        for block in self.blocks:
            nn.init.constant_(block.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(block.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].weight, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)

    def unpatchify(self, x):
        """:376-389 (pure re-layout of the [N, T, p*p*C] kernel output)."""
        c = self.out_channels
        p = self.x_embedder.patch_size[0]
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.reshape(shape=(x.shape[0], h, w, p, p, c))
        x = torch.einsum('nhwpqc->nchpwq', x)
        return x.reshape(shape=(x.shape[0], c, h * p, h * p))

    def forward(self, x, t=None, y=None):
        """:391-418.  The activation dtype is read from the ambient autocast state once; the body then runs with
        autocast disabled so the few torch re-layout ops around the kernels stay f32 (the reference's output is
        f32 too: accelerate converts outputs to fp32)."""
        dtype = _act_dtype(self.precision)
        if dtype != torch.float32 and self.hidden_size % 64 != 0:
            dtype = torch.float32         # the 16-bit MFMA GEMMs contract in steps of 64: other widths (no registry entry has one) keep f32 activations
        with torch.autocast(device_type="cuda", enabled=False):
            x = self.x_embedder(x, self.pos_embed[0])
            t = self.t_embedder(t)
            y = self.y_embedder(y, self.training)
            c = t + y
            sc = _SiluFn.apply(c)
            # a block output of this chain has exactly one consumer (the next block / the final layer) unless someone taps it with a
            # module hook: only then may a block's backward re-use the incoming gradient buffer, and only when that holds for the
            # whole chain do consecutive backward passes hand work to each other (_GradChain)
            hooked = [bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks) for m in self.blocks]
            fl = self.final_layer
            chain = None
            if torch.is_grad_enabled() and not self.use_checkpoint and not any(hooked) and \
                    not (fl._forward_hooks or fl._forward_pre_hooks or fl._backward_hooks):
                chain = _GradChain()
            # the adaLN Linears of all blocks as ONE bf16 GEMM (_AdaLNAllFn): training needs the chain (its shared gradient buffer),
            # inference just takes the forward.  f32 mode, checkpointing, hooked blocks and batches that are not a multiple of 8 (the
            # bf16 TN GEMM's alignment) keep the per-block f32 GEMMs.
            mod_all = None
            if self._use_batched_adaln() and dtype == torch.bfloat16 and not self.use_checkpoint and len(self.blocks) <= 64 and sc.shape[0] % 8 == 0 and \
                    (chain is not None or not torch.is_grad_enabled()):
                lins = [b.adaLN_modulation[1] for b in self.blocks]
                mod_all = _AdaLNAllFn.apply(sc, not torch.is_grad_enabled(), *[l.weight for l in lins], *[l.bias for l in lins])
                if chain is not None:
                    chain.mod_cols = mod_all.shape[1]
            for i, block in enumerate(self.blocks):
                if self.use_checkpoint:
                    x = checkpoint(block, x, c, self.feat_rope, sc, dtype, not hooked[i], use_reentrant=True)
                else:
                    x = block(x, c, self.feat_rope, sc, dtype, not hooked[i], chain, i, self.direct_param_grads and chain is not None, mod_all)
            x = self.final_layer(x, c, sc, dtype, chain, len(self.blocks))
            x = self.unpatchify(x)
            if self.learn_sigma:
                x, _ = x.chunk(2, dim=1)
        return x

    def _use_batched_adaln(self) -> bool:
        if self.batched_adaln is not None:
            return bool(self.batched_adaln)
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and torch.is_grad_enabled())

    def forward_with_cfg(self, x, t, y, cfg_scale, cfg_interval=None, cfg_interval_start=None):
        """:420-442 (CFG on the first three channels only; interval gate on t[0])."""
        half = x[: len(x) // 2]
        combined = torch.cat([half, half], dim=0)
        model_out = self.forward(combined, t, y)
        eps, rest = model_out[:, :3], model_out[:, 3:]
        cond_eps, uncond_eps = torch.split(eps, len(eps) // 2, dim=0)
        half_eps = uncond_eps + cfg_scale * (cond_eps - uncond_eps)
        if cfg_interval is True:
            timestep = t[0]
            if timestep < cfg_interval_start:
                half_eps = cond_eps
        eps = torch.cat([half_eps, half_eps], dim=0)
        return torch.cat([eps, rest], dim=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, extra_tokens=0):
    """Reference name kept (:444-458); the table itself is built in ldmae_amd/tables.py."""
    return sincos_2d(embed_dim, grid_size, np.float64, cls_token=bool(cls_token and extra_tokens > 0))


# ----------------------------------------------------------------------------- registry (:498-531)
def LightningDiT_XL_1(**kw): return LightningDiT(depth=28, hidden_size=1152, patch_size=1, num_heads=16, **kw)
def LightningDiT_XL_2(**kw): return LightningDiT(depth=28, hidden_size=1152, patch_size=2, num_heads=16, **kw)
def LightningDiT_L_2(**kw): return LightningDiT(depth=24, hidden_size=1024, patch_size=2, num_heads=16, **kw)
def LightningDiT_B_1(**kw): return LightningDiT(depth=12, hidden_size=768, patch_size=1, num_heads=12, **kw)
def LightningDiT_B_2(**kw): return LightningDiT(depth=12, hidden_size=768, patch_size=2, num_heads=12, **kw)
def LightningDiT_1p0B_1(**kw): return LightningDiT(depth=24, hidden_size=1536, patch_size=1, num_heads=24, **kw)
def LightningDiT_1p0B_2(**kw): return LightningDiT(depth=24, hidden_size=1536, patch_size=2, num_heads=24, **kw)
def LightningDiT_1p6B_1(**kw): return LightningDiT(depth=28, hidden_size=1792, patch_size=1, num_heads=28, **kw)
def LightningDiT_1p6B_2(**kw): return LightningDiT(depth=28, hidden_size=1792, patch_size=2, num_heads=28, **kw)


LightningDiT_models = {
    'LightningDiT-B/1': LightningDiT_B_1, 'LightningDiT-B/2': LightningDiT_B_2,
    'LightningDiT-L/2': LightningDiT_L_2,
    'LightningDiT-XL/1': LightningDiT_XL_1, 'LightningDiT-XL/2': LightningDiT_XL_2,
    'LightningDiT-1p0B/1': LightningDiT_1p0B_1, 'LightningDiT-1p0B/2': LightningDiT_1p0B_2,
    'LightningDiT-1p6B/1': LightningDiT_1p6B_1, 'LightningDiT-1p6B/2': LightningDiT_1p6B_2,
}
