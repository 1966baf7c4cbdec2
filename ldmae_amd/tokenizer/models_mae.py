"""VMAE tokenizer (masked-autoencoder ViT) on the gfx950 kernels -- mirror of the reference module
``tokenizer/models_mae.py`` (citations relative to /root/reference/LDMAE/tokenizer/models_mae.py).

Hot path = masked-token ENCODER: patch-embed GEMM(+pos) -> ``random_masking`` (in-LDS stable sort, bit-exact) -> gather
kept tokens -> 12 pre-LN ViT blocks (LayerNorm, qkv GEMM, flash attention on the kept subset, proj GEMM + residual,
fc1 GEMM + exact GELU, fc2 GEMM + residual) -> LayerNorm.  Each block is one autograd Function (forward + backward
kernel sequences).  head_dim is 16: under bf16 autocast the attention core is the bf16 flash kernel with the K/V images and
fragments zero-padded to 32 columns in LDS / registers (attention.hip); in f32 mode everything runs on the exact-f32 MFMA path.
The decoder / ``encode`` / ``decode`` / ``decode_to_images`` docking functions re-use the same block kernels
(inference only for the RGB smoothing conv).  Unshipped variants (gradual_resol, cls token, pred_with_conv, perceptual loss) raise
NotImplementedError; `down_nonlinear` (MLP latent maps of the f8d16 / f8d16_flexible archs) is implemented.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from functools import partial
from typing import Optional

import numpy as np
import weakref

import torch
import torch.nn as nn

from ldmae_amd import ops
from ldmae_amd.models.lightningdit import PatchEmbed, _act_dtype, _wcopies
from . import fused_encoder
from .util.misc import DiagonalGaussianDistribution
from .util.pos_embed import get_2d_sincos_pos_embed


class Config:
    def __init__(self, scaling_factor):
        self.scaling_factor = scaling_factor


@dataclass
class DecoderOutput:
    sample: torch.Tensor
    commit_loss: Optional[torch.Tensor] = None


@dataclass
class EncoderOutput:
    latent: torch.Tensor

    def sample(self):
        return self.latent

    def mode(self):
        return self.latent


@dataclass
class MAEOutput:
    latent_dist: object


# ----------------------------------------------------------------------------- host-side image preprocessing (:85-103, :935-950)
def center_crop_arr(pil_image, image_size):
    """ADM's centre crop as the reference applies it (:85-103): halve with a BOX filter while the short side is >= 2x the target, one
    BICUBIC resize that brings the short side to the target, then the central image_size x image_size window."""
    from PIL import Image
    while min(pil_image.size) >= 2 * image_size:
        pil_image = pil_image.resize((pil_image.size[0] // 2, pil_image.size[1] // 2), resample=Image.BOX)
    scale = image_size / min(pil_image.size)
    pil_image = pil_image.resize((round(pil_image.size[0] * scale), round(pil_image.size[1] * scale)), resample=Image.BICUBIC)
    arr = np.asarray(pil_image)
    top, left = (arr.shape[0] - image_size) // 2, (arr.shape[1] - image_size) // 2
    return Image.fromarray(arr[top:top + image_size, left:left + image_size])


class ImageTransform:
    """What ``img_transform`` returns (:935-950) without torchvision: centre crop -> horizontal flip with probability p_hflip (one
    ``torch.rand(1)`` draw per image, like RandomHorizontalFlip) -> CHW float in [0, 1] -> (x - 0.5) / 0.5."""

    def __init__(self, p_hflip, img_size):
        self.p_hflip, self.img_size = float(p_hflip), int(img_size)

    def __call__(self, pil_image):
        arr = np.asarray(center_crop_arr(pil_image.convert("RGB"), self.img_size))
        if torch.rand(1) < self.p_hflip:
            arr = arr[:, ::-1]
        x = torch.from_numpy(np.array(arr)).permute(2, 0, 1).to(torch.float32).div_(255.0)
        return x.sub_(0.5).div_(0.5)


import contextlib


@contextlib.contextmanager
def reference_tf32():
    """`torch.backends.cuda.matmul.allow_tf32 = True` for the duration of a driver's tokenizer calls, restored afterwards.  The reference's
    drivers set the flag for the whole process (inference.py:79, extract_features.py:2-3); the drivers here are also called in-process (tests,
    bench.py), so they scope it.  With it on, f32 forward-only calls of the tokenizer run TF32-class (MaskedAutoencoderViT._docking_dtype)."""
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = True
    try:
        yield
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev


# ----------------------------------------------------------------------------- autograd Functions
class _ViTBlockFn(torch.autograd.Function):
    """Block.forward (:176-187): x += proj(attn(LN1(x))); x += fc2(gelu(fc1(LN2(x))))."""

    @staticmethod
    def forward(ctx, x, H, eps, dtype, inplace, fwd_only, direct, yres_f32, n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b):
        B, N, D = x.shape
        M, hd = B * N, D // H
        x2 = x.contiguous().view(M, D)
        # forward-only calls (encode / decode under no_grad; the flag comes from the module: grad mode is always off in here) skip what only
        # the backward pass reads
        bwd = not fwd_only and any(ctx.needs_input_grad)
        Wqkv, WqkvT = _wcopies(qkvw, dtype, bwd)
        Wp, WpT = _wcopies(pw, dtype, bwd)
        W1, W1T = _wcopies(f1w, dtype, bwd)
        W2, W2T = _wcopies(f2w, dtype, bwd)
        h1, mu1, rs1 = ops.layernorm_fwd(x2, n1w, n1b, dtype, eps)
        qkv = ops.gemm_nt(h1, Wqkv, qkvb)                                     # activation dtype: bf16 under autocast
        if dtype == torch.float16 and hd != 16:
            raise RuntimeError("ldmae_amd: the fp16 kernel family covers head_dim 16 (the shipped VMAE heads)")
        if dtype in (torch.bfloat16, torch.float16) or (hd == 16 and not bwd):
            # flash kernel on the packed qkv as the Linear wrote it (bf16: head_dim 16 padded to 32 in LDS; f32 inference at head_dim 16: the
            # 16x16x4-MFMA kernel reads the packed rows too -- no head-major relayout.  The f32 BACKWARD kernels take head-major q / k / v.)
            q = k = v = None
            o, lse = ops.attention_fwd_qkv(qkv, B, N, H, hd, hd ** -0.5)
        else:
            q, k, v = ops.heads_split(qkv, B, N, H, hd)
            o, lse = ops.attention_fwd(q, k, v, hd ** -0.5)                    # [B,N,D]
            qkv = None
        oa = o.view(M, D)
        # yres_f32 (the TF32-class docking calls: fp16 operands = inputs rounded to a 10-bit mantissa, f32 accumulation): like a TF32 Linear, the
        # branch output joins the residual stream UNROUNDED.  Under fp16 / bf16 AUTOCAST the Linear's output is the activation type, rounded.
        ydt = torch.float32 if yres_f32 else None
        xmid, _ = ops.gemm_nt_gate_res(oa, Wp, pb, x2, None, N, save_y=False, y_dtype=ydt)
        h2, mu2, rs2 = ops.layernorm_fwd(xmid, n2w, n2b, dtype, eps)
        act, pre = ops.gemm_nt_gelu(h2, W1, f1b, save_pre=bwd)
        xout, _ = ops.gemm_nt_gate_res(act, W2, f2b, xmid, None, N, save_y=False, y_dtype=ydt)
        if not bwd:
            return xout.view(B, N, D)
        ctx.save_for_backward(x2, h1, mu1, rs1, qkv, q, k, v, o, lse, oa, xmid, h2, mu2, rs2, act, pre, n1w, n2w, WqkvT, WpT, W1T, W2T)
        ctx.dims = (B, N, D, H, hd, dtype)
        ctx.inplace = bool(inplace)
        ctx.x_ref = weakref.ref(x)            # backward looks for TENSOR hooks on the block's input (see the `_ldmae_cast` hand-off there)
        # opt-in of the training driver (MaskedAutoencoderViT.direct_param_grads; every .grad is a view of its gradient slab): the block's
        # twelve parameter gradients are ADDED into their .grad by the producing kernels (the TN GEMM's reduce with beta = 1; one
        # ldmae_multi_add for the eight vectors) instead of by twelve AccumulateGrad passes -- as models/lightningdit.py does for the DiT
        plist = (n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b)
        ctx.plist = plist if direct and all(p_.grad is not None and p_.grad.dtype == torch.float32 and p_.grad.is_contiguous() for p_ in plist) else None
        return xout.view(B, N, D)

    @staticmethod
    def backward(ctx, gout):
        x2, h1, mu1, rs1, qkv, q, k, v, o, lse, oa, xmid, h2, mu2, rs2, act, pre, n1w, n2w, WqkvT, WpT, W1T, W2T = ctx.saved_tensors
        B, N, D, H, hd, dtype = ctx.dims
        M = B * N
        dx = gout.contiguous().view(M, D)
        if not ctx.inplace and dx.data_ptr() == gout.data_ptr():
            dx = dx.clone()                   # the incoming gradient may have other consumers: never modify it (see _DiTBlockFn)
        pl = ctx.plist

        def dw(dy, xin, i):           # weight + bias gradient of Linear i (index of its weight in plist; its bias is plist[i + 1])
            if pl is not None:        # both straight into their slab views (beta = 1): no zero-filled bias buffer, no add afterwards
                ops.gemm_tn(dy, xin, out=pl[i].grad, beta=1.0, with_bias=True, dbias_out=pl[i + 1].grad.view(-1))
                return None, None
            return ops.gemm_tn(dy, xin, with_bias=True)
        half = dtype != torch.float32
        # MLP branch.  The residual gradient rounded to the activation type is the operand of both Linear backwards of a branch: the LayerNorm
        # backward that last updated dx writes that copy beside it (ops.layernorm_bwd(cast=True)); across blocks it rides on the returned tensor
        # (`_ldmae_cast`: the next block's backward finds it only if autograd hands over that very tensor, unmodified) -- else one cast pass.
        stash = getattr(gout, "_ldmae_cast", None) if half else None
        dy2 = stash[0] if stash is not None and stash[1] == gout._version and stash[0].dtype == dtype and dx.data_ptr() == stash[2] else ops.cast(dx, dtype)
        dW2, db2 = dw(dy2, act, 10)
        dpre = ops.gemm_nt_gelu_bwd(dy2, W2T, pre)                 # fc2's input gradient with the GELU backward in the GEMM epilogue
        dW1, db1 = dw(dpre, h2, 8)
        # attention branch
        if half:
            dn2w, dn2b, dy1 = ops.layernorm_bwd(ops.gemm_nt(dpre, W1T, grad=True), xmid, n2w, mu2, rs2, dx, cast=True)
        else:
            dn2w, dn2b = ops.layernorm_bwd(ops.gemm_nt(dpre, W1T, grad=True), xmid, n2w, mu2, rs2, dx)
            dy1 = dx
        dWp, dbp = dw(dy1, oa, 4)
        do = ops.gemm_nt(dy1, WpT, grad=True)
        if qkv is not None:
            dqkv = ops.attention_bwd_qkv(qkv, o, do, lse, B, N, H, hd, hd ** -0.5)
        else:
            dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, hd ** -0.5)
            dqkv = ops.heads_merge(dq, dk, dv, B, N, H, hd)
        dWqkv, dbqkv = dw(dqkv, h1, 2)
        if half and ctx.inplace:
            dn1w, dn1b, dxc = ops.layernorm_bwd(ops.gemm_nt(dqkv, WqkvT, grad=True), x2, n1w, mu1, rs1, dx, cast=True)
        else:
            dn1w, dn1b = ops.layernorm_bwd(ops.gemm_nt(dqkv, WqkvT, grad=True), x2, n1w, mu1, rs1, dx)
            dxc = None
        if pl is not None:
            small = [(0, dn1w), (1, dn1b), (6, dn2w), (7, dn2b)]          # (the four bias gradients went into the slab with their weights)
            ops.multi_add_([pl[i].grad for i, _ in small], [g_ for _, g_ in small])
            for p_ in pl:              # a data-parallel reducer's post-accumulate hook does not fire for these: tell it (no-op without one)
                r = getattr(p_, "_ldmae_grad_ready", None)
                if r is not None:
                    r(p_)
            dn1w = dn1b = dbqkv = dbp = dn2w = dn2b = db1 = db2 = None
        gx = dx.view(B, N, D)
        # the rounded copy rides on the returned tensor only while nothing can touch the gradient between the two blocks: module hooks switch
        # `inplace` off (_chain_ok), and a TENSOR hook on the block's input (x.register_hook: it runs on gx before the previous block's backward,
        # and one that edits it through a raw-pointer op would not bump `_version`) drops the hand-off -- the previous block then casts dx itself
        xin = ctx.x_ref()
        if dxc is not None and xin is not None and not xin._backward_hooks:
            gx._ldmae_cast = (dxc, gx._version, dx.data_ptr())
        return (gx, None, None, None, None, None, None, None, dn1w, dn1b, dWqkv, dbqkv, dWp, dbp, dn2w, dn2b, dW1, db1, dW2, db2)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1])
        out, mu, rs = ops.layernorm_fwd(x2, w, b, torch.float32, eps)
        ctx.save_for_backward(x2, w, mu, rs)
        return out.view(shp)

    @staticmethod
    def backward(ctx, g):
        x2, w, mu, rs = ctx.saved_tensors
        dx = torch.zeros_like(x2)
        dw, db = ops.layernorm_bwd(g.contiguous().view_as(x2), x2, w, mu, rs, dx)
        return dx.view(g.shape), dw, db, None


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T + b (f32 in/out; small latent projections :307-308, 378)."""

    @staticmethod
    def forward(ctx, x, w, b):
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1])
        ctx.save_for_backward(x2, w)
        return ops.gemm_nt(x2, w, b).view(*shp[:-1], -1)

    @staticmethod
    def backward(ctx, g):
        x2, w = ctx.saved_tensors
        g2 = g.contiguous().view(x2.shape[0], -1)
        dx = ops.gemm_nt(g2, ops.cast_weight(w, torch.float32, True, False)[1]).view(*g.shape[:-1], -1) if ctx.needs_input_grad[0] else None
        return dx, ops.gemm_tn(g2, x2), ops.colsum(g2)


def _linear(x, w, b):
    """_LinearFn with output widths off the GEMMs' 16-grid (the prediction head of a patch-14 tokenizer: 14 * 14 * 3 = 588, mae_vit_huge_patch14) zero-padded:
    extra zero rows of the weight, sliced off the result; the weight gradient comes back through the pad cut to the real rows."""
    n = w.shape[0]
    if n % 16 == 0:
        return _LinearFn.apply(x, w, b)
    pad = (-n) % 16
    return _LinearFn.apply(x, torch.nn.functional.pad(w, (0, 0, 0, pad)), torch.nn.functional.pad(b, (0, pad)))[..., :n]


class _Conv3x3Fn(torch.autograd.Function):
    """conv_smoother (:254,275): 3x3 / stride 1 / pad 1 on the RGB prediction, f32."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        return ops.conv3x3(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx, dw, db = ops.conv3x3_bwd(g.contiguous(), x, w, need_dx=ctx.needs_input_grad[0])
        return dx, dw, db


class _MaeLossFn(torch.autograd.Function):
    """forward_loss (:733-754) without norm_pix_loss, in IMAGE space: (mask_loss, visible_loss) from the smoothing conv's output image, the input
    images and the patch mask -- all patches have p*p*C elements, so the two means over patches are weighted sums over pixels (csrc/vmae.hip:
    mae_loss_fwd / bwd); patch counts and the two backward coefficients stay on the device."""

    @staticmethod
    def forward(ctx, pred_img, imgs, mask, p):
        sums = ops.mae_loss_fwd(pred_img, imgs, mask, p)
        cm = mask.sum()
        cnt = torch.stack([cm, mask.numel() - cm]) * float(p * p * imgs.shape[1])
        out = sums / cnt
        ctx.save_for_backward(pred_img, imgs, mask, cnt)
        ctx.p = p
        return out[0], out[1]

    @staticmethod
    def backward(ctx, gm, gv):
        pred_img, imgs, mask, cnt = ctx.saved_tensors
        z = torch.zeros((), device=cnt.device)
        coef = torch.stack([gm if gm is not None else z, gv if gv is not None else z]).float() / cnt
        return ops.mae_loss_bwd(pred_img, imgs, mask, coef, ctx.p), None, None, None


class _GatherFn(torch.autograd.Function):
    """torch.gather(x, 1, ids_keep) on token rows (:486)."""

    @staticmethod
    def forward(ctx, x, ids_keep):
        ctx.save_for_backward(ids_keep)
        ctx.L = x.shape[1]
        return ops.gather_rows(x.contiguous(), ids_keep)

    @staticmethod
    def backward(ctx, g):
        (ids_keep,) = ctx.saved_tensors
        return ops.scatter_rows(g, ids_keep, ctx.L), None


class _RestoreTokensFn(torch.autograd.Function):
    """The decoder's input (:536-541): cat([x, mask_token.repeat]) -> gather(ids_restore) -> + decoder_pos_embed as one kernel each way
    (ops.restore_tokens; backward: kept rows gathered back, the mask token's gradient = column sums of the masked rows)."""

    @staticmethod
    def forward(ctx, x, mask_token, pos, ids_restore):
        ctx.save_for_backward(ids_restore)
        ctx.keep, ctx.mshape = x.shape[1], mask_token.shape
        return ops.restore_tokens(x, mask_token.reshape(-1), pos.reshape(pos.shape[-2], pos.shape[-1]), ids_restore)

    @staticmethod
    def backward(ctx, g):
        (ids_restore,) = ctx.saved_tensors
        dx, dm = ops.restore_tokens_bwd(g, ids_restore, ctx.keep, need_mask_grad=ctx.needs_input_grad[1])
        return (dx if ctx.needs_input_grad[0] else None), (dm.view(ctx.mshape) if dm is not None else None), None, None


# ----------------------------------------------------------------------------- modules
class Mlp(nn.Module):
    """timm Mlp parameter layout (fc1 / fc2), computed inside _ViTBlockFn."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)


class Attention(nn.Module):
    """:117-128 parameter container."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    """:149-187 with init_values=None, drop_path=0 (the shipped archs)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., init_values=None, drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        if init_values or drop_path > 0 or drop > 0 or attn_drop > 0 or not qkv_bias:
            raise NotImplementedError("ldmae_amd Block: LayerScale / DropPath / dropout / bias-free qkv are not on the accelerated path")
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)
        self.precision = None

    def forward(self, x, _inplace_grad=False, dtype=None, tf32_class=False):
        """dtype: the activation type of THIS call (the model's encoder / decoder loops pass the type they resolved -- a caller that has
        switched autocast off around the stack passes what it read before doing so); None: the block's own `precision` setting / autocast.
        tf32_class: the call is one of the TF32-class docking calls (decided where the type is resolved, MaskedAutoencoderViT._docking_dtype --
        never here: the model's loops run with autocast switched off, so the ambient autocast state says nothing about the caller's): fp16
        operands with the branch outputs joining the residual stream UNROUNDED, as a TF32 Linear's do.  fp16 autocast training is fp16 with
        tf32_class False: Linear outputs rounded to fp16, as torch's autocast produces them.  `last_dtype` / `last_tf32_class` record what the
        call ran in (tests)."""
        a, m = self.attn, self.mlp
        dtype = dtype if dtype is not None else _act_dtype(self.precision, allow_f16=True)
        D_, rows_ = self.norm1.weight.numel(), x.shape[0] * x.shape[1]
        if dtype != torch.float32 and (D_ % 64 != 0 or m.fc1.weight.shape[0] % 64 != 0):
            # the 16-bit MFMA GEMMs contract in steps of 64: widths off that grid (96: the pre-training tree's mae_for_ldmae_f8d16_small,
            # VMAE/models_mae.py:1036-1048) keep f32 activations on the exact-f32 kernels -- more precision than the caller's autocast asked for, never less
            dtype = torch.float32
        if dtype == torch.float16 and not tf32_class and (D_ // a.num_heads != 16 or rows_ % 8 != 0):
            # fp16 AUTOCAST (the reference's torch.amp.autocast('cuda'), engine_pretrain.py:51-57) on a geometry the fp16 kernel family does not cover -- it is built
            # for the shipped heads of 16 and whole groups of 8 token rows (no fallback GEMM): f32 activations, by the same rule
            dtype = torch.float32
        self.last_dtype = dtype
        yres_f32 = self.last_tf32_class = bool(tf32_class) and dtype == torch.float16
        return _ViTBlockFn.apply(x.float(), a.num_heads, self.norm1.eps, dtype, _inplace_grad,
                                 not torch.is_grad_enabled(), bool(getattr(self, "direct_param_grads", False)), yres_f32, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                                 self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)


class MLP_dim_resize(nn.Module):
    """:232-242 -- to_latent / from_latent of the `down_nonlinear` archs (mae_for_ldmae_f8d16, _flexible): Linear -> exact GELU -> Linear, hidden
    width 4 * latent_dim.  The two Linears run on the GEMM kernels (_LinearFn); the GELU between them acts on [tokens, 64] and stays in torch."""

    def __init__(self, input_dim, hidden_dim, output_dim):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, output_dim))

    def forward(self, x):
        h = torch.nn.functional.gelu(_LinearFn.apply(x, self.layers[0].weight, self.layers[0].bias))
        return _LinearFn.apply(h, self.layers[2].weight, self.layers[2].bias)


def _latent_map(mod, x):
    """to_latent / from_latent: a plain Linear (:316-317) or MLP_dim_resize (:311-314)."""
    return mod(x) if isinstance(mod, MLP_dim_resize) else _LinearFn.apply(x, mod.weight, mod.bias)


class conv_decoder_pred(nn.Module):
    """:244-281 with pred_with_conv=False: linear -> unpatchify -> 3x3 conv on RGB -> patchify."""

    def __init__(self, decoder_embed_dim, patch_size, in_chans, pred_with_conv=False):
        super().__init__()
        if pred_with_conv:
            raise NotImplementedError("ldmae_amd conv_decoder_pred: pred_with_conv=True is not shipped")
        self.p = patch_size
        self.pred_with_conv = False
        self.linear_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.conv_smoother = nn.Conv2d(in_chans, in_chans, 3, 1, 1)

    def forward(self, x, return_image=False):
        h = w = int(x.shape[1] ** .5)
        x = _linear(x, self.linear_pred.weight, self.linear_pred.bias)
        x = x.reshape(x.shape[0], h, w, self.p, self.p, 3)
        x = torch.einsum('nhwpqc->nchpwq', x).reshape(x.shape[0], 3, h * self.p, w * self.p)
        img = _Conv3x3Fn.apply(x, self.conv_smoother.weight, self.conv_smoother.bias)
        x = img.reshape(img.shape[0], 3, h, self.p, w, self.p)
        pred = torch.einsum('nchpwq->nhwpqc', x).reshape(x.shape[0], h * w, self.p * self.p * 3)
        return (pred, img) if return_image else pred            # img: what the pre-training loss reads (MaskedAutoencoderViT.forward)


class MaskedAutoencoderViT(nn.Module):
    """:283-973 -- constructor signature of the reference; accelerated for the shipped tokenizer configuration."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                 decoder_depth=8, decoder_num_heads=16, mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False, latent_dim=32,
                 ldmae_mode=False, scaling_factor=0.9654248952865601, no_cls=True, gradual_resol=False, finetune_downsample_layer=None,
                 down_nonlinear=False, kl_loss_weight=None, smooth_output=False, pred_with_conv=False, perceptual_loss=None,
                 perceptual_loss_ratio=1.0, fixed_std=None):
        """The constructor of BOTH reference trees: LDMAE/tokenizer/models_mae.py:287-292 and the pre-training tree's VMAE/models_mae.py:288-293, which
        adds `perceptual_loss_ratio` and `fixed_std`.  `kl_form` (attribute; "tokenizer" | "vmae"): which tree's posterior KL the training forward
        uses -- they differ (tokenizer/util/misc.DiagonalGaussianDistribution); passing `fixed_std` selects "vmae", ldmae_amd/vmae_pretrain.py sets it."""
        super().__init__()
        self.fixed_std, self.kl_form = fixed_std, ("vmae" if fixed_std is not None else "tokenizer")
        if gradual_resol or not no_cls or perceptual_loss is not None:
            raise NotImplementedError("ldmae_amd MaskedAutoencoderViT: gradual_resol / cls token / perceptual loss are "
                                      "not used by the shipped tokenizer (mae_for_ldmae_f8d16_prev, inference.py:133-137)")
        self.perceptual_loss, self.smooth_output, self.gradual_resol, self.kl_loss_weight = None, smooth_output, False, kl_loss_weight
        enc_lat = 2 * latent_dim if kl_loss_weight is not None else latent_dim
        if down_nonlinear:                                   # :311-314 (from_latent ends at EMBED_dim; decoder_embed then maps it to the decoder width)
            self.to_latent = MLP_dim_resize(embed_dim, latent_dim * 4, enc_lat)
            self.from_latent = MLP_dim_resize(latent_dim, latent_dim * 4, embed_dim)
        else:
            self.to_latent = nn.Linear(embed_dim, enc_lat)
            # VMAE/models_mae.py:320 (the pre-training tree): the latent maps back to the ENCODER width, and decoder_embed (:371) takes it to the decoder's.
            # LDMAE/tokenizer/models_mae.py:317 says decoder_embed_dim here, which only runs when the two widths are equal (decoder_embed expects embed_dim):
            # same shapes wherever that copy works at all; the asymmetric archs (mae_for_ldmae_f8d16_asym_small, mae_vit_*_dec512d8b) need this form
            self.from_latent = nn.Linear(latent_dim, embed_dim)
        self.config = Config(scaling_factor=scaling_factor)
        self.ldmae_mode, self.img_size, self.patch_size = ldmae_mode, img_size, patch_size
        self.latent_resolution = img_size // patch_size
        self.tile_latent_min_size = self.latent_resolution
        self.latent_dim, self.no_cls, self.num_extra_tokens = latent_dim, no_cls, 0
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        if not self.ldmae_mode:
            self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
                                             for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        if smooth_output:
            self.decoder_pred = conv_decoder_pred(decoder_embed_dim, patch_size, in_chans, pred_with_conv)
        else:
            self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.norm_pix_loss = norm_pix_loss
        self.precision = None
        self.fused_encoder = os.environ.get("LDMAE_VMAE_FUSED", "1") != "0"      # False / LDMAE_VMAE_FUSED=0: always the per-layer kernels (A/B and parity tests)
        self.direct_param_grads = False
        self.initialize_weights()

    def initialize_weights(self):
        """:432-465."""
        g = int(self.patch_embed.num_patches ** .5)
        self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.pos_embed.shape[-1], g)).float().unsqueeze(0))
        self.decoder_pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], g)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        if not self.ldmae_mode:
            torch.nn.init.normal_(self.mask_token, std=.02)

        def _init(m):
            if isinstance(m, nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self.apply(_init)

    def set_direct_param_grads(self, on=True):
        """Opt-in of a training driver whose optimizer keeps every .grad as a view of one gradient slab (vmae_pretrain.build_optimizer): the
        blocks add their parameter gradients into .grad themselves (see _ViTBlockFn)."""
        self.direct_param_grads = bool(on)
        for b in list(self.blocks) + list(self.decoder_blocks):
            b.direct_param_grads = bool(on)
        return self

    def set_precision(self, dtype):
        self.precision = dtype
        for b in list(self.blocks) + list(self.decoder_blocks):
            b.precision = dtype
        return self

    # ---- layout helpers (:467-495)
    def patchify(self, imgs):
        p = self.patch_embed.patch_size[0]
        h = w = imgs.shape[2] // p
        x = imgs.reshape(imgs.shape[0], 3, h, p, w, p)
        return torch.einsum('nchpwq->nhwpqc', x).reshape(imgs.shape[0], h * w, p ** 2 * 3)

    def unpatchify(self, x):
        p = self.patch_embed.patch_size[0]
        h = w = int(x.shape[1] ** .5)
        x = x.reshape(x.shape[0], h, w, p, p, 3)
        return torch.einsum('nhwpqc->nchpwq', x).reshape(x.shape[0], 3, h * p, h * p)

    def random_masking(self, x, mask_ratio, noise=None):
        """:472-497.  The uniform noise is drawn on the device RNG exactly like the reference (pass `noise` to pin it); the
        argsort / ids_restore / mask / gather run in the HIP kernels (stable: ties broken by index)."""
        N, L, D = x.shape
        len_keep = int(L * (1 - mask_ratio))
        if noise is None:
            noise = torch.rand(N, L, device=x.device)
        ids_keep, mask, ids_restore = ops.random_masking(noise.float().contiguous(), len_keep)
        return _GatherFn.apply(x.float(), ids_keep), mask, ids_restore

    def _embed(self, x, dtype=None):
        return self.patch_embed(x, self.pos_embed[0], dtype if dtype != torch.float16 else torch.float32)

    def _docking_dtype(self, blocks, rows=None):
        """(activation type, tf32_class) of a forward-only stack call.  f32 calls (no autocast, no `set_precision`) made while the caller has set
        ``torch.backends.cuda.matmul.allow_tf32 = True`` -- what the reference's drivers do before they call `_encode` / `decode`
        (inference.py:79, extract_features.py:2-3) -- run TF32-CLASS: gfx950 has no TF32 MFMA, but fp16 has exactly TF32's 10-bit mantissa at the
        bf16 rate, and the operands it is used for (LayerNorm outputs, q / k / v, softmax probabilities, GELU outputs) are O(1); accumulation,
        residual stream, LayerNorm and softmax statistics stay f32, as under TF32.  End to end (12 blocks) 5.6e-4 relative against exact f32 --
        the same as emulated TF32 (tests/test_gpu_mae.py).  With the flag off (torch's default) f32 calls stay on the exact-f32 MFMA kernels,
        the 1e-4 parity path.  LDMAE_TF32=0 keeps them there regardless."""
        dtype = _act_dtype(self.precision, allow_f16=True)
        if dtype != torch.float32 or torch.is_grad_enabled() or not torch.backends.cuda.matmul.allow_tf32 or os.environ.get("LDMAE_TF32", "1") == "0":
            return dtype, False
        # the fp16 whole-line GEMM has no fallback kernel: token rows in whole groups of 8 (M % 8 == 0; `rows` = B * tokens of this call) and widths on
        # 128-B lines, or the call stays on the exact-f32 kernels (an odd batch of an odd-grid image, e.g. 72 px -> 81 tokens at patch 8)
        ok = all(b.norm1.weight.numel() // b.attn.num_heads == 16 and b.norm1.weight.numel() % 64 == 0 and b.mlp.fc1.weight.shape[0] % 64 == 0 for b in blocks)
        ok = ok and (rows is None or rows % 8 == 0)
        return (torch.float16, True) if ok else (dtype, False)

    @staticmethod
    def _chain_ok(blk):
        """A block of the encoder / decoder chain feeds only the next block or the closing LayerNorm (whose backward allocates the
        gradient buffer it hands on) unless a module hook taps it: only then may its backward re-use the incoming gradient."""
        return not (blk._forward_hooks or blk._forward_pre_hooks or blk._backward_hooks)

    def _run(self, blocks, x, dtype=None, tf32_class=False):
        for blk in blocks:
            x = blk(x, self._chain_ok(blk), dtype, tf32_class)
        return x

    def forward_encoder(self, x, mask_ratio, noise=None):
        """:499-523."""
        dtype, tf32 = self._docking_dtype(self.blocks, rows=x.shape[0] * int(self.patch_embed.num_patches * (1 - mask_ratio)))
        with torch.autocast(device_type="cuda", enabled=False):
            # inference in bf16 on the shipped geometry with 256 kept tokens (mask_ratio 0.75): mask FIRST (it depends on the noise alone),
            # embed only the kept quarter of the patches, then the whole stack + the closing LayerNorm as ONE kernel, one workgroup per
            # image, the residual stream in registers (csrc/vmae_fused.hip)
            pe = self.patch_embed
            L = pe.num_patches
            keep = int(L * (1 - mask_ratio))
            if (self.fused_encoder and dtype in (torch.bfloat16, torch.float16) and not torch.is_grad_enabled() and x.dim() == 4 and
                    fused_encoder.supported(self, keep, self.pos_embed.shape[-1], tiled=True) and all(self._chain_ok(blk) for blk in self.blocks)):
                if noise is None:
                    noise = torch.rand(x.shape[0], L, device=x.device)             # the draw random_masking makes (:480)
                ids_keep, mask, ids_restore = ops.random_masking(noise.float().contiguous(), keep)
                w2d = pe.proj.weight.view(pe.proj.weight.shape[0], -1)
                xk = ops.patch_embed_kept(x, ids_keep, self.pos_embed[0], w2d, pe.proj.bias, pe.patch_size[0], dtype if dtype != torch.float16 else torch.float32)
                # 256 kept tokens in bf16: the whole stack in one launch; 512 / 768 / 1024 (mask_ratio 0.5 / 0.25 / 0) and the TF32-class (fp16)
                # calls: the tiled form
                if keep == fused_encoder.TOKENS and dtype == torch.bfloat16:
                    return fused_encoder.encoder_forward(self, xk), mask, ids_restore
                return fused_encoder.encoder_forward_tiled(self, xk, dtype=dtype), mask, ids_restore
            x = self._embed(x, dtype)
            x, mask, ids_restore = self.random_masking(x, mask_ratio, noise)
            x = self._run(self.blocks, x, dtype, tf32)
            x = _LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x, mask, ids_restore

    def forward_decoder(self, x, ids_restore, dtype=None, return_image=False):
        """:525-554 (no cls token).  `dtype`: activation type of the decoder blocks; a caller that has switched autocast off around this call
        (forward) passes the type it read BEFORE doing so -- the blocks would otherwise see "no autocast" and run their f32 kernels (the
        1024-token decoder of the pre-training step did: 250 of its 304 ms)."""
        dtype = dtype if dtype is not None else _act_dtype(self.precision, allow_f16=True)
        x = _LinearFn.apply(x, self.decoder_embed.weight, self.decoder_embed.bias)
        if x.dtype == torch.float32 and x.shape[2] % 4 == 0 and x.shape[2] <= 512 and not self.decoder_pos_embed.requires_grad and x.shape[1] <= ids_restore.shape[1]:
            # kept rows back in place + mask tokens + position embedding in ONE pass each way (ldmae_restore_tokens)
            x = _RestoreTokensFn.apply(x, self.mask_token, self.decoder_pos_embed, ids_restore)
        else:                                # (a trainable position table, odd widths: the reference's four tensor ops)
            mask_tokens = self.mask_token.repeat(x.shape[0], ids_restore.shape[1] - x.shape[1], 1)
            x_ = torch.cat([x, mask_tokens], dim=1)
            x = torch.gather(x_, dim=1, index=ids_restore.unsqueeze(-1).repeat(1, 1, x.shape[2]))
            x = x + self.decoder_pos_embed
        x = self._run(self.decoder_blocks, x, dtype)
        x = _LayerNormFn.apply(x, self.decoder_norm.weight, self.decoder_norm.bias, self.decoder_norm.eps)
        if return_image:                     # (pred, smoothed image): only with the conv_decoder_pred head
            return self.decoder_pred(x, return_image=True)
        return self.decoder_pred(x) if not isinstance(self.decoder_pred, nn.Linear) else \
            _linear(x, self.decoder_pred.weight, self.decoder_pred.bias)

    def forward_loss(self, imgs, pred, mask, visible_loss_ratio=0.5):
        """:733-754."""
        target = self.patchify(imgs)
        if self.norm_pix_loss:
            mean, var = target.mean(dim=-1, keepdim=True), target.var(dim=-1, keepdim=True)
            target = (target - mean) / (var + 1.e-6) ** .5
        loss = ((pred - target) ** 2).mean(dim=-1)
        visible_loss = (loss * (1 - mask)).sum() / (1 - mask).sum()
        mask_loss = (loss * mask).sum() / mask.sum()
        return (1 - visible_loss_ratio) * mask_loss + visible_loss_ratio * visible_loss, visible_loss, mask_loss

    def forward(self, imgs, mask_ratio=0.75, visible_loss_ratio=0.5, _noise=None, _eps=None):
        """forward_vanilla (:756-790), trainable end to end on the kernels (the VMAE pre-training step of engine_pretrain.py:51-76):
        masked encoder -> to_latent -> KL + posterior sample -> from_latent -> decoder (mask tokens, blocks, RGB smoothing conv) ->
        masked / visible reconstruction loss.  `_noise` [B, L] / `_eps` [B, latent, kept] (tests): host-drawn masking noise and
        posterior-sample noise instead of the device RNG draws the reference makes at the same two places."""
        if self.ldmae_mode:
            raise NotImplementedError("ldmae_amd: ldmae_mode (decoder fine-tuning with LPIPS) is out of scope")
        dtype = _act_dtype(self.precision, allow_f16=True)           # read before autocast is switched off below
        latent, mask, ids_restore = self.forward_encoder(imgs, mask_ratio, noise=_noise)
        with torch.autocast(device_type="cuda", enabled=False):
            latent = _latent_map(self.to_latent, latent)
            kl_loss = None
            if self.kl_loss_weight is not None:
                B, N, D = latent.shape
                posterior = DiagonalGaussianDistribution(latent.permute(0, 2, 1), fixed_std=self.fixed_std, pretrain_tree=self.kl_form == "vmae")
                kl = posterior.kl()
                kl_loss = torch.sum(kl) / kl.shape[0] / N
                latent = (posterior.sample() if _eps is None else posterior.mean + posterior.std * _eps).permute(0, 2, 1)
            latent = _latent_map(self.from_latent, latent.contiguous())
            pe = self.patch_embed.patch_size[0]
            if isinstance(self.decoder_pred, conv_decoder_pred) and not self.norm_pix_loss and pe % 4 == 0 and imgs.shape[1] == 3:
                # the loss straight from the smoothing conv's output image (no patchify of target or prediction, one kernel each way)
                pred, pimg = self.forward_decoder(latent, ids_restore, dtype, return_image=True)
                mask_loss, vis_loss = _MaeLossFn.apply(pimg, imgs.float().contiguous(), mask, pe)
                loss = (1 - visible_loss_ratio) * mask_loss + visible_loss_ratio * vis_loss
            else:
                pred = self.forward_decoder(latent, ids_restore, dtype)
                loss, vis_loss, mask_loss = self.forward_loss(imgs, pred, mask, visible_loss_ratio)
            if kl_loss is not None:
                loss = loss + self.kl_loss_weight * kl_loss
        return loss, pred, mask, vis_loss, mask_loss, kl_loss

    # ---- docking functions (:817-973)
    def _encode(self, x):
        dtype, tf32 = self._docking_dtype(self.blocks, rows=x.shape[0] * self.patch_embed.num_patches)
        with torch.autocast(device_type="cuda", enabled=False):
            x = self._embed(x, dtype)
            if (self.fused_encoder and dtype in (torch.bfloat16, torch.float16) and not torch.is_grad_enabled() and x.dim() == 3 and
                    fused_encoder.supported(self, x.shape[1], x.shape[2], tiled=True) and all(self._chain_ok(blk) for blk in self.blocks)):
                # bf16 / TF32-class inference on the shipped geometry: per block three launches instead of seven, the residual stream of a 256-token
                # tile in registers through proj / LayerNorm / MLP (csrc/vmae_fused.hip, MODE 1 / 2), the closing LayerNorm in the last of them
                x = fused_encoder.encoder_forward_tiled(self, x.float(), dtype=dtype)
            else:
                x = self._run(self.blocks, x, dtype, tf32)
                x = _LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps)
            x = _latent_map(self.to_latent, x)
        g = self.latent_resolution
        return x.reshape(x.shape[0], g, g, -1).permute(0, 3, 1, 2)

    def encode(self, x, return_dict=True):
        m = self._encode(x)
        p = DiagonalGaussianDistribution(m) if self.kl_loss_weight is not None else EncoderOutput(m)
        return MAEOutput(latent_dist=p) if return_dict else (p,)

    def decode(self, z, return_dict=True, generator=None):
        dtype, tf32 = self._docking_dtype(self.decoder_blocks, rows=z.shape[0] * z.shape[2] * z.shape[3])
        with torch.autocast(device_type="cuda", enabled=False):
            B = z.shape[0]
            x = z.float().permute(0, 2, 3, 1).reshape(B, -1, z.shape[1]).contiguous()
            x = _latent_map(self.from_latent, x)
            x = _LinearFn.apply(x, self.decoder_embed.weight, self.decoder_embed.bias) + self.decoder_pos_embed
            if (self.fused_encoder and dtype in (torch.bfloat16, torch.float16) and not torch.is_grad_enabled() and
                    fused_encoder.supported(self, x.shape[1], x.shape[2], tiled=True, which="dec") and
                    all(self._chain_ok(blk) for blk in self.decoder_blocks)):
                # the shipped decoder has the encoder's geometry (192 wide, 12 heads): bf16 / TF32-class inference runs it on the tiled fused kernels too
                x = fused_encoder.encoder_forward_tiled(self, x.float().contiguous(), which="dec", dtype=dtype)
            else:
                x = self._run(self.decoder_blocks, x, dtype, tf32)
                x = _LayerNormFn.apply(x, self.decoder_norm.weight, self.decoder_norm.bias, self.decoder_norm.eps)
            x = self.decoder_pred(x) if not isinstance(self.decoder_pred, nn.Linear) else \
                _linear(x, self.decoder_pred.weight, self.decoder_pred.bias)
            img = self.unpatchify(x)
        return DecoderOutput(sample=img) if return_dict else (img,)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def img_transform(self, p_hflip=0, img_size=None):
        """:935-950 -- PIL image -> normalised CHW tensor (extract_features.py:105-108 builds its two ImageFolders with it)."""
        return ImageTransform(p_hflip, img_size if img_size is not None else self.img_size)

    # ---- the token-level docking functions of the reference (:625-703: what evaluate_tokenizer.py and the probing heads call), on the kernels above
    def ldmae_encoding(self, imgs, use_mode=False, return_kl=False):
        """:625-654 (no cls token): all patches -> encoder -> to_latent -> posterior mode / sample, as TOKENS [B, N, latent]."""
        mom = self._encode(imgs)                                             # [B, (2) latent, g, g]
        B, C = mom.shape[0], mom.shape[1]
        latent = mom.reshape(B, C, -1)                                       # B D HW
        posterior = None
        if self.kl_loss_weight is not None:
            posterior = DiagonalGaussianDistribution(latent)
            latent = posterior.mode() if use_mode else posterior.sample()
        latent = latent.permute(0, 2, 1)
        if return_kl:
            return latent, posterior.kl()
        return latent

    def ldmae_decoding(self, x):
        """:656-691: latent tokens [B, N, latent] -> from_latent -> decoder -> prediction head, as patch tokens [B, N, p * p * 3] (unpatchify is the caller's)."""
        g = self.latent_resolution
        z = x.permute(0, 2, 1).reshape(x.shape[0], x.shape[2], g, g)
        return self.patchify(self.decode(z, return_dict=False)[0])

    def reconstruct(self, imgs, use_mode=True, return_kl=False):
        """:693-703."""
        with torch.no_grad():
            enc = self.ldmae_encoding(imgs, use_mode=use_mode, return_kl=return_kl)
        x, kl_val = enc if return_kl else (enc, None)
        x = self.ldmae_decoding(x)
        return (x, kl_val) if return_kl else x

    def forward_vanilla(self, imgs, mask_ratio=0.75, visible_loss_ratio=0.5):
        """:756-790 -- the training forward under its reference name."""
        return self.forward(imgs, mask_ratio=mask_ratio, visible_loss_ratio=visible_loss_ratio)

    def encode_images(self, images):
        with torch.no_grad():
            return self.encode(images.cuda(), return_dict=False)[0].sample()

    def decode_to_images(self, z):
        """:963-973 -> uint8 NHWC numpy."""
        with torch.no_grad():
            images = self.decode(z.cuda(), return_dict=False)[0]
            # clamp, convert to uint8 and go to NHWC ON THE DEVICE, then copy 1 byte per value (the reference's .to("cpu", dtype=uint8) moves
            # the f32 image and converts / permutes on the host: 30 of 36 ms per 64 images); same truncation toward zero for values in [0, 255]
            img8 = torch.clamp(127.5 * images.float() + 128.0, 0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            return img8.cpu().numpy()


def _ln(**kw):
    return partial(nn.LayerNorm, eps=1e-6)


# registry (:977-1083): the archs built on the 192/384-wide ViT used by LDMAE
def mae_for_ldmae(**kwargs):
    return MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=12, num_heads=12, decoder_embed_dim=192, decoder_depth=12,
                                decoder_num_heads=12, mlp_ratio=4, norm_layer=_ln(), latent_dim=32, **kwargs)


mae_for_ldmae_f8d32 = mae_for_ldmae


def mae_for_ldmae_f8d16_prev(**kwargs):
    return MaskedAutoencoderViT(patch_size=8, embed_dim=192, depth=12, num_heads=12, decoder_embed_dim=192, decoder_depth=12,
                                decoder_num_heads=12, mlp_ratio=4, norm_layer=_ln(), latent_dim=16, **kwargs)


def mae_for_ldmae_f8d16_small(**kwargs):
    """VMAE/models_mae.py:1036-1041 (the pre-training tree's registry only): 96 wide, 8 heads of 12 -- zero-padded to the head-dim-16 kernels."""
    return MaskedAutoencoderViT(patch_size=8, embed_dim=96, depth=12, num_heads=8, decoder_embed_dim=96, decoder_depth=12,
                                decoder_num_heads=8, mlp_ratio=4, norm_layer=_ln(), latent_dim=16, **kwargs)


def mae_for_ldmae_f8d16_asym_small(**kwargs):
    """VMAE/models_mae.py:1043-1048 (the pre-training tree's registry only): 96-wide encoder (8 heads of 12), the shipped 192-wide decoder."""
    return MaskedAutoencoderViT(patch_size=8, embed_dim=96, depth=12, num_heads=8, decoder_embed_dim=192, decoder_depth=12,
                                decoder_num_heads=12, mlp_ratio=4, norm_layer=_ln(), latent_dim=16, **kwargs)


def mae_for_ldmae_f8d16_prev_large(**kwargs):
    return MaskedAutoencoderViT(patch_size=8, embed_dim=384, depth=12, num_heads=16, decoder_embed_dim=384, decoder_depth=12,
                                decoder_num_heads=16, mlp_ratio=4, norm_layer=_ln(), latent_dim=16, **kwargs)


def mae_for_ldmae_f8d32_flexible(**kwargs):
    return MaskedAutoencoderViT(patch_size=8, embed_dim=192, depth=12, num_heads=12, decoder_embed_dim=192, decoder_depth=12,
                                decoder_num_heads=12, mlp_ratio=4, norm_layer=_ln(), latent_dim=32, **kwargs)


def mae_for_ldmae_16d(**kwargs):
    return MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=12, num_heads=12, decoder_embed_dim=192, decoder_depth=12,
                                decoder_num_heads=12, mlp_ratio=4, norm_layer=_ln(), latent_dim=16, **kwargs)


def mae_for_ldmae_f16d32(**kwargs):
    return MaskedAutoencoderViT(img_size=128, patch_size=16, embed_dim=192, depth=12, num_heads=12, decoder_embed_dim=192, decoder_depth=12,
                                decoder_num_heads=12, mlp_ratio=4, norm_layer=_ln(), latent_dim=32, **kwargs)


# The rest of the reference's registry (:1006-1083), as thin entries over the same class: geometries only.  The f8d16 / f8d16_flexible archs
# (`down_nonlinear`: MLP_dim_resize latent maps; 384-wide decoder with 24 heads of 16) run on the per-layer kernels.
def mae_for_ldmae_f8d16(**kwargs):
    return MaskedAutoencoderViT(patch_size=8, embed_dim=192, depth=12, num_heads=12, decoder_embed_dim=384, decoder_depth=12,
                                decoder_num_heads=24, mlp_ratio=4, norm_layer=_ln(), latent_dim=16, down_nonlinear=True, **kwargs)


mae_for_ldmae_f8d16_flexible = mae_for_ldmae_f8d16


def mae_for_ldmae_f16d32_large(**kwargs):
    # finetune_downsample_layer only acts with gradual_resol (:351), which is not implemented (and raises when asked for)
    return MaskedAutoencoderViT(img_size=128, patch_size=16, embed_dim=384, depth=12, num_heads=12, decoder_embed_dim=384, decoder_depth=12,
                                decoder_num_heads=12, mlp_ratio=4, norm_layer=_ln(), latent_dim=32, finetune_downsample_layer=4, **kwargs)


def mae_vit_base_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=8,
                                decoder_num_heads=16, mlp_ratio=4, norm_layer=_ln(), **kwargs)


def mae_vit_base_patch16_dec128d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=128, decoder_depth=8,
                                decoder_num_heads=16, mlp_ratio=4, norm_layer=_ln(), **kwargs)


def mae_vit_large_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512, decoder_depth=8,
                                decoder_num_heads=16, mlp_ratio=4, norm_layer=_ln(), **kwargs)


def mae_vit_huge_patch14_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=14, embed_dim=1280, depth=32, num_heads=16, decoder_embed_dim=512, decoder_depth=8,
                                decoder_num_heads=16, mlp_ratio=4, norm_layer=_ln(), **kwargs)


mae_vit_base_patch16 = mae_vit_base_patch16_dec512d8b
mae_vit_large_patch16 = mae_vit_large_patch16_dec512d8b
mae_vit_huge_patch14 = mae_vit_huge_patch14_dec512d8b
mae_vit_base_patch16_128 = mae_vit_base_patch16_dec128d8b
