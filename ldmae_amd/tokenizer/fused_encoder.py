"""Host side of the one-kernel VMAE encoder (csrc/vmae_fused.hip; reference blocks: tokenizer/models_mae.py:149-187, 499-523).

Packs the encoder's weights into the byte stream the kernel's LDS ring consumes, slot after slot: per block 36 steps of
[panel 0: 12 KiB][panel 1: 12 KiB][f32 vectors: 2 KiB]; a panel is 12 ready-made 1-KiB MFMA A-operand fragments (lane L = (h, row)
holds W[32 rb + row][16 ks + PERM16[8 h .. 8 h + 7]]), PERM16 being the order in which a 32x32 C/D block hands its rows on as the
k-index of the next product (the kernel feeds LayerNorm / GELU / attention outputs to the next MFMA straight from the accumulator).

steps 2hp, 2hp+1 (head pair hp = 0..5):  [Wqkv rows of q_hp | of k_hp]  then  [Wqkv rows of v_hp | Wproj columns 32hp..32hp+31]
steps 12 + c (hidden chunk c = 0..23):   [fc1 rows 32c..32c+31 | fc2 columns 32c..32c+31]
vectors (512 floats): [0:192] row vector #1, [192:384] #2, [384:512] misc -- see `_pack_block`.
"""
from __future__ import annotations

import weakref

import torch

from ldmae_amd import ops

PERM16 = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]
TOKENS, DIM, HEADS, HIDDEN, STEPS = 256, 192, 12, 768, 36


_PACK_DTYPE = torch.bfloat16        # set by pack_encoder for the duration of one packing


def _frags(w: torch.Tensor) -> torch.Tensor:
    """[R, K] (R % 32 == 0, K % 16 == 0) -> [R/32, K/16, 512] operand fragments (bf16, or fp16 for the TF32-class blob)."""
    R, K = w.shape
    f = (w.clamp(-65504.0, 65504.0) if _PACK_DTYPE == torch.float16 else w).to(_PACK_DTYPE).reshape(R // 32, 32, K // 16, 16)[..., PERM16]
    return f.reshape(R // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(R // 32, K // 16, 512).contiguous()


def _vec(dev, v1=None, v2=None, misc=()):
    v = torch.zeros(512, dtype=torch.float32, device=dev)
    if v1 is not None:
        v[0:192] = v1
    if v2 is not None:
        v[192:384] = v2
    o = 384
    for m in misc:
        v[o:o + m.numel()] = m
        o += m.numel()
    return v


def _pack_block(blk, final_norm=None):
    """-> list of 36 (panel0 [12,512] bf16, panel1 [12,512] bf16, vec [512] f32)."""
    a, m = blk.attn, blk.mlp
    dev = a.qkv.weight.device
    fq = _frags(a.qkv.weight.detach())                 # [18, 12, 512]: row blocks 0-5 q, 6-11 k, 12-17 v
    f1 = _frags(m.fc1.weight.detach())                 # [24, 12, 512]
    bq, b1 = a.qkv.bias.detach().float(), m.fc1.bias.detach().float()
    steps = []
    for hp in range(6):
        fp = _frags(a.proj.weight.detach()[:, 32 * hp:32 * hp + 32]).reshape(12, 512)       # (feature block, k-step)
        steps.append((fq[hp], fq[6 + hp],
                      _vec(dev, blk.norm1.weight.detach() if hp == 0 else None, blk.norm1.bias.detach() if hp == 0 else None,
                           (bq[32 * hp:32 * hp + 32], bq[192 + 32 * hp:192 + 32 * hp + 32]))))
        steps.append((fq[12 + hp], fp, _vec(dev, a.proj.bias.detach() if hp == 5 else None, None, (bq[384 + 32 * hp:384 + 32 * hp + 32],))))
    for c in range(24):
        f2 = _frags(m.fc2.weight.detach()[:, 32 * c:32 * c + 32]).reshape(12, 512)
        v1 = v2 = None
        if c == 0:
            v1, v2 = blk.norm2.weight.detach(), blk.norm2.bias.detach()
        elif c == 23:
            v1 = m.fc2.bias.detach()
        elif c == 22 and final_norm is not None:        # read after the loop, from the second-to-last slot
            v1, v2 = final_norm.weight.detach(), final_norm.bias.detach()
        steps.append((f1[c], f2, _vec(dev, v1, v2, (b1[32 * c:32 * c + 32],))))
    return steps


def pack_encoder(blocks, final_norm, dtype=torch.bfloat16) -> torch.Tensor:
    global _PACK_DTYPE
    parts = []
    _PACK_DTYPE = dtype
    try:
        for i, blk in enumerate(blocks):
            for p0, p1, v in _pack_block(blk, final_norm if i == len(blocks) - 1 else None):
                parts += [p0.reshape(-1).view(torch.uint8), p1.reshape(-1).view(torch.uint8), v.view(torch.uint8)]
    finally:
        _PACK_DTYPE = torch.bfloat16
    blob = torch.cat(parts)
    assert blob.numel() == len(blocks) * STEPS * (2 * 12 * 1024 + 2048)
    return blob


def _stack(model, which: str):
    """(blocks, closing LayerNorm) of the encoder ("enc") or the decoder ("dec": the shipped tokenizer's decoder has the encoder's geometry)."""
    return (model.blocks, model.norm) if which == "enc" else (model.decoder_blocks, model.decoder_norm)


def supported(model, tokens: int, dim: int, tiled: bool = False, which: str = "enc") -> bool:
    """The fused kernel covers exactly the shipped block geometry on a 256-token sequence (mask_ratio 0.75 of 1024 patches); `tiled`:
    the two-launches-per-block form, any whole number of 256-token tiles (the docking encoder / decoder on all 1024 patches)."""
    blocks, norm = _stack(model, which)
    if len(blocks) == 0:
        return False
    b0 = blocks[0]

    def packable(blk):          # _pack_block reads every bias: a model built with qkv_bias=False (or on another device / type) keeps the per-layer kernels
        ts = (blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight, blk.attn.proj.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
              blk.mlp.fc2.weight, blk.mlp.fc2.bias, blk.norm1.weight, blk.norm1.bias, blk.norm2.weight, blk.norm2.bias)
        return all(t is not None and t.is_cuda and t.dtype == torch.float32 for t in ts)
    return ((tokens % TOKENS == 0 and tokens > 0 if tiled else tokens == TOKENS) and dim == DIM and b0.attn.qkv.weight.shape[1] == DIM and
            b0.attn.num_heads == HEADS and b0.mlp.fc1.weight.shape[0] == HIDDEN and
            isinstance(norm, torch.nn.LayerNorm) and norm.bias is not None and all(packable(blk) for blk in blocks) and
            all(abs(blk.norm1.eps - norm.eps) < 1e-12 and abs(blk.norm2.eps - norm.eps) < 1e-12 for blk in blocks))


_CACHE: dict = {}


def encoder_blob(model, which: str = "enc", dtype=torch.bfloat16) -> torch.Tensor:
    """Packed weights of the stack's blocks + closing LayerNorm, rebuilt when any of those parameters changed (version counters, storage,
    and ops.WEIGHT_EPOCH for writes through the flat optimizer slab)."""
    blocks, norm = _stack(model, which)
    ps = [p for blk in blocks for p in blk.parameters()] + list(norm.parameters())
    stamp = (ops.WEIGHT_EPOCH,) + tuple((p.data_ptr(), p._version) for p in ps)
    hit = _CACHE.get((id(model), which, dtype))
    if hit is not None and hit[0]() is model and hit[1] == stamp:      # the weak reference guards against a recycled id()
        return hit[2]
    for k in [k for k, v in _CACHE.items() if v[0]() is None]:         # models that are gone: drop their 11 MB blobs
        del _CACHE[k]
    blob = pack_encoder(blocks, norm, dtype)
    _CACHE[(id(model), which, dtype)] = (weakref.ref(model), stamp, blob)
    return blob


def encoder_forward(model, x: torch.Tensor) -> torch.Tensor:
    """x [B, 256, 192] f32 (gathered kept tokens) -> LayerNorm(blocks(x)) [B, 256, 192] f32, one launch."""
    return ops.vmae_encoder_fwd(x, encoder_blob(model), len(model.blocks), DIM, HEADS, HIDDEN, model.norm.eps)


def encoder_forward_tiled(model, x: torch.Tensor, which: str = "enc", dtype=torch.bfloat16) -> torch.Tensor:
    """x [B, k * 256, 192] f32 (every patch of the image) -> LayerNorm(blocks(x)), two launches per block from the same blob; `which`:
    the encoder stack or the decoder stack; `dtype`: bf16, or fp16 = the TF32-class form (its own blob)."""
    blocks, norm = _stack(model, which)
    return ops.vmae_encoder_fwd_tiled(x, encoder_blob(model, which, dtype), len(blocks), DIM, HEADS, HIDDEN, norm.eps, f16=dtype == torch.float16)
