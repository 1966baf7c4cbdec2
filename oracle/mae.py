"""Oracle (test infrastructure): VMAE masked-token encoder (and decoder) restated.

Integer parts (random_masking) are numpy with a *stable* sort; float parts are
fp32 torch.  Citations relative to /root/reference/LDMAE/tokenizer/models_mae.py.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class MAEConfig:
    """mae_for_ldmae_f8d16_prev (models_mae.py:992-997) + call-site kwargs
    (LDMAE/inference.py:133-137)."""
    img_size: int = 256
    patch_size: int = 8
    in_chans: int = 3
    embed_dim: int = 192
    depth: int = 12
    num_heads: int = 12
    decoder_embed_dim: int = 192
    decoder_depth: int = 12
    decoder_num_heads: int = 12
    mlp_ratio: float = 4.0
    latent_dim: int = 16
    kl: bool = True            # kl_loss_weight is not None -> encoder predicts mean & logvar (:300-303)
    ldmae_mode: bool = False   # False -> decoder owns a learnable mask_token (:380-381)
    ln_eps: float = 1e-6
    down_nonlinear: bool = False   # to_latent / from_latent = MLP_dim_resize (Linear -> GELU -> Linear, hidden 4*latent; models_mae.py:232-242, 311-314)

    @property
    def grid(self):
        return self.img_size // self.patch_size

    @property
    def num_patches(self):
        return self.grid * self.grid


def sincos_pos_embed_2d_f32(embed_dim: int, grid_size: int) -> np.ndarray:
    """tokenizer/util/pos_embed.py:20-67 -- same as the DiT table but omega is float32."""
    coords = np.arange(grid_size, dtype=np.float32)
    gw, gh = np.meshgrid(coords, coords)

    def one_axis(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float32)
        omega /= dim / 2.0
        omega = 1.0 / 10000 ** omega
        ang = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)

    return np.concatenate([one_axis(embed_dim // 2, gw), one_axis(embed_dim // 2, gh)], axis=1)


def len_keep(L: int, mask_ratio: float) -> int:
    """models_mae.py:479."""
    return int(L * (1 - mask_ratio))


def random_masking_ids(noise: np.ndarray, mask_ratio: float):
    """models_mae.py:481-495 on a given noise[N, L] (fp32).

    Returns (ids_keep i64 [N, keep], mask f32 [N, L] in {0,1}, ids_restore i64 [N, L]).
    Ties in ``noise`` are broken by index (stable sort) -- the documented contract
    of the HIP kernel; torch.argsort on tie-free input gives the same result.
    """
    N, L = noise.shape
    keep = len_keep(L, mask_ratio)
    ids_shuffle = np.argsort(noise, axis=1, kind="stable")
    ids_restore = np.argsort(ids_shuffle, axis=1, kind="stable")
    mask = np.ones((N, L), dtype=np.float32)
    mask[:, :keep] = 0
    mask = np.take_along_axis(mask, ids_restore, axis=1)
    return ids_shuffle[:, :keep].astype(np.int64), mask, ids_restore.astype(np.int64)


def random_masking(x, noise, mask_ratio):
    """models_mae.py:472-497: gather the kept tokens."""
    ids_keep, mask, ids_restore = random_masking_ids(noise.numpy(), mask_ratio)
    ids_keep = torch.from_numpy(ids_keep)
    xm = torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, x.shape[-1]))
    return xm, torch.from_numpy(mask), torch.from_numpy(ids_restore)


def mae_attention(sd, pre, x, num_heads):
    """models_mae.py:130-147: softmax((q k^T) * hd^-0.5) v, then proj."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def mae_block(sd, pre, x, num_heads, eps=1e-6):
    """models_mae.py:176-187 (LayerScale / DropPath are Identity in shipped archs);
    timm Mlp = fc1 -> exact GELU -> fc2."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps)
    x = x + mae_attention(sd, pre + "attn.", h, num_heads)
    h = F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps)
    h = F.linear(F.gelu(F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])),
                 sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return x + h


def patch_embed(sd, imgs, cfg: MAEConfig):
    """timm PatchEmbed (Conv2d k=s=8, flatten, transpose) + pos_embed (models_mae.py:501-505, no_cls)."""
    h = F.conv2d(imgs, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=cfg.patch_size)
    return h.flatten(2).transpose(1, 2) + sd["pos_embed"]


def forward_encoder(sd, imgs, noise, mask_ratio, cfg: MAEConfig):
    """models_mae.py:499-523 with the uniform noise supplied by the caller."""
    x = patch_embed(sd, imgs, cfg)
    x, mask, ids_restore = random_masking(x, noise, mask_ratio)
    for i in range(cfg.depth):
        x = mae_block(sd, f"blocks.{i}.", x, cfg.num_heads, cfg.ln_eps)
    x = F.layer_norm(x, (cfg.embed_dim,), sd["norm.weight"], sd["norm.bias"], cfg.ln_eps)
    return x, mask, ids_restore


def latent_map(sd, x, name):
    """to_latent / from_latent: a Linear (models_mae.py:316-317) or, with down_nonlinear, MLP_dim_resize (:232-242: Linear -> exact GELU -> Linear)."""
    if name + ".weight" in sd:
        return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])
    h = F.gelu(F.linear(x, sd[name + ".layers.0.weight"], sd[name + ".layers.0.bias"]))
    return F.linear(h, sd[name + ".layers.2.weight"], sd[name + ".layers.2.bias"])


def encode_moments(sd, imgs, cfg: MAEConfig):
    """models_mae.py:819-838 (_encode): full-sequence encoder -> to_latent -> [B, 2*latent, h, w]."""
    x = patch_embed(sd, imgs, cfg)
    for i in range(cfg.depth):
        x = mae_block(sd, f"blocks.{i}.", x, cfg.num_heads, cfg.ln_eps)
    x = F.layer_norm(x, (cfg.embed_dim,), sd["norm.weight"], sd["norm.bias"], cfg.ln_eps)
    x = latent_map(sd, x, "to_latent")
    B = x.shape[0]
    return x.reshape(B, cfg.grid, cfg.grid, -1).permute(0, 3, 1, 2)


def decoder_pred(sd, x, cfg: MAEConfig):
    """conv_decoder_pred with pred_with_conv=False (models_mae.py:257-281):
    linear -> unpatchify -> 3x3 conv on RGB -> patchify."""
    p, g = cfg.patch_size, cfg.grid
    x = F.linear(x, sd["decoder_pred.linear_pred.weight"], sd["decoder_pred.linear_pred.bias"])
    x = x.reshape(x.shape[0], g, g, p, p, 3)
    x = torch.einsum("nhwpqc->nchpwq", x).reshape(x.shape[0], 3, g * p, g * p)
    x = F.conv2d(x, sd["decoder_pred.conv_smoother.weight"], sd["decoder_pred.conv_smoother.bias"], padding=1)
    x = x.reshape(x.shape[0], 3, g, p, g, p)
    return torch.einsum("nchpwq->nhwpqc", x).reshape(x.shape[0], g * g, p * p * 3)


def decode(sd, z, cfg: MAEConfig):
    """models_mae.py:865-887: latent [B, latent, h, w] -> image [B, 3, H, W]."""
    B = z.shape[0]
    x = z.permute(0, 2, 3, 1).reshape(B, cfg.num_patches, -1)
    x = latent_map(sd, x, "from_latent")
    x = F.linear(x, sd["decoder_embed.weight"], sd["decoder_embed.bias"]) + sd["decoder_pos_embed"]
    for i in range(cfg.decoder_depth):
        x = mae_block(sd, f"decoder_blocks.{i}.", x, cfg.decoder_num_heads, cfg.ln_eps)
    x = F.layer_norm(x, (cfg.decoder_embed_dim,), sd["decoder_norm.weight"], sd["decoder_norm.bias"], cfg.ln_eps)
    x = decoder_pred(sd, x, cfg)
    p, g = cfg.patch_size, cfg.grid
    x = x.reshape(B, g, g, p, p, 3)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(B, 3, g * p, g * p)


def patchify(imgs, cfg: MAEConfig):
    """models_mae.py:449-460: [N,3,H,W] -> [N, L, p*p*3] in (p, q, c) order."""
    p, g = cfg.patch_size, cfg.grid
    x = imgs.reshape(imgs.shape[0], 3, g, p, g, p)
    return torch.einsum("nchpwq->nhwpqc", x).reshape(imgs.shape[0], g * g, p * p * 3)


def posterior_kl(mean, logvar, kl_form="tokenizer", fixed_std=None):
    """DiagonalGaussianDistribution.kl() per sample (sum over every other dim), the two trees of the reference:
    "tokenizer" = LDMAE/tokenizer/util/misc.py:102-107: 0.5 sum(mean^2 + var - 1 - logvar);
    "vmae" = the PRE-TRAINING tree, VMAE/util/misc.py:103-125: with fixed_std 0.5 sum(var / s^2 - 1 - logvar + log s^2) (:105-116),
    without it the variance-only form 0.5 sum(var - 1 - logvar) (:118-125: the mean^2 line is commented out there)."""
    var = torch.exp(logvar)
    if kl_form == "tokenizer":
        assert fixed_std is None
        return 0.5 * torch.sum(mean ** 2 + var - 1.0 - logvar, dim=[1, 2])
    if fixed_std is not None:
        fv = torch.tensor(fixed_std) ** 2
        return 0.5 * torch.sum(var / fv - 1.0 - logvar + torch.log(fv), dim=[1, 2])
    return 0.5 * torch.sum(var - 1.0 - logvar, dim=[1, 2])


def forward_vanilla(sd, imgs, noise, eps, mask_ratio, visible_loss_ratio, kl_loss_weight, cfg: MAEConfig, kl_form="tokenizer", fixed_std=None):
    """MaskedAutoencoderViT.forward_vanilla + forward_decoder + forward_loss (models_mae.py:756-790, 525-554, 733-754) with the two
    random draws supplied by the caller: `noise` [B, L] (random_masking, :480) and `eps` [B, latent, kept] (posterior.sample,
    util/misc.py:87-96).  kl_form / fixed_std: which tree's KL (posterior_kl; VMAE/models_mae.py:773-807 is otherwise the same function).
    Returns (loss, pred, mask, vis_loss, mask_loss, kl_loss)."""
    latent, mask, ids_restore = forward_encoder(sd, imgs, noise, mask_ratio, cfg)
    latent = latent_map(sd, latent, "to_latent")
    B, N, D = latent.shape
    mom = latent.permute(0, 2, 1)                                     # B D HW
    mean, logvar = torch.chunk(mom, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    kl = posterior_kl(mean, logvar, kl_form, fixed_std)
    kl_loss = torch.sum(kl) / kl.shape[0] / N
    latent = (mean + torch.exp(0.5 * logvar) * eps).permute(0, 2, 1)
    x = latent_map(sd, latent, "from_latent")
    x = F.linear(x, sd["decoder_embed.weight"], sd["decoder_embed.bias"])
    mask_tokens = sd["mask_token"].repeat(x.shape[0], ids_restore.shape[1] - x.shape[1], 1)
    x_ = torch.cat([x, mask_tokens], dim=1)
    x = torch.gather(x_, 1, ids_restore.unsqueeze(-1).repeat(1, 1, x.shape[2])) + sd["decoder_pos_embed"]
    for i in range(cfg.decoder_depth):
        x = mae_block(sd, f"decoder_blocks.{i}.", x, cfg.decoder_num_heads, cfg.ln_eps)
    x = F.layer_norm(x, (cfg.decoder_embed_dim,), sd["decoder_norm.weight"], sd["decoder_norm.bias"], cfg.ln_eps)
    pred = decoder_pred(sd, x, cfg)
    per_patch = ((pred - patchify(imgs, cfg)) ** 2).mean(dim=-1)
    vis_loss = (per_patch * (1 - mask)).sum() / (1 - mask).sum()
    mask_loss = (per_patch * mask).sum() / mask.sum()
    loss = (1 - visible_loss_ratio) * mask_loss + visible_loss_ratio * vis_loss + kl_loss_weight * kl_loss
    return loss, pred, mask, vis_loss, mask_loss, kl_loss


def to_uint8_images(img):
    """models_mae.py:970-972 (decode_to_images tail)."""
    return torch.clamp(127.5 * img + 128.0, 0, 255).permute(0, 2, 3, 1).to(torch.uint8).numpy()


def param_shapes(cfg: MAEConfig, with_decoder: bool = True) -> dict:
    D, Dd, p = cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_size
    Hm = int(D * cfg.mlp_ratio)
    enc_lat = cfg.latent_dim * (2 if cfg.kl else 1)
    s = {"patch_embed.proj.weight": (D, cfg.in_chans, p, p), "patch_embed.proj.bias": (D,)}

    def blk(pre, d, hm):
        return {pre + "norm1.weight": (d,), pre + "norm1.bias": (d,),
                pre + "attn.qkv.weight": (3 * d, d), pre + "attn.qkv.bias": (3 * d,),
                pre + "attn.proj.weight": (d, d), pre + "attn.proj.bias": (d,),
                pre + "norm2.weight": (d,), pre + "norm2.bias": (d,),
                pre + "mlp.fc1.weight": (hm, d), pre + "mlp.fc1.bias": (hm,),
                pre + "mlp.fc2.weight": (d, hm), pre + "mlp.fc2.bias": (d,)}
    for i in range(cfg.depth):
        s.update(blk(f"blocks.{i}.", D, Hm))
    s.update({"norm.weight": (D,), "norm.bias": (D,)})
    hid = 4 * cfg.latent_dim
    if cfg.down_nonlinear:              # MLP_dim_resize(embed_dim, 4 latent, enc_lat) / (latent, 4 latent, EMBED_dim) (:313-314)
        s.update({"to_latent.layers.0.weight": (hid, D), "to_latent.layers.0.bias": (hid,),
                  "to_latent.layers.2.weight": (enc_lat, hid), "to_latent.layers.2.bias": (enc_lat,)})
    else:
        s.update({"to_latent.weight": (enc_lat, D), "to_latent.bias": (enc_lat,)})
    if with_decoder:
        if cfg.down_nonlinear:
            s.update({"from_latent.layers.0.weight": (hid, cfg.latent_dim), "from_latent.layers.0.bias": (hid,),
                      "from_latent.layers.2.weight": (D, hid), "from_latent.layers.2.bias": (D,)})
        else:
            # VMAE/models_mae.py:320: back to the ENCODER width (decoder_embed, :371, then maps it to the decoder's); the tokenizer copy's
            # decoder_embed_dim (LDMAE/tokenizer/models_mae.py:317) is the same shape wherever that copy runs at all (equal widths)
            s.update({"from_latent.weight": (D, cfg.latent_dim), "from_latent.bias": (D,)})
        s.update({"decoder_embed.weight": (Dd, D), "decoder_embed.bias": (Dd,)})
        if not cfg.ldmae_mode:
            s["mask_token"] = (1, 1, Dd)
        for i in range(cfg.decoder_depth):
            s.update(blk(f"decoder_blocks.{i}.", Dd, int(Dd * cfg.mlp_ratio)))
        s.update({"decoder_norm.weight": (Dd,), "decoder_norm.bias": (Dd,),
                  "decoder_pred.linear_pred.weight": (p * p * cfg.in_chans, Dd),
                  "decoder_pred.linear_pred.bias": (p * p * cfg.in_chans,),
                  "decoder_pred.conv_smoother.weight": (cfg.in_chans, cfg.in_chans, 3, 3),
                  "decoder_pred.conv_smoother.bias": (cfg.in_chans,)})
    return s


def fixed_tables(cfg: MAEConfig) -> dict:
    pe = torch.from_numpy(sincos_pos_embed_2d_f32(cfg.embed_dim, cfg.grid)).float().unsqueeze(0)
    dpe = torch.from_numpy(sincos_pos_embed_2d_f32(cfg.decoder_embed_dim, cfg.grid)).float().unsqueeze(0)
    return {"pos_embed": pe, "decoder_pos_embed": dpe}
