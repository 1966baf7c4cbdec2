"""VMAE masked-token encoder (tokenizer/models_mae.py mirror) on the HIP kernels vs the CPU oracle and the goldens
generated from the reference: masks / ids_restore bit-exact, latents within 1e-4 (f32)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import mae as omae
from weights import det_randn, det_weights

pytestmark = pytest.mark.gpu


def build(cfg_kw, sd, img_size):
    from ldmae_amd.tokenizer import models_mae
    m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=False, no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=img_size, **cfg_kw)
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def full_sd(cfg, seed=2):
    sd = det_weights(omae.param_shapes(cfg), seed)
    sd.update(omae.fixed_tables(cfg))
    return sd


def test_forward_encoder_matches_reference_golden(golden):
    g = golden("mae")
    cfg = omae.MAEConfig()
    m = build({}, full_sd(cfg), 256)
    imgs = det_randn("mae_img", (2, 3, 256, 256), 2).clamp(-1, 1).cuda()
    noise = torch.from_numpy(g["mae_noise"]).cuda()
    for tag, ratio in (("75", 0.75), ("25", 0.25)):
        with torch.no_grad():
            lat, mask, ids = m.forward_encoder(imgs, ratio, noise=noise)
        np.testing.assert_array_equal(mask.cpu().numpy(), g[f"mae{tag}_mask"])            # bit-exact
        np.testing.assert_array_equal(ids.cpu().numpy(), g[f"mae{tag}_ids_restore"])       # bit-exact
        assert list(lat.shape) == list(g[f"mae{tag}_lat_shape"])
        assert rel_err(lat[:, :4].cpu(), g[f"mae{tag}_lat_head"]) < 1e-4
        assert abs(float(lat.double().norm()) - float(g[f"mae{tag}_lat_norm"])) < 1e-4 * float(g[f"mae{tag}_lat_norm"])
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            lat16, mask16, _ = m.forward_encoder(imgs, ratio, noise=noise)
        assert torch.equal(mask16, mask) and rel_err(lat16.cpu(), lat.cpu()) < 3e-2
        # ... and the bf16 one-kernel encoder DIRECTLY against the reference's own latents (not only through the HIP f32 result)
        assert rel_err(lat16[:, :4].float().cpu(), g[f"mae{tag}_lat_head"]) < 3e-2
        assert abs(float(lat16.double().norm()) - float(g[f"mae{tag}_lat_norm"])) < 1e-2 * float(g[f"mae{tag}_lat_norm"])


def test_encode_decode_docking_matches_reference_golden(golden):
    g = golden("mae")
    cfg = omae.MAEConfig()
    m = build({}, full_sd(cfg), 256)
    imgs = det_randn("mae_img", (2, 3, 256, 256), 2).clamp(-1, 1).cuda()
    with torch.no_grad():
        mom = m._encode(imgs)
        post = m.encode(imgs).latent_dist
        rec = m.decode(mom[:, :16]).sample
    assert rel_err(mom[:, :, :2, :2].cpu(), g["mae_moments_head"]) < 1e-4
    assert abs(float(mom.double().norm()) - float(g["mae_moments_norm"])) < 1e-4 * float(g["mae_moments_norm"])
    assert torch.equal(post.mean, mom[:, :16])
    assert rel_err(rec[:, :, :4, :4].cpu(), g["mae_rec_head"]) < 1e-4
    assert abs(float(rec.double().norm()) - float(g["mae_rec_norm"])) < 1e-4 * float(g["mae_rec_norm"])
    img8 = m.decode_to_images(mom[:, :16])
    assert img8.dtype == np.uint8 and img8.shape == (2, 256, 256, 3)
    assert (np.abs(img8[:, :4, :4].astype(int) - g["mae_img8_head"].astype(int)) <= 1).all()
    # the token-level docking functions (models_mae.py:625-703: ldmae_encoding -> ldmae_decoding = reconstruct) are the same computation as tokens
    lat, kl = m.ldmae_encoding(imgs, use_mode=True, return_kl=True)
    assert lat.shape == (2, 1024, 16) and torch.equal(lat, mom[:, :16].reshape(2, 16, -1).permute(0, 2, 1)) and kl.shape == (2,)
    rt = m.reconstruct(imgs, use_mode=True)
    assert rt.shape == (2, 1024, 192) and rel_err(m.unpatchify(rt)[:, :, :4, :4].cpu(), g["mae_rec_head"]) < 1e-4
    torch.manual_seed(3)
    a_ = m.ldmae_encoding(imgs)                                   # a posterior SAMPLE: mean + std * noise
    assert a_.shape == lat.shape and not torch.equal(a_, lat) and bool(torch.isfinite(a_).all())
    # the bf16 docking paths (tiled fused encoder, fused decoder stack) DIRECTLY against the reference goldens
    m.set_precision(torch.bfloat16)
    try:
        with torch.no_grad():
            mom16 = m._encode(imgs)
            rec16 = m.decode(mom[:, :16]).sample
    finally:
        m.set_precision(None)
    assert rel_err(mom16[:, :, :2, :2].float().cpu(), g["mae_moments_head"]) < 3e-2
    assert abs(float(mom16.double().norm()) - float(g["mae_moments_norm"])) < 1e-2 * float(g["mae_moments_norm"])
    assert rel_err(rec16[:, :, :4, :4].float().cpu(), g["mae_rec_head"]) < 3e-2
    assert abs(float(rec16.double().norm()) - float(g["mae_rec_norm"])) < 1e-2 * float(g["mae_rec_norm"])


@pytest.mark.parametrize("ratio", [0.75, 0.8, 0.3])
def test_masked_encoder_gradients_vs_oracle(ratio):
    """img 128 -> 256 patches, mask 0.75 -> 64 kept tokens; every encoder parameter gradient vs torch autograd on the oracle.
    Ratios 0.8 / 0.3 keep int(256 * 0.2) = 51 / 179 tokens (models_mae.py:480): ragged attention tiles and GEMM row counts."""
    cfg = omae.MAEConfig(img_size=128, depth=2)
    shapes = omae.param_shapes(cfg)
    sd = full_sd(cfg, seed=3)
    from ldmae_amd.tokenizer import models_mae
    m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192,
                                        decoder_depth=12, decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16,
                                        no_cls=True, kl_loss_weight=1e-6, smooth_output=True)
    own = m.state_dict()
    own.update({k: v for k, v in sd.items() if k in own and own[k].shape == v.shape})
    m.load_state_dict(own)
    m = m.cuda().train()
    imgs = det_randn("img128", (2, 3, 128, 128), 4).clamp(-1, 1)
    noise = torch.rand(2, 256, generator=torch.Generator().manual_seed(9))
    enc_keys = [k for k in shapes if k.startswith(("patch_embed", "blocks.0.", "blocks.1.", "norm."))]
    leaves = {k: own[k].clone().requires_grad_(True) for k in enc_keys}
    osd = dict(own)
    osd.update(leaves)
    olat, omask, oids = omae.forward_encoder(osd, imgs, noise, ratio, cfg)
    w = det_randn("w", tuple(olat.shape), 5)
    (olat * w).sum().backward()
    lat, mask, ids = m.forward_encoder(imgs.cuda(), ratio, noise=noise.cuda())
    assert lat.shape[1] == int(256 * (1 - ratio))
    assert torch.equal(mask.cpu(), omask) and torch.equal(ids.cpu(), oids)
    assert rel_err(lat.detach().cpu(), olat.detach()) < 1e-4
    (lat * w.cuda()).sum().backward()
    params = dict(m.named_parameters())
    for k in enc_keys:
        assert rel_err(params[k].grad.cpu(), leaves[k].grad) < 2e-4, k
    if ratio != 0.75:                  # the same ragged shapes through the bf16 kernels (flash attention on the packed qkv, bf16 TN GEMM)
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lat16, mask16, _ = m.forward_encoder(imgs.cuda(), ratio, noise=noise.cuda())
        assert torch.equal(mask16.cpu(), omask) and rel_err(lat16.detach().float().cpu(), olat.detach()) < 3e-2
        (lat16.float() * w.cuda()).sum().backward()
        for k in ("blocks.0.attn.qkv.weight", "blocks.1.mlp.fc2.weight", "patch_embed.proj.weight"):
            assert rel_err(params[k].grad.cpu(), leaves[k].grad) < 6e-2, k


def test_small_archs_of_the_pretraining_tree_vs_oracle():
    """The two archs only the pre-training tree registers (VMAE/models_mae.py:1036-1048): mae_for_ldmae_f8d16_small (96 wide, 8 heads of 12) and
    _asym_small (that encoder, the 192-wide decoder), at depth 1 / 64 px with the tree's own KL (variance-only): loss and every gradient against
    the oracle in f32, bf16 autocast close to it, the docking calls finite.  Head dim 12 runs the head-dim-16 kernels on zero-padded heads."""
    from ldmae_amd.tokenizer import models_mae
    imgs = det_randn("img64s", (2, 3, 64, 64), 4).clamp(-1, 1)
    noise = torch.rand(2, 64, generator=torch.Generator().manual_seed(5))
    eps = torch.randn(2, 16, 16, generator=torch.Generator().manual_seed(6))
    for dec_dim, dec_heads in ((96, 8), (192, 12)):
        cfg = omae.MAEConfig(img_size=64, embed_dim=96, num_heads=8, depth=1, decoder_embed_dim=dec_dim, decoder_num_heads=dec_heads, decoder_depth=1)
        sd = full_sd(cfg, seed=11)
        m = models_mae.MaskedAutoencoderViT(img_size=64, patch_size=8, embed_dim=96, depth=1, num_heads=8, decoder_embed_dim=dec_dim, decoder_depth=1,
                                            decoder_num_heads=dec_heads, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16, no_cls=True,
                                            kl_loss_weight=1e-3, smooth_output=True)
        m.kl_form = "vmae"
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train()
        keys = [k for k in omae.param_shapes(cfg)]
        leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
        osd = dict(sd)
        osd.update(leaves)
        ol = omae.forward_vanilla(osd, imgs, noise, eps, 0.75, 0.5, 1e-3, cfg, kl_form="vmae")[0]
        ol.backward()
        loss = m(imgs.cuda(), 0.75, 0.5, _noise=noise.cuda(), _eps=eps.cuda())[0]
        assert abs(float(loss) - float(ol)) < 1e-4 * abs(float(ol)), dec_dim
        loss.backward()
        params = dict(m.named_parameters())
        for k in keys:
            assert rel_err(params[k].grad.cpu(), leaves[k].grad) < 2e-4, (dec_dim, k)
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss16 = m(imgs.cuda(), 0.75, 0.5, _noise=noise.cuda(), _eps=eps.cuda())[0]
        loss16.backward()
        assert abs(float(loss16) - float(ol)) < 2e-2 * abs(float(ol)), dec_dim
        with torch.no_grad():
            rec = m.decode(m._encode(imgs.cuda())[:, :16]).sample
        assert rec.shape == (2, 3, 64, 64) and bool(torch.isfinite(rec).all())
    sm, asym = models_mae.mae_for_ldmae_f8d16_small(img_size=64, no_cls=True), models_mae.mae_for_ldmae_f8d16_asym_small(img_size=64, no_cls=True)
    assert sm.pos_embed.shape[-1] // sm.blocks[0].attn.num_heads == 12 and asym.decoder_pos_embed.shape[-1] // asym.decoder_blocks[0].attn.num_heads == 16
    assert asym.from_latent.weight.shape == (96, 16) and asym.decoder_embed.weight.shape == (192, 96)      # VMAE/models_mae.py:320,371


def test_head_dim_24_arch_pretraining_step_vs_oracle(golden):
    """mae_for_ldmae_f8d16_prev_large's geometry (384 wide, 16 heads of 24: a head dim the attention kernels are not instantiated for -- they run
    on zero-padded heads, ops.attention_fwd) at depth 1: loss and every parameter gradient of the pre-training step against the oracle in f32,
    and bf16 autocast close to it."""
    cfg = omae.MAEConfig(img_size=64, embed_dim=384, num_heads=16, depth=1, decoder_embed_dim=384, decoder_num_heads=16, decoder_depth=1)
    sd = full_sd(cfg, seed=8)
    from ldmae_amd.tokenizer import models_mae
    m = models_mae.MaskedAutoencoderViT(img_size=64, patch_size=8, embed_dim=384, depth=1, num_heads=16, decoder_embed_dim=384, decoder_depth=1,
                                        decoder_num_heads=16, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16, no_cls=True,
                                        kl_loss_weight=1e-3, smooth_output=True)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    imgs = det_randn("img64p", (2, 3, 64, 64), 4).clamp(-1, 1)
    g = golden("mae_archs")                                     # the reference's own run of this geometry (make_golden.py: gen_mae_archs)
    noise, eps = torch.from_numpy(g["ar_h24_noise"]), torch.from_numpy(g["ar_h24_eps"])
    keys = [k for k in omae.param_shapes(cfg)]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
    osd = dict(sd)
    osd.update(leaves)
    ol = omae.forward_vanilla(osd, imgs, noise, eps, 0.75, 0.5, 1e-3, cfg)[0]
    ol.backward()
    loss, pred = m(imgs.cuda(), 0.75, 0.5, _noise=noise.cuda(), _eps=eps.cuda())[:2]
    assert abs(float(loss) - float(ol)) < 1e-4 * abs(float(ol))
    assert abs(float(loss) - float(g["ar_h24_loss"][0])) < 1e-4 * abs(float(ol)) and rel_err(pred.detach()[:, :6, :24].cpu(), g["ar_h24_pred_head"]) < 1e-4
    loss.backward()
    params = dict(m.named_parameters())
    for k in keys:
        assert rel_err(params[k].grad.cpu(), leaves[k].grad) < 2e-4, k
    norms = np.array([float(params[k].grad.double().norm()) for k in sorted(keys)])
    np.testing.assert_allclose(norms, g["ar_h24_grad_norms"], rtol=5e-4, atol=1e-8)      # ... and the reference's own gradient norms
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss16 = m(imgs.cuda(), 0.75, 0.5, _noise=noise.cuda(), _eps=eps.cuda())[0]
    loss16.backward()
    assert abs(float(loss16) - float(ol)) < 2e-2 * abs(float(ol))
    assert rel_err(params["blocks.0.attn.qkv.weight"].grad.cpu(), leaves["blocks.0.attn.qkv.weight"].grad) < 8e-2
    with torch.no_grad():                                        # docking calls of the same arch
        rec = m.decode(m._encode(imgs.cuda())[:, :16]).sample
    assert rec.shape == (2, 3, 64, 64) and bool(torch.isfinite(rec).all())


def test_down_nonlinear_wide_decoder_arch_vs_oracle(golden):
    """mae_for_ldmae_f8d16's geometry (models_mae.py:1006-1011: 192-wide encoder, 384-wide decoder with 24 heads of 16, MLP_dim_resize latent
    maps) at depth 1: the pre-training step (loss, every gradient) and the docking encode / decode against the oracle in f32."""
    cfg = omae.MAEConfig(img_size=64, depth=1, decoder_embed_dim=384, decoder_num_heads=24, decoder_depth=1, down_nonlinear=True)
    sd = full_sd(cfg, seed=9)
    from ldmae_amd.tokenizer import models_mae
    m = models_mae.MaskedAutoencoderViT(img_size=64, patch_size=8, embed_dim=192, depth=1, num_heads=12, decoder_embed_dim=384, decoder_depth=1,
                                        decoder_num_heads=24, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16, no_cls=True,
                                        kl_loss_weight=1e-3, smooth_output=True, down_nonlinear=True)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    imgs = det_randn("img64d", (2, 3, 64, 64), 4).clamp(-1, 1)
    g = golden("mae_archs")                                     # the reference's own run of this geometry
    noise, eps = torch.from_numpy(g["ar_dn_noise"]), torch.from_numpy(g["ar_dn_eps"])
    keys = [k for k in omae.param_shapes(cfg)]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
    osd = dict(sd)
    osd.update(leaves)
    ol = omae.forward_vanilla(osd, imgs, noise, eps, 0.75, 0.5, 1e-3, cfg)[0]
    ol.backward()
    loss = m(imgs.cuda(), 0.75, 0.5, _noise=noise.cuda(), _eps=eps.cuda())[0]
    assert abs(float(loss) - float(ol)) < 1e-4 * abs(float(ol))
    loss.backward()
    params = dict(m.named_parameters())
    for k in keys:
        assert rel_err(params[k].grad.cpu(), leaves[k].grad) < 2e-4, k
    with torch.no_grad():
        mom = m.eval()._encode(imgs.cuda())
        rec = m.decode(mom[:, :16]).sample
        omom = omae.encode_moments(sd, imgs, cfg)
        orec = omae.decode(sd, omom[:, :16], cfg)
    assert rel_err(mom.cpu(), omom) < 1e-4 and rel_err(rec.cpu(), orec) < 1e-4
    assert abs(float(loss) - float(g["ar_dn_loss"][0])) < 1e-4 * abs(float(ol))
    norms = np.array([float(params[k].grad.double().norm()) for k in sorted(keys)])
    np.testing.assert_allclose(norms, g["ar_dn_grad_norms"], rtol=5e-4, atol=1e-8)
    assert rel_err(mom[:, :, :2, :2].cpu(), g["ar_dn_moments_head"]) < 1e-4 and rel_err(rec[:, :, :4, :4].cpu(), g["ar_dn_rec_head"]) < 1e-4


def test_patch16_tokenizer_geometry_vs_reference_golden(golden):
    """The patch-16 tokenizers of the registry (mae_for_ldmae_f16d32, models_mae.py:1020-1025) at depth 1 / 64 px: 16 tokens per image (every
    attention tile ragged), 768-element patches through the patch embed, the RGB conv and the image-space loss -- the pre-training step and the
    docking calls against the reference's own run (tests/golden/mae_archs.npz)."""
    g = golden("mae_archs")
    cfg = omae.MAEConfig(img_size=64, patch_size=16, depth=1, decoder_depth=1)
    sd = full_sd(cfg, seed=7)
    from ldmae_amd.tokenizer import models_mae
    m = models_mae.MaskedAutoencoderViT(img_size=64, patch_size=16, embed_dim=192, depth=1, num_heads=12, decoder_embed_dim=192, decoder_depth=1,
                                        decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16, no_cls=True,
                                        kl_loss_weight=1e-3, smooth_output=True)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    imgs = det_randn("img64q", (2, 3, 64, 64), 4).clamp(-1, 1).cuda()
    loss, pred, mask = m(imgs, 0.75, 0.5, _noise=torch.from_numpy(g["ar_p16_noise"]).cuda(), _eps=torch.from_numpy(g["ar_p16_eps"]).cuda())[:3]
    np.testing.assert_array_equal(mask.cpu().numpy(), g["ar_p16_mask"])
    assert abs(float(loss) - float(g["ar_p16_loss"][0])) < 1e-4 * abs(float(g["ar_p16_loss"][0]))
    assert rel_err(pred.detach()[:, :6, :24].cpu(), g["ar_p16_pred_head"]) < 1e-4
    loss.backward()
    params = dict(m.named_parameters())
    keys = [str(k) for k in g["ar_p16_keys"]]
    np.testing.assert_allclose(np.array([float(params[k].grad.double().norm()) for k in keys]), g["ar_p16_grad_norms"], rtol=5e-4, atol=1e-8)
    with torch.no_grad():
        mom = m.eval()._encode(imgs)
        rec = m.decode(mom[:, :16]).sample
    assert rel_err(mom[:, :, :2, :2].cpu(), g["ar_p16_moments_head"]) < 1e-4 and rel_err(rec[:, :, :4, :4].cpu(), g["ar_p16_rec_head"]) < 1e-4


def test_pretraining_step_loss_and_all_grads_vs_oracle():
    """SURVEY 8(f)4 minimal slice: the VMAE pre-training forward (masked encoder -> KL posterior -> decoder with mask tokens and the
    RGB smoothing conv -> masked / visible loss, models_mae.py:733-790) and EVERY parameter gradient against torch autograd on the
    oracle, 128-px images, encoder / decoder depth 2, f32, host-drawn masking noise and posterior noise."""
    cfg = omae.MAEConfig(img_size=128, depth=2, decoder_depth=2)
    sd = full_sd(cfg, seed=6)
    from ldmae_amd.tokenizer import models_mae
    m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192,
                                        decoder_depth=2, decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16,
                                        no_cls=True, kl_loss_weight=1e-3, smooth_output=True)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    imgs = det_randn("img128p", (2, 3, 128, 128), 4).clamp(-1, 1)
    noise = torch.rand(2, 256, generator=torch.Generator().manual_seed(9))
    eps = det_randn("peps", (2, 16, 64), 5)
    keys = [k for k in omae.param_shapes(cfg)]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
    osd = dict(sd)
    osd.update(leaves)
    ol, opred, omask, ovis, omsk, okl = omae.forward_vanilla(osd, imgs, noise, eps, 0.75, 0.5, 1e-3, cfg)
    ol.backward()
    loss, pred, mask, vis, msk, kl = m(imgs.cuda(), 0.75, 0.5, _noise=noise.cuda(), _eps=eps.cuda())
    assert torch.equal(mask.cpu(), omask)
    for a, b in ((loss, ol), (vis, ovis), (msk, omsk), (kl, okl)):
        assert abs(float(a) - float(b)) < 1e-4 * abs(float(b)), (float(a), float(b))
    assert rel_err(pred.detach().cpu(), opred.detach()) < 1e-4
    loss.backward()
    params = dict(m.named_parameters())
    worst = 0.0
    for k in keys:
        e = rel_err(params[k].grad.cpu(), leaves[k].grad)
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
    print("worst grad rel err", worst)
    # the same step under bf16 autocast stays close (conv / loss stay f32)
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss16 = m(imgs.cuda(), 0.75, 0.5, _noise=noise.cuda(), _eps=eps.cuda())[0]
    loss16.backward()
    assert abs(float(loss16) - float(ol)) < 2e-2 * abs(float(ol))
    g, r = params["decoder_pred.conv_smoother.weight"].grad.cpu(), leaves["decoder_pred.conv_smoother.weight"].grad
    assert rel_err(g, r) < 5e-2
    # ... and under fp16 autocast (what the reference's engine_pretrain.py:51-57 trains in): the fp16 kernel family, forward and backward --
    # three more mantissa bits than bf16: every gradient within 1e-2 of the f32 oracle (loss scale 1: nothing overflows at this size)
    from ldmae_amd import _lib
    m.zero_grad(set_to_none=True)
    _lib.launch_counts(reset=True)
    with torch.autocast("cuda", dtype=torch.float16):
        lossh = m(imgs.cuda(), 0.75, 0.5, _noise=noise.cuda(), _eps=eps.cuda())[0]
    lossh.backward()
    c = _lib.launch_counts(reset=True)
    assert c["nt_f16"] > 0 and c["tn_f16"] > 0 and c["attn_f16"] == 2 * (len(m.blocks) + len(m.decoder_blocks)) and c["attn_bf16"] == 0 and c["nt_bf16"] == 0, c
    assert abs(float(lossh) - float(ol)) < 3e-3 * abs(float(ol))
    worst16 = max(rel_err(params[k].grad.cpu(), leaves[k].grad) for k in keys)
    print("fp16 worst grad rel err", worst16)
    assert worst16 < 1e-2, worst16


def test_vmae_pretrain_driver_steps():
    """A few optimizer steps of the pre-training driver (engine_pretrain.py counterpart) on a small geometry: loss finite and falling on a
    repeated batch, weight decay applied to matrices only."""
    import argparse
    from ldmae_amd import vmae_pretrain as vp
    from ldmae_amd.tokenizer import models_mae
    torch.manual_seed(0)
    m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192, decoder_depth=2,
                                        decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16, no_cls=True,
                                        kl_loss_weight=1e-6, smooth_output=True).cuda()
    opt = vp.build_optimizer(m, 1e-3, 0.05)
    args = argparse.Namespace(accum_iter=2, lr=1e-3, min_lr=0.0, warmup_epochs=0, epochs=10, fixed_lr=False, precision="bf16", mask_ratio=0.75,
                              visible_loss_ratio=0.5, print_freq=1000)
    x = torch.rand(8, 3, 128, 128, generator=torch.Generator().manual_seed(1)) * 2 - 1     # 256 patches -> 64 kept tokens (attention tiles: N % 64 == 0)
    loader = [(x, 0)] * 24
    ln_w = m.norm.weight.detach().clone()
    first = vp.train_one_epoch(m, loader[:2], opt, 0, args, log=lambda s: None)
    last = vp.train_one_epoch(m, loader, opt, 1, args, log=lambda s: None)
    assert opt.step_count == 13 and np.isfinite(last["loss"]) and last["loss"] < first["loss"]
    assert 0 < last["lr"] < 1e-3 and torch.isfinite(m.norm.weight).all() and not torch.equal(m.norm.weight, ln_w)
    # under bf16 autocast BOTH stacks run their bf16 kernels: forward() switches autocast off around the decoder, whose blocks must be
    # told the activation type read before that (they once saw "no autocast" and ran the f32 kernels: 250 of 304 ms per step at batch 256)
    assert all(b.last_dtype == torch.bfloat16 and b.precision is None for b in list(m.blocks) + list(m.decoder_blocks))
    with torch.no_grad():
        m(x.cuda())                                # no autocast: everything back on the f32 path
    assert all(b.last_dtype == torch.float32 and b.precision is None for b in list(m.blocks) + list(m.decoder_blocks))


def test_bf16_calls_are_dispatched_to_the_bf16_kernels():
    """Which kernel FAMILY every path of the tokenizer runs, by count (ldmae_launch_counts).  Under bf16 autocast / set_precision(bf16) no
    block GEMM and no attention call may fall to the f32 kernels -- the pre-training decoder once did, silently (304 instead of 79 ms per
    step); the only f32 GEMMs left are the thin latent projections (to_latent / from_latent / decoder_embed / decoder_pred), which are f32
    by policy.  Counted on the shipped geometry with depth 2 + 2 at 128 px."""
    from ldmae_amd import _lib
    from ldmae_amd.tokenizer import models_mae
    torch.manual_seed(0)
    m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192, decoder_depth=2,
                                        decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16, no_cls=True,
                                        kl_loss_weight=1e-6, smooth_output=True).cuda()
    x = torch.rand(8, 3, 128, 128, device="cuda") * 2 - 1
    nblk = len(m.blocks) + len(m.decoder_blocks)

    def counted(fn):
        _lib.launch_counts(reset=True)
        fn()
        torch.cuda.synchronize()
        return _lib.launch_counts(reset=True)

    def train_step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = m(x, mask_ratio=0.75, visible_loss_ratio=0.5)[0]
        loss.backward()
    c = counted(train_step)
    # per block: 4 forward Linears + 4 dX GEMMs (NT), 4 dW GEMMs (TN), attention forward + backward -- all bf16
    assert c["nt_bf16"] >= 8 * nblk and c["tn_bf16"] >= 4 * nblk and c["attn_bf16"] == 2 * nblk and c["attn_f32"] == 0, c
    assert c["nt_f32"] <= 12 and c["tn_f32"] <= 8, c                     # the latent / embedding / prediction Linears only
    m.eval().set_precision(torch.bfloat16)
    with torch.no_grad():
        c = counted(lambda: m._encode(x))            # whole 256-token tiles: the tiled fused kernels (no NT GEMM per block) around the bf16 flash kernel
        assert c["attn_bf16"] == len(m.blocks) and c["attn_f32"] == 0 and c["nt_f32"] <= 2, c
        m.fused_encoder = False
        c = counted(lambda: m._encode(x))            # the per-layer kernels
        assert c["nt_bf16"] >= 4 * len(m.blocks) and c["attn_bf16"] == len(m.blocks) and c["attn_f32"] == 0 and c["nt_f32"] <= 2, c
        m.fused_encoder = True
        z = torch.randn(8, 16, 16, 16, device="cuda")
        c = counted(lambda: m.decode(z))             # the decoder has the encoder's geometry here (as in the shipped tokenizer): tiled fused kernels too
        assert c["attn_bf16"] == len(m.decoder_blocks) and c["attn_f32"] == 0 and c["nt_f32"] <= 4, c
        m.fused_encoder = False
        c = counted(lambda: m.decode(z))
        assert c["nt_bf16"] >= 4 * len(m.decoder_blocks) and c["attn_bf16"] == len(m.decoder_blocks) and c["attn_f32"] == 0 and c["nt_f32"] <= 4, c
        m.fused_encoder = True
    m.set_precision(None)
    with torch.no_grad():
        c = counted(lambda: m._encode(x))                                # no autocast, no precision: the f32 family, and only it
        assert c["nt_bf16"] == 0 and c["attn_bf16"] == 0 and c["attn_f32"] == len(m.blocks), c


def test_fused_encoder_is_independent_of_the_batch_at_full_size():
    """BASELINE config 4 at its full size.  The one-kernel encoder at the bench size: 256 images = one workgroup per CU, every CU busy.  Each
    image's tokens must equal, bit for bit, what the kernel gives for that image alone / in a small batch (no cross-workgroup state, no
    dependence on the grid) -- and images 0 and 255 of the full batch are compared in VALUE with the CPU oracle run on those two images
    (mask / ids_restore bit-exact, latents at the bf16 tolerance)."""
    cfg = omae.MAEConfig()
    sd = full_sd(cfg)
    m = build({}, sd, 256)
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.rand(256, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    noise = torch.rand(256, 1024, device="cuda", generator=g)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        full, mask, ids = m.forward_encoder(x, 0.75, noise=noise)
        assert full.shape == (256, 256, 192) and torch.isfinite(full).all()
        for sl in (slice(0, 1), slice(97, 102), slice(251, 256)):
            part = m.forward_encoder(x[sl], 0.75, noise=noise[sl])[0]
            assert torch.equal(part, full[sl]), sl
    pick = [0, 255]
    olat, omask, oids = omae.forward_encoder(sd, x[pick].cpu(), noise[pick].cpu(), 0.75, cfg)
    np.testing.assert_array_equal(mask[pick].cpu().numpy(), np.asarray(omask))                 # bit-exact
    np.testing.assert_array_equal(ids[pick].cpu().numpy(), np.asarray(oids))                   # bit-exact
    assert rel_err(full[pick].float().cpu(), torch.as_tensor(np.asarray(olat))) < 3e-2


def test_tiled_encoder_is_independent_of_the_batch_at_full_size():
    """The docking encoder (`_encode`: all 1024 patches) in bf16 at the bench size -- 256 images = 1024 tiles of 256 tokens through the
    q|k|v and proj / MLP kernels of csrc/vmae_fused.hip and the flash kernel between them: an image's latent must equal, bit for bit, what
    the same kernels give for that image alone / in a small batch."""
    cfg = omae.MAEConfig()
    sd = full_sd(cfg)
    m = build({}, sd, 256)
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.rand(256, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        full = m._encode(x)
        assert full.shape[0] == 256 and torch.isfinite(full).all()
        for sl in (slice(0, 1), slice(97, 102), slice(251, 256)):
            assert torch.equal(m._encode(x[sl]), full[sl]), sl
    # values at the full batch: the posterior moments of images 0 and 255 against the CPU oracle on those two images
    omom = omae.encode_moments(sd, x[[0, 255]].cpu(), cfg)
    assert rel_err(full[[0, 255]].float().cpu(), torch.as_tensor(np.asarray(omom))) < 3e-2


def test_vmae_direct_param_grads_equal_autograd_accumulation():
    """The pre-training driver's opt-in (set_direct_param_grads: the blocks add their twelve parameter gradients into the slab views
    themselves -- TN GEMM reduce with beta = 1, one multi_add for the vectors) gives the same gradient slab, bit for bit, as autograd's
    AccumulateGrad, over two accumulated micro-steps, bf16."""
    from ldmae_amd import vmae_pretrain as vp
    from ldmae_amd.tokenizer import models_mae
    slabs = {}
    for direct in (False, True):
        torch.manual_seed(0)
        m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192, decoder_depth=2,
                                            decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16, no_cls=True,
                                            kl_loss_weight=1e-6, smooth_output=True).cuda()
        opt = vp.build_optimizer(m, 1e-3, 0.05)
        m.set_direct_param_grads(direct)
        opt.zero_grad()
        g = torch.Generator().manual_seed(3)
        for _ in range(2):
            x = (torch.rand(8, 3, 128, 128, generator=g) * 2 - 1).cuda()
            torch.manual_seed(11)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = m(x, mask_ratio=0.75, visible_loss_ratio=0.5)[0]
            loss.backward()
        slabs[direct] = opt.flat.grads.clone()
        assert all(b.direct_param_grads == direct for b in m.blocks)
    assert float(slabs[True].abs().sum()) > 0 and torch.equal(slabs[True], slabs[False])


def test_vmae_pretrain_bf16_tracks_f32_over_50_steps():
    """Row f4's precision statement, with evidence.  The reference pre-trains under fp16 autocast + GradScaler (VMAE/engine_pretrain.py:51-57,
    util/misc.py:406-435); the kernels here have no fp16 path and run bf16 (8 significant bits against fp16's 11, f32's exponent range).
    What that costs over an optimisation trajectory: 50 optimizer steps of the pre-training driver on the mae_train geometry (128 px,
    depth 2 + 2, batch 8, a fresh batch per step, identical initial weights, masking noise and posterior draws) in bf16 and in f32 --
    the loss curves stay within 2 % of each other at EVERY step (measured: 0.6 % worst step, 0.2 % over the last ten), no step is skipped
    by the loss scaler, and both fall by the same amount."""
    import argparse
    from ldmae_amd import vmae_pretrain as vp
    from ldmae_amd.tokenizer import models_mae
    curves = {}
    for prec in ("fp32", "bf16", "fp16"):
        torch.manual_seed(0)
        m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192, decoder_depth=2,
                                            decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16, no_cls=True,
                                            kl_loss_weight=1e-6, smooth_output=True).cuda()
        opt = vp.build_optimizer(m, 1e-3, 0.05)
        args = argparse.Namespace(accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=0, epochs=100, fixed_lr=True, precision=prec, mask_ratio=0.75,
                                  visible_loss_ratio=0.5, print_freq=1)
        g = torch.Generator().manual_seed(5)
        loader = [(torch.rand(8, 3, 128, 128, generator=g) * 2 - 1, 0) for _ in range(50)]
        scaler = vp.LossScaler(enabled=prec != "fp32")
        losses = []
        torch.manual_seed(123)                       # the device RNG behind the masking noise and the posterior sample
        vp.train_one_epoch(m, loader, opt, 0, args, log=lambda s: losses.append(float(s.split("loss: ")[1].split()[0])), scaler=scaler)
        assert len(losses) == 50 and opt.step_count + scaler.skipped == 50
        assert scaler.skipped == 0 or prec == "fp16", (prec, scaler.skipped)      # only fp16 has a range to overflow
        print(prec, "skipped steps", scaler.skipped, "final loss scale", scaler.scale)
        curves[prec] = np.array(losses)
    f, b, h = curves["fp32"], curves["bf16"], curves["fp16"]
    relh = np.abs(h - f) / f
    print("fp16 vs f32 pre-training loss: worst step", relh.max(), "last ten", abs(h[-10:].mean() - f[-10:].mean()) / f[-10:].mean())
    assert relh.max() < 2e-2 and abs(h[-10:].mean() - f[-10:].mean()) / f[-10:].mean() < 1e-2
    rel = np.abs(b - f) / f
    print("bf16 vs f32 pre-training loss: worst step", rel.max(), "last ten", abs(b[-10:].mean() - f[-10:].mean()) / f[-10:].mean(), "f32 first/last", f[0], f[-1])
    assert f[-10:].mean() < 0.8 * f[:5].mean()                                  # it trains
    assert rel.max() < 2e-2 and abs(b[-10:].mean() - f[-10:].mean()) / f[-10:].mean() < 1e-2


def test_pretraining_forward_and_grads_vs_reference_golden(golden):
    """The product's pre-training forward + backward against the REFERENCE's own MaskedAutoencoderViT.forward outputs and gradient
    norms (tests/golden/mae_train.npz, generated by importing tokenizer/models_mae.py:733-790, 811-815), on the recorded draws."""
    g = golden("mae_train")
    cfg = omae.MAEConfig(img_size=128, depth=2, decoder_depth=2)
    sd = full_sd(cfg, seed=6)
    from ldmae_amd.tokenizer import models_mae
    m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192,
                                        decoder_depth=2, decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16,
                                        no_cls=True, kl_loss_weight=1e-3, smooth_output=True)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    imgs = det_randn("img128p", (2, 3, 128, 128), 4).clamp(-1, 1).cuda()
    names = sorted(k for k, p in m.named_parameters() if p.requires_grad)
    assert names == [str(k) for k in g["mt_keys"]]
    for tag in ("a", "b"):
        ratio, vlr = (float(v) for v in g[f"mt{tag}_ratio"])
        m.zero_grad(set_to_none=True)
        loss, pred, mask, vis, msk, kl = m(imgs, ratio, vlr, _noise=torch.from_numpy(g[f"mt{tag}_noise"]).cuda(),
                                           _eps=torch.from_numpy(g[f"mt{tag}_eps"]).cuda())
        np.testing.assert_array_equal(mask.cpu().numpy(), g[f"mt{tag}_mask"])                       # bit-exact
        np.testing.assert_allclose([float(loss), float(vis), float(msk), float(kl)], g[f"mt{tag}_loss"], rtol=1e-4)
        assert rel_err(pred.detach()[:, :6, :24].cpu(), g[f"mt{tag}_pred_head"]) < 1e-4
        loss.backward()
        params = dict(m.named_parameters())
        norms = np.array([float(params[k].grad.double().norm()) for k in names])
        np.testing.assert_allclose(norms, g[f"mt{tag}_grad_norms"], rtol=3e-4, atol=1e-8)
        assert rel_err(params["decoder_pred.conv_smoother.weight"].grad.cpu(), g[f"mt{tag}_grad_smoother"]) < 2e-4


def test_pretraining_tree_kl_forms_vs_reference_golden(golden):
    """The step VMAE/engine_pretrain.py:51-57 actually runs is the PRE-TRAINING tree's (VMAE/models_mae.py:773-807), whose posterior KL differs from the
    tokenizer tree's: variance-only without `fixed_std` (VMAE/util/misc.py:118-125), against N(mean, fixed_std^2) with it (train_ae.sh:33: 1e-3).  The
    product with `kl_form = "vmae"` / `fixed_std` -- what ldmae_amd/vmae_pretrain.py builds -- against that tree's own outputs and gradient norms
    (tests/golden/vmae_tree.npz, generated from /root/reference/VMAE)."""
    g = golden("vmae_tree")
    cfg = omae.MAEConfig(img_size=128, depth=2, decoder_depth=2)
    sd = full_sd(cfg, seed=6)
    from ldmae_amd.tokenizer import models_mae
    imgs = det_randn("img128p", (2, 3, 128, 128), 4).clamp(-1, 1).cuda()
    for tag in ("n", "f", "h"):
        ratio, vlr, fs = (float(v) for v in g[f"vt{tag}_cfg"])
        m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192,
                                            decoder_depth=2, decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16,
                                            no_cls=True, kl_loss_weight=1e-3, smooth_output=True, fixed_std=None if fs < 0 else fs)
        m.kl_form = "vmae"
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train()
        names = sorted(k for k, p in m.named_parameters() if p.requires_grad)
        assert names == [str(k) for k in g["vt_keys"]]
        loss, pred, mask, vis, msk, kl = m(imgs, ratio, vlr, _noise=torch.from_numpy(g[f"vt{tag}_noise"]).cuda(), _eps=torch.from_numpy(g[f"vt{tag}_eps"]).cuda())
        np.testing.assert_array_equal(mask.cpu().numpy(), g[f"vt{tag}_mask"])
        np.testing.assert_allclose([float(loss), float(vis), float(msk), float(kl)], g[f"vt{tag}_loss"], rtol=1e-4)
        loss.backward()
        params = dict(m.named_parameters())
        norms = np.array([float(params[k].grad.double().norm()) for k in names])
        np.testing.assert_allclose(norms, g[f"vt{tag}_grad_norms"], rtol=3e-4, atol=1e-8)
        assert rel_err(params["to_latent.bias"].grad.cpu(), g[f"vt{tag}_grad_to_latent_bias"]) < 2e-4


def test_loss_scaler_protocol_skips_non_finite_steps():
    """The reference's GradScaler protocol (VMAE/util/misc.py:406-435, engine_pretrain.py:72-76) on the flat slab: a non-finite gradient
    skips the step and halves the scale; clean steps apply 1/scale inside the fused AdamW kernel and grow the scale on schedule."""
    from ldmae_amd import vmae_pretrain as vp
    lin = torch.nn.Linear(64, 64).cuda()
    opt = vp.build_optimizer(lin, 1e-2, 0.0)
    sc = vp.LossScaler(init_scale=1024.0, growth_interval=2)
    w0 = lin.weight.detach().clone()
    x = torch.randn(8, 64, device="cuda")
    (lin(x).pow(2).mean() * sc.scale).backward()
    ref = lin.weight.grad.detach().clone() / sc.scale
    opt.flat.grads[3] = float("inf")
    assert sc.step(opt) is None and sc.scale == 512.0 and sc.skipped == 1 and torch.equal(lin.weight, w0) and opt.step_count == 0
    opt.zero_grad()
    (lin(x).pow(2).mean() * sc.scale).backward()
    gn = sc.step(opt)
    assert opt.step_count == 1 and not torch.equal(lin.weight, w0) and abs(gn - float(torch.cat([ref.flatten(), lin.bias.grad.flatten() / sc.scale]).norm())) < 1e-3 * gn
    # first AdamW step moves every weight by lr * sign(g) (m / sqrt(v) = +-1): the 1/scale factor reached the kernel
    moved = (lin.weight.detach() - w0)
    nz = ref.abs() > 1e-6
    assert torch.allclose(moved[nz], -1e-2 * torch.sign(ref[nz]), atol=2e-4)
    opt.zero_grad()
    (lin(x).pow(2).mean() * sc.scale).backward()
    sc.step(opt)
    assert sc.scale == 1024.0                      # two clean steps: growth


def test_tiled_encoder_kernels_match_per_layer_path(golden):
    """`_encode` (models_mae.py:819-833: every patch, no masking) in bf16 through the tiled form of the fused encoder -- per block: q|k|v
    kernel, flash attention, proj / MLP kernel with the residual stream in registers -- against the per-layer bf16 kernels and the f32 path on
    the reference-pinned weights; under autograd and in f32 the per-layer kernels stay."""
    from ldmae_amd.tokenizer import fused_encoder
    cfg = omae.MAEConfig()
    m = build({}, full_sd(cfg), 256)
    imgs = det_randn("mae_img", (5, 3, 256, 256), 2).clamp(-1, 1).cuda()
    calls = []
    orig = fused_encoder.encoder_forward_tiled
    fused_encoder.encoder_forward_tiled = lambda model, x, which="enc", **kw: (calls.append(tuple(x.shape)), orig(model, x, which, **kw))[1]
    try:
        with torch.no_grad():
            lat32 = m._encode(imgs)                                                          # f32: never the fused kernels
            with torch.autocast("cuda", dtype=torch.bfloat16):
                fused = m._encode(imgs)
                m.fused_encoder = False
                layered = m._encode(imgs)
                m.fused_encoder = True
                one = m._encode(imgs[3:4])
        assert calls == [(5, 1024, 192), (1, 1024, 192)]
        assert fused.shape == lat32.shape and fused.dtype == torch.float32
        assert rel_err(fused.cpu(), lat32.cpu()) < 3e-2 and rel_err(layered.cpu(), lat32.cpu()) < 3e-2
        assert rel_err(fused.cpu(), layered.cpu()) < 2e-2
        assert torch.equal(one[0], fused[3])
        # the decoder stack of the shipped tokenizer has the same geometry: `decode` (-> decode_to_images) takes the same kernels with its own blob
        zlat = lat32[:, :16].contiguous()
        with torch.no_grad():
            rec32 = m.decode(zlat).sample
            with torch.autocast("cuda", dtype=torch.bfloat16):
                rec_f = m.decode(zlat).sample
                m.fused_encoder = False
                rec_l = m.decode(zlat).sample
                m.fused_encoder = True
                rec_1 = m.decode(zlat[3:4]).sample
        assert calls[-2:] == [(5, 1024, 192), (1, 1024, 192)] and len(calls) == 4
        assert rel_err(rec_f.cpu(), rec32.cpu()) < 3e-2 and rel_err(rec_l.cpu(), rec32.cpu()) < 3e-2 and rel_err(rec_f.cpu(), rec_l.cpu()) < 2e-2
        assert torch.equal(rec_1[0], rec_f[3])
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lat_g = m._encode(imgs[:1])
        assert lat_g.requires_grad and len(calls) == 4
    finally:
        fused_encoder.encoder_forward_tiled = orig


def test_fused_encoder_kernel_matches_per_layer_path(golden):
    """csrc/vmae_fused.hip (the whole encoder stack in one launch, one workgroup per image) against the per-layer bf16 kernels and the f32
    path, on the reference-pinned weights / images / masking noise of mae.npz at mask_ratio 0.75 (256 kept tokens)."""
    from ldmae_amd.tokenizer import fused_encoder
    g = golden("mae")
    cfg = omae.MAEConfig()
    m = build({}, full_sd(cfg), 256)
    imgs = det_randn("mae_img", (5, 3, 256, 256), 2).clamp(-1, 1).cuda()
    noise = torch.rand(5, 1024, generator=torch.Generator().manual_seed(3)).cuda()
    calls = []
    orig = fused_encoder.encoder_forward
    fused_encoder.encoder_forward = lambda model, x: (calls.append(tuple(x.shape)), orig(model, x))[1]
    try:
        with torch.no_grad():
            lat32, mask, _ = m.forward_encoder(imgs, 0.75, noise=noise)                       # f32: never the fused kernel
            with torch.autocast("cuda", dtype=torch.bfloat16):
                fused, mask_f, _ = m.forward_encoder(imgs, 0.75, noise=noise)
                m.fused_encoder = False
                layered, _, _ = m.forward_encoder(imgs, 0.75, noise=noise)
                m.fused_encoder = True
                other, _, _ = m.forward_encoder(imgs, 0.5, noise=noise)                       # 512 kept tokens: the tiled form of the fused kernels
                m.fused_encoder = False
                other_l, _, _ = m.forward_encoder(imgs, 0.5, noise=noise)
                m.fused_encoder = True
                odd, _, _ = m.forward_encoder(imgs, 0.7, noise=noise)                         # 307 kept tokens: per-layer path
        assert calls == [(5, 256, 192)] and other.shape[1] == 512 and odd.shape[1] == 307
        assert rel_err(other.cpu(), other_l.cpu()) < 2e-2
        assert fused.dtype == torch.float32 and torch.equal(mask_f, mask)
        assert rel_err(fused.cpu(), lat32.cpu()) < 3e-2 and rel_err(layered.cpu(), lat32.cpu()) < 3e-2
        assert rel_err(fused.cpu(), layered.cpu()) < 2e-2
        # every image is independent (one workgroup each): a batch of one gives the same bits
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            one, _, _ = m.forward_encoder(imgs[3:4], 0.75, noise=noise[3:4])
        assert torch.equal(one[0], fused[3])
        # the packed weights follow the parameters
        with torch.no_grad():
            m.blocks[7].mlp.fc2.weight.mul_(1.5)          # (a uniform bias shift would be removed again by the next LayerNorm)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                moved, _, _ = m.forward_encoder(imgs, 0.75, noise=noise)
        assert rel_err(moved.cpu(), fused.cpu()) > 1e-3
        # under autograd the encoder stays on the differentiable per-layer path
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lat_g, _, _ = m.forward_encoder(imgs[:1], 0.75, noise=noise[:1])
        assert lat_g.requires_grad and len(calls) == 3
    finally:
        fused_encoder.encoder_forward = orig


def test_patch_embed_of_kept_tokens_equals_embed_then_gather():
    """Inference masks first and embeds the kept quarter of the patches (ldmae_patch_gather + the gated-residual GEMM epilogue with the
    gathered pos rows as residual): same bits as the reference order, embed every patch then gather (models_mae.py:502-510)."""
    from ldmae_amd import ops
    cfg = omae.MAEConfig()
    m = build({}, full_sd(cfg), 256)
    imgs = det_randn("mae_img", (3, 3, 256, 256), 5).clamp(-1, 1).cuda()
    noise = torch.rand(3, 1024, generator=torch.Generator().manual_seed(4)).cuda()
    pe = m.patch_embed
    w2d = pe.proj.weight.view(192, -1)
    for dtype in (torch.bfloat16, torch.float32):
        with torch.no_grad():
            full = m._embed(imgs, dtype)
            ids_keep, mask, ids_restore = ops.random_masking(noise, 256)
            ref = ops.gather_rows(full.contiguous(), ids_keep)
            got = ops.patch_embed_kept(imgs, ids_keep, m.pos_embed[0], w2d, pe.proj.bias, 8, dtype)
        assert got.shape == ref.shape == (3, 256, 192) and torch.equal(got, ref), dtype


def test_tf32_class_docking_path_matches_reference_golden(golden):
    """The reference's drivers set torch.backends.cuda.matmul.allow_tf32 = True before they call `_encode` / `decode` in f32 (inference.py:79,
    extract_features.py:2-3).  With that flag on, f32 forward-only calls run the TF32-CLASS kernels (fp16 operands = TF32's 10-bit mantissa, f32
    accumulation: LDMAE_F16 family of the C ABI): within 1e-3 of the reference's own f32 results (TF32's error class; emulated TF32 gives 5.6e-4
    on the oracle), counted as the f16 family.  With the flag off (torch's default) the same calls stay on the exact-f32 kernels at 1e-4."""
    from ldmae_amd import _lib
    g = golden("mae")
    cfg = omae.MAEConfig()
    m = build({}, full_sd(cfg), 256)
    imgs = det_randn("mae_img", (2, 3, 256, 256), 2).clamp(-1, 1).cuda()
    noise = torch.from_numpy(g["mae_noise"]).cuda()
    prev = torch.backends.cuda.matmul.allow_tf32
    try:
        torch.backends.cuda.matmul.allow_tf32 = True
        _lib.launch_counts(reset=True)
        with torch.no_grad():
            mom = m._encode(imgs)
            c_enc = _lib.launch_counts(reset=True)
            rec = m.decode(mom[:, :16]).sample
            c_dec = _lib.launch_counts(reset=True)
            lat, mask, ids = m.forward_encoder(imgs, 0.75, noise=noise)
        nb, nd = len(m.blocks), len(m.decoder_blocks)
        # default: the tiled fused kernels on fp16 operands (csrc/vmae_fused.hip, F16) around the fp16 flash kernel -- no per-layer GEMM launches
        assert c_enc["nt_f16"] == 0 and c_enc["attn_f16"] == nb and c_enc["attn_f32"] == 0 and c_enc["nt_bf16"] == 0 and c_enc["attn_bf16"] == 0, c_enc
        assert c_dec["nt_f16"] == 0 and c_dec["attn_f16"] == nd and c_dec["attn_f32"] == 0 and c_dec["nt_bf16"] == 0 and c_dec["attn_bf16"] == 0, c_dec
        # the per-layer fp16 kernels (model.fused_encoder = False; geometries the fused kernels do not cover): same error class, own launch family
        m.fused_encoder = False
        _lib.launch_counts(reset=True)
        with torch.no_grad():
            mom_pl = m._encode(imgs)
            c_enc = _lib.launch_counts(reset=True)
            rec_pl = m.decode(mom_pl[:, :16]).sample
            c_dec = _lib.launch_counts(reset=True)
        m.fused_encoder = True
        assert c_enc["nt_f16"] == 4 * nb and c_enc["attn_f16"] == nb and c_enc["attn_f32"] == 0 and c_enc["nt_bf16"] == 0, c_enc
        assert c_dec["nt_f16"] == 4 * nd and c_dec["attn_f16"] == nd and c_dec["attn_f32"] == 0 and c_dec["nt_bf16"] == 0, c_dec
        assert rel_err(mom_pl[:, :, :2, :2].cpu(), g["mae_moments_head"]) < 1e-3 and rel_err(rec_pl[:, :, :4, :4].cpu(), g["mae_rec_head"]) < 1e-3
        assert rel_err(mom.cpu(), mom_pl.cpu()) < 1e-3 and rel_err(rec.cpu(), rec_pl.cpu()) < 1.5e-3
        assert rel_err(mom[:, :, :2, :2].cpu(), g["mae_moments_head"]) < 1e-3
        assert abs(float(mom.double().norm()) - float(g["mae_moments_norm"])) < 1e-3 * float(g["mae_moments_norm"])
        assert rel_err(rec[:, :, :4, :4].cpu(), g["mae_rec_head"]) < 1e-3
        assert abs(float(rec.double().norm()) - float(g["mae_rec_norm"])) < 1e-3 * float(g["mae_rec_norm"])
        np.testing.assert_array_equal(mask.cpu().numpy(), g["mae75_mask"])                   # masks / indices stay bit-exact
        np.testing.assert_array_equal(ids.cpu().numpy(), g["mae75_ids_restore"])
        assert rel_err(lat[:, :4].cpu(), g["mae75_lat_head"]) < 1e-3
        # a training call is NOT switched (the fp16 family is forward-only): gradients on -> exact f32
        _lib.launch_counts(reset=True)
        m.forward_encoder(imgs, 0.75, noise=noise)[0].sum().backward()
        c = _lib.launch_counts(reset=True)
        assert c["nt_f16"] == 0 and c["attn_f16"] == 0 and c["nt_f32"] > 0, c
        torch.backends.cuda.matmul.allow_tf32 = False
        with torch.no_grad():
            mom32 = m._encode(imgs)
        c = _lib.launch_counts(reset=True)
        assert c["nt_f16"] == 0 and c["attn_f16"] == 0 and c["attn_f32"] == nb, c
        assert rel_err(mom32[:, :, :2, :2].cpu(), g["mae_moments_head"]) < 1e-4
        assert 1e-6 < rel_err(mom.cpu(), mom32.cpu()) < 1e-3                                  # the two paths really differ, by TF32's margin
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev


def test_fp16_pretraining_loss_scaler_backs_off_on_overflow():
    """fp16 autocast has a range to overflow: with an absurd initial loss scale the gradient GEMMs / attention backward produce INFINITIES
    (LDMAE_EPI_F16_INF: gradient outputs are not saturated), the scaler sees the non-finite slab, skips the step and halves the scale
    (VMAE/util/misc.py:413-430 via torch.amp.GradScaler) until steps go through; parameters stay finite and the loss falls."""
    import argparse
    from ldmae_amd import vmae_pretrain as vp
    from ldmae_amd.tokenizer import models_mae
    torch.manual_seed(0)
    m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192, decoder_depth=2,
                                        decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16, no_cls=True,
                                        kl_loss_weight=1e-6, smooth_output=True).cuda()
    opt = vp.build_optimizer(m, 1e-3, 0.05)
    args = argparse.Namespace(accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=0, epochs=100, fixed_lr=True, precision="fp16", mask_ratio=0.75,
                              visible_loss_ratio=0.5, print_freq=1)
    x = torch.rand(8, 3, 128, 128, generator=torch.Generator().manual_seed(1)) * 2 - 1
    scaler = vp.LossScaler(init_scale=2.0 ** 36, enabled=True)
    losses = []
    vp.train_one_epoch(m, [(x, 0)] * 40, opt, 0, args, log=lambda s: losses.append(float(s.split("loss: ")[1].split()[0])), scaler=scaler)
    print("skipped", scaler.skipped, "scale", scaler.scale, "steps", opt.step_count, "loss", losses[0], losses[-1])
    assert scaler.skipped >= 5 and scaler.scale == 2.0 ** 36 * 0.5 ** scaler.skipped and opt.step_count == 40 - scaler.skipped and opt.step_count >= 10
    assert all(torch.isfinite(p).all() for p in m.parameters()) and np.isfinite(losses).all() and losses[-1] < losses[0]
    # fp16 AUTOCAST training is fp16 with ROUNDED branch outputs (what torch's autocast Linear hands to the residual add); only the TF32-class
    # docking calls join the residual stream unrounded -- decided where the type is resolved, not from the (switched-off) autocast state in the block
    blocks = list(m.blocks) + list(m.decoder_blocks)
    assert all(b.last_dtype == torch.float16 and b.last_tf32_class is False for b in blocks)
    prev = torch.backends.cuda.matmul.allow_tf32
    try:
        torch.backends.cuda.matmul.allow_tf32 = True
        m.fused_encoder = False
        with torch.no_grad():
            mom = m.eval()._encode(x.cuda())                      # 8 x 256 tokens: whole groups of 8 rows -> TF32-class
            assert all(b.last_dtype == torch.float16 and b.last_tf32_class is True for b in m.blocks)
            m.decode(mom[:3, :16])                                 # 3 x 256 rows: fine too
            # rows that are not whole groups of 8 (an odd batch of an odd grid: 81 tokens): the fp16 GEMM has no fallback -> exact f32 kernels
            assert m._docking_dtype(m.blocks, rows=81) == (torch.float32, False) and m._docking_dtype(m.blocks, rows=88) == (torch.float16, True)
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev


def test_tensor_hook_between_blocks_sees_and_changes_the_gradient():
    """The block backward hands the bf16-rounded residual gradient to the previous block on the gradient tensor itself (`_ldmae_cast`).  A
    TENSOR hook on a block's input runs in between and may edit that gradient -- also through an op that never bumps torch's version
    counter (every kernel of this library writes through raw pointers): the hand-off must then be dropped, not used stale.  A hook that adds
    a fixed perturbation through such a raw-pointer op must give exactly the gradients of the same hook written with torch's own in-place
    add, and a hook that does nothing must give the unhooked gradients bit for bit."""
    from ldmae_amd import ops
    from ldmae_amd.tokenizer import models_mae
    torch.manual_seed(0)
    m = models_mae.MaskedAutoencoderViT(img_size=64, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192, decoder_depth=3,
                                        decoder_num_heads=12, mlp_ratio=4, norm_layer=models_mae._ln(), latent_dim=16, no_cls=True,
                                        kl_loss_weight=1e-6, smooth_output=True).cuda().train()
    imgs = (torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 2 - 1).cuda()
    noise = torch.rand(4, 64, generator=torch.Generator().manual_seed(2)).cuda()
    eps = torch.randn(4, 16, 16, generator=torch.Generator().manual_seed(3)).cuda()
    delta = (torch.randn(4 * 64 * 192, generator=torch.Generator().manual_seed(4)) * 1e-3).cuda()

    def grads(hook):
        run = m._run

        def hooked_run(blocks, x, dtype=None, tf32_class=False):
            for i, blk in enumerate(blocks):
                x = blk(x, m._chain_ok(blk), dtype, tf32_class)
                if hook is not None and blocks is m.decoder_blocks and i == 0:
                    x.register_hook(hook)
            return x
        m._run = hooked_run
        try:
            m.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = m(imgs, 0.75, 0.5, _noise=noise, _eps=eps)[0]
            loss.backward()
        finally:
            m._run = run
        return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    base = grads(None)
    idle = grads(lambda g: None)
    assert all(torch.equal(base[k], idle[k]) for k in base)

    def raw(g):                                   # edits the gradient in place through the C ABI: no version bump
        ops.multi_add_([g.view(-1)], [delta])

    def plain(g):
        g.view(-1).add_(delta)
    a, b = grads(raw), grads(plain)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert any(not torch.equal(a[k], base[k]) for k in a)         # and the perturbation did reach the earlier blocks


def test_every_registry_arch_trains_one_step_in_every_precision():
    """Every constructor of BOTH trees' registries (LDMAE/tokenizer/models_mae.py:977-1083, VMAE/models_mae.py:1014-1134) at depth 1: one pre-training step
    (loss + backward) in f32, bf16 autocast and fp16 autocast (the reference's own, engine_pretrain.py:51-57), then the docking calls.  Geometries off a kernel
    family's grid fall back to MORE precision, never less (fp16 autocast off heads of 16 -> f32 activations; widths off the 16-bit GEMMs' 64-grid -> f32;
    head dims 12 / 24 / 80 -> zero-padded heads; patch 14: patch-embed K = 588 and the prediction head's N = 588 zero-padded to the GEMMs' 16-grid;
    D = 1280 LayerNorm).  A coverage sweep: finite loss / gradients / reconstructions -- parity per geometry is the other tests' job."""
    from ldmae_amd.tokenizer import models_mae as mm
    names = ["mae_for_ldmae", "mae_for_ldmae_f8d32", "mae_for_ldmae_f8d16_prev", "mae_for_ldmae_f8d16_small", "mae_for_ldmae_f8d16_asym_small",
             "mae_for_ldmae_f8d16_prev_large", "mae_for_ldmae_f8d16", "mae_for_ldmae_f8d16_flexible", "mae_for_ldmae_f16d32", "mae_for_ldmae_f16d32_large",
             "mae_for_ldmae_f8d32_flexible", "mae_for_ldmae_16d", "mae_vit_base_patch16_dec512d8b", "mae_vit_base_patch16_dec128d8b",
             "mae_vit_large_patch16_dec512d8b", "mae_vit_huge_patch14_dec512d8b"]
    for n in names:
        f = getattr(mm, n)
        for prec in ("fp32", "bf16", "fp16"):
            torch.manual_seed(0)
            try:
                p = f(no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=64).patch_embed.patch_size[0]
                S = p * 8 if p != 14 else 112
                m = f(no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=S)
            except TypeError:                                   # constructors that fix img_size themselves, as in the reference
                m = f(no_cls=True, kl_loss_weight=1e-6, smooth_output=True)
                S = m.img_size
            m.blocks, m.decoder_blocks = m.blocks[:1], m.decoder_blocks[:1]
            m = m.cuda().train()
            x = torch.rand(3, 3, S, S, device="cuda") * 2 - 1
            with torch.autocast("cuda", dtype=torch.float16 if prec == "fp16" else torch.bfloat16, enabled=prec != "fp32"):
                loss = m(x, mask_ratio=0.75, visible_loss_ratio=0.5)[0]
            loss.backward()
            assert bool(torch.isfinite(loss)) and all(torch.isfinite(q.grad).all() for q in m.parameters() if q.grad is not None), (n, prec)
            m.eval()
            with torch.no_grad():
                rec = m.decode(m._encode(x)[:, :m.latent_dim]).sample
            assert rec.shape == x.shape and bool(torch.isfinite(rec).all()), (n, prec)
            del m
