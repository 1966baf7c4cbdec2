"""Pin the oracle: every golden vector produced by importing the reference
(tests/golden/make_golden.py) must be reproduced by the CPU restatement."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import dit, mae, train, transport
from weights import DIT_FLAG_VARIANTS, det_randn, det_weights

TINY = dit.DiTConfig(input_size=8, patch_size=1, in_channels=16, hidden_size=192, depth=2, num_heads=3,
                     num_classes=10, class_dropout_prob=0.5)
B1 = dit.DiTConfig(**dit.DIT_B_1)


def sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def tiny_sd(seed=1):
    sd = det_weights(dit.param_shapes(TINY), seed)
    sd.update(dit.fixed_tables(TINY))
    return sd


def test_b1_tables_bit_exact(golden):
    g = golden("kernels")
    tb = dit.fixed_tables(B1)
    assert sha(tb["pos_embed"]) == str(g["b1_pos_sha"])
    assert sha(tb["feat_rope.freqs_cos"]) == str(g["b1_cos_sha"])
    assert sha(tb["feat_rope.freqs_sin"]) == str(g["b1_sin_sha"])
    np.testing.assert_array_equal(tb["feat_rope.freqs_cos"][[0, 1, 33, 1023]].numpy(), g["b1_cos_rows"])


def test_b1_state_dict_keys_and_param_count(golden):
    g = golden("kernels")
    shapes = dit.param_shapes(B1)
    keys = sorted(list(shapes) + ["feat_rope.freqs_cos", "feat_rope.freqs_sin"])
    assert keys == [str(k) for k in g["b1_keys"]]
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["b1_nparams"])


def test_dit_tiny_forward(golden):
    g = golden("dit_tiny")
    sd = tiny_sd()
    taps = {}
    out = dit.dit_forward(sd, torch.from_numpy(g["dit_xt"]), torch.from_numpy(g["dit_t"]),
                          torch.from_numpy(g["dit_y"]), TINY, True, torch.from_numpy(g["dit_drop"]), taps)
    assert bool(g["dit_drop"].any()) and not bool(g["dit_drop"].all())
    assert rel_err(taps["blocks.0.out"], g["dit_blk0"]) < 2e-6
    assert rel_err(taps["blocks.1.out"], g["dit_blk1"]) < 2e-6
    assert rel_err(out, g["dit_out"]) < 2e-6


def test_dit_tiny_loss_and_grads(golden):
    g = golden("dit_tiny")
    sd = tiny_sd()
    # the oracle's own host draws must reproduce the reference's (x0 torch, t numpy, drop torch)
    torch.manual_seed(5)
    np.random.seed(5)
    x1 = torch.from_numpy(g["tl_x1"])
    t, x0, _ = transport.sample(x1)
    drop = torch.rand(2) < TINY.class_dropout_prob
    np.testing.assert_array_equal(x0.numpy(), g["tl_x0"])
    np.testing.assert_array_equal(t.numpy(), g["tl_t"])
    np.testing.assert_array_equal(drop.numpy(), g["tl_drop"])
    y = torch.from_numpy(g["dit_y"])
    keys = train.trainable_keys(TINY)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
    full = dict(sd)
    full.update(leaves)
    terms = transport.training_losses(lambda xt, tt: dit.dit_forward(full, xt, tt, y, TINY, True, drop), x1, t, x0)
    assert rel_err(terms["loss"], g["tl_loss_b"]) < 2e-6
    assert rel_err(terms["pred"], g["tl_pred"]) < 2e-6
    terms["loss"].mean().backward()
    names = [str(n) for n in g["tl_grad_names"]]
    assert set(names) == set(keys)
    for i, k in enumerate(names):
        gr = leaves[k].grad
        assert abs(float(gr.double().norm()) - g["tl_grad_norm"][i]) <= 2e-5 * g["tl_grad_norm"][i] + 1e-9, k
        np.testing.assert_allclose(gr.flatten()[:8].numpy(), g["tl_grad_head"][i], rtol=2e-4, atol=2e-7, err_msg=k)


def test_cfg_and_euler(golden):
    g = golden("dit_tiny")
    sd = tiny_sd()
    z, y = torch.from_numpy(g["cfg_z"]), torch.from_numpy(g["cfg_y"])
    lo = dit.dit_forward_with_cfg(sd, z, torch.full((4,), 0.05), y, TINY, 4.0, True, 0.10)
    hi = dit.dit_forward_with_cfg(sd, z, torch.full((4,), 0.50), y, TINY, 4.0, True, 0.10)
    assert rel_err(lo, g["cfg_lo"]) < 2e-6 and rel_err(hi, g["cfg_hi"]) < 2e-6
    grid = transport.shifted_time_grid(4, 0.3)
    traj = transport.euler_ode(lambda x, t: dit.dit_forward_with_cfg(sd, x, t, y, TINY, 4.0, True, 0.10), z, grid)
    assert rel_err(traj[-1], g["euler_last"]) < 5e-6


def test_kernel_vectors_real_width(golden):
    g = golden("kernels")
    x = det_randn("k5_x", (2, 8, 768), 3)
    w = 1 + 0.1 * det_randn("k5_w", (768,), 3)
    sh, sc = 0.3 * det_randn("k5_sh", (2, 768), 3), 0.3 * det_randn("k5_sc", (2, 768), 3)
    assert rel_err(dit.modulate(dit.rmsnorm(x, w), sh, sc), g["k5_out"]) < 1e-6
    cos, sin = dit.rope_tables(64, 32)
    rq = dit.apply_rope(det_randn("k8_q", (1, 2, 1024, 64), 3), cos, sin)
    assert sha(rq) == str(g["k8_sha"])
    sdf = det_weights({"w12.weight": (4096, 768), "w12.bias": (4096,), "w3.weight": (768, 2048), "w3.bias": (768,)}, 4)
    assert rel_err(dit.swiglu(sdf, "", det_randn("k11_x", (8, 768), 3)), g["k11_out"]) < 2e-6
    sda = det_weights({"qkv.weight": (2304, 768), "qkv.bias": (2304,), "q_norm.weight": (64,), "k_norm.weight": (64,),
                       "proj.weight": (768, 768), "proj.bias": (768,)}, 6)
    ao = dit.attention(sda, "", det_randn("k9_x", (1, 1024, 768), 3) * 0.5, B1, cos, sin)
    assert rel_err(ao[0, :4], g["k9_head"]) < 5e-6 and rel_err(ao[0, -4:], g["k9_tail"]) < 5e-6
    assert abs(float(ao.double().norm()) - float(g["k9_norm"])) < 1e-5 * float(g["k9_norm"])
    np.testing.assert_allclose(dit.timestep_embedding(torch.tensor([0.0, 0.25, 0.9])).numpy(), g["k2_emb"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(transport.shifted_time_grid(250, 0.3).numpy(), g["euler_grid"], rtol=0, atol=1e-7)
    np.random.seed(11)
    np.testing.assert_array_equal(transport.sample_logit_normal(16).numpy(), g["lognorm_t_seed11"])


def test_mae_masking_bit_exact_and_encoder(golden):
    g = golden("mae")
    cfg = mae.MAEConfig()
    shapes = mae.param_shapes(cfg)
    sd = det_weights(shapes, 2)
    sd.update(mae.fixed_tables(cfg))
    assert sha(sd["pos_embed"]) == str(g["mae_pos_sha"])
    assert sorted(list(shapes) + ["pos_embed", "decoder_pos_embed"]) == [str(k) for k in g["mae_keys"]]
    assert sum(int(np.prod(s)) for s in shapes.values()) + 2 * 1024 * 192 == int(g["mae_nparams"])
    imgs = det_randn("mae_img", (2, 3, 256, 256), 2).clamp(-1, 1)
    noise = torch.from_numpy(g["mae_noise"])
    for tag, ratio in (("75", 0.75), ("25", 0.25)):
        lat, mask, ids = mae.forward_encoder(sd, imgs, noise, ratio, cfg)
        np.testing.assert_array_equal(mask.numpy(), g[f"mae{tag}_mask"])          # bit-exact
        np.testing.assert_array_equal(ids.numpy(), g[f"mae{tag}_ids_restore"])     # bit-exact
        assert list(lat.shape) == list(g[f"mae{tag}_lat_shape"])
        assert rel_err(lat[:, :4], g[f"mae{tag}_lat_head"]) < 5e-6
        assert abs(float(lat.double().norm()) - float(g[f"mae{tag}_lat_norm"])) < 1e-5 * float(g[f"mae{tag}_lat_norm"])
    mom = mae.encode_moments(sd, imgs, cfg)
    assert rel_err(mom[:, :, :2, :2], g["mae_moments_head"]) < 5e-6
    rec = mae.decode(sd, mom[:, :16], cfg)
    assert rel_err(rec[:, :, :4, :4], g["mae_rec_head"]) < 1e-5
    img8 = mae.to_uint8_images(rec)
    assert (np.abs(img8[:, :4, :4].astype(int) - g["mae_img8_head"].astype(int)) <= 1).all()


def test_masking_ties_are_broken_by_index():
    noise = np.zeros((1, 8), dtype=np.float32)
    noise[0, 5] = -1
    keep, mask, restore = mae.random_masking_ids(noise, 0.5)
    assert keep.tolist() == [[5, 0, 1, 2]]
    assert mask.tolist() == [[0, 0, 0, 1, 1, 0, 1, 1]]


def test_adamw_restatement_matches_torch_optim():
    torch.manual_seed(0)
    p0 = {"a": torch.randn(5, 7), "b": torch.randn(11)}
    ref = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    opt = torch.optim.AdamW(ref.values(), lr=2e-4, weight_decay=0, betas=(0.9, 0.95))
    sd = {k: v.clone() for k, v in p0.items()}
    st = train.AdamWState(list(sd), sd)
    for _ in range(5):
        grads = {k: torch.randn_like(v) for k, v in sd.items()}
        for k in ref:
            ref[k].grad = grads[k].clone()
        opt.step()
        train.adamw_step(sd, grads, st)
    for k in sd:
        np.testing.assert_allclose(sd[k].numpy(), ref[k].detach().numpy(), rtol=1e-6, atol=1e-7)


def test_curve_fixture_present(golden):
    g = golden("curve")
    assert len(g["curve_losses"]) == 100


def test_mae_pretraining_forward_and_grads_vs_reference_golden(golden):
    """SURVEY 8(f)4 pin: oracle.mae.forward_vanilla (and torch autograd through it) against the reference's own
    MaskedAutoencoderViT.forward + backward (tokenizer/models_mae.py:733-790, 811-815) on the recorded masking / posterior noise."""
    g = golden("mae_train")
    cfg = mae.MAEConfig(img_size=128, depth=2, decoder_depth=2)
    sd = det_weights(mae.param_shapes(cfg), 6)
    sd.update(mae.fixed_tables(cfg))
    keys = sorted(mae.param_shapes(cfg))
    assert keys == [str(k) for k in g["mt_keys"]]                 # the reference's trainable-parameter names
    imgs = det_randn("img128p", (2, 3, 128, 128), 4).clamp(-1, 1)
    for tag in ("a", "b"):
        ratio, vlr = (float(v) for v in g[f"mt{tag}_ratio"])
        leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
        osd = dict(sd)
        osd.update(leaves)
        loss, pred, mask, vis, mloss, kl = mae.forward_vanilla(osd, imgs, torch.from_numpy(g[f"mt{tag}_noise"]), torch.from_numpy(g[f"mt{tag}_eps"]),
                                                               ratio, vlr, 1e-3, cfg)
        np.testing.assert_array_equal(mask.numpy(), g[f"mt{tag}_mask"])                       # bit-exact
        np.testing.assert_allclose([float(loss), float(vis), float(mloss), float(kl)], g[f"mt{tag}_loss"], rtol=2e-5)
        assert rel_err(pred.detach()[:, :6, :24], g[f"mt{tag}_pred_head"]) < 1e-5
        assert abs(float(pred.detach().double().norm()) - float(g[f"mt{tag}_pred_norm"])) < 1e-5 * float(g[f"mt{tag}_pred_norm"])
        loss.backward()
        norms = np.array([float(leaves[k].grad.double().norm()) for k in keys])
        np.testing.assert_allclose(norms, g[f"mt{tag}_grad_norms"], rtol=2e-4, atol=1e-9)
        assert rel_err(leaves["decoder_pred.conv_smoother.weight"].grad, g[f"mt{tag}_grad_smoother"]) < 1e-4
        assert rel_err(leaves["to_latent.weight"].grad[:4, :8], g[f"mt{tag}_grad_to_latent_head"]) < 1e-4


def test_other_registry_geometries_vs_reference_golden(golden):
    """The reference itself on the registry's other geometries (tests/golden/make_golden.py: gen_mae_archs): 'dn' = mae_for_ldmae_f8d16
    (down_nonlinear MLP_dim_resize latent maps, 384-wide decoder, 24 heads of 16), 'h24' = mae_for_ldmae_f8d16_prev_large (16 heads of 24) --
    pins oracle.mae's restatement of both (pre-training step with its gradients, docking encode / decode)."""
    g = golden("mae_archs")
    cfgs = {"dn": (mae.MAEConfig(img_size=64, depth=1, decoder_embed_dim=384, decoder_num_heads=24, decoder_depth=1, down_nonlinear=True), 9, "img64d"),
            "h24": (mae.MAEConfig(img_size=64, embed_dim=384, num_heads=16, depth=1, decoder_embed_dim=384, decoder_num_heads=16, decoder_depth=1), 8, "img64p"),
            "p16": (mae.MAEConfig(img_size=64, patch_size=16, depth=1, decoder_depth=1), 7, "img64q")}
    for tag, (cfg, seed, iname) in cfgs.items():
        sd = det_weights(mae.param_shapes(cfg), seed)
        sd.update(mae.fixed_tables(cfg))
        keys = sorted(mae.param_shapes(cfg))
        assert keys == [str(k) for k in g[f"ar_{tag}_keys"]]                     # the reference's parameter names (incl. to_latent.layers.0 / .2)
        imgs = det_randn(iname, (2, 3, 64, 64), 4).clamp(-1, 1)
        leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
        osd = dict(sd)
        osd.update(leaves)
        loss, pred, mask, vis, mloss, kl = mae.forward_vanilla(osd, imgs, torch.from_numpy(g[f"ar_{tag}_noise"]), torch.from_numpy(g[f"ar_{tag}_eps"]),
                                                               0.75, 0.5, 1e-3, cfg)
        np.testing.assert_array_equal(mask.numpy(), g[f"ar_{tag}_mask"])
        np.testing.assert_allclose([float(loss), float(vis), float(mloss), float(kl)], g[f"ar_{tag}_loss"], rtol=2e-5)
        assert rel_err(pred.detach()[:, :6, :24], g[f"ar_{tag}_pred_head"]) < 1e-5
        loss.backward()
        norms = np.array([float(leaves[k].grad.double().norm()) for k in keys])
        np.testing.assert_allclose(norms, g[f"ar_{tag}_grad_norms"], rtol=2e-4, atol=1e-9)
        with torch.no_grad():
            mom = mae.encode_moments(sd, imgs, cfg)
            rec = mae.decode(sd, mom[:, :16], cfg)
        assert rel_err(mom[:, :, :2, :2], g[f"ar_{tag}_moments_head"]) < 1e-5 and rel_err(rec[:, :, :4, :4], g[f"ar_{tag}_rec_head"]) < 1e-5
        assert abs(float(rec.double().norm()) - float(g[f"ar_{tag}_rec_norm"])) < 1e-5 * float(g[f"ar_{tag}_rec_norm"])


DIT_VARIANTS = {"p2": (dict(input_size=16, patch_size=2, in_channels=4, hidden_size=192, depth=1, num_heads=3, num_classes=10, class_dropout_prob=0.1,
                            learn_sigma=True), 3, (2, 4, 16, 16)),
                "hd72": (dict(input_size=8, patch_size=1, in_channels=16, hidden_size=576, depth=1, num_heads=8, num_classes=10, class_dropout_prob=0.1),
                         4, (2, 16, 8, 8))}


def test_dit_variants_vs_reference_golden(golden):
    """oracle.dit on the geometries the shipped config does not exercise, pinned on the reference's own eval forward (make_golden.py:
    gen_dit_variants): patch size 2 with learn_sigma (the /2 registry entries) and head_dim 72 (XL's heads)."""
    g = golden("dit_variants")
    for tag, (kw, seed, xs) in DIT_VARIANTS.items():
        cfg = dit.DiTConfig(**kw)
        sd = det_weights(dit.param_shapes(cfg), seed)
        sd.update(dit.fixed_tables(cfg))
        x, t, y = det_randn("x", xs, 1), torch.tensor([0.2, 0.7]), torch.tensor([1, 5])
        out = dit.dit_forward(sd, x, t, y, cfg, train=False)
        assert out.shape == g[f"dv_{tag}_out"].shape and rel_err(out, g[f"dv_{tag}_out"]) < 2e-6, tag



def test_dit_block_flags_vs_reference_golden(golden):
    """oracle.dit with every LightningDiTBlock flag flipped away from the shipped imagenet YAML -- first of all use_qknorm=False, the reference's
    CelebA-HQ configuration (configs/celeba_hq/lightningdit_b_vmae_f8d16_cfg.yaml:30; README.md:108-110) -- pinned on the reference's own
    train-mode forward, loss and every parameter gradient (make_golden.py: gen_dit_flags), plus the state-dict key set each variant has."""
    g = golden("dit_flags")
    xt, t, tgt = det_randn("xt", (2, 16, 8, 8), 7), torch.tensor([0.3, 0.8]), det_randn("tgt", (2, 16, 8, 8), 11)
    base = dict(input_size=8, patch_size=1, in_channels=16, hidden_size=192, depth=2, num_heads=3, num_classes=10, class_dropout_prob=0.5)
    for n, (tag, over) in enumerate(DIT_FLAG_VARIANTS.items()):
        cfg = dit.DiTConfig(**{**base, **over})
        sd = det_weights(dit.param_shapes(cfg), 20 + n)
        tabs = dit.fixed_tables(cfg)
        assert sorted(list(sd) + list(tabs)) == [str(k) for k in g[f"df_{tag}_keys"]], tag
        sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        sd.update(tabs)
        out = dit.dit_forward(sd, xt, t, torch.from_numpy(g[f"df_{tag}_y"]), cfg, True, torch.from_numpy(g[f"df_{tag}_drop"]))
        loss = ((out - tgt) ** 2).mean()
        loss.backward()
        assert rel_err(out, g[f"df_{tag}_out"]) < 2e-6, tag
        assert abs(float(loss) - float(g[f"df_{tag}_loss"])) < 2e-6 * float(g[f"df_{tag}_loss"]), tag
        names = [str(k) for k in g[f"df_{tag}_grad_names"]]
        assert sorted(names) == sorted(k for k in sd if sd[k].requires_grad), tag
        for k, gn, gh in zip(names, g[f"df_{tag}_grad_norm"], g[f"df_{tag}_grad_head"]):
            assert abs(float(sd[k].grad.double().norm()) - gn) <= 2e-5 * gn + 1e-9, (tag, k)
            assert rel_err(sd[k].grad.flatten()[:8], gh) < 1e-4 or float(np.abs(gh).max()) < 1e-6, (tag, k)
    cfg = dit.DiTConfig(input_size=8, hidden_size=768, depth=1, num_heads=12, num_classes=1, use_qknorm=False)
    sd = det_weights(dit.param_shapes(cfg), 31)
    sd.update(dit.fixed_tables(cfg))
    out = dit.dit_forward(sd, det_randn("x", (2, 16, 8, 8), 1), torch.tensor([0.2, 0.7]), torch.tensor([0, 0]), cfg, train=False)
    assert rel_err(out, g["df_noqk768_out"]) < 2e-6


def test_pretraining_tree_kl_forms_vs_reference_golden(golden):
    """The PRE-TRAINING tree's own step (VMAE/models_mae.py:773-807 with the KL of VMAE/util/misc.py:103-125 -- what VMAE/engine_pretrain.py:51-57 runs;
    tests/golden/vmae_tree.npz, make_golden.py: gen_vmae_tree, generated from /root/reference/VMAE): the variance-only KL without fixed_std (no mean^2 term in
    that tree), and the fixed_std form (train_ae.sh:33 passes 1e-3) at two values.  oracle.mae.forward_vanilla(kl_form="vmae", fixed_std=...): loss terms,
    mask and every gradient norm."""
    g = golden("vmae_tree")
    cfg = mae.MAEConfig(img_size=128, depth=2, decoder_depth=2)
    sd = det_weights(mae.param_shapes(cfg), 6)
    sd.update(mae.fixed_tables(cfg))
    keys = sorted(mae.param_shapes(cfg))
    assert keys == [str(k) for k in g["vt_keys"]]
    imgs = det_randn("img128p", (2, 3, 128, 128), 4).clamp(-1, 1)
    for tag in ("n", "f", "h"):
        ratio, vlr, fs = (float(v) for v in g[f"vt{tag}_cfg"])
        leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
        osd = dict(sd)
        osd.update(leaves)
        loss, pred, mask, vis, mloss, kl = mae.forward_vanilla(osd, imgs, torch.from_numpy(g[f"vt{tag}_noise"]), torch.from_numpy(g[f"vt{tag}_eps"]),
                                                               ratio, vlr, 1e-3, cfg, kl_form="vmae", fixed_std=None if fs < 0 else fs)
        np.testing.assert_array_equal(mask.numpy(), g[f"vt{tag}_mask"])
        np.testing.assert_allclose([float(loss), float(vis), float(mloss), float(kl)], g[f"vt{tag}_loss"], rtol=3e-5)
        loss.backward()
        norms = np.array([float(leaves[k].grad.double().norm()) for k in keys])
        np.testing.assert_allclose(norms, g[f"vt{tag}_grad_norms"], rtol=3e-4, atol=1e-9)
        assert rel_err(leaves["to_latent.bias"].grad, g[f"vt{tag}_grad_to_latent_bias"]) < 1e-4
    # and the two trees really differ: the tokenizer tree's KL on the same draws is another number
    tk = mae.forward_vanilla(sd, imgs, torch.from_numpy(g["vtn_noise"]), torch.from_numpy(g["vtn_eps"]), 0.75, 0.5, 1e-3, cfg)[5]
    assert abs(float(tk) - float(g["vtn_loss"][3])) > 1e-2 * float(g["vtn_loss"][3])
