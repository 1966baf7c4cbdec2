#!/usr/bin/env python3
"""Generate the committed golden vectors by IMPORTING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 TORCH_COMPILE_DISABLE=1 python tests/golden/make_golden.py [--curve]

The reference's third-party imports that are absent here (timm, fairscale,
torchdiffeq, diffusers, torchvision, taming) are replaced by minimal stand-ins
inserted into ``sys.modules`` *before* import (recipe: SURVEY.md §8c).  The
stand-ins are harness code, not product code.  Outputs are data only
(inputs / expected outputs, hashes); no reference source text is stored.

Weights are regenerated on both sides from ``weights.det_weights`` so the
fixtures stay small.
"""
import argparse
import hashlib
import os
import sys
import types
from dataclasses import dataclass

os.environ.setdefault("TORCH_COMPILE_DISABLE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/LDMAE"
sys.path.insert(0, HERE)
from weights import det_weights, det_randn, DIT_FLAG_VARIANTS  # noqa: E402


# --------------------------------------------------------------------------- shims
def install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
            super().__init__()
            self.patch_size = (patch_size, patch_size)
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
            self.norm = nn.Identity()

        def forward(self, x):
            return self.norm(self.proj(x).flatten(2).transpose(1, 2))

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features or in_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    class DropPath(nn.Identity):
        def __init__(self, p=0.0):
            super().__init__()

    mod("timm")
    mod("timm.models")
    mod("timm.models.vision_transformer", PatchEmbed=PatchEmbed, Mlp=Mlp, DropPath=DropPath)
    mod("fairscale")
    mod("fairscale.nn")
    mod("fairscale.nn.model_parallel")
    mod("fairscale.nn.model_parallel.initialize")
    mod("fairscale.nn.model_parallel.layers", ColumnParallelLinear=object, ParallelEmbedding=object,
        RowParallelLinear=object)

    def odeint(fn, x, t, method="euler", atol=None, rtol=None):
        assert method == "euler"
        xs = [x]
        for k in range(len(t) - 1):
            x = x + (t[k + 1] - t[k]) * fn(t[k], x)
            xs.append(x)
        return torch.stack(xs)
    mod("torchdiffeq", odeint=odeint)

    class BaseOutput:            # attribute access only; subclasses are @dataclass
        pass
    mod("diffusers", ConfigMixin=object, ModelMixin=object)
    mod("diffusers.utils", BaseOutput=BaseOutput)

    class _Inert:
        def __init__(self, *a, **k):
            pass
    tv = mod("torchvision")
    tr = mod("torchvision.transforms", **{n: _Inert for n in
             ["Compose", "Lambda", "RandomHorizontalFlip", "ToTensor", "Normalize", "RandomResizedCrop", "Resize"]})
    tr.functional = mod("torchvision.transforms.functional")
    tv.transforms = tr
    mod("torchvision.datasets", ImageFolder=_Inert)
    mod("taming")
    mod("taming.modules")
    mod("taming.modules.losses")
    mod("taming.modules.losses.lpips", LPIPS=_Inert)


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def load_det(model, seed=0, skip=()):
    shapes = {k: tuple(v.shape) for k, v in model.named_parameters()}
    w = det_weights(shapes, seed)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k in w and k not in skip:
                p.copy_(w[k])


# --------------------------------------------------------------------------- DiT tiny
TINY = dict(input_size=8, patch_size=1, in_channels=16, hidden_size=192, depth=2, num_heads=3,
            num_classes=10, class_dropout_prob=0.5)
FLAGS = dict(use_qknorm=True, use_swiglu=True, use_rope=True, use_rmsnorm=True, wo_shift=False)


def gen_dit_tiny(out):
    from models.lightningdit import LightningDiT
    from transport import create_transport
    torch.manual_seed(0)
    m = LightningDiT(**TINY, **FLAGS)
    load_det(m, seed=1)
    m.train()
    B = 2
    x1 = det_randn("x1", (B, 16, 8, 8), 7)
    y = torch.tensor([3, 7])
    # model-only forward with controlled label drop (sample 1 dropped)
    taps = {}
    hooks = [blk.register_forward_hook(lambda mod, i, o, n=n: taps.__setitem__(n, o.detach().clone()))
             for n, blk in enumerate(m.blocks)]
    torch.manual_seed(123)
    drop = torch.rand(B) < TINY["class_dropout_prob"]
    torch.manual_seed(123)
    t_in = torch.tensor([0.3, 0.8])
    xt_in = det_randn("xt", (B, 16, 8, 8), 7)
    o = m(xt_in, t_in, y)
    for h in hooks:
        h.remove()
    out.update(dit_xt=xt_in.numpy(), dit_t=t_in.numpy(), dit_y=y.numpy(), dit_drop=drop.numpy(),
               dit_out=o.detach().numpy(), dit_blk0=taps[0].numpy(), dit_blk1=taps[1].numpy())
    # full loss + grads through reference transport with seeded RNGs
    tr = create_transport("Linear", "velocity", None, None, None, use_cosine_loss=False, use_lognorm=True)
    torch.manual_seed(5)
    np.random.seed(5)
    terms = tr.training_losses(m, x1, dict(y=y))
    loss = terms["loss"].mean()
    loss.backward()
    # replay the draws to record them
    torch.manual_seed(5)
    np.random.seed(5)
    x0 = torch.randn_like(x1)
    z = np.random.standard_normal(B)
    t = torch.tensor(1 / (1 + np.exp(-z)), dtype=torch.float32)
    drop2 = torch.rand(B) < TINY["class_dropout_prob"]
    out.update(tl_x1=x1.numpy(), tl_x0=x0.numpy(), tl_t=t.numpy(), tl_drop=drop2.numpy(),
               tl_loss_b=terms["loss"].detach().numpy(), tl_pred=terms["pred"].detach().numpy())
    names, gn, gs, gh = [], [], [], []
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        gn.append(float(p.grad.double().norm()))
        gs.append(float(p.grad.double().sum()))
        gh.append(p.grad.flatten()[:8].numpy().copy())
    out.update(tl_grad_names=np.array(names), tl_grad_norm=np.array(gn), tl_grad_sum=np.array(gs),
               tl_grad_head=np.stack(gh))
    # CFG forward (eval) on doubled batch + 3-step Euler sample with shift
    m.eval()
    from transport import Sampler
    z0 = det_randn("z0", (2, 16, 8, 8), 9)
    zz = torch.cat([z0, z0], 0)
    yy = torch.tensor([1, 4, 10, 10])
    with torch.no_grad():
        o_cfg_lo = m.forward_with_cfg(zz, torch.full((4,), 0.05), yy, 4.0, True, 0.10)
        o_cfg_hi = m.forward_with_cfg(zz, torch.full((4,), 0.50), yy, 4.0, True, 0.10)
        fn = Sampler(tr).sample_ode(sampling_method="euler", num_steps=4, atol=1e-6, rtol=1e-3, reverse=False,
                                    timestep_shift=0.3)
        traj = fn(zz, m.forward_with_cfg, y=yy, cfg_scale=4.0, cfg_interval=True, cfg_interval_start=0.10)
    out.update(cfg_z=zz.numpy(), cfg_y=yy.numpy(), cfg_lo=o_cfg_lo.numpy(), cfg_hi=o_cfg_hi.numpy(),
               euler_last=traj[-1].numpy())


# --------------------------------------------------------------------------- tables / kernels at real width
def gen_tables_and_kernels(out):
    from models.lightningdit import LightningDiT_models, modulate
    from models.rmsnorm import RMSNorm
    from models.swiglu_ffn import SwiGLUFFN
    from transport.integrators import ode
    m = LightningDiT_models["LightningDiT-B/1"](input_size=32, num_classes=1000, in_channels=16, **FLAGS)
    out.update(b1_pos_sha=sha(m.pos_embed), b1_pos_head=m.pos_embed[0, :3, :].numpy().copy(),
               b1_pos_tail=m.pos_embed[0, -2:, :].numpy().copy(),
               b1_cos_sha=sha(m.feat_rope.freqs_cos), b1_sin_sha=sha(m.feat_rope.freqs_sin),
               b1_cos_rows=m.feat_rope.freqs_cos[[0, 1, 33, 1023]].numpy().copy(),
               b1_sin_rows=m.feat_rope.freqs_sin[[0, 1, 33, 1023]].numpy().copy(),
               b1_nparams=np.array(sum(p.numel() for p in m.parameters())),
               b1_keys=np.array(sorted(m.state_dict().keys())))
    # RMSNorm + modulate at width 768
    x = det_randn("k5_x", (2, 8, 768), 3)
    w = 1 + 0.1 * det_randn("k5_w", (768,), 3)
    sh, sc = 0.3 * det_randn("k5_sh", (2, 768), 3), 0.3 * det_randn("k5_sc", (2, 768), 3)
    n = RMSNorm(768)
    with torch.no_grad():
        n.weight.copy_(w)
        out["k5_out"] = modulate(n(x), sh, sc).numpy()
    # RoPE on [1, 2, 1024, 64]
    q = det_randn("k8_q", (1, 2, 1024, 64), 3)
    rq = m.feat_rope(q)
    out.update(k8_head=rq[0, :, :4].numpy().copy(), k8_tail=rq[0, :, -4:].numpy().copy(), k8_sha=sha(rq))
    # SwiGLU 768 -> 2048 -> 768 on 8 rows
    f = SwiGLUFFN(768, 2048)
    load_det(f, seed=4)
    with torch.no_grad():
        out["k11_out"] = f(det_randn("k11_x", (8, 768), 3)).numpy()
    # attention of block 0 on one sample at full sequence
    a = m.blocks[0].attn
    load_det(a, seed=6)
    with torch.no_grad():
        ao = a(det_randn("k9_x", (1, 1024, 768), 3) * 0.5, rope=m.feat_rope)
    out.update(k9_head=ao[0, :4].numpy().copy(), k9_tail=ao[0, -4:].numpy().copy(), k9_sha=sha(ao),
               k9_norm=np.array(float(ao.double().norm())))
    # timestep embedding + shifted Euler grid
    from models.lightningdit import TimestepEmbedder
    out["k2_emb"] = TimestepEmbedder.timestep_embedding(torch.tensor([0.0, 0.25, 0.9]), 256).numpy()
    o = ode(drift=None, t0=0, t1=1, sampler_type="euler", num_steps=250, atol=1e-6, rtol=1e-3, timestep_shift=0.3)
    out["euler_grid"] = o.t.numpy()
    # logit-normal t draw
    from transport import create_transport
    tr = create_transport("Linear", "velocity", None, None, None, use_cosine_loss=False, use_lognorm=True)
    np.random.seed(11)
    out["lognorm_t_seed11"] = tr.sample_logit_normal(0, 1, size=16).numpy()


# --------------------------------------------------------------------------- VMAE
def gen_mae(out):
    sys.path.insert(0, os.path.join(REF, "tokenizer"))
    from tokenizer import models_mae
    torch.manual_seed(0)
    m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=False, no_cls=True, kl_loss_weight=1e-6, smooth_output=True,
                                            img_size=256)
    out.update(mae_nparams=np.array(sum(p.numel() for p in m.parameters())),
               mae_keys=np.array(sorted(m.state_dict().keys())),
               mae_pos_sha=sha(m.pos_embed), mae_pos_head=m.pos_embed[0, :3].numpy().copy())
    load_det(m, seed=2, skip=("pos_embed", "decoder_pos_embed"))
    m.eval()
    imgs = det_randn("mae_img", (2, 3, 256, 256), 2).clamp(-1, 1)
    for tag, ratio in (("75", 0.75), ("25", 0.25)):
        torch.manual_seed(42)
        noise = torch.rand(2, 1024)
        torch.manual_seed(42)
        with torch.no_grad():
            lat, mask, ids_restore = m.forward_encoder(imgs, ratio)
        srt = np.sort(noise.numpy(), axis=1)
        assert (np.diff(srt, axis=1) > 0).all(), "noise has ties; pick another seed"
        out.update({f"mae{tag}_mask": mask.numpy(), f"mae{tag}_ids_restore": ids_restore.numpy(),
                    f"mae{tag}_lat_head": lat[:, :4].numpy().copy(), f"mae{tag}_lat_sha": sha(lat),
                    f"mae{tag}_lat_norm": np.array(float(lat.double().norm())),
                    f"mae{tag}_lat_shape": np.array(lat.shape)})
        out["mae_noise"] = noise.numpy()
    with torch.no_grad():
        mom = m._encode(imgs)
        z = mom[:, :16]
        rec = m.decode(z).sample
    out.update(mae_moments_head=mom[:, :, :2, :2].numpy().copy(), mae_moments_norm=np.array(float(mom.double().norm())),
               mae_rec_head=rec[:, :, :4, :4].numpy().copy(), mae_rec_norm=np.array(float(rec.double().norm())))
    img8 = torch.clamp(127.5 * rec + 128.0, 0, 255).permute(0, 2, 3, 1).to(torch.uint8).numpy()
    out["mae_img8_sha"] = np.array(hashlib.sha256(img8.tobytes()).hexdigest())
    out["mae_img8_head"] = img8[:, :4, :4].copy()


def gen_mae_train(out):
    """SURVEY 8(f)4 pin: the reference's own pre-training forward ``MaskedAutoencoderViT.forward(imgs, mask_ratio, visible_loss_ratio)``
    (tokenizer/models_mae.py:811-815 -> forward_vanilla :756-790 -> forward_loss :733-754) and its backward, at 128 px with encoder /
    decoder depth 2 (same block classes and widths as mae_for_ldmae_f8d16_prev), f32, eager.  The two random draws of the step are
    recorded as inputs: the masking noise (torch.rand in random_masking, :480) is reproduced by re-seeding, the posterior noise
    (randn_tensor in DiagonalGaussianDistribution.sample, util/misc.py:87-96) is captured from the call itself."""
    from functools import partial
    sys.path.insert(0, os.path.join(REF, "tokenizer"))
    from tokenizer import models_mae
    misc = sys.modules[models_mae.DiagonalGaussianDistribution.__module__]      # `util.misc`, the module object models_mae itself imported
    m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192,
                                        decoder_depth=2, decoder_num_heads=12, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                        latent_dim=16, no_cls=True, kl_loss_weight=1e-3, smooth_output=True)
    load_det(m, seed=6, skip=("pos_embed", "decoder_pos_embed"))
    m.train()
    out["mt_keys"] = np.array(sorted(k for k, p in m.named_parameters() if p.requires_grad))
    imgs = det_randn("img128p", (2, 3, 128, 128), 4).clamp(-1, 1)
    captured = []
    orig = misc.randn_tensor

    def spy(*a, **k):
        t = orig(*a, **k)
        captured.append(t.detach().clone())
        return t
    misc.randn_tensor = spy
    try:
        for tag, ratio, vlr in (("a", 0.75, 0.5), ("b", 0.5, 0.25)):
            captured.clear()
            m.zero_grad(set_to_none=True)
            torch.manual_seed(77)
            noise = torch.rand(2, 256)
            assert (np.diff(np.sort(noise.numpy(), axis=1), axis=1) > 0).all(), "noise has ties; pick another seed"
            torch.manual_seed(77)
            loss, pred, mask, vis, mask_loss, kl = m(imgs, ratio, vlr)
            assert len(captured) == 1, "expected exactly one posterior draw"
            loss.backward()
            out.update({f"mt{tag}_ratio": np.array([ratio, vlr]), f"mt{tag}_noise": noise.numpy(), f"mt{tag}_eps": captured[0].numpy(),
                        f"mt{tag}_loss": np.array([float(loss), float(vis), float(mask_loss), float(kl)], dtype=np.float64),
                        f"mt{tag}_mask": mask.detach().numpy(), f"mt{tag}_pred_head": pred.detach()[:, :6, :24].numpy().copy(),
                        f"mt{tag}_pred_norm": np.array(float(pred.detach().double().norm())),
                        f"mt{tag}_grad_norms": np.array([float(p.grad.double().norm()) for k, p in sorted(m.named_parameters())
                                                         if p.requires_grad], dtype=np.float64),
                        f"mt{tag}_grad_smoother": m.decoder_pred.conv_smoother.weight.grad.numpy().copy(),
                        f"mt{tag}_grad_to_latent_head": m.to_latent.weight.grad[:4, :8].numpy().copy()})
    finally:
        misc.randn_tensor = orig


def gen_vmae_tree(out):
    """The PRE-TRAINING tree's own step (VMAE/models_mae.py:773-807 -> forward_loss :741-771, KL from VMAE/util/misc.py:103-135), which is what
    VMAE/engine_pretrain.py:51-57 runs: same geometry, weights and draws as gen_mae_train, three KL settings -- `fixed_std=None` (the VARIANCE-ONLY
    KL 0.5 sum(var - 1 - logvar): no mean^2 term in this tree, misc.py:118-125), `fixed_std=1e-3` (train_ae.sh:33: 0.5 sum(var / s^2 - 1 - logvar + log s^2),
    :105-116) and 0.5.  Must run in its own process (`--only vmae_tree`): the tree's top-level `util` package collides with LDMAE/tokenizer/util."""
    from functools import partial
    assert "util" not in sys.modules and "models_mae" not in sys.modules, "run with --only vmae_tree (fresh process)"
    sys.path.insert(0, "/root/reference/VMAE")
    import models_mae
    misc = sys.modules[models_mae.DiagonalGaussianDistribution.__module__]
    assert misc.__file__.startswith("/root/reference/VMAE/"), misc.__file__
    imgs = det_randn("img128p", (2, 3, 128, 128), 4).clamp(-1, 1)
    for tag, fixed_std, ratio, vlr in (("n", None, 0.75, 0.5), ("f", 1e-3, 0.25, 0.5), ("h", 0.5, 0.5, 0.25)):
        m = models_mae.MaskedAutoencoderViT(img_size=128, patch_size=8, embed_dim=192, depth=2, num_heads=12, decoder_embed_dim=192,
                                            decoder_depth=2, decoder_num_heads=12, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                            latent_dim=16, no_cls=True, kl_loss_weight=1e-3, smooth_output=True, fixed_std=fixed_std)
        load_det(m, seed=6, skip=("pos_embed", "decoder_pos_embed"))
        m.train()
        captured = []
        orig = misc.randn_tensor

        def spy(*a, **k):
            t = orig(*a, **k)
            captured.append(t.detach().clone())
            return t
        misc.randn_tensor = spy
        try:
            torch.manual_seed(77)
            noise = torch.rand(2, 256)
            torch.manual_seed(77)
            loss, pred, mask, vis, mask_loss, kl, p_loss = m(imgs, ratio, vlr)
            assert len(captured) == 1
            loss.backward()
        finally:
            misc.randn_tensor = orig
        out.update({f"vt{tag}_cfg": np.array([ratio, vlr, -1.0 if fixed_std is None else fixed_std]), f"vt{tag}_noise": noise.numpy(), f"vt{tag}_eps": captured[0].numpy(),
                    f"vt{tag}_loss": np.array([float(loss), float(vis), float(mask_loss), float(kl)], dtype=np.float64),
                    f"vt{tag}_mask": mask.detach().numpy(), f"vt{tag}_pred_norm": np.array(float(pred.detach().double().norm())),
                    f"vt{tag}_grad_norms": np.array([float(p.grad.double().norm()) for k, p in sorted(m.named_parameters()) if p.requires_grad], dtype=np.float64),
                    f"vt{tag}_grad_to_latent_head": m.to_latent.weight.grad[:4, :8].numpy().copy(), f"vt{tag}_grad_to_latent_bias": m.to_latent.bias.grad.numpy().copy()})
    out["vt_keys"] = np.array(sorted(k for k, p in m.named_parameters() if p.requires_grad))


def gen_dit_variants(out):
    """The reference's LightningDiT on the geometries the shipped config does not exercise (eval forward, f32): 'p2' = patch size 2 with
    learn_sigma (the /2 registry entries; x_embedder conv stride 2, unpatchify with p = 2, 2x out channels), 'hd72' = head_dim 72 (XL's heads)
    at width 576.  Same deterministic weights / inputs as tests/test_gpu_dit.py."""
    from models.lightningdit import LightningDiT
    for tag, kw, seed, xs in (("p2", dict(input_size=16, patch_size=2, in_channels=4, hidden_size=192, depth=1, num_heads=3, num_classes=10,
                                          class_dropout_prob=0.1, learn_sigma=True), 3, (2, 4, 16, 16)),
                              ("hd72", dict(input_size=8, patch_size=1, in_channels=16, hidden_size=576, depth=1, num_heads=8, num_classes=10,
                                            class_dropout_prob=0.1), 4, (2, 16, 8, 8))):
        m = LightningDiT(**kw, **FLAGS)
        load_det(m, seed=seed)
        m.eval()
        x, t, y = det_randn("x", xs, 1), torch.tensor([0.2, 0.7]), torch.tensor([1, 5])
        with torch.no_grad():
            o = m(x, t, y)
        out.update({f"dv_{tag}_out": o.numpy(), f"dv_{tag}_norm": np.array(float(o.double().norm()))})


def gen_dit_flags(out):
    """The reference's LightningDiT with each block flag flipped away from the shipped imagenet YAML (f32, eager): per variant a TRAIN-mode
    forward at the tiny geometry (label drop drawn inside forward from a re-seeded torch RNG and recorded), a quadratic loss against a fixed
    target and every parameter gradient; for 'noqk' (the CelebA-HQ configuration) also an eval forward at the real B/1 width (768, 12 heads
    of 64, depth 1).  Same deterministic weights / inputs as the tests."""
    from models.lightningdit import LightningDiT
    B = 2
    xt, t, tgt = det_randn("xt", (B, 16, 8, 8), 7), torch.tensor([0.3, 0.8]), det_randn("tgt", (B, 16, 8, 8), 11)
    for n, (tag, over) in enumerate(DIT_FLAG_VARIANTS.items()):
        kw = {**TINY, **FLAGS, **over}
        m = LightningDiT(**kw)
        load_det(m, seed=20 + n)
        m.train()
        y = torch.tensor([0, 0]) if kw["num_classes"] == 1 else torch.tensor([3, 7])
        torch.manual_seed(40 + n)
        drop = torch.rand(B) < kw["class_dropout_prob"]
        torch.manual_seed(40 + n)
        o = m(xt, t, y)
        loss = ((o - tgt) ** 2).mean()
        loss.backward()
        names = [k for k, p in m.named_parameters() if p.grad is not None]
        grads = dict(m.named_parameters())
        out.update({f"df_{tag}_drop": drop.numpy(), f"df_{tag}_y": y.numpy(), f"df_{tag}_out": o.detach().numpy(), f"df_{tag}_loss": np.array(float(loss)),
                    f"df_{tag}_keys": np.array(sorted(m.state_dict().keys())), f"df_{tag}_grad_names": np.array(names),
                    f"df_{tag}_grad_norm": np.array([float(grads[k].grad.double().norm()) for k in names]),
                    f"df_{tag}_grad_head": np.stack([grads[k].grad.flatten()[:8].numpy().copy() for k in names])})
    kw = dict(input_size=8, patch_size=1, in_channels=16, hidden_size=768, depth=1, num_heads=12, num_classes=1, class_dropout_prob=0.1)
    m = LightningDiT(**kw, **{**FLAGS, "use_qknorm": False})
    load_det(m, seed=31)
    m.eval()
    x = det_randn("x", (2, 16, 8, 8), 1)
    with torch.no_grad():
        o = m(x, torch.tensor([0.2, 0.7]), torch.tensor([0, 0]))
    out.update(df_noqk768_out=o.numpy(), df_noqk768_norm=np.array(float(o.double().norm())))


def gen_mae_archs(out):
    """The registry's other geometries, pinned on the reference itself at depth 1 / 64 px (f32, eager; same recording of the two random draws
    as gen_mae_train): 'dn' = mae_for_ldmae_f8d16 (:1006-1011: down_nonlinear MLP_dim_resize latent maps, 384-wide decoder with 24 heads of 16),
    'h24' = mae_for_ldmae_f8d16_prev_large (:999-1004: 384 wide, 16 heads of 24), 'p16' = the patch-16 tokenizers (mae_for_ldmae_f16d32, :1020-1025)."""
    from functools import partial
    sys.path.insert(0, os.path.join(REF, "tokenizer"))
    from tokenizer import models_mae
    misc = sys.modules[models_mae.DiagonalGaussianDistribution.__module__]
    archs = {"dn": (dict(embed_dim=192, num_heads=12, decoder_embed_dim=384, decoder_num_heads=24, down_nonlinear=True), 9, "img64d", "pepsd"),
             "h24": (dict(embed_dim=384, num_heads=16, decoder_embed_dim=384, decoder_num_heads=16), 8, "img64p", "peps24"),
             "p16": (dict(embed_dim=192, num_heads=12, decoder_embed_dim=192, decoder_num_heads=12, patch_size=16), 7, "img64q", "peps16")}
    captured = []
    orig = misc.randn_tensor

    def spy(*a, **k):
        t = orig(*a, **k)
        captured.append(t.detach().clone())
        return t
    misc.randn_tensor = spy
    try:
        for tag, (kw, seed, iname, _) in archs.items():
            kw = dict(kw)
            ps = kw.pop("patch_size", 8)
            m = models_mae.MaskedAutoencoderViT(img_size=64, patch_size=ps, depth=1, decoder_depth=1, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                                latent_dim=16, no_cls=True, kl_loss_weight=1e-3, smooth_output=True, **kw)
            load_det(m, seed=seed, skip=("pos_embed", "decoder_pos_embed"))
            m.train()
            imgs = det_randn(iname, (2, 3, 64, 64), 4).clamp(-1, 1)
            captured.clear()
            torch.manual_seed(78)
            noise = torch.rand(2, (64 // ps) ** 2)
            assert (np.diff(np.sort(noise.numpy(), axis=1), axis=1) > 0).all(), "noise has ties; pick another seed"
            torch.manual_seed(78)
            loss, pred, mask, vis, mask_loss, kl = m(imgs, 0.75, 0.5)
            assert len(captured) == 1
            loss.backward()
            keys = sorted(k for k, p in m.named_parameters() if p.requires_grad)
            out.update({f"ar_{tag}_keys": np.array(keys), f"ar_{tag}_noise": noise.numpy(), f"ar_{tag}_eps": captured[0].numpy(),
                        f"ar_{tag}_loss": np.array([float(loss), float(vis), float(mask_loss), float(kl)], dtype=np.float64),
                        f"ar_{tag}_mask": mask.detach().numpy(), f"ar_{tag}_pred_head": pred.detach()[:, :6, :24].numpy().copy(),
                        f"ar_{tag}_pred_norm": np.array(float(pred.detach().double().norm())),
                        f"ar_{tag}_grad_norms": np.array([float(dict(m.named_parameters())[k].grad.double().norm()) for k in keys], dtype=np.float64)})
            m.eval()
            with torch.no_grad():
                mom = m._encode(imgs)
                rec = m.decode(mom[:, :16]).sample
            out.update({f"ar_{tag}_moments_head": mom[:, :, :2, :2].numpy().copy(), f"ar_{tag}_moments_norm": np.array(float(mom.double().norm())),
                        f"ar_{tag}_rec_head": rec[:, :, :4, :4].numpy().copy(), f"ar_{tag}_rec_norm": np.array(float(rec.double().norm()))})
    finally:
        misc.randn_tensor = orig


# --------------------------------------------------------------------------- 100-step loss curve, B/1 bs=4
def gen_dataset(out):
    """the reference's own ``ImgLatentDataset`` (datasets/img_latent_dataset.py:16-93) on a tiny generated shard
    directory: shard contents (inputs) + the stats it caches + the (feature, label) items it returns under a seeded numpy / torch
    RNG, for both the moments-sampling configuration of the shipped YAML (latent_norm, sample, multiplier 1.0) and the plain one."""
    import shutil
    import tempfile
    from safetensors.torch import save_file
    from datasets.img_latent_dataset import ImgLatentDataset
    g = torch.Generator().manual_seed(123)
    shards = []
    for s, n in enumerate((5, 3)):
        shards.append({"latents": torch.randn(n, 8, 4, 4, generator=g), "latents_flip": torch.randn(n, 8, 4, 4, generator=g),
                       "labels": torch.randint(0, 1000, (n,), generator=g)})
        for k, v in shards[-1].items():
            out[f"ds_shard{s}_{k}"] = v.numpy()
    for tag, kw in (("a", dict(latent_norm=True, latent_multiplier=1.0, sample=True)), ("b", dict(latent_norm=False, latent_multiplier=0.18215, sample=False)),
                    ("c", dict(latent_norm=True, latent_multiplier=0.5, sample=False))):
        d = tempfile.mkdtemp()
        for s, sh in enumerate(shards):
            save_file(sh, os.path.join(d, f"latents_rank00_shard{s:03d}.safetensors"), metadata={"total_size": str(len(sh["labels"])), "dtype": "torch.float32", "device": "cpu"})
        np.random.seed(7)
        torch.manual_seed(7)
        ds = ImgLatentDataset(d, **kw)
        assert len(ds) == 8
        if kw["latent_norm"]:
            out[f"ds_{tag}_mean"], out[f"ds_{tag}_std"] = ds._latent_mean.numpy(), ds._latent_std.numpy()
        order = [3, 0, 7, 5, 5, 1, 6, 2, 4, 0]
        feats, labels = zip(*[ds[i] for i in order])
        out[f"ds_{tag}_order"], out[f"ds_{tag}_feat"], out[f"ds_{tag}_label"] = np.array(order), torch.stack(feats).numpy(), torch.stack(labels).numpy()
        shutil.rmtree(d)


IMG_CASES = ((300, 200, 64), (130, 97, 64), (64, 64, 64), (517, 389, 64), (97, 260, 32))      # (width, height, target)


def synth_image(w, h, seed):
    """Seeded test image: smooth colour ramps + noise (so that BOX / BICUBIC resampling has structure to act on)."""
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([xx * 255.0 / max(w - 1, 1), yy * 255.0 / max(h - 1, 1), (xx + yy) % 256], axis=-1)
    return np.clip(base + r.randint(-40, 41, size=(h, w, 3)), 0, 255).astype(np.uint8)


def gen_images(out):
    """The reference's own ``center_crop_arr`` (tokenizer/models_mae.py:85-103; what ``img_transform`` :935-950 starts with) on seeded
    images: pins the host-side preprocessing of extract_features.py (BOX halvings, the BICUBIC resize, the crop window)."""
    from PIL import Image
    sys.path.insert(0, os.path.join(REF, "tokenizer"))
    from tokenizer import models_mae
    for i, (w, h, size) in enumerate(IMG_CASES):
        res = models_mae.center_crop_arr(Image.fromarray(synth_image(w, h, 100 + i)), size)
        out[f"crop{i}"] = np.asarray(res)


def ref_style_init(model, seed):
    """Reference init *scheme* (zero adaLN / final, Xavier Linears) with name-keyed
    values so the oracle / HIP side can rebuild the identical start point."""
    import math
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k == "pos_embed":
                continue
            z = det_randn(k, p.shape, seed)
            if "norm" in k and k.endswith("weight"):
                p.fill_(1.0)
            elif k.endswith(".bias") or "adaLN_modulation" in k or k.startswith("final_layer.linear"):
                p.zero_()
            elif k.startswith("y_embedder") or k.startswith("t_embedder"):
                p.copy_(0.02 * z)
            else:
                fan_out, fan_in = p.shape[0], int(np.prod(p.shape[1:]))
                p.copy_(z * math.sqrt(2.0 / (fan_in + fan_out)))


def gen_curve(out, steps=100, B=4):
    import time
    from copy import deepcopy
    from collections import OrderedDict
    from models.lightningdit import LightningDiT_models
    from transport import create_transport
    torch.manual_seed(0)
    m = LightningDiT_models["LightningDiT-B/1"](input_size=32, num_classes=1000, in_channels=16,
                                                 class_dropout_prob=0.1, **FLAGS)
    ref_style_init(m, seed=10)
    ema = deepcopy(m)
    for p in ema.parameters():
        p.requires_grad = False
    m.train()
    tr = create_transport("Linear", "velocity", None, None, None, use_cosine_loss=False, use_lognorm=True)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-4, weight_decay=0, betas=(0.9, 0.95))
    torch.manual_seed(1234)
    np.random.seed(1234)
    losses = []
    x0_sha = None
    for s in range(steps):
        t0 = time.time()
        x = torch.randn(B, 16, 32, 32)
        y = torch.randint(0, 1000, (B,))
        if s == 0:
            st, ns = torch.get_rng_state(), np.random.get_state()
            x0_sha = sha(torch.randn_like(x))
            torch.set_rng_state(st)
            np.random.set_state(ns)
        loss = tr.training_losses(m, x, dict(y=y))["loss"].mean()
        loss.backward()
        opt.step()
        opt.zero_grad()
        with torch.no_grad():
            ep, mp = OrderedDict(ema.named_parameters()), OrderedDict(m.named_parameters())
            for k, p in mp.items():
                ep[k].mul_(0.9999).add_(p.data, alpha=1 - 0.9999)
        losses.append(float(loss))
        print(f"[curve] step {s} loss {losses[-1]:.6f} ({time.time() - t0:.1f}s)", flush=True)
    sdm, sde = m.state_dict(), ema.state_dict()
    probe = ["blocks.0.attn.qkv.weight", "blocks.11.mlp.w3.weight", "final_layer.linear.weight",
             "blocks.5.adaLN_modulation.1.weight", "y_embedder.embedding_table.weight"]
    out.update(curve_losses=np.array(losses), curve_x0_sha=np.array(x0_sha), curve_probe=np.array(probe),
               curve_param_norm=np.array([float(sdm[k].double().norm()) for k in probe]),
               curve_ema_norm=np.array([float(sde[k].double().norm()) for k in probe]),
               curve_param_head=np.stack([sdm[k].flatten()[:8].numpy() for k in probe]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", action="store_true", help="also run the 100-step B/1 bs=4 loss curve (~15 min)")
    ap.add_argument("--only-curve", action="store_true")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--only", default="", help="comma-separated subset of {dit_tiny,kernels,mae,mae_train,mae_archs,dit_variants,dit_flags,vmae_tree,dataset,images}")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    install_shims()
    sys.path.insert(0, REF)
    if not args.only_curve:
        gens = (("dit_tiny", gen_dit_tiny), ("kernels", gen_tables_and_kernels), ("mae", gen_mae), ("mae_train", gen_mae_train),
                ("mae_archs", gen_mae_archs), ("dit_variants", gen_dit_variants), ("dit_flags", gen_dit_flags), ("vmae_tree", gen_vmae_tree), ("dataset", gen_dataset), ("images", gen_images))
        for name, fn in gens:
            if args.only and name not in args.only.split(","):
                continue
            if name == "vmae_tree" and args.only != "vmae_tree":
                continue                      # own process only: `--only vmae_tree` (its `util` package collides with LDMAE/tokenizer/util)
            out = {}
            fn(out)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
            print("wrote", name, sorted(out)[:6], "...")
    if args.curve or args.only_curve:
        out = {}
        gen_curve(out)
        np.savez_compressed(os.path.join(HERE, "curve.npz"), **out)
        print("wrote curve")


if __name__ == "__main__":
    main()
