"""Host-side logic of the driver counterparts (config / dataset / checkpoint plumbing) -- no GPU needed."""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_yaml_is_the_reference_api():
    cfg = yaml.safe_load(open(os.path.join(ROOT, "ldmae_amd/configs/imagenet/lightningdit_b_vmae_f8d16_cfg.yaml")))
    assert set(cfg) == {"ckpt_path", "data", "vae", "model", "train", "optimizer", "transport", "sample"}
    assert cfg["model"] == dict(model_type="LightningDiT-B/1", use_qknorm=True, use_swiglu=True, use_rope=True, use_rmsnorm=True,
                                wo_shift=False, in_chans=16)
    assert cfg["optimizer"] == dict(lr=0.0002, beta2=0.95) and cfg["train"]["global_batch_size"] == 256
    assert cfg["transport"]["use_lognorm"] is True and cfg["sample"]["timestep_shift"] == 0.3 and cfg["sample"]["cfg_scale"] == 10.0
    import ldmae_amd.train_accum as t
    m = t.build_model(cfg)
    assert m.hidden_size == 768 and m.depth == 12 and m.x_embedder.num_patches == 1024 and m.in_channels == 16
    # the reference's second documented configuration (README.md:108-110): unconditional CelebA-HQ, no QK-norm; and both launcher configs
    cel = yaml.safe_load(open(os.path.join(ROOT, "ldmae_amd/configs/celeba_hq/lightningdit_b_vmae_f8d16_cfg.yaml")))
    assert set(cel) == set(cfg) and cel["model"] == {**cfg["model"], "use_qknorm": False}
    assert cel["data"]["num_classes"] == 1 and cel["data"]["name"] == "celebahq" and cel["sample"]["cfg_scale"] == 0 and cel["sample"]["per_proc_batch_size"] == 128
    assert cel["train"]["global_batch_size"] == 1024 and cel["train"]["max_steps"] == 60000 and cel["vae"]["model_name"] == "vmae"
    mc = t.build_model(cel)
    assert mc.depth == 12 and mc.y_embedder.dropout_prob == 0 and mc.y_embedder.embedding_table.weight.shape[0] == 1
    assert not any("q_norm" in k for k in mc.state_dict())
    for n, procs in (("4gpu", 4), ("8gpu", 8)):
        acc = yaml.safe_load(open(os.path.join(ROOT, f"ldmae_amd/configs/accelerator/{n}.yaml")))
        assert acc["num_processes"] == procs and acc["mixed_precision"] == "bf16" and acc["distributed_type"] == "MULTI_GPU"


def test_latent_dataset_roundtrip(tmp_path):
    from safetensors.torch import save_file
    sys.path.insert(0, os.path.join(ROOT, "ldmae_amd"))
    from ldmae_amd.datasets.img_latent_dataset import ImgLatentDataset, SyntheticLatentDataset
    d = tmp_path / "lat_sample"
    d.mkdir()
    torch.manual_seed(0)
    for s in range(2):
        save_file({"latents": torch.randn(5, 32, 4, 4), "latents_flip": torch.randn(5, 32, 4, 4), "labels": torch.arange(5) + 10 * s},
                  str(d / f"latents_rank00_shard{s:03d}.safetensors"), metadata={"total_size": "5"})
    ds = ImgLatentDataset(str(d), latent_norm=True, latent_multiplier=1.0, sample=True)
    assert len(ds) == 10 and os.path.exists(d / "latents_stats.pt")
    x, y = ds[7]
    assert x.shape == (16, 4, 4) and int(y) == 12          # posterior sample of the 32-channel moments
    assert ds._latent_mean.shape == (1, 16, 1, 1)
    s = SyntheticLatentDataset(length=4, channels=16, size=8)
    a, b = s[1], s[1]
    assert torch.equal(a[0], b[0]) and a[0].shape == (16, 8, 8)


def test_latent_dataset_matches_reference_golden(tmp_path):
    """SURVEY 8(f)3 pinned: shards rebuilt from the golden inputs, then our ImgLatentDataset under the same numpy / torch seeds must
    reproduce what the reference's class returned (tests/golden/make_golden.py:gen_dataset imports it): cached stats (np.random.choice
    pick + posterior sample), the per-item flip pick (np.random.uniform), posterior sample (torch.randn), normalisation, multiplier."""
    from safetensors.torch import save_file
    sys.path.insert(0, os.path.join(ROOT, "ldmae_amd"))
    from ldmae_amd.datasets.img_latent_dataset import ImgLatentDataset
    g = np.load(os.path.join(ROOT, "tests", "golden", "dataset.npz"))
    for tag, kw in (("a", dict(latent_norm=True, latent_multiplier=1.0, sample=True)), ("b", dict(latent_norm=False, latent_multiplier=0.18215, sample=False)),
                    ("c", dict(latent_norm=True, latent_multiplier=0.5, sample=False))):
        d = tmp_path / tag
        d.mkdir()
        for s in range(2):
            save_file({k: torch.from_numpy(g[f"ds_shard{s}_{k}"]) for k in ("latents", "latents_flip", "labels")},
                      str(d / f"latents_rank00_shard{s:03d}.safetensors"))
        np.random.seed(7)
        torch.manual_seed(7)
        ds = ImgLatentDataset(str(d), **kw)
        assert len(ds) == 8
        if kw["latent_norm"]:
            assert np.array_equal(ds._latent_mean.numpy(), g[f"ds_{tag}_mean"]) and np.array_equal(ds._latent_std.numpy(), g[f"ds_{tag}_std"])
        items = [ds[int(i)] for i in g[f"ds_{tag}_order"]]
        assert np.array_equal(torch.stack([x for x, _ in items]).numpy(), g[f"ds_{tag}_feat"])
        assert np.array_equal(torch.stack([y for _, y in items]).numpy(), g[f"ds_{tag}_label"])


def test_weight_init_special_case():
    import ldmae_amd.train_accum as t
    from ldmae_amd.models.lightningdit import LightningDiT
    kw = dict(input_size=8, patch_size=1, hidden_size=192, depth=1, num_heads=3, num_classes=10, use_qknorm=True, use_swiglu=True,
              use_rope=True, use_rmsnorm=True)
    big, small = LightningDiT(in_channels=32, **kw), LightningDiT(in_channels=16, **kw)
    ck = {"model": {"module." + k: v for k, v in big.state_dict().items()}}
    t.load_weights_with_shape_check(small, ck, rank=1)
    assert torch.equal(small.x_embedder.proj.weight[:, :16], big.x_embedder.proj.weight[:, :16])
    assert torch.equal(small.blocks[0].attn.qkv.weight, big.blocks[0].attn.qkv.weight)


def test_vmae_pretrain_schedule_and_param_groups():
    """Host logic of the VMAE pre-training driver: the half-cycle cosine schedule with warm-up (reference VMAE/util/lr_sched.py:9-18) and
    the timm weight-decay split as contiguous groups of the flat parameter slab."""
    import math
    from ldmae_amd.vmae_pretrain import cosine_lr, no_decay
    from ldmae_amd.optim import FlatParams
    assert cosine_lr(0.0, 1e-3, 0.0, 40, 400) == 0.0 and abs(cosine_lr(20.0, 1e-3, 0.0, 40, 400) - 5e-4) < 1e-12
    assert abs(cosine_lr(40.0, 1e-3, 1e-5, 40, 400) - 1e-3) < 1e-12 and abs(cosine_lr(400.0, 1e-3, 1e-5, 40, 400) - 1e-5) < 1e-12
    mid = cosine_lr(220.0, 1e-3, 0.0, 40, 400)
    assert abs(mid - 0.5e-3 * (1 + math.cos(math.pi * 0.5))) < 1e-12 and cosine_lr(7.0, 3e-4, 0, 40, 400, fixed_lr=True) == 3e-4
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.LayerNorm(16), torch.nn.Linear(16, 4))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    flat = FlatParams(net, group_fn=no_decay)
    assert set(flat.groups) == {0, 1} and flat.groups[0][0] == 0 and flat.groups[0][1] == flat.groups[1][0] and flat.groups[1][1] == flat.n_trainable
    for n, p in net.named_parameters():
        lo, hi = flat.groups[no_decay(n, p)]
        assert lo <= flat.offsets[n][0] < hi and torch.equal(p.data, before[n])           # values unchanged, each name inside its group


def _synth_image(w, h, seed):          # same generator as tests/golden/make_golden.py:synth_image
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([xx * 255.0 / max(w - 1, 1), yy * 255.0 / max(h - 1, 1), (xx + yy) % 256], axis=-1)
    return np.clip(base + r.randint(-40, 41, size=(h, w, 3)), 0, 255).astype(np.uint8)


def test_center_crop_matches_reference_golden():
    """Host preprocessing of extract_features.py: our center_crop_arr against the reference's own function on the same seeded images
    (tests/golden/make_golden.py:gen_images imports it; BOX halvings + BICUBIC resize + crop window), then the full img_transform."""
    from PIL import Image
    from ldmae_amd.tokenizer.models_mae import ImageTransform, center_crop_arr
    g = np.load(os.path.join(ROOT, "tests", "golden", "images.npz"))
    cases = ((300, 200, 64), (130, 97, 64), (64, 64, 64), (517, 389, 64), (97, 260, 32))
    for i, (w, h, size) in enumerate(cases):
        img = Image.fromarray(_synth_image(w, h, 100 + i))
        got = np.asarray(center_crop_arr(img, size))
        assert got.shape == (size, size, 3) and np.array_equal(got, g[f"crop{i}"]), i
        plain, flipped = ImageTransform(0.0, size)(img), ImageTransform(1.0, size)(img)
        want = torch.from_numpy(g[f"crop{i}"].copy()).permute(2, 0, 1).float() / 255.0
        assert torch.equal(plain, (want - 0.5) / 0.5) and torch.equal(flipped, plain.flip(-1))
        assert plain.dtype == torch.float32 and float(plain.min()) >= -1.0 and float(plain.max()) <= 1.0


def test_image_folder_order_and_labels(tmp_path):
    """torchvision ImageFolder semantics without torchvision: classes = sorted dir names, samples sorted inside a class, non-images skipped."""
    from PIL import Image
    from ldmae_amd.datasets.image_folder import ImageFolder
    for c, names in (("n02", ("b.png", "a.JPEG", "notes.txt")), ("n01", ("z.png",)), ("n10", ("sub/k.bmp", "c.png"))):
        for n in names:
            p = tmp_path / c / n
            p.parent.mkdir(parents=True, exist_ok=True)
            if n.endswith(".txt"):
                p.write_text("x")
            else:
                Image.fromarray(_synth_image(20, 12, len(str(p)))).save(str(p), format="PNG" if n.lower().endswith("jpeg") else None)
    ds = ImageFolder(str(tmp_path))
    assert ds.classes == ["n01", "n02", "n10"] and ds.class_to_idx == {"n01": 0, "n02": 1, "n10": 2}
    rel = [(os.path.relpath(p, tmp_path), t) for p, t in ds.samples]
    assert rel == [("n01/z.png", 0), ("n02/a.JPEG", 1), ("n02/b.png", 1), ("n10/c.png", 2), ("n10/sub/k.bmp", 2)]
    img, t = ds[3]
    assert img.size == (20, 12) and img.mode == "RGB" and t == 2


def test_mae_arch_registry_has_every_reference_constructor():
    """tokenizer/models_mae.py:977-1083: sixteen constructor names + four aliases.  All build (on the meta device: geometry only); the two
    `down_nonlinear` archs carry MLP_dim_resize latent maps with the reference's state-dict keys (:232-242, 311-314)."""
    from ldmae_amd.tokenizer import models_mae as mm
    names = ["mae_for_ldmae", "mae_for_ldmae_f8d32", "mae_for_ldmae_f8d16_prev", "mae_for_ldmae_f8d16_prev_large", "mae_for_ldmae_f8d16",
             "mae_for_ldmae_f8d16_flexible", "mae_for_ldmae_f16d32", "mae_for_ldmae_f16d32_large", "mae_for_ldmae_f8d32_flexible", "mae_for_ldmae_16d",
             "mae_vit_base_patch16_dec512d8b", "mae_vit_base_patch16_dec128d8b", "mae_vit_large_patch16_dec512d8b", "mae_vit_huge_patch14_dec512d8b",
             "mae_vit_base_patch16", "mae_vit_large_patch16", "mae_vit_huge_patch14", "mae_vit_base_patch16_128",
             "mae_for_ldmae_f8d16_small", "mae_for_ldmae_f8d16_asym_small"]       # the last two: the pre-training tree's registry only (VMAE/models_mae.py:1036-1048)
    geo = {"mae_for_ldmae_f16d32_large": (384, 12, 384, 64), "mae_vit_base_patch16_dec128d8b": (768, 12, 128, 196), "mae_vit_large_patch16": (1024, 24, 512, 196),
           "mae_for_ldmae_f8d16_prev": (192, 12, 192, 784), "mae_for_ldmae_f8d16": (192, 12, 384, 784),
           "mae_for_ldmae_f8d16_small": (96, 12, 96, 784), "mae_for_ldmae_f8d16_asym_small": (96, 12, 192, 784)}
    for n in names:
        f = getattr(mm, n)
        if n in ("mae_for_ldmae_f8d16", "mae_for_ldmae_f8d16_flexible"):
            m = f(kl_loss_weight=1e-6)
            sd = m.state_dict()
            assert sd["to_latent.layers.0.weight"].shape == (64, 192) and sd["to_latent.layers.2.weight"].shape == (32, 64)
            assert sd["from_latent.layers.0.weight"].shape == (64, 16) and sd["from_latent.layers.2.weight"].shape == (192, 64)
            assert sd["decoder_embed.weight"].shape == (384, 192) and m.decoder_blocks[0].attn.num_heads == 24
        if n in geo:
            with torch.device("meta"):
                try:
                    m = f()
                except Exception:            # initialize_weights fills real tensors: geometry checked on the CPU for the small ones only
                    m = None
            if m is None and geo[n][0] <= 384:
                m = f()
            if m is not None:
                assert (m.pos_embed.shape[-1], len(m.blocks), m.decoder_pos_embed.shape[-1], m.pos_embed.shape[1]) == geo[n]


# ----------------------------------------------------------------------------- VMAE pre-training driver: host side (main_pretrain.py:111-192, misc.py:488-531)
def _png_tree(root, n=6, size=(48, 40)):
    from PIL import Image
    rng = np.random.default_rng(0)
    paths = []
    for i in range(n):
        d = root / ("a" if i % 2 else "b")
        d.mkdir(parents=True, exist_ok=True)
        p = d / f"img_{i}.png"
        Image.fromarray(rng.integers(0, 256, size=(size[1], size[0], 3), dtype=np.uint8)).save(p)
        paths.append(str(p))
    return sorted(paths)


def test_vmae_pretrain_image_input(tmp_path):
    """The three dataset branches of main_pretrain.py:111-192 on PIL alone: 'imagenet' in the path -> ImageFolder(<path>/train) with the
    random-resized-crop + flip transform; any other tree -> every image, resized to a square, a random label."""
    import argparse
    from ldmae_amd import vmae_pretrain as vp
    tree = tmp_path / "pics"
    paths = _png_tree(tree)
    ds = vp.get_dataset(argparse.Namespace(synthetic=False, data_path=str(tree), input_size=32))
    assert isinstance(ds, vp.FlatImageTree) and ds.paths == paths and len(ds) == 6
    x, y = ds[0]
    assert x.shape == (3, 32, 32) and x.dtype == torch.float32 and -1.0 <= float(x.min()) and float(x.max()) <= 1.0 and 0 <= y < 1000
    from PIL import Image
    ref = torch.from_numpy(np.asarray(Image.open(paths[0]).convert("RGB").resize((32, 32), Image.BILINEAR)).copy()).permute(2, 0, 1).float() / 255
    assert torch.equal(x, (ref - 0.5) / 0.5)
    inet = tmp_path / "imagenet_x"
    _png_tree(inet / "train")
    ds2 = vp.get_dataset(argparse.Namespace(synthetic=False, data_path=str(inet), input_size=32))
    assert type(ds2).__name__ == "ImageFolder" and ds2.classes == ["a", "b"] and len(ds2) == 6
    torch.manual_seed(0)
    crops = [ds2[i][0] for i in range(6)]
    assert all(c.shape == (3, 32, 32) and torch.isfinite(c).all() for c in crops)
    # the crop box: area fraction in [0.75, 1], aspect ratio in [3/4, 4/3], inside the image
    t = vp.RandomResizedCropFlip(32)
    for _ in range(50):
        top, left, ch, cw = t._box(48, 40)
        assert 0 <= top and 0 <= left and top + ch <= 40 and left + cw <= 48
        assert 0.74 <= ch * cw / (48 * 40) <= 1.0 + 1e-9 and 0.74 <= cw / ch <= 1.35


def test_vmae_pretrain_pos_embed_resize():
    from ldmae_amd import vmae_pretrain as vp
    pe = torch.randn(1, 16 * 16, 24)
    out = vp.resize_pos_embed(pe, 32)
    ref = torch.nn.functional.interpolate(pe.reshape(1, 16, 16, 24).permute(0, 3, 1, 2), size=(32, 32), mode="bilinear", align_corners=False)
    assert out.shape == (1, 1024, 24) and torch.equal(out, ref.permute(0, 2, 3, 1).reshape(1, -1, 24))
    assert torch.equal(vp.resize_pos_embed(pe, 16), pe)


def test_resume_from_a_reference_style_optimizer_state(tmp_path):
    """A checkpoint as the REFERENCE's save_model writes it (VMAE/util/misc.py:474-481): 'optimizer' = torch.optim.AdamW.state_dict() over
    timm's [no_decay, decay] groups (main_pretrain.py:258-259).  load_model maps its moments by parameter order into the slabs (step count
    included); a state that fits nothing is refused BEFORE the optimizer is touched, and load_model then still restores model / epoch / scaler."""
    import argparse
    from ldmae_amd import vmae_pretrain as vp
    from ldmae_amd.optim import AdamWEMA, FlatParams

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.pos_embed = torch.nn.Parameter(torch.zeros(1, 4, 8), requires_grad=False)
            self.decoder_pos_embed = torch.nn.Parameter(torch.zeros(1, 4, 8), requires_grad=False)
            self.a, self.n, self.b = torch.nn.Linear(8, 16), torch.nn.LayerNorm(16), torch.nn.Linear(16, 4)
    torch.manual_seed(0)
    ref = Net()
    named = [(n, p) for n, p in ref.named_parameters() if p.requires_grad]
    nd = [p for n, p in named if p.ndim <= 1 or n.endswith(".bias")]
    dc = [p for n, p in named if not (p.ndim <= 1 or n.endswith(".bias"))]
    topt = torch.optim.AdamW([{"params": nd, "weight_decay": 0.0}, {"params": dc, "weight_decay": 0.05}], lr=1e-3, betas=(0.9, 0.95))
    for _ in range(3):
        topt.zero_grad()
        ref.b(ref.n(ref.a(torch.randn(5, 8)))).pow(2).sum().backward()
        topt.step()
    path = tmp_path / "checkpoint-7.pth"
    torch.save({"model": ref.state_dict(), "optimizer": topt.state_dict(), "epoch": 7, "scaler": {"scale": 1024.0, "_growth_tracker": 5}, "args": None}, path)
    net = Net()
    opt = AdamWEMA(net, flat=FlatParams(net, group_fn=vp.no_decay), weight_decay=0.05, group_weight_decay={0: 0.05, 1: 0.0})    # as vp.build_optimizer
    scaler = vp.LossScaler()
    args = argparse.Namespace(resume=str(path), start_epoch=0)
    msgs = []
    assert vp.load_model(args, net, opt, scaler, log=msgs.append) == 8
    assert opt.step_count == 3 and scaler.scale == 1024.0
    tstate = {id(p): topt.state[p] for g in topt.param_groups for p in g["params"]}
    for n, p in named:
        o, k = opt.flat.offsets[n]
        assert torch.equal(opt.m[o:o + k], tstate[id(p)]["exp_avg"].reshape(-1)) and torch.equal(opt.v[o:o + k], tstate[id(p)]["exp_avg_sq"].reshape(-1)), n
        assert torch.equal(dict(net.named_parameters())[n], p)
    assert torch.equal(opt.ema, opt.flat.params)                      # a torch state has no EMA: restarted from the loaded weights
    # one list of model.parameters() (LDMAE/train_accum.py:121) is understood too
    t1 = torch.optim.AdamW([p for _, p in named], lr=1e-3)
    t1.zero_grad(); ref.b(ref.n(ref.a(torch.randn(5, 8)))).pow(2).sum().backward(); t1.step()
    opt.load_state_dict(t1.state_dict())
    assert opt.step_count == 1
    # ... and back: the state this optimizer WRITES is a torch.optim.AdamW.state_dict() the reference's load_model can feed to its optimizer
    # (misc.py:523-525), group for group, and reading it again reproduces the slabs bit for bit
    opt.load_state_dict(topt.state_dict())
    out = opt.torch_adamw_state_dict()
    assert [len(g["params"]) for g in out["param_groups"]] == [len(nd), len(dc)] and [g["weight_decay"] for g in out["param_groups"]] == [0.0, 0.05]
    assert set(out["param_groups"][0]) == set(topt.state_dict()["param_groups"][0])
    t2 = torch.optim.AdamW([{"params": nd, "weight_decay": 0.0}, {"params": dc, "weight_decay": 0.05}], lr=1e-3, betas=(0.9, 0.95))
    t2.load_state_dict(out)
    for g_a, g_b in zip(t2.param_groups, topt.param_groups):
        for p_a, p_b in zip(g_a["params"], g_b["params"]):
            assert all(torch.equal(t2.state[p_a][k], topt.state[p_b][k]) for k in ("exp_avg", "exp_avg_sq")) and float(t2.state[p_a]["step"]) == 3
    net2 = Net()
    opt2 = AdamWEMA(net2, flat=FlatParams(net2, group_fn=vp.no_decay))
    opt2.load_state_dict(out)
    assert opt2.step_count == 3 and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    single = AdamWEMA(Net()).torch_adamw_state_dict()                 # one list of parameters, never stepped: torch's empty state
    assert len(single["param_groups"]) == 1 and len(single["param_groups"][0]["params"]) == len(named) and single["state"] == {}
    # states that fit nothing: refused with the optimizer untouched; load_model logs and goes on
    m0, step0 = opt.m.clone(), opt.step_count
    bad = topt.state_dict()
    bad["state"][0]["exp_avg"] = torch.zeros(3)
    for broken in (bad, {"step": 9, "m": opt.m}, {"state": {}, "param_groups": [{"params": [0]}, {"params": [1]}, {"params": [2]}]}):
        with pytest.raises(RuntimeError):
            opt.load_state_dict(broken)
        assert opt.step_count == step0 and torch.equal(opt.m, m0)
    torch.save({"model": ref.state_dict(), "optimizer": bad, "epoch": 2, "scaler": None, "args": None}, path)
    msgs.clear()
    assert vp.load_model(args, net, opt, scaler, log=msgs.append) == 3 and any("not restored" in str(m) for m in msgs)


def test_vmae_pretrain_cli_takes_the_references_flag_set():
    """VMAE/main_pretrain.py:37-91's flags parse here; what this driver cannot do is refused by name BEFORE anything touches the GPU: the LPIPS term
    (`--perceptual_loss_ratio`, needs the VGG weights), decoder fine-tuning (`--tune_decoder`), and the two forms no shipped script uses."""
    from ldmae_amd import vmae_pretrain as vp
    for bad in (["--perceptual_loss_ratio", "0.5"], ["--tune_decoder"], ["--pred_with_conv"], ["--gradual_resol"], ["--device", "cpu"]):
        with pytest.raises(SystemExit) as e:
            vp.main(["--synthetic", "--no_cls", "--smooth_output", "--fixed_std", "1e-3", "--log_dir", "x", "--pin_mem", "--world_size", "8", "--local-rank", "0",
                     "--dist_url", "env://"] + bad)
        assert e.value.code == 2


def test_sample_folder_name_follows_the_reference_rule():
    """inference.py:45-52: <model>-ckpt-<stem>-<method>-<steps>, lower-cased; the guidance suffix only when cfg_scale > 1."""
    import yaml
    from ldmae_amd.inference import sample_folder_name, DEMO_LABELS
    cfg = yaml.safe_load(open(os.path.join(ROOT, "ldmae_amd/configs/imagenet/lightningdit_b_vmae_f8d16_cfg.yaml")))
    s = cfg["sample"]
    base = f"lightningdit-b-1-ckpt-0100000-{s['sampling_method']}-{s['num_sampling_steps']}".lower()
    assert sample_folder_name(cfg, cfg["ckpt_path"], 1.0) == base
    assert sample_folder_name(cfg, cfg["ckpt_path"]) == base + f"-interval{s['cfg_interval_start']:.2f}-cfg{s['cfg_scale']:.2f}-shift{s['timestep_shift']:.2f}"
    cel = yaml.safe_load(open(os.path.join(ROOT, "ldmae_amd/configs/celeba_hq/lightningdit_b_vmae_f8d16_cfg.yaml")))
    assert sample_folder_name(cel, cel["ckpt_path"]).startswith("lightningdit-b-1-ckpt-0060000-") and "cfg" not in sample_folder_name(cel, cel["ckpt_path"])
    assert DEMO_LABELS == [975, 3, 207, 387, 388, 88, 979, 279]


def test_pe_reset_writes_the_resized_checkpoint(tmp_path):
    """Stage 2 of VMAE/train_ae.sh (pe_reset.py:20-77): both position embeddings resized to the target grid (bilinear, the resume path's rule),
    everything else untouched, written as <checkpoint>_pe.pth; both spellings of the path flag (pe_reset.py:90 / train_ae.sh:66) parse."""
    from ldmae_amd import pe_reset
    from ldmae_amd.tokenizer import models_mae
    from ldmae_amd.vmae_pretrain import resize_pos_embed
    torch.manual_seed(0)
    m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=True, no_cls=True, img_size=64, smooth_output=True, kl_loss_weight=None)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["pos_embed"] = torch.randn_like(sd["pos_embed"])
    sd["decoder_pos_embed"] = torch.randn_like(sd["decoder_pos_embed"])
    path = tmp_path / "checkpoint-90.pth"
    torch.save({"model": sd, "epoch": 90}, path)
    out = pe_reset.main(["--ckpt_dir", str(path), "--input_size", "128"])
    assert out == str(tmp_path / "checkpoint-90_pe.pth")
    new = torch.load(out, map_location="cpu")
    assert new["epoch"] == 90 and new["model"]["pos_embed"].shape == (1, 256, sd["pos_embed"].shape[2])
    assert torch.equal(new["model"]["pos_embed"], resize_pos_embed(sd["pos_embed"], 16))
    assert torch.equal(new["model"]["decoder_pos_embed"], resize_pos_embed(sd["decoder_pos_embed"], 16))
    assert all(torch.equal(new["model"][k], v) for k, v in sd.items() if "pos_embed" not in k)
    big = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=True, no_cls=True, img_size=128, smooth_output=True, kl_loss_weight=None)
    assert not big.load_state_dict(new["model"], strict=False).unexpected_keys
    same = pe_reset.main(["--chkpt_dir", str(path), "--input_size", "64"])          # same grid: a plain copy
    assert torch.equal(torch.load(same, map_location="cpu")["model"]["pos_embed"], sd["pos_embed"])


def test_launcher_scripts_mirror_the_references(tmp_path):
    """run_train.sh / run_inference.sh / run_fast_inference.sh / run_extract_feature.sh (shared body _launch.sh) and train_ae.sh: executable, valid
    bash, and -- run here against a stand-in `python` that prints its arguments -- the command each one starts: the driver, the process count from
    GPUS_PER_NODE x WORLD_SIZE (machines), RANK as the machine index, the reference script's own default port, PRECISION exported, extra
    arguments handed on to the driver."""
    import subprocess
    d = os.path.join(ROOT, "ldmae_amd")
    for name in ("run_train.sh", "run_inference.sh", "run_fast_inference.sh", "run_extract_feature.sh", "train_ae.sh"):
        assert os.access(os.path.join(d, name), os.X_OK), name
    for name in ("run_train.sh", "run_inference.sh", "run_fast_inference.sh", "run_extract_feature.sh", "train_ae.sh", "_launch.sh"):
        assert subprocess.run(["bash", "-n", os.path.join(d, name)]).returncode == 0, name
    shim = tmp_path / "bin"
    shim.mkdir()
    (shim / "python").write_text('#!/bin/bash\necho "ARGS $@"\necho "ENV PRECISION=$PRECISION RANK=${RANK:-unset} WORLD_SIZE=${WORLD_SIZE:-unset}"\n')
    os.chmod(shim / "python", 0o755)
    env = {"PATH": f"{shim}:/usr/bin:/bin", "LDMAE_USE_TORCHRUN": "1"}

    def run(name, *args, **extra):
        r = subprocess.run(["bash", os.path.join(d, name), *args], env={**env, **extra}, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return r.stdout
    out = run("run_train.sh", "configs/imagenet/x.yaml", "--synthetic", GPUS_PER_NODE="4", WORLD_SIZE="2", RANK="1", PRECISION="fp32")
    assert "--nnodes 2 --node-rank 1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 1235 train_accum.py --config configs/imagenet/x.yaml --synthetic" in out
    assert "ENV PRECISION=fp32 RANK=unset WORLD_SIZE=unset" in out           # WORLD_SIZE / RANK meant machines: not leaked to the launcher
    out = run("run_inference.sh", "c.yaml")
    assert "--nnodes 1 --node-rank 0 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 1237 inference.py --config c.yaml" in out and "PRECISION=bf16" in out
    out = run("run_fast_inference.sh", "c.yaml", GPUS_PER_NODE="8", WORLD_SIZE="4", MASTER_PORT="2000")
    assert "--nnodes 1 --node-rank 0 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2000 inference.py --config c.yaml --demo" in out
    out = run("run_extract_feature.sh", "c.yaml", GPUS_PER_NODE="2")
    assert "--nproc-per-node 2 --master-addr 127.0.0.1 --master-port 1235 extract_features.py --config c.yaml" in out
    out = run("train_ae.sh", GPUS_PER_NODE="2", DATA_PATH="/d/imagenet", OUT=str(tmp_path / "w"))
    assert "vmae_pretrain.py --batch_size 128 --no_cls --accum_iter 2" in out and "--fixed_std 1e-3" in out and "--data_path /d/imagenet" in out
    assert f"pe_reset.py --model_name mae_for_ldmae_f8d16_prev --ckpt_dir {tmp_path / 'w'}/checkpoint-90.pth" in out and "perceptual" not in out.split("Stage 3")[0]
