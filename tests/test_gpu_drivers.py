"""The train / sampling driver counterparts end to end on one GPU (tiny geometry, synthetic latents)."""
import copy
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tiny_cfg(tmp_path, dataset="imagenet"):
    cfg = yaml.safe_load(open(os.path.join(ROOT, f"ldmae_amd/configs/{dataset}/lightningdit_b_vmae_f8d16_cfg.yaml")))
    cfg = copy.deepcopy(cfg)
    cfg["data"].update(image_size=64, num_workers=0)           # 64 / 8 = 8x8 latents -> 64 tokens
    cfg["train"].update(global_batch_size=8, output_dir=str(tmp_path), exp_name="t", log_every=2, ckpt_every=3, max_steps=3)
    return cfg


@pytest.mark.parametrize("dataset", ["imagenet", "celeba_hq"])      # both documented run_train.sh configurations (README.md:104-110)
def test_train_driver_checkpoint_resume_and_sampler(tmp_path, monkeypatch, dataset):
    import ldmae_amd.train_accum as t
    from ldmae_amd.models import lightningdit as L
    monkeypatch.setitem(L.LightningDiT_models, "LightningDiT-B/1", lambda **kw: L.LightningDiT(depth=2, hidden_size=192, patch_size=1, num_heads=3, **kw))
    cfg = tiny_cfg(tmp_path, dataset)
    torch.manual_seed(0)
    np.random.seed(0)
    model, opt = t.do_train(cfg, synthetic=True)
    ck_path = tmp_path / "t" / "checkpoints" / "0000003.pt"
    assert ck_path.exists() and (tmp_path / "t" / "log.txt").exists()
    ck = torch.load(ck_path, map_location="cpu")
    assert set(ck) == {"model", "ema", "opt", "config"}                       # train_accum.py:275-280
    assert set(ck["model"]) == set(model.state_dict()) == set(ck["ema"])
    assert opt.step_count == 3 and all(torch.isfinite(v).all() for v in ck["model"].values())
    # EMA moved only slightly away from its start, parameters moved more
    w, e = ck["model"]["blocks.0.attn.qkv.weight"], ck["ema"]["blocks.0.attn.qkv.weight"]
    assert 0 < float((w - e).abs().max()) < 1e-2
    # resume picks the checkpoint up and continues to step 5
    cfg2 = tiny_cfg(tmp_path, dataset)
    cfg2["train"].update(resume=True, max_steps=5, ckpt_every=100)
    model2, opt2 = t.do_train(cfg2, synthetic=True)
    assert opt2.step_count == 2
    # sampling: EMA weights -> CFG Euler sampler -> finite latents of the right shape
    from ldmae_amd.inference import build_sampler, sample_latents
    m = t.build_model(cfg)
    m.load_state_dict(ck["ema"])
    m = m.cuda().eval()
    cfg["sample"]["num_sampling_steps"] = 5
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lat, y = sample_latents(m, build_sampler(cfg), 4, 4.0 if dataset == "imagenet" else cfg["sample"]["cfg_scale"], 0.1, torch.device("cuda"),
                                cfg["data"]["num_classes"])
    assert lat.shape == (4, 16, 8, 8) and torch.isfinite(lat).all()


def test_sampler_skips_the_unconditional_half_only_where_it_is_unused(tmp_path):
    """sample_latents runs the steps below the guidance-interval start on the conditional half alone (forward_with_cfg applies no guidance
    there and the kept samples never see the unconditional half's output).  The returned latents must equal, bit for bit, what the sampler
    gives with the reference's forward_with_cfg on the doubled batch at EVERY step, and the shortcut must actually be taken."""
    import ldmae_amd.train_accum as t
    from ldmae_amd.inference import build_sampler, sample_latents
    cfg = tiny_cfg(tmp_path)
    torch.manual_seed(3)
    m = t.build_model(cfg).cuda().eval()
    with torch.no_grad():
        for n, p in m.named_parameters():                         # the zero-initialised adaLN / final layers would make every output equal
            if "adaLN_modulation" in n or n.startswith("final_layer.linear"):
                p.copy_(torch.randn_like(p) * 0.05)
    cfg["sample"].update(num_sampling_steps=12, timestep_shift=0.3)
    fn = build_sampler(cfg)
    halves, fulls = [], []
    orig_fwd, orig_cfg = m.forward, m.forward_with_cfg
    m.forward = lambda x, tt, y: (halves.append(len(x)), orig_fwd(x, tt, y))[1]
    m.forward_with_cfg = lambda *a, **k: (fulls.append(len(a[0])), orig_cfg(*a, **k))[1]
    g = torch.Generator(device="cuda").manual_seed(5)
    lat, y = sample_latents(m, fn, 4, 4.0, 0.3, torch.device("cuda"), cfg["data"]["num_classes"], generator=g)
    assert halves.count(4) >= 2 and len(fulls) >= 2 and halves.count(4) + len(fulls) == 11      # 12 grid points = 11 steps: some on 4 samples, the rest on 8
    # the reference's loop: forward_with_cfg at every step
    m.forward, m.forward_with_cfg = orig_fwd, orig_cfg
    g = torch.Generator(device="cuda").manual_seed(5)
    z = torch.randn(4, m.in_channels, 8, 8, device="cuda", generator=g)
    y2 = torch.randint(0, cfg["data"]["num_classes"], (4,), device="cuda", generator=g)
    zz, yy = torch.cat([z, z]), torch.cat([y2, torch.full((4,), cfg["data"]["num_classes"], device="cuda")])
    with torch.no_grad():
        ref = fn(zz, m.forward_with_cfg, y=yy, cfg_scale=4.0, cfg_interval=True, cfg_interval_start=0.3)[-1][:4]
    assert torch.equal(y, y2) and torch.equal(lat, ref)
    # bf16 autocast: the batched adaLN GEMM needs a batch that is a multiple of 8.  n = 4: the half batch (4) would take the per-block f32
    # adaLN path and the doubled batch (8) the batched bf16 one -> the shortcut must NOT be taken; n = 8: both batched -> taken.  Either way
    # the latents equal the reference's loop bit for bit.
    for n, taken in ((4, False), (8, True)):
        halves.clear(); fulls.clear()
        m.forward = lambda x, tt, y: (halves.append(len(x)), orig_fwd(x, tt, y))[1]
        m.forward_with_cfg = lambda *a, **k: (fulls.append(len(a[0])), orig_cfg(*a, **k))[1]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            g = torch.Generator(device="cuda").manual_seed(5)
            lat, y = sample_latents(m, fn, n, 4.0, 0.3, torch.device("cuda"), cfg["data"]["num_classes"], generator=g)
            assert (halves.count(n) >= 2) == taken and halves.count(n) + len(fulls) == 11, (n, halves, fulls)
            m.forward, m.forward_with_cfg = orig_fwd, orig_cfg
            g = torch.Generator(device="cuda").manual_seed(5)
            z = torch.randn(n, m.in_channels, 8, 8, device="cuda", generator=g)
            y2 = torch.randint(0, cfg["data"]["num_classes"], (n,), device="cuda", generator=g)
            zz, yy = torch.cat([z, z]), torch.cat([y2, torch.full((n,), cfg["data"]["num_classes"], device="cuda")])
            with torch.no_grad():
                ref = fn(zz, m.forward_with_cfg, y=yy, cfg_scale=4.0, cfg_interval=True, cfg_interval_start=0.3)[-1][:n]
        assert torch.equal(y, y2) and torch.equal(lat, ref), n


@pytest.mark.parametrize("dataset,cfg_scale", [("imagenet", 4.0), ("celeba_hq", 0)])
def test_do_sample_end_to_end_writes_pngs(tmp_path, monkeypatch, dataset, cfg_scale):
    """SURVEY 8(f)1 end to end (reference inference.py:264-299): EMA checkpoint -> shifted-grid Euler with CFG -> latent de-normalisation
    (z * std / multiplier + mean) -> VMAE decode_to_images -> one PNG per sample, indexed i * world + rank + total; the PNG encoding runs
    on the writer thread.  'celeba_hq': the reference's unconditional configuration (cfg_scale 0 -> the branch without the doubled batch,
    inference.py:282-284; num_classes 1; no QK-norm)."""
    from PIL import Image
    import ldmae_amd.inference as inf
    import ldmae_amd.train_accum as t
    from ldmae_amd.models import lightningdit as L
    from ldmae_amd.tokenizer import models_mae
    monkeypatch.setitem(L.LightningDiT_models, "LightningDiT-B/1", lambda **kw: L.LightningDiT(depth=2, hidden_size=192, patch_size=1, num_heads=3, **kw))
    cfg = tiny_cfg(tmp_path, dataset)
    cfg["data"].update(data_path=str(tmp_path / "feat"), latent_multiplier=1.0)       # 'sample' key present -> the dir gets the _sample suffix
    cfg["vae"]["weight_path"] = str(tmp_path / "vmae.pth")
    cfg["sample"].update(num_sampling_steps=2, per_proc_batch_size=4, fid_num=8, cfg_scale=cfg_scale)
    torch.manual_seed(0)
    dit = t.build_model(cfg)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():                                  # a zero-initialised final layer would make every sample equal its noise
        for n, p in dit.named_parameters():
            if "adaLN_modulation" in n or n.startswith("final_layer.linear"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    torch.save({"ema": dit.state_dict(), "model": dit.state_dict()}, tmp_path / "ckpt.pt")
    vae = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=True, no_cls=True, kl_loss_weight=True, smooth_output=True, img_size=64)
    torch.save({"model": vae.state_dict()}, cfg["vae"]["weight_path"])
    os.makedirs(str(tmp_path / "feat_sample"))
    torch.save({"mean": torch.randn(1, 16, 1, 1, generator=g) * 0.1, "std": torch.rand(1, 16, 1, 1, generator=g) + 0.5},
               tmp_path / "feat_sample" / "latents_stats.pt")
    out = inf.do_sample(cfg, str(tmp_path / "ckpt.pt"))
    # the directory rule of inference.py:45-58: <output_dir>/<exp_name>/<model>-ckpt-<stem>-<method>-<steps>[-interval..-cfg..-shift..]
    tail = "-interval0.10-cfg4.00-shift0.30" if cfg_scale > 1 else ""
    assert out == os.path.join(cfg["train"]["output_dir"], cfg["train"]["exp_name"], "lightningdit-b-1-ckpt-ckpt-euler-2" + tail)
    files = sorted(os.listdir(out))
    assert files == [f"{i:06d}.png" for i in range(8)]                       # 2 batches of 4, world 1
    ims = [np.asarray(Image.open(os.path.join(out, f))) for f in files]
    assert all(im.shape == (64, 64, 3) and im.dtype == np.uint8 for im in ims)
    assert len({im.tobytes() for im in ims}) == 8 and all(im.std() > 0 for im in ims)      # eight different, non-constant images
    # a folder that already holds more than fid_num PNGs is left alone (inference.py:69-77)
    Image.fromarray(ims[0]).save(os.path.join(out, "extra.png"))
    stamp = {f: os.path.getmtime(os.path.join(out, f)) for f in os.listdir(out)}
    assert inf.do_sample(cfg, str(tmp_path / "ckpt.pt")) == out
    assert stamp == {f: os.path.getmtime(os.path.join(out, f)) for f in os.listdir(out)}
    # --demo (inference.py:54-57, 219-262): eight single-image calls with guidance on every step of the unshifted grid (the fixed ImageNet
    # classes when guided, class 0 eight times otherwise), one 2 x 4 sheet under ./demo_images, nothing returned
    monkeypatch.chdir(tmp_path)
    assert inf.do_sample(cfg, str(tmp_path / "ckpt.pt"), demo=True) is None
    sheet = np.asarray(Image.open(tmp_path / "demo_images" / f"{cfg['train']['exp_name']}_cfg{cfg_scale}_ckpt_demo_samples.png"))
    assert sheet.shape == (128, 256, 3)
    tiles = [sheet[i * 64:(i + 1) * 64, j * 64:(j + 1) * 64] for i in range(2) for j in range(4)]
    assert len({t_.tobytes() for t_ in tiles}) == 8 and all(t_.std() > 0 for t_ in tiles)
    # the command line of run_inference.sh: --config only, checkpoint from the YAML's ckpt_path (inference.py:324-327), PRECISION from the launcher's
    # environment (fp32 -> the exact-f32 kernels: a second folder of eight PNGs, close to but not bit-equal with the bf16 ones)
    cfg["ckpt_path"] = str(tmp_path / "0000007.pt")
    os.replace(tmp_path / "ckpt.pt", cfg["ckpt_path"])
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    monkeypatch.setenv("PRECISION", "fp32")
    out32 = inf.main(["--config", str(tmp_path / "cfg.yaml")])
    assert os.path.basename(out32) == "lightningdit-b-1-ckpt-0000007-euler-2" + tail and out32 != out
    ims32 = [np.asarray(Image.open(os.path.join(out32, f"{i:06d}.png"))).astype(np.int32) for i in range(8)]
    diff = np.mean([np.abs(a_ - b_.astype(np.int32)).mean() for a_, b_ in zip(ims32, ims)])
    assert diff < 8.0, diff                                      # same seeds, same noise: bf16 against f32 sampling, in uint8 steps
    monkeypatch.setenv("PRECISION", "fp8")
    with pytest.raises(SystemExit):
        inf.main(["--config", str(tmp_path / "cfg.yaml")])
    monkeypatch.delenv("PRECISION")
    # the diffusers AutoencoderKL branch of the reference (inference.py:137-167) is refused by name
    cfg["vae"]["model_name"] = "sdv3_f8d16"
    with pytest.raises(NotImplementedError, match="VMAE"):
        inf.do_sample(cfg, cfg["ckpt_path"], str(tmp_path / "other"))


def test_png_writer_surfaces_errors(tmp_path):
    from ldmae_amd.inference import PngWriter
    w = PngWriter()
    w.put([np.zeros((4, 4, 3), np.uint8)], [str(tmp_path / "nodir" / "x.png")])
    with pytest.raises(Exception):
        w.close()


def test_extract_features_writes_reference_shards(tmp_path):
    """SURVEY 8(f)3, the WRITER (reference extract_features.py:103-218): an image folder -> VMAE moments of every image and of its mirror
    image -> shards named / keyed / tagged as the reference writes them -> latents_stats.pt; read back through ImgLatentDataset."""
    from PIL import Image
    from safetensors import safe_open
    import ldmae_amd.extract_features as ef
    from ldmae_amd.datasets.img_latent_dataset import ImgLatentDataset
    from ldmae_amd.tokenizer import models_mae
    rs = np.random.RandomState(0)
    root = tmp_path / "1K_dataset"
    n_img = 0
    for c in ("n03", "n01", "n02"):
        (root / "train" / c).mkdir(parents=True)
        for i in range(4 if c != "n02" else 3):
            w, h = int(rs.randint(70, 140)), int(rs.randint(70, 140))
            Image.fromarray(rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).save(str(root / "train" / c / f"img{i}.png"))
            n_img += 1
    vae = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=True, no_cls=True, kl_loss_weight=True, smooth_output=True, img_size=64)
    torch.manual_seed(3)
    with torch.no_grad():
        for p in vae.parameters():
            if p.requires_grad and p.ndim > 1:
                p.copy_(torch.randn_like(p) * 0.05)
    torch.save({"model": vae.state_dict()}, tmp_path / "vmae.pth")
    cfg = dict(vae=dict(model_name="vmae_f8d16", weight_path=str(tmp_path / "vmae.pth")),
               data=dict(origin_path=str(root), name="imagenet", sample=True))
    import argparse
    args = argparse.Namespace(data_split="train", output_dir=None, image_size=64, batch_size=4, seed=42, num_workers=0, precision="fp32",
                              synthetic=0)
    out_dir, files = ef.main(args, cfg)
    assert out_dir == str(tmp_path / "vmae_feature_imagenet_train_64_sample")            # extract_features.py:49-52
    # 11 images, batch 4 -> 3 batches; a shard closes after 10000 // 4 batches, so everything lands in the remainder shard
    assert [os.path.basename(f) for f in files] == ["latents_rank00_shard000.safetensors"]
    with safe_open(files[0], framework="pt") as h:
        assert sorted(h.keys()) == ["labels", "latents", "latents_flip"]
        assert h.metadata() == {"total_size": str(n_img), "dtype": "torch.float32", "device": "cpu"}
        lat, flip, labels = h.get_tensor("latents"), h.get_tensor("latents_flip"), h.get_tensor("labels")
    assert lat.shape == (n_img, 32, 8, 8) and flip.shape == lat.shape and lat.dtype == torch.float32 and labels.dtype == torch.int64
    assert labels.tolist() == [0] * 4 + [1] * 3 + [2] * 4                                # classes sorted: n01, n02, n03
    # contents: the tokenizer's own _encode of the transformed images, in ImageFolder order; the flipped loader sees mirror images
    from ldmae_amd.datasets.image_folder import ImageFolder
    vae = vae.cuda().eval()
    ds = ImageFolder(str(root / "train"), transform=vae.img_transform(p_hflip=0.0, img_size=64))
    x = torch.stack([ds[i][0] for i in range(n_img)]).cuda()
    with torch.no_grad(), models_mae.reference_tf32():          # the writer encodes under the reference's allow_tf32 switch (extract_features.py:2-3)
        want, want_flip = vae._encode(x).float().cpu(), vae._encode(x.flip(-1)).float().cpu()
    assert torch.allclose(lat, want, atol=1e-5, rtol=1e-5) and torch.allclose(flip, want_flip, atol=1e-5, rtol=1e-5)
    assert not torch.allclose(lat, flip, atol=1e-3)
    # the reader + cached stats
    assert os.path.exists(os.path.join(out_dir, "latents_stats.pt"))
    rd = ImgLatentDataset(out_dir, latent_norm=True, latent_multiplier=1.0, sample=True)
    assert len(rd) == n_img
    f, y = rd[5]
    assert f.shape == (16, 8, 8) and int(y) == 1 and torch.isfinite(f).all()
    # shard-closing rule with a small shard size: 3 batches of 4 at shard_images = 8 -> shards of 8 and 3 images
    loaders = [torch.utils.data.DataLoader(ImageFolder(str(root / "train"), transform=vae.img_transform(p_hflip=p, img_size=64)), batch_size=4)
               for p in (0.0, 1.0)]
    files2 = ef.extract(vae, loaders, str(tmp_path / "two"), batch_size=4, sample=False, shard_images=8, log=lambda *_: None)
    assert [os.path.basename(f) for f in files2] == ["latents_rank00_shard000.safetensors", "latents_rank00_shard001.safetensors"]
    with safe_open(files2[0], framework="pt") as h0, safe_open(files2[1], framework="pt") as h1:
        a, b = h0.get_tensor("latents"), h1.get_tensor("latents")
        assert a.shape == (8, 16, 8, 8) and b.shape == (3, 16, 8, 8)                      # posterior mode: the mean half of the moments
        assert torch.allclose(torch.cat([a, b]), want[:, :16], atol=1e-5, rtol=1e-5)


def test_latent_prologue_matches_reference_dataset_golden(tmp_path):
    """SURVEY 8(f)3 'fused GPU prologue': ImgLatentDataset(raw=True) + LatentPrologue (one HIP kernel per batch) must give what the
    REFERENCE's ImgLatentDataset returned per item for the same shards, flip picks and noise (tests/golden/dataset.npz, generated by
    importing the reference class): posterior sample from the moments, (x - mean) / std, multiplier."""
    from safetensors.torch import save_file
    from ldmae_amd.datasets.img_latent_dataset import ImgLatentDataset, LatentPrologue
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
        ds = ImgLatentDataset(str(d), raw=True, **kw)                  # stats: same RNG consumption as the reference's constructor
        if kw["latent_norm"]:
            assert np.allclose(ds._latent_mean.numpy(), g[f"ds_{tag}_mean"], atol=1e-6) and np.allclose(ds._latent_std.numpy(), g[f"ds_{tag}_std"], atol=1e-6)
        stored, labels, noise = [], [], []
        for i in g[f"ds_{tag}_order"]:
            x, y = ds[int(i)]                                          # np.random.uniform: the flip pick
            stored.append(x); labels.append(y)
            if kw["sample"]:
                noise.append(torch.randn(1, x.shape[0] // 2, *x.shape[1:]))      # the draw DiagonalGaussianDistribution.sample makes per item
        pro = LatentPrologue(ds).cuda()
        out = pro(torch.stack(stored).cuda(), noise=torch.cat(noise).cuda() if noise else None)
        want = torch.from_numpy(g[f"ds_{tag}_feat"])
        assert out.shape == want.shape and out.dtype == torch.float32
        assert torch.allclose(out.cpu(), want, atol=2e-6, rtol=2e-6), (tag, float((out.cpu() - want).abs().max()))
        assert torch.equal(torch.stack(labels), torch.from_numpy(g[f"ds_{tag}_label"]))
    # device-generator noise: right statistics, reproducible under a seeded generator
    gen = torch.Generator(device="cuda").manual_seed(5)
    mom = torch.cat([torch.zeros(64, 4, 8, 8), torch.full((64, 4, 8, 8), 2 * float(np.log(3.0)))], dim=1).cuda()      # mean 0, std 3
    ds.sample, ds.latent_norm, ds.latent_multiplier = True, False, 1.0
    pro = LatentPrologue(ds).cuda()
    a = pro(mom, generator=gen)
    b = pro(mom, generator=torch.Generator(device="cuda").manual_seed(5))
    assert torch.equal(a, b) and abs(float(a.std()) - 3.0) < 0.1 and abs(float(a.mean())) < 0.1
