#!/usr/bin/env python3
"""Sampling driver -- counterpart of the reference's LDMAE/inference.py (`run_inference.sh`): EMA checkpoint -> shifted-grid
Euler ODE with classifier-free guidance (CFG on the first three channels, interval gate) -> latent de-normalisation -> VMAE
``decode_to_images`` -> PNGs.  Ranks are independent replicas (seed = global_seed * world + rank, inference.py:87)."""
import argparse
import math
import os
import sys

import torch
import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
for p in (_HERE, os.path.dirname(_HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

from ldmae_amd.models.lightningdit import _act_dtype        # noqa: E402
from ldmae_amd.tokenizer import models_mae                  # noqa: E402
from ldmae_amd.train_accum import build_model               # noqa: E402
from ldmae_amd.transport import Sampler, create_transport   # noqa: E402


def build_sampler(cfg):
    t, s = cfg['transport'], cfg['sample']
    tr = create_transport(t['path_type'], t['prediction'], t['loss_weight'], t['train_eps'], t['sample_eps'],
                          use_cosine_loss=t.get('use_cosine_loss', False), use_lognorm=t.get('use_lognorm', False))
    if s['mode'] != "ODE":
        raise NotImplementedError(f"Sampling mode {s['mode']} is not supported.")
    return Sampler(tr).sample_ode(sampling_method=s['sampling_method'], num_steps=s['num_sampling_steps'], atol=s['atol'], rtol=s['rtol'],
                                  reverse=s['reverse'], timestep_shift=s.get('timestep_shift', 0))


@torch.no_grad()
def sample_latents(model, sample_fn, n, cfg_scale, cfg_interval_start, device, num_classes=1000, generator=None, truncation=None,
                   labels=None, cfg_interval=True):
    """inference.py:264-292: z ~ N(0,I); CFG on a doubled batch with the null class; Euler; drop the null half.
    `truncation`: the resampling loop of :266-272 (|z| > bound redrawn, at most 100 rounds).  `labels` / `cfg_interval=False`: the demo
    branch (:219-232), fixed classes and guidance on every step."""
    latent = model.x_embedder.img_size[0]
    z = torch.randn(n, model.in_channels, latent, latent, device=device, generator=generator)
    if truncation is not None:
        for _ in range(100):
            bad = z.abs() > truncation
            if not bool(bad.any()):
                break
            z[bad] = torch.randn(int(bad.sum()), device=device, generator=generator)
    if labels is None:
        y = torch.randint(0, num_classes, (n,), device=device, generator=generator)
    else:
        y = torch.as_tensor(labels, device=device, dtype=torch.long)
    if cfg_scale > 1.0:
        z = torch.cat([z, z], 0)
        y = torch.cat([y, torch.full((n,), num_classes, device=device)], 0)
        def cfg_forward(x, t, y, cfg_scale, cfg_interval=None, cfg_interval_start=None):
            # Below the interval start forward_with_cfg applies NO guidance (lightningdit.py:436-439): the conditional half gets its own
            # output, and the unconditional half's output is never used for the samples that are kept -- every step rebuilds the doubled
            # batch from the first half of the state (:423-424) and only that half is returned (inference.py:288).  So those steps run the
            # conditional half alone (the model is bitwise independent of the batch: tests/test_gpu_dit.py); 27 % of the 250 steps of the
            # shipped configuration (interval start 0.10, timestep shift 0.3).  The gate reads t[0] on the host, as the reference's does.
            # Valid ONLY for a fixed-step, per-sample-independent integrator (the shipped Euler / Heun grid): the second half returned here is
            # the conditional output again, which an adaptive solver's error norm would see.  And only while the half batch takes the same
            # adaLN path as the doubled one (the batched bf16 adaLN GEMM needs a batch that is a multiple of 8; an n of 4, 12, 20 ... would put
            # the half on the per-block f32 path and the guided steps on the batched bf16 one: not bit-for-bit the doubled batch any more).
            same_path = _act_dtype(getattr(model, "precision", None)) != torch.bfloat16 or (len(x) // 2) % 8 == 0 or len(x) % 8 != 0
            if cfg_interval is True and cfg_interval_start and float(t[0]) < cfg_interval_start and same_path:
                half = len(x) // 2
                out = model.forward(x[:half], t[:half], y[:half])
                return torch.cat([out, out], dim=0)
            return model.forward_with_cfg(x, t, y, cfg_scale, cfg_interval, cfg_interval_start)
        out = sample_fn(z, cfg_forward, y=y, cfg_scale=cfg_scale, cfg_interval=cfg_interval, cfg_interval_start=cfg_interval_start)[-1]
        out, _ = out.chunk(2, dim=0)
    else:
        out = sample_fn(z, model.forward, y=y)[-1]
    return out, y[:n]


class PngWriter:
    """PNG encoding on a host thread: the reference writes each image synchronously inside the sampling loop (inference.py:293-297),
    so the GPU sits idle while 256 PNGs are compressed; here the loop hands the uint8 batch over and goes on enqueueing the next
    batch's Euler steps.  Bounded queue (2 batches) so host memory stays flat; `close()` drains it and re-raises a writer error."""

    def __init__(self, depth=2):
        import queue
        import threading
        self.q, self.err = queue.Queue(maxsize=depth), None
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _run(self):
        from PIL import Image
        while True:
            item = self.q.get()
            if item is None:
                return
            try:
                imgs, paths = item
                for im, path in zip(imgs, paths):
                    Image.fromarray(im).save(path)
            except Exception as e:          # surfaced by close()
                self.err = e

    def put(self, imgs, paths):
        if self.err is not None:
            raise self.err
        self.q.put((imgs, paths))

    def close(self):
        self.q.put(None)
        self.th.join()
        if self.err is not None:
            raise self.err


DEMO_LABELS = [975, 3, 207, 387, 388, 88, 979, 279]          # inference.py:223


def sample_folder_name(cfg, ckpt_path, cfg_scale=None):
    """The directory name rule of inference.py:45-52 (the FID tooling downstream finds the PNGs by it)."""
    s = cfg['sample']
    name = f"{cfg['model']['model_type'].replace('/', '-')}-ckpt-{ckpt_path.split('/')[-1].split('.')[0]}-{s['sampling_method']}-{s['num_sampling_steps']}".lower()
    cfg_scale = s['cfg_scale'] if cfg_scale is None else cfg_scale
    if cfg_scale > 1.0:
        name += f"-interval{s.get('cfg_interval_start', 0):.2f}" + f"-cfg{cfg_scale:.2f}" + f"-shift{s.get('timestep_shift', 0):.2f}"
    return name


def build_vae(cfg, device):
    """inference.py:129-136.  The reference's other branch (`ae/dae/vae/sdv3` -> a diffusers AutoencoderKL, :137-167) is a different tokenizer
    family, outside this package: refused by name."""
    kind = cfg['vae']['model_name'].split("_")[0]
    if kind != 'vmae':
        raise NotImplementedError(f"vae.model_name {cfg['vae']['model_name']!r}: only the VMAE tokenizer ('vmae_*') is built here; "
                                  "the reference's diffusers AutoencoderKL branch is out of scope")
    vae = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=True, no_cls=True, kl_loss_weight=True, smooth_output=True,
                                              img_size=cfg['data']['image_size'])
    vck = torch.load(cfg['vae']['weight_path'], map_location='cpu')
    vae.load_state_dict(vck['model'], strict=False)
    return vae.to(device).eval()


def latent_stats(cfg, device):
    """inference.py:203-217: the statistics come through the latent dataset -- the cached latents_stats.pt when it is there (downloaded with the
    checkpoint), otherwise computed from the shards of data_path and cached, exactly as the reference's ImgLatentDataset.get_latent_stats does."""
    data_dir = cfg['data']['data_path'] + ('_sample' if 'sample' in cfg['data'] else '')
    cache = os.path.join(data_dir, "latents_stats.pt")
    if os.path.exists(cache):
        stats = torch.load(cache)
        mean, std = stats['mean'], stats['std']
    else:
        from ldmae_amd.datasets.img_latent_dataset import ImgLatentDataset
        mean, std = ImgLatentDataset(data_dir, latent_norm=cfg['data'].get('latent_norm', False), latent_multiplier=cfg['data'].get('latent_multiplier', 0.18215),
                                     sample=cfg['data'].get('sample', False)).get_latent_stats()
    return mean.to(device), std.to(device)


def do_sample(cfg, ckpt_path, out_dir=None, num_samples=None, precision="bf16", cfg_scale=None, demo=False):
    """inference.py:40-300.  `out_dir=None` -> <train.output_dir>/<train.exp_name>/<sample_folder_name> as the reference; a folder that already
    holds more than `fid_num` PNGs is left alone (:69-77).  `demo=True`: the eight fixed classes, guidance on every step, unshifted grid, one
    2 x 4 sheet under ./demo_images (:54-57, :219-262) written by rank 0; returns None as the reference does."""
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    s = cfg['sample']
    cfg_scale = s['cfg_scale'] if cfg_scale is None else cfg_scale
    if out_dir is None:
        out_dir = os.path.join(cfg['train']['output_dir'], cfg['train']['exp_name'], sample_folder_name(cfg, ckpt_path, cfg_scale))
    want = num_samples or s['fid_num']
    if not demo and os.path.isdir(out_dir) and sum(f.endswith('.png') for f in os.listdir(out_dir)) > want:
        if rank == 0:
            print(f"Found more than {want} PNG files in {out_dir}, skip sampling.")
        return out_dir
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    torch.manual_seed(cfg['train']['global_seed'] * world + rank)
    model = build_model(cfg, learn_sigma=cfg['model'].get('learn_sigma', False))        # inference.py:336-348
    ck = torch.load(ckpt_path, map_location='cpu')
    model.load_state_dict(ck["ema"] if "ema" in ck else ck)
    model = model.to(device).eval()
    if demo:                                                  # :54-57 -- the demo sheet is drawn with guidance on every step of the unshifted grid
        cfg = dict(cfg, sample=dict(s, cfg_interval_start=0, timestep_shift=0))
        s = cfg['sample']
    sample_fn = build_sampler(cfg)
    vae = build_vae(cfg, device)
    mean, std = latent_stats(cfg, device)
    mult = cfg['data'].get('latent_multiplier', 0.18215)

    def decode(lat):
        # inference.py:79 of the reference sets allow_tf32: the f32 decode runs TF32-class (fp16 operands = TF32's mantissa, f32
        # accumulation: the LDMAE_F16 kernel family); LDMAE_TF32=0 keeps the exact-f32 kernels
        with models_mae.reference_tf32():
            return vae.decode_to_images(lat * std / mult + mean)                      # uint8 NHWC on the host (inference.py:290-292)

    if demo:
        if rank != 0:
            return None
        import numpy as np
        from PIL import Image
        sheet = []
        for label in (DEMO_LABELS if cfg_scale > 1.0 else [0] * 8):                    # one image per call, as :224-245 (the noise stream depends on it)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=precision == "bf16"):
                lat, _ = sample_latents(model, sample_fn, 1, cfg_scale, 0, device, cfg['data']['num_classes'], labels=[label], cfg_interval=False)
            sheet.append(decode(lat)[0])
        h, w = sheet[0].shape[:2]
        grid = np.zeros((2 * h, 4 * w, 3), np.uint8)
        for k, im in enumerate(sheet):
            i, j = divmod(k, 4)
            grid[i * h:(i + 1) * h, j * w:(j + 1) * w] = im
        os.makedirs('demo_images', exist_ok=True)
        ckpt_iter = ckpt_path.split("/")[-1][:-3]
        Image.fromarray(grid).save(f"demo_images/{cfg['train']['exp_name']}_cfg{cfg_scale}_{ckpt_iter}_demo_samples.png")
        return None

    n = s['per_proc_batch_size']
    total = int(math.ceil(want / (n * world)) * n * world)
    # :266-272 -- the reference looks the switch up under the key 'trunaction' (sic) and the bound under 'truncation'; a config written for it
    # behaves the same here
    trunc = s['truncation'] if 'trunaction' in s else None
    os.makedirs(out_dir, exist_ok=True)
    done = 0
    writer = PngWriter()
    try:
        for it in range(total // (n * world)):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=precision == "bf16"):
                lat, _ = sample_latents(model, sample_fn, n, cfg_scale, s.get('cfg_interval_start', 0), device, cfg['data']['num_classes'], truncation=trunc)
            imgs = decode(lat)
            writer.put(imgs, [f"{out_dir}/{i * world + rank + done:06d}.png" for i in range(len(imgs))])   # index rule: inference.py:294
            done += n * world
    finally:
        writer.close()
    return out_dir


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=str, default='configs/lightningdit_b_ldmvae_f16d16.yaml')           # inference.py:318
    ap.add_argument('--demo', action='store_true', default=False)                                          # inference.py:319
    ap.add_argument('--ckpt', type=str, default=None, help="default: the config's ckpt_path (inference.py:324-327)")
    ap.add_argument('--out', type=str, default=None, help="default: <output_dir>/<exp_name>/<the reference's folder name>")
    a = ap.parse_args(argv)
    c = yaml.safe_load(open(a.config))
    if a.ckpt is None:
        assert 'ckpt_path' in c, "ckpt_path must be specified in config"
    # run_inference.sh passes --mixed_precision $PRECISION (default bf16) to the launcher, which exports ACCELERATE_MIXED_PRECISION; the
    # reference's sampler never prepares the model, so there the flag is inert and it samples in f32 with TF32 matmuls -- here it selects the
    # activation type (bf16 = BASELINE config 5; PRECISION=fp32 / no -> the exact-f32 kernels)
    precision = os.environ.get("PRECISION") or os.environ.get("ACCELERATE_MIXED_PRECISION") or "bf16"
    precision = {"no": "fp32"}.get(precision, precision)
    if precision not in ("bf16", "fp32"):
        raise SystemExit(f"PRECISION={precision!r}: the sampler runs in bf16 or fp32")
    folder = do_sample(c, a.ckpt or c['ckpt_path'], a.out, demo=a.demo, precision=precision)
    if not a.demo and int(os.environ.get("RANK", 0)) == 0:
        # inference.py:352-367 goes on to an Inception FID against data.fid_reference_file (tools/calculate_fid.py); that needs the Inception
        # weights, which this package does not carry (evaluation tools are out of scope)
        print(f"samples written to {folder}; FID (tools/calculate_fid.py in the reference) is not part of this package")
    return folder


if __name__ == "__main__":
    main()
