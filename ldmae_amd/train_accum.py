#!/usr/bin/env python3
"""Train driver for LightningDiT on MI355X -- counterpart of the reference's LDMAE/train_accum.py with the same YAML, the same
``--config`` flag, the same log / checkpoint layout, launched by the same ``run_train.sh`` (``accelerate launch`` or torchrun: both
export RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).

Differences, all on purpose: the gradient exchange is owned here (``GradBucketReducer``: RCCL all-reduce of contiguous slab
slices on a side stream, overlapped with backward) instead of ``DDP(DDP(model))``; AdamW + EMA are one fused kernel pass
(``AdamWEMA``); the loss is logged from a device accumulator, so there is no ``loss.item()`` sync per micro-step.
"""
import argparse
import json
import logging
import os
import sys
from glob import glob
from time import time

import numpy as np
import torch
import torch.distributed as dist
import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)          # so `models`, `transport`, `tokenizer`, `datasets` resolve to this tree
if os.path.dirname(_HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(_HERE))

from ldmae_amd import ops                                 # noqa: E402
from ldmae_amd.distributed import GradBucketReducer, batched_adaln_pays      # noqa: E402
from ldmae_amd.models.lightningdit import LightningDiT_models  # noqa: E402
from ldmae_amd.optim import AdamWEMA, adaln_first                      # noqa: E402
from ldmae_amd.transport import create_transport         # noqa: E402


def load_config(path):
    with open(path) as f:
        return yaml.safe_load(f)


def create_logger(logging_dir, rank):
    logger = logging.getLogger("ldmae_amd.train")
    if rank == 0:
        logging.basicConfig(level=logging.INFO, format='[\033[34m%(asctime)s\033[0m] %(message)s', datefmt='%Y-%m-%d %H:%M:%S',
                            handlers=[logging.StreamHandler(), logging.FileHandler(f"{logging_dir}/log.txt")])
    else:
        logger.addHandler(logging.NullHandler())
    return logger


def build_model(cfg, **extra):
    """train_accum.py:73-90.  `extra`: keywords only the sampler's call passes (inference.py:346 `learn_sigma`)."""
    ds = cfg['vae'].get('downsample_ratio', 16)
    assert cfg['data']['image_size'] % ds == 0, "Image size must be divisible by the VAE downsample ratio."
    m = cfg['model']
    return LightningDiT_models[m['model_type']](
        input_size=cfg['data']['image_size'] // ds, num_classes=cfg['data']['num_classes'], use_qknorm=m['use_qknorm'],
        use_swiglu=m.get('use_swiglu', False), use_rope=m.get('use_rope', False), use_rmsnorm=m.get('use_rmsnorm', False),
        wo_shift=m.get('wo_shift', False), in_channels=m.get('in_chans', 4), use_checkpoint=m.get('use_checkpoint', False),
        class_dropout_prob=0 if cfg['data']['num_classes'] == 1 else 0.1, **extra)


def load_weights_with_shape_check(model, checkpoint, rank=0):
    """train_accum.py:308-334 (incl. the first-16-input-channels special case for x_embedder.proj.weight)."""
    sd = model.state_dict()
    for name, p in checkpoint['model'].items():
        name = name.replace('module.', '')
        if name not in sd:
            if rank == 0:
                print(f"Parameter '{name}' not found in model, skipping.")
        elif p.shape == sd[name].shape:
            sd[name].copy_(p)
        elif name == 'x_embedder.proj.weight':
            w = torch.zeros_like(sd[name])
            w[:, :16] = p[:, :16]
            sd[name] = w
        elif rank == 0:
            print(f"Skipping loading parameter '{name}' due to shape mismatch: checkpoint {p.shape}, model {sd[name].shape}")
    model.load_state_dict(sd, strict=False)
    return model


def make_loader(cfg, per_gpu, rank, world, synthetic):
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    d = cfg['data']
    if synthetic:
        from ldmae_amd.datasets.img_latent_dataset import SyntheticLatentDataset
        ds_ = SyntheticLatentDataset(channels=cfg['model'].get('in_chans', 4), size=d['image_size'] // cfg['vae'].get('downsample_ratio', 16),
                                     num_classes=d['num_classes'], seed=cfg['train'].get('global_seed', 0))
    else:
        from ldmae_amd.datasets.img_latent_dataset import ImgLatentDataset
        path = d['data_path'] + ('_sample' if 'sample' in d else '')                 # key presence, train_accum.py:124-125
        # data.gpu_prologue (not a reference key): posterior sampling + normalisation per batch on the GPU instead of per item in the workers
        ds_ = ImgLatentDataset(data_dir=path, latent_norm=d.get('latent_norm', False), latent_multiplier=d.get('latent_multiplier', 0.18215),
                               sample=d.get('sample', False), raw=bool(d.get('gpu_prologue', False)))
    sampler = DistributedSampler(ds_, num_replicas=world, rank=rank, shuffle=True, seed=cfg['train'].get('global_seed', 0)) if world > 1 else None
    # workers come from a fork server: forking THIS process (GPU runtime threads, pin-memory thread, RCCL threads) can hand a child a lock
    # that is held at the moment of the fork and never released in the child (extract_features.py saw such workers hang)
    nw = d.get('num_workers', 0)
    return ds_, DataLoader(ds_, batch_size=per_gpu, shuffle=sampler is None, sampler=sampler, num_workers=nw,
                           multiprocessing_context="forkserver" if nw > 0 else None,
                           pin_memory=True, drop_last=True)


def do_train(cfg, synthetic=False, max_steps=None, precision=None):
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    # LDMAE_DIST_BACKEND=gloo + LDMAE_DEVICE=0: several ranks share ONE GPU (rehearsal of the multi-rank launch on a 1-GPU box)
    backend = os.environ.get("LDMAE_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LDMAE_DEVICE", local))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the only form this driver supports (RCCL needs it)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    tr_cfg = cfg['train']
    exp_dir = f"{tr_cfg['output_dir']}/{tr_cfg['exp_name']}"
    ckpt_dir = f"{exp_dir}/checkpoints"
    if rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True)
    if world > 1:
        dist.barrier()
    logger = create_logger(exp_dir, rank)
    precision = precision or os.environ.get("PRECISION") or os.environ.get("ACCELERATE_MIXED_PRECISION") or "bf16"
    precision = {"no": "fp32"}.get(precision, precision)

    model = build_model(cfg)
    if 'weight_init' in tr_cfg:
        ck = torch.load(tr_cfg['weight_init'], map_location='cpu')
        load_weights_with_shape_check(model, ck, rank)
    model = model.to(device).train()
    o = cfg['optimizer']
    # Batched adaLN (models.lightningdit._AdaLNAllFn) finishes the adaLN weight gradients of EVERY block at the very end of backward, so those
    # weights sit at the FRONT of the gradient slab (optim.adaln_first): the reducer cuts its buckets from the end backwards, the front ones
    # are the last to be all-reduced anyway, and every other bucket still starts under backward.  The data-parallel step is then the same
    # program as the single-GPU one (round 3 switched the batched form off for world > 1).
    opt = AdamWEMA(model, lr=o['lr'], betas=(0.9, o['beta2']), weight_decay=0.0, ema_decay=0.9999, front_fn=adaln_first)
    reducer = GradBucketReducer(opt.flat)
    # world > 1: those front buckets (170 MB for B/1, 890 MB for XL/1) are all-reduced fully exposed, which can cost more than the ~0.125 ms per
    # block that batching saves -- chosen by size until a multi-rank run has measured it (distributed.batched_adaln_pays; the exposed time is
    # logged below); LDMAE_BATCHED_ADALN=0|1 overrides
    adaln_bytes = sum(p.numel() * 4 for n, p in opt.flat.trainable if adaln_first(n))
    env = os.environ.get("LDMAE_BATCHED_ADALN")
    model.batched_adaln = (env != "0") if env is not None else batched_adaln_pays(adaln_bytes, len(model.blocks), reducer.world)
    reducer.measure_exposed = reducer.world > 1          # two events per step: the first multi-GPU run shows which choice wins
    # world > 1: LDMAE_DP_GEMM_LAUNCH=tile|persistent (default: what bench.py --dp-config measured faster with the reducer's hooks live)
    ops.set_gemm_launch_mode(os.environ.get("LDMAE_DP_GEMM_LAUNCH", reducer.recommended_gemm_launch_mode()) if reducer.world > 1 else "persistent")
    model.direct_param_grads = True       # every .grad is a slab view and backward is a plain loss.backward(): dW goes straight into the slab
    reducer.broadcast_params(0)
    opt.ema.copy_(opt.flat.params)                               # update_ema(ema, model, decay=0), train_accum.py:166
    t = cfg['transport']
    transport = create_transport(t['path_type'], t['prediction'], t['loss_weight'], t['train_eps'], t['sample_eps'],
                                 use_cosine_loss=t.get('use_cosine_loss', False), use_lognorm=t.get('use_lognorm', False))
    train_steps = 0
    if tr_cfg.get('resume', False):
        files = sorted(glob(f"{ckpt_dir}/*.pt"))                 # by step (the reference sorts by file size, :176)
        if files:
            ck = torch.load(files[-1], map_location='cpu')
            model.load_state_dict(ck['model'])
            opt.ema.copy_(opt.flat.params)
            for k, v in ck['ema'].items():
                if k in opt.flat.offsets:
                    opt.flat.view(opt.ema, k, v.shape).copy_(v)
            train_steps = int(os.path.basename(files[-1]).split('.')[0])
            logger.info(f"Resuming training from checkpoint: {files[-1]}")
    per_gpu = int(np.round(tr_cfg['global_batch_size'] / world))
    accum = int(tr_cfg['gradient_accumulation_steps'])
    dataset, loader = make_loader(cfg, per_gpu, rank, world, synthetic)
    prologue = None
    if getattr(dataset, "raw", False):
        from ldmae_amd.datasets.img_latent_dataset import LatentPrologue
        prologue = LatentPrologue(dataset).to(device)
    if 'valid_path' in cfg['data']:
        # train_accum.py:148-163,288-297 builds a validation loader and calls an `evaluate` that the reference never defines (NameError at the
        # first checkpoint step); no shipped config sets the key.  Said once instead of dying after ckpt_every steps.
        logger.info(f"data.valid_path={cfg['data']['valid_path']!r} is ignored: the reference's validation pass calls an undefined evaluate()")
    logger.info(f"LightningDiT Parameters: {sum(p.numel() for p in model.parameters()) / 1e6:.2f}M; {len(dataset):,} samples; "
                f"batch {per_gpu}/gpu x {world} gpus x {accum} accumulation; precision {precision}")
    max_steps = max_steps or tr_cfg['max_steps']
    running = torch.zeros((), device=device)
    log_steps, micro, start = 0, 0, time()
    opt.zero_grad()
    epoch = 0
    while train_steps < max_steps:
        if getattr(loader, "sampler", None) is not None and hasattr(loader.sampler, "set_epoch"):
            loader.sampler.set_epoch(epoch)              # reshuffle per epoch, as the reference's DataLoader(shuffle=True) does
        epoch += 1
        for x, y in loader:
            x, y = x.to(device, non_blocking=True), y.to(device, non_blocking=True)
            if prologue is not None:
                x = prologue(x)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=precision == "bf16"):
                terms = transport.training_losses(model, x, dict(y=y))
            loss = terms["loss"].mean()
            if 'cos_loss' in terms:
                loss = loss + terms["cos_loss"].mean()
            running += terms["loss"].mean().detach()
            # the slab accumulates the local micro-step gradients; ONE all-reduce on the last micro-step gives sum over ranks of
            # the local sums = what the reference's per-micro-step DDP mean accumulates to (train_accum.py:223-244), times world
            reducer.sync = micro + 1 == accum
            (loss / accum).backward()
            micro += 1
            if micro < accum:
                continue
            clip = o.get('max_grad_norm', None)
            scale = reducer.finish()
            if clip is not None:
                gn = opt.flat.grads.norm() * scale
                scale = scale * float(min(1.0, clip / (float(gn) + 1e-6)))
            opt.step(grad_scale=scale)
            opt.zero_grad()
            micro = 0
            log_steps += 1
            train_steps += 1
            if train_steps % tr_cfg['log_every'] == 0:
                torch.cuda.synchronize()
                avg = running / (log_steps * accum)
                if world > 1:
                    dist.all_reduce(avg, op=dist.ReduceOp.SUM)
                comm = f", exposed all-reduce {reducer.exposed_comm_ms() / max(log_steps, 1):.2f} ms/step (batched adaLN {'on' if model.batched_adaln else 'off'})" if world > 1 else ""
                logger.info(f"(step={train_steps:07d}) Train Loss: {avg.item() / world:.4f}, Train Steps/Sec: {log_steps / (time() - start):.2f}{comm}")
                running.zero_()
                log_steps, start = 0, time()
            if train_steps % tr_cfg['ckpt_every'] == 0 and train_steps > 0:
                if rank == 0:
                    path = f"{ckpt_dir}/{train_steps:07d}.pt"
                    torch.save({"model": model.state_dict(), "ema": opt.ema_state_dict(), "opt": opt.torch_adamw_state_dict(), "config": cfg}, path)   # train_accum.py:275-280: "opt" = AdamW.state_dict()
                    logger.info(f"Saved checkpoint to {path}")
                if world > 1:
                    dist.barrier()
            if train_steps >= max_steps:
                break
    logger.info("Done!")
    return model, opt


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=str, default='configs/debug.yaml')
    ap.add_argument('--synthetic', action='store_true', help='N(0,1) latents instead of the safetensors shards')
    ap.add_argument('--max-steps', type=int, default=None)
    a = ap.parse_args()
    do_train(load_config(a.config), synthetic=a.synthetic, max_steps=a.max_steps)
