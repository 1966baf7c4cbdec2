#!/usr/bin/env python3
"""VMAE pre-training step driver on the HIP kernels -- counterpart of the reference's VMAE/engine_pretrain.py:21-110 and the optimizer /
schedule set-up of VMAE/main_pretrain.py:236-300 (minimal slice: no ImageFolder I/O, no LPIPS, no TensorBoard).

    python ldmae_amd/vmae_pretrain.py --synthetic --epochs 1 --steps-per-epoch 10 --batch_size 64

What is kept: forward_vanilla loss (masked / visible MSE + KL), per-iteration half-cycle cosine LR with linear warm-up
(util/lr_sched.py:9-25), AdamW(betas 0.9 / 0.95) with timm's ``param_groups_weight_decay`` split (no decay on biases and other
1-D parameters, main_pretrain.py:258-259), gradient accumulation, and the GradScaler PROTOCOL of the reference's
``NativeScalerWithGradNormCount`` (VMAE/util/misc.py:406-435: scale the loss, unscale, SKIP the optimizer step when a gradient is
inf / nan and halve the scale, double it after 2000 clean steps) -- ``LossScaler`` below, with 1/scale folded into the fused AdamW
kernel's grad_scale.  What differs on purpose: the activation type is bf16 where the reference's ``torch.amp.autocast('cuda')`` gives
fp16 (engine_pretrain.py:51-57).  bf16 keeps f32's exponent range, so the scaler never has an overflow to back off from in practice
(it is kept for the skip-on-non-finite semantics and so that checkpoints carry the same ``amp_scaler`` state); it has 8 significant
bits against fp16's 11, which is what the bf16-vs-f32 tolerance in tests/test_gpu_mae.py (loss within 2e-2 relative) prices.  The
kernels have no fp16 path.  Also: the fused AdamW kernel on a grouped contiguous slab instead of torch.optim.AdamW.
"""
import argparse
import math
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
for p in (_HERE, os.path.dirname(_HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

from ldmae_amd.optim import AdamWEMA, FlatParams           # noqa: E402
from ldmae_amd.tokenizer import models_mae                # noqa: E402


def cosine_lr(epoch, lr, min_lr, warmup_epochs, epochs, fixed_lr=False):
    """util/lr_sched.py:9-18 (`epoch` is fractional: data_iter_step / len(loader) + epoch, engine_pretrain.py:46)."""
    if fixed_lr:
        return lr
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1. + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


def no_decay(name, param):
    """timm optim_factory.param_groups_weight_decay: 1-D parameters and biases are not decayed.  -> group id (0 decay, 1 no decay)."""
    return 1 if param.ndim <= 1 or name.endswith(".bias") else 0


def build_optimizer(model, lr, weight_decay):
    flat = FlatParams(model, group_fn=no_decay)
    if hasattr(model, "set_direct_param_grads") and os.environ.get("LDMAE_DIRECT_GRADS", "1") != "0":
        model.set_direct_param_grads(True)        # every .grad is a slab view from here on: the blocks add their gradients into it themselves
    return AdamWEMA(model, lr=lr, betas=(0.9, 0.95), weight_decay=weight_decay, ema_decay=0.0, flat=flat,
                    group_weight_decay={0: weight_decay, 1: 0.0})


class LossScaler:
    """torch.amp.GradScaler's protocol (defaults init_scale 65536, growth 2, backoff 0.5, growth_interval 2000) as the reference drives it
    through NativeScalerWithGradNormCount.__call__ (VMAE/util/misc.py:413-430), on the flat gradient slab: ``scale`` multiplies the loss
    before backward; ``step`` looks for a non-finite gradient (one reduction over the slab), SKIPS the optimizer step if there is one,
    otherwise steps with 1/scale folded into the fused kernel; then updates the scale.  Returns the unscaled gradient norm, like the
    reference's get_grad_norm_ (None when the step was skipped)."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval, self.enabled = growth_factor, backoff_factor, growth_interval, enabled
        self._good, self.skipped = 0, 0

    def step(self, opt):
        g = opt.flat.grads
        norm = torch.linalg.vector_norm(g)                       # inf / nan anywhere in the slab makes the norm non-finite
        if self.enabled and not bool(torch.isfinite(norm)):
            self.skipped += 1
            self._good = 0
            self.scale *= self.backoff_factor
            return None
        opt.step(grad_scale=1.0 / self.scale)
        if self.enabled:
            self._good += 1
            if self._good >= self.growth_interval:
                self.scale *= self.growth_factor
                self._good = 0
        return float(norm) / self.scale

    def state_dict(self):
        return {"scale": self.scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self._good}

    def load_state_dict(self, sd):
        self.scale, self._good = float(sd["scale"]), int(sd.get("_growth_tracker", 0))


def train_one_epoch(model, loader, opt, epoch, args, log=print, scaler=None):
    """engine_pretrain.py:21-110 without the metric logger.  `scaler`: a LossScaler (the reference's loss_scaler argument); None = plain
    steps."""
    model.train(True)
    opt.zero_grad()
    n = len(loader)
    stats = None
    for it, (samples, _) in enumerate(loader):
        if it % args.accum_iter == 0:
            opt.lr = cosine_lr(it / n + epoch, args.lr, args.min_lr, args.warmup_epochs, args.epochs, args.fixed_lr)
        samples = samples.cuda(non_blocking=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.precision == "bf16"):
            loss, _, _, vis_loss, mask_loss, kl_loss = model(samples, mask_ratio=args.mask_ratio, visible_loss_ratio=args.visible_loss_ratio)
        if not math.isfinite(float(loss)):
            raise RuntimeError(f"Loss is {float(loss)}, stopping training")
        (loss * (scaler.scale if scaler is not None else 1.0) / args.accum_iter).backward()
        if (it + 1) % args.accum_iter == 0:
            if scaler is not None:
                scaler.step(opt)
            else:
                opt.step()
            opt.zero_grad()
        stats = dict(loss=float(loss), vis_loss=float(vis_loss), mask_loss=float(mask_loss), kl_loss=float(kl_loss) if kl_loss is not None else 0.0,
                     lr=opt.lr)
        if it % args.print_freq == 0:
            log(f"Epoch: [{epoch}] [{it}/{n}] " + "  ".join(f"{k}: {v:.6f}" for k, v in stats.items()))
    return stats


class _SyntheticImages(torch.utils.data.Dataset):
    def __init__(self, n, size):
        self.n, self.size = n, size

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(i)
        return torch.rand(3, self.size, self.size, generator=g) * 2 - 1, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mae_for_ldmae_f8d16_prev")
    ap.add_argument("--input_size", type=int, default=256)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=400)
    ap.add_argument("--accum_iter", type=int, default=1)
    ap.add_argument("--mask_ratio", type=float, default=0.75)
    ap.add_argument("--visible_loss_ratio", type=float, default=0.5)
    ap.add_argument("--kl_loss_weight", type=float, default=1e-6)
    ap.add_argument("--weight_decay", type=float, default=0.05)
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--blr", type=float, default=1e-3)
    ap.add_argument("--min_lr", type=float, default=0.)
    ap.add_argument("--warmup_epochs", type=int, default=40)
    ap.add_argument("--fixed_lr", action="store_true")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--print_freq", type=int, default=20)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--steps-per-epoch", type=int, default=100)
    args = ap.parse_args()
    if not args.synthetic:
        raise NotImplementedError("image-folder input is host I/O outside the kernel path: use --synthetic (SURVEY 8f rank 4)")
    if args.lr is None:
        args.lr = args.blr * args.batch_size * args.accum_iter / 256          # main_pretrain.py:249-252
    model = getattr(models_mae, args.model)(ldmae_mode=False, no_cls=True, kl_loss_weight=args.kl_loss_weight, smooth_output=True,
                                            img_size=args.input_size).cuda()
    opt = build_optimizer(model, args.lr, args.weight_decay)
    loader = torch.utils.data.DataLoader(_SyntheticImages(args.batch_size * args.steps_per_epoch, args.input_size), batch_size=args.batch_size,
                                         drop_last=True)
    scaler = LossScaler(enabled=args.precision == "bf16")          # main_pretrain.py:268: loss_scaler = NativeScaler()
    for epoch in range(args.epochs):
        print("Averaged stats:", train_one_epoch(model, loader, opt, epoch, args, scaler=scaler), "skipped steps:", scaler.skipped)


if __name__ == "__main__":
    main()
