#!/usr/bin/env python3
"""VMAE pre-training driver on the HIP kernels -- counterpart of the reference's VMAE/main_pretrain.py:74-300 + VMAE/engine_pretrain.py:21-110
(SURVEY 8f rank 4): image-folder input (main_pretrain.py:111-192), DistributedSampler + data-parallel gradient exchange (:204-215, 254-256),
checkpoint save / resume with the position-embedding resize (VMAE/util/misc.py:468-531).  No LPIPS (`ldmae_mode`), no TensorBoard.

    python ldmae_amd/vmae_pretrain.py --data_path /data/imagenet --output_dir out --batch_size 64          # <data_path>/train/<class>/*.JPEG
    python ldmae_amd/vmae_pretrain.py --data_path /data/my_pngs --output_dir out                          # any tree of images
    python -m torch.distributed.run --nproc-per-node 8 ldmae_amd/vmae_pretrain.py --data_path ...          # one rank per GPU over RCCL
    python ldmae_amd/vmae_pretrain.py --synthetic --epochs 1 --steps-per-epoch 10 --batch_size 64

The command line takes main_pretrain.py:37-91's flag set (VMAE/train_ae.sh:26-46 parses; `--perceptual_loss_ratio`, `--tune_decoder`, `--pred_with_conv`,
`--gradual_resol` are refused by name).  The posterior KL is the PRE-TRAINING tree's (round 6): VMAE/util/misc.py:103-125 differs from the tokenizer copy the
LDMAE drivers import -- no mean^2 term, and `--fixed_std s` (train_ae.sh:33 passes 1e-3) = KL against N(mean, s^2); `--kl_form tokenizer` restores the other.
One deliberate difference: the reference parses `--visible_loss_ratio` and never passes it on (engine_pretrain.py:57 calls model(samples, mask_ratio=...): its
runs train with the model's default 0.5); here the flag takes effect -- pass 0.5 (the default) to reproduce the reference's runs.

What is kept: forward_vanilla loss (masked / visible MSE + KL), per-iteration half-cycle cosine LR with linear warm-up
(util/lr_sched.py:9-25), AdamW(betas 0.9 / 0.95) with timm's ``param_groups_weight_decay`` split (no decay on biases and other
1-D parameters, main_pretrain.py:258-259), gradient accumulation, and the GradScaler PROTOCOL of the reference's
``NativeScalerWithGradNormCount`` (VMAE/util/misc.py:406-435: scale the loss, unscale, SKIP the optimizer step when a gradient is
inf / nan and halve the scale, double it after 2000 clean steps) -- ``LossScaler`` below, with 1/scale folded into the fused AdamW
kernel's grad_scale.  Activation type: ``--precision fp16`` is the reference's ``torch.amp.autocast('cuda')`` (engine_pretrain.py:51-57) on the
fp16 kernel family (LDMAE_F16, round 5: forward and backward of the blocks; gradient GEMM outputs overflow to infinity, which the scaler
answers by skipping the step and halving the scale); the default stays bf16 (f32's exponent range: the scaler never has an overflow to back
off from; 8 significant bits against fp16's 11 -- tests/test_gpu_mae.py prices both against f32 over 50 steps).  Also: the fused AdamW kernel on a
grouped contiguous slab instead of torch.optim.AdamW; the host reads the loss where the reference prints it, not on every micro-step.
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

from ldmae_amd.distributed import GradBucketReducer        # noqa: E402
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

    def step(self, opt, grad_scale=1.0):
        """grad_scale: what the gradient slab still has to be multiplied by besides 1 / scale (1 / world after a summing all-reduce)."""
        g = opt.flat.grads
        norm = torch.linalg.vector_norm(g)                       # inf / nan anywhere in the slab makes the norm non-finite
        if not bool(torch.isfinite(norm)):                      # (the one host read of the step; after a data-parallel all-reduce every rank sees the same slab)
            if not self.enabled:
                # no loss scaling (fp32): a non-finite gradient is a non-finite loss (engine_pretrain.py:68-70 stops there) -- never step on it
                raise RuntimeError("Loss is not finite, stopping training (non-finite gradient slab)")
            self.skipped += 1
            self._good = 0
            self.scale *= self.backoff_factor
            return None
        opt.step(grad_scale=grad_scale / self.scale)
        if self.enabled:
            self._good += 1
            if self._good >= self.growth_interval:
                self.scale *= self.growth_factor
                self._good = 0
        return float(norm) * grad_scale / self.scale

    def state_dict(self):
        return {"scale": self.scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self._good}

    def load_state_dict(self, sd):
        self.scale, self._good = float(sd["scale"]), int(sd.get("_growth_tracker", 0))


def train_one_epoch(model, loader, opt, epoch, args, log=print, scaler=None, reducer=None):
    """engine_pretrain.py:21-110 without the metric logger.  `scaler`: a LossScaler (the reference's loss_scaler argument); None = plain
    steps.  `reducer`: a GradBucketReducer over opt.flat (data parallel: the slab accumulates the local micro-steps and is all-reduced once,
    on the last one; parameters that got no gradient -- DDP(find_unused_parameters=True), main_pretrain.py:255 -- keep their zeros and their
    buckets are launched by finish()).
    The host reads the loss only where the reference PRINTS it (every print_freq iterations) and at the end of the epoch -- the reference's
    per-iteration `loss.item()` (engine_pretrain.py:59-63) stalls the launch queue on every micro-step; a non-finite loss still stops the
    run, at the next check."""
    model.train(True)
    opt.zero_grad()
    n = len(loader)
    stats, bad = None, torch.zeros((), dtype=torch.float32, device="cuda")
    world = reducer.world if reducer is not None else 1
    if scaler is None:
        scaler = LossScaler(enabled=False)        # plain steps, but never on a non-finite gradient slab (LossScaler.step raises: all ranks, same step)

    def read(st):
        # a non-finite LOSS stops the run where the reference prints it; the ranks agree first (MAX), so nobody is left waiting in a collective.
        # Steps taken since it appeared were skipped (scaler on) or refused (scaler off) by LossScaler.step: no update was made from it.
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if float(bad) > 0:
            raise RuntimeError("Loss is not finite, stopping training")
        return {k: (float(v) if torch.is_tensor(v) else v) for k, v in st.items()}
    for it, (samples, _) in enumerate(loader):
        if it % args.accum_iter == 0:
            opt.lr = cosine_lr(it / n + epoch, args.lr, args.min_lr, args.warmup_epochs, args.epochs, args.fixed_lr)
        samples = samples.cuda(non_blocking=True)
        with torch.autocast("cuda", dtype=torch.float16 if args.precision == "fp16" else torch.bfloat16, enabled=args.precision in ("bf16", "fp16")):
            loss, _, _, vis_loss, mask_loss, kl_loss = model(samples, mask_ratio=args.mask_ratio, visible_loss_ratio=args.visible_loss_ratio)
        bad.clamp_(min=(~torch.isfinite(loss.detach())).float())
        last = (it + 1) % args.accum_iter == 0
        if reducer is not None:
            reducer.sync = last
        (loss * scaler.scale / args.accum_iter).backward()
        if last:
            gscale = reducer.finish() if reducer is not None else 1.0          # 1 / world: the mean over the ranks, as DDP
            scaler.step(opt, gscale)
            opt.zero_grad()
        stats = dict(loss=loss.detach(), vis_loss=vis_loss.detach(), mask_loss=mask_loss.detach(),
                     kl_loss=kl_loss.detach() if kl_loss is not None else 0.0, lr=opt.lr)
        if it % args.print_freq == 0:
            log(f"Epoch: [{epoch}] [{it}/{n}] " + "  ".join(f"{k}: {v:.6f}" for k, v in read(stats).items()))
    out = read(stats) if stats is not None else None
    if out is not None and world > 1:                          # the reference logs the mean over the ranks (misc.all_reduce_mean)
        import torch.distributed as dist
        t = torch.tensor([out["loss"]], device="cuda")
        dist.all_reduce(t)
        out["loss"] = float(t) / world
    return out


class _SyntheticImages(torch.utils.data.Dataset):
    def __init__(self, n, size):
        self.n, self.size = n, size

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        # a pool of 64 images drawn once (a fresh 256 x 256 draw per sample made the --synthetic run host-bound: ~1 ms per image)
        if not hasattr(self, "pool"):
            g = torch.Generator().manual_seed(0)
            self.pool = torch.rand(64, 3, self.size, self.size, generator=g) * 2 - 1
        return self.pool[i % 64], 0


class _SyntheticLoader:
    """--synthetic: `steps` batches per epoch out of four random batches kept on the device (uniform in [-1, 1], like _SyntheticImages)."""

    def __init__(self, steps, batch, size, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        self.batches = [torch.rand(batch, 3, size, size, device="cuda", generator=g) * 2 - 1 for _ in range(4)]
        self.steps = steps

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.batches[i % 4], 0


# ----------------------------------------------------------------------------- image input (main_pretrain.py:111-192), PIL only
def _to_normalised_tensor(img):
    """ToTensor() + Normalize(mean 0.5, std 0.5) (main_pretrain.py:155)."""
    import numpy as np
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32).div_(255.0)
    return x.sub_(0.5).div_(0.5)


class RandomResizedCropFlip:
    """RandomResizedCrop(size, scale=(0.75, 1.0), ratio=(3/4, 4/3), bicubic) + RandomHorizontalFlip + ToTensor + Normalize: the 'imagenet'
    / 'laion' transform of main_pretrain.py:161-175, by torchvision's published algorithm (ten draws of (area fraction, log-uniform aspect
    ratio), the first crop that fits; otherwise the central crop clamped to the ratio range) on torch's RNG."""

    def __init__(self, size, scale=(0.75, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
        self.size, self.scale, self.ratio = int(size), scale, ratio

    def _box(self, w, h):
        area = w * h
        lr = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        for _ in range(10):
            ta = area * float(torch.empty(1).uniform_(self.scale[0], self.scale[1]))
            ar = math.exp(float(torch.empty(1).uniform_(lr[0], lr[1])))
            cw, ch = int(round(math.sqrt(ta * ar))), int(round(math.sqrt(ta / ar)))
            if 0 < cw <= w and 0 < ch <= h:
                return int(torch.randint(0, h - ch + 1, (1,))), int(torch.randint(0, w - cw + 1, (1,))), ch, cw
        r = w / h
        if r < self.ratio[0]:
            cw, ch = w, int(round(w / self.ratio[0]))
        elif r > self.ratio[1]:
            ch, cw = h, int(round(h * self.ratio[1]))
        else:
            cw, ch = w, h
        return (h - ch) // 2, (w - cw) // 2, ch, cw

    def __call__(self, img):
        from PIL import Image
        top, left, ch, cw = self._box(*img.size)
        img = img.crop((left, top, left + cw, top + ch)).resize((self.size, self.size), Image.BICUBIC)
        if torch.rand(1) < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        return _to_normalised_tensor(img)


class ResizeSquare:
    """Resize((S, S)) + ToTensor + Normalize: the custom-folder transform (main_pretrain.py:186-190)."""

    def __init__(self, size):
        self.size = int(size)

    def __call__(self, img):
        from PIL import Image
        return _to_normalised_tensor(img.resize((self.size, self.size), Image.BILINEAR))


class FlatImageTree(torch.utils.data.Dataset):
    """CustomImageDataset (main_pretrain.py:117-150): every image under `root` (recursive, sorted), a random label in [0, 1000)."""

    def __init__(self, root, transform):
        from ldmae_amd.datasets.image_folder import IMG_EXTENSIONS
        self.paths = sorted(os.path.join(d, f) for d, _, fs in os.walk(root, followlinks=True) for f in fs if f.lower().endswith(IMG_EXTENSIONS))
        if not self.paths:
            raise FileNotFoundError(f"no image files under {root}")
        self.transform = transform

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        from PIL import Image
        with open(self.paths[i], "rb") as f:
            img = Image.open(f).convert("RGB")
        return self.transform(img), int(torch.randint(0, 1000, (1,)))


def get_dataset(args):
    """main_pretrain.py:111-192: 'imagenet' in the path -> ImageFolder(<path>/train) with the random-resized-crop transform; 'laion' -> the same
    transform over the image tree (the reference's HuggingFace 'imagefolder' loader falls back to exactly that); anything else -> the
    tree resized to a square."""
    if args.synthetic:
        return _SyntheticImages(args.batch_size * args.steps_per_epoch * max(1, int(os.environ.get("WORLD_SIZE", 1))), args.input_size)
    if "imagenet" in args.data_path:
        from ldmae_amd.datasets.image_folder import ImageFolder
        return ImageFolder(os.path.join(args.data_path, "train"), transform=RandomResizedCropFlip(args.input_size))
    if "laion" in args.data_path:
        return FlatImageTree(args.data_path, RandomResizedCropFlip(args.input_size))
    return FlatImageTree(args.data_path, ResizeSquare(args.input_size))


# ----------------------------------------------------------------------------- checkpoints (VMAE/util/misc.py:468-531)
def resize_pos_embed(pos_embed, new_size):
    """[1, H*H, D] -> [1, new*new, D], bilinear, align_corners=False (misc.py:488-499)."""
    _, hw, d = pos_embed.shape
    h = int(hw ** 0.5)
    assert h * h == hw
    x = torch.nn.functional.interpolate(pos_embed.reshape(1, h, h, d).permute(0, 3, 1, 2), size=(new_size, new_size), mode="bilinear", align_corners=False)
    return x.permute(0, 2, 3, 1).reshape(1, -1, d)


def save_model(args, epoch, model, opt, scaler, rank=0):
    """checkpoint-<epoch>.pth = {model, optimizer, epoch, scaler, args} (misc.py:468-484), written by rank 0."""
    if rank != 0:
        return None
    os.makedirs(args.output_dir, exist_ok=True)
    path = os.path.join(args.output_dir, f"checkpoint-{epoch}.pth")
    # 'optimizer' in torch.optim.AdamW's own format over timm's [no_decay, decay] groups: what the reference's save_model writes and its load_model
    # feeds to optimizer.load_state_dict (misc.py:474-481, 523-525) -- a checkpoint written here resumes there and the other way round
    torch.save({"model": model.state_dict(), "optimizer": opt.torch_adamw_state_dict(), "epoch": epoch,
                "scaler": scaler.state_dict() if scaler is not None else None, "args": vars(args)}, path)
    return path


def load_model(args, model, opt, scaler, log=print):
    """--resume (misc.py:501-531): state dict with strict=False, both position embeddings resized when the grid differs, optimizer / epoch /
    scaler restored when present.  Returns the epoch to start from."""
    if not args.resume:
        return args.start_epoch
    ck = torch.load(args.resume, map_location="cpu", weights_only=False)
    sd = ck["model"]
    if sd["pos_embed"].shape[1] != model.pos_embed.shape[1]:
        new = int(model.pos_embed.shape[1] ** 0.5)
        log(f"latent resolution is {new} x {new}, reshape pos embedding ({tuple(sd['pos_embed'].shape)} -> {new * new} positions)")
        sd["pos_embed"] = resize_pos_embed(sd["pos_embed"], new)
        sd["decoder_pos_embed"] = resize_pos_embed(sd["decoder_pos_embed"], new)
        resized = True
    else:
        resized = False
    log(str(model.load_state_dict(sd, strict=False)))
    from ldmae_amd import ops
    ops.invalidate_weight_cache()
    log(f"Resume checkpoint {args.resume}")
    start = args.start_epoch
    if "optimizer" in ck and "epoch" in ck:
        if not resized:                     # (a resized grid changes the slab: the moments of the old layout do not apply)
            # the reference's save_model (misc.py:474-481) and ours both write torch.optim.AdamW.state_dict(): its moments are mapped by parameter
            # order (AdamWEMA.load_torch_adamw_state); checkpoints of earlier rounds carry the slab state with its layout record.  A state that
            # fits neither is skipped with a message (model / epoch / scaler are still restored).
            try:
                opt.load_state_dict(ck["optimizer"])
                opt.ema.copy_(opt.flat.params) if "param_groups" in ck["optimizer"] else None     # a torch state has no EMA: restart it from the weights
            except (RuntimeError, KeyError, TypeError) as ex:
                log(f"optimizer state of {args.resume} not restored ({type(ex).__name__}: {ex}); continuing with fresh moments")
        start = int(ck["epoch"]) + 1
        if scaler is not None and ck.get("scaler") is not None:
            scaler.load_state_dict(ck["scaler"])
        log("With optim & sched!")
    return start


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mae_for_ldmae_f8d16_prev")
    ap.add_argument("--input_size", type=int, default=256)
    ap.add_argument("--batch_size", type=int, default=64, help="per GPU")
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
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="fp16 = the reference's torch.amp.autocast('cuda') (engine_pretrain.py:51-57): the loss scaler then really protects against overflow")
    ap.add_argument("--print_freq", type=int, default=20)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--steps-per-epoch", type=int, default=100)
    ap.add_argument("--data_path", default="", help="'imagenet' in the path: <path>/train/<class>/<image>; otherwise any tree of images")
    ap.add_argument("--output_dir", default="./output_dir")
    ap.add_argument("--save_epochs", type=int, default=10)
    ap.add_argument("--resume", default="")
    ap.add_argument("--start_epoch", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--num_workers", type=int, default=8)
    ap.add_argument("--no_pin_mem", action="store_false", dest="pin_mem")
    # ---- the rest of VMAE/main_pretrain.py:37-91's flag set, so that the command line of VMAE/train_ae.sh:26-46 parses here
    ap.add_argument("--fixed_std", type=float, default=None, help="main_pretrain.py:43 -> the posterior KL against N(mean, fixed_std^2) (VMAE/util/misc.py:105-116)")
    ap.add_argument("--kl_form", default="vmae", choices=["vmae", "tokenizer"],
                    help="which tree's posterior KL: 'vmae' = the pre-training tree this driver mirrors (VMAE/util/misc.py:103-125: no mean^2 term; --fixed_std form), "
                         "'tokenizer' = LDMAE/tokenizer/util/misc.py:102-107 (with mean^2)")
    ap.add_argument("--no_cls", action="store_true", help="accepted (train_ae.sh passes it): this driver always builds the tokenizer without a class token")
    ap.add_argument("--smooth_output", action="store_true", help="accepted (train_ae.sh passes it): the RGB smoothing convolution is always on")
    ap.add_argument("--norm_pix_loss", action="store_true")
    ap.add_argument("--perceptual_loss_ratio", type=float, default=None, help="LPIPS term (main_pretrain.py:189-203): NOT available (needs the VGG weights); refused if given")
    ap.add_argument("--tune_decoder", action="store_true", help="stage 3 of train_ae.sh (decoder fine-tuning with LPIPS): out of scope, refused")
    ap.add_argument("--pred_with_conv", action="store_true", help="refused: not a shipped form")
    ap.add_argument("--gradual_resol", action="store_true", help="refused: not a shipped form")
    ap.add_argument("--log_dir", default=None, help="accepted and ignored (TensorBoard is out of scope)")
    ap.add_argument("--device", default="cuda", help="accepted; must be cuda")
    ap.add_argument("--world_size", type=int, default=1, help="accepted and ignored: WORLD_SIZE of the launcher's environment counts")
    ap.add_argument("--local-rank", "--local_rank", dest="local_rank_arg", type=int, default=-1, help="accepted and ignored: LOCAL_RANK of the environment counts")
    ap.add_argument("--dist_on_itp", action="store_true", help="accepted and ignored")
    ap.add_argument("--dist_url", default="env://", help="accepted and ignored")
    ap.add_argument("--pin_mem", action="store_true", dest="pin_mem_flag", help="accepted (pinning is the default here; --no_pin_mem turns it off)")
    args = ap.parse_args(argv)
    if not args.synthetic and not args.data_path:
        ap.error("--data_path (an image folder) or --synthetic")
    refused = [n for n in ("tune_decoder", "pred_with_conv", "gradual_resol") if getattr(args, n)] + (["perceptual_loss_ratio"] if args.perceptual_loss_ratio is not None else [])
    if refused or args.device != "cuda":
        ap.error(f"not available in ldmae_amd/vmae_pretrain.py: {refused or args.device} (LPIPS needs the VGG weights -- SURVEY 2.1 #18 marks the LPIPS "
                 "decoder fine-tuning OUT; pred_with_conv / gradual_resol are not shipped forms)")
    import torch.distributed as dist
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    # LDMAE_DIST_BACKEND=gloo + LDMAE_DEVICE=0: several ranks share ONE GPU (rehearsal of the multi-rank launch on a 1-GPU box), as train_accum.py
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
    log = print if rank == 0 else (lambda *a, **k: None)
    torch.manual_seed(args.seed + rank)                                       # main_pretrain.py:99-101: seed + rank
    eff = args.batch_size * args.accum_iter * world
    if args.lr is None:
        args.lr = args.blr * eff / 256                                        # main_pretrain.py:236-239
    log(f"actual lr: {args.lr:.2e}  accumulate grad iterations: {args.accum_iter}  effective batch size: {eff}")
    dataset = get_dataset(args)
    sampler = torch.utils.data.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True)    # :204-207
    if args.synthetic:
        # device-resident batches (the kernels' own speed, not the host's: collating 256 x 3 x 256 x 256 floats per step kept the GPU idle 90 % of the time)
        loader = _SyntheticLoader(args.steps_per_epoch, args.batch_size, args.input_size, args.seed + rank)
    else:
        loader = torch.utils.data.DataLoader(dataset, sampler=sampler, batch_size=args.batch_size, num_workers=args.num_workers, pin_memory=args.pin_mem,
                                             drop_last=True, multiprocessing_context="forkserver" if args.num_workers > 0 else None)
    torch.manual_seed(args.seed)                                              # every rank builds the same initial weights
    model = getattr(models_mae, args.model)(ldmae_mode=False, no_cls=True, kl_loss_weight=args.kl_loss_weight, smooth_output=True, norm_pix_loss=args.norm_pix_loss,
                                            img_size=args.input_size, fixed_std=args.fixed_std).cuda()          # main_pretrain.py:193-205
    model.kl_form = args.kl_form          # the pre-training tree's posterior KL (VMAE/util/misc.py) unless asked otherwise
    torch.manual_seed(args.seed + rank)
    opt = build_optimizer(model, args.lr, args.weight_decay)
    scaler = LossScaler(enabled=args.precision in ("bf16", "fp16"))           # main_pretrain.py:260: loss_scaler = NativeScaler()
    start = load_model(args, model, opt, scaler, log)
    reducer = GradBucketReducer(opt.flat) if world > 1 else None
    if reducer is not None:
        from ldmae_amd import ops
        ops.set_gemm_launch_mode(os.environ.get("LDMAE_DP_GEMM_LAUNCH", reducer.recommended_gemm_launch_mode()))
        reducer.broadcast_params(0)
    log(f"number of params (M): {sum(p.numel() for p in model.parameters() if p.requires_grad) / 1e6:.2f}; {len(dataset)} images; "
        f"{len(loader)} iterations per epoch and rank")
    for epoch in range(start, args.epochs):
        sampler.set_epoch(epoch)
        stats = train_one_epoch(model, loader, opt, epoch, args, log=log, scaler=scaler, reducer=reducer)
        log("Averaged stats:", stats, "skipped steps:", scaler.skipped)
        if args.output_dir and (epoch % args.save_epochs == 0 or epoch + 1 == args.epochs):
            p_ = save_model(args, epoch, model, opt, scaler, rank)
            if p_:
                log(f"Saved checkpoint to {p_}")
            if world > 1:
                dist.barrier()
    if world > 1:
        dist.destroy_process_group()
    return model, opt


if __name__ == "__main__":
    main()
