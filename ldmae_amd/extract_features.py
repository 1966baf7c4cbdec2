#!/usr/bin/env python3
"""Latent-shard WRITER on the HIP kernels -- counterpart of the reference's ``extract_features.py``: run the VMAE
tokenizer over an image folder twice (plain and horizontally flipped), write ``latents_rank{r:02d}_shard{n:03d}.safetensors`` with the keys
``latents`` / ``latents_flip`` / ``labels`` and the metadata strings the reference writes (extract_features.py:163-212), then let rank 0
build ``latents_stats.pt`` through ImgLatentDataset (:215-218).  ``datasets.img_latent_dataset.ImgLatentDataset`` reads the result.

    torchrun --nproc_per_node N -m ldmae_amd.extract_features --config configs/imagenet/lightningdit_b_vmae_f8d16_cfg.yaml
    python -m ldmae_amd.extract_features --config <cfg> --synthetic 512          # no image folder: seeded random images

Same as the reference: one process per GPU, DistributedSampler(shuffle=False) so rank r sees samples r, r + world, ...; a shard is closed
after ``10000 // batch_size`` batches; with ``data.sample`` in the config the shards hold the posterior MOMENTS ``_encode`` returns
([n, 2 * latent, h, w]: the dataset samples from them per item), otherwise the posterior mode.  Different on purpose: the encoder runs on
this package's kernels (bf16 activations with --precision bf16, the f32 MFMA path by default like the reference's un-autocast call), each
batch's latents go to PINNED host memory on a side stream while the next batch encodes (the reference keeps up to 2 x 1.3 GB per shard on
the GPU and copies at shard end), and only the vmae tokenizer exists here (the reference's SD-VAE branch is out of scope)."""
import argparse
import os
import sys
from datetime import datetime

import torch
import torch.distributed as dist
import yaml
from safetensors.torch import save_file
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

_HERE = os.path.dirname(os.path.abspath(__file__))
for p in (_HERE, os.path.dirname(_HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

from ldmae_amd.datasets.image_folder import ImageFolder               # noqa: E402
from ldmae_amd.datasets.img_latent_dataset import ImgLatentDataset    # noqa: E402
from ldmae_amd.tokenizer import models_mae                            # noqa: E402


def shard_name(output_dir, rank, n):
    return os.path.join(output_dir, f"latents_rank{rank:02d}_shard{n:03d}.safetensors")        # extract_features.py:176


class _HostSink:
    """Device latents -> pinned host buffers on a side stream; ``take()`` hands back one CPU tensor per key once its copies are done."""

    def __init__(self):
        self.stream, self.parts = torch.cuda.Stream(), {"latents": [], "latents_flip": [], "labels": []}

    def put(self, key, t):
        if not t.is_cuda:
            self.parts[key].append((t.clone(), None))
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        t.record_stream(self.stream)
        self.parts[key].append((h, ev))

    def batches(self):
        return len(self.parts["latents"])

    def take(self):
        out = {}
        for k, lst in self.parts.items():
            for _, ev in lst:
                if ev is not None:
                    ev.synchronize()
            out[k] = torch.cat([h for h, _ in lst], dim=0).contiguous()
            self.parts[k] = []
        return out


def extract(tokenizer, loaders, output_dir, rank=0, batch_size=64, sample=True, shard_images=10000, log=print):
    """extract_features.py:141-212.  `loaders` = (plain, flipped) over the same samples in the same order.  Returns the shard paths."""
    os.makedirs(output_dir, exist_ok=True)
    sink, written, run_images = _HostSink(), [], 0
    total = len(loaders[0].dataset)

    def flush():
        d = sink.take()
        path = shard_name(output_dir, rank, len(written))
        lat = d["latents"]
        save_file(d, path, metadata={"total_size": f"{lat.shape[0]}", "dtype": f"{lat.dtype}", "device": f"{lat.device}"})
        written.append(path)
        if rank == 0:
            log(f"Saved {path}  " + "  ".join(f"{k} {tuple(v.shape)}" for k, v in d.items()))

    for batch_idx, pair in enumerate(zip(*loaders)):
        run_images += pair[0][0].shape[0]
        if rank == 0 and (run_images % 100 == 0 or batch_idx % 8 == 7):      # the reference's rule (:147) never fires at batch sizes like 64 / 256
            log(f"{datetime.now()} processing {run_images} of {total} images")
        for which, (x, y) in enumerate(pair):
            # extract_features.py:2-3 of the reference set allow_tf32 at import: the tokenizer's f32 `_encode` runs TF32-class (fp16 operands =
            # TF32's mantissa, f32 accumulation; 5.6e-4 against exact f32); LDMAE_TF32=0 keeps the exact-f32 kernels
            with torch.no_grad(), models_mae.reference_tf32():
                x = x.cuda(non_blocking=True)
                z = tokenizer._encode(x) if sample else tokenizer.encode(x).latent_dist.mode()
                z = z.float().contiguous()
            if batch_idx == 0 and which == 0 and rank == 0:
                log(f"latent shape {tuple(z.shape)} dtype {z.dtype}")
            sink.put("latents" if which == 0 else "latents_flip", z)
            if which == 0:
                sink.put("labels", y)
        if sink.batches() == shard_images // batch_size:
            flush()
    if sink.batches() > 0:                       # remainder: fewer than shard_images images (:186-212)
        flush()
    return written


class _SyntheticImages(Dataset):
    """Seeded random images in [-1, 1] (already 'transformed'); `flip` mirrors them, as the p_hflip = 1 transform does."""

    def __init__(self, n, size, flip, num_classes=1000):
        self.n, self.size, self.flip, self.num_classes = n, size, flip, num_classes

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(i)
        x = torch.rand(3, self.size, self.size, generator=g) * 2 - 1
        return (x.flip(-1) if self.flip else x), int(torch.randint(0, self.num_classes, (1,), generator=g))


def main(args, cfg):
    if os.environ.get("LDMAE_DUMP_AFTER"):            # debugging aid: all thread stacks after N seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LDMAE_DUMP_AFTER"]), exit=True)
    if not torch.cuda.is_available():
        raise RuntimeError("extract_features needs a GPU: there is no CPU path in this package")
    if "RANK" in os.environ:                                           # launched by torchrun / accelerate (:27-41)
        dist.init_process_group("nccl" if os.environ.get("LDMAE_DIST_BACKEND", "nccl") == "nccl" else "gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        print("Not launched under torch.distributed: running in local mode.")
        rank, world = 0, 1
    device = int(os.environ.get("LDMAE_DEVICE", rank % torch.cuda.device_count()))
    torch.manual_seed(args.seed)                 # the same initial weights on every rank when no checkpoint is loaded (--synthetic)
    torch.cuda.set_device(device)
    model_name = cfg["vae"]["model_name"].split("_")[0]
    if model_name != "vmae":
        raise NotImplementedError(f"tokenizer '{model_name}': only the vmae tokenizer is built here (the SD-VAE branch is out of scope)")
    sample = "sample" in cfg["data"]                                   # :51, :150-153: key presence, not its value
    out_root = args.output_dir or os.path.dirname(cfg["data"]["origin_path"])
    output_dir = os.path.join(out_root, f"{model_name}_feature_{cfg['data'].get('name', 'imagenet')}_{args.data_split}_{args.image_size}")
    if sample:
        output_dir += "_sample"
    tokenizer = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=True, no_cls=True, kl_loss_weight=True, smooth_output=True, img_size=args.image_size)
    chkpt = cfg["vae"].get("weight_path")
    if chkpt and os.path.exists(chkpt):
        msg = tokenizer.load_state_dict(torch.load(chkpt, map_location="cpu")["model"], strict=False)
        if rank == 0:
            print(model_name, msg)
    elif not args.synthetic:
        raise FileNotFoundError(f"vae.weight_path {chkpt!r} not found")
    tokenizer = tokenizer.cuda().eval()
    torch.manual_seed(args.seed + rank)          # extract_features.py:33,41: per-rank seed for everything after the model
    if args.precision == "bf16":
        tokenizer.set_precision(torch.bfloat16)
    if args.synthetic:
        sets = [_SyntheticImages(args.synthetic, args.image_size, flip) for flip in (False, True)]
    else:
        root = os.path.join(cfg["data"]["origin_path"], args.data_split)
        sets = [ImageFolder(root, transform=tokenizer.img_transform(p_hflip=p, img_size=args.image_size)) for p in (0.0, 1.0)]
    # workers from a FORK SERVER, not forked from this process: it has the GPU runtime, its helper threads and (from the first loader on) a
    # pin-memory thread running, and a child forked while one of them holds a lock inherits that lock forever -- seen as loader workers that
    # never deliver a batch (2 of 6 runs with 2 x 8 workers).  Datasets and transforms are plain picklable objects.
    mp_ctx = "forkserver" if args.num_workers > 0 else None
    loaders = [DataLoader(ds, batch_size=args.batch_size, shuffle=False,
                          sampler=DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False, seed=args.seed),
                          num_workers=args.num_workers, pin_memory=True, drop_last=False, multiprocessing_context=mp_ctx,
                          persistent_workers=False) for ds in sets]
    if rank == 0:
        print(f"Total data in one loop: {len(sets[0])}")
    files = extract(tokenizer, loaders, output_dir, rank=rank, batch_size=args.batch_size, sample=sample)
    if world > 1:
        dist.barrier()
    if rank == 0:                                                     # computes and caches latents_stats.pt (:215-218)
        ImgLatentDataset(output_dir, latent_norm=True, sample=cfg["data"]["sample"] if sample else False)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return output_dir, files


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_split", type=str, default="train")
    ap.add_argument("--output_dir", type=str, default=None, help="default: dirname(data.origin_path), like the reference")
    ap.add_argument("--output_path", type=str, default="/data/dataset/imagenet/", help="accepted and ignored, as in the reference (extract_features.py:226 parses it; :45 derives the output from data.origin_path)")
    ap.add_argument("--image_size", type=int, default=256)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--num_workers", type=int, default=8)
    ap.add_argument("--config", type=str, default="configs/debug.yaml")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--synthetic", type=int, default=0, help="N seeded random images instead of an image folder")
    a = ap.parse_args()
    with open(a.config) as f:
        main(a, yaml.safe_load(f))
