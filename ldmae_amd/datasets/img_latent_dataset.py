"""Latent dataset over safetensors shards (reference: LDMAE/datasets/img_latent_dataset.py; writer extract_features.py:163-212).

Shard keys: ``latents`` / ``latents_flip`` [n, 2*C, h, w] (posterior moments when ``sample``) and ``labels`` [n];
``latents_stats.pt`` caches the channel-wise {mean, std} [1, C, 1, 1].  Host I/O only -- not on the kernel path.
"""
import os
from glob import glob

import numpy as np
import torch
from safetensors import safe_open
from torch.utils.data import Dataset

from ldmae_amd.tokenizer.util.misc import DiagonalGaussianDistribution


class ImgLatentDataset(Dataset):
    def __init__(self, data_dir, latent_norm=True, latent_multiplier=1.0, sample=False, raw=False):
        """raw=True (not in the reference): __getitem__ stops after the shard read and the flip pick and returns the stored tensor
        (moments when `sample`); posterior sampling, normalisation and the multiplier then run per BATCH on the GPU -- LatentPrologue."""
        self.data_dir, self.latent_norm, self.latent_multiplier, self.sample, self.raw = data_dir, latent_norm, latent_multiplier, sample, raw
        self.files = sorted(glob(os.path.join(data_dir, "*.safetensors")))
        self.index = []                                    # global idx -> (file, idx in file)
        for f in self.files:
            with safe_open(f, framework="pt", device="cpu") as h:
                n = h.get_slice("labels").get_shape()[0]
            self.index.extend((f, i) for i in range(n))
        if latent_norm:
            self._latent_mean, self._latent_std = self.get_latent_stats()

    def get_latent_stats(self):
        cache = os.path.join(self.data_dir, "latents_stats.pt")
        if os.path.exists(cache):
            st = torch.load(cache)
        else:
            st = self.compute_latent_stats()
            torch.save(st, cache)
        return st["mean"], st["std"]

    def _read(self, f, i, key):
        with safe_open(f, framework="pt", device="cpu") as h:
            return h.get_slice(key)[i:i + 1]

    def compute_latent_stats(self):
        n = min(10000, len(self.index))
        lat = []
        for idx in np.random.choice(len(self.index), n, replace=False):
            f, i = self.index[idx]
            x = self._read(f, i, "latents")
            lat.append(DiagonalGaussianDistribution(x).sample() if self.sample else x)
        lat = torch.cat(lat, dim=0)
        return {"mean": lat.mean(dim=[0, 2, 3], keepdim=True), "std": lat.std(dim=[0, 2, 3], keepdim=True)}

    def __len__(self):
        return len(self.index)

    def __getitem__(self, idx):
        f, i = self.index[idx]
        key = "latents" if np.random.uniform(0, 1) > 0.5 else "latents_flip"      # per-item flip pick (:79)
        x, y = self._read(f, i, key), self._read(f, i, "labels")
        if self.raw:
            return x.squeeze(0), y.squeeze(0)
        if self.sample:
            x = DiagonalGaussianDistribution(x).sample()
        if self.latent_norm:
            x = (x - self._latent_mean) / self._latent_std
        return (x * self.latent_multiplier).squeeze(0), y.squeeze(0)


class LatentPrologue(torch.nn.Module):
    """The part of ImgLatentDataset.__getitem__ after the shard read (:82-93), as one HIP kernel per batch (ops.latent_prologue,
    SURVEY 8f rank 3 "fused GPU prologue"): x = ((mean + std * noise) - latent_mean) / latent_std * multiplier.  Built from a dataset
    opened with raw=True; the noise comes from the DEVICE generator (the reference draws it on the host inside the DataLoader workers:
    same distribution, a different stream of numbers) unless the caller passes it."""

    def __init__(self, dataset):
        super().__init__()
        self.sample, self.multiplier, self.norm = bool(dataset.sample), float(dataset.latent_multiplier), bool(dataset.latent_norm)
        if self.norm:
            self.register_buffer("latent_mean", dataset._latent_mean.float().reshape(-1).clone())
            self.register_buffer("latent_std", dataset._latent_std.float().reshape(-1).clone())

    def forward(self, stored, noise=None, generator=None):
        from ldmae_amd import ops
        if self.sample and noise is None:
            B, C2, H, W = stored.shape
            noise = torch.randn(B, C2 // 2, H, W, device=stored.device, dtype=torch.float32, generator=generator)
        return ops.latent_prologue(stored, noise, self.latent_mean if self.norm else None, self.latent_std if self.norm else None,
                                   self.multiplier, self.sample)


class SyntheticLatentDataset(Dataset):
    """N(0,1) latents + uniform labels (what the channel-normalised real latents look like); for benchmarks / smoke runs."""

    def __init__(self, length=1 << 20, channels=16, size=32, num_classes=1000, seed=0):
        self.length, self.shape, self.num_classes, self.seed = length, (channels, size, size), num_classes, seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        return torch.randn(self.shape, generator=g), torch.randint(0, self.num_classes, (1,), generator=g)[0]
