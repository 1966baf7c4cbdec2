"""Oracle (test infrastructure): one optimizer step of LDMAE/train_accum.py.

Restates the loop body train_accum.py:204-246 (loss -> backward -> AdamW ->
zero_grad -> EMA) on the functional oracle model.  AdamW is restated
explicitly (torch.optim.AdamW, torch==2.2.0 pinned by requirements.txt:4,
single-tensor formula, amsgrad=False, maximize=False) so the HIP fused
AdamW+EMA kernel has a line-by-line checker; tests/test_oracle_golden.py
checks the restatement against torch.optim.AdamW itself.
"""
from __future__ import annotations

import math
import time

import numpy as np
import torch

from . import dit, transport


def trainable_keys(cfg: dit.DiTConfig):
    """Everything in named_parameters() except the frozen pos_embed (lightningdit.py:314)."""
    return [k for k in dit.param_shapes(cfg) if k != "pos_embed"]


class AdamWState:
    def __init__(self, keys, sd):
        self.m = {k: torch.zeros_like(sd[k]) for k in keys}
        self.v = {k: torch.zeros_like(sd[k]) for k in keys}
        self.step = 0


def adamw_step(sd, grads, st: AdamWState, lr=2e-4, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.0):
    """torch/optim/adamw.py _single_tensor_adamw; hyper-parameters from train_accum.py:121."""
    st.step += 1
    bc1 = 1 - beta1 ** st.step
    bc2 = 1 - beta2 ** st.step
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for k, g in grads.items():
        p = sd[k]
        p.mul_(1 - lr * weight_decay)
        st.m[k].lerp_(g, 1 - beta1)
        st.v[k].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (st.v[k].sqrt() / bc2_sqrt).add_(eps)
        p.addcdiv_(st.m[k], denom, value=-step_size)


def ema_update(ema_sd, sd, keys, decay=0.9999):
    """train_accum.py:336-347: over named_parameters *including* frozen pos_embed."""
    for k in keys:
        ema_sd[k].mul_(decay).add_(sd[k], alpha=1 - decay)


def loss_and_grads(sd, cfg, x1, y, t, x0, drop_ids):
    keys = trainable_keys(cfg)
    leaves = {k: sd[k].detach().requires_grad_(True) for k in keys}
    full = dict(sd)
    full.update(leaves)
    terms = transport.training_losses(lambda xt, tt: dit.dit_forward(full, xt, tt, y, cfg, True, drop_ids), x1, t, x0)
    loss = terms["loss"].mean()
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys])
    return loss.detach(), dict(zip(keys, grads)), terms["pred"].detach()


def train_steps(sd, cfg, batches, lr=2e-4, beta2=0.95, ema_decay=0.9999, log=None):
    """Run len(batches) optimizer steps.  Each batch is (x1, y, t, x0, drop_ids) drawn
    by the caller on the host in the reference's order (x0 torch RNG, t numpy RNG,
    label-drop torch RNG; SURVEY.md §7 'RNG parity').  Returns losses and mutates sd."""
    keys = trainable_keys(cfg)
    ema_keys = keys + ["pos_embed"]
    ema = {k: sd[k].clone() for k in ema_keys}
    st = AdamWState(keys, sd)
    losses, times = [], []
    for i, (x1, y, t, x0, drop) in enumerate(batches):
        t0 = time.perf_counter()
        loss, grads, _ = loss_and_grads(sd, cfg, x1, y, t, x0, drop)
        with torch.no_grad():
            adamw_step(sd, grads, st, lr=lr, beta2=beta2)
            ema_update(ema, sd, ema_keys, ema_decay)
        times.append(time.perf_counter() - t0)
        losses.append(float(loss))
        if log:
            log(i, losses[-1], times[-1])
    return losses, ema, times


def draw_batch(B, cfg: dit.DiTConfig, gen_seeded: bool = True):
    """Host-side draws in the reference's order for one micro-step on synthetic
    latents (SURVEY.md §8d cfg 1): x1, y from the torch RNG; then x0 (randn_like),
    t (numpy), label-drop (torch.rand(B) < p)."""
    x1 = torch.randn(B, cfg.in_channels, cfg.input_size, cfg.input_size)
    y = torch.randint(0, cfg.num_classes, (B,))
    t, x0, _ = transport.sample(x1)
    drop = torch.rand(B) < cfg.class_dropout_prob
    return x1, y, t, x0, drop
