"""Flat parameter / gradient storage and the fused AdamW + EMA step.

Replaces ``torch.optim.AdamW(...).step(); opt.zero_grad(); update_ema(ema, model)`` of the reference
loop (LDMAE/train_accum.py:121, 240-243, 336-347) with ONE kernel pass over contiguous f32 buffers
(p, g, m, v, ema): 36 B/parameter of HBM traffic instead of ~150 small launches x 2 for the EMA
alone.  The same contiguous gradient buffer is what the data-parallel reducer all-reduces
(``ldmae_amd.distributed``).
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import ops

_ALIGN = 64   # elements (256 B): keeps every parameter view 16-B aligned for the GEMM operand loads


class FlatParams:
    """Moves every parameter of ``module`` into one contiguous f32 buffer (trainable first, frozen after)
    and gives trainable parameters ``.grad`` views into a second one.  Names / shapes / state_dict are
    unchanged; ``module.to(device)`` must happen BEFORE this."""

    def __init__(self, module: torch.nn.Module, group_fn=None, front_fn=None):
        """group_fn(name, param) -> int (optional): trainable parameters are laid out group by group (stable inside a group), so every
        group is ONE contiguous slice ``self.groups[g] = (lo, hi)`` -- per-group optimizer hyper-parameters (timm-style weight-decay
        groups of the VMAE pre-training) still cost one fused launch per group, not one per parameter.
        front_fn(name) -> bool (optional): matching trainable parameters are laid out FIRST (stable).  The gradient reducer cuts its buckets
        from the END of the slab backwards, in the order backward completes them, so parameters whose gradients only complete at the very end
        of backward belong at the front: the adaLN weights of a LightningDiT with batched adaLN (`adaln_first`)."""
        named = [(n, p) for n, p in module.named_parameters()]
        self.trainable = [(n, p) for n, p in named if p.requires_grad]
        gid = {n: (group_fn(n, p) if group_fn else 0) for n, p in self.trainable}
        self.trainable.sort(key=lambda np_: (gid[np_[0]], 0 if (front_fn and front_fn(np_[0])) else 1))
        self.n_front = sum(1 for n, _ in self.trainable if front_fn and front_fn(n))
        self.frozen = [(n, p) for n, p in named if not p.requires_grad]
        dev = named[0][1].device
        self.offsets = OrderedDict()
        self.groups = {}
        off = 0
        for group in (self.trainable, self.frozen):
            for n, p in group:
                if p.dtype != torch.float32:
                    raise RuntimeError(f"FlatParams: parameter {n} is {p.dtype}; master weights must be float32")
                self.offsets[n] = (off, p.numel())
                off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
                if group is self.trainable:
                    self.groups[gid[n]] = (self.groups.get(gid[n], (self.offsets[n][0], 0))[0], off)
            if group is self.trainable:
                self.n_trainable = off
        self.total = off
        self.params = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_trainable, dtype=torch.float32, device=dev)
        for n, p in self.trainable + self.frozen:
            o, k = self.offsets[n]
            self.params[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.params[o:o + k].view(p.shape)
        self.attach_grads()

    def attach_grads(self):
        for n, p in self.trainable:
            o, k = self.offsets[n]
            p.grad = self.grads[o:o + k].view(p.shape)

    def view(self, flat: torch.Tensor, name: str, shape) -> torch.Tensor:
        o, k = self.offsets[name]
        return flat[o:o + k].view(shape)


def adaln_first(name: str) -> bool:
    """front_fn for a LightningDiT trained with batched adaLN (models.lightningdit._AdaLNAllFn): the modulation Linears of the blocks, whose
    weight gradients are two GEMMs over ALL blocks at the end of backward."""
    return name.startswith("blocks.") and ".adaLN_modulation." in name


class AdamWEMA:
    """AdamW(lr, betas, eps, weight_decay) on the trainable slice + EMA(decay) over ALL parameters (the
    reference's EMA includes the frozen ``pos_embed``, train_accum.py:343-347)."""

    def __init__(self, module, lr=2e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, ema_decay=0.9999, flat: FlatParams = None,
                 group_weight_decay=None, front_fn=None):
        """group_weight_decay: {group id: weight decay} for a ``FlatParams(module, group_fn)`` layout (default: `weight_decay` everywhere)."""
        self.module = module
        self.flat = flat or FlatParams(module, front_fn=front_fn)
        self.lr, self.betas, self.eps, self.weight_decay, self.ema_decay = lr, betas, eps, weight_decay, ema_decay
        self.group_weight_decay = group_weight_decay
        f = self.flat
        self.m = torch.zeros_like(f.grads)
        self.v = torch.zeros_like(f.grads)
        self.ema = f.params.clone()          # == update_ema(ema, model, decay=0) at start (train_accum.py:166)
        self.step_count = 0

    def zero_grad(self):
        self.flat.grads.zero_()
        self.flat.attach_grads()

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        """One optimizer step + EMA update; ``grad_scale`` folds 1/world_size (and 1/accum) into the same pass."""
        f = self.flat
        self.step_count += 1
        n = f.n_trainable
        spans = [(0, n, self.weight_decay)] if not self.group_weight_decay else \
            [(lo, hi, self.group_weight_decay.get(g, self.weight_decay)) for g, (lo, hi) in sorted(f.groups.items())]
        for lo, hi, wd in spans:
            ops.adamw_ema(f.params[lo:hi], f.grads[lo:hi], self.m[lo:hi], self.v[lo:hi], self.ema[lo:hi], self.step_count, self.lr,
                          self.betas[0], self.betas[1], self.eps, wd, self.ema_decay, grad_scale)
        if f.total > n:
            ops.ema_only(self.ema[n:], f.params[n:], self.ema_decay)

    def ema_state_dict(self):
        """EMA weights under the module's parameter names (+ buffers copied as-is), loadable by
        ``model.load_state_dict`` -- the ``"ema"`` entry of the reference checkpoint (train_accum.py:275-280)."""
        sd = OrderedDict()
        params = dict(self.module.named_parameters())
        for k, v in self.module.state_dict().items():
            sd[k] = self.flat.view(self.ema, k, params[k].shape).clone() if k in params else v.clone()
        return sd

    def layout(self):
        """(name, offset, numel) of every parameter in slab order: m / v / ema are raw slabs, and the order depends on `front_fn` / `group_fn`."""
        return [(n, int(o), int(k)) for n, (o, k) in self.flat.offsets.items()]

    def state_dict(self):
        return {"step": self.step_count, "m": self.m, "v": self.v, "ema": self.ema, "layout": self.layout(),
                "hyper": dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, ema_decay=self.ema_decay)}

    def load_state_dict(self, sd):
        """Restores the moments and the EMA.  The slabs carry their layout: a state saved under ANOTHER slab order (before `adaln_first`, other
        groups) is re-ordered by name; one saved without a layout record is refused unless its order is known to be this one.  A
        ``torch.optim.AdamW.state_dict()`` ('state' + 'param_groups': what the reference's save_model writes, VMAE/util/misc.py:474-481; LDMAE/
        train_accum.py:275-280 'opt') goes through `load_torch_adamw_state`.  Everything is validated BEFORE this optimizer is touched: a state
        that is refused leaves step count, moments and EMA as they were."""
        if "param_groups" in sd and "state" in sd:
            return self.load_torch_adamw_state(sd)
        lay = sd.get("layout")
        if lay is None or not all(k in sd for k in ("step", "m", "v", "ema")):
            raise RuntimeError("AdamWEMA.load_state_dict: the saved state has no 'layout' record (written before round 5) or lacks step / m / v / "
                               "ema; the order of its slabs cannot be verified -- restore the model and the by-name EMA instead (train_accum.py does)")
        mine = self.flat.offsets
        same = [tuple(x) for x in lay] == self.layout()
        if not same and ({n for n, _, _ in lay} != set(mine) or any(mine[n][1] != k for n, _, k in lay)):
            raise RuntimeError("AdamWEMA.load_state_dict: the saved state belongs to a different set of parameters")
        if same and (sd["m"].numel() != self.m.numel() or sd["v"].numel() != self.v.numel() or sd["ema"].numel() != self.ema.numel()):
            raise RuntimeError("AdamWEMA.load_state_dict: slab sizes differ from the recorded layout")
        self.step_count = int(sd["step"])
        if same:
            self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.ema.copy_(sd["ema"])
            return
        n_tr = self.flat.n_trainable
        for n, o, k in lay:                                  # same parameters, other order: copy slab by slab, by name
            d = mine[n][0]
            self.ema[d:d + k].copy_(sd["ema"][o:o + k])
            if d < n_tr:
                self.m[d:d + k].copy_(sd["m"][o:o + k]); self.v[d:d + k].copy_(sd["v"][o:o + k])

    @torch.no_grad()
    def load_torch_adamw_state(self, sd):
        """Moments of a ``torch.optim.AdamW.state_dict()`` into the slabs.  torch numbers the parameters group by group in the order the groups
        list them; the reference builds its groups either as ONE list of ``model.parameters()`` (LDMAE/train_accum.py:121) or with timm's
        ``param_groups_weight_decay`` (VMAE/main_pretrain.py:258-259: [no_decay, decay], each in named_parameters order, no_decay = 1-D
        parameters and names ending in '.bias'; frozen parameters in neither).  Both orders are reconstructed from this module and every state
        tensor must have its parameter's size, or the state is refused untouched.  The EMA is not part of a torch optimizer: it is left as it
        is (the caller restores it by name, or keeps the freshly initialised copy of the loaded parameters)."""
        named = [(n, p) for n, p in self.module.named_parameters() if p.requires_grad]
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(groups) == 1:
            order = [n for n, _ in named]
        elif len(groups) == 2:
            nd = lambda n, p: p.ndim <= 1 or n.endswith(".bias")      # noqa: E731
            first, second = [n for n, p in named if nd(n, p)], [n for n, p in named if not nd(n, p)]
            if (len(groups[0]["params"]), len(groups[1]["params"])) != (len(first), len(second)):
                raise RuntimeError("AdamWEMA.load_torch_adamw_state: the two parameter groups are not timm's [no_decay, decay] split of this model")
            order = first + second
        else:
            raise RuntimeError(f"AdamWEMA.load_torch_adamw_state: {len(groups)} parameter groups; one (model.parameters()) or timm's two are understood")
        if len(ids) != len(order):
            raise RuntimeError(f"AdamWEMA.load_torch_adamw_state: the state lists {len(ids)} parameters, this model has {len(order)} trainable ones")
        shapes, st, steps = dict(named), sd["state"], []
        for i, n in zip(ids, order):
            e = st.get(i)
            if e is None:
                continue                                                 # never stepped (no gradient so far): zero moments, as torch would create them
            if any(k not in e for k in ("exp_avg", "exp_avg_sq", "step")) or e["exp_avg"].numel() != shapes[n].numel() or e["exp_avg_sq"].numel() != shapes[n].numel():
                raise RuntimeError(f"AdamWEMA.load_torch_adamw_state: state entry {i} does not fit parameter {n}")
            steps.append(int(e["step"]))
        self.m.zero_(); self.v.zero_()
        for i, n in zip(ids, order):
            e = st.get(i)
            if e is None:
                continue
            o, k = self.flat.offsets[n]
            self.m[o:o + k].copy_(e["exp_avg"].reshape(-1)); self.v[o:o + k].copy_(e["exp_avg_sq"].reshape(-1))
        self.step_count = max(steps) if steps else 0                     # one bias-correction step for the fused kernel: torch's per-parameter steps agree

    def torch_adamw_state_dict(self):
        """The inverse of `load_torch_adamw_state`: this optimizer's step count and moments as a ``torch.optim.AdamW.state_dict()``, so that a
        checkpoint written here resumes under the REFERENCE's drivers (VMAE/util/misc.py:523-525 `optimizer.load_state_dict(checkpoint
        ['optimizer'])`; LDMAE/train_accum.py:180 keeps its own call commented out).  Groups as the reference builds them: one list of the trainable
        parameters, or timm's [no_decay, decay] when this optimizer was built with per-group weight decay.  The group dicts carry torch's full
        key set (taken from a throw-away torch.optim.AdamW over empty tensors), the state tensors are views of the slabs in parameter shape."""
        named = [(n, p) for n, p in self.module.named_parameters() if p.requires_grad]
        if len(self.flat.groups) == 2:
            gwd = self.group_weight_decay or {}
            nd = lambda n, p: p.ndim <= 1 or n.endswith(".bias")      # noqa: E731
            first, second = [n for n, p in named if nd(n, p)], [n for n, p in named if not nd(n, p)]
            f = self.flat
            gid = lambda n: next(g for g, (lo, hi) in f.groups.items() if lo <= f.offsets[n][0] < hi)      # noqa: E731
            wds = [{gwd.get(gid(n), self.weight_decay) for n in part} for part in (first, second)]
            if any(len(w) > 1 for w in wds):
                raise RuntimeError("AdamWEMA.torch_adamw_state_dict: the weight-decay groups of this optimizer are not timm's [no_decay, decay] split")
            parts = [(first, wds[0].pop() if wds[0] else 0.0), (second, wds[1].pop() if wds[1] else self.weight_decay)]
        elif len(self.flat.groups) > 2:
            raise RuntimeError(f"AdamWEMA.torch_adamw_state_dict: {len(self.flat.groups)} parameter groups; one or timm's two can be written")
        else:
            parts = [([n for n, _ in named], self.weight_decay)]
        dummy = {n: torch.nn.Parameter(torch.empty(0)) for n, _ in named}
        shell = torch.optim.AdamW([{"params": [dummy[n] for n in part], "weight_decay": wd} for part, wd in parts],
                                  lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.weight_decay)
        sd = shell.state_dict()
        shapes = dict(named)
        if self.step_count > 0:
            order = [n for part, _ in parts for n in part]
            for i, n in enumerate(order):
                sd["state"][i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.flat.view(self.m, n, shapes[n].shape),
                                  "exp_avg_sq": self.flat.view(self.v, n, shapes[n].shape)}
        return sd

    @torch.no_grad()
    def swap_in_ema(self):
        """Exchange the live parameters with their EMA (sampling from the EMA weights inside a training process; call again to swap
        back).  Writes the slab directly, so the forward-only weight-copy cache is invalidated."""
        f = self.flat
        tmp = f.params.clone()
        f.params.copy_(self.ema)
        self.ema.copy_(tmp)
        ops.invalidate_weight_cache()
