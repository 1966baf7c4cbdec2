#!/usr/bin/env python3
"""Headline benchmark: LightningDiT-B/1 (f8d16) flow-matching TRAIN STEP throughput on synthetic 32x32x16 latents.

    python bench.py --gpus N --steps K --warmup W

N > 1: either the driver starts the ranks (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...: RANK /
LOCAL_RANK / WORLD_SIZE in the environment), or -- WORLD_SIZE unset -- bench.py starts them itself the same way (the parent never
touches the GPU, starts torch.distributed.run as a child, relays rank 0's JSON line and exits with the child's code).  One rank per
GPU over RCCL; a run whose world size differs from --gpus fails instead of reporting a mislabelled number.

One "step" = what LDMAE/train_accum.py:204-246 does per optimizer step at gradient_accumulation_steps=1:
transport.training_losses (x0 ~ N(0,I), t ~ logit-normal, xt, ut) -> model fwd (bf16 autocast) -> velocity MSE ->
backward (+ bucketed RCCL all-reduce overlapped on a side stream when N > 1) -> fused AdamW + EMA.  Inputs (latents,
labels) are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

METRIC = "DiT-B train-step images/sec (32×32×16 latents) at 1/2/4/8 MI355X"
FLOPS_PER_IMAGE = 638.22e9        # 212.74 GFLOP fwd x 3 (fwd+bwd), algorithmic
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md chip table)
NOMINAL_CLOCK_GHZ = 2.4           # the shader clock that peak is quoted at (2.4 GHz x 256 CUs x 4 SIMDs x 1024 FLOP/clk)


def build(device, per_gpu_batch, force_hooks=False, celeba=False):
    from ldmae_amd.distributed import GradBucketReducer
    from ldmae_amd.models.lightningdit import LightningDiT_models
    from ldmae_amd.optim import AdamWEMA, adaln_first
    from ldmae_amd.transport import create_transport
    # configs/imagenet/lightningdit_b_vmae_f8d16_cfg.yaml: model / optimizer / transport sections
    # celeba: configs/celeba_hq/lightningdit_b_vmae_f8d16_cfg.yaml through train_accum.py:79-90 (num_classes 1 -> class_dropout_prob 0, use_qknorm false)
    model = LightningDiT_models["LightningDiT-B/1"](input_size=32, num_classes=1 if celeba else 1000, use_qknorm=not celeba, use_swiglu=True, use_rope=True,
                                                     use_rmsnorm=True, wo_shift=False, in_channels=16, use_checkpoint=False,
                                                     class_dropout_prob=0 if celeba else 0.1)
    # the reference zero-initialises adaLN / final layer; random-init them so every kernel does real work from step 0
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "adaLN_modulation" in n or n.startswith("final_layer.linear"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    model = model.to(device).train()
    # the adaLN weights FIRST in the slab: the batched adaLN backward finishes them at the very end of backward, and the reducer's buckets are
    # cut from the end of the slab backwards -- so N > 1 runs the same program as N = 1 (ldmae_amd/train_accum.py does the same)
    opt = AdamWEMA(model, lr=2e-4, betas=(0.9, 0.95), weight_decay=0.0, ema_decay=0.9999, front_fn=adaln_first)
    reducer = GradBucketReducer(opt.flat, force_hooks=force_hooks)
    # batched adaLN under data parallelism: by size until a multi-rank run has measured the exposed tail (distributed.batched_adaln_pays;
    # B/1: 170 MB -> on, XL/1: 890 MB -> off); LDMAE_BATCHED_ADALN=0|1 overrides -- as ldmae_amd/train_accum.py
    from ldmae_amd.distributed import batched_adaln_pays
    env = os.environ.get("LDMAE_BATCHED_ADALN")
    adaln_bytes = sum(p.numel() * 4 for n, p in opt.flat.trainable if adaln_first(n))
    model.batched_adaln = (env != "0") if env is not None else batched_adaln_pays(adaln_bytes, len(model.blocks), reducer.world)
    model.direct_param_grads = os.environ.get("LDMAE_DIRECT_GRADS", "1") != "0"       # every .grad is a slab view and backward is a plain loss.backward(): dW goes straight into the slab
    reducer.broadcast_params(0)
    opt.ema.copy_(opt.flat.params)
    transport = create_transport("Linear", "velocity", None, None, None, use_cosine_loss=False, use_lognorm=True)
    return model, opt, reducer, transport


def train_step(model, opt, reducer, transport, x, y):
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = transport.training_losses(model, x, dict(y=y))["loss"].mean()
    loss.backward()
    opt.step(grad_scale=reducer.finish())
    return loss


def kernel_source_sha(files=("gemm.hip", "gemm_nt_lines.hip", "gemm_nt_common.h", "common.h")):
    """sha256[:16] of the dominant kernel's sources: a committed PMC figure is only quoted for the kernel it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "ldmae_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_pmc():
    """The newest committed PMC run of this bench (tools/pmc_bench.sh -> profiles/rNN_pmc_bench.json).  PMC needs rocprofv3 around the
    process, so the counters cannot be read inside this run; the committed figures are quoted ONLY while the kernel sources still hash to the
    values recorded with them.  Returns (dict or None, relative path)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bench.json")), reverse=True):
        try:
            with open(f) as fh:
                return json.load(fh), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def measured_traffic():
    """Counters of the dominant kernel (bf16 NT GEMM) from the committed PMC run: HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate
    passes, the gfx950 correction of MI355X_MICROARCH.md), matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 SIMDs x
    GRBM_GUI_ACTIVE / 8)) and the clock the chip held.  Null when the kernel sources have changed since.  Returns (dict, note)."""
    d, rel = committed_pmc()
    none = {"traffic": None, "mfma_busy": None, "clock_ghz": None}
    if d is None:
        return none, None
    if d.get("kernel_source_sha") != kernel_source_sha():
        return none, f"{rel} was measured on other kernel sources (sha {d.get('kernel_source_sha')}); not quoted"
    g = d["gemm_nt"]
    return ({"traffic": round(g["hbm_bytes_per_launch"]), "mfma_busy": round(g["mfma_busy"], 4) if "mfma_busy" in g else None,
             "clock_ghz": round(g["clock_ghz"], 3) if "clock_ghz" in g else None},
            rel + " (committed rocprofv3 --pmc run of this bench on these kernel sources)")


# the dominant HBM-bound kernel of the step: rmsnorm_mod_bwd_kernel<GATE> (RMSNorm + modulate backward with the gated-residual backward of the
# branch below it, elementwise.hip).  Algorithmic bytes per launch at bs = 256 (M = 262144 rows of D = 768): reads dout (bf16),
# x (f32), the running dx (f32), y (bf16); writes dx (f32), the branch gradient (bf16): 2 + 4 + 4 + 2 + 4 + 2 = 18 B per element.
ROWWISE_BYTES_PER_ELEMENT = 18


def hbm_block(batch, tokens=1024, width=768):
    """`hbm` block of the JSON line: achieved HBM rate of the dominant HBM-bound kernel = algorithmic bytes per launch / its average duration in
    the committed rocprofv3 run (and the PMC-measured bytes beside it), against the 8 TB/s peak.  Null figures when elementwise.hip changed."""
    d, rel = committed_pmc()
    algo = batch * tokens * width * ROWWISE_BYTES_PER_ELEMENT
    blk = {"bound": "hbm", "kernel": "rmsnorm_mod_bwd_kernel<GATE> (RMSNorm + modulate backward + gated-residual backward, 24 launches per step)",
           "algorithmic_bytes_per_launch": algo, "peak": 8000.0, "unit": "GB/s", "achieved": None, "frac": None, "traffic": None, "source": None}
    if d is None or "rmsnorm_mod_bwd" not in d:
        return blk
    if d.get("rowwise_source_sha") != kernel_source_sha(("elementwise.hip", "common.h")):
        blk["source"] = f"{rel} was measured on other kernel sources; not quoted"
        return blk
    r = d["rmsnorm_mod_bwd"]
    us = r.get("avg_duration_us_in_FETCH_SIZE_pass") or r.get("avg_duration_us_in_WRITE_SIZE_pass")
    if us and batch == 256:
        blk["achieved"] = round(algo / (us * 1e-6) / 1e9, 1)
        blk["frac"] = round(blk["achieved"] / 8000.0, 4)
        blk["avg_launch_us"] = round(us, 1)
    blk["traffic"] = round(r["hbm_bytes_per_launch"]) if "hbm_bytes_per_launch" in r else None
    blk["source"] = rel + " (rocprofv3 --pmc + --kernel-trace over this bench: duration and FETCH_SIZE x2 + WRITE_SIZE per launch)"
    return blk


def usable_cores():
    """(threads, rule): host cores the CPU baseline runs on = min(scheduler affinity, cgroup CPU quota, CPUs visible / GPUs visible).
    The last term is the share of a multi-GPU host that belongs to this job's GPUs (a 1-GPU box carved out of a 256-CPU, 8-GPU node
    shows every CPU in os.cpu_count() although only its 1/8 share is schedulable without contention)."""
    ncpu = os.cpu_count() or 1
    n, rule = ncpu, [f"os.cpu_count()={ncpu}"]
    try:
        aff = len(os.sched_getaffinity(0))
        rule.append(f"affinity={aff}")
        n = min(n, aff)
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            q = None
            if path.endswith("cpu.max") and parts[0] != "max":
                q = max(1, int(int(parts[0]) / int(parts[1]) + 0.5))
            elif path.endswith("cfs_quota_us") and int(parts[0]) > 0:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    q = max(1, int(int(parts[0]) / int(g.read()) + 0.5))
            if q:
                rule.append(f"cgroup_quota={q}")
                n = min(n, q)
        except Exception:
            pass
    # GPUs physically on the host (KFD topology nodes with SIMDs), not the ones this container may use
    host_gpus = 0
    try:
        import glob
        for pth in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            with open(pth) as f:
                if any(l.startswith("simd_count") and int(l.split()[1]) > 0 for l in f):
                    host_gpus += 1
    except Exception:
        pass
    vis = max(1, torch.cuda.device_count())
    if host_gpus > vis:
        share = max(1, ncpu * vis // host_gpus)
        rule.append(f"host share = {ncpu} CPUs x {vis}/{host_gpus} GPUs = {share}")
        n = min(n, share)
    elif n == ncpu and ncpu > 64:
        # nothing narrowed a very large host and its GPU count is hidden from this container: assume the usual 8-GPU node
        share = max(1, ncpu * vis // 8)
        rule.append(f"assumed 8-GPU host share = {ncpu} CPUs x {vis}/8 = {share}")
        n = min(n, share)
    return max(1, n), "min(" + ", ".join(rule) + ")"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(max_seconds=30.0):
    """The CPU oracle (a port of the reference step; the reference itself cannot travel) on this box's host cores, SURVEY 8(d):
    BASELINE config 1 -- LightningDiT-B/1, bs=4, fp32, AdamW + EMA, eager, torch.set_num_threads(os.cpu_count()).
    Bounded sample: 1 untimed step + as many timed steps (at most 8) as fit in `max_seconds` of CPU work."""
    from oracle import dit as odit, train as otrain
    cfg = odit.DiTConfig(**odit.DIT_B_1)
    # every host core THIS process may run on: the GPU box shows all of the host's CPUs in os.cpu_count() but pins a 1-GPU job to its
    # share (16); asking torch for 256 threads there oversubscribes 16 cores and a single step takes minutes
    ncores, rule = usable_cores()
    torch.set_num_threads(ncores)
    torch.manual_seed(0)
    np.random.seed(0)
    sd = odit.init_weights(cfg)
    for k in sd:
        if "adaLN_modulation" in k or k.startswith("final_layer.linear"):
            sd[k] = torch.randn(sd[k].shape) * 0.02
    times = []
    t_begin = time.perf_counter()
    batches = [otrain.draw_batch(4, cfg) for _ in range(9)]
    keys = otrain.trainable_keys(cfg)
    st = otrain.AdamWState(keys, sd)
    ema = {k: sd[k].clone() for k in keys + ["pos_embed"]}
    for i, (x1, y, t, x0, drop) in enumerate(batches):
        t0 = time.perf_counter()
        _, grads, _ = otrain.loss_and_grads(sd, cfg, x1, y, t, x0, drop)
        with torch.no_grad():
            otrain.adamw_step(sd, grads, st)
            otrain.ema_update(ema, sd, keys + ["pos_embed"])
        if i > 0:
            times.append(time.perf_counter() - t0)
        if times and time.perf_counter() - t_begin + times[-1] > max_seconds:
            break
    sps = float(np.median(times))
    return {"value": round(4.0 / sps, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "cpu": cpu_model(),
            "sample": f"LightningDiT-B/1 bs=4 fp32 eager CPU step (fwd+bwd+AdamW+EMA), median of {len(times)} timed steps after 1 warm-up "
                      f"(bounded to ~{max_seconds:.0f} s of CPU work), {sps:.2f} s/step, torch {torch.__version__}, "
                      f"{torch.get_num_threads()} threads = {rule}"}


def profiler_attached():
    """rocprofv3 (or another rocprofiler-sdk tool) is wrapped around this process: its preloaded library lives in every child too, so
    nothing here may spawn processes then, and sampler threads would only perturb the counters."""
    env = os.environ
    return any(k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF_")) for k in env) or "rocprofiler" in env.get("LD_PRELOAD", "")


def _hwmon_dir(device_index):
    """sysfs hwmon directory of the amdgpu card behind HIP device `device_index` (matched by PCI address), or None."""
    import glob
    want = None
    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except Exception:
        pass
    cands = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if not os.path.exists(os.path.join(hw, "power1_average")) and not os.path.exists(os.path.join(hw, "power1_input")):
            continue
        slot = ""
        try:
            with open(os.path.join(os.path.dirname(os.path.dirname(hw)), "uevent")) as f:
                slot = next((l.split("=", 1)[1].strip() for l in f if l.startswith("PCI_SLOT_NAME=")), "")
        except Exception:
            pass
        cands.append((hw, slot))
    for hw, slot in cands:
        if want and slot.lower().startswith(want):
            return hw
    return cands[0][0] if len(cands) == 1 else None


class PowerSampler:
    """Package power and shader clock read IN PROCESS from the card's sysfs hwmon files (power1_average in uW, freq1_input in Hz) at
    ~4 Hz on a host thread while the timed loop runs: no child process (a rocm-smi child would inherit a profiler's preloaded library
    and re-exec through `env` with the GPU initialised -- forbidden on this pool), no GPU work, no synchronisation.  Off with
    --no-power, and automatically whenever a profiler is attached.  Reported beside the roofline because the MFMA kernels of this step
    run at the package power limit (profiles/r02_pmc_mfma.md): the clock they hold is part of the measurement."""

    def __init__(self, enabled=True, device_index=0):
        import threading
        self.hw = _hwmon_dir(device_index) if enabled and not profiler_attached() else None
        self.enabled = self.hw is not None
        self.samples, self._stop, self._th = [], False, threading.Thread(target=self._run, daemon=True)

    def _read(self, name):
        try:
            with open(os.path.join(self.hw, name)) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop:
            w = self._read("power1_average")
            if w is None:
                w = self._read("power1_input")
            c = self._read("freq1_input")
            if w is not None:
                self.samples.append((w / 1e6, c / 1e6 if c else None))
            time.sleep(0.25)

    def __enter__(self):
        if self.enabled:
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self.enabled:
            self._th.join(timeout=2)

    def summary(self):
        s = self.samples[1:] if len(self.samples) > 2 else self.samples      # the first sample may predate the loop
        if not s:
            return None
        w = [x[0] for x in s]
        c = [x[1] for x in s if x[1]]
        cap = self._read("power1_cap")
        return {"package_w_mean": round(sum(w) / len(w), 1), "package_w_max": round(max(w), 1), "package_w_cap": round(cap / 1e6, 1) if cap else None,
                "sclk_mhz_mean": round(sum(c) / len(c)) if c else None, "samples": len(s),
                "source": f"sysfs {self.hw}/power1_average + freq1_input read in-process at 4 Hz during the timed loop (rank 0)"}


def timed_loop(fn, steps, warmup, world):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def gemm_roofline(lib, fn, steps=2):
    """Short profiled pass AFTER the timed region: every bf16 NT GEMM launch is bracketed by HIP events on its launch stream inside
    the library (ldmae_prof_*); achieved = sum(2MNK) / sum(duration)."""
    import ctypes as C
    lib.ldmae_prof_enable(1)
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    lib.ldmae_prof_enable(0)
    ms, fl, nl = C.c_double(), C.c_double(), C.c_long()
    lib.ldmae_prof_collect(C.byref(ms), C.byref(fl), C.byref(nl))
    tf = (fl.value / 1e12) / (ms.value / 1e3) if ms.value > 0 else 0.0
    pmc, src = measured_traffic()
    return {"bound": "mfma", "kernel": "gemm_nt_lines_kernel (bf16 NT GEMM, whole-line ring: every Linear fwd + dX; gemm_nt_persist_kernel for shapes off 128-B lines)",
            "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4),
            "traffic": pmc["traffic"], "traffic_unit": "HBM bytes per launch (PMC: FETCH_SIZE x2 + WRITE_SIZE)",
            "mfma_busy": pmc["mfma_busy"], "clock_ghz": pmc["clock_ghz"],
            # the same rate against the peak AT THE CLOCK THE CHIP HELD in that kernel (the package sits at its power cap: DESIGN section 6);
            # `frac` above stays the headline, against the 2.4-GHz spec peak
            "frac_at_held_clock": round(tf / (PEAK_BF16_TFLOPS * pmc["clock_ghz"] / NOMINAL_CLOCK_GHZ), 4) if pmc["clock_ghz"] else None,
            "counters_unit": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 SIMDs x kernel cycles); clock_ghz = GRBM_GUI_ACTIVE / 8 / duration (profiled pass)",
            "traffic_source": src,
            "launches": int(nl.value), "avg_launch_ms": round(ms.value / max(1, nl.value), 4),
            "avg_launch_gflop": round(fl.value / max(1, nl.value) / 1e9, 2),
            "measured": f"{steps} profiled steps after the timed region (HIP events per launch on the launch stream)"}


def bench_dropin(args, device):
    """The DROP-IN mode, measured (verdict r04 item 4): what the reference's own train_accum.py runs when its imports resolve to this tree --
    the model built through the drop-in names with the reference's constructor call (train_accum.py:79-90), the STOCK torch.optim.AdamW
    (:121; foreach on CUDA), the reference's EMA as a deep copy updated by a per-parameter mul_ / add_ loop (:336-347), gradients delivered by
    autograd's AccumulateGrad into ordinary .grad tensors: none of the slab / fused-optimizer / direct-gradient machinery the headline uses.
    Same model, batch, precision and kernels otherwise; the difference to the headline is itemised by timing the two optimizer parts alone."""
    import copy
    from ldmae_amd import _dropin, ops
    _dropin.install()
    from models.lightningdit import LightningDiT_models        # the reference driver's import line (train_accum.py:33)
    from transport import create_transport                     # (:34)
    ops.set_gemm_launch_mode("persistent")
    torch.manual_seed(0)
    np.random.seed(0)
    model = LightningDiT_models["LightningDiT-B/1"](input_size=32, num_classes=1000, use_qknorm=True, use_swiglu=True, use_rope=True, use_rmsnorm=True,
                                                    wo_shift=False, in_channels=16, use_checkpoint=False, class_dropout_prob=0.1).to(device)
    g = torch.Generator().manual_seed(0)                         # as the headline: non-zero adaLN / final layers, so every kernel does real work
    with torch.no_grad():
        for n, p_ in model.named_parameters():
            if "adaLN_modulation" in n or n.startswith("final_layer.linear"):
                p_.copy_(torch.randn(p_.shape, generator=g, device="cpu").to(p_.device) * 0.02)
    ema = copy.deepcopy(model).to(device)
    for p_ in ema.parameters():
        p_.requires_grad_(False)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, weight_decay=0, betas=(0.9, 0.95))
    transport = create_transport("Linear", "velocity", None, None, None, use_cosine_loss=False, use_lognorm=True)
    x = torch.randn(args.batch, 16, 32, 32, device=device)
    y = torch.randint(0, 1000, (args.batch,), device=device)

    @torch.no_grad()
    def update_ema(decay=0.9999):                               # the reference's loop: one mul_ and one add_ per parameter
        ep = dict(ema.named_parameters())
        for n, p_ in model.named_parameters():
            ep[n].mul_(decay).add_(p_.data, alpha=1 - decay)

    def fwd_bwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = transport.training_losses(model, x, dict(y=y))["loss"].mean()
        loss.backward()
        return loss

    def step():
        loss = fwd_bwd()
        opt.step()
        opt.zero_grad()
        update_ema()
        return loss
    for _ in range(2):
        step()
    el, loss = timed_loop(step, args.steps, 1, 1)

    def part(fn, iters=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / iters * 1e3, 3)
    fwd_bwd()                                                   # leave gradients in place for opt.step()
    t_opt, t_ema = part(opt.step), part(update_ema)
    opt.zero_grad()
    t_fb = part(lambda: (fwd_bwd(), opt.zero_grad()))
    return {"ms_per_step": round(el / args.steps * 1e3, 3), "images_per_s": round(args.batch * args.steps / el, 2), "steps": args.steps,
            "loss": round(float(loss.detach()), 5), "optimizer": "torch.optim.AdamW (stock, foreach)", "ema": "deep copy + per-parameter mul_/add_ loop",
            "parts_ms": {"forward_backward_autograd_grads": t_fb, "adamw_step": t_opt, "update_ema_loop": t_ema},
            "note": "reference train_accum.py's own optimizer / EMA / gradient hand-off on this tree's model (drop-in mode); the headline replaces "
                    "them by one fused AdamW + EMA pass over parameter / gradient slabs and GEMM-side gradient accumulation"}


def bench_dp_config(args, device, launch_modes=("tile", "persistent")):
    """The DATA-PARALLEL program on one GPU: a world of ONE RCCL rank (init_process_group("nccl"), librccl loaded), the gradient reducer
    with its hooks, side stream and per-bucket all-reduce launches live (force_hooks), adaLN weights first in the slab, batched adaLN on --
    timed with both GEMM launch modes.  What the first 8-GPU run adds to this is the collectives' own time and their CU share."""
    from ldmae_amd import ops
    import socket
    own = not dist.is_initialized()
    if own:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=device)
    res = {}
    try:
        model, opt, reducer, transport = build(device, args.batch, force_hooks=True)
        reducer.measure_exposed = True
        torch.manual_seed(0); np.random.seed(0)
        x = torch.randn(args.batch, 16, 32, 32, device=device)
        y = torch.randint(0, 1000, (args.batch,), device=device)
        step = lambda: train_step(model, opt, reducer, transport, x, y)
        for mode in launch_modes:
            ops.set_gemm_launch_mode(mode)
            for _ in range(2):
                step()
            reducer.exposed_comm_ms()
            el, loss = timed_loop(step, args.steps, 1, 1)
            res[mode] = {"ms_per_step": round(el / args.steps * 1e3, 3), "exposed_reduce_wait_ms": round(reducer.exposed_comm_ms() / (args.steps + 1), 3)}
        res.update({"buckets": len(reducer.buckets), "adaln_params_first": opt.flat.n_front, "batched_adaln": bool(model.batched_adaln),
                    "backend": "rccl, world of 1 rank", "steps": args.steps, "loss": round(float(loss.item()), 5)})
        del model, opt, reducer
    finally:
        ops.set_gemm_launch_mode("auto")
        if own:
            dist.destroy_process_group()
    return res


def bench_dit(args, world, rank, device, lib, backend):
    from ldmae_amd import ops
    model, opt, reducer, transport = build(device, args.batch)
    # world > 1: one tile per workgroup or persistent, a per-call flag of the C ABI (LDMAE_DP_GEMM_LAUNCH; default = what --dp-config measured)
    ops.set_gemm_launch_mode(os.environ.get("LDMAE_DP_GEMM_LAUNCH", reducer.recommended_gemm_launch_mode()) if world > 1 else "persistent")
    reducer.measure_exposed = True
    seed = 0 * world + rank                      # inference.py:87 convention
    torch.manual_seed(seed)
    np.random.seed(seed)
    x = torch.randn(args.batch, 16, 32, 32, device=device)     # latents are channel-normalised -> N(0,1) (img_latent_dataset.py:86-88)
    y = torch.randint(0, 1000, (args.batch,), device=device)
    step = lambda: train_step(model, opt, reducer, transport, x, y)
    for _ in range(min(2, args.warmup)):
        step()
    reducer.exposed_comm_ms()                                  # drop warm-up samples
    with PowerSampler(enabled=rank == 0 and not args.no_power, device_index=device.index or 0) as ps:
        elapsed, loss = timed_loop(step, args.steps, max(0, args.warmup - 2), world)
    exposed = reducer.exposed_comm_ms() / max(1, args.steps + max(0, args.warmup - 2))
    if world > 1:
        tt = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final_loss = float(loss.item())
    if not np.isfinite(final_loss):
        raise RuntimeError("non-finite loss in bench")
    # SURVEY 8(d) defines the step from the H2D of the batch: a second, short loop hands every step a FRESH batch from pinned host memory
    # (16.8 MB per 256 latents, async copy on the compute stream).  Reported beside `value`, which keeps its inputs resident in HBM.
    hsteps = max(2, min(10, args.steps))
    xh = [torch.randn(args.batch, 16, 32, 32).pin_memory() for _ in range(2)]
    yh = [torch.randint(0, 1000, (args.batch,)).pin_memory() for _ in range(2)]
    cnt = [0]

    def step_h2d():
        i = cnt[0] & 1
        cnt[0] += 1
        return train_step(model, opt, reducer, transport, xh[i].to(device, non_blocking=True), yh[i].to(device, non_blocking=True))
    el_h, _ = timed_loop(step_h2d, hsteps, 1, world)
    reducer.exposed_comm_ms()
    timeline = None
    if world > 1 and reducer.overlap:
        # one more step with per-bucket events: when backward released each bucket, how long its all-reduce took, how much ran past backward
        reducer.measure_timeline = True
        step()
        torch.cuda.synchronize()
        timeline = reducer.bucket_timeline()
        reducer.measure_timeline = False
    roof = gemm_roofline(lib, step)
    if rank != 0:
        return None
    ips = args.batch * world * args.steps / elapsed
    out = {
        "metric": METRIC, "value": round(ips, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "LightningDiT-B/1 f8d16 bf16 train step (fwd+bwd+AdamW+EMA), 1024 tokens x 768, synthetic ImageNet-256 "
                               "latents 32x32x16", "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "parallelism": f"dp{world}", "loss": round(final_loss, 5)},
        "step_mfma_frac": round(FLOPS_PER_IMAGE * ips / world / (PEAK_BF16_TFLOPS * 1e12), 4),
        "roofline": roof,
        "hbm": hbm_block(args.batch),
        "h2d_inclusive": {"ms_per_step": round(el_h / hsteps * 1e3, 3), "images_per_s": round(args.batch * world * hsteps / el_h, 2), "steps": hsteps,
                          "note": "same step with a fresh pinned-host batch copied H2D (non_blocking) every step, SURVEY 8(d); rank 0 clock"},
        "power": ps.summary(),
    }
    # the whole step against the peak at the clock it held: the mean shader clock sampled in process during the timed loop when the sensor is
    # readable, else the committed PMC run's clock of the dominant kernel (labelled either way)
    held, src = None, None
    if out["power"] and out["power"].get("sclk_mhz_mean"):
        held, src = out["power"]["sclk_mhz_mean"] / 1e3, "power.sclk_mhz_mean (sysfs, this run's timed loop)"
    elif roof.get("clock_ghz"):
        held, src = roof["clock_ghz"], "roofline.clock_ghz (committed PMC run, NT GEMM)"
    out["step_mfma_frac_at_held_clock"] = round(out["step_mfma_frac"] * NOMINAL_CLOCK_GHZ / held, 4) if held else None
    out["held_clock_ghz"], out["held_clock_source"] = (round(held, 3), src) if held else (None, None)
    if world > 1:
        out["comm"] = {"backend": "rccl" if backend == "nccl" else backend, "rccl_ranks": world if backend == "nccl" else 0,
                       "grad_bytes_per_step": int(opt.flat.n_trainable) * 4, "buckets": len(reducer.buckets),
                       "exposed_comm_ms_per_step_rank0": round(exposed, 3),
                       "gemm_launch_mode": "one tile per workgroup" if ops.gemm_launch_mode() == "tile" else "persistent",
                       "batched_adaln": bool(model._use_batched_adaln()) if hasattr(model, "_use_batched_adaln") else None,
                       "bucket_timeline_rank0": timeline}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    return out


def bench_celeba(args, device):
    """The reference's second documented training configuration (README.md:108-110: run_train.sh configs/celeba_hq/lightningdit_b_vmae_f8d16_cfg.yaml):
    LightningDiT-B/1 without QK-norm, one class, no label drop -- the same train step at bs 256 bf16.  The attention front end is RoPE only,
    the forward softmax tracks its running maximum (no norm, no proven score bound), the backward epilogues apply the rotation's adjoint."""
    model, opt, reducer, transport = build(device, args.batch, celeba=True)
    torch.manual_seed(0)
    np.random.seed(0)
    x = torch.randn(args.batch, 16, 32, 32, device=device)
    y = torch.zeros(args.batch, dtype=torch.long, device=device)
    step = lambda: train_step(model, opt, reducer, transport, x, y)
    elapsed, loss = timed_loop(step, args.steps, args.warmup, 1)
    ips = args.batch * args.steps / elapsed
    if not np.isfinite(float(loss.item())):
        raise RuntimeError("non-finite loss in the celeba bench")
    return {"metric": "DiT-B/1 train step, CelebA-HQ configuration (use_qknorm false, num_classes 1), images/s on one MI355X", "value": round(ips, 2),
            "unit": "images/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3), "steps": args.steps, "warmup": args.warmup, "dtype": "bf16",
            "per_gpu_batch": args.batch, "loss": round(float(loss.item()), 5),
            "step_mfma_frac": round(FLOPS_PER_IMAGE * ips / (PEAK_BF16_TFLOPS * 1e12), 4)}


def bench_vmae(args, world, rank, device, lib):
    """BASELINE config 4: VMAE masked-token encoder (mae_for_ldmae_f8d16_prev, mask_ratio 0.75) on 256x256 random images, bf16 autocast,
    forward_encoder (patch-embed -> random_masking -> gather -> 12 blocks on the kept tokens -> LayerNorm).  Replicas only for N > 1."""
    from ldmae_amd.tokenizer import models_mae
    m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=False, no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=256).to(device).eval()
    g = torch.Generator(device=device).manual_seed(rank)
    x = torch.rand(args.batch, 3, 256, 256, device=device, generator=g) * 2 - 1
    noise = torch.rand(args.batch, 1024, device=device, generator=g)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return m.forward_encoder(x, 0.75, noise=noise)[0]
    elapsed, lat = timed_loop(step, args.steps, args.warmup, world)
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        return None
    kept = lat.shape[1]
    ips = args.batch * world * args.steps / elapsed
    flops = 3.395e9                      # 3.32 GFLOP/sample encoder @ keep 256 + 0.075 patch embed
    # Since round 3 the 12 blocks + closing LayerNorm are ONE kernel (activations never leave the CU), so the step's HBM traffic is the image
    # in (f32), the kept patches + their embedding once, and the latent tokens out: ~1.8 MB per image, 2-3 % of the HBM roofline at this rate.
    # What bounds it is the matrix / vector / LDS work of width-192, head-dim-16 blocks: priced against the dense bf16 MFMA peak on the
    # algorithmic FLOPs (the HBM figure is reported beside it).
    abytes = 3 * 256 * 256 * 4 + 256 * 192 * (2 + 4 + 4 + 4)
    return {
        "metric": "VMAE encoder kept-tokens/sec (mask_ratio 0.75, 256x256 images) on MI355X", "value": round(ips * kept, 1), "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "images_per_s": round(ips, 1),
        "config": {"workload": "VMAE mae_for_ldmae_f8d16_prev forward_encoder, mask_ratio 0.75, 256x256x3 random images (BASELINE config 4)",
                   "per_gpu_batch": args.batch, "kept_tokens_per_image": int(kept), "parallelism": f"replicas{world}"},
        "roofline": {"bound": "mfma", "kernel": "whole forward_encoder step (patch embed of the kept tokens + vmae_encoder_fwd_kernel: 12 blocks in one launch)",
                     "achieved": round(flops * ips / world / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(flops * ips / world / (PEAK_BF16_TFLOPS * 1e12), 4), "traffic": None,
                     "algorithmic_gflop_per_image": flops / 1e9, "hbm_bytes_per_image": abytes,
                     "hbm_frac_of_peak": round(abytes * ips / world / 8e12, 4)},
    }


def bench_xl_sample(args, world, rank, device, lib):
    """BASELINE config 5 kernel reuse: LightningDiT-XL/1 (depth 28, width 1152, 16 heads, head_dim 72) CFG forward in bf16, inference
    only.  One step = one forward_with_cfg on a doubled batch (what every Euler step of run_inference.sh costs; 250 of them per image)."""
    from ldmae_amd.models.lightningdit import LightningDiT_models
    n = args.batch                                               # per-GPU images per sampling batch (doubled for CFG): 256 -> CFG batch 512,
                                                                 # the reference's `per_proc_batch_size: 256` (lightningdit_b_vmae_f8d16_cfg.yaml:77)
    model = LightningDiT_models["LightningDiT-XL/1"](input_size=32, num_classes=1000, use_qknorm=True, use_swiglu=True, use_rope=True,
                                                      use_rmsnorm=True, wo_shift=False, in_channels=16, class_dropout_prob=0.1)
    gsd = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for nm, p in model.named_parameters():
            if "adaLN_modulation" in nm or nm.startswith("final_layer.linear"):
                p.copy_(torch.randn(p.shape, generator=gsd) * 0.02)
    model = model.to(device).eval()
    g = torch.Generator(device=device).manual_seed(rank)
    z = torch.randn(n, 16, 32, 32, device=device, generator=g)
    z = torch.cat([z, z], 0)
    y = torch.cat([torch.randint(0, 1000, (n,), device=device, generator=g), torch.full((n,), 1000, device=device)], 0)
    t = torch.full((2 * n,), 0.5, device=device)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return model.forward_with_cfg(z, t, y, 10.0, True, 0.10)
    elapsed, out = timed_loop(step, args.steps, args.warmup, world)
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if not bool(torch.isfinite(out).all()):
        raise RuntimeError("non-finite XL output")
    # the sampler (ldmae_amd/inference.py: sample_latents) runs the steps below the guidance-interval start on the conditional half alone
    # (forward_with_cfg applies no guidance there and the kept samples never see the other half): time that step too
    def half_step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return model.forward(z[:n], t[:n], y[:n])
    elapsed_h, _ = timed_loop(half_step, args.steps, args.warmup, world)
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    ms_h = elapsed_h / args.steps * 1e3
    from ldmae_amd.transport.integrators import shifted_grid
    grid = shifted_grid(0.0, 1.0, 250, 0.3)                   # the shipped sampling section: 250 Euler steps, timestep_shift 0.3
    unguided = float((grid[:-1] < 0.10).float().mean())       # share of the steps below cfg_interval_start 0.10
    per_image_s = 249 * (unguided * ms_h + (1 - unguided) * ms) / 1e3
    fl = 1049.04e9 * 2 * n                                    # 1049.04 GFLOP per sample forward
    return {
        "metric": "LightningDiT-XL/1 CFG sampling samples/sec (Euler 250 steps) on MI355X", "value": round(n * world / (250 * ms / 1e3), 3),
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "LightningDiT-XL/1 f8d16 bf16 forward_with_cfg (cfg 10.0, interval 0.10) on a doubled batch; samples/s "
                               "implied for 250 Euler steps (BASELINE config 5, inference-only kernel reuse)",
                   "per_gpu_images": n, "cfg_batch": 2 * n, "parallelism": f"replicas{world}"},
        "sampler_as_run": {"samples_per_s": round(n * world / per_image_s, 3), "ms_per_unguided_step": round(ms_h, 3), "unguided_step_share": round(unguided, 3),
                           "note": "sample_latents runs the steps below cfg_interval_start (0.10; shifted grid, 0.3) on the conditional half alone: same "
                                   "samples bit for bit (tests/test_gpu_drivers.py); `value` keeps the doubled batch at all 250 steps"},
        "roofline": {"bound": "mfma", "kernel": "whole CFG forward (algorithmic 1049.04 GFLOP per sample)", "achieved": round(fl / (ms / 1e3) / 1e12, 1),
                     "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / (ms / 1e3) / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": None},
    }


def bench_vmae_paths(device, batch=256):
    """The tokenizer calls the reference's drivers actually make, per 256 images (the encoder line above is BASELINE config 4 as written:
    masked forward_encoder, no gradients): `_encode` over all 1024 tokens (extract_features.py:150 -> models_mae.py:819-833), `decode_to_images`
    (1024-token decoder + uint8 / NHWC + D2H, inference.py:290-292 -> :963-973), in the f32 the reference runs them in and in bf16
    (set_precision), and one optimizer step of VMAE pre-training (engine_pretrain.py:51-76: masked encoder + 1024-token decoder, loss,
    backward, fused AdamW; bf16 autocast + loss scaler)."""
    import argparse as _ap
    from ldmae_amd import vmae_pretrain as vp
    from ldmae_amd.tokenizer import models_mae
    out = {"images": batch}
    m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=True, no_cls=True, kl_loss_weight=True, smooth_output=True, img_size=256).to(device).eval()
    g = torch.Generator(device=device).manual_seed(0)
    x = torch.rand(batch, 3, 256, 256, device=device, generator=g) * 2 - 1
    z = torch.randn(batch, 16, 32, 32, device=device, generator=g)

    def ms(fn, iters=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / iters * 1e3, 2)
    tf32_prev = torch.backends.cuda.matmul.allow_tf32
    # f32: exact-f32 MFMA kernels (torch's default, allow_tf32 = False: the parity path); f32_tf32: the same f32 calls with the switch the
    # reference's drivers set (inference.py:79, extract_features.py:2-3) -> the TF32-class fp16 family; bf16: set_precision
    for tag, prec, tf32 in (("f32", None, False), ("f32_tf32", None, True), ("bf16", torch.bfloat16, False)):
        m.set_precision(prec)
        torch.backends.cuda.matmul.allow_tf32 = tf32
        with torch.no_grad():
            out[f"encode_all_tokens_ms_{tag}"] = ms(lambda: m._encode(x))
        out[f"decode_to_images_ms_{tag}"] = ms(lambda: m.decode_to_images(z))
    torch.backends.cuda.matmul.allow_tf32 = tf32_prev
    del m
    free_gpu_memory()
    torch.manual_seed(0)
    pm = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=False, no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=256).to(device)
    opt = vp.build_optimizer(pm, 1.5e-4, 0.05)
    a = _ap.Namespace(accum_iter=1, lr=1.5e-4, min_lr=0.0, warmup_epochs=0, epochs=10, fixed_lr=True, precision="bf16", mask_ratio=0.75,
                      visible_loss_ratio=0.5, print_freq=10 ** 9)
    scaler = vp.LossScaler(enabled=True)
    loader = [(x, 0)]
    stats = None

    def step():
        nonlocal stats
        stats = vp.train_one_epoch(pm, loader, opt, 0, a, log=lambda s_: None, scaler=scaler)
    out["pretrain_step_ms_bf16"] = ms(step, iters=4)
    out["pretrain_loss"] = round(float(stats["loss"]), 5)
    a.precision = "fp16"                          # the reference's own autocast type (engine_pretrain.py:51-57): the fp16 kernel family
    out["pretrain_step_ms_fp16"] = ms(step, iters=4)
    out["pretrain_loss_fp16"] = round(float(stats["loss"]), 5)
    out["note"] = ("wall time per call with a device synchronise at both ends; the reference's drivers run these calls in f32 WITH "
                   "torch.backends.cuda.matmul.allow_tf32 = True: f32_tf32 is that configuration (fp16 operands = TF32's mantissa, f32 accumulation), "
                   "f32 the exact-f32 parity path (flag off)")
    return out


def bench_do_sample(args, device, batches=1):
    """BASELINE config 5 end to end, as run_inference.sh drives it (inference.py:264-299) at the reference's per-process batch: LightningDiT-B/1
    (the yaml's model), 250 shifted Euler steps with CFG 10 / interval 0.10 on a doubled batch of 2 x 256 -> de-normalise -> VMAE
    decode_to_images (1024 tokens) -> PNG files written by the host thread of ldmae_amd.inference.PngWriter.  Random-init weights, synthetic
    latent statistics.  Reports images/s over the whole loop and how long the GPU had nothing enqueued."""
    import shutil
    import tempfile
    import yaml
    from ldmae_amd import inference as inf
    from ldmae_amd.tokenizer import models_mae
    from ldmae_amd.train_accum import build_model
    root = os.path.dirname(os.path.abspath(__file__))
    cfg = yaml.safe_load(open(os.path.join(root, "ldmae_amd/configs/imagenet/lightningdit_b_vmae_f8d16_cfg.yaml")))
    model = build_model(cfg)
    gsd = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for nm, p in model.named_parameters():
            if "adaLN_modulation" in nm or nm.startswith("final_layer.linear"):
                p.copy_(torch.randn(p.shape, generator=gsd) * 0.02)
    model = model.to(device).eval()
    vae = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=True, no_cls=True, kl_loss_weight=True, smooth_output=True, img_size=cfg["data"]["image_size"]).to(device).eval()
    sample_fn = inf.build_sampler(cfg)
    s = cfg["sample"]
    n = s["per_proc_batch_size"] if args.batch == 256 else args.batch
    mean, std, mult = torch.zeros(1, 16, 1, 1, device=device), torch.ones(1, 16, 1, 1, device=device), cfg["data"].get("latent_multiplier", 0.18215)
    out_dir = tempfile.mkdtemp(prefix="ldmae_bench_png_")
    gen = torch.Generator(device=device).manual_seed(0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_png = [0.0]
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        writer = inf.PngWriter()
        gpu_ms = dec_ms = 0.0
        for it in range(batches):
            ev[0].record()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                lat, _ = inf.sample_latents(model, sample_fn, n, s["cfg_scale"], s.get("cfg_interval_start", 0), device, cfg["data"]["num_classes"], generator=gen)
            ev[1].record()
            with models_mae.reference_tf32():                # inference.py:79: the reference decodes in f32 with allow_tf32 on -> the TF32-class family
                imgs = vae.decode_to_images(lat * std / mult + mean)
            ev[2].record()
            writer.put(imgs, [f"{out_dir}/{it * n + i:06d}.png" for i in range(len(imgs))])
            torch.cuda.synchronize()
            gpu_ms += ev[0].elapsed_time(ev[2]); dec_ms += ev[1].elapsed_time(ev[2])
        t1 = time.perf_counter()
        writer.close()
        t2 = time.perf_counter()
        npng = len(os.listdir(out_dir))
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    wall = t2 - t0
    return {"metric": "LightningDiT-B/1 CFG sampling to PNG files, images/s on one MI355X (250 Euler steps, per-process batch of the reference)",
            "value": round(batches * n / wall, 3), "unit": "images/s", "images": batches * n, "png_files": npng, "wall_s": round(wall, 2),
            "euler_250_steps_s": round((gpu_ms - dec_ms) / 1e3 / batches, 2), "ms_per_cfg_forward": round((gpu_ms - dec_ms) / batches / (s["num_sampling_steps"] - 1), 2),
            "decode_to_images_s": round(dec_ms / 1e3 / batches, 3), "png_drain_after_last_batch_s": round(t2 - t1, 2),
            "gpu_idle_share": round(max(0.0, 1.0 - gpu_ms / 1e3 / wall), 4),
            "config": {"workload": "inference.py:264-299 loop: sample_latents (CFG batch 2 x %d, cfg 10.0, interval 0.10, timestep_shift 0.3) -> "
                                   "de-normalise -> decode_to_images -> PngWriter" % n, "batches": batches}}


def spawn_ranks(args, argv):
    """--gpus N > 1 without a launcher around us: start the N ranks (fresh processes, torch.distributed.run on 127.0.0.1) as a CHILD of
    this process -- which has not touched, and never touches, the GPU -- relay rank 0's JSON line and return the child's exit code
    (reference: run_train.sh:13-22 is a launcher; the driver's contract makes bench.py one too)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    other = [l for l in r.stdout.splitlines() if not l.startswith("{")]
    if other:
        print("\n".join(other), file=sys.stderr)
    if r.returncode == 0 and lines:
        print(lines[-1])
    return r.returncode if r.returncode != 0 else (0 if lines else 1)


def free_gpu_memory():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config: 256)")
    ap.add_argument("--workload", default="dit", choices=["dit", "vmae", "xl_sample", "do_sample", "vmae_paths", "celeba"],
                    help="dit = the headline train step (BASELINE config 2/3); vmae = config 4 encoder; xl_sample = config 5 CFG forward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="do not sample package power / clock (sysfs) during the timed loop")
    ap.add_argument("--no-extra", action="store_true", help="dit workload at N=1: skip the vmae / xl_sample / dp_config lines under extra_workloads")
    ap.add_argument("--dropin-config", action="store_true", help="N=1: time the drop-in mode (stock AdamW, the reference's EMA loop, autograd-delivered gradients) and print that line")
    ap.add_argument("--dp-config", action="store_true", help="N=1: time the data-parallel program (world-1 RCCL group, reducer hooks + side stream live) and print that line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a mislabelled number",
                  file=sys.stderr)
        sys.exit(2)
    # LDMAE_BENCH_BACKEND=gloo + LDMAE_BENCH_DEVICE=0 let two ranks share ONE GPU to rehearse the N>1 path on a 1-GPU box
    backend = os.environ.get("LDMAE_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("LDMAE_BENCH_DEVICE", local))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (this driver's only form): set before the first HIP call of the rank
        if backend == "nccl":
            # the RCCL run is refused rather than degraded: every rank needs its own GPU (a rehearsal on fewer GPUs asks for it explicitly:
            # LDMAE_BENCH_BACKEND=gloo LDMAE_BENCH_DEVICE=0), and the group that comes up must really be the nccl (= RCCL) backend
            if torch.cuda.device_count() < world or local >= torch.cuda.device_count():
                sys.exit(f"[bench] --gpus {world} over RCCL needs {world} visible GPUs (found {torch.cuda.device_count()}); "
                         "a rehearsal on one GPU is LDMAE_BENCH_BACKEND=gloo LDMAE_BENCH_DEVICE=0 and is labelled as such")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            if dist.get_backend() != "nccl":
                sys.exit(f"[bench] asked for the nccl (RCCL) backend, the process group came up as {dist.get_backend()!r}: refusing a silent fallback")
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        # one collective through the backend before anything is timed: the ranks that answer are the ranks the line may claim
        probe = torch.ones(1, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(probe)
        if int(probe.item()) != world:
            sys.exit(f"[bench] all-reduce of ones over {world} ranks returned {probe.item()}")

    from ldmae_amd import _lib
    lib = _lib.load()
    if args.dropin_config:
        if world != 1:
            sys.exit("--dropin-config runs on ONE GPU")
        out = {"metric": "DiT-B train step in drop-in mode on one MI355X (ms per step)", "dropin": bench_dropin(args, device)}
    elif args.dp_config:
        if world != 1:
            sys.exit("--dp-config is the data-parallel program on ONE GPU")
        out = {"metric": "DiT-B train step in the data-parallel configuration on one MI355X (ms per step)", "dp_config": bench_dp_config(args, device)}
    elif args.workload == "dit":
        out = bench_dit(args, world, rank, device, lib, backend)
        if world == 1 and not args.no_extra and args.batch == 256:
            # BASELINE configs 4 and 5 on the same box, AFTER the timed headline region (its model and activations are freed first), so
            # that the driver's one record carries them; each is the line `--workload vmae` / `--workload xl_sample` prints on its own
            free_gpu_memory()
            extra = argparse.Namespace(**vars(args))
            extra.steps, extra.warmup = 30, 5
            out["extra_workloads"] = {"vmae": bench_vmae(extra, world, rank, device, lib)}
            free_gpu_memory()
            extra.steps, extra.warmup = 4, 1
            out["extra_workloads"]["xl_sample"] = bench_xl_sample(extra, world, rank, device, lib)
            free_gpu_memory()
            out["extra_workloads"]["vmae_paths"] = bench_vmae_paths(device)
            free_gpu_memory()
            out["extra_workloads"]["do_sample"] = bench_do_sample(extra, device, batches=1)
            free_gpu_memory()
            extra.steps, extra.warmup = 8, 2
            try:
                out["extra_workloads"]["celeba"] = bench_celeba(extra, device)
                out["extra_workloads"]["celeba"]["headline_ms_per_step"] = out["ms_per_step"]
            except Exception as ex:
                out["extra_workloads"]["celeba"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            free_gpu_memory()
            try:
                out["extra_workloads"]["dropin"] = bench_dropin(extra, device)
                out["extra_workloads"]["dropin"]["headline_ms_per_step"] = out["ms_per_step"]
            except Exception as ex:
                out["extra_workloads"]["dropin"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            free_gpu_memory()
            try:
                out["extra_workloads"]["dp_config"] = bench_dp_config(extra, device)
                out["extra_workloads"]["dp_config"]["headline_ms_per_step"] = out["ms_per_step"]
            except Exception as ex:          # a box without a usable RCCL must not cost the headline line
                out["extra_workloads"]["dp_config"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    elif args.workload == "celeba":
        out = bench_celeba(args, device)
    elif args.workload == "vmae":
        out = bench_vmae(args, world, rank, device, lib)
    elif args.workload == "vmae_paths":
        out = bench_vmae_paths(device, args.batch)
    elif args.workload == "do_sample":
        out = bench_do_sample(args, device, batches=max(1, args.steps if args.steps < 10 else 1))
    else:
        out = bench_xl_sample(args, world, rank, device, lib)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
