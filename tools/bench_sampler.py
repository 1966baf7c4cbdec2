#!/usr/bin/env python3
"""Wall time of the Euler / CFG sampling loop (inference.sample_latents) on LightningDiT-B/1, 64 images x 20 steps: host overhead check
(32.8 ms per step = the forward's GPU time).    python tools/bench_sampler.py"""
import os, sys, time, yaml, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ldmae_amd.inference as inf
import ldmae_amd.train_accum as t
cfg = yaml.safe_load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "ldmae_amd/configs/imagenet/lightningdit_b_vmae_f8d16_cfg.yaml")))
cfg["sample"]["num_sampling_steps"] = 20
m = t.build_model(cfg).cuda().eval()
fn = inf.build_sampler(cfg)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lat, _ = inf.sample_latents(m, fn, 64, cfg["sample"]["cfg_scale"], cfg["sample"].get("cfg_interval_start", 0), torch.device("cuda"), 1000)
    torch.cuda.synchronize(); t1 = time.time()
    print(f"sample_latents 64 images x 20 Euler steps (CFG batch 128): {1e3 * (t1 - t0):.1f} ms = {1e3 * (t1 - t0) / 20:.1f} ms per step")
