#!/usr/bin/env python3
"""A few VMAE pre-training steps (engine_pretrain.py:51-76; 256 images of 256 x 256, mask ratio 0.75) for rocprofv3 --kernel-trace --stats: which
kernels carry the step, and what is left to ATen.
    python tools/pretrain_profile.py [bf16|fp16] [steps]"""
import argparse as _ap, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import vmae_pretrain as vp
from ldmae_amd.tokenizer import models_mae
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.manual_seed(0)
pm = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=False, no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=256).cuda()
opt = vp.build_optimizer(pm, 1.5e-4, 0.05)
a = _ap.Namespace(accum_iter=1, lr=1.5e-4, min_lr=0.0, warmup_epochs=0, epochs=10, fixed_lr=True, precision=prec, mask_ratio=0.75, visible_loss_ratio=0.5,
                  print_freq=10 ** 9)
scaler = vp.LossScaler(enabled=True)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand(256, 3, 256, 256, device="cuda", generator=g) * 2 - 1
for _ in range(steps):
    vp.train_one_epoch(pm, [(x, 0)], opt, 0, a, log=lambda s_: None, scaler=scaler)
torch.cuda.synchronize()
