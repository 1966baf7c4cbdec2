#!/usr/bin/env python3
"""Time csrc/vmae_fused.hip alone (the one-kernel VMAE encoder, batch 256 = one workgroup per CU) for whatever library LDMAE_HIP_LIB names
(ablation builds: -DVF_DBG=1 no GELU, 2 no softmax / PV, 4 no MLP).   python tools/bench_vmae_fused.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import ops
from ldmae_amd.tokenizer import fused_encoder, models_mae

m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=False, no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=256).cuda().eval()
x = torch.randn(256, 256, 192, device="cuda")
blob = fused_encoder.encoder_blob(m)
fn = lambda: ops.vmae_encoder_fwd(x, blob, 12, 192, 12, 768, 1e-6)
for _ in range(3): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): fn()
b.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get("LDMAE_HIP_LIB", "libldmae_hip.so")), f"{a.elapsed_time(b) / 20:.3f} ms per encoder pass (12 blocks, 256 images)")
