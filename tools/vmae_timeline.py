#!/usr/bin/env python3
"""In-kernel timeline of the fused VMAE encoder's MLP steps (shader-clock stamps of waves 0 and 4 of two workgroups: ring wait + barrier,
fc1 + GELU, bias / convert, fc2).  Needs a TIMING build of csrc/vmae_fused.hip (-DVF_TL=1: the stamps overwrite the head of the output and
the extra registers spill -- never the product library):
    hipcc ... -DVF_TL=1 -c vmae_fused.hip -o build/vf_tl.o && hipcc -shared -o ../libldmae_hip_tl.so build/{core,gemm,elementwise,attention,vmae}.o build/vf_tl.o
    LDMAE_HIP_LIB=.../libldmae_hip_tl.so python tools/vmae_timeline.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ldmae_amd import ops
from ldmae_amd.tokenizer import fused_encoder, models_mae
m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=False, no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=256).cuda().eval()
x = torch.randn(256, 256, 192, device="cuda")
blob = fused_encoder.encoder_blob(m)
for _ in range(3): out = ops.vmae_encoder_fwd(x, blob, 12, 192, 12, 768, 1e-6)
torch.cuda.synchronize()
o = out.view(256, 8, 32, 192)
for img in (0, 100):
    for w in (0, 4):
        t = o[img, w, 0, :40].contiguous().view(torch.int64).cpu().tolist()
        for c in range(4):
            s = t[c * 5:(c + 1) * 5]
            print(f"img {img} wave {w} chunk {c+4}: slot-wait {s[1]-s[0]:5d}  fc1+gelu {s[2]-s[1]:5d}  bias/cvt {s[3]-s[2]:5d}  fc2 {s[4]-s[3]:5d}  total {s[4]-s[0]:5d}" + (f"  next-start +{t[(c+1)*5]-s[4]}" if c < 3 else ""))
