#!/usr/bin/env python3
"""Launch a few GEMMs of one shape (for rocprofv3 --pmc runs).  python3 tools/one_gemm.py kind N K [variant]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LDMAE_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ldmae_amd", "libldmae_hip_diag.so"))   # A/B knobs live in the diagnostic build only (make -C ldmae_amd/csrc diag)
from ldmae_amd import _lib, ops
kind, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
v = int(sys.argv[4]) if len(sys.argv) > 4 else 0
M = 262144
if kind in ("nt", "tn"):
    _lib.load().ldmae_tune(8 if kind == "nt" else 4, v)      # nt: 2 = one tile per workgroup; tn: 3 / 4 = 16-wave variants
g = torch.Generator(device="cuda").manual_seed(0)
if kind == "nt":
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    for _ in range(3):
        ops.gemm_nt(a, w, None)
elif kind == "tn":
    a = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    b = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    for _ in range(3):
        ops.gemm_tn(a, b)
elif kind in ("attnpv", "attnpvb"):      # the DiT block's forward attention (hd 64, QK-norm + RoPE): tracked maximum / static shift from the norm-weight bound
    from ldmae_amd.models.pos_embed import VisionRotaryEmbeddingFast
    B, H, NN, hd = 256, 12, 1024, 64
    qkv = torch.randn(B, NN, 3, H, hd, device="cuda", generator=g).to(torch.bfloat16)
    wq, wk = torch.ones(hd, device="cuda"), torch.ones(hd, device="cuda")
    rope = VisionRotaryEmbeddingFast(dim=hd // 2, pt_seq_len=32).cuda()
    q, k, _ = ops.qknorm_rope_fwd(qkv, wq, wk, rope.freqs_cos, rope.freqs_sin, B, NN, H, hd, copy_v=False)
    bound = ops.qk_score_bound(wq, wk, hd, hd ** -0.5) if kind == "attnpvb" else None
    for _ in range(3):
        ops.attention_fwd_pv(q, k, qkv, hd ** -0.5, bound=bound)
elif kind == "attn16":        # VMAE attention forward at 1024 tokens: packed qkv, 12 heads of 16 (the decoder / _encode shape)
    B, H, NN, hd = 256, 12, 1024, 16
    qkv = torch.randn(B * NN, 3 * H * hd, device="cuda", generator=g).to(torch.bfloat16)
    for _ in range(3):
        ops.attention_fwd_qkv(qkv, B, NN, H, hd, hd ** -0.5)
elif kind == "vmaet":         # the tiled VMAE encoder at 1024 tokens, 256 images (MODE 1 once, then MODE 3 + flash attention per block, MODE 2 last)
    from ldmae_amd.tokenizer import fused_encoder, models_mae
    m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=False, no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=256).cuda().eval()
    x = torch.randn(256, 1024, 192, device="cuda", generator=g)
    blob = fused_encoder.encoder_blob(m)
    for _ in range(3):
        ops.vmae_encoder_fwd_tiled(x, blob, 12, 192, 12, 768, 1e-6)
elif kind == "vmae":          # the one-kernel VMAE encoder, 256 images (one workgroup per CU)
    from ldmae_amd.tokenizer import fused_encoder, models_mae
    m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=False, no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=256).cuda().eval()
    x = torch.randn(256, 256, 192, device="cuda", generator=g)
    blob = fused_encoder.encoder_blob(m)
    for _ in range(3):
        ops.vmae_encoder_fwd(x, blob, 12, 192, 12, 768, 1e-6)
else:
    B, H, NN, hd = 256, 12, 1024, 64
    q, k, vv = (torch.randn(B, H, NN, hd, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))
    do = torch.randn(B, NN, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    for _ in range(3):
        o, lse = ops.attention_fwd(q, k, vv, 0.125)
        ops.attention_bwd(q, k, vv, o, do, lse, 0.125)
torch.cuda.synchronize()
