"""The drop-in seam on the GPU: the model is built through the reference's import names and constructor call
(/root/reference/LDMAE/train_accum.py:33-34, 79-90, 106-114) in a fresh interpreter with only <repo>/ldmae_amd on PYTHONPATH, and
the two module-level callables the reference exposes besides the block -- `feat_rope(t)` (models/pos_embed.py:135) and
`block.attn(x, rope)` (models/lightningdit.py:66-91) -- match the CPU oracle."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import rel_err
from oracle import dit as odit
from weights import det_randn, det_weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import yaml, torch, numpy as np
from models.lightningdit import LightningDiT_models                    # train_accum.py:33
from transport import create_transport, Sampler                        # train_accum.py:34
train_config = yaml.safe_load(open(CFG))
latent_size = train_config['data']['image_size'] // train_config['vae']['downsample_ratio']
model = LightningDiT_models[train_config['model']['model_type']](      # train_accum.py:79-90, verbatim
    input_size=latent_size,
    num_classes=train_config['data']['num_classes'],
    use_qknorm=train_config['model']['use_qknorm'],
    use_swiglu=train_config['model']['use_swiglu'] if 'use_swiglu' in train_config['model'] else False,
    use_rope=train_config['model']['use_rope'] if 'use_rope' in train_config['model'] else False,
    use_rmsnorm=train_config['model']['use_rmsnorm'] if 'use_rmsnorm' in train_config['model'] else False,
    wo_shift=train_config['model']['wo_shift'] if 'wo_shift' in train_config['model'] else False,
    in_channels=train_config['model']['in_chans'] if 'in_chans' in train_config['model'] else 4,
    use_checkpoint=train_config['model']['use_checkpoint'] if 'use_checkpoint' in train_config['model'] else False,
    class_dropout_prob=0 if train_config['data']['num_classes'] == 1 else 0.1,
)
transport = create_transport(                                           # train_accum.py:106-114
    train_config['transport']['path_type'], train_config['transport']['prediction'], train_config['transport']['loss_weight'],
    train_config['transport']['train_eps'], train_config['transport']['sample_eps'],
    use_cosine_loss=train_config['transport']['use_cosine_loss'] if 'use_cosine_loss' in train_config['transport'] else False,
    use_lognorm=train_config['transport']['use_lognorm'] if 'use_lognorm' in train_config['transport'] else False)
celeba = train_config['data']['num_classes'] == 1                       # configs/celeba_hq: no QK-norm weights (12 x 2 x 64), a one-row label table
assert sum(p.numel() for p in model.parameters()) == (131122960 - 1536 - 1000 * 768 if celeba else 131122960)   # SURVEY 8c: 131.12296 M for B/1
assert model.in_channels == 16 and model.x_embedder.patch_size[0] == 1 and model.x_embedder.num_patches == 1024
model = model.to("cuda").train()
opt = torch.optim.AdamW(model.parameters(), lr=2e-4, weight_decay=0, betas=(0.9, 0.95))   # train_accum.py:121: the STOCK optimizer
torch.manual_seed(0); np.random.seed(0)
x = torch.randn(2, 16, 32, 32, device="cuda"); y = torch.randint(0, train_config['data']['num_classes'], (2,), device="cuda")
with torch.autocast("cuda", dtype=torch.bfloat16):                      # what accelerate --mixed_precision bf16 sets up
    loss = transport.training_losses(model, x, dict(y=y))["loss"].mean()
loss.backward()
g = model.final_layer.linear.weight.grad
assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
opt.step()
import sys, ldmae_amd._lib as L
assert L._lib is not None, "the HIP library was not loaded"
assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules)
print("LOSS %.6f" % float(loss))
'''


@pytest.mark.parametrize("dataset", ["imagenet", "celeba_hq"])      # README.md:104-110: the two documented run_train.sh invocations
def test_reference_constructor_call_through_dropin_names(tmp_path, dataset):
    cfg = os.path.join(ROOT, "ldmae_amd", "configs", dataset, "lightningdit_b_vmae_f8d16_cfg.yaml")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = os.path.join(ROOT, "ldmae_amd")
    r = subprocess.run([sys.executable, "-c", f"CFG = {cfg!r}\n" + DRIVER], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    loss = float(r.stdout.strip().split("LOSS")[-1])
    assert 1.5 < loss < 2.6          # zero-initialised final layer: pred = 0, loss = E|x1 - x0|^2 ~ 2


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
def test_feat_rope_callable_matches_oracle(dtype, tol):
    from ldmae_amd.models.pos_embed import VisionRotaryEmbeddingFast
    rope = VisionRotaryEmbeddingFast(dim=32, pt_seq_len=8).cuda()            # head_dim 64, N = 64
    cos, sin = odit.rope_tables(64, 8)
    t = det_randn("rope_t", (2, 3, 64, 64), 5).to(dtype)
    tg = t.cuda().requires_grad_(True)
    out = rope(tg)
    ref = odit.apply_rope(t.float(), cos, sin)
    assert out.dtype == dtype and rel_err(out.float().cpu(), ref) < tol
    w = det_randn("rope_w", (2, 3, 64, 64), 6)
    (out.float() * w.cuda()).sum().backward()
    tr = t.float().requires_grad_(True)
    (odit.apply_rope(tr, cos, sin) * w).sum().backward()
    assert rel_err(tg.grad.float().cpu(), tr.grad) < tol


def test_attention_module_callable_matches_oracle():
    from ldmae_amd.models.lightningdit import LightningDiTBlock
    from ldmae_amd.models.pos_embed import VisionRotaryEmbeddingFast
    cfg = odit.DiTConfig(input_size=8, patch_size=1, in_channels=16, hidden_size=192, depth=1, num_heads=3, num_classes=10)
    sd = det_weights(odit.param_shapes(cfg), 3)
    blk = LightningDiTBlock(192, 3, use_qknorm=True, use_swiglu=True, use_rmsnorm=True).cuda()
    blk.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items() if k.startswith("blocks.0.")})
    rope = VisionRotaryEmbeddingFast(dim=32, pt_seq_len=8).cuda()
    cos, sin = odit.rope_tables(64, 8)
    x = det_randn("attn_x", (2, 64, 192), 9)
    xg = x.cuda().requires_grad_(True)
    out = blk.attn(xg, rope)                                                  # the reference's call form (lightningdit.py:248)
    xr = x.clone().requires_grad_(True)
    osd = {k: v.clone().requires_grad_(k.startswith("blocks.0.attn.")) for k, v in sd.items()}
    ref = odit.attention(osd, "blocks.0.attn.", xr, cfg, cos, sin)
    assert rel_err(out.cpu(), ref.detach()) < 1e-4
    w = det_randn("attn_w", (2, 64, 192), 10)
    (out * w.cuda()).sum().backward()
    (ref * w).sum().backward()
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-4
    for name in ("qkv.weight", "qkv.bias", "q_norm.weight", "k_norm.weight", "proj.weight", "proj.bias"):
        got = dict(blk.attn.named_parameters())[name].grad
        assert rel_err(got.cpu(), osd["blocks.0.attn." + name].grad) < 2e-4, name
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ob = blk.attn(x.cuda(), rope)
    assert ob.dtype == torch.bfloat16 and rel_err(ob.float().cpu(), ref.detach()) < 3e-2
