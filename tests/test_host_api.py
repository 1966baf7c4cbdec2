"""CPU-side checks: the C-ABI library loads and exports every symbol of include/ldmae_hip.h, the module mirror has the
reference's API surface and state-dict keys, and the product path fails loudly without a GPU (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ldmae_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "ldmae_hip.h")).read()
    declared = set(re.findall(r"\b(ldmae_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(handle, name), name
    lib = _lib.load()
    assert lib.ldmae_arch() == b"gfx950" and lib.ldmae_version().startswith(b"ldmae_hip")


def test_module_api_and_state_dict_keys(golden):
    from ldmae_amd.models.lightningdit import LightningDiT_models
    g = golden("kernels")
    assert set(LightningDiT_models) == {'LightningDiT-B/1', 'LightningDiT-B/2', 'LightningDiT-L/2', 'LightningDiT-XL/1', 'LightningDiT-XL/2',
                                        'LightningDiT-1p0B/1', 'LightningDiT-1p0B/2', 'LightningDiT-1p6B/1', 'LightningDiT-1p6B/2'}
    # exactly the call of train_accum.py:79-90
    m = LightningDiT_models['LightningDiT-B/1'](input_size=32, num_classes=1000, use_qknorm=True, use_swiglu=True, use_rope=True,
                                                 use_rmsnorm=True, wo_shift=False, in_channels=16, use_checkpoint=False, class_dropout_prob=0.1)
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["b1_keys"]]
    assert sum(p.numel() for p in m.parameters()) == int(g["b1_nparams"])
    assert m.in_channels == 16 and m.x_embedder.patch_size[0] == 1 and m.x_embedder.num_patches == 1024
    assert not m.pos_embed.requires_grad
    # reference init: adaLN and final layer zero, norms one (lightningdit.py:364-374)
    assert float(m.blocks[3].adaLN_modulation[1].weight.abs().sum()) == 0 and float(m.final_layer.linear.weight.abs().sum()) == 0
    assert hasattr(m, "forward_with_cfg")


def test_every_block_flag_constructs_with_the_references_keys(golden):
    """Every LightningDiTBlock flag combination the reference's constructor takes builds a model whose state dict has the reference's keys
    (tests/golden/dit_flags.npz: key sets of the reference's own modules) -- LayerNorm blocks have no norm parameters, nn.LayerNorm QK-norm has
    biases, the timm Mlp has fc1 / fc2, no QK-norm has no q_norm / k_norm entries (nn.Identity, lightningdit.py:60-61), wo_shift a 4x adaLN.
    The reference's CelebA-HQ kwargs (train_accum.py:79-90 on configs/celeba_hq/...yaml) at B/1: a one-row label table, 12 x 2 x 64 fewer weights."""
    from ldmae_amd.models.lightningdit import LightningDiT, LightningDiT_models
    from weights import DIT_FLAG_VARIANTS
    g = golden("dit_flags")
    base = dict(use_qknorm=True, use_swiglu=True, use_rope=True, use_rmsnorm=True, wo_shift=False, num_classes=10)
    for tag, over in DIT_FLAG_VARIANTS.items():
        t = LightningDiT(input_size=8, patch_size=1, in_channels=16, hidden_size=192, depth=2, num_heads=3, class_dropout_prob=0.5, **{**base, **over})
        assert sorted(t.state_dict().keys()) == [str(k) for k in g[f"df_{tag}_keys"]], tag
    m = LightningDiT_models['LightningDiT-B/1'](input_size=32, num_classes=1, use_qknorm=False, use_swiglu=True, use_rope=True,
                                                 use_rmsnorm=True, wo_shift=False, in_channels=16, use_checkpoint=False, class_dropout_prob=0)
    keys = set(m.state_dict().keys())
    assert not any("q_norm" in k or "k_norm" in k for k in keys) and m.y_embedder.embedding_table.weight.shape == (1, 768)
    assert keys == {str(k) for k in golden("kernels")["b1_keys"] if "q_norm" not in str(k) and "k_norm" not in str(k)}
    assert sum(p.numel() for p in m.parameters()) == 131122960 - 12 * 2 * 64 - 1000 * 768


def test_no_cpu_fallback():
    from ldmae_amd.models.lightningdit import LightningDiT
    m = LightningDiT(input_size=8, patch_size=1, in_channels=16, hidden_size=192, depth=1, num_heads=3, num_classes=10,
                     use_qknorm=True, use_swiglu=True, use_rope=True, use_rmsnorm=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(2, 16, 8, 8), torch.rand(2), torch.tensor([1, 2]))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ldmae_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dirpath, f)


def test_transport_draws_match_reference_order(golden):
    from ldmae_amd.transport import Sampler, create_transport
    g = golden("dit_tiny")
    tr = create_transport("Linear", "velocity", None, None, None, use_cosine_loss=False, use_lognorm=True)
    assert tr.train_eps == 0 and tr.sample_eps == 0
    torch.manual_seed(5)
    np.random.seed(5)
    t, x0, _ = tr.sample(torch.from_numpy(g["tl_x1"]))
    np.testing.assert_array_equal(x0.numpy(), g["tl_x0"])
    np.testing.assert_array_equal(t.numpy(), g["tl_t"])
    fn = Sampler(tr).sample_ode(sampling_method="euler", num_steps=250, atol=1e-6, rtol=1e-3, reverse=False, timestep_shift=0.3)
    np.testing.assert_allclose(fn.__self__.t.numpy(), golden("kernels")["euler_grid"], atol=1e-7)
    # a linear "model" integrates exactly: dx/dt = 1 -> x(1) = x(0) + 1
    z = torch.zeros(2, 3)
    assert torch.allclose(fn(z, lambda x, t: torch.ones_like(x))[-1], torch.ones(2, 3), atol=1e-6)


def test_flat_params_views_and_buckets():
    from ldmae_amd.distributed import GradBucketReducer
    from ldmae_amd.optim import FlatParams
    net = torch.nn.Sequential(torch.nn.Linear(10, 7), torch.nn.Linear(7, 3))
    net[1].bias.requires_grad_(False)
    ref = {k: v.clone() for k, v in net.state_dict().items()}
    flat = FlatParams(net)
    for k, v in net.state_dict().items():
        assert torch.equal(v, ref[k])
    assert flat.n_trainable % 64 == 0 and flat.total > flat.n_trainable
    net(torch.randn(4, 10)).sum().backward()
    assert net[0].weight.grad.data_ptr() == flat.grads.data_ptr()        # grads accumulate straight into the slab
    assert float(flat.grads.abs().sum()) > 0
    red = GradBucketReducer(flat, bucket_bytes=128)
    covered = sorted((lo, hi) for lo, hi, _ in red.buckets)
    assert covered[0][0] == 0 and covered[-1][1] == flat.n_trainable
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert red.finish() == 1.0


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    """`--gpus 8` under a launcher that started 1 rank must fail, not print a 1-rank number labelled n_gpus 8 (round-2 verdict)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()


def test_out_of_scope_transport_configurations_raise_by_design():
    """SURVEY 2.1 #7 marks the VP / GVP plans, noise / score prediction, loss weighting and the SDE sampler OUT of the hot path; the
    reference's create_transport (transport/__init__.py:3-72) accepts them, this one must say so instead of training something else."""
    from ldmae_amd.transport import Sampler, create_transport
    for kw in (dict(path_type="VP"), dict(path_type="GVP"), dict(prediction="noise"), dict(prediction="score"), dict(loss_weight="velocity"),
               dict(loss_weight="likelihood")):
        with pytest.raises(NotImplementedError, match="hot path"):
            create_transport(**kw)
    t = create_transport("Linear", "velocity", None, None, None)
    with pytest.raises(NotImplementedError):
        Sampler(t).sample_sde()


def test_reference_method_names_exist():
    """Names a caller of the reference finds on the two model classes (LDMAE/tokenizer/models_mae.py, LDMAE/models/lightningdit.py) -- beyond forward / state dict."""
    from ldmae_amd.models.lightningdit import LabelEmbedder, LightningDiT
    from ldmae_amd.tokenizer.models_mae import EncoderOutput, MaskedAutoencoderViT
    for n in ("patchify", "unpatchify", "random_masking", "forward_encoder", "forward_decoder", "forward_loss", "forward_vanilla", "forward", "_encode", "encode", "decode",
              "ldmae_encoding", "ldmae_decoding", "reconstruct", "encode_images", "decode_to_images", "img_transform"):
        assert callable(getattr(MaskedAutoencoderViT, n)), n
    for n in ("forward", "forward_with_cfg", "unpatchify", "initialize_weights"):
        assert callable(getattr(LightningDiT, n)), n
    assert callable(LabelEmbedder.token_drop) and callable(EncoderOutput.mode)
    le = LabelEmbedder(10, 8, 0.1)
    out = le.token_drop(torch.tensor([1, 2, 3]), force_drop_ids=torch.tensor([1, 0, 1]))
    assert out.tolist() == [10, 2, 10]
