"""Oracle (test infrastructure): functional fp32 restatement of LightningDiT.

Weights are a flat ``dict[str, Tensor]`` with the reference's state-dict keys
(SURVEY.md §8a).  All functions are pure; gradients come from torch autograd on
CPU.  Citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class DiTConfig:
    """Constructor arguments of LightningDiT (LDMAE/models/lightningdit.py:279-297)."""
    input_size: int = 32
    patch_size: int = 1
    in_channels: int = 16
    hidden_size: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    class_dropout_prob: float = 0.1
    num_classes: int = 1000
    learn_sigma: bool = False
    # block flags (lightningdit.py:292-296; defaults = the shipped imagenet YAML, configs/imagenet/lightningdit_b_vmae_f8d16_cfg.yaml:26-33;
    # configs/celeba_hq/...yaml:30 sets use_qknorm false)
    use_qknorm: bool = True
    use_swiglu: bool = True
    use_rope: bool = True
    use_rmsnorm: bool = True
    wo_shift: bool = False

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads

    @property
    def grid(self):
        return self.input_size // self.patch_size

    @property
    def num_tokens(self):
        return self.grid * self.grid

    @property
    def mlp_hidden(self):
        # LDMAE/models/lightningdit.py:213,217: int(2/3 * int(hidden*mlp_ratio)); :219-224 (timm Mlp): int(hidden*mlp_ratio)
        full = int(self.hidden_size * self.mlp_ratio)
        return int(2 / 3 * full) if self.use_swiglu else full

    @property
    def out_channels(self):
        return self.in_channels * (2 if self.learn_sigma else 1)


DIT_B_1 = dict(depth=12, hidden_size=768, patch_size=1, num_heads=12)     # lightningdit.py:507-508
DIT_XL_1 = dict(depth=28, hidden_size=1152, patch_size=1, num_heads=16)   # lightningdit.py:498-499


# --------------------------------------------------------------------------- tables
def sincos_pos_embed_2d(embed_dim: int, grid_size: int) -> np.ndarray:
    """2-D sin-cos table, float64 omega (lightningdit.py:444-491).

    Half of the channels encode the *w* coordinate first ("here w goes first",
    :451), each half is [sin | cos].
    """
    coords = np.arange(grid_size, dtype=np.float32)
    gw, gh = np.meshgrid(coords, coords)          # gw[i, j] = j, gh[i, j] = i

    def one_axis(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float64) / (dim / 2.0)
        omega = 1.0 / 10000 ** omega
        ang = pos.reshape(-1)[:, None] * omega[None, :]
        return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)

    return np.concatenate([one_axis(embed_dim // 2, gw), one_axis(embed_dim // 2, gh)], axis=1)


def rope_tables(head_dim: int, grid: int, theta: float = 10000.0):
    """freqs_cos / freqs_sin [grid*grid, head_dim] of VisionRotaryEmbeddingFast
    (LDMAE/models/pos_embed.py:96-133) as built at lightningdit.py:317-323
    (dim = head_dim // 2, pt_seq_len = ft_seq_len = grid)."""
    dim = head_dim // 2
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))   # [dim/2]
    t = torch.arange(grid) / grid * grid                                            # :121
    f = t[:, None] * freqs[None, :]                                                 # [grid, dim/2]
    f = f.repeat_interleave(2, dim=-1)                                              # '(n r)', r=2  -> [grid, dim]
    fh = f[:, None, :].expand(grid, grid, dim)                                      # row index
    fw = f[None, :, :].expand(grid, grid, dim)                                      # col index
    full = torch.cat([fh, fw], dim=-1).reshape(grid * grid, 2 * dim)
    return full.cos(), full.sin()


# --------------------------------------------------------------------------- pieces
def rmsnorm(x, weight, eps: float = 1e-6):
    """LDMAE/models/rmsnorm.py:51-77."""
    xf = x.float()
    y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x)
    return y * weight


def modulate(x, shift, scale):
    """lightningdit.py:26-30 (shift None: wo_shift)."""
    if shift is None:
        return x * (1 + scale.unsqueeze(1))
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def block_norm(sd, key, x, cfg):
    """norm1 / norm2 / norm_final: RMSNorm with weight (lightningdit.py:203-204,259), or -- use_rmsnorm=False -- LayerNorm without affine,
    eps 1e-6 (:200-201,257)."""
    if cfg.use_rmsnorm:
        return rmsnorm(x, sd[key + ".weight"])
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)


def qk_norm(sd, key, x, cfg):
    """q_norm / k_norm (lightningdit.py:56-61): Identity without use_qknorm; RMSNorm(head_dim), or nn.LayerNorm(head_dim) (affine, eps 1e-5)
    when use_rmsnorm is off."""
    if not cfg.use_qknorm:
        return x
    if cfg.use_rmsnorm:
        return rmsnorm(x, sd[key + ".weight"])
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], 1e-5)


def rotate_pairs(x):
    """pos_embed.py:38-42: (x0, x1) -> (-x1, x0) on interleaved pairs."""
    x = x.reshape(*x.shape[:-1], -1, 2)
    return torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(*x.shape[:-2], -1)


def apply_rope(t, cos, sin):
    """pos_embed.py:135."""
    return t * cos + rotate_pairs(t) * sin


def timestep_embedding(t, dim: int = 256, max_period: int = 10000):
    """lightningdit.py:109-131 (t is NOT scaled by 1000)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def t_embedder(sd, t):
    """lightningdit.py:133-137."""
    h = F.linear(timestep_embedding(t), sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])
    return F.linear(F.silu(h), sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])


def y_embedder(sd, y, cfg: DiTConfig, train: bool, drop_ids=None):
    """lightningdit.py:152-169.  ``drop_ids`` (bool[B]) replaces the in-forward
    ``torch.rand(B) < p`` draw when given; when None and ``train`` the draw is
    made here from the global torch RNG exactly like the reference."""
    if drop_ids is None and train and cfg.class_dropout_prob > 0:
        drop_ids = torch.rand(y.shape[0]) < cfg.class_dropout_prob
    if drop_ids is not None:
        y = torch.where(drop_ids, torch.full_like(y, cfg.num_classes), y)
    return sd["y_embedder.embedding_table.weight"][y]


def patch_embed(sd, x, cfg: DiTConfig):
    """timm PatchEmbed as used at lightningdit.py:309,402: Conv2d(k=s=p) ->
    flatten(2).transpose(1,2), then + pos_embed."""
    h = F.conv2d(x, sd["x_embedder.proj.weight"], sd["x_embedder.proj.bias"], stride=cfg.patch_size)
    return h.flatten(2).transpose(1, 2) + sd["pos_embed"]


def attention(sd, pre, x, cfg: DiTConfig, cos, sin, taps=None):
    """lightningdit.py:66-91 (qk-norm [optional] + rope [optional] + SDPA, scale 1/sqrt(hd))."""
    B, N, C = x.shape
    H, hd = cfg.num_heads, cfg.head_dim
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q, k = qk_norm(sd, pre + "q_norm", q, cfg), qk_norm(sd, pre + "k_norm", k, cfg)
    if cfg.use_rope:                                                                 # :71-73
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    s = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    o = s.softmax(dim=-1) @ v
    o = o.transpose(1, 2).reshape(B, N, C)
    if taps is not None:
        taps[pre + "q"], taps[pre + "k"], taps[pre + "v"], taps[pre + "o"] = q, k, v, o
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def swiglu(sd, pre, x):
    """LDMAE/models/swiglu_ffn.py:31-36."""
    x12 = F.linear(x, sd[pre + "w12.weight"], sd[pre + "w12.bias"])
    x1, x2 = x12.chunk(2, dim=-1)
    return F.linear(F.silu(x1) * x2, sd[pre + "w3.weight"], sd[pre + "w3.bias"])


def gelu_mlp(sd, pre, x):
    """timm Mlp as built at lightningdit.py:219-224: fc1 -> GELU(approximate="tanh") -> fc2 (drop = 0)."""
    h = F.gelu(F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"]), approximate="tanh")
    return F.linear(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def block(sd, i, x, c, cfg: DiTConfig, cos, sin, taps=None):
    """lightningdit.py:239-250."""
    p = f"blocks.{i}."
    mod = F.linear(F.silu(c), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"])
    if cfg.wo_shift:                                                                 # :241-244
        scale_msa, gate_msa, scale_mlp, gate_mlp = mod.chunk(4, dim=1)
        shift_msa = shift_mlp = None
    else:
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod.chunk(6, dim=1)
    xm = modulate(block_norm(sd, p + "norm1", x, cfg), shift_msa, scale_msa)
    x = x + gate_msa.unsqueeze(1) * attention(sd, p + "attn.", xm, cfg, cos, sin, taps)
    xm2 = modulate(block_norm(sd, p + "norm2", x, cfg), shift_mlp, scale_mlp)
    x = x + gate_mlp.unsqueeze(1) * (swiglu if cfg.use_swiglu else gelu_mlp)(sd, p + "mlp.", xm2)
    if taps is not None:
        taps[p + "xm1"], taps[p + "xm2"], taps[p + "out"] = xm, xm2, x
    return x


def final_layer(sd, x, c, cfg: DiTConfig = None):
    """lightningdit.py:267-272."""
    mod = F.linear(F.silu(c), sd["final_layer.adaLN_modulation.1.weight"], sd["final_layer.adaLN_modulation.1.bias"])
    shift, scale = mod.chunk(2, dim=1)
    x = modulate(block_norm(sd, "final_layer.norm_final", x, cfg or DiTConfig()), shift, scale)
    return F.linear(x, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"])


def unpatchify(x, cfg: DiTConfig):
    """lightningdit.py:376-389."""
    c, p = cfg.out_channels, cfg.patch_size
    h = w = int(x.shape[1] ** 0.5)
    x = x.reshape(x.shape[0], h, w, p, p, c)
    x = torch.einsum("nhwpqc->nchpwq", x)
    return x.reshape(x.shape[0], c, h * p, w * p)


def dit_forward(sd, x, t, y, cfg: DiTConfig, train: bool = True, drop_ids=None, taps=None):
    """LightningDiT.forward, lightningdit.py:391-418."""
    cos, sin = (sd["feat_rope.freqs_cos"], sd["feat_rope.freqs_sin"]) if cfg.use_rope else (None, None)     # :317-325
    h = patch_embed(sd, x, cfg)
    c = t_embedder(sd, t) + y_embedder(sd, y, cfg, train, drop_ids)
    if taps is not None:
        taps["x_embed"], taps["c"] = h, c
    for i in range(cfg.depth):
        h = block(sd, i, h, c, cfg, cos, sin, taps)
    out = unpatchify(final_layer(sd, h, c, cfg), cfg)
    if cfg.learn_sigma:
        out, _ = out.chunk(2, dim=1)
    return out


def dit_forward_with_cfg(sd, x, t, y, cfg: DiTConfig, cfg_scale, cfg_interval=None, cfg_interval_start=None):
    """lightningdit.py:420-442: CFG on channels [:3] only; interval gate on t[0]."""
    half = x[: len(x) // 2]
    out = dit_forward(sd, torch.cat([half, half], 0), t, y, cfg, train=False)
    eps, rest = out[:, :3], out[:, 3:]
    cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
    half_eps = uncond + cfg_scale * (cond - uncond)
    if cfg_interval is True and t[0] < cfg_interval_start:
        half_eps = cond
    return torch.cat([torch.cat([half_eps, half_eps], 0), rest], dim=1)


# --------------------------------------------------------------------------- weights
def param_shapes(cfg: DiTConfig) -> dict:
    """State-dict key -> shape of every *parameter* (SURVEY.md §8a key list)."""
    D, Hs, p = cfg.hidden_size, cfg.mlp_hidden, cfg.patch_size
    s = {
        "pos_embed": (1, cfg.num_tokens, D),
        "x_embedder.proj.weight": (D, cfg.in_channels, p, p),
        "x_embedder.proj.bias": (D,),
        "t_embedder.mlp.0.weight": (D, 256), "t_embedder.mlp.0.bias": (D,),
        "t_embedder.mlp.2.weight": (D, D), "t_embedder.mlp.2.bias": (D,),
        "y_embedder.embedding_table.weight": (cfg.num_classes + (1 if cfg.class_dropout_prob > 0 else 0), D),
    }
    nmod = 4 if cfg.wo_shift else 6
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s.update({
            b + "attn.qkv.weight": (3 * D, D), b + "attn.qkv.bias": (3 * D,),
            b + "attn.proj.weight": (D, D), b + "attn.proj.bias": (D,),
            b + "adaLN_modulation.1.weight": (nmod * D, D), b + "adaLN_modulation.1.bias": (nmod * D,),
        })
        if cfg.use_rmsnorm:                   # LayerNorm(elementwise_affine=False) has no parameters (:200-201)
            s.update({b + "norm1.weight": (D,), b + "norm2.weight": (D,)})
        if cfg.use_qknorm:                    # nn.Identity otherwise (:60-61); nn.LayerNorm carries a bias
            s.update({b + "attn.q_norm.weight": (cfg.head_dim,), b + "attn.k_norm.weight": (cfg.head_dim,)})
            if not cfg.use_rmsnorm:
                s.update({b + "attn.q_norm.bias": (cfg.head_dim,), b + "attn.k_norm.bias": (cfg.head_dim,)})
        if cfg.use_swiglu:
            s.update({b + "mlp.w12.weight": (2 * Hs, D), b + "mlp.w12.bias": (2 * Hs,), b + "mlp.w3.weight": (D, Hs), b + "mlp.w3.bias": (D,)})
        else:
            s.update({b + "mlp.fc1.weight": (Hs, D), b + "mlp.fc1.bias": (Hs,), b + "mlp.fc2.weight": (D, Hs), b + "mlp.fc2.bias": (D,)})
    if cfg.use_rmsnorm:
        s["final_layer.norm_final.weight"] = (D,)
    s.update({
        "final_layer.linear.weight": (p * p * cfg.out_channels, D),
        "final_layer.linear.bias": (p * p * cfg.out_channels,),
        "final_layer.adaLN_modulation.1.weight": (2 * D, D),
        "final_layer.adaLN_modulation.1.bias": (2 * D,),
    })
    return s


def fixed_tables(cfg: DiTConfig) -> dict:
    """pos_embed (frozen parameter, lightningdit.py:350-351) + RoPE buffers."""
    pe = torch.from_numpy(sincos_pos_embed_2d(cfg.hidden_size, cfg.grid)).float().unsqueeze(0)
    if not cfg.use_rope:                      # feat_rope = None (:324-325): no buffers
        return {"pos_embed": pe}
    cos, sin = rope_tables(cfg.head_dim, cfg.grid)
    return {"pos_embed": pe, "feat_rope.freqs_cos": cos, "feat_rope.freqs_sin": sin}


def init_weights(cfg: DiTConfig, generator=None) -> dict:
    """Reference initialisation scheme (lightningdit.py:340-374): Xavier-uniform
    Linears, zero biases, N(0,.02) embedders, zero adaLN / final linear, unit
    RMSNorm weights.  Not RNG-order compatible with the reference (parity tests
    use explicit weights); the distribution is the same."""
    sd = {}
    for k, shp in param_shapes(cfg).items():
        if k == "pos_embed":
            continue
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("norm_final.weight") \
                or k.endswith("q_norm.weight") or k.endswith("k_norm.weight"):
            sd[k] = torch.ones(shp)
        elif k.endswith(".bias"):
            sd[k] = torch.zeros(shp)
        elif "adaLN_modulation" in k or k.startswith("final_layer.linear"):
            sd[k] = torch.zeros(shp)
        elif k.startswith("y_embedder") or k.startswith("t_embedder"):
            sd[k] = torch.randn(shp, generator=generator) * 0.02
        else:
            fan_out = shp[0]
            fan_in = int(np.prod(shp[1:]))
            a = math.sqrt(6.0 / (fan_in + fan_out))
            sd[k] = (torch.rand(shp, generator=generator) * 2 - 1) * a
    sd.update(fixed_tables(cfg))
    return sd
