"""Parity of every HIP kernel against the CPU oracle / a plain f32-f64 torch restatement of the same op,
called through the C ABI (ldmae_amd.ops -> ctypes -> libldmae_hip.so).

Tolerances: f32 path 1e-4 relative (north-star contract; most kernels are ~1e-6); bf16 path is checked
against the same math on bf16-rounded inputs with 2e-2 (bf16 has 8 significant bits); integer / index
outputs are bit-exact."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import dit as odit
from oracle import mae as omae
from oracle import train as otrain

pytestmark = pytest.mark.gpu

F32, BF16 = torch.float32, torch.bfloat16
TOL = {F32: 1e-4, BF16: 2e-2}


@pytest.fixture(scope="module")
def ops():
    from ldmae_amd import _lib, ops
    assert _lib.load().ldmae_arch() == b"gfx950"
    return ops


def dev(t, dtype=None):
    t = t.cuda()
    return t.to(dtype) if dtype is not None else t


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dtype):
    """Round to the activation dtype and back (what the kernel sees)."""
    return t.to(dtype).float()


# ----------------------------------------------------------------------------- GEMMs
@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("M,N,K", [(128, 384, 192), (200, 80, 64), (256, 16, 768), (64, 4608, 192), (1024, 768, 2048)])
def test_gemm_nt_bias(ops, dtype, M, N, K):
    a, b, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = q(a, dtype).double() @ q(b, dtype).double().T + bias.double()
    out = ops.gemm_nt(dev(a, dtype), dev(b, dtype), dev(bias))
    assert out.dtype == dtype and rel_err(out.float().cpu(), ref) < TOL[dtype]
    out32 = ops.gemm_nt(dev(a, dtype), dev(b, dtype), dev(bias), out_dtype=F32)
    assert rel_err(out32.cpu(), ref) < (1e-5 if dtype == F32 else 1e-6 + 1e-5)
    # beta = 1 accumulates
    ops.gemm_nt(dev(a, dtype), dev(b, dtype), None, out=out32, beta=1.0)
    assert rel_err(out32.cpu(), 2 * ref - bias.double()) < 1e-4


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gemm_nt_gate_res_pos_gelu(ops, dtype):
    B, T, D, K = 3, 64, 192, 128
    M = B * T
    a, w, bias = rnd(M, K, seed=1), rnd(D, K, seed=2, scale=K ** -0.5), rnd(D, seed=3)
    xin, mod = rnd(M, D, seed=4), rnd(B, 6 * D, seed=5)
    gate = mod[:, 2 * D:3 * D]
    y = q(a, dtype).double() @ q(w, dtype).double().T + bias.double()
    ref = xin.double() + gate.double().repeat_interleave(T, 0) * y
    modd = dev(mod)
    xo, ysave = ops.gemm_nt_gate_res(dev(a, dtype), dev(w, dtype), dev(bias), dev(xin), modd[:, 2 * D:3 * D], T)
    assert rel_err(xo.cpu(), ref) < TOL[dtype] and rel_err(ysave.float().cpu(), y) < TOL[dtype]
    # ungated residual (VMAE)
    xo2, _ = ops.gemm_nt_gate_res(dev(a, dtype), dev(w, dtype), dev(bias), dev(xin), None, T, save_y=False)
    assert rel_err(xo2.cpu(), xin.double() + y) < TOL[dtype]
    pos = rnd(T, D, seed=6)
    a32, w32 = rnd(M, 16, seed=7), rnd(D, 16, seed=8)
    o = ops.gemm_nt_pos(dev(a32), dev(w32), dev(bias), dev(pos), T)
    assert rel_err(o.cpu(), a32.double() @ w32.double().T + bias.double() + pos.double().repeat(B, 1)) < 1e-5
    g, pre = ops.gemm_nt_gelu(dev(a, dtype), dev(w, dtype), dev(bias))
    assert rel_err(pre.float().cpu(), y) < TOL[dtype]
    assert rel_err(g.float().cpu(), torch.nn.functional.gelu(y)) < TOL[dtype]
    # GELU backward fused into the dX GEMM of the next Linear: same bits as the GEMM followed by ldmae_gelu_bwd
    dy2, w2t = dev(rnd(M, 64, seed=9), dtype), dev(rnd(D, 64, seed=10, scale=0.125), dtype)
    assert torch.equal(ops.gemm_nt_gelu_bwd(dy2, w2t, pre), ops.gelu_bwd(ops.gemm_nt(dy2, w2t), pre))


def test_gemm_nt_persistent_multi_tile_ragged(ops):
    """More 256x256 tiles than CUs with ragged edges: every persistent workgroup walks several tiles (next-tile prefetch before
    the epilogue, XCD-owned row-block ranges with an uneven tail).  Checked against an f32 torch product of the same bf16 data;
    plain-bias, gated-residual, SwiGLU and SwiGLU-bwd epilogues, and the one-tile-per-workgroup launch mode."""
    from ldmae_amd import _lib
    M, N, K, T = 256 * 41 + 100, 2304 + 64, 192, 4
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randn(M, K, device="cuda", generator=g).to(BF16)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(BF16)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = a.float() @ w.float().T + bias
    outs = {}
    for mode in (0, 2):
        ops.set_gemm_launch_mode("tile" if mode == 2 else "persistent")      # per-call flag of the C ABI (LDMAE_EPI_TILE_LAUNCH)
        out = ops.gemm_nt(a, w, bias)
        outs[mode] = out
        assert rel_err(out.float().cpu(), ref.cpu()) < TOL[BF16]
        assert torch.equal(out[-1].float(), ops.gemm_nt(a[-1:].contiguous().expand(8, K).contiguous(), w, bias)[0].float())
        xin = torch.randn(M, N, device="cuda", generator=g)
        gate = torch.randn(M // T, N, device="cuda", generator=g)
        xo, y = ops.gemm_nt_gate_res(a, w, bias, xin, gate, T)
        assert torch.equal(y, out)
        # the residual adds y AS STORED (bf16) -- what the reference's autocast Linear hands to `x + gate * branch` (lightningdit.py:248-249)
        assert rel_err(xo.cpu(), torch.addcmul(xin, gate.repeat_interleave(T, 0), y.float()).cpu()) < 1e-6
        xo32, _ = ops.gemm_nt_gate_res(a, w, bias, xin, gate, T, save_y=False, y_dtype=torch.float32)      # y_dtype f32: the unrounded product
        assert rel_err(xo32.cpu(), (xin + gate.repeat_interleave(T, 0) * ref).cpu()) < 1e-5
    ops.set_gemm_launch_mode("persistent")
    assert not hasattr(_lib.load(), "ldmae_tune")    # the product library has no process-wide knobs (csrc/probe/ldmae_diag.h is a separate build)
    assert torch.equal(outs[0], outs[2])             # persistent and one-tile-per-workgroup launches (the multi-rank mode): bitwise equal
    Hs = 1280                                        # N = 2560 = 10 tile columns, 420 tiles
    w12 = (torch.randn(2 * Hs, K, device="cuda", generator=g) * K ** -0.5).to(BF16)
    b12 = torch.randn(2 * Hs, device="cuda", generator=g)
    h12, hid = ops.gemm_nt_swiglu(a, w12, b12)
    h12_ref = ops.gemm_nt(a, w12, b12)
    assert torch.equal(h12, h12_ref) and torch.equal(hid, ops.swiglu_fwd(h12_ref))
    pre = ops.gemm_nt(a, w, bias)                       # [M, N] ragged multi-tile: the GELU-backward epilogue on the persistent path
    w2t = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(BF16)
    assert torch.equal(ops.gemm_nt_gelu_bwd(a, w2t, pre), ops.gelu_bwd(ops.gemm_nt(a, w2t), pre))
    w3t = (torch.randn(Hs, K, device="cuda", generator=g) * K ** -0.5).to(BF16)
    dh12, db12 = ops.gemm_nt_swiglu_bwd(a, w3t, h12, with_bias=True)
    unf = ops.swiglu_bwd(ops.gemm_nt(a, w3t), h12)
    # same formula in both kernels; the compiler may contract an FMA differently: at most a last-bit difference in a few of 27 M values
    nbad = int((dh12 != unf).sum())
    assert nbad <= 1e-6 * dh12.numel() and rel_err(dh12.float().cpu(), unf.float().cpu()) < 1e-6, nbad
    assert rel_err(db12.cpu(), dh12.float().sum(0).cpu()) < 1e-5


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("M,N,K", [(128, 384, 192), (4096, 16, 768), (8192, 576, 192), (256, 1152, 200), (16384, 768, 2048), (32768, 2304, 768)])
def test_gemm_tn(ops, dtype, M, N, K):
    if dtype == BF16 and K % 8:
        pytest.skip("bf16 needs K % 8 == 0")
    a, b = rnd(M, N, seed=1), rnd(M, K, seed=2)
    ref = q(a, dtype).double().T @ q(b, dtype).double()
    out = ops.gemm_tn(dev(a, dtype), dev(b, dtype))
    assert out.dtype == F32 and rel_err(out.cpu(), ref) < 1e-5
    out2 = ops.gemm_tn(dev(a, dtype), dev(b, dtype), out=out.clone(), beta=1.0)
    assert rel_err(out2.cpu(), 2 * ref) < 1e-5
    # deterministic: two launches are bitwise identical
    assert torch.equal(ops.gemm_tn(dev(a, dtype), dev(b, dtype)), out)
    out3, db = ops.gemm_tn(dev(a, dtype), dev(b, dtype), with_bias=True)      # fused bias gradient = column sums of a
    assert torch.equal(out3, out) and rel_err(db.cpu(), q(a, dtype).double().sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_colsum_cast(ops, dtype):
    x = rnd(1000, 576, seed=3)
    assert rel_err(ops.colsum(dev(x, dtype)).cpu(), q(x, dtype).double().sum(0)) < 1e-5
    w = rnd(80, 200, seed=4)
    s, t = ops.cast_weight(dev(w), dtype)
    assert torch.equal(s.cpu(), w.to(dtype)) and torch.equal(t.cpu(), w.to(dtype).T.contiguous())
    assert torch.equal(ops.cast(dev(rnd(1003, seed=5)), BF16).cpu(), rnd(1003, seed=5).to(BF16))


# ----------------------------------------------------------------------------- norm / elementwise
@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("D", [192, 768, 1152])
def test_rmsnorm_modulate(ops, dtype, D):
    B, T = 2, 64
    M = B * T
    x = rnd(M, D, seed=1).requires_grad_(True)
    w = (1 + 0.1 * rnd(D, seed=2)).requires_grad_(True)
    mod = (0.3 * rnd(B, 6 * D, seed=3)).requires_grad_(True)
    sh, sc = mod[:, 3 * D:4 * D], mod[:, 4 * D:5 * D]
    ref = odit.modulate(odit.rmsnorm(x.view(B, T, D), w), sh, sc).view(M, D)
    g = rnd(M, D, seed=4)
    gq = q(g, dtype)
    ref.backward(gq)
    modd = dev(mod.detach())
    out, rstd = ops.rmsnorm_modulate_fwd(dev(x.detach()), dev(w.detach()), modd[:, 3 * D:4 * D], modd[:, 4 * D:5 * D], T, dtype)
    assert rel_err(out.float().cpu(), ref.detach()) < (1e-5 if dtype == F32 else 1e-2)
    dx = dev(rnd(M, D, seed=5))
    dx0 = dx.clone()
    dmod = torch.zeros(B, 6 * D, device="cuda")
    dw = ops.rmsnorm_modulate_bwd(dev(g, dtype), dev(x.detach()), dev(w.detach()), modd[:, 4 * D:5 * D], rstd, dx,
                                  dmod[:, 3 * D:4 * D], dmod[:, 4 * D:5 * D], T)
    assert rel_err((dx - dx0).cpu(), x.grad) < 1e-4
    assert rel_err(dw.cpu(), w.grad) < 1e-4
    assert rel_err(dmod.cpu(), mod.grad) < 1e-4


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("D,T", [(768, 64), (192, 128), (1152, 64)])
def test_rmsnorm_modulate_bwd_gate_fused(ops, dtype, D, T):
    """The one-pass form (norm2 backward + the attention branch's gate backward) against the two separate entry points it replaces:
    same arithmetic, same partial layout and summation order, so every output is bitwise equal."""
    B = 3
    M = B * T
    x, w, g = dev(rnd(M, D, seed=1)), dev(1 + 0.1 * rnd(D, seed=2)), dev(rnd(M, D, seed=4), dtype)
    mod = dev(0.3 * rnd(B, 6 * D, seed=3))
    y = dev(rnd(M, D, seed=6), dtype)
    _, rstd = ops.rmsnorm_modulate_fwd(x, w, mod[:, 3 * D:4 * D], mod[:, 4 * D:5 * D], T, dtype)
    dx_a, dx_b = dev(rnd(M, D, seed=5)), dev(rnd(M, D, seed=5))
    dmod_a, dmod_b = torch.zeros(B, 6 * D, device="cuda"), torch.zeros(B, 6 * D, device="cuda")
    dw_a = ops.rmsnorm_modulate_bwd(g, x, w, mod[:, 4 * D:5 * D], rstd, dx_a, dmod_a[:, 3 * D:4 * D], dmod_a[:, 4 * D:5 * D], T)
    dy_a, db_a = ops.gate_bwd(dx_a, y, mod[:, 2 * D:3 * D], dmod_a[:, 2 * D:3 * D], T, dtype, with_bias=True)
    dw_b, dy_b, db_b = ops.rmsnorm_modulate_bwd_gate(g, x, w, mod[:, 4 * D:5 * D], rstd, dx_b, dmod_b[:, 3 * D:4 * D], dmod_b[:, 4 * D:5 * D],
                                                     y, mod[:, 2 * D:3 * D], dmod_b[:, 2 * D:3 * D], T, dtype)
    for name, a, b_ in (("dx", dx_a, dx_b), ("dmod", dmod_a, dmod_b), ("dw", dw_a, dw_b), ("dy", dy_a, dy_b), ("dbias", db_a, db_b)):
        assert torch.equal(a, b_), name


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("hd,H", [(64, 3), (72, 2)])
def test_qknorm_rope(ops, dtype, hd, H):
    B, grid = 2, 8
    N = grid * grid
    cos, sin = odit.rope_tables(hd, grid)
    qkv = q(rnd(B, N, 3, H, hd, seed=1), dtype).requires_grad_(True)
    wq, wk = (1 + 0.1 * rnd(hd, seed=2)).requires_grad_(True), (1 + 0.1 * rnd(hd, seed=3)).requires_grad_(True)
    t = qkv.permute(2, 0, 3, 1, 4)
    rq = odit.apply_rope(odit.rmsnorm(t[0], wq), cos, sin)
    rk = odit.apply_rope(odit.rmsnorm(t[1], wk), cos, sin)
    gq_, gk_, gv_ = q(rnd(B, H, N, hd, seed=4), dtype), q(rnd(B, H, N, hd, seed=5), dtype), q(rnd(B, H, N, hd, seed=6), dtype)
    (rq * gq_).sum().add((rk * gk_).sum()).add((t[2] * gv_).sum()).backward()
    qd, kd, vd = ops.qknorm_rope_fwd(dev(qkv.detach(), dtype), dev(wq.detach()), dev(wk.detach()), dev(cos), dev(sin), B, N, H, hd)
    tol = 1e-5 if dtype == F32 else 1e-2
    assert rel_err(qd.float().cpu(), rq.detach()) < tol and rel_err(kd.float().cpu(), rk.detach()) < tol
    assert torch.equal(vd.float().cpu(), t[2].detach().contiguous())
    dqkv, dwq, dwk = ops.qknorm_rope_bwd(dev(gq_, dtype), dev(gk_, dtype), dev(gv_, dtype), dev(qkv.detach(), dtype), dev(wq.detach()),
                                         dev(wk.detach()), dev(cos), dev(sin), B, N, H, hd)
    assert rel_err(dqkv.float().cpu(), qkv.grad) < tol
    assert rel_err(dwq.cpu(), wq.grad) < 1e-4 and rel_err(dwk.cpu(), wk.grad) < 1e-4
    dqkv2, _, _, db = ops.qknorm_rope_bwd(dev(gq_, dtype), dev(gk_, dtype), dev(gv_, dtype), dev(qkv.detach(), dtype), dev(wq.detach()),
                                          dev(wk.detach()), dev(cos), dev(sin), B, N, H, hd, with_bias=True)     # fused qkv bias gradient
    assert torch.equal(dqkv2, dqkv)
    assert rel_err(db.cpu(), dqkv.float().cpu().reshape(B * N, 3 * H * hd).sum(0)) < 1e-5


@pytest.mark.parametrize("hd,H,N", [(64, 3, 128), (64, 12, 1024), (128, 2, 192)])
def test_attention_bwd_pv_qknorm_fused(ops, hd, H, N):
    """Attention backward with the QK-RMSNorm / RoPE backward folded into its epilogues against the two entry points it replaces
    (attention_bwd_pv + qknorm_rope_bwd): same per-element formulas on the same bf16-rounded dq / dk, so dqkv agrees to bf16 rounding of
    isolated elements (the row sums are formed over 8 instead of 16 lanes) and the weight / bias gradient sums to f32 summation order."""
    B, grid = 2, int(N ** 0.5) if int(N ** 0.5) ** 2 == N else None
    g = torch.Generator().manual_seed(11)
    qkv = (torch.randn(B, N, 3, H, hd, generator=g)).to(BF16).cuda()
    wq, wk = (1 + 0.1 * torch.randn(hd, generator=g)).cuda(), (1 + 0.1 * torch.randn(hd, generator=g)).cuda()
    cos, sin = torch.rand(N, hd, generator=g).cuda(), torch.rand(N, hd, generator=g).cuda()      # any table: the kernels only index it
    do = torch.randn(B, N, H * hd, generator=g).to(BF16).cuda()
    q, k, _ = ops.qknorm_rope_fwd(qkv, wq, wk, cos, sin, B, N, H, hd, copy_v=False)
    o, lse = ops.attention_fwd_pv(q, k, qkv, hd ** -0.5)
    dq, dk, dqkv_a = ops.attention_bwd_pv(q, k, qkv, o, do, lse, hd ** -0.5)
    dqkv_a, dwq_a, dwk_a, db_a = ops.qknorm_rope_bwd(dq, dk, None, qkv, wq, wk, cos, sin, B, N, H, hd, with_bias=True, dqkv=dqkv_a)
    dqkv_b, dwq_b, dwk_b, db_b = ops.attention_bwd_pv_qknorm(q, k, qkv, o, do, lse, hd ** -0.5, wq, wk, cos, sin)
    assert torch.equal(dqkv_a[:, :, 2], dqkv_b[:, :, 2])                      # dv: identical path
    assert rel_err(dqkv_b.float().cpu(), dqkv_a.float().cpu()) < 2e-3
    assert (dqkv_a != dqkv_b).float().mean().item() < 0.02
    assert rel_err(dwq_b.cpu(), dwq_a.cpu()) < 1e-4 and rel_err(dwk_b.cpu(), dwk_a.cpu()) < 1e-4
    assert rel_err(db_b.cpu(), db_a.cpu()) < 1e-3
    assert rel_err(db_b.cpu(), dqkv_b.float().cpu().reshape(B * N, 3 * H * hd).sum(0)) < 1e-5      # = column sums of dqkv as stored


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("hd,H", [(64, 3), (72, 2), (128, 2)])
def test_rope_only_front_end_without_qknorm(ops, dtype, hd, H):
    """ldmae_qknorm_rope_fwd / _bwd with wq = wk = NULL and tables: the attention front end of a block built with use_qknorm=False (q_norm =
    k_norm = nn.Identity, lightningdit.py:60-61,69-74; the reference's CelebA-HQ YAML).  Forward = the rotation of the head-major split, what
    ldmae_rope gives (with and without the head-major copy of v -- the 16-B bf16 kernel and the generic one -- bit for bit the same); backward = its adjoint
    and the qkv bias gradient; identity tables (use_rope=False) return the split itself."""
    B, grid = 2, 8
    N = grid * grid
    cos, sin = odit.rope_tables(hd, grid)
    qkv = q(rnd(B, N, 3, H, hd, seed=1), dtype)
    t = qkv.permute(2, 0, 3, 1, 4)
    rq, rk = odit.apply_rope(t[0], cos, sin), odit.apply_rope(t[1], cos, sin)
    qkv_d = dev(qkv, dtype)
    qd, kd, vd = ops.qknorm_rope_fwd(qkv_d, None, None, dev(cos), dev(sin), B, N, H, hd)
    tol = 1e-6 if dtype == F32 else 1e-2
    assert rel_err(qd.float().cpu(), rq) < tol and rel_err(kd.float().cpu(), rk) < tol
    assert torch.equal(vd.float().cpu(), t[2].contiguous())
    qs, ks, _ = ops.heads_split(qkv_d, B, N, H, hd)
    # against the standalone rotation kernel: the same products, contracted into FMAs differently by the compiler (last-bit differences)
    assert rel_err(qd.float(), ops.rope(qs, dev(cos), dev(sin)).float()) < tol and rel_err(kd.float(), ops.rope(ks, dev(cos), dev(sin)).float()) < tol
    q2, k2, v2 = ops.qknorm_rope_fwd(qkv_d, None, None, dev(cos), dev(sin), B, N, H, hd, copy_v=False)
    assert v2 is None and torch.equal(q2, qd) and torch.equal(k2, kd)
    qi, ki, _ = ops.qknorm_rope_fwd(qkv_d, None, None, torch.ones_like(dev(cos)), torch.zeros_like(dev(sin)), B, N, H, hd)
    assert torch.equal(qi, qs) and torch.equal(ki, ks)
    gq_, gk_, gv_ = (dev(q(rnd(B, H, N, hd, seed=s_), dtype), dtype) for s_ in (4, 5, 6))
    dqkv, dwq, dwk, db = ops.qknorm_rope_bwd(gq_, gk_, gv_, qkv_d, None, None, dev(cos), dev(sin), B, N, H, hd, with_bias=True)
    assert dwq is None and dwk is None
    ref = ops.heads_merge(ops.rope(gq_, dev(cos), dev(sin), transposed=True), ops.rope(gk_, dev(cos), dev(sin), transposed=True), gv_, B, N, H, hd)
    assert rel_err(dqkv.reshape(ref.shape).float(), ref.float()) < tol
    assert torch.equal(dqkv.reshape(B, N, 3, H, hd)[:, :, 2], ref.reshape(B, N, 3, H, hd)[:, :, 2])      # dv: a copy
    assert rel_err(db.cpu(), dqkv.float().cpu().reshape(B * N, 3 * H * hd).sum(0)) < 1e-5


@pytest.mark.parametrize("hd,H,N", [(64, 3, 128), (64, 12, 1024), (128, 2, 192)])
def test_attention_bwd_pv_rope_only_fused(ops, hd, H, N):
    """The fused attention backward with wq = wk = NULL (use_qknorm=False): the epilogues apply the rotation's adjoint only.  Against
    attention_bwd_pv + the rope-only qknorm_rope_bwd: dv identical, dq / dk the same bf16 values (one rounding of the same f32 products
    either way, then an exact rotation of rounded inputs -- equal up to the rounding of the rotated sum), bias gradient = column sums."""
    B = 2
    g = torch.Generator().manual_seed(12)
    qkv = (torch.randn(B, N, 3, H, hd, generator=g)).to(BF16).cuda()
    ang = torch.rand(N, hd // 2, generator=g) * 6.28
    cos, sin = ang.cos().repeat_interleave(2, 1).cuda(), ang.sin().repeat_interleave(2, 1).cuda()
    do = torch.randn(B, N, H * hd, generator=g).to(BF16).cuda()
    q, k, _ = ops.qknorm_rope_fwd(qkv, None, None, cos, sin, B, N, H, hd, copy_v=False)
    o, lse = ops.attention_fwd_pv(q, k, qkv, hd ** -0.5)
    dq, dk, dqkv_a = ops.attention_bwd_pv(q, k, qkv, o, do, lse, hd ** -0.5)
    dqkv_a, _, _, db_a = ops.qknorm_rope_bwd(dq, dk, None, qkv, None, None, cos, sin, B, N, H, hd, with_bias=True, dqkv=dqkv_a)
    dqkv_b, dwq_b, dwk_b, db_b = ops.attention_bwd_pv_qknorm(q, k, qkv, o, do, lse, hd ** -0.5, None, None, cos, sin)
    assert dwq_b is None and dwk_b is None
    assert torch.equal(dqkv_a[:, :, 2], dqkv_b[:, :, 2])
    assert rel_err(dqkv_b.float().cpu(), dqkv_a.float().cpu()) < 2e-3
    assert (dqkv_a != dqkv_b).float().mean().item() < 0.02
    assert rel_err(db_b.cpu(), db_a.cpu()) < 1e-3
    assert rel_err(db_b.cpu(), dqkv_b.float().cpu().reshape(B * N, 3 * H * hd).sum(0)) < 1e-5
    # and against f64 math: softmax attention on the rotated heads, gradients rotated back
    qd, kd = q.double().cpu().requires_grad_(True), k.double().cpu().requires_grad_(True)
    vd = qkv[:, :, 2].permute(0, 2, 1, 3).double().cpu().requires_grad_(True)
    ref = torch.softmax(qd @ kd.transpose(-1, -2) * hd ** -0.5, -1) @ vd
    (ref * do.view(B, N, H, hd).permute(0, 2, 1, 3).double().cpu()).sum().backward()
    rot_t = lambda g_: g_ * cos.double().cpu() - odit.rotate_pairs(g_ * sin.double().cpu())      # noqa: E731   (adjoint of t*cos + rot(t)*sin)
    want = torch.stack([rot_t(qd.grad), rot_t(kd.grad), vd.grad], 0).permute(1, 3, 0, 2, 4)      # [B,N,3,H,hd]
    assert rel_err(dqkv_b.float().cpu(), want.float()) < 2e-2


@pytest.mark.parametrize("hd,H,N", [(64, 3, 256), (72, 2, 192), (64, 2, 200)])
def test_attention_fwd_static_shift_from_the_qknorm_bound(ops, hd, H, N):
    """attention_fwd_pv with the score bound of QK-normalised heads (ldmae_qk_score_bound: hd max|wq| max|wk| scale log2e): the kernel then
    shifts every exponent by that bound instead of tracking a running maximum.  Same softmax: o and lse against the f64 reference and
    against the tracked form; the bound holds for the scores the kernel sees; a bound above 50 keeps the tracked form, bit for bit."""
    B = 2
    g = torch.Generator().manual_seed(5)
    qkv = (2 * torch.randn(B, N, 3, H, hd, generator=g)).to(BF16).cuda()
    wq, wk = (1 + 0.2 * torch.randn(hd, generator=g)).cuda(), (1 + 0.2 * torch.randn(hd, generator=g)).cuda()
    ang = torch.rand(N, hd // 2, generator=g) * 6.28
    cos, sin = ang.cos().repeat_interleave(2, 1).cuda(), ang.sin().repeat_interleave(2, 1).cuda()      # a rotation per pair: norms are kept
    scale = hd ** -0.5
    q, k, _ = ops.qknorm_rope_fwd(qkv, wq, wk, cos, sin, B, N, H, hd, copy_v=False)
    bound = ops.qk_score_bound(wq, wk, hd, scale)
    assert abs(float(bound) - hd * float(wq.abs().max()) * float(wk.abs().max()) * scale * 1.4426950408889634 * 1.02) < 1e-3 * float(bound)
    smax = float((q.double() @ k.double().transpose(-1, -2)).abs().max()) * scale * 1.4426950408889634
    assert smax <= float(bound) <= 50
    o_t, lse_t = ops.attention_fwd_pv(q, k, qkv, scale)
    o_s, lse_s = ops.attention_fwd_pv(q, k, qkv, scale, bound=bound)
    v = qkv[:, :, 2].permute(0, 2, 1, 3)
    ro, rl = _attn_ref(q.double().cpu(), k.double().cpu(), v.double().cpu(), scale)
    assert rel_err(o_s.float().cpu(), ro.float()) < 1e-2 and rel_err(o_t.float().cpu(), ro.float()) < 1e-2
    assert (lse_s.cpu() - rl.float()).abs().max() < 2e-2 and (lse_s - lse_t).abs().max() < 2e-2
    assert rel_err(o_s.float().cpu(), o_t.float().cpu()) < 5e-3
    big = torch.full((1,), 80.0, device="cuda")
    o_b, lse_b = ops.attention_fwd_pv(q, k, qkv, scale, bound=big)
    assert torch.equal(o_b, o_t) and torch.equal(lse_b, lse_t)


@pytest.mark.parametrize("hd,H,N", [(16, 3, 512), (32, 2, 640)])
def test_attention_fwd_qkv_static_shift_from_the_key_norm_pass(ops, hd, H, N):
    """attention_fwd_qkv on long sequences of small heads: one pass over the k slots (ldmae_k_norm_max) + each query's own norm bound the
    scores, the flash kernel then runs without a running maximum.  Same softmax as the tracked form and as the f64 reference; the maxima
    are the true maxima; huge keys (bound > 50) fall back to the tracked form bit for bit."""
    import ldmae_amd.ops as opsmod
    B = 2
    g = torch.Generator().manual_seed(7)
    qkv = torch.randn(B * N, 3 * H * hd, generator=g).to(BF16).cuda()
    scale = hd ** -0.5
    old = opsmod.BOUNDED_ATTENTION_MIN_SCORES
    try:
        opsmod.BOUNDED_ATTENTION_MIN_SCORES = 1 << 62
        o_t, lse_t = ops.attention_fwd_qkv(qkv, B, N, H, hd, scale)
        opsmod.BOUNDED_ATTENTION_MIN_SCORES = 0
        o_s, lse_s = ops.attention_fwd_qkv(qkv, B, N, H, hd, scale)
        o_f = ops.attention_fwd_qkv(qkv.float(), B, N, H, hd, scale)[0] if hd == 16 else None       # f32 packed qkv: never the (bf16) key-norm pass
        big = (qkv.float() * 40).to(BF16)
        o_b, lse_b = ops.attention_fwd_qkv(big, B, N, H, hd, scale)
        opsmod.BOUNDED_ATTENTION_MIN_SCORES = 1 << 62
        o_bt, lse_bt = ops.attention_fwd_qkv(big, B, N, H, hd, scale)
    finally:
        opsmod.BOUNDED_ATTENTION_MIN_SCORES = old
    q, k, v = (qkv.view(B, N, 3, H, hd)[:, :, i].permute(0, 2, 1, 3).double().cpu() for i in range(3))
    ro, rl = _attn_ref(q, k, v, scale)
    assert rel_err(o_s.float().cpu(), ro.float()) < 1e-2 and rel_err(o_t.float().cpu(), ro.float()) < 1e-2
    assert (lse_s.cpu() - rl.float()).abs().max() < 2e-2 and rel_err(o_s.float().cpu(), o_t.float().cpu()) < 5e-3
    assert torch.equal(o_b, o_bt) and torch.equal(lse_b, lse_bt)           # bound far above 50: the tracked form
    if o_f is not None:
        assert rel_err(o_f.cpu(), ro.float()) < 1e-5
    kmax = torch.empty(B * H, 2, device="cuda")
    from ldmae_amd._lib import call
    from ldmae_amd.ops import ptr, stream
    call("ldmae_k_norm_max", ptr(qkv), ptr(kmax), B, N, H, hd, stream())
    want = (k.float() ** 2).sum(-1).amax(-1).reshape(-1)
    assert torch.equal(kmax[:, 0].cpu(), torch.zeros(B * H)) and rel_err(kmax[:, 1].cpu(), want) < 1e-6


def _attn_ref(qq, kk, vv, scale):
    s = (qq @ kk.transpose(-2, -1)) * scale
    o = s.softmax(-1) @ vv
    B, H, N, hd = qq.shape
    return o.transpose(1, 2).reshape(B, N, H * hd), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype,hd,N", [(F32, 64, 64), (F32, 64, 192), (F32, 16, 256), (F32, 72, 128), (BF16, 64, 64), (BF16, 64, 192),
                                        (BF16, 64, 1024), (BF16, 72, 256), (BF16, 128, 128), (BF16, 16, 256), (BF16, 16, 1024),
                                        (BF16, 32, 128), (BF16, 72, 1024),
                                        # token counts that are not a multiple of 64 (models_mae.py:472-497: int(L * (1 - mask_ratio)) kept tokens;
                                        # N + 1 with a cls token): the last tile of every sweep is ragged
                                        (BF16, 16, 200), (BF16, 16, 257), (BF16, 64, 200), (BF16, 64, 257), (BF16, 64, 40), (BF16, 72, 100),
                                        (F32, 16, 200), (F32, 64, 257), (F32, 64, 40)])
def test_attention(ops, dtype, hd, N):
    """bf16 head dims 16 / 72 run on the flash kernels with the LDS images zero-padded to 32 / 96 columns (no padded HBM copies)."""
    B, H = 2, 3
    mk = lambda s: q(rnd(B, H, N, hd, seed=s), dtype).double().requires_grad_(True)
    qq, kk, vv = mk(1), mk(2), mk(3)
    scale = hd ** -0.5
    o, lse = _attn_ref(qq, kk, vv, scale)
    g = q(rnd(B, N, H * hd, seed=4), dtype).double()
    o.backward(g)
    od, lsed = ops.attention_fwd(dev(qq.detach().float(), dtype), dev(kk.detach().float(), dtype), dev(vv.detach().float(), dtype), scale)
    tol = 2e-5 if dtype == F32 else 2e-2
    assert rel_err(od.float().cpu(), o.detach()) < tol
    assert rel_err(lsed.cpu(), lse.detach()) < (1e-5 if dtype == F32 else 2e-3)
    dq, dk, dv = ops.attention_bwd(dev(qq.detach().float(), dtype), dev(kk.detach().float(), dtype), dev(vv.detach().float(), dtype),
                                   od, dev(g.float(), dtype), lsed, scale)
    for got, ref in ((dq, qq.grad), (dk, kk.grad), (dv, vv.grad)):
        assert rel_err(got.float().cpu(), ref) < (1e-4 if dtype == F32 else 3e-2)


def test_attention_ragged_n_writes_nothing_past_n(ops):
    """Ragged N, packed VMAE layout, last (batch, head) of the buffer: rows past N do not exist -- guard pages of NaN canaries placed
    right behind every output must survive, and NaNs placed right behind the INPUTS must not leak into any result."""
    B, H, N, hd = 2, 12, 204, 16                                     # mask_ratio 0.8 of 1024 patches keeps 204 tokens
    tot = B * N * 3 * H * hd
    buf = torch.full((tot + 4096,), float("nan"), device="cuda").to(BF16)
    buf[:tot] = dev(rnd(tot, seed=1), BF16)
    qkv = buf[:tot].view(B * N, 3 * H * hd)
    dobuf = torch.full((B * N * H * hd + 4096,), float("nan"), device="cuda").to(BF16)
    dobuf[:B * N * H * hd] = dev(rnd(B * N * H * hd, seed=2), BF16)
    do = dobuf[:B * N * H * hd].view(B, N, H * hd)
    o, lse = ops.attention_fwd_qkv(qkv, B, N, H, hd, hd ** -0.5)
    dqkv = ops.attention_bwd_qkv(qkv, o, do, lse, B, N, H, hd, hd ** -0.5)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all() and torch.isfinite(dqkv.float()).all()
    qh, kh, vh = (t.float().cpu().double().requires_grad_(True) for t in ops.heads_split(qkv, B, N, H, hd))
    oref, lref = _attn_ref(qh, kh, vh, hd ** -0.5)
    oref.backward(do.float().cpu().double())
    assert rel_err(o.float().cpu(), oref.detach()) < 2e-2 and rel_err(lse.cpu(), lref.detach()) < 2e-3
    ref = ops.heads_merge(dev(qh.grad.float(), BF16), dev(kh.grad.float(), BF16), dev(vh.grad.float(), BF16), B, N, H, hd)
    assert rel_err(dqkv.float().cpu(), ref.float().cpu()) < 3e-2


def test_gemm_tn_ragged_rows(ops):
    """Weight gradient over a token count that is not a multiple of 64 (B * kept tokens of a ragged VMAE batch): the rows past the end of
    the last split are fetched as zeros."""
    M, N, K = 2 * 204, 576, 192
    a, b = dev(rnd(M, N, seed=1), BF16), dev(rnd(M, K, seed=2), BF16)
    out, db = ops.gemm_tn(a, b, with_bias=True)
    ref = a.float().cpu().double().T @ b.float().cpu().double()
    assert rel_err(out.cpu(), ref) < 1e-5 and rel_err(db.cpu(), a.float().cpu().double().sum(0)) < 1e-5


@pytest.mark.parametrize("hd,N", [(16, 256), (64, 128), (16, 200)])
def test_attention_packed_qkv_equals_head_major(ops, hd, N):
    """The VMAE path: flash attention straight on the packed token-major qkv [B,N,3,H,hd] (q / k / v read, dq / dk / dv written in
    place) gives bitwise the results of the head-major path behind the relayout kernels."""
    B, H = 2, 12
    qkv = dev(rnd(B * N, 3 * H * hd, seed=1), BF16)
    do = dev(rnd(B, N, H * hd, seed=2), BF16)
    qh, kh, vh = ops.heads_split(qkv, B, N, H, hd)
    o1, lse1 = ops.attention_fwd(qh, kh, vh, hd ** -0.5)
    o2, lse2 = ops.attention_fwd_qkv(qkv, B, N, H, hd, hd ** -0.5)
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2)
    dq, dk, dv = ops.attention_bwd(qh, kh, vh, o1, do, lse1, hd ** -0.5)
    dqkv = ops.attention_bwd_qkv(qkv, o2, do, lse2, B, N, H, hd, hd ** -0.5)
    assert torch.equal(dqkv, ops.heads_merge(dq, dk, dv, B, N, H, hd))


def test_attention_softmax_rescale_branch(ops):
    """Force the running max to jump at a chosen key tile (guide rule 26): one huge score late in the sequence."""
    B, H, N, hd = 1, 1, 256, 64
    qq, kk, vv = rnd(B, H, N, hd, seed=1), rnd(B, H, N, hd, seed=2), rnd(B, H, N, hd, seed=3)
    kk[0, 0, 200] = qq[0, 0, 5] * 8.0
    o, lse = _attn_ref(qq.double(), kk.double(), vv.double(), hd ** -0.5)
    for dtype in (F32, BF16):
        od, lsed = ops.attention_fwd(dev(qq, dtype), dev(kk, dtype), dev(vv, dtype), hd ** -0.5)
        o_r, lse_r = _attn_ref(q(qq, dtype).double(), q(kk, dtype).double(), q(vv, dtype).double(), hd ** -0.5)
        assert rel_err(od.float().cpu(), o_r) < (2e-5 if dtype == F32 else 2e-2)
        assert rel_err(lsed.cpu(), lse_r) < 2e-3


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_swiglu_gate_silu(ops, dtype):
    M, Hs, B, T, D = 128, 512, 2, 64, 192
    h12 = q(rnd(M, 2 * Hs, seed=1), dtype).requires_grad_(True)
    x1, x2 = h12.chunk(2, -1)
    ref = torch.nn.functional.silu(x1) * x2
    g = q(rnd(M, Hs, seed=2), dtype)
    ref.backward(g)
    tol = 1e-5 if dtype == F32 else 1e-2
    assert rel_err(ops.swiglu_fwd(dev(h12.detach(), dtype)).float().cpu(), ref.detach()) < tol
    assert rel_err(ops.swiglu_bwd(dev(g, dtype), dev(h12.detach(), dtype)).float().cpu(), h12.grad) < tol
    dxo, y, mod = rnd(M, D, seed=3), q(rnd(M, D, seed=4), dtype), rnd(B, 6 * D, seed=5)
    modd = dev(mod)
    dmod = torch.zeros(B, 6 * D, device="cuda")
    dy = ops.gate_bwd(dev(dxo), dev(y, dtype), modd[:, 5 * D:], dmod[:, 5 * D:], T, dtype)
    gate = mod[:, 5 * D:].repeat_interleave(T, 0)
    assert rel_err(dy.float().cpu(), dxo * gate) < tol
    assert rel_err(dmod[:, 5 * D:].cpu(), (dxo * y).view(B, T, D).sum(1)) < 1e-5
    dy2, db = ops.gate_bwd(dev(dxo), dev(y, dtype), modd[:, 5 * D:], None, T, dtype, with_bias=True)     # producer-side bias gradient
    assert torch.equal(dy2, dy)
    assert rel_err(db.cpu(), dy.float().cpu().sum(0)) < 1e-5          # column sums of dy exactly as stored
    if dtype == F32:
        c = rnd(5, 192, seed=6).requires_grad_(True)
        torch.nn.functional.silu(c).backward(g[:5, :192])
        assert rel_err(ops.silu_fwd(dev(c.detach())).cpu(), torch.nn.functional.silu(c.detach())) < 1e-6
        assert rel_err(ops.silu_bwd(dev(g[:5, :192].contiguous()), dev(c.detach())).cpu(), c.grad) < 1e-5


def test_fused_swiglu_gemm_epilogues_match_unfused(ops):
    M, D, Hs = 512, 192, 512
    a, w12, b12 = dev(rnd(M, D, seed=1), BF16), dev(rnd(2 * Hs, D, seed=2, scale=D ** -0.5), BF16), dev(rnd(2 * Hs, seed=3))
    h12, hid = ops.gemm_nt_swiglu(a, w12, b12)
    h12_ref = ops.gemm_nt(a, w12, b12)
    assert torch.equal(h12, h12_ref) and torch.equal(hid, ops.swiglu_fwd(h12_ref))
    dy, w3t = dev(rnd(M, D, seed=4), BF16), dev(rnd(Hs, D, seed=5, scale=D ** -0.5), BF16)
    assert torch.equal(ops.gemm_nt_swiglu_bwd(dy, w3t, h12), ops.swiglu_bwd(ops.gemm_nt(dy, w3t), h12))
    dh12, db12 = ops.gemm_nt_swiglu_bwd(dy, w3t, h12, with_bias=True)          # epilogue-side bias gradient of w12
    assert rel_err(db12.cpu(), dh12.float().sum(0).cpu()) < 1e-5
    # f32 path: unfused kernels behind the same entry points
    a32, w32 = dev(rnd(64, D, seed=1)), dev(rnd(2 * Hs, D, seed=2, scale=D ** -0.5))
    h32, hid32 = ops.gemm_nt_swiglu(a32, w32, b12)
    x1, x2 = h32.chunk(2, -1)
    assert rel_err(hid32.cpu(), (torch.nn.functional.silu(x1) * x2).cpu()) < 1e-5


def test_embedders(ops, golden):
    g = golden("kernels")
    t = torch.tensor([0.0, 0.25, 0.9])
    np.testing.assert_allclose(ops.timestep_embedding(dev(t)).cpu().numpy(), g["k2_emb"], atol=2e-6)      # vs reference golden
    table = rnd(11, 192, seed=1)
    y, drop = torch.tensor([3, 7, 3, 0]), torch.tensor([0, 1, 0, 0], dtype=torch.uint8)
    out = ops.label_embed_fwd(dev(table), dev(y), dev(drop), 10)
    assert torch.equal(out.cpu(), table[torch.tensor([3, 10, 3, 0])])
    gg = rnd(4, 192, seed=2)
    ref = torch.zeros(11, 192).index_add_(0, torch.tensor([3, 10, 3, 0]), gg)
    assert rel_err(ops.label_embed_bwd(dev(gg), dev(y), dev(drop), 10, 11).cpu(), ref) < 1e-6


def test_adamw_ema_matches_oracle(ops):
    n = 4096 + 64
    p0, g1, g2 = rnd(n, seed=1), rnd(n, seed=2), rnd(n, seed=3)
    sd = {"p": p0.clone()}
    st = otrain.AdamWState(["p"], sd)
    ema = {"p": p0.clone()}
    p, m, v, e = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), dev(p0)
    for step, g in enumerate((g1, g2), 1):
        otrain.adamw_step(sd, {"p": g}, st)
        otrain.ema_update(ema, sd, ["p"])
        ops.adamw_ema(p, dev(g), m, v, e, step, 2e-4, 0.9, 0.95, 1e-8, 0.0, 0.9999)
    np.testing.assert_allclose(p.cpu().numpy(), sd["p"].numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(e.cpu().numpy(), ema["p"].numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(v.cpu().numpy(), st.v["p"].numpy(), rtol=2e-6, atol=1e-12)


# ----------------------------------------------------------------------------- VMAE
@pytest.mark.parametrize("L,ratio", [(1024, 0.75), (1024, 0.25), (256, 0.75), (200, 0.5)])
def test_random_masking_bit_exact(ops, L, ratio):
    N = 5
    noise = torch.rand(N, L, generator=torch.Generator().manual_seed(L))
    noise[1, 10:20] = noise[1, 5]           # ties -> index order
    noise[2] = 0.5                          # all equal
    keep = omae.len_keep(L, ratio)
    ids_keep, mask, ids_restore = ops.random_masking(dev(noise), keep)
    rk, rm, rr = omae.random_masking_ids(noise.numpy(), ratio)
    np.testing.assert_array_equal(ids_restore.cpu().numpy(), rr)
    np.testing.assert_array_equal(mask.cpu().numpy(), rm)
    np.testing.assert_array_equal(ids_keep.cpu().numpy(), rk)
    x = rnd(N, L, 192, seed=1)
    xm = ops.gather_rows(dev(x), ids_keep)
    assert torch.equal(xm.cpu(), torch.gather(x, 1, torch.from_numpy(rk).unsqueeze(-1).expand(-1, -1, 192)))
    back = ops.scatter_rows(xm, ids_keep, L).cpu()
    assert torch.equal(back, x * (1 - torch.from_numpy(rm)).unsqueeze(-1))


@pytest.mark.parametrize("L,ratio,D", [(1024, 0.75, 192), (256, 0.5, 384), (200, 0.25, 512), (64, 0.0, 192)])
def test_restore_tokens_is_the_references_cat_gather_add(ops, L, ratio, D):
    """ldmae_restore_tokens / _bwd against the four tensor ops of the reference's forward_decoder (models_mae.py:536-541: mask_token.repeat,
    cat, gather by ids_restore, + decoder_pos_embed) and their autograd: forward and the kept rows' gradient bit for bit (an add / a copy),
    the mask token's gradient to summation order."""
    B = 3
    keep = omae.len_keep(L, ratio)
    noise = torch.rand(B, L, generator=torch.Generator().manual_seed(L + D))
    _, _, ids_restore = ops.random_masking(dev(noise), keep)
    x = rnd(B, keep, D, seed=1).requires_grad_(True)
    mtok, pos = rnd(1, 1, D, seed=2).requires_grad_(True), rnd(1, L, D, seed=3)
    ids = ids_restore.cpu()
    x_ = torch.cat([x, mtok.repeat(B, L - keep, 1)], dim=1)
    ref = torch.gather(x_, 1, ids.unsqueeze(-1).repeat(1, 1, D)) + pos
    g = rnd(B, L, D, seed=4)
    ref.backward(g)
    out = ops.restore_tokens(dev(x.detach()), dev(mtok.detach().reshape(-1)), dev(pos[0]), ids_restore)
    assert torch.equal(out.cpu(), ref.detach())
    dx, dm = ops.restore_tokens_bwd(dev(g), ids_restore, keep)
    assert torch.equal(dx.cpu(), x.grad)
    if keep < L:
        assert rel_err(dm.cpu(), mtok.grad.reshape(-1)) < 1e-5
    else:
        assert float(dm.abs().max()) == 0.0
    dx2, dm2 = ops.restore_tokens_bwd(dev(g), ids_restore, keep, need_mask_grad=False)
    assert dm2 is None and torch.equal(dx2, dx)


def test_random_masking_golden(ops, golden):
    g = golden("mae")
    for tag, ratio in (("75", 0.75), ("25", 0.25)):
        _, mask, ids = ops.random_masking(dev(torch.from_numpy(g["mae_noise"])), omae.len_keep(1024, ratio))
        np.testing.assert_array_equal(mask.cpu().numpy(), g[f"mae{tag}_mask"])
        np.testing.assert_array_equal(ids.cpu().numpy(), g[f"mae{tag}_ids_restore"])


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_layernorm_gelu(ops, dtype):
    M, D = 300, 192
    x = rnd(M, D, seed=1).requires_grad_(True)
    w, b = (1 + 0.1 * rnd(D, seed=2)).requires_grad_(True), (0.1 * rnd(D, seed=3)).requires_grad_(True)
    ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6)
    g = q(rnd(M, D, seed=4), dtype)
    ref.backward(g)
    out, mean, rstd = ops.layernorm_fwd(dev(x.detach()), dev(w.detach()), dev(b.detach()), dtype)
    assert rel_err(out.float().cpu(), ref.detach()) < (1e-5 if dtype == F32 else 1e-2)
    dx = torch.zeros(M, D, device="cuda")
    dw, db = ops.layernorm_bwd(dev(g, dtype), dev(x.detach()), dev(w.detach()), mean, rstd, dx)
    assert rel_err(dx.cpu(), x.grad) < 1e-4 and rel_err(dw.cpu(), w.grad) < 1e-4 and rel_err(db.cpu(), b.grad) < 1e-4
    if dtype != F32:       # the second output of the block backward: the UPDATED residual gradient rounded to the activation type, bit for bit a cast
        acc = dev(rnd(M, D, seed=6))
        ref_acc = acc.clone()
        ops.layernorm_bwd(dev(g, dtype), dev(x.detach()), dev(w.detach()), mean, rstd, ref_acc)
        dw2, db2, dxc = ops.layernorm_bwd(dev(g, dtype), dev(x.detach()), dev(w.detach()), mean, rstd, acc, cast=True)
        assert torch.equal(acc, ref_acc) and torch.equal(dxc, ops.cast(acc, dtype)) and torch.equal(dw2, dw) and torch.equal(db2, db)
    pre = q(rnd(M, D, seed=5), dtype).requires_grad_(True)
    torch.nn.functional.gelu(pre).backward(g)
    assert rel_err(ops.gelu_bwd(dev(g, dtype), dev(pre.detach(), dtype)).float().cpu(), pre.grad) < (1e-5 if dtype == F32 else 1e-2)


@pytest.mark.parametrize("shape", [(3, 3, 64, 64), (2, 3, 17, 20), (2, 3, 9, 10)])
def test_conv3x3_rgb_and_its_backward(ops, shape):
    """The RGB smoothing convolution (models_mae.py:254,275) and its backward: the four-pixels-per-thread kernels (row length % 4 == 0) and the
    generic ones (the last shape) against torch's conv2d in f64."""
    B, C, Hh, Ww = shape
    x = rnd(*shape, seed=1).double().requires_grad_(True)
    w = (0.3 * rnd(C, C, 3, 3, seed=2)).double().requires_grad_(True)
    b = (0.1 * rnd(C, seed=3)).double().requires_grad_(True)
    g = rnd(*shape, seed=4).double()
    ref = torch.nn.functional.conv2d(x, w, b, padding=1)
    ref.backward(g)
    out = ops.conv3x3(dev(x.detach().float()), dev(w.detach().float()), dev(b.detach().float()))
    assert rel_err(out.cpu(), ref.detach()) < 1e-6
    dx, dw, db = ops.conv3x3_bwd(dev(g.float()), dev(x.detach().float()), dev(w.detach().float()))
    assert rel_err(dx.cpu(), x.grad) < 1e-6 and rel_err(dw.cpu(), w.grad) < 1e-5 and rel_err(db.cpu(), b.grad) < 1e-5


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("hd", [8, 24, 80])
def test_attention_head_dims_outside_the_instantiated_set(ops, dtype, hd):
    """Head dims the kernels are not instantiated for (24: mae_for_ldmae_f8d16_prev_large, 80: mae_vit_huge, 8: the 128-wide decoder) run the next
    larger kernel on zero-padded copies -- exact, since zero columns add nothing to q . k and give zero output columns.  Head-major and packed
    entry points, forward and backward, against f64 softmax attention."""
    if dtype == F32 and hd == 80:
        with pytest.raises(RuntimeError, match="160 KiB of LDS"):                # the f32 BACKWARD stops at head_dim 72 (LDS): loud, not wrong
            z = torch.zeros(1, 1, 64, 128, device="cuda")
            ops.attention_bwd(z, z, z, torch.zeros(1, 64, 128, device="cuda"), torch.zeros(1, 64, 128, device="cuda"), torch.zeros(1, 1, 64, device="cuda"), 1.0)
        return
    B, H, N = 2, 3, 200
    g = torch.Generator().manual_seed(hd)
    qkv = q(torch.randn(B * N, 3 * H * hd, generator=g), dtype)
    do = q(torch.randn(B, N, H * hd, generator=g), dtype)
    scale = hd ** -0.5
    qq, kk, vv = (qkv.view(B, N, 3, H, hd)[:, :, i].permute(0, 2, 1, 3).double().requires_grad_(True) for i in range(3))
    ro = (torch.softmax((qq @ kk.transpose(-1, -2)) * scale, -1) @ vv).permute(0, 2, 1, 3).reshape(B, N, H * hd)
    (ro * do.double()).sum().backward()
    tol = 1e-5 if dtype == F32 else 2e-2
    qh, kh, vh = (dev(t.detach(), dtype).contiguous() for t in (qq, kk, vv))
    o, lse = ops.attention_fwd(qh, kh, vh, scale)
    assert o.shape == (B, N, H * hd) and rel_err(o.float().cpu(), ro.detach()) < tol
    for got, ref in zip(ops.attention_bwd(qh, kh, vh, o, dev(do, dtype), lse, scale), (qq, kk, vv)):
        assert got.shape == ref.shape and rel_err(got.float().cpu(), ref.grad) < tol
    if dtype == BF16:                                                            # the packed form the ViT blocks call
        o2, lse2 = ops.attention_fwd_qkv(dev(qkv, dtype), B, N, H, hd, scale)
        assert torch.equal(o2, o)
        dqkv = ops.attention_bwd_qkv(dev(qkv, dtype), o2, dev(do, dtype), lse2, B, N, H, hd, scale).view(B, N, 3, H, hd)
        for i, ref in enumerate((qq, kk, vv)):
            assert rel_err(dqkv[:, :, i].permute(0, 2, 1, 3).float().cpu(), ref.grad) < tol


def test_mae_loss_in_image_space(ops):
    """forward_loss (models_mae.py:733-754: patchify the target, per-patch mean of the squared error, means over masked / visible patches) against the
    image-space kernels: values and the gradient with respect to the predicted image, f64 reference."""
    B, p, g = 3, 8, 5
    img = rnd(B, 3, g * p, g * p, seed=1).double()
    pred = rnd(B, 3, g * p, g * p, seed=2).double().requires_grad_(True)
    mask = (rnd(B, g * g, seed=3) > 0.3).double()
    patch = lambda x: torch.einsum('nchpwq->nhwpqc', x.reshape(B, 3, g, p, g, p)).reshape(B, g * g, p * p * 3)      # noqa: E731
    loss = ((patch(pred) - patch(img)) ** 2).mean(-1)
    ml, vl = (loss * mask).sum() / mask.sum(), (loss * (1 - mask)).sum() / (1 - mask).sum()
    (0.7 * ml + 0.3 * vl).backward()
    sums = ops.mae_loss_fwd(dev(pred.detach().float()), dev(img.float()), dev(mask.float()), p).cpu().double()
    P = p * p * 3
    assert abs(float(sums[0] / (mask.sum() * P)) - float(ml)) < 1e-6 * float(ml) and abs(float(sums[1] / ((1 - mask).sum() * P)) - float(vl)) < 1e-6 * float(vl)
    coef = torch.tensor([0.7 / float(mask.sum() * P), 0.3 / float((1 - mask).sum() * P)], device="cuda")
    d = ops.mae_loss_bwd(dev(pred.detach().float()), dev(img.float()), dev(mask.float()), coef, p)
    assert rel_err(d.cpu(), pred.grad) < 1e-6


def test_errors_are_loud(ops):
    a = torch.zeros(128, 100, device="cuda", dtype=BF16)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.gemm_nt(a, a)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm_nt(torch.zeros(4, 16), torch.zeros(4, 16))


@pytest.mark.parametrize("K", [16, 32])
def test_thin_products_vs_float64(ops, K):
    """ldmae_thin_nt / ldmae_thin_tn (contraction 16 / 32: the DiT patch embedding at patch 1 and the final layer's dX) against float64,
    incl. bias, the position table, bf16 output, a ragged row count and bitwise run-to-run equality of the chunked reduction."""
    g = torch.Generator(device="cuda").manual_seed(K)
    M, N, Tn = 2 * 520 + 8, 772, 520                      # not a multiple of the 512-row chunk; N % 4 == 0 only
    t, w, b = (torch.randn(*s, device="cuda", generator=g) for s in ((M, K), (N, K), (N,)))
    pos = torch.randn(Tn + 4, N, device="cuda", generator=g)[:Tn]
    ref = t.double() @ w.double().t() + b.double()
    out = ops.thin_nt(t, w, b)
    assert float((out.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    rows = torch.arange(M, device="cuda") % Tn
    outp = ops.thin_nt(t, w, b, pos, Tn)
    assert float((outp.double() - (ref + pos.double()[rows])).abs().max()) < 1e-4 * float(ref.abs().max())
    outb = ops.thin_nt(t, w, None, None, 0, torch.bfloat16)
    assert outb.dtype == torch.bfloat16 and float((outb.double() - (ref - b.double())).abs().max()) < 1e-2 * float(ref.abs().max())
    gr = torch.randn(M, N, device="cuda", generator=g)
    dW, db = ops.thin_tn(gr, t)
    rW, rb = gr.double().t() @ t.double(), gr.double().sum(0)
    assert float((dW.double() - rW).abs().max()) < 1e-4 * float(rW.abs().max()) and float((db.double() - rb).abs().max()) < 1e-4 * float(rb.abs().max())
    dW2, db2 = ops.thin_tn(gr, t)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)
    assert not ops.thin_ok(770, 16) and not ops.thin_ok(768, 24)


def test_round3_entry_points_reject_bad_arguments(ops):
    """Error behaviour of the entry points added in round 3: a negative status + message (RuntimeError through the wrappers), never a launch."""
    f = lambda *s: torch.randn(*s, device="cuda")
    with pytest.raises(RuntimeError, match="K=24"):
        ops.thin_nt(f(64, 24), f(32, 24))
    with pytest.raises(RuntimeError, match="multiple of 4"):
        ops.thin_nt(f(64, 16), f(30, 16))
    with pytest.raises(RuntimeError, match="K=8"):
        ops.thin_tn(f(64, 32), f(64, 8))
    with pytest.raises(RuntimeError, match="count=65"):
        ops.cast_stack([f(8, 8) for _ in range(65)], torch.bfloat16)
    with pytest.raises(RuntimeError, match="one size"):
        ops.cast_stack([f(8, 8), f(8, 16)], torch.bfloat16)
    with pytest.raises(RuntimeError, match="equal size"):
        ops.multi_add_([f(16)], [f(8)])
    with pytest.raises(RuntimeError, match="H\\*W=6"):
        ops.latent_prologue(f(2, 8, 2, 3), f(2, 4, 2, 3))
    # and the happy paths of the two small helpers
    a, b = [f(5), f(1000, 3)], [f(5), f(1000, 3)]
    want = [x + y for x, y in zip(a, b)]
    ops.multi_add_(a, b)
    assert all(torch.equal(x, w) for x, w in zip(a, want))
    ws = [f(16, 24) for _ in range(3)]
    st = ops.cast_stack(ws, torch.bfloat16)
    assert st.shape == (48, 24) and torch.equal(st, torch.cat(ws).to(torch.bfloat16))


def test_thin_products_at_the_bench_shape_match_the_generic_gemms(ops):
    """At the real patch-embedding shape (M = 256 x 1024 rows, N = 768, K = 16) the thin kernels against the generic f32 MFMA GEMMs they
    replaced on this shape (same inputs; f32 summation order differs: 1e-5)."""
    g = torch.Generator(device="cuda").manual_seed(3)
    M, N, K, T = 256 * 1024, 768, 16, 1024
    t, w, b, pos = (torch.randn(*s, device="cuda", generator=g) for s in ((M, K), (N, K), (N,), (T, N)))
    a, ref = ops.thin_nt(t, w, b, pos, T), ops.gemm_nt_pos(t, w, b, pos, T)
    assert float((a - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    gr = torch.randn(M, N, device="cuda", generator=g)
    dW, db = ops.thin_tn(gr, t)
    rW, rb = ops.gemm_tn(gr, t), ops.colsum(gr)
    assert float((dW - rW).abs().max()) < 1e-5 * float(rW.abs().max()) and float((db - rb).abs().max()) < 1e-5 * float(rb.abs().max())
    wt = torch.randn(N, K, device="cuda", generator=g)            # final layer dX: [M, 16] @ [16, 768] -> bf16
    dx, rx = ops.thin_nt(t, wt, out_dtype=torch.bfloat16), ops.gemm_nt(t, wt, out_dtype=torch.bfloat16)
    assert float((dx.float() - rx.float()).abs().max()) <= 2 ** -7 * float(rx.float().abs().max())


@pytest.mark.parametrize("R,C", [(768, 2304), (4096, 768), (192, 576), (100, 36), (64, 64)])
def test_cast_weight_straight_and_transposed(ops, R, C):
    """ldmae_cast_weight: bf16 / f32 copies of an f32 master weight, straight and transposed, on the 64x64-tile path (sides % 64 == 0) and
    the 32x32 fallback -- exact (one rounding per element)."""
    w = torch.randn(R, C, device="cuda", generator=torch.Generator(device="cuda").manual_seed(R + C))
    for dtype in (torch.bfloat16, torch.float32):
        a, at = ops.cast_weight(w, dtype, transposed=True, straight=True)
        assert torch.equal(a, w.to(dtype)) and torch.equal(at, w.t().contiguous().to(dtype))
        _, only_t = ops.cast_weight(w, dtype, transposed=True, straight=False)
        assert torch.equal(only_t, at)


# ----------------------------------------------------------------------------- the reference's real-width kernel goldens, straight into the HIP kernels
def test_reference_kernel_goldens_through_the_hip_kernels(ops, golden):
    """tests/golden/kernels.npz holds what the REFERENCE's own modules return at the real B/1 width (make_golden.py: RMSNorm + modulate at
    768, feat_rope on [1, 2, 1024, 64], SwiGLUFFN 768 -> 2048 -> 768, block 0's Attention at 1024 tokens).  test_oracle_golden.py pins the
    oracle to them; here the same vectors go DIRECTLY through the HIP kernels (C ABI), f32 at 1e-4 and bf16 at 2e-2."""
    from weights import det_randn, det_weights
    from ldmae_amd.models.lightningdit import Attention
    from ldmae_amd.models.pos_embed import VisionRotaryEmbeddingFast
    from ldmae_amd.models.swiglu_ffn import SwiGLUFFN
    g = golden("kernels")
    # k5: RMSNorm(768) + modulate (rmsnorm.py:51-77, lightningdit.py modulate)
    x = det_randn("k5_x", (2, 8, 768), 3)
    w = 1 + 0.1 * det_randn("k5_w", (768,), 3)
    sh, sc = 0.3 * det_randn("k5_sh", (2, 768), 3), 0.3 * det_randn("k5_sc", (2, 768), 3)
    for dtype in (F32, BF16):
        out, _ = ops.rmsnorm_modulate_fwd(dev(x.view(16, 768)), dev(w), dev(sh), dev(sc), 8, dtype)
        assert rel_err(out.float().cpu().view(2, 8, 768), g["k5_out"]) < TOL[dtype], dtype
    # k8: 2-D RoPE on [1, 2, 1024, 64] (pos_embed.py:96-133): head / tail rows and, in f32, the reference's own hash if the bits agree
    rope = VisionRotaryEmbeddingFast(32, pt_seq_len=32).cuda()
    q8 = det_randn("k8_q", (1, 2, 1024, 64), 3)
    rq = rope(dev(q8)).cpu()
    assert rel_err(rq[0, :, :4], g["k8_head"]) < 1e-5 and rel_err(rq[0, :, -4:], g["k8_tail"]) < 1e-5
    rq16 = rope(dev(q8, BF16)).float().cpu()
    assert rel_err(rq16[0, :, :4], g["k8_head"]) < TOL[BF16]
    # k11: SwiGLUFFN (swiglu_ffn.py:15-36) with the golden's weights on 8 rows
    sdf = det_weights({"w12.weight": (4096, 768), "w12.bias": (4096,), "w3.weight": (768, 2048), "w3.bias": (768,)}, 4)
    f = SwiGLUFFN(768, 2048).cuda()
    f.load_state_dict(sdf)
    x11 = det_randn("k11_x", (8, 768), 3)
    with torch.no_grad():
        assert rel_err(f(dev(x11)).cpu(), g["k11_out"]) < TOL[F32]
        assert rel_err(f(dev(x11, BF16)).float().cpu(), g["k11_out"]) < TOL[BF16]
    # k9: block 0's Attention (lightningdit.py:32-91: qkv -> QK-RMSNorm -> RoPE -> softmax attention -> proj) on one sample, 1024 tokens
    sda = det_weights({"qkv.weight": (2304, 768), "qkv.bias": (2304,), "q_norm.weight": (64,), "k_norm.weight": (64,),
                       "proj.weight": (768, 768), "proj.bias": (768,)}, 6)
    a = Attention(768, num_heads=12, qkv_bias=True, qk_norm=True, use_rmsnorm=True).cuda()
    a.load_state_dict(sda)
    x9 = det_randn("k9_x", (1, 1024, 768), 3) * 0.5
    with torch.no_grad():
        ao = a(dev(x9), rope=rope).cpu()
        assert rel_err(ao[0, :4], g["k9_head"]) < TOL[F32] and rel_err(ao[0, -4:], g["k9_tail"]) < TOL[F32]
        assert abs(float(ao.double().norm()) - float(g["k9_norm"])) < 1e-4 * float(g["k9_norm"])
        a.precision = BF16
        ao16 = a(dev(x9), rope=rope).float().cpu()
        assert rel_err(ao16[0, :4], g["k9_head"]) < TOL[BF16] and rel_err(ao16[0, -4:], g["k9_tail"]) < TOL[BF16]


@pytest.mark.parametrize("M,N,K", [(256 * 41 + 104, 2304 + 64, 192), (8, 768, 64), (264, 264, 128), (16384, 768, 768), (2048 + 8, 1152, 1152)])
def test_whole_line_nt_kernel_is_bitwise_equal_to_the_half_line_kernel(ops, M, N, K):
    """gemm_nt_lines_kernel (128-B LDS rows, seamless five-slot ring; the default for these shapes) against gemm_nt_persist_kernel (per-call flag
    LDMAE_EPI_HALF_LINES): the same products in the same order -> the same bits, for every fused epilogue, ragged row and column tiles
    (M, N multiples of 8 only), one / two / many 64-deep blocks per tile, persistent and one-tile-per-workgroup launches."""
    g = torch.Generator(device="cuda").manual_seed(11)
    rb = lambda *s: torch.randn(*s, device="cuda", generator=g).to(BF16)
    rf = lambda *s: torch.randn(*s, device="cuda", generator=g)
    a, w, bias = rb(M, K), rb(N, K) * K ** -0.5, rf(N)
    T = 8 if M % 8 == 0 and M >= 8 else 1
    xin, gate, pos = rf(M, N), rf(M // T, N), rf(T, N)
    cases = {
        "bias": lambda: ops.gemm_nt(a, w, bias),
        "bias_f32out": lambda: ops.gemm_nt(a, w, bias, out_dtype=F32),
        "gate_res": lambda: ops.gemm_nt_gate_res(a, w, bias, xin, gate, T),
        "gelu": lambda: ops.gemm_nt_gelu(a, w, bias, save_pre=True),
        "pos": lambda: ops.gemm_nt_pos(a, w, bias, pos, T),
    }
    pre = rb(M, N)
    cases["gelu_bwd"] = lambda: ops.gemm_nt_gelu_bwd(a, w, pre)
    if N % 256 == 0 and N >= 512:
        cases["swiglu"] = lambda: ops.gemm_nt_swiglu(a, w, bias)
    h12 = rb(M, 2 * N)
    cases["swiglu_bwd"] = lambda: ops.gemm_nt_swiglu_bwd(a, w, h12, with_bias=True)
    flat = lambda o: [t for t in (o if isinstance(o, (tuple, list)) else (o,)) if torch.is_tensor(t)]
    try:
        for mode in ("persistent", "tile"):
            ops.set_gemm_launch_mode(mode)
            for name, fn in cases.items():
                ops.set_gemm_half_lines(True)
                ref = [t.clone() for t in flat(fn())]
                ops.set_gemm_half_lines(False)
                got = flat(fn())
                assert len(ref) == len(got) and all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(ref, got)), (name, mode)
        # and against the arithmetic, once
        ops.set_gemm_half_lines(False)
        assert rel_err(ops.gemm_nt(a, w, bias).float().cpu(), (a.float() @ w.float().T + bias).cpu()) < TOL[BF16]
    finally:
        ops.set_gemm_half_lines(False)
        ops.set_gemm_launch_mode("persistent")


# ----------------------------------------------------------------------------- the fp16 kernel family (LDMAE_F16), kernel by kernel
F16 = torch.float16


@pytest.mark.parametrize("M,N,K", [(8192, 576, 192), (4104, 768, 768), (264, 192, 64)])
def test_f16_gemm_nt_epilogues(ops, M, N, K):
    """fp16 operands (v_mfma_f32_16x16x32_f16, f32 accumulation) against f64 products of the same fp16-rounded data: bias (fp16 and f32 out),
    exact-erf GELU with its pre-activation copy, gated residual adding the UNROUNDED product (the TF32-class form) and the rounded one (autocast),
    the GELU-backward gradient GEMM.  Forward outputs saturate at 65504, gradient outputs overflow to infinity (LDMAE_EPI_F16_INF)."""
    a, w, bias = dev(rnd(M, K, seed=1), F16), dev(rnd(N, K, seed=2, scale=K ** -0.5), F16), dev(rnd(N, seed=3))
    ref = a.double().cpu() @ w.double().cpu().T + bias.double().cpu()
    out = ops.gemm_nt(a, w, bias)
    assert out.dtype == F16 and rel_err(out.float().cpu(), ref) < 1e-3
    assert rel_err(ops.gemm_nt(a, w, bias, out_dtype=F32).cpu(), ref) < 2e-5
    act, pre = ops.gemm_nt_gelu(a, w, bias, save_pre=True)
    assert rel_err(pre.float().cpu(), ref) < 1e-3 and rel_err(act.float().cpu(), torch.nn.functional.gelu(ref)) < 1e-3
    T = 8
    xin, gate = dev(rnd(M, N, seed=4)), dev(rnd(M // T, N, seed=5))
    xo, _ = ops.gemm_nt_gate_res(a, w, bias, xin, gate, T, save_y=False, y_dtype=F32)
    assert rel_err(xo.cpu(), xin.double().cpu() + gate.double().cpu().repeat_interleave(T, 0) * ref) < 2e-5
    xo16, y16 = ops.gemm_nt_gate_res(a, w, bias, xin, gate, T, save_y=True)
    assert y16.dtype == F16 and torch.equal(y16, out)
    assert rel_err(xo16.cpu(), torch.addcmul(xin, gate.repeat_interleave(T, 0), y16.float()).cpu()) < 1e-6
    dy = dev(rnd(M, K, seed=6), F16)
    got = ops.gemm_nt_gelu_bwd(dy, w, pre)                       # [M, N] = (dy @ w^T) * gelu'(pre)
    v = pre.double().cpu()
    dgelu = 0.5 * (1 + torch.erf(v / 2 ** 0.5)) + v * torch.exp(-0.5 * v * v) / (2 * torch.pi) ** 0.5
    assert rel_err(got.float().cpu(), (dy.double().cpu() @ w.double().cpu().T).half().double() * dgelu) < 2e-3
    # overflow: forward saturates, a gradient GEMM gives infinity
    ab, wb = (a.float().abs() * 1e3).to(F16), (w.float().abs() * 40).to(F16)      # finite operands, products far beyond 65504
    assert torch.isfinite(ab).all() and torch.isfinite(wb).all()
    big = ops.gemm_nt(ab, wb, None)
    assert torch.isfinite(big).all() and float(big.float().abs().max()) == 65504.0
    bigg = ops.gemm_nt(ab, wb, None, grad=True)
    assert torch.isinf(bigg).any() and not torch.isnan(bigg).any()


@pytest.mark.parametrize("M,N,K", [(4096, 192, 768), (8192, 576, 192), (51 * 8, 192, 192), (16384, 768, 192)])
def test_f16_gemm_tn(ops, M, N, K):
    a, b = dev(rnd(M, N, seed=1), F16), dev(rnd(M, K, seed=2), F16)
    ref = a.double().cpu().T @ b.double().cpu()
    out, db = ops.gemm_tn(a, b, with_bias=True)
    assert out.dtype == F32 and rel_err(out.cpu(), ref) < 1e-5 and rel_err(db.cpu(), a.double().cpu().sum(0)) < 1e-5
    out2 = ops.gemm_tn(a, b, out=out.clone(), beta=1.0)
    assert rel_err(out2.cpu(), 2 * ref) < 1e-5 and torch.equal(ops.gemm_tn(a, b), out)


@pytest.mark.parametrize("H,N", [(12, 256), (3, 200), (12, 1024)])
def test_f16_attention_fwd_bwd_head_dim_16(ops, H, N):
    """The head-dim-16 flash kernels on fp16 operands (packed qkv, ragged N too) against f64 softmax attention of the same fp16-rounded data:
    fp16's 11-bit mantissa gives 8x the accuracy of the bf16 instantiation."""
    B, hd = 2, 16
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B * N, 3 * H * hd, generator=g).to(F16).cuda()
    do = torch.randn(B, N, H * hd, generator=g).to(F16).cuda()
    scale = hd ** -0.5
    o, lse = ops.attention_fwd_qkv(qkv, B, N, H, hd, scale)
    q, k, v = (qkv.view(B, N, 3, H, hd)[:, :, i].permute(0, 2, 1, 3).double().cpu().requires_grad_(True) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * scale
    ro = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, H * hd)
    assert o.dtype == F16 and rel_err(o.float().cpu(), ro.detach()) < 2e-3
    assert (lse.cpu() - torch.logsumexp(s, -1).detach().float()).abs().max() < 5e-3
    (ro * do.double().cpu()).sum().backward()
    dqkv = ops.attention_bwd_qkv(qkv, o, do, lse, B, N, H, hd, scale).view(B, N, 3, H, hd)
    for i, t in enumerate((q, k, v)):
        assert rel_err(dqkv[:, :, i].permute(0, 2, 1, 3).float().cpu(), t.grad) < 4e-3, i
    o16, _ = ops.attention_fwd_qkv(qkv.to(BF16), B, N, H, hd, scale)
    assert rel_err(o.float().cpu(), ro.detach()) < 0.5 * rel_err(o16.float().cpu(), ro.detach())      # really the fp16 instantiation


def test_f16_layernorm_cast_colsum(ops):
    x = dev(rnd(4096, 192, seed=1) * 3 + 0.5)
    w, b = dev(1 + 0.1 * rnd(192, seed=2)), dev(0.1 * rnd(192, seed=3))
    y, mu, rs = ops.layernorm_fwd(x, w, b, F16)
    ref = torch.nn.functional.layer_norm(x.double().cpu(), (192,), w.double().cpu(), b.double().cpu(), 1e-6)
    assert y.dtype == F16 and rel_err(y.float().cpu(), ref) < 5e-4
    assert torch.equal(ops.cast(x, F16), x.half()) and torch.equal(ops.cast(x.half(), F32), x.half().float())
    wt, wtt = ops.cast_weight(dev(rnd(192, 576, seed=4)), F16)
    assert torch.equal(wt, dev(rnd(192, 576, seed=4)).half()) and torch.equal(wtt, wt.t().contiguous())
    assert torch.isinf(ops.cast(x * 1e6, F16)).any()                          # plain casts overflow to infinity (gradients under a loss scaler)
    assert rel_err(ops.colsum(y).cpu(), y.double().cpu().sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("M,D,rpb,shift", [(128, 768, 64, True), (96, 192, 32, True), (64, 1152, 16, False)])
def test_layernorm_modulate_without_affine(ops, dtype, M, D, rpb, shift):
    """ldmae_layernorm_modulate_fwd / _bwd / _bwd_gate (rmsnorm_modulate_* with w=None): nn.LayerNorm(D, elementwise_affine=False, eps=1e-6)
    + modulate, the norm of a block built with use_rmsnorm=False (lightningdit.py:200-201,26-30), against torch autograd; the gate-fused form
    against the unfused pair bit for bit."""
    B = M // rpb
    x = rnd(M, D, seed=1).add(0.3).requires_grad_(True)                 # a non-zero row mean: the centring matters
    mod = (0.2 * rnd(B, 3 * D, seed=2)).requires_grad_(True)
    sh, sc = (mod[:, :D] if shift else None), mod[:, D:2 * D]
    xn = torch.nn.functional.layer_norm(x, (D,), None, None, 1e-6).view(B, rpb, D)
    ref = (xn * (1 + sc.unsqueeze(1)) + (sh.unsqueeze(1) if shift else 0)).view(M, D)
    g = q(rnd(M, D, seed=3), dtype)
    ref.backward(g)
    modd = dev(mod.detach())
    shd, scd = (modd[:, :D] if shift else None), modd[:, D:2 * D]
    out, rstd = ops.rmsnorm_modulate_fwd(dev(x.detach()), None, shd, scd, rpb, dtype, 1e-6)
    assert rel_err(out.float().cpu(), ref.detach()) < (1e-5 if dtype == F32 else 1e-2)
    dmod = torch.zeros(B, 3 * D, device="cuda")
    dx = dev(rnd(M, D, seed=4))
    dx0 = dx.clone()
    dw = ops.rmsnorm_modulate_bwd(dev(g, dtype), dev(x.detach()), None, scd, rstd, dx, dmod[:, :D] if shift else None, dmod[:, D:2 * D], rpb)
    assert dw is None
    assert rel_err((dx - dx0).cpu(), x.grad) < 1e-4
    assert rel_err(dmod[:, D:2 * D].cpu(), mod.grad[:, D:2 * D]) < 1e-4
    if shift:
        assert rel_err(dmod[:, :D].cpu(), mod.grad[:, :D]) < 1e-4
    # the gate-fused form == the unfused pair (rmsnorm_modulate_bwd, then gate_bwd of the updated dx)
    y, gate = dev(q(rnd(M, D, seed=5), dtype), dtype), modd[:, 2 * D:]
    dxa, dxb = dx0.clone(), dx0.clone()
    dma, dmb = torch.zeros_like(dmod), torch.zeros_like(dmod)
    ops.rmsnorm_modulate_bwd(dev(g, dtype), dev(x.detach()), None, scd, rstd, dxa, dma[:, :D] if shift else None, dma[:, D:2 * D], rpb)
    dya, dba = ops.gate_bwd(dxa, y, gate, dma[:, 2 * D:], rpb, dtype, with_bias=True)
    _, dyb, dbb = ops.rmsnorm_modulate_bwd_gate(dev(g, dtype), dev(x.detach()), None, scd, rstd, dxb, dmb[:, :D] if shift else None, dmb[:, D:2 * D],
                                                y, gate, dmb[:, 2 * D:], rpb, dtype)
    assert torch.equal(dxa, dxb) and torch.equal(dya, dyb) and torch.equal(dma, dmb) and torch.equal(dba, dbb)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gelu_tanh(ops, dtype):
    """nn.GELU(approximate="tanh") and its backward (the timm Mlp of a use_swiglu=False block, lightningdit.py:208,219-224)."""
    x = (3 * rnd(300, 512, seed=1)).requires_grad_(True)
    xq = q(x.detach(), dtype).requires_grad_(True)
    ref = torch.nn.functional.gelu(xq, approximate="tanh")
    g = q(rnd(300, 512, seed=2), dtype)
    ref.backward(g)
    out = ops.gelu_tanh_fwd(dev(xq.detach(), dtype))
    dx = ops.gelu_tanh_bwd(dev(g, dtype), dev(xq.detach(), dtype))
    tol = 1e-6 if dtype == F32 else 1e-2
    assert rel_err(out.float().cpu(), ref.detach()) < tol and rel_err(dx.float().cpu(), xq.grad) < tol
    big = torch.tensor([-40.0, -12.0, 12.0, 40.0, 0.0])
    ob = ops.gelu_tanh_fwd(dev(big))
    assert torch.isfinite(ob).all() and rel_err(ob.cpu(), torch.nn.functional.gelu(big, approximate="tanh")) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,H,K", [(2, 128, 4, 192), (1, 256, 12, 768), (4, 128, 8, 64), (256, 1024, 12, 768)])      # last: BASELINE config 2's block, full size
@pytest.mark.parametrize("with_norm", [True, False])
def test_qkv_gemm_with_fused_qknorm_rope_is_bitwise_the_pair(B, N, H, K, with_norm):
    """ldmae_gemm_nt_qkv_rope (lightningdit.py:68-74 in one kernel: the qkv Linear with q_norm / k_norm / RoPE in the GEMM epilogue) against the pair it
    replaces, ldmae_gemm_nt + ldmae_qknorm_rope_fwd: the packed qkv, q2 and k2 must be BITWISE equal (same arithmetic on the same bf16-rounded values);
    with_norm=False: the RoPE-only form of use_qknorm=False.  Forward-only calls (store_raw_qk=False) skip the pre-norm q / k thirds and nothing else."""
    from ldmae_amd import ops
    hd = 64
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + N + H)
    a = torch.randn(B * N, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(3 * H * hd, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(3 * H * hd, device="cuda", generator=g) * 0.1
    wq = 1 + 0.2 * torch.randn(hd, device="cuda", generator=g) if with_norm else None
    wk = 1 + 0.2 * torch.randn(hd, device="cuda", generator=g) if with_norm else None
    ang = torch.rand(N, hd // 2, device="cuda", generator=g) * 6.28
    cos, sin = ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous()
    assert ops.gemm_nt_qkv_rope_ok(a, w, B, N, H, hd)
    ref = ops.gemm_nt(a, w, bias)
    q_ref, k_ref, _ = ops.qknorm_rope_fwd(ref, wq, wk, cos, sin, B, N, H, hd, 1e-6, copy_v=False)
    qkv, q2, k2 = ops.gemm_nt_qkv_rope(a, w, bias, wq, wk, cos, sin, B, N, H, hd, 1e-6)
    assert torch.equal(qkv, ref) and torch.equal(q2, q_ref) and torch.equal(k2, k_ref)
    if B * N > 65536:
        return                                      # full size: the bitwise statement above is the test (the f32 torch check below would need 10 GB)
    # against plain f32 torch as well (the pair is itself pinned by the reference goldens: test_qknorm_rope...)
    x = (a.float() @ w.float().t() + bias).to(torch.bfloat16).float().view(B, N, 3, H, hd)
    def front(t, wn):
        if wn is not None:
            t = t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6) * wn
        rot = torch.stack((-t[..., 1::2], t[..., 0::2]), -1).flatten(-2)
        return (t * cos[None, :, None, :] + rot * sin[None, :, None, :]).permute(0, 2, 1, 3)
    assert (q2.float() - front(x[:, :, 0], wq)).abs().max() < 0.04 and (k2.float() - front(x[:, :, 1], wk)).abs().max() < 0.04
    qkv_f, q2_f, k2_f = ops.gemm_nt_qkv_rope(a, w, bias, wq, wk, cos, sin, B, N, H, hd, 1e-6, store_raw_qk=False)
    assert torch.equal(q2_f, q_ref) and torch.equal(k2_f, k_ref) and torch.equal(qkv_f.view(B * N, 3, H * hd)[:, 2], ref.view(B * N, 3, H * hd)[:, 2])
    # without a bias, and one tile per workgroup
    ref0 = ops.gemm_nt(a, w, None)
    q0, k0, _ = ops.qknorm_rope_fwd(ref0, wq, wk, cos, sin, B, N, H, hd, 1e-6, copy_v=False)
    ops.set_gemm_launch_mode("tile")
    try:
        r0 = ops.gemm_nt_qkv_rope(a, w, None, wq, wk, cos, sin, B, N, H, hd, 1e-6)
    finally:
        ops.set_gemm_launch_mode("auto")
    assert torch.equal(r0[0], ref0) and torch.equal(r0[1], q0) and torch.equal(r0[2], k0)


@pytest.mark.gpu
def test_qkv_rope_fused_gemm_refuses_what_it_does_not_cover():
    from ldmae_amd import ops
    a = torch.zeros(192, 64, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(3 * 4 * 64, 64, device="cuda", dtype=torch.bfloat16)
    assert not ops.gemm_nt_qkv_rope_ok(a, w, 3, 64, 4, 64)                       # 64 tokens per sample: a wave's 128 rows would straddle samples
    assert not ops.gemm_nt_qkv_rope_ok(a.float(), w.float(), 1, 256, 4, 64)      # f32
    assert not ops.gemm_nt_qkv_rope_ok(torch.zeros(256, 64, device="cuda", dtype=torch.bfloat16), torch.zeros(3 * 4 * 72, 64, device="cuda", dtype=torch.bfloat16), 1, 256, 4, 72)
    t = torch.zeros(64, 64, device="cuda")
    with pytest.raises(RuntimeError, match="qkv_rope"):
        ops.gemm_nt_qkv_rope(a, w, None, None, None, t, t, 3, 64, 4, 64)
