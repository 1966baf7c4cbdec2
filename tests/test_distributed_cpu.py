"""world_size-2 gloo test of the data-parallel gradient path (the N>1 path of bench.py / the train driver): bucketed
all-reduce of the flat gradient slab launched from post-accumulate-grad hooks == sum of per-rank gradients."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ldmae_amd.distributed import GradBucketReducer
    from ldmae_amd.optim import FlatParams
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(), torch.nn.Linear(32, 4))
    if rank == 1:                                   # desynchronise, then broadcast_params must repair it
        with torch.no_grad():
            net[0].weight.add_(1.0)
    flat = FlatParams(net)
    red = GradBucketReducer(flat, bucket_bytes=512)           # several buckets
    red.broadcast_params(0)
    assert len(red.buckets) >= 3
    torch.manual_seed(100 + rank)
    x = torch.randn(8, 16)
    # this rank's own gradient, computed without touching .grad (no hooks fire for autograd.grad)
    own = torch.autograd.grad(net(x).pow(2).mean(), [p for _, p in flat.trainable])
    local = torch.zeros_like(flat.grads)
    for (n, p), g in zip(flat.trainable, own):
        o, k = flat.offsets[n]
        local[o:o + k] = g.reshape(-1)
    for it in range(2):                             # two steps: counters re-arm
        flat.grads.zero_()
        net(x).pow(2).mean().backward()             # hooks launch the bucket all-reduces while backward runs
        scale = red.finish()
        assert scale == 0.5
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered)
    ok = torch.allclose(flat.grads, expect, rtol=1e-6, atol=1e-7)
    # gradient accumulation (accum = 2) as the train driver runs it: the slab accumulates both micro-steps locally and is
    # all-reduced ONCE (sync only on the last micro-step) -> sum over ranks of (g1 + g2); the round-1 form (all-reduce of the
    # accumulating slab on every micro-step) gave world * g1 + g2
    torch.manual_seed(200 + rank)
    x2 = torch.randn(8, 16)
    own2 = torch.autograd.grad(net(x2).pow(2).mean(), [p for _, p in flat.trainable])
    local2 = local.clone()
    for (n, p), g in zip(flat.trainable, own2):
        o, k = flat.offsets[n]
        local2[o:o + k] += g.reshape(-1)
    flat.grads.zero_()
    for mi, xx in enumerate((x, x2)):
        red.sync = mi == 1
        net(xx).pow(2).mean().backward()
        if mi == 0:
            assert not red._works           # nothing launched on a non-final micro-step
    red.finish()
    gathered2 = [torch.zeros_like(local2) for _ in range(world)]
    dist.all_gather(gathered2, local2)
    ok = ok and torch.allclose(flat.grads, sum(gathered2), rtol=1e-6, atol=1e-7)
    params = [torch.zeros_like(flat.params) for _ in range(world)]
    dist.all_gather(params, flat.params)
    same = torch.equal(params[0], params[1])
    if rank == 0:
        out.put((ok, same))
    dist.barrier()
    try:                                  # results are out, both ranks are past the barrier: gloo's teardown racing the peer's is not what is tested
        dist.destroy_process_group()
    except Exception as e:                # noqa: BLE001
        print("destroy_process_group:", e)


def test_bucketed_allreduce_matches_sum_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for r, p in enumerate(procs):
        p.join(120)
        assert p.exitcode == 0, f"rank {r} exit code {p.exitcode}"
    ok, same = q.get(timeout=5)
    assert ok and same


# ----------------------------------------------------------------------------- world 8 on the REAL DiT-B/1 gradient slab (pre-flight for the 8-GPU node)
class _B1Layout:
    """The parameter names / shapes / trainability of LightningDiT-B/1 (oracle.dit.param_shapes = the reference's state-dict keys) with
    uninitialised storage: what FlatParams and GradBucketReducer see of the real model, without building it."""

    def __init__(self):
        from oracle import dit as odit
        self.shapes = odit.param_shapes(odit.DiTConfig(**odit.DIT_B_1))
        self.params = [(n, torch.nn.Parameter(torch.empty(s), requires_grad=(n != "pos_embed"))) for n, s in self.shapes.items()]

    def named_parameters(self):
        return iter(self.params)


def _worker8(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ldmae_amd.distributed import GradBucketReducer
    from ldmae_amd.optim import FlatParams, adaln_first
    flat = FlatParams(_B1Layout(), front_fn=adaln_first)
    red = GradBucketReducer(flat)                                    # the driver's default: 64 MiB buckets
    launched = []
    launch = red._launch
    red._launch = lambda bi: (launched.append(bi), launch(bi))[1]
    names = [n for n, _ in flat.trainable]
    ada = [n for n in names if adaln_first(n)]
    rest = [n for n in names if not adaln_first(n)]
    hooks = {n: p._ldmae_grad_ready for n, p in flat.trainable}
    params = dict(flat.trainable)
    # layout facts the 8-GPU run relies on
    facts = dict(n_buckets=len(red.buckets), slab_mb=flat.n_trainable * 4 / 2 ** 20, n_front=flat.n_front,
                 ada_buckets=sorted({red.param_bucket[n] for n in ada}), rest_buckets=sorted({red.param_bucket[n] for n in rest}))
    pattern = (torch.arange(flat.n_trainable, dtype=torch.float32) % 251) - 125            # small integers: every sum below is exact in f32

    def backward_like(scale):
        """what a micro-step's backward does to the slab: gradients are ADDED, and the hooks fire in the order backward finishes
        parameters -- last layers first, the blocks' adaLN weights (batched adaLN: two GEMMs at the very end) last of all"""
        flat.grads.add_(pattern * scale)
        for n in list(reversed(rest)) + ada:
            hooks[n](params[n])

    # accumulation of two micro-steps: nothing may be launched on the first (DDP's no_sync), one all-reduce of the local sum on the second
    flat.grads.zero_()
    red.sync = False
    backward_like(float(rank + 1))
    none_early = not launched and not red._works
    red.sync = True
    backward_like(float(2 * (rank + 1)))
    scale = red.finish()
    expect = pattern * float(3 * sum(r + 1 for r in range(world)))
    exact = torch.equal(flat.grads, expect)
    # end of the slab first, in order, while "backward" runs; the buckets that hold adaLN weights last (in whatever order the batched adaLN
    # backward hands its gradients over)
    first_ada = min(red.param_bucket[n] for n in ada)
    order_ok = launched[:first_ada] == list(range(first_ada)) and sorted(launched[first_ada:]) == list(range(first_ada, len(red.buckets)))
    # second optimizer step: the counters re-armed
    launched.clear()
    flat.grads.zero_()
    backward_like(1.0)
    red.finish()
    again = torch.equal(flat.grads, pattern * float(world)) and sorted(launched) == list(range(len(red.buckets)))
    if rank == 0:
        out.put(dict(facts, none_early=none_early, scale=scale, exact=exact, order_ok=order_ok, again=again))
    dist.barrier()
    try:
        dist.destroy_process_group()
    except Exception as e:                # noqa: BLE001
        print("destroy_process_group:", e)


def test_world8_reducer_on_the_real_b1_slab():
    """Eight gloo ranks (CPU tensors) run GradBucketReducer on the real DiT-B/1 gradient slab (130 M f32 = 497 MiB, adaLN weights first):
    bucket count, the adaLN buckets launched LAST, `sync = False` accumulation, result == the exact sum over the ranks, scale = 1/8."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for r, p in enumerate(procs):
        p.join(600)
        assert p.exitcode == 0, f"rank {r} exit code {p.exitcode}"
    res = q.get(timeout=5)
    assert res["none_early"] and res["exact"] and res["order_ok"] and res["again"], res
    assert res["scale"] == 1.0 / world
    assert 495 < res["slab_mb"] < 500 and res["n_buckets"] == 8 and res["n_front"] == 24, res
    # the blocks' adaLN parameters (24 tensors, 162 MiB) sit at the front: they fill the LAST buckets, and only those
    assert res["ada_buckets"] == list(range(res["ada_buckets"][0], res["n_buckets"])), res
    assert max(res["rest_buckets"]) <= res["ada_buckets"][0], res
