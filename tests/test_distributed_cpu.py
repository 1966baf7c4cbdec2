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
