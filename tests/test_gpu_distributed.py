"""The N > 1 path on the one GPU of the test box: two fresh ranks (gloo rendezvous, both on cuda:0) run the HIP tiny model with the
post-accumulate-grad hooks and the side-stream bucket launches of ``GradBucketReducer``; the reduced gradient slab must equal the
single-process gradient on the concatenated batch.  Also: the train driver comes up under a torchrun-style
two-process launch (RANK / LOCAL_RANK / WORLD_SIZE from the launcher, reference run_train.sh:13-22)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    sys.path.insert(0, ROOT)
    from oracle import dit as odit
    from weights import det_randn, det_weights
    from ldmae_amd.models.lightningdit import LightningDiT
    cfg = odit.DiTConfig(input_size=8, patch_size=1, in_channels=16, hidden_size=192, depth=2, num_heads=3, num_classes=10, class_dropout_prob=0.0)
    sd = det_weights(odit.param_shapes(cfg), 1)
    sd.update(odit.fixed_tables(cfg))
    m = LightningDiT(input_size=8, patch_size=1, in_channels=16, hidden_size=192, depth=2, num_heads=3, num_classes=10, class_dropout_prob=0.0,
                     use_qknorm=True, use_swiglu=True, use_rope=True, use_rmsnorm=True)
    m.load_state_dict(sd)
    x = det_randn("ddp_x", (8, 16, 8, 8), 1)
    t = torch.linspace(0.1, 0.9, 8)
    y = torch.arange(8) % 10
    tgt = det_randn("ddp_tgt", (8, 16, 8, 8), 2)
    return m.cuda().train(), x, t, y, tgt


def _grad_slab(m, flat, x, t, y, tgt):
    flat.grads.zero_()
    loss = ((m(x.cuda(), t.cuda(), y.cuda()) - tgt.cuda()) ** 2).mean()
    loss.backward()
    return loss


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("LDMAE_TUNE", None)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    m, x, t, y, tgt = _tiny()
    from ldmae_amd import _lib
    from ldmae_amd.distributed import GradBucketReducer
    from ldmae_amd.optim import AdamWEMA, adaln_first
    from ldmae_amd import ops
    # what a model gets when NOBODY configures it (the reference's own train_accum.py under accelerate / DDP through the drop-in): in a world
    # of two ranks the per-block adaLN (its weight gradients complete with their block) and one-tile-per-workgroup GEMM launches
    assert m.batched_adaln is None and not m._use_batched_adaln() and ops.gemm_launch_mode() == "tile"
    with torch.no_grad():
        assert m._use_batched_adaln()                                    # forward-only calls keep the batched form
    opt = AdamWEMA(m, lr=1e-3, front_fn=adaln_first)                     # the drivers' layout: adaLN weights first in the slab ...
    m.batched_adaln = True                                               # ... so the batched form stays on under the reducer
    names = [n for n, _ in opt.flat.trainable]
    assert opt.flat.n_front == 4 and all(adaln_first(n) for n in names[:4]) and not any(adaln_first(n) for n in names[4:])
    red = GradBucketReducer(opt.flat, bucket_bytes=256 << 10)            # several buckets, launched from the hooks on the side stream
    red.broadcast_params(0)
    m.direct_param_grads = True      # as the train drivers do: block weight gradients go straight into the slab and notify the reducer by callback
    assert len(red.buckets) >= 3 and red.overlap
    assert all(adaln_first(n) for n in red.buckets[-1][2][-4:])          # the adaLN weights sit in the bucket that is reduced last
    assert red.recommended_gemm_launch_mode() == "tile"                  # the reducer only recommends
    ops.set_gemm_launch_mode(red.recommended_gemm_launch_mode())         # ... the driver sets it: one tile per workgroup, per call
    red.measure_exposed = True
    sl = slice(rank * 4, rank * 4 + 4)
    for it in range(2):                                                  # twice: the counters re-arm
        _grad_slab(m, opt.flat, x[sl], t[sl], y[sl], tgt[sl])
        scale = red.finish()
    torch.cuda.synchronize()
    exposed = red.exposed_comm_ms()
    if rank == 0:
        q.put(((opt.flat.grads.cpu() * scale).numpy(), exposed))      # by value: a torch tensor travels as a file descriptor that needs THIS process alive when the parent unpickles it
    dist.barrier()
    try:                                  # the result is out and both ranks are past the barrier: a peer that closed its sockets first
        dist.destroy_process_group()      # ("connection closed by peer" in gloo's teardown) is not a failure of what this test checks
    except Exception as e:                # noqa: BLE001
        print("destroy_process_group:", e)


def test_two_ranks_one_gpu_reduced_grads_equal_concatenated_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, exposed = q.get(timeout=300)
    got = torch.from_numpy(got)
    for r, p in enumerate(procs):
        p.join(120)
        assert p.exitcode == 0, f"rank {r} exit code {p.exitcode}"
    m, x, t, y, tgt = _tiny()
    from ldmae_amd.optim import AdamWEMA, adaln_first
    opt = AdamWEMA(m, lr=1e-3, front_fn=adaln_first)                      # same slab layout as the ranks
    _grad_slab(m, opt.flat, x, t, y, tgt)                                 # single process, all 8 samples: mean loss = mean of the halves' means
    ref = opt.flat.grads.cpu()
    err = float((got - ref).norm() / ref.norm())
    print("2-rank reduced slab vs single-process slab: rel err", err, "exposed comm ms", exposed)
    assert err < 1e-5 and exposed >= 0.0


def _world1_rccl_worker(port, q):
    """One RCCL rank: librccl loads, the reducer's hooks fire, every bucket is all-reduced on the side stream and finish() joins it."""
    os.environ.pop("LDMAE_TUNE", None)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    m, x, t, y, tgt = _tiny()
    from ldmae_amd.distributed import GradBucketReducer
    from ldmae_amd.optim import AdamWEMA, adaln_first
    opt = AdamWEMA(m, lr=1e-3, front_fn=adaln_first)
    m.batched_adaln = True
    m.direct_param_grads = True

    def slab():
        opt.flat.grads.zero_()
        with torch.autocast("cuda", dtype=torch.bfloat16):               # batch 8, bf16: the batched adaLN path runs
            loss = ((m(x.cuda(), t.cuda(), y.cuda()).float() - tgt.cuda()) ** 2).mean()
        loss.backward()
        return opt.flat.grads.clone()
    ref = slab()                                                          # no process group, no hooks
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    red = GradBucketReducer(opt.flat, bucket_bytes=256 << 10, force_hooks=True)
    assert red.active and red.world == 1 and len(red.buckets) >= 3 and red.overlap and red.recommended_gemm_launch_mode() == "tile"
    red.broadcast_params(0)
    red.measure_exposed = True
    launched = []
    orig = red._launch
    red._launch = lambda bi: (launched.append(bi), orig(bi))[1]
    got = None
    for _ in range(2):                                                    # twice: the counters re-arm
        launched.clear()
        got = slab()
        scale = red.finish()
        assert sorted(launched) == list(range(len(red.buckets))) and scale == 1.0
    torch.cuda.synchronize()
    front = sorted({red.param_bucket[n] for n, _ in opt.flat.trainable if adaln_first(n)})
    q.put((bool(torch.equal(got, ref)), list(launched), front, red.exposed_comm_ms()))
    try:
        dist.destroy_process_group()
    except Exception as e:                # noqa: BLE001  (teardown only: everything the test checks is in the queue)
        print("destroy_process_group:", e)


def test_world_of_one_rccl_rank_walks_the_reducer():
    """init_process_group("nccl") with a world of ONE rank on the test GPU (RCCL itself executes: the 8-GPU node is the driver's), the
    reducer armed with force_hooks: every bucket goes through dist.all_reduce on the side stream, the bucket with the adaLN weights (front of
    the slab) is launched LAST, and the slab equals the hook-free gradient bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_world1_rccl_worker, args=(_free_port(), q))
    p.start()
    same, launched, front, exposed = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0, f"exit code {p.exitcode}"
    # the buckets that hold adaLN weights (front of the slab = highest bucket indices) are among the last launched: only the bucket with the
    # embedders' parameters, whose gradients complete after the batched adaLN backward, may come between / after them
    assert same and exposed >= 0.0 and front == list(range(front[0], len(launched))), (launched, front)
    assert set(front) <= set(launched[-(len(front) + 1):]) and launched[0] == 0, (launched, front)


def test_bench_dp_config_line():
    """`bench.py --dp-config`: the data-parallel program (world-1 RCCL group, hooks + side stream, adaLN weights first, batched adaLN) timed
    with both GEMM launch modes on one GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LDMAE_TUNE")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dp-config", "--steps", "2", "--batch", "8", "--no-power"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["dp_config"]
    assert d["tile"]["ms_per_step"] > 0 and d["persistent"]["ms_per_step"] > 0 and d["buckets"] >= 2 and d["batched_adaln"] and d["adaln_params_first"] == 24


def test_train_driver_comes_up_under_two_process_launch(tmp_path):
    cfg = yaml.safe_load(open(os.path.join(ROOT, "ldmae_amd/configs/imagenet/lightningdit_b_vmae_f8d16_cfg.yaml")))
    cfg["data"].update(image_size=64, num_workers=0)                      # 8x8 latents: 64 tokens at the real B/1 width
    cfg["train"].update(global_batch_size=8, output_dir=str(tmp_path), exp_name="ddp", log_every=1, ckpt_every=2, max_steps=2,
                        gradient_accumulation_steps=2)
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, LDMAE_DIST_BACKEND="gloo", LDMAE_DEVICE="0", PRECISION="bf16", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LDMAE_TUNE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "ldmae_amd", "train_accum.py"), "--config", str(cfg_path), "--synthetic"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    log = (tmp_path / "ddp" / "log.txt").read_text()
    assert "batch 4/gpu x 2 gpus x 2 accumulation" in log and "(step=0000002)" in log and "Done!" in log
    ck = torch.load(tmp_path / "ddp" / "checkpoints" / "0000002.pt", map_location="cpu")
    assert all(torch.isfinite(v).all() for v in ck["model"].values())
    loss = float(log.split("(step=0000002) Train Loss: ")[1].split(",")[0])
    assert np.isfinite(loss) and 0.1 < loss < 10.0


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) must itself bring up 2 ranks (reference launcher:
    run_train.sh:13-22) and label the line n_gpus = 2 with a `comm` block; here the two ranks share cuda:0 over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LDMAE_BENCH_BACKEND="gloo", LDMAE_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-power"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16 and line["config"]["parallelism"] == "dp2"
    assert line["comm"]["backend"] == "gloo" and line["comm"]["buckets"] >= 1 and line["comm"]["gemm_launch_mode"] == "one tile per workgroup"
    assert line["value"] > 0 and line["roofline"]["achieved"] > 0 and "cpu_baseline" not in line
    # per-bucket timeline of one step (when backward released the bucket, how long its all-reduce took, how much ran past backward)
    tl = line["comm"]["bucket_timeline_rank0"]
    assert len(tl) == line["comm"]["buckets"] and all(b["allreduce_ms"] >= 0 and b["released_ms_before_backward_end"] >= -1e-3 for b in tl), tl
    assert line["comm"]["rccl_ranks"] == 0            # a gloo rehearsal is never labelled as an RCCL run


def test_bench_refuses_an_rccl_run_without_one_gpu_per_rank():
    """`--gpus 2` with the default (nccl = RCCL) backend on a box with fewer GPUs than ranks must fail loudly, not fall back to anything."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LDMAE_BENCH_BACKEND", "LDMAE_BENCH_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8", "--no-power"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "needs 2 visible GPUs" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-1500:]


def test_extract_features_two_ranks_share_the_dataset(tmp_path):
    """The shard writer under a launcher (reference extract_features.py:27-41, 109-118): 2 ranks (gloo, both on cuda:0) over 10 synthetic
    images -> rank r encodes samples r, r + 2, ... (DistributedSampler, shuffle=False), each writes latents_rank{r:02d}_shard000, rank 0
    alone builds latents_stats.pt after the barrier; the reader sees all 10."""
    import yaml
    from safetensors import safe_open
    cfg = dict(vae=dict(model_name="vmae_f8d16", weight_path=""), data=dict(origin_path=str(tmp_path / "ds" / "root"), name="imagenet", sample=True))
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LDMAE_DIST_BACKEND="gloo", LDMAE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "-m", "ldmae_amd.extract_features", "--config", str(tmp_path / "cfg.yaml"), "--synthetic", "10",
                        "--image_size", "64", "--batch_size", "2", "--num_workers", "0"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = tmp_path / "ds" / "vmae_feature_imagenet_train_64_sample"
    files = sorted(f for f in os.listdir(out) if f.endswith(".safetensors"))
    assert files == ["latents_rank00_shard000.safetensors", "latents_rank01_shard000.safetensors"] and os.path.exists(out / "latents_stats.pt")
    sys.path.insert(0, ROOT)
    from ldmae_amd.extract_features import _SyntheticImages
    ds = _SyntheticImages(10, 64, False)
    for rank, f in enumerate(files):
        with safe_open(str(out / f), framework="pt") as h:
            lat, labels = h.get_tensor("latents"), h.get_tensor("labels")
            assert h.metadata()["total_size"] == "5"
        assert lat.shape == (5, 32, 8, 8) and labels.tolist() == [ds[i][1] for i in range(rank, 10, 2)]      # rank-strided, in order
    from ldmae_amd.datasets.img_latent_dataset import ImgLatentDataset
    assert len(ImgLatentDataset(str(out), latent_norm=True, sample=True)) == 10


def test_vmae_pretrain_two_ranks_on_a_folder_of_pngs(tmp_path):
    """VMAE pre-training as main_pretrain.py drives it (reference :204-215, 254-256): 2 ranks (gloo, both on cuda:0) over a folder of PNGs --
    DistributedSampler, gradient slab all-reduced by GradBucketReducer (unused parameters keep their zero gradient), fused AdamW, a
    checkpoint per `save_epochs`; then ONE process resumes that checkpoint at twice the resolution (position embeddings resized,
    util/misc.py:488-531) and trains on."""
    from PIL import Image
    rng = np.random.default_rng(1)
    pics = tmp_path / "pics"
    for i in range(16):
        d = pics / f"d{i % 3}"
        d.mkdir(parents=True, exist_ok=True)
        Image.fromarray(rng.integers(0, 256, size=(72, 80, 3), dtype=np.uint8)).save(d / f"{i}.png")
    out = tmp_path / "out"
    env = dict(os.environ, LDMAE_DIST_BACKEND="gloo", LDMAE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LDMAE_TUNE", None)
    script = os.path.join(ROOT, "ldmae_amd", "vmae_pretrain.py")
    common = ["--data_path", str(pics), "--output_dir", str(out), "--batch_size", "4", "--num_workers", "0", "--warmup_epochs", "1",
              "--print_freq", "1", "--save_epochs", "1"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script, "--input_size", "64", "--epochs", "2"] + common
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "effective batch size: 8" in r.stdout and "16 images; 2 iterations per epoch and rank" in r.stdout, r.stdout[-1500:]
    ck = torch.load(out / "checkpoint-1.pth", map_location="cpu", weights_only=False)
    assert ck["epoch"] == 1 and ck["model"]["pos_embed"].shape[1] == 64 and all(torch.isfinite(v).all() for v in ck["model"].values())
    # the optimizer entry is a torch.optim.AdamW.state_dict() over timm's [no_decay, decay] groups (what the reference's load_model feeds to its optimizer)
    og = ck["optimizer"]["param_groups"]
    assert len(og) == 2 and [g["weight_decay"] for g in og] == [0.0, 0.05] and len(ck["optimizer"]["state"]) == sum(len(g["params"]) for g in og)
    assert all(float(e["step"]) == 4 for e in ck["optimizer"]["state"].values()) and ck["scaler"]["scale"] > 0
    losses = [float(l.split("loss: ")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("Epoch: [")]
    assert len(losses) == 4 and all(np.isfinite(losses)) and losses[-1] < losses[0]
    # resume on one rank at 128 px: 64 -> 256 positions
    # ... with the flag set VMAE/train_ae.sh:26-46 passes to main_pretrain.py (minus the LPIPS term, which needs the VGG weights): --no_cls / --smooth_output /
    # --log_dir are accepted, --fixed_std selects the pre-training tree's KL against N(mean, fixed_std^2) (VMAE/util/misc.py:105-116)
    r2 = subprocess.run([sys.executable, script, "--input_size", "128", "--epochs", "3", "--resume", str(out / "checkpoint-1.pth"), "--no_cls", "--smooth_output",
                         "--fixed_std", "1e-3", "--mask_ratio", "0.25", "--visible_loss_ratio", "0.75", "--kl_loss_weight", "1e-6", "--log_dir", str(out),
                         "--blr", "1.0e-4", "--weight_decay", "0.05"] + common,
                        env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout[-1500:] + r2.stderr[-3000:]
    assert "reshape pos embedding" in r2.stdout and "Epoch: [2]" in r2.stdout and "Epoch: [1]" not in r2.stdout
    ck2 = torch.load(out / "checkpoint-2.pth", map_location="cpu", weights_only=False)
    assert ck2["model"]["pos_embed"].shape[1] == 256 and ck2["epoch"] == 2
