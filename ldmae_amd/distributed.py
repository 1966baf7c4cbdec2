"""Data-parallel gradient exchange for the train step (reference: DDP via accelerate,
LDMAE/train_accum.py:105,188,230 -- bucketed all-reduce(SUM)/world of 130 M f32 gradients per micro-step).

MI355X design: gradients already live in ONE contiguous f32 buffer (``optim.FlatParams``) laid out in
forward order, so buckets are plain slices taken from the END of the buffer backwards (the order the
backward pass finishes them).  Each bucket is all-reduced with RCCL (torch.distributed backend "nccl")
on a side HIP stream as soon as the last gradient in it has been accumulated, overlapping the rest of
backward; the 1/world scaling is folded into the fused AdamW kernel (``AdamWEMA.step(grad_scale)``).
xGMI is point-to-point (7 links x ~153 GB/s), so buckets are large (default 64 MiB: ~8 collectives per
step) rather than DDP's 25 MiB.  Works unchanged on CPU tensors with the gloo backend (tests).

RCCL's collective kernels hold some CUs while a bucket overlaps backward; a persistent GEMM (exactly one workgroup per CU for
the whole launch) would run its displaced workgroups as a second round.  The DRIVER that builds a reducer with world > 1 therefore
asks for one-tile-per-workgroup GEMM launches, ``ops.set_gemm_launch_mode(reducer.recommended_gemm_launch_mode())`` -- a per-call
flag of the C ABI (LDMAE_EPI_TILE_LAUNCH), not library state; this class does not touch it.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def batched_adaln_pays(adaln_bytes: int, n_blocks: int, world: int, busbw_gbps: float = None, saving_ms_per_block: float = 0.125) -> bool:
    """Should a data-parallel train step keep BATCHED adaLN (models.lightningdit._AdaLNAllFn)?  Batching saves ~0.125 ms per block
    (measured on B/1: 1.5 ms for 12 blocks) but finishes the adaLN weight gradients of every block at the very END of backward, so their
    buckets (`adaln_bytes` of the slab; 170 MB for B/1, 890 MB for XL/1) are all-reduced fully exposed.  Until a multi-rank run measures the
    exposed time (train_accum logs it), the choice is made by size: batched only while a ring all-reduce of those bytes,
    2 (w - 1) / w * bytes / bus bandwidth, is below the saving.  LDMAE_XGMI_BUSBW_GBPS sets the bandwidth assumed (default 250 GB/s, the low
    end of what RCCL reaches on an 8-GPU xGMI node for 100-MB messages); LDMAE_BATCHED_ADALN=0|1 overrides the whole rule."""
    if world <= 1:
        return True
    bw = busbw_gbps if busbw_gbps is not None else float(os.environ.get("LDMAE_XGMI_BUSBW_GBPS", "250"))
    exposed_ms = 2.0 * (world - 1) / world * adaln_bytes / (bw * 1e9) * 1e3
    return exposed_ms < saving_ms_per_block * n_blocks


class GradBucketReducer:
    def __init__(self, flat, process_group=None, bucket_bytes: int = 64 << 20, overlap: bool = True, force_hooks: bool = False):
        """force_hooks: arm the hooks / side stream / bucket launches for a world of ONE rank too (bench.py --dp-config and the world-1
        RCCL test: the data-parallel program on a single GPU; needs an initialised process group)."""
        self.flat = flat
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force_hooks and dist.is_initialized())
        self.is_cuda = flat.grads.is_cuda
        self.overlap = overlap and self.is_cuda
        self.stream = torch.cuda.Stream() if self.overlap else None
        # buckets: contiguous [lo, hi) element ranges, built from the last parameter backwards
        names = [n for n, _ in flat.trainable]
        self.buckets, self.param_bucket = [], {}
        cap = max(1, bucket_bytes // 4)
        hi = flat.n_trainable
        cur = []
        for n in reversed(names):
            o, _ = flat.offsets[n]
            cur.append(n)
            if hi - o >= cap:
                self.buckets.append((o, hi, cur))
                hi, cur = o, []
        if cur:
            self.buckets.append((0, hi, cur))
        for bi, (_, _, ns) in enumerate(self.buckets):
            for n in ns:
                self.param_bucket[n] = bi
        self._pending = [len(ns) for _, _, ns in self.buckets]
        self._seen = [set() for _ in self.buckets]
        self._works = []
        self._hooks = []
        self._exposed = []
        self.measure_exposed = False      # bench.py sets it: two timing events per step, drained by exposed_comm_ms(); off in training
        self.measure_timeline = False     # bench.py, one step: per-bucket launch / completion events -> bucket_timeline()
        self._tl, self._tl_done = [], None
        # DDP.no_sync() equivalent: with gradient accumulation the slab holds the LOCAL sum of the micro-step gradients and is
        # all-reduced once, on the last micro-step (`sync = True` before that backward).  All-reducing the accumulating slab on
        # every micro-step would re-sum earlier micro-steps across ranks (world * g1 + g2).
        self.sync = True
        if self.active:
            for n, p in flat.trainable:
                h = self._make_hook(n)
                self._hooks.append(p.register_post_accumulate_grad_hook(h))
                p._ldmae_grad_ready = h          # for gradients written into .grad without AccumulateGrad (models.lightningdit._dw_into_grad)

    def recommended_gemm_launch_mode(self) -> str:
        return "tile" if (self.active and self.is_cuda) else "persistent"

    def _make_hook(self, name):
        def hook(_p):
            if not self.sync:
                return
            bi = self.param_bucket[name]
            if name in self._seen[bi]:        # idempotent: a parameter may be announced by its post-accumulate hook AND by the callback
                return                        # of a gradient written straight into .grad (autograd may still visit its AccumulateGrad)
            self._seen[bi].add(name)
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi):
        lo, hi, _ = self.buckets[bi]
        buf = self.flat.grads[lo:hi]
        if self.overlap:
            tl = self.measure_timeline
            ev = torch.cuda.Event(enable_timing=tl)
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                if tl:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                if tl:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(self.stream)
                    self._tl.append((bi, (hi - lo) * 4, ev, e0, e1))
        else:
            self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self) -> float:
        """Call after backward: launches buckets whose hooks did not all fire (unused parameters), waits for the
        collectives, re-arms the counters.  Returns the grad scale (1/world) to hand to ``AdamWEMA.step``."""
        if self.active and self.sync:
            for bi, pend in enumerate(self._pending):
                if pend > 0:
                    self._launch(bi)
            if self.is_cuda and self.measure_timeline and self.overlap:
                self._tl_done = torch.cuda.Event(enable_timing=True)      # "backward has finished" on the compute stream
                self._tl_done.record(torch.cuda.current_stream())
            ev0 = ev1 = None
            if self.is_cuda and self.measure_exposed:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(torch.cuda.current_stream())
            for w in self._works:
                w.wait()
            if self.overlap:
                torch.cuda.current_stream().wait_stream(self.stream)
            if ev0 is not None:
                ev1.record(torch.cuda.current_stream())
                self._exposed.append((ev0, ev1))
        self._works = []
        self._pending = [len(ns) for _, _, ns in self.buckets]
        self._seen = [set() for _ in self.buckets]
        return 1.0 / self.world

    def exposed_comm_ms(self) -> float:
        """Sum over the steps since the last call of the time the compute stream waited for the gradient all-reduce after backward
        had finished (HIP events around the wait in finish(); synchronises).  0 on CPU / world 1."""
        tot = 0.0
        for e0, e1 in self._exposed:
            e1.synchronize()
            tot += e0.elapsed_time(e1)
        self._exposed = []
        return tot

    def bucket_timeline(self):
        """Per bucket of the steps since `measure_timeline` was set (synchronises): when backward released it (ms before the end of backward),
        how long its all-reduce took on the side stream, and how much of that ran past the end of backward (= exposed).  What the first
        multi-GPU run needs to see to size buckets and to judge the late adaLN buckets."""
        out = []
        if self._tl_done is None:
            return out
        self._tl_done.synchronize()
        for bi, nbytes, ready, e0, e1 in self._tl:
            e1.synchronize()
            out.append({"bucket": bi, "mbytes": round(nbytes / 2 ** 20, 1), "released_ms_before_backward_end": round(ready.elapsed_time(self._tl_done), 3),
                        "start_delay_ms": round(ready.elapsed_time(e0), 3), "allreduce_ms": round(e0.elapsed_time(e1), 3),
                        "past_backward_end_ms": round(max(0.0, self._tl_done.elapsed_time(e1)), 3)})
        self._tl, self._tl_done = [], None
        return out

    def broadcast_params(self, src: int = 0):
        """DDP-constructor equivalent: replicate rank `src` parameters (and nothing else) to every rank."""
        if self.world > 1:
            dist.broadcast(self.flat.params, src=src, group=self.pg)
            if self.is_cuda:
                from . import ops
                ops.invalidate_weight_cache()        # the slab was rewritten behind the parameter views' version counters
