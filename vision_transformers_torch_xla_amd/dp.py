"""Data parallelism for the flat-buffer ViT: one process per GPU, RCCL all-reduce of gradient buckets over xGMI,
launched from inside backward on a side HIP stream so that communication overlaps the remaining backward kernels.

Replaces the reference's two gradient-sync mechanisms: xm.optimizer_step's all-reduce of every gradient
(engine.py:185) and torch DistributedDataParallel (main.py:855-858).  Semantics kept: gradients are averaged over
replicas before the optimizer step; parameters start identical on every rank (rank 0's are broadcast at wrap time,
DDP's behaviour); with gradient accumulation only the last micro-step communicates.

Because the model's gradients live in one flat buffer laid out in the order backward finalises them, a "bucket" is
just the next contiguous slice [done, ready): no per-parameter hooks, no copies into staging buffers.  xGMI is
point-to-point, so a ring all-reduce is bound by one link (~153 GB/s/GPU): ViT-B's 346 MB of fp32 gradients take
about 4 ms per step, against a 57 ms step of which 37 ms are backward; buckets of >= 32 MB keep the per-collective
latency negligible while leaving >= 10 buckets to overlap.
"""
from __future__ import annotations

import contextlib
import os
import sys

import torch
import torch.distributed as dist
import torch.nn as nn


_DEBUG = os.environ.get("VITK_DP_DEBUG") == "1"


class GradReducer:
    def __init__(self, flat_grad_getter, process_group=None, bucket_bytes=32 << 20, single_rank_collectives=False):
        self._get = flat_grad_getter
        self.pg = process_group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # a one-rank group normally skips communication; a rehearsal on a one-GPU box can ask for the collectives anyway
        self._skip = self.world <= 1 and not (single_rank_collectives and dist.is_initialized())
        self.enabled = True
        self._done = 0
        self._works = []
        self._stream = None
        self._events = []
        self._after = None        # latest event of a stream other than the current one that also writes gradients
        self.launched = []        # (start, end) slices reduced in the last backward, for tests / introspection
        self._new_pass = True     # the next report starts a new backward pass
        self.timing = False       # bracket every collective with timed events on the reducer's stream (diagnostics, see comm_ms)
        self._timed = []          # (start event, end event, elements) of the collectives launched while `timing` was on

    def _side_stream(self, device):
        if self._stream is None or self._stream.device != device:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def ready(self, upto: int, final: bool, after=None):
        """Backward reports that gradients in [0, upto) are final once everything queued on the current stream has run - and,
        when backward writes some gradients on another stream, once `after` (an event recorded on that stream; later events
        supersede earlier ones) has completed.  Reduce [done, upto) once it is a bucket's worth (or at the end)."""
        if not self.enabled or self._skip:
            self.launched = []
            if final:
                self._done = 0
                self._after = None
            return
        if self._new_pass:
            self.launched = []
            self._new_pass = False
        if after is not None:
            self._after = after
        # a bucket's worth is ready - or less than a bucket is left after this point: then what is ready goes now, so that the
        # last collective of the step, the only one nothing overlaps, is as small as it can be
        remaining = self._get().numel() - upto
        if final or upto - self._done >= self.bucket_elems or (upto > self._done and remaining < self.bucket_elems):
            self._launch(self._done, upto)
            self._done = upto
        if final:
            self._done = 0
            self._after = None
            self._new_pass = True
            # Whatever follows backward on this stream (clipping, the optimizer step, a reader of .grad) reads the AVERAGED
            # gradients: the stream waits here for every bucket, as torch DDP's backward does.  All collectives are already
            # launched, so nothing of the overlap is lost.
            self._join()

    def _launch(self, a, b):
        if b <= a:
            return
        g = self._get()[a:b]
        self.launched.append((a, b))
        if _DEBUG:
            print(f"[dp] rank {dist.get_rank(self.pg)} all-reduce [{a}, {b})", file=sys.stderr, flush=True)
        # RCCL averages inside the collective; gloo (CPU rehearsal, or several ranks sharing one GPU) has no AVG:
        # pre-scale, then SUM
        native_avg = dist.get_backend(self.pg) == "nccl"
        if g.is_cuda:
            cur = torch.cuda.current_stream(g.device)
            side = self._side_stream(g.device)
            ev = torch.cuda.Event()
            ev.record(cur)                      # the slice is complete once everything queued so far has run
            side.wait_event(ev)
            if self._after is not None:
                side.wait_event(self._after)
            with torch.cuda.stream(side):
                if self.timing:
                    t_start = torch.cuda.Event(enable_timing=True)
                    t_start.record(side)
                if native_avg:
                    dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.pg)
                else:
                    # gloo rehearsal with GPU tensors: stage through the host on this side stream and reduce there.
                    # (ProcessGroupGloo's own device path takes a fresh pool stream per call and stalls after a few
                    # dozen calls when two ranks share a device.)
                    host = g.to("cpu")                   # waits for the slice (side stream is ordered after `ev`)
                    host.mul_(1.0 / self.world)
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.pg)
                    g.copy_(host)
                done = torch.cuda.Event(enable_timing=self.timing)
                done.record(side)
                if self.timing:
                    self._timed.append((t_start, done, b - a))
            self._events.append(done)
        else:
            g.mul_(1.0 / self.world)
            self._works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _join(self):
        for w in self._works:
            w.wait()
        self._works = []
        if self._events:
            cur = torch.cuda.current_stream()
            for ev in self._events:
                cur.wait_event(ev)
            self._events = []

    def comm_ms(self, reset=True):
        """(milliseconds, collectives, elements) of the collectives launched while `timing` was on: the time between each
        collective's start on the reducer's stream (its inputs ready) and its end - transfer time plus the wait for the slowest
        peer.  Synchronises with the reducer's stream."""
        total, n, elems = 0.0, 0, 0
        for t0, t1, cnt in self._timed:
            t1.synchronize()
            total += t0.elapsed_time(t1)
            n += 1
            elems += cnt
        if reset:
            self._timed = []
        return total, n, elems

    def wait(self):
        """Make the current stream (or the host, for CPU tensors) wait for every outstanding bucket.  Backward already does
        this when it reports its last gradients, so after a complete backward this is a no-op kept for callers that reduce
        by hand; calling it any number of times is harmless."""
        self._join()


class DistributedDataParallel(nn.Module):
    """Drop-in for torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu]) as the reference calls it
    (main.py:855-858), for the flat-buffer VisionTransformer (optionally inside StudentWithDistillation)."""

    def __init__(self, module, device_ids=None, find_unused_parameters=False, process_group=None, bucket_bytes=32 << 20,
                 broadcast_parameters=True, single_rank_collectives=False):
        super().__init__()
        self.module = module
        core = module.student if hasattr(module, "student") else module
        if not hasattr(core, "_flat"):
            raise RuntimeError("DistributedDataParallel: expected a vision_transformers_torch_xla_amd VisionTransformer")
        self._core = core
        self.reducer = GradReducer(core._ensure_grad_buffer, process_group, bucket_bytes, single_rank_collectives)
        core._grad_ready_hook = self.reducer.ready
        if broadcast_parameters and dist.is_initialized() and not self.reducer._skip:
            if not core._views_ok():
                core._flatten()
            dist.broadcast(core._flat["p"], src=0, group=process_group)
            core._weights_version = None

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        old = self.reducer.enabled
        self.reducer.enabled = False
        try:
            yield
        finally:
            self.reducer.enabled = old
