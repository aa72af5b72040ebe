"""Host-side support for the training loop: meters, schedules, distributed helpers, EMA, accuracy.

Mirrors utils/__init__.py of the reference: SmoothedValue / MetricLogger (:103-275, cross-rank sync :122-142),
cosine_scheduler (:667-684), get_world_size / get_rank / is_main_process (:428-449), init_distributed_mode (:482-547,
here RCCL through torch.distributed's "nccl" backend, one process per GPU), and timm's ModelEma / accuracy that
engine.py:12-13 imports.
"""
from __future__ import annotations

import datetime
import math
import os
import time
from collections import defaultdict, deque

import numpy as np
import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


class SmoothedValue(object):
    """Window median/average plus global average of a series (utils/__init__.py:103-191)."""

    def __init__(self, window_size=20, fmt=None):
        if fmt is None:
            fmt = "{median:.4f} ({global_avg:.4f})"
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0
        self.fmt = fmt

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """SUM of [count, total] over ranks; the window is not synchronised (utils/__init__.py:122-142)."""
        if not is_dist_avail_and_initialized():
            return
        dev = torch.device("cuda", torch.cuda.current_device()) if (torch.cuda.is_available() and dist.get_backend() == "nccl") else torch.device("cpu")
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        t = t.tolist()
        self.count = int(t[0])
        self.total = t[1]

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger(object):
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("{}: {}".format(name, str(meter)) for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        """One packed fp64 all-reduce for all meters (the reference does one barrier + all_reduce per meter;
        the sums are identical)."""
        if not is_dist_avail_and_initialized() or not self.meters:
            return
        names = sorted(self.meters)
        dev = torch.device("cuda", torch.cuda.current_device()) if (torch.cuda.is_available() and dist.get_backend() == "nccl") else torch.device("cpu")
        t = torch.tensor([[self.meters[n].count, self.meters[n].total] for n in names], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        for n, (c, tot) in zip(names, t.tolist()):
            self.meters[n].count = int(c)
            self.meters[n].total = tot

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        i = 0
        header = header or ""
        start_time = time.time()
        end = time.time()
        iter_time = SmoothedValue(fmt="{avg:.4f}")
        data_time = SmoothedValue(fmt="{avg:.4f}")
        n = len(iterable) if hasattr(iterable, "__len__") else None
        for obj in iterable:
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if print_freq and (i % print_freq == 0 or (n is not None and i == n - 1)):
                if n is not None:
                    eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                    print(self.delimiter.join([header, f"[{i}/{n}]", f"eta: {eta}", str(self), f"time: {iter_time}", f"data: {data_time}"]))
                else:
                    print(self.delimiter.join([header, f"[{i}]", str(self), f"time: {iter_time}", f"data: {data_time}"]))
            i += 1
            end = time.time()
        total_time = time.time() - start_time
        print("{} Total time: {} ({:.4f} s / it)".format(header, str(datetime.timedelta(seconds=int(total_time))), total_time / max(i, 1)))


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """Linear warm-up then half-cosine, one value per iteration (utils/__init__.py:667-684)."""
    warmup_schedule = np.array([])
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    if warmup_epochs > 0:
        warmup_schedule = np.linspace(start_warmup_value, base_value, warmup_iters)
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / (len(iters)))) for i in iters])
    schedule = np.concatenate((warmup_schedule, schedule))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def accuracy(output, target, topk=(1,)):
    """timm.utils.accuracy: top-k accuracy in percent (engine.py:405)."""
    maxk = min(max(topk), output.size()[1])
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.reshape(1, -1).expand_as(pred))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100.0 / batch_size for k in topk]


def init_distributed_mode(args=None, backend=None):
    """One process per GPU from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*): binds the GPU and creates
    the RCCL process group ("nccl" backend on ROCm).  Falls back to a single process when the env is absent
    (utils/__init__.py:482-547)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args is not None:
        args.rank, args.world_size, args.gpu, args.distributed = rank, world, local, world > 1
    if world <= 1:
        return False
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank,
                                timeout=datetime.timedelta(minutes=30), **kw)
    dist.barrier()
    return True


class ModelEma:
    """Exponential moving average of a model's weights: ema = decay * ema + (1 - decay) * model after every optimizer
    step (timm.utils.ModelEma, used at engine.py:198-202,273-274).  For the flat-buffer ViT the update is one fused HIP
    launch over the whole parameter buffer; `ema` is a deep copy usable for evaluation."""

    def __init__(self, model, decay=0.9999, device="", resume=""):
        import copy
        self.ema = copy.deepcopy(model)
        self.ema.eval()
        self.decay = decay
        for p in self.ema.parameters():
            p.requires_grad_(False)
        if hasattr(self.ema, "_sync_weights") and next(model.parameters()).is_cuda:
            self.ema._sync_weights(force=True)

    def update(self, model):
        from . import ops
        from ._lib import VitkError
        src = model.module if hasattr(model, "module") else model
        if not (hasattr(src, "_flat") and hasattr(self.ema, "_flat") and src._flat["p"].is_cuda):
            raise VitkError("ModelEma.update: expected the flat-buffer VisionTransformer on the GPU (no torch fallback)")
        if not self.ema._views_ok():
            self.ema._flatten()
        ops.ema_update(self.ema._flat["p"], src._flat["p"], self.decay)
        self.ema._weights_version = None          # bf16 copies are refreshed on the EMA model's next forward
