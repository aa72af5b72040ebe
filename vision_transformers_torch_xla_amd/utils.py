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
from collections import defaultdict

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


class SmoothedValue:
    """Statistics of one scalar series with the attribute names the reference's meter has (utils/__init__.py:103-191):
    `median`, `avg`, `max`, `value` over the latest `window_size` values, `global_avg` = total / count over everything
    (`update(value, n)` adds value * n to `total` and n to `count`), `fmt` for printing.

    Built for a loop that must not stop the GPU: `update` also takes a 0-d device tensor and keeps it as it is; pending
    device scalars are fetched together - one stack, one copy - when a statistic is read, when `resolve()` is called, or
    when PENDING_LIMIT of them have piled up.  The window is a fixed ring of floats."""

    PENDING_LIMIT = 256

    def __init__(self, window_size=20, fmt=None):
        self.fmt = "{median:.4f} ({global_avg:.4f})" if fmt is None else fmt
        self._ring = np.zeros(max(int(window_size), 1), dtype=np.float64)
        self._seen = 0                      # values pushed so far; the ring holds the last min(_seen, size) of them
        self._count, self._total = 0, 0.0
        self._pending = []                  # (0-d device tensor, n) in arrival order

    # -- writing
    def update(self, value, n=1):
        if isinstance(value, torch.Tensor) and value.is_cuda:
            self._pending.append((value.detach().reshape(()).to(torch.float32), n))
            if len(self._pending) >= self.PENDING_LIMIT:
                self.resolve()
            return
        if self._pending:
            self.resolve()                  # keep the series in arrival order
        self._push(float(value), n)

    def _push(self, v, n):
        self._ring[self._seen % len(self._ring)] = v
        self._seen += 1
        self._count += n
        self._total += v * n

    def resolve(self, values=None):
        """Move the pending device scalars into the statistics (`values`: their host values when a caller fetched them
        together with other meters')."""
        if not self._pending:
            return
        if values is None:
            values = torch.stack([t for t, _ in self._pending]).tolist()
        pending, self._pending = self._pending, []
        for (_, n), v in zip(pending, values):
            self._push(v, n)

    # -- the totals are plain attributes to callers (cross-rank sync and checkpoint code assign them)
    @property
    def count(self):
        self.resolve()
        return self._count

    @count.setter
    def count(self, v):
        self.resolve()
        self._count = v

    @property
    def total(self):
        self.resolve()
        return self._total

    @total.setter
    def total(self, v):
        self.resolve()
        self._total = v

    def _window(self):
        self.resolve()
        k = min(self._seen, len(self._ring))
        if k == 0:
            raise ValueError("SmoothedValue: no values yet")
        return self._ring[:k]

    @property
    def median(self):
        w = np.sort(self._window())
        return float(w[(len(w) - 1) // 2])          # the lower middle value, as torch.median picks it

    @property
    def avg(self):
        return float(self._window().mean())

    @property
    def max(self):
        return float(self._window().max())

    @property
    def value(self):
        self.resolve()
        if self._seen == 0:
            raise ValueError("SmoothedValue: no values yet")
        return float(self._ring[(self._seen - 1) % len(self._ring)])

    @property
    def global_avg(self):
        return self.total / self.count

    def synchronize_between_processes(self):
        """count and total become their sums over all ranks; the window stays local (utils/__init__.py:122-142)"""
        if not is_dist_avail_and_initialized():
            return
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=_collective_device())
        dist.barrier()
        dist.all_reduce(t)
        c, tot = t.tolist()
        self._count, self._total = int(c), tot

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


def _collective_device():
    """where a small tensor must live to be all-reduced: the current GPU under RCCL, the host otherwise"""
    if torch.cuda.is_available() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class MetricLogger:
    """Named SmoothedValue meters, created on first use (`logger.meters[name]`, `logger.update(name=value)`, `logger.name`),
    printed as "name: meter" joined by `delimiter` (utils/__init__.py:194-275).  Device scalars handed to `update` stay on
    the GPU until `flush()` - or any read - fetches ALL meters' pending values in one transfer; the cross-rank sync is one
    packed fp64 all-reduce for every meter instead of a barrier + all-reduce per meter (the sums are the same)."""

    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **named_values):
        for name, v in named_values.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                if v.numel() != 1:
                    raise TypeError(f"MetricLogger.update({name}=...): expected a scalar, got shape {tuple(v.shape)}")
                if not v.is_cuda:
                    v = v.item()
            elif not isinstance(v, (int, float, np.integer, np.floating)):
                raise TypeError(f"MetricLogger.update({name}=...): expected a number or a 0-d tensor, got {type(v).__name__}")
            self.meters[name].update(v)

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def flush(self):
        """one device -> host transfer for every pending scalar of every meter"""
        waiting = [(m, len(m._pending)) for m in self.meters.values() if m._pending]
        if not waiting:
            return
        flat = torch.stack([t for m, _ in waiting for t, _ in m._pending]).tolist()
        at = 0
        for m, k in waiting:
            m.resolve(flat[at:at + k])
            at += k

    def __getattr__(self, name):
        meters = self.__dict__.get("meters")
        if meters is not None and name in meters:
            return meters[name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __str__(self):
        self.flush()
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        self.flush()
        if not is_dist_avail_and_initialized() or not self.meters:
            return
        names = sorted(self.meters)                      # the same order on every rank
        packed = torch.tensor([[self.meters[n].count, self.meters[n].total] for n in names], dtype=torch.float64, device=_collective_device())
        dist.all_reduce(packed)
        for n, (c, tot) in zip(names, packed.tolist()):
            self.meters[n].count, self.meters[n].total = int(c), tot

    def log_every(self, iterable, print_freq, header=None):
        """iterate, timing the loop body ("time") and the wait for the next item ("data"); a status line every `print_freq` items"""
        header = header or ""
        length = len(iterable) if hasattr(iterable, "__len__") else None
        per_item, waiting = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        began = mark = time.time()
        seen = 0
        for seen, item in enumerate(iterable, 1):
            waiting.update(time.time() - mark)
            yield item
            per_item.update(time.time() - mark)
            i = seen - 1
            if print_freq and (i % print_freq == 0 or i + 1 == length):
                fields = [header, f"[{i}/{length}]" if length is not None else f"[{i}]"]
                if length is not None:
                    fields.append("eta: " + str(datetime.timedelta(seconds=int(per_item.global_avg * (length - i)))))
                print(self.delimiter.join(fields + [str(self), f"time: {per_item}", f"data: {waiting}"]))
            mark = time.time()
        spent = time.time() - began
        print(f"{header} Total time: {datetime.timedelta(seconds=int(spent))} ({spent / max(seen, 1):.4f} s / it)")


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """One value per training iteration: a linear ramp start_warmup_value -> base_value over the warm-up iterations
    (warmup_steps > 0 overrides warmup_epochs * niter_per_ep for its LENGTH, but the ramp only exists when
    warmup_epochs > 0), then half a cosine from base_value to final_value over what is left (utils/__init__.py:667-684;
    the values are np.array_equal to the reference's, hence math.cos per element and the same expression order)."""
    total = epochs * niter_per_ep
    ramp_len = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    ramp = np.linspace(start_warmup_value, base_value, ramp_len) if warmup_epochs > 0 else np.array([])
    tail_len = total - ramp_len
    tail = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / tail_len)) for i in range(tail_len)])
    schedule = np.concatenate((ramp, tail))
    if len(schedule) != total:
        # the reference asserts the same condition
        raise AssertionError(f"cosine_scheduler: {len(schedule)} values for {total} iterations (warm-up longer than the run, or warmup_steps without warmup_epochs)")
    return schedule


def accuracy(output, target, topk=(1,)):
    """Top-k accuracy in percent, one 0-d tensor per k (timm.utils.accuracy as engine.py:405 calls it): the share of rows
    whose target is among the k largest logits."""
    widest = min(max(topk), output.shape[1])
    best = output.topk(widest, dim=1).indices                    # [B, widest], largest first
    hit = best == target.reshape(-1, 1)
    rows = target.shape[0]
    return [hit[:, :min(k, widest)].any(dim=1).float().sum() * 100.0 / rows for k in topk]


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
        src = src.student if hasattr(src, "student") else src        # knowledge distillation: the student is what is averaged (main.py:747-748)
        if not (hasattr(src, "_flat") and hasattr(self.ema, "_flat") and src._flat["p"].is_cuda):
            raise VitkError("ModelEma.update: expected the flat-buffer VisionTransformer on the GPU (no torch fallback)")
        if not self.ema._views_ok():
            self.ema._flatten()
        ops.ema_update(self.ema._flat["p"], src._flat["p"], self.decay)
        self.ema._weights_version = None          # bf16 copies are refreshed on the EMA model's next forward


# ---------------------------------------------------------------------------------------------- checkpoints
def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def _plain_args(args):
    """argparse.Namespace -> dict of plain values, so that the checkpoint loads with torch.load(weights_only=True)"""
    src = vars(args) if hasattr(args, "__dict__") else dict(args or {})
    ok = (int, float, str, bool, type(None))
    return {k: (v if isinstance(v, ok) else [x for x in v] if isinstance(v, (list, tuple)) and all(isinstance(x, ok) for x in v) else str(v))
            for k, v in src.items()}


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """checkpoint-{epoch}.pth under args.output_dir, written by rank 0: {'model', 'optimizer', 'epoch', 'scaler', 'args'
    [, 'model_ema']} with timm parameter names, then the checkpoint save_ckpt_num * save_ckpt_freq epochs back is removed
    (utils/__init__.py:686-732).  'args' is stored as a dict of plain values."""
    from pathlib import Path
    output_dir = Path(args.output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    checkpoint_path = output_dir / ("checkpoint-%s.pth" % str(epoch))
    to_save = {
        "model": {k: v.detach().cpu() for k, v in model_without_ddp.state_dict().items()},
        "optimizer": optimizer.state_dict(),
        "epoch": epoch,
        "scaler": loss_scaler.state_dict() if loss_scaler is not None else None,
        "args": _plain_args(args),
    }
    if model_ema is not None:
        ema = model_ema.ema if hasattr(model_ema, "ema") else model_ema
        to_save["model_ema"] = {k: v.detach().cpu() for k, v in ema.state_dict().items()}
    save_on_master(to_save, checkpoint_path)
    if is_main_process() and isinstance(epoch, int):
        to_del = epoch - getattr(args, "save_ckpt_num", 3) * getattr(args, "save_ckpt_freq", 1)
        old_ckpt = output_dir / ("checkpoint-%s.pth" % to_del)
        if os.path.exists(old_ckpt):
            os.remove(old_ckpt)


def _load_checkpoint_file(path):
    """torch.load that executes nothing from the file; argparse.Namespace (what the reference pickles under 'args') is
    allowed by name.  URLs are refused: there is no network."""
    import argparse
    if str(path).startswith("http"):
        raise RuntimeError("checkpoint URLs cannot be fetched in this build (no network): pass a local file")
    if str(path).endswith(".safetensors"):          # models/_helpers.py:62-65: a flat name -> tensor file, nothing to execute
        import safetensors.torch
        return safetensors.torch.load_file(str(path), device="cpu")
    with torch.serialization.safe_globals([argparse.Namespace]):
        return torch.load(path, map_location="cpu", weights_only=True)


def load_checkpoint_state_dict(checkpoint_path, use_ema=True):
    """The state dict of a checkpoint file as models/_helpers.py:44-86 (load_state_dict) picks it: '.safetensors' or torch file;
    'state_dict_ema' / 'model_ema' first when use_ema, then 'state_dict', then 'model', else the file's top level; the 'module.'
    (DistributedDataParallel) and '_orig_mod.' (torch.compile) prefixes removed from the keys (:30-41).  A missing file raises
    FileNotFoundError like the reference."""
    if not (checkpoint_path and os.path.isfile(checkpoint_path)):
        print("No checkpoint found at '{}'".format(checkpoint_path))
        raise FileNotFoundError()
    checkpoint = _load_checkpoint_file(checkpoint_path)
    key = ""
    if isinstance(checkpoint, dict):
        if use_ema and checkpoint.get("state_dict_ema", None) is not None:
            key = "state_dict_ema"
        elif use_ema and checkpoint.get("model_ema", None) is not None:
            key = "model_ema"
        elif "state_dict" in checkpoint:
            key = "state_dict"
        elif "model" in checkpoint:
            key = "model"
    sd = checkpoint[key] if key else checkpoint
    out = {}
    for k, v in sd.items():
        for pre in ("module.", "_orig_mod."):
            if k.startswith(pre):
                k = k[len(pre):]
        out[k] = v
    return out


def save_safetensors(model, path):
    """model.state_dict() (timm's keys and shapes) as a .safetensors file"""
    import safetensors.torch
    safetensors.torch.save_file({k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}, str(path))


def auto_load_model(args, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """Resume from args.resume, or with args.auto_resume from the highest-numbered checkpoint-*.pth in args.output_dir:
    restores the model, and when the file holds them the optimizer state, args.start_epoch = epoch + 1, the EMA weights
    and the scaler (utils/__init__.py:735-770)."""
    import glob
    from pathlib import Path
    output_dir = Path(args.output_dir)
    if getattr(args, "auto_resume", False) and len(getattr(args, "resume", "") or "") == 0:
        latest_ckpt = -1
        for ckpt in glob.glob(os.path.join(output_dir, "checkpoint-*.pth")):
            t = ckpt.split("-")[-1].split(".")[0]
            if t.isdigit():
                latest_ckpt = max(int(t), latest_ckpt)
        if latest_ckpt >= 0:
            args.resume = os.path.join(output_dir, "checkpoint-%d.pth" % latest_ckpt)
        print("Auto resume checkpoint: %s" % getattr(args, "resume", ""))
    if getattr(args, "resume", ""):
        checkpoint = _load_checkpoint_file(args.resume)
        model_without_ddp.load_state_dict(checkpoint["model"])
        print("Resume checkpoint %s" % args.resume)
        if "optimizer" in checkpoint and "epoch" in checkpoint:
            optimizer.load_state_dict(checkpoint["optimizer"])
            if not isinstance(checkpoint["epoch"], str):        # 'best' / 'best-ema' cannot be resumed from
                args.start_epoch = checkpoint["epoch"] + 1
            else:
                assert getattr(args, "eval", False), "Does not support resuming with checkpoint-best"
            if getattr(args, "model_ema", False) and model_ema is not None:
                ema = model_ema.ema if hasattr(model_ema, "ema") else model_ema
                ema.load_state_dict(checkpoint["model_ema"] if "model_ema" in checkpoint else checkpoint["model"])
            if "scaler" in checkpoint and loss_scaler is not None and checkpoint["scaler"] is not None:
                loss_scaler.load_state_dict(checkpoint["scaler"])
            print("With optim & sched!")


def load_state_dict(model, state_dict, prefix="", ignore_missing="relative_position_index"):
    """Non-strict load that reports instead of raising: missing / unexpected keys are printed, keys containing one of the
    '|'-separated ignore_missing patterns are reported separately (utils/__init__.py:550-596).  Returns
    (missing, unexpected) besides printing."""
    sd = {k[len(prefix):] if prefix and k.startswith(prefix) else k: v for k, v in state_dict.items()}
    own = model.state_dict()
    error_msgs = [f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from checkpoint, the shape in current model is "
                  f"{tuple(own[k].shape)}." for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    loadable = {k: v for k, v in sd.items() if k not in own or tuple(own[k].shape) == tuple(v.shape)}
    res = model.load_state_dict(loadable, strict=False)
    missing = [k for k in res.missing_keys if not any(p in k for p in ignore_missing.split("|"))]
    ignored = [k for k in res.missing_keys if k not in missing]
    if missing:
        print("Weights of {} not initialized from pretrained model: {}".format(model.__class__.__name__, missing))
    if res.unexpected_keys:
        print("Weights from pretrained model not used in {}: {}".format(model.__class__.__name__, res.unexpected_keys))
    if ignored:
        print("Ignored weights of {} not initialized from pretrained model: {}".format(model.__class__.__name__, ignored))
    if error_msgs:
        print("\n".join(error_msgs))
    return missing, list(res.unexpected_keys)


def resample_abs_pos_embed(posemb, new_size, old_size=None, num_prefix_tokens=1, interpolation="bicubic", antialias=True):
    """Learned absolute position table [1, prefix + gh*gw, D] -> [1, prefix + new_h*new_w, D]: the grid part is resampled
    (bicubic, antialiased) in fp32, prefix tokens are carried over (timm resample_abs_pos_embed, the step
    checkpoint_filter_fn applies at models/vision_transformer.py:1441-1451; needed for a 224 -> 384 px fine-tune)."""
    import torch.nn.functional as F
    num_pos_tokens = posemb.shape[1]
    num_new_tokens = new_size[0] * new_size[1] + num_prefix_tokens
    if num_new_tokens == num_pos_tokens and new_size[0] == new_size[1]:
        return posemb
    if old_size is None:
        hw = int(math.sqrt(num_pos_tokens - num_prefix_tokens))
        old_size = hw, hw
    prefix, grid = (posemb[:, :num_prefix_tokens], posemb[:, num_prefix_tokens:]) if num_prefix_tokens else (None, posemb)
    embed_dim = grid.shape[-1]
    orig_dtype = grid.dtype
    grid = grid.float().reshape(1, old_size[0], old_size[1], -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=new_size, mode=interpolation, antialias=antialias)
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, embed_dim).to(orig_dtype)
    return torch.cat([prefix, grid], dim=1) if prefix is not None else grid


def load_finetune_checkpoint(model, path, model_key="model|module", model_prefix=""):
    """--finetune as main.py:653-674 does it (pick the state dict by model_key, drop a head of another size, non-strict
    load) plus the position-table resample the reference leaves out, so that a 224 px checkpoint initialises a 384 px
    model."""
    checkpoint = _load_checkpoint_file(path)
    print("Load ckpt from %s" % path)
    checkpoint_model = None
    for key in model_key.split("|"):
        if key in checkpoint:
            checkpoint_model = checkpoint[key]
            print("Load state_dict by model_key = %s" % key)
            break
    if checkpoint_model is None:
        checkpoint_model = checkpoint
    checkpoint_model = dict(checkpoint_model)
    state_dict = model.state_dict()
    for k in ["head.weight", "head.bias", "head_dist.weight", "head_dist.bias"]:
        if k in checkpoint_model and k in state_dict and checkpoint_model[k].shape != state_dict[k].shape:
            print(f"Removing key {k} from pretrained checkpoint")
            del checkpoint_model[k]
    pe = checkpoint_model.get("pos_embed")
    if pe is not None and "pos_embed" in state_dict and pe.shape != state_dict["pos_embed"].shape:
        grid = model.patch_embed.grid_size
        checkpoint_model["pos_embed"] = resample_abs_pos_embed(pe, (grid[0], grid[1]), num_prefix_tokens=model.num_prefix_tokens)
    return load_state_dict(model, checkpoint_model, prefix=model_prefix)
