"""train_one_epoch / evaluate with the reference's signatures (engine.py:19-24, :340), driving the HIP model.

Kept from the reference (engine.py): the update_freq bookkeeping (:84-88), the per-iteration overwrite of every param
group's lr (x lr_scale) and weight_decay from the schedules (:98-103), loss / update_freq before backward, optimizer
step + zero_grad + EMA update every update_freq micro-steps (:268-274), optional grad-norm clipping (:175-177), train
accuracy = argmax equality when mixup is off (:282-285), meters "loss", "lr", "class_acc" and the returned
{name: global_avg} after a cross-rank sync (:290-299, :328-333); evaluate(): CrossEntropyLoss, top-1 / top-5,
meters "loss", "acc1", "acc5" (:340-430).

Changed on purpose (MI355X-first, same results): `tpu`, `use_amp` and `loss_scaler` are accepted and ignored (the model
computes in bf16 with fp32 masters by construction; the reference's CUDA-AMP branch never trains, engine.py:109-118 vs
:245-255); device scalars (loss, accuracy) are collected on the GPU and fetched in batches instead of one
`.item()` host sync per step (engine.py:294-323); gradient averaging across ranks is the bucketed RCCL all-reduce
overlapped with backward (dp.py) in place of xm.optimizer_step (:185).
"""
from __future__ import annotations

import math
import os
import sys
from typing import Iterable, Optional

import torch

from . import ops, utils
from .losses import CrossEntropyLoss


def _core_model(model):
    m = model.module if hasattr(model, "module") else model
    return m.student if hasattr(m, "student") else m


def clip_grad_norm_(model, max_norm: float):
    """torch.nn.utils.clip_grad_norm_ (L2) over the flat gradient buffer: one sum-of-squares reduction and one
    scaling launch, the norm never leaves the GPU.  Returns the total norm as a 0-d GPU tensor."""
    core = _core_model(model)
    g = core._ensure_grad_buffer()
    if not hasattr(core, "_sumsq"):
        core._sumsq = torch.zeros(1, dtype=torch.float32, device=g.device)
    ops.sumsq(g, core._sumsq, ws=core._ws)
    ops.clip_grad_scale(g, core._sumsq, max_norm)
    return core._sumsq.sqrt()[0]


class GraphedStep:
    """forward + loss + backward of one training step captured ONCE in a HIP graph and replayed for every later step of the same
    shapes: the few hundred launches of a step then cost the host one call, which is what bounds the small workloads (ViT-Tiny at
    64 images: 6 ms of Python per step against 4 ms of kernels).  The reference has no counterpart (XLA compiles its step once
    per shape, engine.py:137-192, which plays the same role on TPU).

    What makes a step replayable: inputs are copied into fixed buffers; stochastic depth reads its per-sample factors from a fixed
    table that is re-drawn before every replay (VisionTransformer.begin_static_droppath: every branch is computed for every sample
    and scaled, as timm does, instead of the compact kept-samples batches whose shapes change per step); the optimizer step, EMA,
    clipping and metrics stay outside the graph (learning rate and weight decay change per step and are kernel arguments).
    Results are those of the eager step with VITK_DROPPATH_COMPACT=0, bit for bit.  Not used with a gradient reducer (collectives
    are not captured) or with gradient accumulation."""

    WARMUP = 2          # eager steps before the capture (allocator, lazy attributes, weight copies)
    total_replays = 0   # replays made by all instances (diagnostics, tests)

    def __init__(self, model, criterion):
        self.model, self.criterion = model, criterion
        self.core = _core_model(model)
        self.graph = None
        self.seen = 0
        self.x = self.y = self.output = self.loss = None

    @staticmethod
    def wanted(model, update_freq):
        return (os.environ.get("VITK_GRAPH_STEP") == "1" and update_freq == 1 and getattr(model, "reducer", None) is None
                and hasattr(_core_model(model), "begin_static_droppath"))

    def _fits(self, samples, targets):
        return (self.x is not None and samples.shape == self.x.shape and samples.dtype == self.x.dtype
                and targets.shape == self.y.shape and targets.dtype == self.y.dtype)

    def step(self, samples, targets):
        """-> (output, loss) of this step, gradients in place; eager for the first WARMUP steps and for odd shapes (a last,
        smaller batch)"""
        core = self.core
        if self.x is None:
            self.x, self.y = samples.clone(), targets.clone()
            core.begin_static_droppath(targets.shape[0], samples.device)      # samples may be a patch matrix: count the labels
        if not self._fits(samples, targets):
            output = self.model(samples)            # another batch size draws its own factors (forward checks the table's size)
            loss = self.criterion(output, targets)
            loss.backward()
            return output, loss
        self.x.copy_(samples)
        self.y.copy_(targets)
        core.refresh_static_droppath()
        if self.graph is None and self.seen >= self.WARMUP:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.output = self.model(self.x)
                self.loss = self.criterion(self.output, self.y)
                self.loss.backward()
            self.graph = graph
        self.seen += 1
        if self.graph is None:
            output = self.model(self.x)
            loss = self.criterion(output, self.y)
            loss.backward()
            return output, loss
        self.graph.replay()
        GraphedStep.total_replays += 1
        # a replay runs no Python: what the eager backward does on the host besides launching kernels is done here - every
        # parameter's .grad is the view of the flat gradient buffer the kernels just wrote, and the buffer counts as valid - so a
        # reader of p.grad (torch's clip_grad_norm_, a gradient logger, the per-parameter path of the optimizer) sees this step
        for n, p in zip(core._pnames, core._plist):
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != core._gview[n].data_ptr()):
                p.grad = core._gview[n]
        core._grad_valid = True
        # the graph's outputs are overwritten by the next replay: hand out copies
        output = tuple(o.clone() for o in self.output) if isinstance(self.output, tuple) else self.output.clone()
        return output, self.loss.clone()

    def pause(self):
        """leave the replayable mode (the model draws its own stochastic-depth factors again) but keep the captured graph and the
        table it reads: resume() continues with both in the next epoch"""
        self._parked = self.core._static_drop
        self.core.end_static_droppath()

    def resume(self):
        if getattr(self, "_parked", None) is not None:
            self.core._static_drop = self._parked        # the same table at the same address: the captured graph reads it
            self._parked = None

    def close(self):
        self.core.end_static_droppath()
        self.graph = None
        self._parked = None


def train_one_epoch(model: torch.nn.Module, criterion: torch.nn.Module, data_loader: Iterable,
                    optimizer: torch.optim.Optimizer, device: torch.device, epoch: int, loss_scaler=None, max_norm: float = 0,
                    model_ema=None, mixup_fn=None, log_writer=None, wandb_logger=None, start_steps=None,
                    lr_schedule_values=None, wd_schedule_values=None, num_training_steps_per_epoch=None, update_freq=None,
                    use_amp=False, tpu: bool = False, log_freq: int = 10):
    model.train(True)
    update_freq = update_freq or 1
    start_steps = start_steps or 0
    if log_writer is not None and num_training_steps_per_epoch is not None:
        log_writer.set_step(epoch * num_training_steps_per_epoch * update_freq)
    metric_logger = utils.MetricLogger(delimiter="  ")      # device scalars stay on the GPU until the meters are read
    header = f"Epoch: [{epoch}]"
    reducer = getattr(model, "reducer", None)
    optimizer.zero_grad()
    graphed = None
    if GraphedStep.wanted(model, update_freq):      # VITK_GRAPH_STEP=1: the captured step is kept on the model from epoch to epoch
        holder = _core_model(model)                 # (its deep copies - an EMA model - are rebuilt from scratch and do not inherit it)
        graphed = getattr(holder, "_vitk_graphed_step", None)
        if graphed is None or graphed.criterion is not criterion or graphed.model is not model:
            graphed = GraphedStep(model, criterion)
            holder._vitk_graphed_step = graphed
        else:
            graphed.resume()

    try:
        for data_iter_step, (samples, targets) in enumerate(data_loader):
            step = data_iter_step // update_freq
            if num_training_steps_per_epoch is not None and step >= num_training_steps_per_epoch:
                continue
            it = start_steps + step
            if (lr_schedule_values is not None or wd_schedule_values is not None) and data_iter_step % update_freq == 0:
                for param_group in optimizer.param_groups:
                    if lr_schedule_values is not None:
                        param_group["lr"] = lr_schedule_values[it] * param_group.get("lr_scale", 1.0)
                    if wd_schedule_values is not None and param_group.get("weight_decay", None) is not None:
                        param_group["weight_decay"] = wd_schedule_values[it]

            if mixup_fn is not None:
                samples, targets = mixup_fn(samples, targets)
            samples = samples.to(device, non_blocking=True)
            targets = targets.to(device, non_blocking=True)

            last_micro = (data_iter_step + 1) % update_freq == 0
            if reducer is not None:
                reducer.enabled = last_micro          # gradient accumulation: only the last micro-step communicates
            if graphed is not None:
                output, loss = graphed.step(samples, targets)
            else:
                output = model(samples)
                loss = criterion(output, targets)
                if update_freq > 1:
                    loss = loss / update_freq
                loss.backward()
            if last_micro:
                if reducer is not None:
                    reducer.wait()
                if max_norm is not None and max_norm > 0:
                    clip_grad_norm_(model, max_norm)
                optimizer.step()
                optimizer.zero_grad()
                if model_ema is not None:
                    model_ema.update(model)

            class_acc = None
            if mixup_fn is None:
                out0 = output[0] if isinstance(output, tuple) else output
                class_acc = (ops.argmax_rows(out0.detach().float()) == targets).float().mean()

            if data_iter_step % log_freq == 0 or data_iter_step < 5:
                metric_logger.update(loss=loss.detach())
                if class_acc is not None:
                    metric_logger.meters["class_acc"].update(class_acc, n=targets.size(0))
                metric_logger.update(lr=optimizer.param_groups[0]["lr"])
            if log_writer is not None:          # a per-step host log is the caller's choice: it costs one sync per step
                lv = metric_logger.meters["loss"].value if metric_logger.meters["loss"].count else float("nan")
                log_writer.update(loss=lv, lr=optimizer.param_groups[0]["lr"])
            # the experiment tracker gets the same record at the console's frequency (engine.py:311-323): the keys and the
            # step numbering of the reference; reading the scalars is the one host sync a caller who passes a logger asks for
            if wandb_logger is not None and (data_iter_step % log_freq == 0 or data_iter_step < 5):
                record = {"train/loss": float(loss.detach()), "train/learning_rate": optimizer.param_groups[0]["lr"], "train/epoch": epoch,
                          "train/step": start_steps + data_iter_step if start_steps else data_iter_step}
                if class_acc is not None:
                    record["train/accuracy"] = float(class_acc)
                wandb_logger._wandb.log(record)

    finally:
        if graphed is not None:       # also when the loop raises: the model must not keep reading the static DropPath table
            graphed.pause()
    metric_logger.flush()
    if "loss" in metric_logger.meters and not math.isfinite(metric_logger.meters["loss"].global_avg):
        print("Loss is {}, stopping training".format(metric_logger.meters["loss"].global_avg))
        sys.exit(1)
    metric_logger.synchronize_between_processes()
    if utils.is_main_process():
        print(header, "Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def evaluate(data_loader, model, device, use_amp=False, tpu: bool = False):
    criterion = CrossEntropyLoss()
    metric_logger = utils.MetricLogger(delimiter="  ")
    model.eval()
    for batch in data_loader:
        images, target = batch[0], batch[-1]
        images = images.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        output = model(images)
        loss = criterion(output, target)
        acc1, acc5 = utils.accuracy(output, target, topk=(1, 5))
        bs = target.shape[0]
        metric_logger.update(loss=loss)
        metric_logger.meters["acc1"].update(acc1, n=bs)
        metric_logger.meters["acc5"].update(acc5, n=bs)
    metric_logger.flush()
    metric_logger.synchronize_between_processes()
    if utils.is_main_process() and metric_logger.meters:
        print("* Acc@1 {top1.global_avg:.3f} Acc@5 {top5.global_avg:.3f} loss {losses.global_avg:.3f}".format(
            top1=metric_logger.meters["acc1"], top5=metric_logger.meters["acc5"], losses=metric_logger.meters["loss"]))
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
