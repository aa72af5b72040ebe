"""CPU restatement of the reference training / evaluation loop (test oracle; see oracle/__init__.py).

ref_train_one_epoch follows the fp32 branch of engine.train_one_epoch (engine.py:67-103 step bookkeeping and per-iteration
lr / weight-decay overwrite, :257-274 forward/backward/step, :282-299 metrics, :333 returned dict); ref_evaluate follows
engine.evaluate (:340-430).  Optimizer construction follows optim_factory.py:214-249 (AdamW, two parameter groups).
"""
from __future__ import annotations

import torch

from . import ref_ops as R


def make_optimizer(model, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
    groups = R.parameter_groups(model, weight_decay, model.no_weight_decay() if hasattr(model, "no_weight_decay") else ())
    for g in groups:
        g.pop("names")
    return torch.optim.AdamW(groups, lr=lr, weight_decay=0.0, betas=betas, eps=eps)


def ref_train_one_epoch(model, criterion, data_loader, optimizer, start_steps=0, lr_schedule_values=None,
                        wd_schedule_values=None, num_training_steps_per_epoch=None, update_freq=1, forward_kwargs=None,
                        record=None):
    """Returns {'loss', 'lr', 'class_acc'} global averages, like engine.py:333 with mixup off and log_freq=1."""
    model.train(True)
    optimizer.zero_grad()
    tot = {"loss": 0.0, "lr": 0.0, "class_acc": 0.0}
    cnt = {"loss": 0, "lr": 0, "class_acc": 0}
    fk = forward_kwargs or {}
    for data_iter_step, (samples, targets) in enumerate(data_loader):
        step = data_iter_step // update_freq
        if num_training_steps_per_epoch is not None and step >= num_training_steps_per_epoch:
            continue
        it = start_steps + step
        if (lr_schedule_values is not None or wd_schedule_values is not None) and data_iter_step % update_freq == 0:
            for group in optimizer.param_groups:                                   # engine.py:98-103
                if lr_schedule_values is not None:
                    group["lr"] = lr_schedule_values[it] * group.get("lr_scale", 1.0)
                if wd_schedule_values is not None and group.get("weight_decay", None) is not None:
                    group["weight_decay"] = wd_schedule_values[it]
        output = model(samples, **fk)                                              # engine.py:264-265
        loss = criterion(output, targets)
        loss_value = loss.item()
        loss = loss / update_freq                                                  # engine.py:268-274
        loss.backward()
        if record is not None:
            record(data_iter_step, output, loss_value, model)
        if (data_iter_step + 1) % update_freq == 0:
            optimizer.step()
            optimizer.zero_grad()
        out0 = output[0] if isinstance(output, tuple) else output
        class_acc = (out0.max(-1)[-1] == targets).float().mean().item()          # engine.py:283
        tot["loss"] += loss.item(); cnt["loss"] += 1                               # engine.py:294 logs loss/update_freq
        tot["class_acc"] += class_acc * samples.shape[0]; cnt["class_acc"] += samples.shape[0]
        tot["lr"] += optimizer.param_groups[0]["lr"]; cnt["lr"] += 1
    return {k: tot[k] / max(cnt[k], 1) for k in tot}


@torch.no_grad()
def ref_evaluate(data_loader, model, forward_kwargs=None):
    """engine.py:340-430: CrossEntropyLoss, top-1 / top-5 in percent; loss averaged per batch, accuracies per sample."""
    model.eval()
    fk = forward_kwargs or {}
    loss_t, loss_n, a1, a5, n = 0.0, 0, 0.0, 0.0, 0
    for batch in data_loader:
        images, target = batch[0], batch[-1]
        output = model(images, **fk)
        loss = torch.nn.functional.cross_entropy(output, target)
        acc1, acc5 = R.accuracy_topk(output, target, topk=(1, 5))
        bs = images.shape[0]
        loss_t += loss.item(); loss_n += 1
        a1 += acc1.item() * bs; a5 += acc5.item() * bs; n += bs
    return {"loss": loss_t / max(loss_n, 1), "acc1": a1 / max(n, 1), "acc5": a5 / max(n, 1)}
