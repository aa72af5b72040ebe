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


class RefStudentWithDistillation(torch.nn.Module):
    """main.py:836-851 (duplicated at test_kd.py:43-58): in train mode (student(x), teacher(x) under no_grad), in eval mode
    student(x).  forward_kwargs reach both models (the oracle's emulate= switch)."""

    def __init__(self, student_model, teacher_model):
        super().__init__()
        self.student = student_model
        self.teacher = teacher_model

    def forward(self, x, **kw):
        student_logits = self.student(x, **kw)
        if self.training and self.teacher is not None:
            with torch.no_grad():
                teacher_logits = self.teacher(x, **kw)
            return student_logits, teacher_logits
        return student_logits


class RefDistillationLoss(torch.nn.Module):
    """main.py:939-970: (1 - alpha) * base(student, y) + alpha * T^2 * KLDiv_batchmean(log_softmax(s/T), softmax(t/T)) for a
    (student, teacher) tuple, base(outputs, y) for a plain tensor."""

    def __init__(self, base_criterion, alpha=0.7, temperature=4.0):
        super().__init__()
        self.base_criterion, self.alpha, self.temperature = base_criterion, alpha, temperature

    def forward(self, outputs, targets):
        if isinstance(outputs, tuple):
            return R.distillation_loss(outputs[0], outputs[1], targets, self.base_criterion, self.alpha, self.temperature)
        return self.base_criterion(outputs, targets)


class RefModelEma:
    """timm.utils.ModelEma as engine.py:198-202,273-274 uses it: a deep copy in eval mode whose every state-dict entry
    becomes decay * ema + (1 - decay) * model after each optimizer step."""

    def __init__(self, model, decay=0.9999):
        import copy
        self.ema = copy.deepcopy(model)
        self.ema.eval()
        self.decay = decay
        for p in self.ema.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def update(self, model):
        msd = model.state_dict()
        for k, ema_v in self.ema.state_dict().items():
            ema_v.copy_(ema_v * self.decay + (1.0 - self.decay) * msd[k].detach())


def ref_train_one_epoch(model, criterion, data_loader, optimizer, start_steps=0, lr_schedule_values=None,
                        wd_schedule_values=None, num_training_steps_per_epoch=None, update_freq=1, forward_kwargs=None,
                        record=None, max_norm=0, model_ema=None):
    """Returns {'loss', 'lr', 'class_acc'} global averages, like engine.py:333 with mixup off and log_freq=1.
    max_norm > 0 clips the total gradient norm before the optimizer step (engine.py:175-177); model_ema.update(model)
    follows every optimizer step (engine.py:198-202, 273-274)."""
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
            if max_norm is not None and max_norm > 0:                              # engine.py:175-177
                torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], max_norm)
            optimizer.step()
            optimizer.zero_grad()
            if model_ema is not None:                                              # engine.py:198-202, 273-274
                model_ema.update(model)
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
