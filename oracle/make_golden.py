#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle (test infrastructure).

The reference cannot be imported here (its hot path needs the un-vendored `timm`; see oracle/__init__.py), and it
ships no golden vectors of its own, so these fixtures pin the ORACLE (regression guard + the vectors the GPU parity
tests and `smoke()` compare against), not the reference: parity stays "unpinned" with respect to upstream timm.

    python -m oracle.make_golden            # rewrites tests/golden/*.npz deterministically (CPU, fp32, seeds fixed)

Contents (all float32 unless noted):
  vit_tiny_d2_step.npz   2-block ViT-Tiny/16, 40 classes, global_pool='avg', batch 4, seed-0 inputs:
                         x [4,3,224,224] is NOT stored (regenerated from the seed: torch.Generator().manual_seed(1));
                         stores targets, logits (fp32 path), logits_emul (bf16-emulating path), loss, loss_emul,
                         per-parameter gradient norms and three full small gradients, weights-after-one-AdamW-step digests
  vit_tiny_d2_train3.npz the same model through the loop restatement (engine.py:257-299) for 3 consecutive optimizer steps on batches
                         seeded 11, 12, 13 (B = 4), AdamW lr 1e-3 -> cosine schedule with one warm-up epoch, wd 0.05: per step logits
                         and loss (fp32 and bf16-emulating paths), per-parameter gradient norms of step 1, the two parameter groups'
                         member names, the returned {loss, lr, class_acc}, first / last values of both schedules, weight norms at the end
  schedules.npz          cosine_scheduler lr / wd values (utils/__init__.py:667-684 semantics) for two configurations
  ops.npz                single-op vectors: LayerNorm, exact GELU, attention (3 heads x 17 tokens) forward and backward,
                         label-smoothing / soft-target / KD losses with their logits gradients
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from oracle import ref_engine, ref_ops as R   # noqa: E402
from oracle.ref_vit import create_ref_model   # noqa: E402

OUT = ROOT / "tests" / "golden"


def tiny_model(seed=0):
    torch.manual_seed(seed)
    return create_ref_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", drop_path_rate=0.0, depth=2)


def tiny_batch(seed=1, B=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, 224, 224, generator=g), torch.randint(0, 40, (B,), generator=g)


def gen_step():
    torch.set_num_threads(1)
    out = {}
    x, y = tiny_batch()
    out["targets"] = y.numpy()
    for tag, emu in (("", False), ("_emul", True)):
        m = tiny_model()
        m.train()
        logits = m(x, emulate=emu)
        loss = R.label_smoothing_ce(logits, y, 0.1)
        loss.backward()
        out["logits" + tag] = logits.detach().numpy()
        out["loss" + tag] = np.float32(loss.item())
        names = [n for n, _ in m.named_parameters()]
        out["grad_norms" + tag] = np.array([p.grad.norm().item() for _, p in m.named_parameters()], dtype=np.float64)
        for n in ("head.bias", "fc_norm.weight", "blocks.1.attn.proj.bias"):
            out["grad:" + n + tag] = dict(m.named_parameters())[n].grad.numpy().copy()
        if not emu:
            out["param_names"] = np.array(names)
            opt = ref_engine.make_optimizer(m, 1e-3, 0.05)
            opt.step()
            out["weights_after_step_norms"] = np.array([p.detach().norm().item() for _, p in m.named_parameters()], dtype=np.float64)
            out["head_weight_after_step_row0"] = m.head.weight.detach()[0].numpy().copy()
    np.savez_compressed(OUT / "vit_tiny_d2_step.npz", **out)


TRAIN3_SEEDS = (11, 12, 13)


def train3_schedules():
    return (R.cosine_scheduler(1e-3, 1e-6, 2, 3, warmup_epochs=1, start_warmup_value=1e-4), R.cosine_scheduler(0.05, 0.02, 2, 3))


def gen_train3():
    torch.set_num_threads(1)
    out = {}
    data = [tiny_batch(seed=s) for s in TRAIN3_SEEDS]
    lr_sched, wd_sched = train3_schedules()
    out["lr_schedule"], out["wd_schedule"] = lr_sched, wd_sched
    out["targets"] = np.stack([y.numpy() for _, y in data])
    for tag, emu in (("", False), ("_emul", True)):
        m = tiny_model()
        opt = ref_engine.make_optimizer(m, 1e-3, 0.05)
        logits, losses, gnorms = [], [], []

        def record(i, output, loss_value, model):
            logits.append(output.detach().numpy().copy())
            losses.append(loss_value)
            if i == 0:
                gnorms.extend(p.grad.norm().item() for p in model.parameters())
        stats = ref_engine.ref_train_one_epoch(m, lambda o, t: R.label_smoothing_ce(o, t, 0.1), data, opt, 0, lr_sched, wd_sched, 3, 1,
                                               forward_kwargs=dict(emulate=emu), record=record)
        out["logits" + tag] = np.stack(logits)
        out["losses" + tag] = np.array(losses, dtype=np.float64)
        out["grad_norms_step1" + tag] = np.array(gnorms, dtype=np.float64)
        out["stats" + tag] = np.array([stats["loss"], stats["lr"], stats["class_acc"]], dtype=np.float64)
        out["weight_norms_end" + tag] = np.array([p.detach().norm().item() for p in m.parameters()], dtype=np.float64)
        if not emu:
            groups = R.parameter_groups(m, 0.05, m.no_weight_decay())
            out["param_names"] = np.array([n for n, _ in m.named_parameters()])
            out["group_weight_decay"] = np.array([g["weight_decay"] for g in groups])
            for gi, g in enumerate(groups):
                out[f"group{gi}_names"] = np.array(g["names"])
    np.savez_compressed(OUT / "vit_tiny_d2_train3.npz", **out)


def gen_schedules():
    np.savez_compressed(
        OUT / "schedules.npz",
        lr_a=R.cosine_scheduler(4e-3, 1e-6, 3, 7, warmup_epochs=1),
        wd_a=R.cosine_scheduler(0.05, 0.05, 3, 7),
        lr_b=R.cosine_scheduler(1e-3, 1e-5, 5, 4, warmup_epochs=2, warmup_steps=3),
        wd_b=R.cosine_scheduler(0.05, 0.1, 5, 4),
    )


def gen_ops():
    g = torch.Generator().manual_seed(7)
    out = {}
    x = torch.randn(5, 192, generator=g)
    w, b = torch.randn(192, generator=g), torch.randn(192, generator=g)
    out["ln_x"], out["ln_w"], out["ln_b"] = x.numpy(), w.numpy(), b.numpy()
    out["ln_y"] = R.layernorm(x, w, b).numpy()
    gx = torch.linspace(-6, 6, 193)
    out["gelu_x"], out["gelu_y"] = gx.numpy(), R.gelu(gx).numpy()
    qkv = torch.randn(2, 17, 3 * 3 * 64, generator=g).requires_grad_(True)
    do = torch.randn(2, 17, 3 * 64, generator=g)
    o = R.attention(qkv, 3)
    o.backward(do)
    out["attn_qkv"], out["attn_do"], out["attn_o"], out["attn_dqkv"] = qkv.detach().numpy(), do.numpy(), o.detach().numpy(), qkv.grad.numpy()
    logits = (torch.randn(6, 40, generator=g) * 3).requires_grad_(True)
    tgt = torch.randint(0, 40, (6,), generator=g)
    soft = torch.softmax(torch.randn(6, 40, generator=g), -1)
    teacher = torch.randn(6, 40, generator=g) * 2
    out["loss_logits"], out["loss_targets"], out["loss_soft"], out["loss_teacher"] = logits.detach().numpy(), tgt.numpy(), soft.numpy(), teacher.numpy()
    for name, fn in (("ls", lambda z: R.label_smoothing_ce(z, tgt, 0.1)), ("ce", lambda z: R.label_smoothing_ce(z, tgt, 0.0)),
                     ("soft", lambda z: R.soft_target_ce(z, soft)),
                     ("kd", lambda z: R.distillation_loss(z, teacher, tgt, lambda a, t: R.label_smoothing_ce(a, t, 0.1), 0.7, 4.0))):
        logits.grad = None
        v = fn(logits)
        v.backward()
        out[f"loss_{name}"], out[f"loss_{name}_grad"] = np.float32(v.item()), logits.grad.numpy().copy()
    np.savez_compressed(OUT / "ops.npz", **out)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    gen_step()
    gen_train3()
    gen_schedules()
    gen_ops()
    for f in sorted(OUT.glob("*.npz")):
        print(f"{f.relative_to(ROOT)}  {f.stat().st_size / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
