"""CPU tests of the oracle itself.

The reference pins no numeric result for this path and cannot be imported here (un-vendored timm), so the oracle is
pinned two ways: (1) against INDEPENDENT implementations available in this container - a torch.nn composition with
autograd / F.scaled_dot_product_attention, torch's own label-smoothing cross entropy, torch.optim.AdamW, and the Hugging
Face `transformers` ViT built from a local config with the oracle's weights copied in; (2) against the committed golden
fixtures (tests/golden/, written by oracle/make_golden.py) as a regression guard."""
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import ref_engine, ref_ops as R
from oracle.make_golden import tiny_batch, tiny_model
from oracle.ref_vit import create_ref_model
from tests.util import rel_l2

GOLD = Path(__file__).resolve().parent / "golden"


# ---------------------------------------------------------------------------------- independent implementations
def _plain_block(blk, x, heads):
    """a pre-norm block written with stock torch functionals only (no oracle code), on the oracle's parameters"""
    B, N, D = x.shape
    y = F.layer_norm(x, (D,), blk.norm1.weight, blk.norm1.bias, 1e-6)
    q, k, v = F.linear(y, blk.attn.qkv.weight, blk.attn.qkv.bias).reshape(B, N, 3, heads, D // heads).permute(2, 0, 3, 1, 4).unbind(0)
    a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, D)
    x = x + F.linear(a, blk.attn.proj.weight, blk.attn.proj.bias)
    y = F.layer_norm(x, (D,), blk.norm2.weight, blk.norm2.bias, 1e-6)
    return x + F.linear(F.gelu(F.linear(y, blk.mlp.fc1.weight, blk.mlp.fc1.bias)), blk.mlp.fc2.weight, blk.mlp.fc2.bias)


def _plain_forward(ref, x):
    """forward of the oracle's weights through stock torch functionals (conv2d, layer_norm, linear, SDPA, gelu)"""
    D = ref.embed_dim
    t = F.conv2d(x, ref.patch_embed.proj.weight, ref.patch_embed.proj.bias, stride=16).flatten(2).transpose(1, 2)
    t = torch.cat([ref.cls_token.expand(x.shape[0], -1, -1), t], 1) + ref.pos_embed
    for blk in ref.blocks:
        t = _plain_block(blk, t, ref.num_heads)
    if ref.global_pool == "avg":
        f = F.layer_norm(t[:, 1:].mean(1), (D,), ref.fc_norm.weight, ref.fc_norm.bias, 1e-6)
    else:
        f = F.layer_norm(t, (D,), ref.norm.weight, ref.norm.bias, 1e-6)[:, 0]
    return F.linear(f, ref.head.weight, ref.head.bias)


@pytest.mark.parametrize("pool", ["avg", "token"])
def test_oracle_matches_stock_torch_composition(pool):
    torch.manual_seed(0)
    ref = create_ref_model("vit_tiny_patch16_224", num_classes=40, global_pool=pool, depth=2)
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))
    x, y = tiny_batch()
    ref.train()
    out = ref(x)
    want = _plain_forward(ref, x)
    assert rel_l2(out, want) < 2e-6
    # gradients: oracle's explicit attention backward vs autograd through SDPA
    g1 = torch.autograd.grad(R.label_smoothing_ce(out, y, 0.1), list(ref.parameters()))
    g2 = torch.autograd.grad(F.cross_entropy(want, y, label_smoothing=0.1), list(ref.parameters()))
    for (n, _), a, b in zip(ref.named_parameters(), g1, g2):
        assert rel_l2(a, b) < 2e-5, n


def _hf_config(cls, depth=2, **kw):
    return cls(hidden_size=192, num_hidden_layers=depth, num_attention_heads=3, intermediate_size=768, hidden_act="gelu",
               hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-6, image_size=224, patch_size=16,
               num_channels=3, qkv_bias=True, **kw)


def _copy_into_hf(hf, ref, prefix=""):
    """the oracle's weights into a Hugging Face ViT / DeiT encoder (module names of transformers 4 and 5).  Returns
    {oracle parameter name: [hf parameter names whose gradients, concatenated along dim 0, are its gradient]}."""
    sd = hf.state_dict()
    D = ref.embed_dim
    back = {}

    def put(k, v, name, part=False):
        k = prefix + k
        assert k in sd and sd[k].shape == v.shape, (k, tuple(v.shape))
        sd[k] = v.detach().clone()
        back.setdefault(name, []).append(k)

    put("embeddings.cls_token", ref.cls_token, "cls_token")
    put("embeddings.position_embeddings", ref.pos_embed, "pos_embed")
    if ref.distilled:
        put("embeddings.distillation_token", ref.dist_token, "dist_token")
    put("embeddings.patch_embeddings.projection.weight", ref.patch_embed.proj.weight, "patch_embed.proj.weight")
    put("embeddings.patch_embeddings.projection.bias", ref.patch_embed.proj.bias, "patch_embed.proj.bias")
    new_names = prefix + "layers.0.attention.q_proj.weight" in sd       # transformers >= 5 renamed the sub-modules
    for i, blk in enumerate(ref.blocks):
        if new_names:
            p = f"layers.{i}."
            names = dict(q="attention.q_proj", k="attention.k_proj", v="attention.v_proj", o="attention.o_proj", fc1="mlp.fc1", fc2="mlp.fc2")
        else:
            p = f"encoder.layer.{i}."
            names = dict(q="attention.attention.query", k="attention.attention.key", v="attention.attention.value",
                         o="attention.output.dense", fc1="intermediate.dense", fc2="output.dense")
        b = f"blocks.{i}."
        for nm, w_, b_ in zip("qkv", blk.attn.qkv.weight.split(D, 0), blk.attn.qkv.bias.split(D, 0)):      # rows [q ; k ; v]
            put(p + names[nm] + ".weight", w_, b + "attn.qkv.weight"); put(p + names[nm] + ".bias", b_, b + "attn.qkv.bias")
        put(p + names["o"] + ".weight", blk.attn.proj.weight, b + "attn.proj.weight"); put(p + names["o"] + ".bias", blk.attn.proj.bias, b + "attn.proj.bias")
        put(p + "layernorm_before.weight", blk.norm1.weight, b + "norm1.weight"); put(p + "layernorm_before.bias", blk.norm1.bias, b + "norm1.bias")
        put(p + "layernorm_after.weight", blk.norm2.weight, b + "norm2.weight"); put(p + "layernorm_after.bias", blk.norm2.bias, b + "norm2.bias")
        put(p + names["fc1"] + ".weight", blk.mlp.fc1.weight, b + "mlp.fc1.weight"); put(p + names["fc1"] + ".bias", blk.mlp.fc1.bias, b + "mlp.fc1.bias")
        put(p + names["fc2"] + ".weight", blk.mlp.fc2.weight, b + "mlp.fc2.weight"); put(p + names["fc2"] + ".bias", blk.mlp.fc2.bias, b + "mlp.fc2.bias")
    if hasattr(ref, "norm"):
        put("layernorm.weight", ref.norm.weight, "norm.weight"); put("layernorm.bias", ref.norm.bias, "norm.bias")
    hf.load_state_dict(sd)
    return back


def _perturbed(name, pool, seed, **kw):
    torch.manual_seed(seed)
    ref = create_ref_model(name, num_classes=40, global_pool=pool, depth=2, **kw)
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))          # biases, tokens and norm offsets away from their zero / one init
    return ref


def _check_grads_against_hf(ref, hf, back, skip=()):
    hfp = dict(hf.named_parameters())
    worst = 0.0
    for n, p in ref.named_parameters():
        if n in skip:
            continue
        want = torch.cat([hfp[k].grad for k in back[n]], 0)
        e = rel_l2(p.grad, want)
        worst = max(worst, e)
        assert e < 5e-5, (n, e)
    return worst


def test_oracle_matches_huggingface_vit():
    """forward AND every parameter gradient of the oracle ('token' pooling) against transformers' ViT with the same weights"""
    pytest.importorskip("transformers")
    from transformers import ViTConfig, ViTModel
    ref = _perturbed("vit_tiny_patch16_224", "token", 1)
    hf = ViTModel(_hf_config(ViTConfig), add_pooling_layer=False).train()
    back = _copy_into_hf(hf, ref)
    x, y = tiny_batch()
    ref.train()
    feats = hf(pixel_values=x).last_hidden_state[:, 0]
    want = F.linear(feats, ref.head.weight, ref.head.bias)           # the classifier is ours on both sides: its gradient ...
    got = ref(x)
    assert rel_l2(got, want) < 5e-6
    g_head = torch.autograd.grad(F.cross_entropy(want, y, label_smoothing=0.1), [ref.head.weight, ref.head.bias], retain_graph=True)
    F.cross_entropy(want, y, label_smoothing=0.1).backward()
    ref.zero_grad()
    R.label_smoothing_ce(got, y, 0.1).backward()
    assert rel_l2(ref.head.weight.grad, g_head[0]) < 5e-5 and rel_l2(ref.head.bias.grad, g_head[1]) < 5e-5   # ... is compared too
    worst = _check_grads_against_hf(ref, hf, back, skip=("head.weight", "head.bias"))
    print(f"oracle vs Hugging Face ViT: logits {rel_l2(got, want):.1e}, worst parameter gradient {worst:.1e}")


def test_oracle_avg_pool_matches_huggingface_composition():
    """global_pool='avg' (no final norm on the tokens, mean over the patch tokens, fc_norm, head: vision_transformer.py:529,603,616,
    953-990) against transformers' encoder stack composed with stock functionals"""
    pytest.importorskip("transformers")
    from transformers import ViTConfig, ViTModel
    ref = _perturbed("vit_tiny_patch16_224", "avg", 2)
    hf = ViTModel(_hf_config(ViTConfig), add_pooling_layer=False).train()
    back = _copy_into_hf(hf, ref)
    x, y = tiny_batch()
    ref.train()
    last = hf(pixel_values=x, output_hidden_states=True).hidden_states[-1]        # the last block's output, before HF's final LayerNorm
    pooled = F.layer_norm(last[:, 1:].mean(1), (192,), ref.fc_norm.weight, ref.fc_norm.bias, 1e-6)
    want = F.linear(pooled, ref.head.weight, ref.head.bias)
    got = ref(x)
    assert rel_l2(got, want) < 5e-6
    tail = [ref.fc_norm.weight, ref.fc_norm.bias, ref.head.weight, ref.head.bias]
    g_tail = torch.autograd.grad(F.cross_entropy(want, y, label_smoothing=0.1), tail, retain_graph=True)
    F.cross_entropy(want, y, label_smoothing=0.1).backward()
    ref.zero_grad()
    R.label_smoothing_ce(got, y, 0.1).backward()
    for p, g in zip(tail, g_tail):
        assert rel_l2(p.grad, g) < 5e-5
    _check_grads_against_hf(ref, hf, back, skip=("fc_norm.weight", "fc_norm.bias", "head.weight", "head.bias"))


def test_distilled_oracle_matches_huggingface_deit_with_teacher():
    """the distilled variant (models/deit.py:28-119: dist_token after cls_token, 198 positions, two heads, their mean at inference,
    both in distilled training) against transformers' DeiTForImageClassificationWithTeacher: logits of all three outputs and the
    gradients of a loss on both heads"""
    pytest.importorskip("transformers")
    from transformers import DeiTConfig, DeiTForImageClassificationWithTeacher
    ref = _perturbed("deit_tiny_distilled_patch16_224", "token", 3)
    hf = DeiTForImageClassificationWithTeacher(_hf_config(DeiTConfig, num_labels=40)).train()
    back = _copy_into_hf(hf, ref, prefix="deit.")
    sd = hf.state_dict()
    for k, v in (("cls_classifier.weight", ref.head.weight), ("cls_classifier.bias", ref.head.bias),
                 ("distillation_classifier.weight", ref.head_dist.weight), ("distillation_classifier.bias", ref.head_dist.bias)):
        assert sd[k].shape == v.shape
        sd[k] = v.detach().clone()
    hf.load_state_dict(sd)
    back.update({"head.weight": ["cls_classifier.weight"], "head.bias": ["cls_classifier.bias"],
                 "head_dist.weight": ["distillation_classifier.weight"], "head_dist.bias": ["distillation_classifier.bias"]})
    x, y = tiny_batch()
    ref.eval()
    out = hf(pixel_values=x)
    with torch.no_grad():
        assert rel_l2(ref(x), out.logits) < 5e-6                      # inference: the mean of the two heads
    ref.train(); ref.set_distilled_training(True)
    a, b = ref(x)
    assert rel_l2(a, out.cls_logits) < 5e-6 and rel_l2(b, out.distillation_logits) < 5e-6
    ref.zero_grad()
    (R.label_smoothing_ce(a, y, 0.1) + 0.5 * R.label_smoothing_ce(b, y.flip(0), 0.0)).backward()
    (F.cross_entropy(out.cls_logits, y, label_smoothing=0.1) + 0.5 * F.cross_entropy(out.distillation_logits, y.flip(0))).backward()
    worst = _check_grads_against_hf(ref, hf, back)
    print(f"distilled oracle vs Hugging Face DeiT: worst parameter gradient {worst:.1e}")
    ref.set_distilled_training(False)                                # training without the distillation flag: still the mean
    assert rel_l2(ref(x), out.logits) < 5e-6


def test_attention_explicit_backward_matches_autograd():
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(2, 37, 3 * 3 * 64, generator=g, dtype=torch.float64).requires_grad_(True)
    do = torch.randn(2, 37, 3 * 64, generator=g, dtype=torch.float64)
    o = R.attention(qkv, 3)
    o.backward(do)
    g1 = qkv.grad.clone(); qkv.grad = None
    q, k, v = qkv.reshape(2, 37, 3, 3, 64).permute(2, 0, 3, 1, 4)
    o2 = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2, 37, 192)
    o2.backward(do)
    assert (o - o2).abs().max() < 1e-12 and (g1 - qkv.grad).abs().max() < 1e-12


def test_losses_match_torch():
    g = torch.Generator().manual_seed(2)
    z = torch.randn(9, 1000, generator=g) * 3
    t = torch.randint(0, 1000, (9,), generator=g)
    assert abs(R.label_smoothing_ce(z, t, 0.1) - F.cross_entropy(z, t, label_smoothing=0.1)) < 1e-5
    assert abs(R.label_smoothing_ce(z, t, 0.0) - F.cross_entropy(z, t)) < 1e-5
    soft = torch.softmax(torch.randn(9, 1000, generator=g), -1)
    assert abs(R.soft_target_ce(z, soft) - F.cross_entropy(z, soft)) < 1e-5
    onehot = F.one_hot(t, 1000).float()
    assert abs(R.soft_target_ce(z, onehot) - F.cross_entropy(z, t)) < 1e-5
    top1, top5 = R.accuracy_topk(z, t, (1, 5))
    assert abs(top1 - (z.argmax(-1) == t).float().mean() * 100) < 1e-6 and top5 >= top1


def test_adamw_restatement_matches_torch():
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(1000, generator=g)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=3e-3, weight_decay=0.05)
    q, m, v = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    for step in range(1, 4):
        grad = torch.randn(1000, generator=g)
        p.grad = grad.clone(); opt.step()
        q, m, v = R.adamw_reference_step(q, grad, m, v, 3e-3, 0.9, 0.999, 1e-8, 0.05, step)
    assert rel_l2(q, p.detach()) < 1e-6


def test_emulation_stays_at_bf16_noise_floor():
    m = tiny_model()
    x, _ = tiny_batch()
    m.eval()
    with torch.no_grad():
        a, b = m(x), m(x, emulate=True)
    assert 1e-5 < rel_l2(b, a) < 2e-2


def test_parameter_counts_and_groups():
    # models/vision_transformer.py entrypoints with main.py's kwargs: 5 717 416 / 86 567 656 / 304 326 632 parameters
    for name, n in (("vit_tiny_patch16_224", 5717416), ("vit_base_patch16_224", 86567656)):
        m = create_ref_model(name, num_classes=1000, global_pool="avg")
        assert sum(p.numel() for p in m.parameters()) == n
        groups = R.parameter_groups(m, 0.05, m.no_weight_decay())
        assert [g["weight_decay"] for g in groups] == [0.0, 0.05]          # first-seen order: cls_token is no_decay
        nd = set(groups[0]["names"])
        assert {"cls_token", "pos_embed", "fc_norm.weight", "head.bias", "blocks.0.attn.qkv.bias"} <= nd
        assert "blocks.0.attn.qkv.weight" in groups[1]["names"] and "patch_embed.proj.weight" in groups[1]["names"]
        assert len(groups[0]["names"]) + len(groups[1]["names"]) == len(list(m.parameters()))


def test_train_loop_restatement_update_freq():
    """update_freq=2: two micro-batches per optimizer step, loss/2 each, equals one step on the concatenated batch"""
    x, y = tiny_batch(B=4)
    crit = lambda o, t: R.label_smoothing_ce(o, t, 0.1)
    m1, m2 = tiny_model(), tiny_model()
    o1, o2 = ref_engine.make_optimizer(m1, 1e-3, 0.05), ref_engine.make_optimizer(m2, 1e-3, 0.05)
    r1 = ref_engine.ref_train_one_epoch(m1, crit, [(x[:2], y[:2]), (x[2:], y[2:])], o1, update_freq=2, num_training_steps_per_epoch=1)
    r2 = ref_engine.ref_train_one_epoch(m2, crit, [(x, y)], o2, update_freq=1, num_training_steps_per_epoch=1)
    assert set(r1) == {"loss", "lr", "class_acc"}
    assert abs(2 * r1["loss"] - r2["loss"]) < 1e-5            # engine.py:294 logs loss / update_freq
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert rel_l2(a, b) < 1e-3     # first AdamW step is ~lr * sign(g): fp32 summation order decides near-zero gradients


# ----------------------------------------------------------------------------------------------- golden fixtures
def test_golden_step_fixture():
    gold = np.load(GOLD / "vit_tiny_d2_step.npz")
    torch.set_num_threads(1)
    x, y = tiny_batch()
    assert np.array_equal(gold["targets"], y.numpy())
    for tag, emu in (("", False), ("_emul", True)):
        m = tiny_model(); m.train()
        logits = m(x, emulate=emu)
        loss = R.label_smoothing_ce(logits, y, 0.1)
        loss.backward()
        assert rel_l2(logits, torch.from_numpy(gold["logits" + tag])) < (1e-5 if not emu else 5e-3)
        assert abs(loss.item() - float(gold["loss" + tag])) < (1e-5 if not emu else 5e-3) * abs(loss.item())
        norms = np.array([p.grad.norm().item() for _, p in m.named_parameters()])
        assert np.allclose(norms, gold["grad_norms" + tag], rtol=(1e-4 if not emu else 2e-2))
        for n in ("head.bias", "fc_norm.weight", "blocks.1.attn.proj.bias"):
            assert rel_l2(dict(m.named_parameters())[n].grad, torch.from_numpy(gold["grad:" + n + tag])) < (1e-4 if not emu else 2e-2)
    assert list(gold["param_names"]) == [n for n, _ in tiny_model().named_parameters()]


def test_golden_three_step_fixture():
    """the loop restatement reproduces the committed 3-step run (regression guard for oracle/ref_engine.py + ref_vit.py)"""
    from oracle.make_golden import TRAIN3_SEEDS, train3_schedules
    gold = np.load(GOLD / "vit_tiny_d2_train3.npz")
    torch.set_num_threads(1)
    data = [tiny_batch(seed=s) for s in TRAIN3_SEEDS]
    lr_sched, wd_sched = train3_schedules()
    assert np.array_equal(gold["lr_schedule"], lr_sched) and np.array_equal(gold["wd_schedule"], wd_sched)
    m = tiny_model()
    groups = R.parameter_groups(m, 0.05, m.no_weight_decay())
    assert [list(gold[f"group{i}_names"]) for i in range(2)] == [g["names"] for g in groups]
    assert list(gold["group_weight_decay"]) == [g["weight_decay"] for g in groups]
    losses = []
    stats = ref_engine.ref_train_one_epoch(m, lambda o, t: R.label_smoothing_ce(o, t, 0.1), data, ref_engine.make_optimizer(m, 1e-3, 0.05), 0,
                                           lr_sched, wd_sched, 3, 1, record=lambda i, o, l, mod: losses.append(l))
    assert np.allclose(losses, gold["losses"], rtol=1e-5)
    assert np.allclose([stats["loss"], stats["lr"], stats["class_acc"]], gold["stats"], rtol=1e-5)
    assert np.allclose([p.detach().norm().item() for p in m.parameters()], gold["weight_norms_end"], rtol=1e-5)


def test_golden_schedules_and_ops():
    s = np.load(GOLD / "schedules.npz")
    assert np.array_equal(s["lr_a"], R.cosine_scheduler(4e-3, 1e-6, 3, 7, warmup_epochs=1))
    assert np.array_equal(s["wd_b"], R.cosine_scheduler(0.05, 0.1, 5, 4))
    assert s["lr_a"][0] == 0.0 and abs(s["lr_a"][6] - 4e-3) < 1e-12 and len(s["lr_b"]) == 20
    o = np.load(GOLD / "ops.npz")
    t = lambda k: torch.from_numpy(o[k])
    assert rel_l2(R.layernorm(t("ln_x"), t("ln_w"), t("ln_b")), t("ln_y")) < 1e-6
    assert rel_l2(F.layer_norm(t("ln_x"), (192,), t("ln_w"), t("ln_b"), 1e-6), t("ln_y")) < 1e-6
    assert rel_l2(R.gelu(t("gelu_x")), t("gelu_y")) < 1e-6
    assert rel_l2(0.5 * t("gelu_x").double() * (1 + torch.erf(t("gelu_x").double() / 2 ** 0.5)), t("gelu_y")) < 1e-6
    qkv = t("attn_qkv").requires_grad_(True)
    out = R.attention(qkv, 3)
    out.backward(t("attn_do"))
    assert rel_l2(out, t("attn_o")) < 1e-5 and rel_l2(qkv.grad, t("attn_dqkv")) < 1e-5
    z = t("loss_logits").requires_grad_(True)
    R.label_smoothing_ce(z, t("loss_targets"), 0.1).backward()
    assert rel_l2(z.grad, t("loss_ls_grad")) < 1e-5
