"""Parity at the real depth and width of the benchmark configurations: ViT-B/16 (12 blocks), ViT-L/16 (24 blocks),
ViT-B/16 at 384 px (577 tokens) and DeiT-B distilled with a ViT-B teacher (models/vision_transformer.py:2794-2811,
2843-2850; models/deit.py:306-314; main.py:836-851, 939-970) - forward, backward and one AdamW step of the HIP model
against BOTH oracles on identical weights, inputs and stochastic-depth draws (drop_path 0.1, injected masks):

  * the bf16-emulating oracle (same storage points as the kernels): how far two bf16 pipelines of the same policy drift;
  * the fp32 oracle (exact math, the reference's CPU semantics): the distance that matters, and the one that cannot move
    with the kernels - every number measured against it is pinned in tests/golden/fp32_distance.json (tests/util.py).

Besides the end results the residual stream is compared block by block (the oracle's `taps`) and written out as a drift
table (VITK_DRIFT_DIR): the relative distance grows like the square root of the number of rounded stages and stays at the bf16
quantisation floor at depth 12 and 24.  The bars below are a few times the measured values; the recorded values
themselves are held to 1.25 x by the anchor file."""
import copy
import json
import os

import pytest
import torch

from tests.test_model_gpu import _batch, _pair
from tests.util import fp32_anchor, rel_l2

pytestmark = pytest.mark.gpu

# measured against the fp32 oracle on MI355X: residual stream 1.05e-2 after block 12 (ViT-B) and 1.4e-2 after block 24 (ViT-L),
# logits <= 1.1e-2, whole gradient <= 1.9e-2, cosine >= 0.99982
STREAM_TOL = 2e-2
LOGIT_TOL = 2.5e-2
GRAD_TOL = 4e-2
COS_MIN = 0.999

CASES = {
    "vit_b_224": dict(name="vit_base_patch16_224", B=8, size=224, pool="avg"),
    "vit_l_224": dict(name="vit_large_patch16_224", B=4, size=224, pool="avg"),
    "vit_b_384": dict(name="vit_base_patch16_384", B=2, size=384, pool="avg"),
    "deit_b_kd": dict(name="deit_base_distilled_patch16_224", B=4, size=224, pool="token", teacher="vit_base_patch16_224"),
}


class _Args:
    opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None


def _flatgrads(model, names):
    p = dict(model.named_parameters())
    return torch.cat([p[n].grad.detach().reshape(-1).cpu().float() for n in names])


def _flatparams(model, names):
    p = dict(model.named_parameters())
    return torch.cat([p[n].detach().reshape(-1).cpu().float() for n in names])


@pytest.mark.parametrize("case", list(CASES))
def test_full_depth_against_both_oracles(device, case):
    from oracle import ref_engine, ref_ops as R
    from vision_transformers_torch_xla_amd import optim_factory
    from vision_transformers_torch_xla_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy, StudentWithDistillation
    c = CASES[case]
    B, C = c["B"], 1000
    ref_e, m = _pair(c["name"], device, depth=None, num_classes=C, global_pool=c["pool"], dpr=0.1, seed=0)
    ref_f = copy.deepcopy(ref_e)
    depth = m.depth
    names = [n for n, _ in m.named_parameters()]
    x, y = _batch(B, c["size"], C, 31)
    ref_e.train(); ref_f.train(); m.train()
    scales = ref_e.make_drop_scales(B, generator=torch.Generator().manual_seed(9))
    assert sum(int((s == 0).sum()) for pair in scales for s in pair if s is not None) > 0      # some branch really drops a sample

    teacher = None
    if "teacher" in c:
        ref_t, t = _pair(c["teacher"], device, depth=None, num_classes=C, global_pool="avg", seed=7)
        for tm in (ref_t, t):
            tm.eval()
            for p in tm.parameters():
                p.requires_grad = False
        teacher = (ref_t, t)

    lsce = lambda o, tg: R.label_smoothing_ce(o, tg, 0.1)

    def oracle_step(ref, emulate):
        taps = {}
        ref.zero_grad()
        out = ref(x, emulate=emulate, drop_scales=scales, taps=taps)
        if teacher is not None:
            with torch.no_grad():
                tl = teacher[0](x, emulate=emulate)
            loss = R.distillation_loss(out, tl, y, lsce, 0.7, 4.0)
        else:
            loss = lsce(out, y)
        loss.backward()
        grads = _flatgrads(ref, names)
        before = _flatparams(ref, names)
        opt = ref_engine.make_optimizer(ref, _Args.lr, _Args.weight_decay)
        opt.step()
        return dict(out=out.detach(), loss=loss.item(), taps=taps, grads=grads, update=_flatparams(ref, names) - before)

    want = {"emulated": oracle_step(ref_e, True), "fp32": oracle_step(ref_f, False)}

    # ---- HIP: the residual stream entering every block (a forward pass that keeps its intermediates), then the training step
    m._forced_drop_scales = scales
    with torch.no_grad():
        m._sync_weights()
        _, saved = m._forward_impl(x.to(device), m._draw_drop_scales(B, device), save=True)
    N, D = m.num_tokens, m.embed_dim
    stream = [saved["layers"][i][0].float().cpu().reshape(B, N, D) for i in range(depth)]
    del saved
    m.zero_grad()
    opt = optim_factory.create_optimizer(_Args, m)
    if teacher is not None:
        model = StudentWithDistillation(m, teacher[1]).train()
        crit = DistillationLoss(LabelSmoothingCrossEntropy(0.1), 0.7, 4.0)
        outs = model(x.to(device))
        loss = crit(outs, y.to(device))
        got_out = outs[0]
    else:
        got_out = m(x.to(device))
        loss = LabelSmoothingCrossEntropy(0.1)(got_out, y.to(device))
    loss.backward()
    got_grads = _flatgrads(m, names)
    before = _flatparams(m, names)
    opt.step()
    torch.cuda.synchronize()
    got_update = _flatparams(m, names) - before

    table = {}
    for which, w in want.items():
        drift = [rel_l2(stream[i], w["taps"]["x0" if i == 0 else f"{i - 1}.x3"]) for i in range(depth)]
        cos = torch.nn.functional.cosine_similarity(got_grads.double(), w["grads"].double(), dim=0).item()
        table[which] = dict(stream=drift, logits=rel_l2(got_out, w["out"]), loss=abs(loss.item() - w["loss"]) / abs(w["loss"]),
                            gradient=rel_l2(got_grads, w["grads"]), cosine=cos, update=rel_l2(got_update, w["update"]))
    floor = dict(stream=[rel_l2(want["emulated"]["taps"]["x0" if i == 0 else f"{i - 1}.x3"], want["fp32"]["taps"]["x0" if i == 0 else f"{i - 1}.x3"]) for i in range(depth)],
                 logits=rel_l2(want["emulated"]["out"], want["fp32"]["out"]), gradient=rel_l2(want["emulated"]["grads"], want["fp32"]["grads"]))
    table["oracle_emulated_vs_fp32"] = floor
    print(f"\n[{case}] {c['name']} depth {depth}, {B} images, {N} tokens: distance of the HIP model to each oracle")
    print("  block:      " + " ".join(f"{i:>8d}" for i in range(depth)))
    for which in ("emulated", "fp32"):
        print(f"  vs {which:9s}" + " ".join(f"{v:8.1e}" for v in table[which]["stream"]))
    print("  oracles:    " + " ".join(f"{v:8.1e}" for v in floor["stream"]))
    for which in ("emulated", "fp32"):
        t_ = table[which]
        print(f"  vs {which}: logits {t_['logits']:.2e}  loss {t_['loss']:.1e}  gradient {t_['gradient']:.2e} (cosine {t_['cosine']:.6f})  AdamW update {t_['update']:.2e}")
    print(f"  oracle emulated vs fp32: logits {floor['logits']:.2e}  gradient {floor['gradient']:.2e}")
    out_dir = os.environ.get("VITK_DRIFT_DIR")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"drift_{case}.json"), "w") as f:
            json.dump(dict(case=case, model=c["name"], depth=depth, images=B, tokens=N, **table), f, indent=1)

    for which in ("emulated", "fp32"):
        t_ = table[which]
        assert max(t_["stream"]) <= STREAM_TOL, (which, t_["stream"])
        assert t_["logits"] <= LOGIT_TOL and t_["loss"] <= LOGIT_TOL, (which, t_)
        assert t_["gradient"] <= GRAD_TOL and t_["cosine"] >= COS_MIN, (which, t_)
    # the fp32 distances are pinned: a change that moves a rounding point must not move these up
    f32 = table["fp32"]
    fp32_anchor(f"fulldepth.{case}.stream.first_block", f32["stream"][1])
    fp32_anchor(f"fulldepth.{case}.stream.middle_block", f32["stream"][depth // 2])
    fp32_anchor(f"fulldepth.{case}.stream.last_block", f32["stream"][-1])
    fp32_anchor(f"fulldepth.{case}.logits", f32["logits"])
    fp32_anchor(f"fulldepth.{case}.gradient", f32["gradient"])
    fp32_anchor(f"fulldepth.{case}.adamw_update", f32["update"])
