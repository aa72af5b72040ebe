"""The branches of the training loop beside the plain step, on the GPU against the CPU oracle's restatement of the same
branch: knowledge distillation as a workload (main.py:836-851, 939-970; test_kd.py:18-126), the weight EMA
(engine.py:198-202, 273-274), gradient clipping (engine.py:175-177), gradient accumulation (engine.py:84-88, 268-274),
the reference's CPU-runnable configuration at full depth (ViT-Tiny, 64 images, 12 blocks), and checkpoint resume /
fine-tune loading followed by real steps (utils/__init__.py:686-770, main.py:653-674).

Tolerances: two bf16 pipelines decorrelate at the bf16 quantisation floor after a few layers (see test_model_gpu.py), so
results that went through the whole network are held to E2E_TOL (loss) and WEIGHT_TOL (weights after AdamW steps, which
normalise every gradient to about lr); anything that is the same arithmetic on the same inputs is held to 1e-6 or to
bit equality."""
import argparse
import copy

import numpy as np
import pytest
import torch

from tests.test_model_gpu import E2E_TOL, _batch, _pair
from tests.util import rel_l2

pytestmark = pytest.mark.gpu

WEIGHT_TOL = 2e-3


class _Args:
    opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None


def _weights(m, ref):
    refp = dict(ref.named_parameters())
    pv = torch.cat([p.detach().reshape(-1).cpu() for _, p in m.named_parameters()])
    rv = torch.cat([refp[n].detach().reshape(-1) for n, _ in m.named_parameters()])
    return pv, rv


def _lsce(o, t):
    from oracle import ref_ops as R
    return R.label_smoothing_ce(o, t, 0.1)


# ------------------------------------------------------------------------------------------ knowledge distillation
@pytest.mark.parametrize("student,pool", [("vit_tiny_patch16_224", "avg"), ("deit_tiny_distilled_patch16_224", "token")])
def test_kd_workload_matches_oracle(device, student, pool):
    """StudentWithDistillation(student, frozen ViT teacher) + DistillationLoss(LabelSmoothingCE(0.1), alpha 0.7, T 4)
    through train_one_epoch for two optimizer steps, against the oracle's restatement of both classes in the oracle's
    loop.  Also test_kd.py's contract: a tuple in train mode, one tensor in eval mode, no gradient for the teacher."""
    from oracle import ref_engine
    from vision_transformers_torch_xla_amd import engine, optim_factory
    from vision_transformers_torch_xla_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy, StudentWithDistillation
    ref_s, s = _pair(student, device, depth=2, global_pool=pool, seed=0)
    ref_t, t = _pair("vit_small_patch16_224", device, depth=2, global_pool="avg", seed=7)
    for teacher in (ref_t, t):                               # main.py:737-741
        teacher.eval()
        for p in teacher.parameters():
            p.requires_grad = False
    t_before = t._flat["p"].clone()
    model, rmodel = StudentWithDistillation(s, t), ref_engine.RefStudentWithDistillation(ref_s, ref_t)
    crit = DistillationLoss(LabelSmoothingCrossEntropy(0.1), 0.7, 4.0)
    rcrit = ref_engine.RefDistillationLoss(_lsce, 0.7, 4.0)
    B, steps = 4, 2
    data = [_batch(B, 224, 40, 300 + i) for i in range(steps)]

    # the wrapper's contract (test_kd.py:103-120)
    model.train()
    out = model(data[0][0].to(device))
    assert isinstance(out, tuple) and len(out) == 2 and out[0].requires_grad and not out[1].requires_grad
    model.eval()
    with torch.no_grad():
        assert isinstance(model(data[0][0].to(device)), torch.Tensor)

    opt = optim_factory.create_optimizer(_Args, s)           # the student is what is optimised (main.py:866-869)
    ropt = ref_engine.make_optimizer(ref_s, 1e-3, 0.05)
    want = ref_engine.ref_train_one_epoch(rmodel, rcrit, data, ropt, 0, None, None, steps, 1, forward_kwargs=dict(emulate=True))
    got = engine.train_one_epoch(model, crit, data, opt, device, 0, None, 0, None, None, num_training_steps_per_epoch=steps,
                                 update_freq=1, log_freq=1)
    print(f"KD {student}: HIP {got} oracle {want}")
    assert set(got) == {"loss", "lr", "class_acc"}
    assert abs(got["loss"] - want["loss"]) <= E2E_TOL * abs(want["loss"])
    assert abs(got["class_acc"] - want["class_acc"]) <= 1e-6
    pv, rv = _weights(s, ref_s)
    print(f"   student weights after {steps} steps rel {rel_l2(pv, rv):.2e}")
    assert rel_l2(pv, rv) <= WEIGHT_TOL
    assert torch.equal(t._flat["p"], t_before)               # the teacher is untouched
    assert all(p.grad is None for p in t.parameters())


def test_side_stream_placements_change_no_bit(device, monkeypatch):
    """Three things run on the side stream because nothing waits for them right away: the teacher's forward pass beside the
    student's (VITK_KD_OVERLAP), the refresh of the transposed weight copies after an optimizer step (VITK_DEFER_WT) and the
    second stage of the LayerNorm parameter gradients (VITK_DEFER_LN_REDUCE).  Three distillation steps with the placements
    on and with all of them off must agree in every bit of the losses and of the weights -
    a missing wait between the streams would show as a difference (or as a stale weight in the third step's gradients)."""
    from vision_transformers_torch_xla_amd import optim_factory
    from vision_transformers_torch_xla_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy, StudentWithDistillation
    from vision_transformers_torch_xla_amd.vit import create_model
    data = [_batch(6, 224, 40, 900 + i) for i in range(3)]

    def run(flag):
        monkeypatch.setenv("VITK_KD_OVERLAP", flag)
        monkeypatch.setenv("VITK_DEFER_WT", flag)
        monkeypatch.setenv("VITK_DEFER_LN_REDUCE", flag)
        torch.manual_seed(11)
        s = create_model("deit_tiny_distilled_patch16_224", num_classes=40, global_pool="token", drop_path_rate=0.1, depth=3).to(device)
        t = create_model("vit_small_patch16_224", num_classes=40, global_pool="avg", depth=2).to(device).eval()
        for p in t.parameters():
            p.requires_grad = False
        model = StudentWithDistillation(s, t).train()
        crit = DistillationLoss(LabelSmoothingCrossEntropy(0.1), 0.7, 4.0)
        opt = optim_factory.create_optimizer(_Args, s)
        losses = []
        for x, y in data:
            loss = crit(model(x.to(device)), y.to(device))
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), s._flat["p"].detach().cpu().clone(), s._flat["b"].detach().cpu().clone()

    on, off = run("1"), run("0")
    for a, b in zip(on, off):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kd", [False, True])
def test_graphed_step_is_the_eager_step(device, monkeypatch, kd):
    """VITK_GRAPH_STEP=1: train_one_epoch captures forward + loss + backward in a HIP graph after two eager steps and replays it.
    Seven steps (the last one a smaller batch, which runs eagerly) against the eager loop with stochastic depth computed for
    every sample (VITK_DROPPATH_COMPACT=0, the arithmetic the replayable step uses): returned statistics and every weight bit
    must be the same - with the per-step learning rate, the re-drawn DropPath factors and, for distillation, the teacher's
    forward pass on the side stream inside the graph."""
    from vision_transformers_torch_xla_amd import engine, optim_factory, utils
    from vision_transformers_torch_xla_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy, StudentWithDistillation
    from vision_transformers_torch_xla_amd.vit import create_model
    monkeypatch.setenv("VITK_DROPPATH_COMPACT", "0")
    data = [_batch(8, 224, 40, 700 + i) for i in range(6)] + [_batch(4, 224, 40, 777)]
    lr = utils.cosine_scheduler(1e-3, 1e-5, 1, len(data), warmup_epochs=1, warmup_steps=2)

    def run(flag):
        monkeypatch.setenv("VITK_GRAPH_STEP", flag)
        torch.manual_seed(5)
        s = create_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", drop_path_rate=0.2, depth=3).to(device)
        model, crit = s, LabelSmoothingCrossEntropy(0.1)
        if kd:
            t = create_model("vit_small_patch16_224", num_classes=40, global_pool="avg", depth=2).to(device).eval()
            for p in t.parameters():
                p.requires_grad = False
            model, crit = StudentWithDistillation(s, t), DistillationLoss(crit, 0.7, 4.0)
        opt = optim_factory.create_optimizer(_Args, s)
        stats = engine.train_one_epoch(model, crit, data, opt, device, 0, None, 1.0, None, None, lr_schedule_values=lr,
                                       num_training_steps_per_epoch=len(data), update_freq=1, log_freq=1)
        torch.cuda.synchronize()
        assert s._static_drop is None                     # the loop leaves the model in its per-step mode
        return stats, s._flat["p"].detach().cpu().clone()

    before = engine.GraphedStep.total_replays
    st_g, p_g = run("1")
    assert engine.GraphedStep.total_replays - before == 4      # steps 2-5: two eager steps first, the smaller last batch eager
    st_e, p_e = run("0")
    assert engine.GraphedStep.total_replays - before == 4
    assert st_g == st_e
    assert torch.equal(p_g, p_e)


def test_graphed_step_survives_epochs_and_fills_param_grads(device, monkeypatch):
    """The captured step is kept from one train_one_epoch call to the next (one capture, not one per epoch), and after a replay -
    which runs no Python - every parameter's .grad is the view of the flat gradient buffer the replayed kernels wrote: torch's own
    clip_grad_norm_ (a reader of p.grad) sees the same norm as the library's clipping."""
    from vision_transformers_torch_xla_amd import engine, optim_factory, utils
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model
    monkeypatch.setenv("VITK_GRAPH_STEP", "1")
    data = [_batch(8, 224, 40, 800 + i) for i in range(5)]
    torch.manual_seed(5)
    m = create_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", drop_path_rate=0.1, depth=2).to(device)
    crit = LabelSmoothingCrossEntropy(0.1)
    opt = optim_factory.create_optimizer(_Args, m)
    before = engine.GraphedStep.total_replays
    engine.train_one_epoch(m, crit, data, opt, device, 0, None, 0, None, None, update_freq=1, log_freq=1)
    assert engine.GraphedStep.total_replays - before == 3          # two eager steps, then the capture and replays
    held = m._vitk_graphed_step
    assert held.graph is not None and m._static_drop is None
    engine.train_one_epoch(m, crit, data, opt, device, 1, None, 0, None, None, update_freq=1, log_freq=1)
    assert m._vitk_graphed_step is held and engine.GraphedStep.total_replays - before == 8     # the second epoch replays all five steps
    # one more replayed step by hand: gradients through p.grad
    held.resume()
    x, y = data[0]
    m.zero_grad(set_to_none=True)            # what a caller's torch-style zero_grad leaves behind: no .grad at all
    assert all(p.grad is None for p in m.parameters()) and not m._grad_valid
    held.step(x.to(device), y.to(device))
    held.pause()
    assert m._grad_valid and all(p.grad is not None and p.grad.data_ptr() == m._gview[n].data_ptr() for n, p in m.named_parameters())
    flat_norm = m._flat["g"].double().norm().item()
    torch_norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 1e9).item()
    assert abs(torch_norm - flat_norm) <= 1e-5 * flat_norm and flat_norm > 0


def test_kd_loss_gradient_matches_autograd(device):
    """the fused criterion on identical logits: value and d/d(student logits) against autograd through the oracle's formula"""
    from oracle import ref_ops as R
    from vision_transformers_torch_xla_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy
    g = torch.Generator().manual_seed(4)
    s0, t0 = 3 * torch.randn(16, 1000, generator=g), 3 * torch.randn(16, 1000, generator=g)
    y = torch.randint(0, 1000, (16,), generator=g)
    sr = s0.clone().requires_grad_(True)
    want = R.distillation_loss(sr, t0, y, _lsce, 0.7, 4.0)
    want.backward()
    sg = s0.clone().to(device).requires_grad_(True)
    got = DistillationLoss(LabelSmoothingCrossEntropy(0.1), 0.7, 4.0)((sg, t0.to(device)), y.to(device))
    got.backward()
    assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item())
    assert rel_l2(sg.grad, sr.grad) <= 1e-5


# --------------------------------------------------------------------------------------------------------- EMA
def test_model_ema_built_after_a_training_step(device):
    """ModelEma deep-copies the model: the copy must work after the model has run a training forward / backward with
    stochastic depth (pinned-table ring with events, side stream), share nothing with it, and follow
    ema = decay * ema + (1 - decay) * model on every entry of the state dict."""
    from oracle.ref_vit import create_ref_model
    from vision_transformers_torch_xla_amd import optim_factory, utils
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    ref, m = _pair("vit_tiny_patch16_224", device, depth=2, dpr=0.3)
    opt = optim_factory.create_optimizer(_Args, m)
    crit = LabelSmoothingCrossEntropy(0.1)
    x, y = _batch(8, 224, 40, 21)
    m.train()

    def step():
        crit(m(x.to(device)), y.to(device)).backward()
        opt.step()
        opt.zero_grad()

    torch.manual_seed(0)
    step()
    ema = utils.ModelEma(m, decay=0.9)
    assert not ema.ema.training and all(not p.requires_grad for p in ema.ema.parameters())
    assert torch.equal(ema.ema._flat["p"], m._flat["p"]) and ema.ema._flat["p"].data_ptr() != m._flat["p"].data_ptr()
    assert ema.ema._flat["g"] is None                         # an averaged copy never trains: no gradient buffer
    e = {k: v.detach().clone() for k, v in ema.ema.state_dict().items()}
    for _ in range(3):
        step()
        ema.update(m)
        e = {k: 0.9 * e[k] + (1.0 - 0.9) * v.detach() for k, v in m.state_dict().items()}
    for k, v in ema.ema.state_dict().items():
        assert rel_l2(v, e[k]) <= 1e-6, k
    # the averaged copy evaluates like the oracle with the averaged weights
    ref2 = create_ref_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", depth=2)
    ref2.load_state_dict({k: v.cpu() for k, v in ema.ema.state_dict().items()})
    ref2.eval()
    with torch.no_grad():
        assert rel_l2(ema.ema(x.to(device)), ref2(x, emulate=True)) <= E2E_TOL


class _SnapshotLoader:
    """yields the batches and records the model's weights whenever the loop asks for the next one (= after every step)"""

    def __init__(self, data, model):
        self.data, self.model, self.snaps = data, model, []

    def __iter__(self):
        for b in self.data:
            self.snaps.append(self.model._flat["p"].clone())
            yield b
        self.snaps.append(self.model._flat["p"].clone())


def test_train_one_epoch_with_ema_clip_and_accumulation(device):
    """train_one_epoch(max_norm=1, model_ema=..., update_freq=2): 4 micro-batches = 2 optimizer steps.  Against the oracle's
    loop with torch.nn.utils.clip_grad_norm_ and its own EMA; and the EMA against the exact recurrence over the HIP model's
    own weights after each optimizer step."""
    from oracle import ref_engine, ref_ops as R
    from vision_transformers_torch_xla_amd import engine, optim_factory, utils
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    ref, m = _pair("vit_tiny_patch16_224", device, depth=2)
    B, micro, uf = 4, 4, 2
    data = [_batch(B, 224, 40, 400 + i) for i in range(micro)]
    lr_sched = R.cosine_scheduler(1e-3, 1e-6, 2, micro // uf, warmup_epochs=1, start_warmup_value=2e-4)
    wd_sched = R.cosine_scheduler(0.05, 0.01, 2, micro // uf)
    opt = optim_factory.create_optimizer(_Args, m)
    ropt = ref_engine.make_optimizer(ref, 1e-3, 0.05)
    ema, rema = utils.ModelEma(m, decay=0.5), ref_engine.RefModelEma(ref, decay=0.5)
    norms = []
    want = ref_engine.ref_train_one_epoch(ref, _lsce, data, ropt, 0, lr_sched, wd_sched, micro // uf, uf, forward_kwargs=dict(emulate=True),
                                          max_norm=1.0, model_ema=rema,
                                          record=lambda i, o, l, mod: norms.append(torch.cat([p.grad.reshape(-1) for p in mod.parameters()]).norm().item()))
    assert min(norms[uf - 1::uf]) > 1.5                       # the clip is active in every optimizer step of this test
    loader = _SnapshotLoader(data, m)
    got = engine.train_one_epoch(m, LabelSmoothingCrossEntropy(0.1), loader, opt, device, 0, None, 1.0, ema, None,
                                 start_steps=0, lr_schedule_values=lr_sched, wd_schedule_values=wd_sched,
                                 num_training_steps_per_epoch=micro // uf, update_freq=uf, log_freq=1)
    print(f"clip + EMA + update_freq=2: HIP {got} oracle {want}; oracle gradient norms before clipping {norms}")
    assert abs(got["loss"] - want["loss"]) <= E2E_TOL * abs(want["loss"])          # engine.py:268,294: the logged loss is loss / update_freq
    assert abs(got["lr"] - want["lr"]) <= 1e-12 and abs(got["class_acc"] - want["class_acc"]) <= 1e-6
    pv, rv = _weights(m, ref)
    ev, erv = _weights(ema.ema, rema.ema)
    print(f"   weights rel {rel_l2(pv, rv):.2e}, EMA weights rel {rel_l2(ev, erv):.2e}")
    assert rel_l2(pv, rv) <= WEIGHT_TOL and rel_l2(ev, erv) <= WEIGHT_TOL
    # weights change only on the last micro-step of an accumulation group; the EMA follows exactly those weights
    s = loader.snaps
    assert len(s) == micro + 1 and torch.equal(s[0], s[1]) and torch.equal(s[2], s[3]) and not torch.equal(s[1], s[2])
    e = s[0]
    for after_step in (s[2], s[4]):
        e = 0.5 * e + 0.5 * after_step
    assert rel_l2(ema.ema._flat["p"], e) <= 1e-6


def test_clip_grad_norm_matches_torch(device):
    """engine.clip_grad_norm_ on the flat gradient buffer = torch.nn.utils.clip_grad_norm_ on the same gradients"""
    from vision_transformers_torch_xla_amd import engine
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    _, m = _pair("vit_tiny_patch16_224", device, depth=2)
    x, y = _batch(4, 224, 40, 31)
    m.train()
    for max_norm in (0.5, 1e6):                               # clipping / not clipping
        m.zero_grad()
        LabelSmoothingCrossEntropy(0.1)(m(x.to(device)), y.to(device)).backward()
        clones = [p.grad.detach().clone().requires_grad_(False) for p in m.parameters()]
        holders = [torch.nn.Parameter(torch.zeros_like(c)) for c in clones]
        for h, c in zip(holders, clones):
            h.grad = c
        want_norm = torch.nn.utils.clip_grad_norm_(holders, max_norm)
        got_norm = engine.clip_grad_norm_(m, max_norm)
        assert abs(got_norm.item() - want_norm.item()) <= 1e-5 * want_norm.item()
        got = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        want = torch.cat([h.grad.reshape(-1) for h in holders])
        assert rel_l2(got, want) <= 1e-6
        if max_norm > 1e5:
            assert torch.equal(got, torch.cat([c.reshape(-1) for c in clones]))     # below the bound nothing is touched


def test_accumulated_micro_batches_equal_the_concatenated_batch(device):
    """update_freq = 2 (loss / 2 per micro-batch, engine.py:268) gives the gradient of one step on the concatenated batch"""
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    _, m = _pair("vit_tiny_patch16_224", device, depth=2)
    x, y = _batch(8, 224, 40, 41)
    x, y = x.to(device), y.to(device)
    crit = LabelSmoothingCrossEntropy(0.1)
    m.train()
    m.zero_grad()
    crit(m(x), y).backward()
    whole = m._flat["g"].clone()
    m.zero_grad()
    for a in (slice(0, 4), slice(4, 8)):
        (crit(m(x[a]), y[a]) / 2).backward()
    e = rel_l2(m._flat["g"], whole)
    print(f"accumulated vs concatenated gradient rel {e:.2e}")
    assert e <= 1e-3


def test_three_steps_against_the_committed_fixture(device):
    """tests/golden/vit_tiny_d2_train3.npz (written by oracle/make_golden.py): three consecutive optimizer steps of the 2-block
    ViT-Tiny through train_one_epoch - per-step logits and loss, the parameter groups, the returned dict and the weights at
    the end against the stored bf16-emulating oracle run (and the fp32 run at the same end-to-end bar)"""
    from pathlib import Path
    import numpy as np
    from oracle.make_golden import TRAIN3_SEEDS, tiny_batch, tiny_model
    from vision_transformers_torch_xla_amd import engine, optim_factory
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model
    gold = np.load(Path(__file__).resolve().parent / "golden" / "vit_tiny_d2_train3.npz")
    m = create_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", depth=2)
    m.load_state_dict(tiny_model().state_dict())               # the fixture's seeded initial state
    m = m.to(device)
    data = [tiny_batch(seed=s) for s in TRAIN3_SEEDS]
    opt = optim_factory.create_optimizer(_Args, m)
    names = {id(p): n for n, p in m.named_parameters()}
    assert [[names[id(p)] for p in g["params"]] for g in opt.param_groups] == [list(gold["group0_names"]), list(gold["group1_names"])]
    assert [g["weight_decay"] for g in opt.param_groups] == list(gold["group_weight_decay"])
    crit = LabelSmoothingCrossEntropy(0.1)
    seen = []

    def recording(o, t):
        seen.append(o.detach().float().cpu().numpy())
        return crit(o, t)
    got = engine.train_one_epoch(m, recording, data, opt, device, 0, None, 0, None, None, start_steps=0, lr_schedule_values=gold["lr_schedule"],
                                 wd_schedule_values=gold["wd_schedule"], num_training_steps_per_epoch=3, update_freq=1, log_freq=1)
    for tag in ("_emul", ""):
        errs = [rel_l2(torch.from_numpy(a), torch.from_numpy(b)) for a, b in zip(seen, gold["logits" + tag])]
        print(f"fixture{tag or '_fp32'}: per-step logits rel {[f'{e:.1e}' for e in errs]}, loss {got['loss']:.6f} vs {gold['stats' + tag][0]:.6f}")
        assert max(errs) <= E2E_TOL
        assert abs(got["loss"] - gold["stats" + tag][0]) <= E2E_TOL * gold["stats" + tag][0]
        assert abs(got["lr"] - gold["stats" + tag][1]) <= 1e-12 and abs(got["class_acc"] - gold["stats" + tag][2]) <= 1e-6
        wn = np.array([p.detach().norm().item() for p in m.parameters()])
        # Zero-initialised biases consist of three AdamW updates only (about lr * sign(g) each), and the key bias of attention has a
        # mathematically zero gradient (softmax is shift-invariant), so its updates are rounding noise on both sides: the bar is
        # on the vector of norms, not on each parameter.
        dev_ = np.abs(wn - gold["weight_norms_end" + tag]) / gold["weight_norms_end" + tag]
        print(f"   weight norms at the end: worst relative deviation {dev_.max():.1e} (median {np.median(dev_):.1e})")
        assert np.linalg.norm(wn - gold["weight_norms_end" + tag]) <= 2e-3 * np.linalg.norm(wn) and np.median(dev_) <= 1e-3


def test_metric_logger_keeps_device_scalars_on_the_gpu(device):
    """meters take 0-d GPU tensors without a host sync and fetch all of them together when they are read"""
    from vision_transformers_torch_xla_amd import utils
    ml = utils.MetricLogger(delimiter="  ")
    vals = [0.5, 2.0, 3.5]
    for v in vals:
        ml.update(loss=torch.tensor(v, device=device))
        ml.meters["class_acc"].update(torch.tensor(v / 4, device=device), n=8)
        ml.update(lr=0.1)
    assert len(ml.meters["loss"]._pending) == 3 and ml.meters["loss"]._seen == 0      # nothing fetched yet
    ml.flush()
    assert not ml.meters["loss"]._pending and not ml.meters["class_acc"]._pending
    assert ml.loss.count == 3 and ml.loss.global_avg == 2.0 and ml.loss.value == 3.5 and ml.loss.median == 2.0
    assert ml.meters["class_acc"].count == 24 and abs(ml.meters["class_acc"].global_avg - 0.5) < 1e-7
    ml.update(loss=torch.tensor(6.0, device=device))
    assert ml.loss.global_avg == 3.0                       # a read resolves what is pending
    ml.update(loss=torch.tensor(1.0, device=device))
    ml.update(loss=9.0)                                    # host value after a device value: order is kept
    assert ml.loss.value == 9.0 and ml.loss.count == 6


# ------------------------------------------------------------------------ the CPU-runnable configuration, full size
def test_vit_tiny_batch64_full_depth_three_steps(device):
    """The reference's CPU-runnable case (ViT-Tiny/16, 64 images of 224 px, all 12 blocks, timm initialisation, drop_path 0.1)
    for three optimizer steps of train_one_epoch against the CPU oracle's loop.  The stochastic-depth factors are drawn
    once and used by both sides in every step (the two sides draw from different generators otherwise)."""
    from oracle import ref_engine, ref_ops as R
    from oracle.ref_vit import create_ref_model
    from vision_transformers_torch_xla_amd import engine, optim_factory
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model
    torch.manual_seed(0)
    ref = create_ref_model("vit_tiny_patch16_224", num_classes=1000, global_pool="avg", drop_path_rate=0.1)
    m = create_model("vit_tiny_patch16_224", pretrained=False, num_classes=1000, global_pool="avg", drop_path_rate=0.1)
    m.load_state_dict(ref.state_dict())
    m = m.to(device)
    assert sum(p.numel() for p in m.parameters()) == 5_717_416
    B, steps = 64, 3
    data = [_batch(B, 224, 1000, 500 + i) for i in range(steps)]
    ref.train()
    scales = ref.make_drop_scales(B, generator=torch.Generator().manual_seed(8))
    m._forced_drop_scales = scales
    lr_sched = R.cosine_scheduler(4e-3, 1e-6, 2, steps, warmup_epochs=1)
    wd_sched = R.cosine_scheduler(0.05, 0.05, 2, steps)
    opt = optim_factory.create_optimizer(_Args, m)
    ropt = ref_engine.make_optimizer(ref, 4e-3, 0.05)
    losses = []
    want = ref_engine.ref_train_one_epoch(ref, _lsce, data, ropt, 0, lr_sched, wd_sched, steps, 1,
                                          forward_kwargs=dict(emulate=True, drop_scales=scales), record=lambda i, o, l, mod: losses.append(l))
    got = engine.train_one_epoch(m, LabelSmoothingCrossEntropy(0.1), data, opt, device, 0, None, 0, None, None, start_steps=0,
                                 lr_schedule_values=lr_sched, wd_schedule_values=wd_sched, num_training_steps_per_epoch=steps,
                                 update_freq=1, log_freq=1)
    pv, rv = _weights(m, ref)
    print(f"ViT-Tiny B=64 depth 12, 3 steps: HIP {got} oracle {want} (oracle losses {losses}); weights rel {rel_l2(pv, rv):.2e}")
    assert abs(got["loss"] - want["loss"]) <= E2E_TOL * abs(want["loss"])
    assert abs(got["lr"] - want["lr"]) <= 1e-12
    assert abs(got["class_acc"] - want["class_acc"]) <= 1.0 / B + 1e-9        # 192 argmax decisions over 1000 near-tied classes at init
    assert rel_l2(pv, rv) <= 5e-3


# ------------------------------------------------------------------------------------------------ checkpoints
def _ck_args(tmp, **kw):
    a = argparse.Namespace(output_dir=str(tmp), save_ckpt_num=3, save_ckpt_freq=1, auto_resume=True, resume="", model_ema=True,
                           eval=False, start_epoch=0, opt="adamw", lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=None)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_resume_is_bit_identical_to_the_uninterrupted_run(device, tmp_path):
    """two steps, save_model, a fresh model / optimizer / EMA, auto_load_model, a third step: parameters, both Adam moments
    and the EMA weights have the bits of the run that never stopped (utils/__init__.py:686-770)"""
    from vision_transformers_torch_xla_amd import optim_factory, utils
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model
    crit = LabelSmoothingCrossEntropy(0.1)
    data = [tuple(t.to(device) for t in _batch(6, 224, 40, 600 + i)) for i in range(3)]

    def fresh(seed):
        torch.manual_seed(seed)
        m = create_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", depth=3, drop_path_rate=0.2).to(device)
        m.train()
        return m, optim_factory.create_optimizer(_Args, m), utils.ModelEma(m, decay=0.9)

    def step(m, opt, ema, i):
        torch.manual_seed(100 + i)                            # the host-side stochastic-depth draws of step i
        crit(m(data[i][0]), data[i][1]).backward()
        opt.step()
        opt.zero_grad()
        ema.update(m)

    m, opt, ema = fresh(0)
    for i in range(2):
        step(m, opt, ema, i)
    args = _ck_args(tmp_path)
    utils.save_model(args, 0, m, m, opt, None, model_ema=ema)
    step(m, opt, ema, 2)

    m2, opt2, ema2 = fresh(1)                                 # other initial weights: everything must come from the file
    args2 = _ck_args(tmp_path)
    utils.auto_load_model(args2, m2, m2, opt2, None, model_ema=ema2)
    assert args2.start_epoch == 1 and opt2._steps == 2
    step(m2, opt2, ema2, 2)
    torch.cuda.synchronize()
    assert torch.equal(m2._flat["p"], m._flat["p"])
    assert torch.equal(opt2._m, opt._m) and torch.equal(opt2._v, opt._v)
    assert torch.equal(ema2.ema._flat["p"], ema.ema._flat["p"])
    assert torch.equal(m2._flat["b"], m._flat["b"])           # the bf16 copies the GEMMs read


def test_finetune_224_to_384_then_train(device, tmp_path):
    """a 224 px checkpoint initialises a 384 px model with another head (main.py:653-674 + the position-table resample), and
    the loaded model takes a training step that matches the oracle loaded with the same state"""
    from oracle.ref_vit import create_ref_model
    from vision_transformers_torch_xla_amd import optim_factory, utils
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model
    torch.manual_seed(0)
    src = create_model("vit_tiny_patch16_224", num_classes=24, global_pool="avg", depth=1)
    torch.save({"model": src.state_dict()}, tmp_path / "pre.pth")
    dst = create_model("vit_tiny_patch16_224", num_classes=16, global_pool="avg", depth=1, img_size=384).to(device)
    missing, unexpected = utils.load_finetune_checkpoint(dst, tmp_path / "pre.pth")
    assert sorted(missing) == ["head.bias", "head.weight"] and unexpected == []
    assert dst._views_ok() and dst.pos_embed.shape == (1, 577, 192)
    assert torch.equal(dst.blocks[0].mlp.fc1.weight.cpu(), src.blocks[0].mlp.fc1.weight)
    ref = create_ref_model("vit_tiny_patch16_224", num_classes=16, global_pool="avg", depth=1, img_size=384)
    ref.load_state_dict({k: v.cpu() for k, v in dst.state_dict().items()})
    x, y = _batch(2, 384, 16, 51)
    ref.train(); dst.train()
    from oracle import ref_ops as R
    R.label_smoothing_ce(ref(x, emulate=True), y, 0.1).backward()
    opt = optim_factory.create_optimizer(_Args, dst)
    loss = LabelSmoothingCrossEntropy(0.1)(dst(x.to(device)), y.to(device))
    loss.backward()
    refp = dict(ref.named_parameters())
    gv = torch.cat([p.grad.reshape(-1).cpu() for _, p in dst.named_parameters()])
    rv = torch.cat([refp[n].grad.reshape(-1) for n, _ in dst.named_parameters()])
    print(f"384 px fine-tune step: gradient rel {rel_l2(gv, rv):.2e}")
    assert rel_l2(gv, rv) <= 3e-2
    opt.step()
    assert all(torch.isfinite(p).all() for p in dst.parameters())


# ------------------------------------------------------------------------------------------------ the loop learns
def test_memorising_one_batch_follows_the_fp32_oracle(device):
    """Thirty optimizer steps of train_one_epoch on ONE repeated batch (ViT-Small/16, all 12 blocks, 16 images, 100 classes,
    no stochastic depth, plain cross-entropy, AdamW with warm-up) beside the fp32 CPU oracle's loop from the same weights: the
    loss has to FALL the way the exact-arithmetic loop's does - a training step that computes plausible but wrong gradients
    (a wrong sign in one branch, a stale weight copy, a mis-scaled update) passes every one-step comparison within tolerance
    for a step or two and then stops learning.  Two bf16 / fp32 trajectories separate slowly, so the curves are compared with a
    bar that widens with the step; both must have memorised the batch at the end."""
    from oracle import ref_engine, ref_ops as R
    from oracle.ref_vit import create_ref_model
    from vision_transformers_torch_xla_amd import engine, optim_factory
    from vision_transformers_torch_xla_amd.losses import CrossEntropyLoss
    from vision_transformers_torch_xla_amd.vit import create_model
    torch.manual_seed(3)
    ref = create_ref_model("vit_small_patch16_224", num_classes=100, global_pool="avg", drop_path_rate=0.0)
    m = create_model("vit_small_patch16_224", pretrained=False, num_classes=100, global_pool="avg", drop_path_rate=0.0)
    m.load_state_dict(ref.state_dict())
    m = m.to(device)
    B, steps = 16, 30
    batch = _batch(B, 224, 100, 77)
    data = [batch] * steps
    lr_sched = R.cosine_scheduler(5e-4, 1e-5, 1, steps, warmup_epochs=1, warmup_steps=5)
    wd_sched = R.cosine_scheduler(0.05, 0.05, 1, steps)

    class A:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 5e-4, 0.05, 1e-8, None
    opt = optim_factory.create_optimizer(A, m)
    ropt = ref_engine.make_optimizer(ref, 5e-4, 0.05)
    want, got = [], []
    ref_engine.ref_train_one_epoch(ref, torch.nn.CrossEntropyLoss(), data, ropt, 0, lr_sched, wd_sched, steps, 1,
                                   record=lambda i, o, l, mod: want.append(l))

    class Recording(CrossEntropyLoss):
        def forward(self, output, target):
            loss = super().forward(output, target)
            got.append(loss.detach())
            return loss
    stats = engine.train_one_epoch(m, Recording(), data, opt, device, 0, None, 0, None, None, start_steps=0, lr_schedule_values=lr_sched,
                                   wd_schedule_values=wd_sched, num_training_steps_per_epoch=steps, update_freq=1, log_freq=1)
    got = [float(g) for g in got]
    print("memorising one batch: fp32 oracle", [round(v, 3) for v in want], "HIP", [round(v, 3) for v in got])
    assert len(got) == len(want) == steps
    assert abs(got[0] - want[0]) <= 2e-3 * want[0]                      # same weights, same batch: the first loss is the forward pass
    for i, (g, w) in enumerate(zip(got, want)):
        assert abs(g - w) <= (0.01 + 0.003 * i) * max(w, 0.2) + 0.01, (i, g, w)      # measured: <= 0.016 at every step
    assert want[-1] < 0.1 * want[0] and got[-1] < 0.1 * got[0]          # both memorised the batch
    assert stats["class_acc"] > 0.5                                     # averaged over the epoch: wrong at first, right at the end


def test_loggers_receive_the_reference_records(device):
    """log_writer.update(loss=, lr=) every step and wandb_logger._wandb.log({...}) at the console's frequency, with the reference's
    keys and step numbering (engine.py:305-323); class accuracy only when mixup is off"""
    from vision_transformers_torch_xla_amd import engine, optim_factory
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model

    class Writer:
        def __init__(self):
            self.calls, self.steps = [], []

        def set_step(self, step=None):
            self.steps.append(step)

        def update(self, head="scalar", step=None, **kw):
            self.calls.append(kw)

    class Wandb:
        class _W:
            def __init__(self):
                self.records = []

            def log(self, d):
                self.records.append(dict(d))

        def __init__(self):
            self._wandb = Wandb._W()
    torch.manual_seed(0)
    m = create_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", depth=1).to(device)
    opt = optim_factory.create_optimizer(_Args, m)
    data = [_batch(4, 224, 40, 900 + i) for i in range(8)]
    writer, wb = Writer(), Wandb()
    stats = engine.train_one_epoch(m, LabelSmoothingCrossEntropy(0.1), data, opt, device, 2, None, 0, None, None, log_writer=writer,
                                   wandb_logger=wb, start_steps=100, num_training_steps_per_epoch=8, update_freq=1, log_freq=3)
    assert writer.steps == [2 * 8 * 1] and len(writer.calls) == 8 and all(set(c) == {"loss", "lr"} for c in writer.calls)
    recs = wb._wandb.records
    assert [r["train/step"] for r in recs] == [100, 101, 102, 103, 104, 106]          # steps 0-4, then every third
    assert all(set(r) == {"train/loss", "train/learning_rate", "train/epoch", "train/step", "train/accuracy"} for r in recs)
    assert all(r["train/epoch"] == 2 and r["train/learning_rate"] == opt.param_groups[0]["lr"] and np.isfinite(r["train/loss"]) for r in recs)
    assert np.isfinite(stats["loss"])
