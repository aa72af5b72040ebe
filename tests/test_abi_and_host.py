"""CPU tests (no GPU, no kernel launches): the C-ABI library loads and exports every symbol include/vitk.h declares,
argument validation rejects bad calls before any launch, and the host-side mirror of the reference interface
(registry, flat parameter layout, parameter groups, schedules, meters, optimizer planning) behaves like the reference."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def _lib():
    from vision_transformers_torch_xla_amd import _lib
    from vision_transformers_torch_xla_amd.build import build_lib
    if not _lib.LIB_PATH.exists():
        build_lib(verbose=False)
    return _lib


def test_library_exports_every_declared_symbol():
    L = _lib()
    header = (ROOT / "include" / "vitk.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(vitk_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 35
    lib = ctypes.CDLL(str(L.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/vitk.h but not exported by libvitk.so"
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)       # the ctypes table mirrors the header
    assert L.load().vitk_version() == 100


def test_validation_rejects_before_launch():
    L = _lib()
    lib = L.load()
    # null pointers / bad shapes: VITK_EINVAL (-22) and a message, nothing is launched (there is no GPU here)
    assert lib.vitk_layernorm_fwd(None, None, None, None, None, None, 4, 64, 1e-6, None) == -22
    assert b"null" in lib.vitk_last_error()
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.cast(buf, ctypes.c_void_p).value
    p16 = (p + 15) // 16 * 16
    assert lib.vitk_layernorm_fwd(p16, p16, p16, p16, p16, p16, 4, 60, 1e-6, None) == -22       # D % 8 != 0
    assert lib.vitk_layernorm_fwd(p16, p16, p16, p16, p16, p16, 4, 2048, 1e-6, None) == -22     # D > 1024
    assert lib.vitk_gemm_nt(p16, 12, p16, 12, p16, 16, 16, 16, 12, 0, None, None, 0, None, 0, None, 0, None, 0, None, 0, 0, None) == -22
    assert b"multiples of 8" in lib.vitk_last_error()
    assert lib.vitk_gemm_nt(p16, 64, p16, 64, p16, 16, 16, 16, 64, 4, None, None, 0, None, 0, None, 0, None, 0, None, 0, 0, None) == -22
    assert lib.vitk_gemm_tn(p16, 8, p16, 8, p16, 8, 64, 8, 8, None, 0, p16, 16, None) == -28     # workspace too small
    assert lib.vitk_attention_fwd(p16, p16, p16, 1, 700, 1, 0.125, None) == -22                   # N > 608
    assert lib.vitk_gemm_tn_workspace(100864, 3072, 768) > 0 and lib.vitk_layernorm_bwd_workspace(100, 768) > 0
    assert lib.vitk_attention_bwd_workspace(2, 197, 12) == 0 and lib.vitk_attention_bwd_workspace(2, 577, 12) == 0
    assert lib.vitk_attention_bwd(p16, p16, p16, p16, p16, 1, 700, 1, 0.125, None, 0, None) == -22       # N > 608


def test_ops_reject_cpu_tensors():
    L = _lib()
    from vision_transformers_torch_xla_amd import ops
    with pytest.raises(L.VitkError):
        ops.layernorm_fwd(torch.zeros(4, 64, dtype=torch.bfloat16), torch.ones(64), torch.zeros(64))
    with pytest.raises(L.VitkError):
        ops.gemm_nt(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))


# ------------------------------------------------------------------------------------------------ host mirror
def test_create_model_registry_and_errors():
    from vision_transformers_torch_xla_amd._lib import VitkError
    from vision_transformers_torch_xla_amd.vit import create_model, is_model, list_models
    m = create_model("vit_tiny_patch16_224", pretrained=False, num_classes=1000, global_pool="avg", drop_path_rate=0.1,
                     drop_rate=None)       # None kwargs are dropped (models/_factory.py:108)
    assert sum(p.numel() for p in m.parameters()) == 5717416
    assert [round(b.drop_path, 4) for b in m.blocks][:2] == [0.0, round(0.1 / 11, 4)] and abs(m.blocks[-1].drop_path - 0.1) < 1e-7
    assert hasattr(m, "fc_norm") and not hasattr(m, "norm")
    t = create_model("vit_tiny_patch16_224")                                                     # timm default: pool='token'
    assert hasattr(t, "norm") and not hasattr(t, "fc_norm")
    assert create_model("vit_base_patch16_384", depth=1).num_tokens == 577
    d = create_model("deit_tiny_distilled_patch16_224", depth=1)
    assert d.num_prefix_tokens == 2 and "dist_token" in dict(d.named_parameters()) and "head_dist.weight" in d.state_dict()
    assert is_model("my_vit_b") and "vit_large_patch16_224" in list_models()
    with pytest.raises(RuntimeError, match=r"Unknown model \(convnext_tiny\)"):
        create_model("convnext_tiny")
    with pytest.raises(RuntimeError, match="pretrained"):
        create_model("vit_tiny_patch16_224", pretrained=True)
    # every size of models/my_vit.py:84-165 builds: head dims 48 (mini) and 72 (xs) beside 64
    assert [(create_model(n, depth=1).head_dim, create_model(n, depth=1).num_heads) for n in ("my_vit_mini", "my_vit_ti", "my_vit_xs", "my_vit_s")] == \
        [(48, 3), (64, 3), (72, 4), (64, 6)]
    with pytest.raises(VitkError, match="head dim"):
        create_model("vit_tiny_patch16_224", embed_dim=200, num_heads=4, depth=1)               # 50: not a multiple of 8
    with pytest.raises(VitkError, match="tokens exceed"):
        create_model("my_vit_xs", img_size=384, depth=1)                                         # 577 tokens need head dim <= 64
    assert create_model("my_vit_mini", img_size=384, depth=1)._pad_heads                          # 48 runs zero-padded to 64
    with pytest.raises(VitkError):
        m(torch.zeros(1, 3, 224, 224))                                                            # CPU input: no fallback
    assert m.no_weight_decay() == {"pos_embed", "cls_token", "dist_token"}


def test_state_dict_keys_match_oracle_and_roundtrip():
    from oracle.ref_vit import create_ref_model
    from vision_transformers_torch_xla_amd.vit import create_model
    for name, pool in (("vit_tiny_patch16_224", "avg"), ("vit_tiny_patch16_224", "token"), ("deit_tiny_distilled_patch16_224", "token")):
        ref = create_ref_model(name, num_classes=40, global_pool=pool, depth=2)
        m = create_model(name, num_classes=40, global_pool=pool, depth=2)
        assert list(m.state_dict().keys()) != [] and set(m.state_dict().keys()) == set(ref.state_dict().keys())
        for k, v in ref.state_dict().items():
            assert tuple(m.state_dict()[k].shape) == tuple(v.shape), k
        m.load_state_dict(ref.state_dict())
        for k, v in m.state_dict().items():
            assert torch.equal(v, ref.state_dict()[k])
        assert m._views_ok()                        # load_state_dict copies into the flat buffer, parameters stay views


def test_flat_layout_invariants():
    from oracle import ref_ops as R
    from vision_transformers_torch_xla_amd.vit import create_model
    m = create_model("deit_tiny_distilled_patch16_224", num_classes=40, depth=3)
    offs, names = m._offs, m._pnames
    # contiguous, non-overlapping, 16-byte aligned, decay parameters first
    prev_end = 0
    for n, p in zip(names, m._plist):
        assert offs[n] >= prev_end and offs[n] % 4 == 0
        prev_end = offs[n] + p.numel()
        assert p.data_ptr() == m._flat["p"].data_ptr() + 4 * offs[n]
    assert prev_end <= m._total
    groups = R.parameter_groups(m, 0.05, m.no_weight_decay())
    decay_names = next(g["names"] for g in groups if g["weight_decay"] > 0)
    assert set(names[:len(decay_names)]) == set(decay_names)
    assert all(offs[n] < m._decay_end for n in decay_names) and all(offs[n] >= m._decay_end for n in names[len(decay_names):])
    assert offs["dist_token"] == offs["cls_token"] + m.embed_dim                 # one [prefix, D] table
    # gradients become final in increasing address order: head, then blocks L-1..0
    ends = [m._head_end] + [m._layer_end[i] for i in reversed(range(m.depth))]
    assert ends == sorted(ends) and ends[-1] <= m._decay_end
    # deepcopy / zero_grad
    import copy
    m.blocks[0].norm1.weight.requires_grad_(False)
    m._grad_ready_hook = lambda *a, **k: None
    c = copy.deepcopy(m)
    assert c._views_ok() and torch.equal(c._flat["p"], m._flat["p"]) and c._flat["p"].data_ptr() != m._flat["p"].data_ptr()
    assert [n for n, _ in c.named_parameters()] == [n for n, _ in m.named_parameters()]
    assert [p.requires_grad for p in c.parameters()] == [p.requires_grad for p in m.parameters()]
    # runtime objects are rebuilt, not shared or copied; an evaluation copy holds no gradient buffer
    assert c._pin_ring is not m._pin_ring and c._ws is not m._ws and c._grad_ready_hook is None
    assert c._flat["g"] is None and m._flat["g"] is not None
    assert c._ensure_grad_buffer().shape == m._flat["g"].shape and c._gview["head.weight"].shape == m.head.weight.shape
    # timm's parameter order for the distilled model (optimizer state dicts are matched by index)
    assert [n for n, _ in m.named_parameters()][:3] == ["cls_token", "pos_embed", "dist_token"]
    # a re-flatten drops the transposed-weight table together with the buffers it describes
    m._wt_plan, m._wt_plan_key = "stale", ("stale",)
    m._flatten()
    assert m._wt_plan is None and m._wt_plan_key is None and m._wt == {}


def test_parameter_groups_and_optimizer_plan():
    from oracle import ref_ops as R
    from vision_transformers_torch_xla_amd import optim_factory
    from vision_transformers_torch_xla_amd.vit import create_model
    m = create_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", depth=2)
    groups = optim_factory.get_parameter_groups(m, 0.05, m.no_weight_decay())
    want = R.parameter_groups(m, 0.05, m.no_weight_decay())
    assert [(g["weight_decay"], g["lr_scale"], len(g["params"])) for g in groups] == [(g["weight_decay"], g["lr_scale"], len(g["params"])) for g in want]

    class Args:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 4e-3, 0.05, 1e-8, None
    opt = optim_factory.create_optimizer(Args, m)
    assert isinstance(opt, optim_factory.FusedAdamW) and len(opt.param_groups) == 2
    opt._build_plan()
    plan = {g["weight_decay"]: ranges for g, (ranges, loose) in zip(opt.param_groups, opt._plan)}
    assert plan[0.05] == [(0, m._decay_end)] and plan[0.0] == [(m._decay_end, m._total)]       # one launch per group
    assert all(not loose for _, loose in opt._plan)
    sd = opt.state_dict()
    assert len(sd["state"]) == len(list(m.parameters())) and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}

    class Bad:
        opt, lr, weight_decay = "sgd", 0.1, 0.0
    from vision_transformers_torch_xla_amd._lib import VitkError
    with pytest.raises(VitkError):
        optim_factory.create_optimizer(Bad, m)


def test_optimizer_replan_keeps_moments_and_checks_shapes():
    """A re-flattened model (.to(), .cuda(), a loaded state) makes the optimizer rebuild its flat moment buffers: what was
    gathered so far must move into the new ones (the step count, hence the bias correction, carries on), and a state dict of
    another parameter layout must be refused instead of being broadcast into the wrong moments."""
    from vision_transformers_torch_xla_amd import optim_factory
    from vision_transformers_torch_xla_amd._lib import VitkError
    from vision_transformers_torch_xla_amd.vit import create_model
    torch.manual_seed(0)
    m = create_model("deit_tiny_distilled_patch16_224", num_classes=16, depth=1)

    class Args:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
    opt = optim_factory.create_optimizer(Args, m)
    opt._build_plan()
    opt._m.uniform_(-1, 1); opt._v.uniform_(0, 1); opt._steps = 5
    before = {n: (opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()) for n, p in m.named_parameters()}
    old_m = opt._m
    m._flatten()                                           # what .to(device) does: new flat buffers, parameters re-pointed
    assert opt._plan_stale()
    opt._build_plan()
    assert opt._m is not old_m and opt._steps == 5
    for n, p in m.named_parameters():
        assert torch.equal(opt.state[p]["exp_avg"], before[n][0]) and torch.equal(opt.state[p]["exp_avg_sq"], before[n][1]), n
        assert float(opt.state[p]["step"]) == 5.0
    # a state dict whose entries 1 and 2 are swapped (a model that registers dist_token before pos_embed): shapes do not fit
    sd = opt.state_dict()
    name_of = {id(q): n for n, q in m.named_parameters()}
    idx = {name_of[id(p)]: k for g, sg in zip(opt.param_groups, sd["param_groups"]) for p, k in zip(g["params"], sg["params"])}
    sd["state"][idx["pos_embed"]], sd["state"][idx["dist_token"]] = sd["state"][idx["dist_token"]], sd["state"][idx["pos_embed"]]
    with pytest.raises(VitkError, match="does not fit"):
        opt.load_state_dict(sd)


def test_cosine_scheduler_matches_golden():
    from vision_transformers_torch_xla_amd import utils
    s = np.load(ROOT / "tests" / "golden" / "schedules.npz")
    assert np.array_equal(utils.cosine_scheduler(4e-3, 1e-6, 3, 7, warmup_epochs=1), s["lr_a"])
    assert np.array_equal(utils.cosine_scheduler(0.05, 0.05, 3, 7), s["wd_a"])
    assert np.array_equal(utils.cosine_scheduler(1e-3, 1e-5, 5, 4, warmup_epochs=2, warmup_steps=3), s["lr_b"])


def test_metric_logger_semantics():
    from vision_transformers_torch_xla_amd import utils
    ml = utils.MetricLogger(delimiter="  ")
    ml.update(loss=2.0, lr=0.1, skipped=None)
    ml.update(loss=torch.tensor(4.0))
    ml.meters["class_acc"].update(0.5, n=8)
    ml.meters["class_acc"].update(1.0, n=24)
    assert ml.loss.global_avg == 3.0 and ml.loss.count == 2 and ml.loss.value == 4.0
    assert abs(ml.meters["class_acc"].global_avg - 0.875) < 1e-12
    assert "skipped" not in ml.meters and "loss:" in str(ml)
    ml.synchronize_between_processes()                     # single process: no-op
    with pytest.raises(AttributeError):
        ml.nope
    # window statistics: the latest `window_size` values; median = the lower middle one (torch.median); totals over everything
    sv = utils.SmoothedValue(window_size=4, fmt="{median:.1f}|{avg:.2f}|{max:.0f}|{value:.0f}|{global_avg:.3f}")
    for v in (9.0, 1.0, 4.0, 2.0, 8.0):
        sv.update(v)
    want = torch.tensor([1.0, 4.0, 2.0, 8.0])
    assert sv.median == want.median().item() == 2.0 and sv.avg == want.mean().item() and sv.max == 8.0 and sv.value == 8.0
    assert sv.count == 5 and sv.total == 24.0 and str(sv) == "2.0|3.75|8|8|4.800"
    sv.count, sv.total = 10, 5.0                           # the cross-rank sum assigns them
    assert sv.global_avg == 0.5
    with pytest.raises(TypeError):
        ml.update(loss="high")
    top1, top5 = utils.accuracy(torch.tensor([[0.1, 0.9, 0.0, 0.2, 0.3, 0.4, 0.5], [0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.0]]), torch.tensor([1, 6]), (1, 5))
    assert top1.item() == 50.0 and top5.item() == 50.0


def test_init_distributions_follow_the_reference_rules():
    """The constructor's initialisation, as numbers (vision_transformer.py:634-648 init_weights with mode '', :998-1010
    init_weights_vit_timm; deit.py:49-51): every Linear weight and pos_embed ~ N(0, 0.02) truncated at +-2 (absolute, i.e. never
    reached), biases zero, LayerNorm at (1, 0), cls_token ~ N(0, 1e-6), dist_token like pos_embed, the patch convolution with
    torch's Conv2d default (uniform within +-1/sqrt(fan_in) for weight and bias)."""
    from vision_transformers_torch_xla_amd.vit import create_model
    torch.manual_seed(0)
    m = create_model("deit_small_distilled_patch16_224", num_classes=1000)
    sd = dict(m.named_parameters())
    for name, p in sd.items():
        p = p.detach().double()
        if name.endswith(".weight") and p.dim() == 2:                                 # every Linear, heads included
            assert abs(p.mean().item()) < 4 * 0.02 / p.numel() ** 0.5 + 1e-6, name
            assert abs(p.std().item() - 0.02) < 0.02 * 0.02, name
            assert p.abs().max().item() < 0.02 * 6.5, name
            # a normal sample, not a uniform one: the excess kurtosis of N is 0, of U it is -1.2
            kurt = ((p - p.mean()) ** 4).mean().item() / p.var().item() ** 2 - 3.0
            assert abs(kurt) < 0.15, (name, kurt)
        elif "norm" in name and name.endswith(".weight"):
            assert torch.equal(p, torch.ones_like(p)), name
        elif name.endswith(".bias") and not name.startswith("patch_embed"):
            assert torch.equal(p, torch.zeros_like(p)), name
    for name in ("pos_embed", "dist_token"):
        p = sd[name].detach().double()
        assert abs(p.std().item() - 0.02) < 0.02 * 0.1 and p.abs().max().item() < 0.13, name
    cls = sd["cls_token"].detach().double()
    assert 0 < cls.abs().max().item() < 1e-5 and abs(cls.std().item() - 1e-6) < 3e-7
    w, b = sd["patch_embed.proj.weight"].detach().double(), sd["patch_embed.proj.bias"].detach().double()
    bound = 1.0 / (3 * 16 * 16) ** 0.5
    assert w.shape == (384, 3, 16, 16) and w.abs().max().item() <= bound and b.abs().max().item() <= bound
    assert abs(w.std().item() - bound / 3 ** 0.5) < 0.01 * bound          # uniform(-bound, bound): std = bound / sqrt(3)
    kurt = (w ** 4).mean().item() / w.var().item() ** 2 - 3.0
    assert abs(kurt + 1.2) < 0.05
    # stochastic-depth rates: linspace(0, rate, depth) over the blocks (vision_transformer.py:580)
    m2 = create_model("vit_small_patch16_224", drop_path_rate=0.1)
    assert np.allclose([blk.drop_path for blk in m2.blocks], np.linspace(0, 0.1, 12))
