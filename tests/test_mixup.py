"""Mixup / CutMix: the host logic (random draws) against the CPU oracle without a GPU; the device kernels bit-exact against
the oracle on the GPU.  The oracle restates timm.data.Mixup (not vendored by the reference, not installed): parity with
upstream is unpinned, see oracle/ref_mixup.py."""
import numpy as np
import pytest
import torch

from oracle import ref_mixup as RM


def _host():
    from vision_transformers_torch_xla_amd import mixup
    return mixup


@pytest.mark.parametrize("mix,cut", [(0.8, 1.0), (0.8, 0.0), (0.0, 1.0)])
def test_host_draws_match_oracle(mix, cut):
    """same numpy seed -> same (lambda, box) sequence from the product's host code and from the oracle"""
    M = _host()
    shape = (8, 3, 224, 224)
    m = M.Mixup(mixup_alpha=mix, cutmix_alpha=cut, prob=0.9, switch_prob=0.5, label_smoothing=0.1, num_classes=1000)
    np.random.seed(123)
    want = [RM.draw_params(mix, cut, 0.9, 0.5, shape) for _ in range(40)]
    np.random.seed(123)
    got = []
    for _ in range(40):
        lam, use_cutmix = m._params_per_batch()
        box = None
        if lam != 1. and use_cutmix:
            (yl, yh, xl, xh), lam = M.cutmix_bbox_and_lam(shape, lam, correct_lam=True)
            box = (int(yl), int(yh), int(xl), int(xh))
        got.append((lam, box))
    assert got == want
    assert any(b is not None for _, b in got) == (cut > 0)
    assert all(0.0 <= l <= 1.0 for l, _ in got)
    if mix > 0 and cut > 0:
        assert any(b is None and l != 1. for l, b in got)      # both kinds occur


def test_oracle_properties():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (6,), generator=g)
    # soft targets: rows sum to 1, argmax is one of the pair's labels, smoothing floor everywhere
    _, t = RM.apply(x, y, 0.3, None, 10, 0.1)
    assert torch.allclose(t.sum(1), torch.ones(6), atol=1e-6) and (t >= 0.1 / 10 * 0.999).all()
    assert all(int(t[b].argmax()) in (int(y[b]), int(y[5 - b])) for b in range(6))
    # lam = 1: identity on the pixels, plain smoothed one-hot
    x1, t1 = RM.apply(x, y, 1., None, 10, 0.1)
    assert torch.equal(x1, x) and torch.allclose(t1[0, y[0]], torch.tensor(0.91))
    # cutmix: inside the box the flipped batch, outside untouched; corrected lambda = 1 - box area share
    np.random.seed(5)
    lam, box = RM.draw_params(0.0, 1.0, 1.0, 0.5, x.shape)
    yl, yh, xl, xh = box
    assert abs(lam - (1 - (yh - yl) * (xh - xl) / 1024.0)) < 1e-12
    x2, _ = RM.apply(x, y, lam, box, 10, 0.1)
    assert torch.equal(x2[:, :, yl:yh, xl:xh], x.flip(0)[:, :, yl:yh, xl:xh])
    mask = torch.ones(32, 32, dtype=torch.bool)
    mask[yl:yh, xl:xh] = False
    assert torch.equal(x2[:, :, mask], x[:, :, mask])


@pytest.mark.parametrize("mode", ["elem", "pair"])
@pytest.mark.parametrize("mix,cut", [(0.8, 1.0), (0.8, 0.0), (0.0, 1.0)])
def test_per_sample_host_plan_matches_oracle(mode, mix, cut):
    """modes 'elem' / 'pair': same numpy seed -> the plan the host hands to the kernel describes exactly what the oracle's loops do
    (kinds, lambdas, boxes, the lambdas of the soft targets)"""
    from vision_transformers_torch_xla_amd import ops
    M = _host()
    B, C = 8, 10
    m = M.Mixup(mixup_alpha=mix, cutmix_alpha=cut, prob=0.8, switch_prob=0.5, mode=mode, label_smoothing=0.1, num_classes=C)
    g = torch.Generator().manual_seed(3)
    for it in range(6):
        x = torch.randn(B, 3, 32, 32, generator=g)
        y = torch.randint(0, C, (B,), generator=g)
        np.random.seed(50 + it)
        want_x, want_t, want_lam = RM.mix_per_sample(x, y, mode, mix, cut, 0.8, 0.5, C, 0.1)
        np.random.seed(50 + it)
        prm, lam_rows = m._sample_plan(x.shape)
        assert prm.dtype == ops.MIXUP_SAMPLE and prm.itemsize == 32 and lam_rows.dtype == np.float32
        assert np.array_equal(lam_rows, want_lam.astype(np.float32))
        # the plan applied with plain torch on the CPU = the oracle's mixed batch
        out = x.clone()
        for b in range(B):
            k, lam, oml, yl, yh, xl, xh, _ = prm[b].tolist()
            if k == 1:
                out[b] = x[b] * lam + x[B - 1 - b] * oml
            elif k == 2:
                out[b][:, yl:yh, xl:xh] = x[B - 1 - b][:, yl:yh, xl:xh]
        assert torch.equal(out, want_x)
        if mode == "pair":
            assert all(prm[b] == prm[B - 1 - b] for b in range(B)) and np.array_equal(lam_rows, lam_rows[::-1])
        assert torch.allclose(want_t.sum(1), torch.ones(B), atol=1e-6)


def test_per_sample_entry_point_validates_on_the_host():
    """vitk_mixup_elem refuses bad plans before anything is launched (no GPU needed to see that)"""
    import ctypes
    from vision_transformers_torch_xla_amd import _lib, ops
    lib = _lib.load()
    buf = ctypes.create_string_buffer(8192)
    addr = (ctypes.addressof(buf) + 15) // 16 * 16
    prm = np.zeros(4, dtype=ops.MIXUP_SAMPLE)
    assert prm.itemsize == 32                                         # struct vitk_mixup_sample

    def call(plan, B=4, C=3, H=8, W=8):
        return lib.vitk_mixup_elem(addr, B, C, H, W, plan.ctypes.data, addr, None)
    odd = np.zeros(3, dtype=ops.MIXUP_SAMPLE)
    assert call(odd, B=3) != 0 and b"even" in lib.vitk_last_error()
    bad = prm.copy(); bad[1]["kind"] = 3
    assert call(bad) != 0 and b"kind" in lib.vitk_last_error()
    box = prm.copy(); box[2] = (2, 0., 0., 0, 9, 0, 4, 0)            # rows [0, 9) of an 8-row image
    assert call(box) != 0 and b"outside" in lib.vitk_last_error()
    assert call(prm, C=1, H=3, W=3) != 0 and b"multiple of 4" in lib.vitk_last_error()
    assert lib.vitk_mixup_targets_elem(None, addr, 10, 4, 10, addr, 0.9, 0.01, None) != 0


def test_rejects_other_modes_and_odd_batches():
    from vision_transformers_torch_xla_amd._lib import VitkError
    M = _host()
    with pytest.raises(VitkError):
        M.Mixup(mode='half')
    with pytest.raises(AssertionError):
        M.Mixup()(torch.zeros(3, 3, 8, 8), torch.zeros(3, dtype=torch.long))
    with pytest.raises(VitkError, match="patch matrix"):              # what a loader with patch=16 yields is not an image batch
        M.Mixup()(torch.zeros(4 * 196, 768, dtype=torch.bfloat16), torch.zeros(4, dtype=torch.long))


@pytest.mark.gpu
@pytest.mark.parametrize("mix,cut,shape", [(0.8, 1.0, (8, 3, 224, 224)), (0.8, 0.0, (4, 3, 32, 32)), (0.0, 1.0, (6, 3, 64, 48)),
                                           (0.8, 1.0, (2, 1, 16, 16))])
def test_device_mixup_bit_exact(device, mix, cut, shape):
    M = _host()
    C = 1000 if shape[-1] == 224 else 10
    m = M.Mixup(mixup_alpha=mix, cutmix_alpha=cut, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=C)
    g = torch.Generator().manual_seed(shape[0] * 7 + shape[-1])
    kinds = set()
    for it in range(8):
        x = torch.randn(*shape, generator=g)
        y = torch.randint(0, C, (shape[0],), generator=g)
        np.random.seed(1000 + it)
        lam, box = RM.draw_params(mix, cut, 1.0, 0.5, shape)
        want_x, want_t = RM.apply(x, y, lam, box, C, 0.1)
        np.random.seed(1000 + it)
        got_x, got_t = m(x.to(device), y.to(device))
        assert torch.equal(got_x.cpu(), want_x), (it, lam, box)
        assert torch.equal(got_t.cpu(), want_t), (it, lam, box)
        kinds.add(box is not None)
    assert kinds == ({True, False} if (mix > 0 and cut > 0) else {cut > 0})


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["elem", "pair"])
@pytest.mark.parametrize("mix,cut,shape", [(0.8, 1.0, (8, 3, 224, 224)), (0.8, 0.0, (4, 3, 32, 32)), (0.0, 1.0, (6, 3, 64, 48)),
                                           (0.8, 1.0, (2, 1, 16, 18))])
def test_device_mixup_per_sample_modes_bit_exact(device, mode, mix, cut, shape):
    """modes 'elem' and 'pair' on the device, in place, against the oracle's loops over a clone of the batch"""
    M = _host()
    C = 1000 if shape[-1] == 224 else 10
    m = M.Mixup(mixup_alpha=mix, cutmix_alpha=cut, prob=0.8, switch_prob=0.5, mode=mode, label_smoothing=0.1, num_classes=C)
    g = torch.Generator().manual_seed(shape[0] * 11 + shape[-1])
    for it in range(6):
        x = torch.randn(*shape, generator=g)
        y = torch.randint(0, C, (shape[0],), generator=g)
        np.random.seed(2000 + it)
        want_x, want_t, _ = RM.mix_per_sample(x, y, mode, mix, cut, 0.8, 0.5, C, 0.1)
        np.random.seed(2000 + it)
        got_x, got_t = m(x.to(device), y.to(device))
        assert torch.equal(got_x.cpu(), want_x), (mode, it)
        assert torch.equal(got_t.cpu(), want_t), (mode, it)


@pytest.mark.gpu
def test_engine_with_mixup_and_soft_targets(device):
    """the default recipe of the reference: mixup on -> SoftTargetCrossEntropy (main.py:926-929), no class_acc meter"""
    from vision_transformers_torch_xla_amd import engine, optim_factory
    from vision_transformers_torch_xla_amd.losses import SoftTargetCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model
    M = _host()
    torch.manual_seed(0)
    np.random.seed(0)
    model = create_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", depth=2, drop_path_rate=0.1).to(device)

    class A:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
    opt = optim_factory.create_optimizer(A, model)
    mix = M.Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=40)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 3, 224, 224, generator=g)       # host batch, as a loader would hand it over
    y = torch.randint(0, 40, (8,), generator=g)
    stats = engine.train_one_epoch(model, SoftTargetCrossEntropy(), [(x.clone(), y)] * 6, opt, device, 0, None, 0, None, mix,
                                   update_freq=1, log_freq=1)
    assert "class_acc" not in stats and np.isfinite(stats["loss"]) and 2.0 < stats["loss"] < 5.0
