"""Input pipeline tail (flip_lr, value_range, HWC -> CHW, normalise): the oracle's arithmetic on the CPU; the HIP kernel and
the device loader bit-exact against the oracle on the GPU.  The reference holds no fixture for its input pipeline (its
TensorFlow side is not installed here either): parity with it is unpinned, see oracle/ref_preprocess.py."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ref_preprocess as RP


def test_oracle_arithmetic():
    img = np.arange(4 * 8 * 3, dtype=np.uint8).reshape(4, 8, 3) * 2
    t = RP.preprocess_image(img, flip=False, normalize="none")
    assert t.shape == (3, 4, 8) and t.dtype == torch.float32
    assert t[1, 2, 3].item() == np.float32(img[2, 3, 1]) / np.float32(255.0)
    f = RP.preprocess_image(img, flip=True, normalize="none")
    assert torch.equal(f, t.flip(2))
    n = RP.preprocess_image(img, flip=False)
    want = (np.float32(img[0, 0, 2]) / np.float32(255.0) - np.float32(0.406)) / np.float32(0.225)
    assert n[2, 0, 0].item() == want
    # value_range(-1, 1) on floats and the crop window
    fl = np.linspace(0, 1, 6 * 8 * 3, dtype=np.float32).reshape(6, 8, 3)
    r = RP.preprocess_image(fl, flip=False, vmin=-1.0, vmax=1.0, normalize="none", window=(1, 4, 4, 4))
    assert r.shape == (3, 4, 4) and r[0, 0, 0].item() == fl[1, 4, 0] * np.float32(2.0) + np.float32(-1.0)
    assert RP.central_crop_offsets(256, 341, 224) == (16, 58)


def test_abi_rejects_bad_windows_before_launch():
    from vision_transformers_torch_xla_amd import _lib as L
    lib = L.load()
    buf = ctypes.create_string_buffer(4096)
    p16 = (ctypes.cast(buf, ctypes.c_void_p).value + 15) // 16 * 16
    assert lib.vitk_preprocess_images(None, 1, None, p16, 1, 8, 8, 8, 8, 0, 0, 0.0, 1.0, None, None, None) == -22
    assert lib.vitk_preprocess_images(p16, 1, None, p16, 1, 8, 6, 8, 8, 0, 0, 0.0, 1.0, None, None, None) == -22
    assert b"multiple of 4" in lib.vitk_last_error()
    assert lib.vitk_preprocess_images(p16, 1, None, p16, 1, 8, 8, 8, 8, 1, 0, 0.0, 1.0, None, None, None) == -22
    assert b"outside" in lib.vitk_last_error()
    m = (ctypes.c_float * 3)(0.5, 0.5, 0.5)
    assert lib.vitk_preprocess_images(p16, 1, None, p16, 1, 8, 8, 8, 8, 0, 0, 0.0, 1.0, m, None, None) == -22


def test_patch_matrix_entry_points_validate_on_the_host():
    from vision_transformers_torch_xla_amd import _lib as L
    lib = L.load()
    buf = ctypes.create_string_buffer(4096)
    p16 = (ctypes.cast(buf, ctypes.c_void_p).value + 15) // 16 * 16
    # patch size: a multiple of 8 that divides the window
    assert lib.vitk_preprocess_patches(p16, 1, None, p16, 12, 1, 24, 24, 24, 24, 0, 0, 0.0, 1.0, None, None, None) == -22
    assert b"multiple of 8" in lib.vitk_last_error()
    assert lib.vitk_preprocess_patches(p16, 1, None, p16, 16, 1, 24, 32, 24, 32, 0, 0, 0.0, 1.0, None, None, None) == -22
    assert lib.vitk_preprocess_patches(p16, 1, None, p16, 8, 1, 16, 16, 16, 16, 1, 0, 0.0, 1.0, None, None, None) == -22
    assert b"outside" in lib.vitk_last_error()
    assert lib.vitk_preprocess_patches(None, 1, None, p16, 8, 1, 16, 16, 16, 16, 0, 0, 0.0, 1.0, None, None, None) == -22
    assert lib.vitk_resample_patches(p16, 300, p16, p16, p16, 12, 1, 24, 0, 0.0, 1.0, None, None, None) == -22
    assert b"multiple of 8" in lib.vitk_last_error()
    assert lib.vitk_resample_patches(p16, 300, p16, p16, None, 8, 1, 24, 0, 0.0, 1.0, None, None, None) == -22


def test_central_crop_window_matches_oracle():
    from vision_transformers_torch_xla_amd import data
    for h, w in ((256, 341), (256, 256), (300, 257), (224, 224)):
        top, left = RP.central_crop_offsets(h, w, 224)
        assert data.central_crop_window(h, w, 224) == (top, left, 224, 224)
    with pytest.raises(ValueError):
        data.central_crop_window(200, 300, 224)
    assert data.IMAGENET_DEFAULT_MEAN == RP.IMAGENET_DEFAULT_MEAN and data.IMAGENET_DEFAULT_STD == RP.IMAGENET_DEFAULT_STD


# ------------------------------------------------------------------------------------------------------- GPU
def _images(kind, B, H, W, seed):
    rng = np.random.default_rng(seed)
    if kind == "u8":
        return rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    return rng.random(size=(B, H, W, 3), dtype=np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["u8", "f32"])
@pytest.mark.parametrize("B,H,W,window", [(5, 224, 224, None), (3, 32, 20, None), (2, 256, 341, (16, 58, 224, 224)),
                                          (2, 40, 37, (3, 4, 32, 28)), (1, 384, 384, None), (0, 0, 0, None)])
def test_kernel_bit_exact(kind, B, H, W, window):
    from vision_transformers_torch_xla_amd import data, ops
    if B == 0:
        with pytest.raises(Exception):
            ops.preprocess_images(torch.empty((0, 8, 8, 3), dtype=torch.uint8, device="cuda"))
        return
    imgs = _images(kind, B, H, W, seed=B * 1000 + W)
    flips = np.random.default_rng(7).integers(0, 2, size=B).astype(np.uint8)
    flips[0] = 1
    dev = torch.device("cuda:0")
    src = torch.from_numpy(imgs).to(dev)
    fl = torch.from_numpy(flips).to(dev)
    for vr, norm in (((0.0, 1.0), "imagenet"), ((-1.0, 1.0), "none"), ((0.0, 1.0), "none")):
        mean, std = (data.IMAGENET_DEFAULT_MEAN, data.IMAGENET_DEFAULT_STD) if norm == "imagenet" else (None, None)
        got = ops.preprocess_images(src, flip=fl, window=window, value_range=vr, mean=mean, std=std)
        want = RP.preprocess_batch(imgs, flips, vmin=vr[0], vmax=vr[1], normalize=norm, window=window)
        assert got.dtype == torch.float32 and got.shape == want.shape
        assert torch.equal(got.cpu(), want), (kind, vr, norm, (got.cpu() - want).abs().max())
    nofl = ops.preprocess_images(src, window=window)
    assert torch.equal(nofl.cpu(), RP.preprocess_batch(imgs, None, normalize="none", window=window))


def _patch_matrix(nchw, p):
    """the model's patch matrix of an fp32 NCHW batch: row = patch (b, gy, gx), column = c * p^2 + ky * p + kx, rounded to bf16
    once (round to nearest even) - the operand order of the patch-embedding GEMM (vision_transformer.py PatchEmbed: Conv2d with
    kernel = stride = patch, weight [D, C, p, p] flattened)"""
    B, C, H, W = nchw.shape
    return nchw.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B * (H // p) * (W // p), C * p * p).to(torch.bfloat16)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["u8", "f32"])
@pytest.mark.parametrize("B,H,W,window,p", [(5, 224, 224, None, 16), (3, 32, 24, None, 8), (2, 256, 341, (16, 58, 224, 224), 16),
                                            (2, 40, 37, (3, 5, 32, 32), 16), (1, 384, 384, None, 32), (2, 224, 224, None, 14)])
def test_patch_matrix_output_bit_exact(kind, B, H, W, window, p):
    """patch=p: the bf16 patch matrix straight from the NHWC batch == im2col of the fp32 batch the two-kernel route writes"""
    from vision_transformers_torch_xla_amd import _lib, data, ops
    imgs = _images(kind, B, H, W, seed=B * 77 + W)
    flips = np.random.default_rng(11).integers(0, 2, size=B).astype(np.uint8)
    flips[0] = 1
    dev = torch.device("cuda:0")
    src = torch.from_numpy(imgs).to(dev)
    fl = torch.from_numpy(flips).to(dev)
    if p % 8:
        with pytest.raises(_lib.VitkError, match="multiple of 8"):
            ops.preprocess_images(src, flip=fl, window=window, patch=p)
        return
    for vr, norm in (((0.0, 1.0), "imagenet"), ((-1.0, 1.0), "none")):
        mean, std = (data.IMAGENET_DEFAULT_MEAN, data.IMAGENET_DEFAULT_STD) if norm == "imagenet" else (None, None)
        got = ops.preprocess_images(src, flip=fl, window=window, value_range=vr, mean=mean, std=std, patch=p)
        want = _patch_matrix(RP.preprocess_batch(imgs, flips, vmin=vr[0], vmax=vr[1], normalize=norm, window=window), p)
        assert got.dtype == torch.bfloat16 and got.shape == want.shape
        assert torch.equal(got.cpu(), want), (kind, vr, norm)
        two = ops.im2col_patches(ops.preprocess_images(src, flip=fl, window=window, value_range=vr, mean=mean, std=std), p)
        assert torch.equal(got, two)
    nofl = ops.preprocess_images(src, window=window, patch=p)
    assert torch.equal(nofl.cpu(), _patch_matrix(RP.preprocess_batch(imgs, None, normalize="none", window=window), p))


@pytest.mark.gpu
def test_model_takes_the_patch_matrix():
    """VisionTransformer.forward on the loader's patch matrix == forward on the fp32 batch (logits and every gradient, bit for
    bit), through train_one_epoch and evaluate as well (the batch size there comes from the labels)"""
    from vision_transformers_torch_xla_amd import data, engine, optim_factory, vit
    from vision_transformers_torch_xla_amd.losses import CrossEntropyLoss

    class A:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
    dev = torch.device("cuda:0")
    B, n = 8, 3
    host = [(_images("u8", B, 64, 64, seed=40 + i), (np.arange(B) * 3 + i) % 16) for i in range(n)]

    def run(patch):
        torch.manual_seed(5)
        m = vit.VisionTransformer(img_size=64, patch_size=16, embed_dim=128, depth=2, num_heads=2, num_classes=16).to(dev)
        opt = optim_factory.create_optimizer(A, m)
        loader = data.DeviceLoader(host, dev, flip_lr=True, seed=1, patch=patch)
        first = next(iter(loader))[0]
        stats = engine.train_one_epoch(m, CrossEntropyLoss(), loader, opt, dev, epoch=0)
        ev = engine.evaluate(data.DeviceLoader(host, dev, patch=patch), m, dev)
        return first, stats, ev, {k: v.detach().clone() for k, v in m.state_dict().items()}

    img, s_img, e_img, w_img = run(None)
    pat, s_pat, e_pat, w_pat = run(16)
    assert pat.dtype == torch.bfloat16 and pat.shape == (B * 16, 3 * 256) and img.shape == (B, 3, 64, 64)
    assert s_img == s_pat and e_img == e_pat
    for k in w_img:
        assert torch.equal(w_img[k], w_pat[k]), k
    m = vit.VisionTransformer(img_size=64, patch_size=16, embed_dim=128, depth=2, num_heads=2, num_classes=16).to(dev)
    with pytest.raises(Exception, match="patch matrix"):
        m(pat[:, :100].contiguous())


@pytest.mark.gpu
def test_every_u8_value_and_flip_is_an_involution():
    from vision_transformers_torch_xla_amd import data, ops
    dev = torch.device("cuda:0")
    imgs = np.zeros((2, 16, 16, 3), dtype=np.uint8)
    imgs.reshape(-1)[:] = np.arange(imgs.size) % 256          # all 256 byte values in every channel position
    src = torch.from_numpy(imgs).to(dev)
    got = ops.preprocess_images(src, mean=data.IMAGENET_DEFAULT_MEAN, std=data.IMAGENET_DEFAULT_STD)
    assert torch.equal(got.cpu(), RP.preprocess_batch(imgs, None))
    ones = torch.ones(2, dtype=torch.uint8, device=dev)
    flipped = ops.preprocess_images(src, flip=ones, mean=data.IMAGENET_DEFAULT_MEAN, std=data.IMAGENET_DEFAULT_STD)
    assert torch.equal(flipped.flip(3), got)


@pytest.mark.gpu
def test_device_loader_matches_host_pipeline():
    from vision_transformers_torch_xla_amd import data
    B, n = 6, 5
    host = [(_images("u8", B, 256, 288, seed=i), np.arange(B) + 10 * i) for i in range(n)]
    loader = data.DeviceLoader(host, "cuda:0", flip_lr=True, crop=224, seed=3)
    assert len(loader) == n
    for epoch in (0, 1):                                     # no set_epoch: the loader counts its own iterations from 0
        gen = torch.Generator().manual_seed(3 + epoch)
        seen = 0
        for (samples, targets), (imgs, labels) in zip(loader, host):
            flips = (torch.rand(B, generator=gen) < 0.5).numpy()
            want = RP.preprocess_batch(imgs, flips, window=(16, 32, 224, 224))
            assert samples.is_cuda and samples.shape == (B, 3, 224, 224) and targets.dtype == torch.int64
            assert torch.equal(samples.cpu(), want) and torch.equal(targets.cpu(), torch.as_tensor(labels))
            seen += 1
        assert seen == n
    # float sources without flips or normalisation pass through value_range(0, 1) unchanged
    hostf = [(_images("f32", 2, 32, 32, seed=9), np.array([1, 2]))]
    (s, t), = list(data.DeviceLoader(hostf, "cuda:0", normalize="none"))
    assert torch.equal(s.cpu(), torch.from_numpy(hostf[0][0]).permute(0, 3, 1, 2)) and t.tolist() == [1, 2]
