"""Crop / resize stage of the input pipeline on the device (vitk_resample_images, data.ResampleLoader) against the CPU oracle's
restatement of tf.image.resize(antialias=True) (oracle/ref_resize.py), and the oracle itself against an independent
implementation (torch's antialiased interpolation, the same triangle / Keys-cubic filters written by other people).
Bit-exact where both sides run the same float32 operations in the same order; TensorFlow itself cannot run here, so parity
with the reference's host stage is unpinned (see the oracle's header)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_preprocess as RP
from oracle import ref_resize as RR


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # smooth content beside the noise, so that a wrong tap order or weight shows up as a pattern, not just as noise
    yy, xx = np.mgrid[0:h, 0:w]
    grad = ((yy * 255 // max(h - 1, 1)) ^ (xx * 255 // max(w - 1, 1))).astype(np.uint8)
    return np.where(rng.random((h, w, 1)) < 0.5, base, grad[..., None].repeat(3, 2)).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ oracle (CPU)
@pytest.mark.parametrize("kernel,mode", [("triangle", "bilinear"), ("keyscubic", "bicubic")])
@pytest.mark.parametrize("h,w,oh,ow", [(64, 48, 24, 24), (37, 91, 224, 224), (300, 200, 64, 80), (50, 50, 50, 50)])
def test_oracle_resize_matches_torch_antialias(kernel, mode, h, w, oh, ow):
    img = RR.to_float01(_img(h, w, h * 7 + w))
    got = RR.resize_antialias(img, oh, ow, kernel)
    want = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(oh, ow), mode=mode, antialias=True, align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(got - want).max() <= 2e-5, np.abs(got - want).max()
    if (h, w) == (oh, ow):
        assert np.array_equal(got, img)                    # same size: one tap of weight 1


def test_oracle_spans_and_geometry():
    # weights of every output index sum to 1 and stay inside the image; a constant image stays constant
    for in_size, out_size in ((500, 224), (100, 224), (224, 224), (3, 7), (4000, 224)):
        for k in ("triangle", "keyscubic"):
            for s, w in RR.compute_spans(in_size, out_size, k):
                assert 0 <= s and s + len(w) <= in_size and abs(float(w.sum()) - 1.0) <= 1e-5
    const = np.full((40, 30, 3), 0.625, np.float32)
    assert np.abs(RR.resize_antialias(const, 17, 23, "keyscubic") - 0.625).max() <= 1e-6
    # resize_small / central crop: the shorter side becomes `size`; crop offsets as tf.image.resize_with_crop_or_pad
    assert RR.resize_small_size(375, 500, 256) == (256, 341) and RR.resize_small_size(500, 375, 256) == (341, 256)
    assert RR.resize_small_size(100, 80, 256) == (320, 256)
    a = np.arange(10 * 12 * 3, dtype=np.float32).reshape(10, 12, 3)
    assert np.array_equal(RR.crop_or_pad(a, 6), a[2:8, 3:9])
    p = RR.crop_or_pad(a[:4, :5], 8)
    assert np.array_equal(p[2:6, 1:6], a[:4, :5]) and p.sum() == a[:4, :5].sum()
    # the sampled boxes respect the area and aspect bounds of decode_jpeg_and_inception_crop
    rng = np.random.default_rng(0)
    for _ in range(300):
        h, w = (int(v) for v in rng.integers(16, 800, 2))
        y, x, bh, bw = RR.sample_inception_box(h, w, rng)
        assert 0 <= y and 0 <= x and y + bh <= h and x + bw <= w and bh > 0 and bw > 0
        if (bh, bw) != (h, w):
            assert 0.08 * h * w - 1e-9 <= bh * bw <= h * w and 0.70 <= bw / bh <= 1.40


def test_host_geometry_matches_oracle():
    from vision_transformers_torch_xla_amd import data
    for seed in range(100):
        h, w = (int(v) for v in np.random.default_rng(seed).integers(12, 900, 2))
        assert data.sample_inception_box(h, w, np.random.default_rng(500 + seed)) == RR.sample_inception_box(h, w, np.random.default_rng(500 + seed))
        assert data.resize_small_size(h, w, 256) == RR.resize_small_size(h, w, 256)
    imgs = [_img(40, 60, 1), _img(300, 200, 2), _img(256, 256, 3)]
    flat, offs = data.pack_images(imgs)
    assert all(o % 16 == 0 for o in offs) and all(np.array_equal(flat[o:o + im.size].reshape(im.shape), im) for o, im in zip(offs, imgs))
    d = data.eval_descriptors(imgs, offs, 64, 56)
    for b, im in enumerate(imgs):
        nh, nw = RR.resize_small_size(im.shape[0], im.shape[1], 64)
        assert (d[b]["out_h"], d[b]["out_w"]) == (nh, nw) and (d[b]["win_y"], d[b]["win_x"]) == ((nh - 56) // 2, (nw - 56) // 2)
    small = data.eval_descriptors([_img(10, 30, 4)], [0], 20, 56)       # 20 x 60 after the resize: padded rows, cropped columns
    assert (small[0]["win_y"], small[0]["win_x"]) == (-18, 2)


def test_crop_origin_range_epoch_seeding_and_oversize_images():
    from vision_transformers_torch_xla_amd import data
    # origin uniform over [0, H - h): with a 2-row image and 1-row boxes the origin H - h itself is never drawn ...
    rng = np.random.default_rng(3)
    tops = set()
    for _ in range(400):
        y, x, bh, bw = data.sample_inception_box(40, 40, rng, area=(0.9, 0.95), ratio=(1.0, 1.0))
        if (bh, bw) != (40, 40):
            assert y + bh < 40 and x + bw < 40, (y, x, bh, bw)
            tops.add(y)
    assert tops and 0 in tops
    # images the kernel cannot take whole are halved (2 x 2 means) until they fit; others come back untouched
    im = _img(8, 8, 0)
    assert data.fit_for_resample(im, 224) is im
    wide = np.zeros((4, 2 * data.MAX_SOURCE_WIDTH, 3), np.uint8)
    wide[0::2, 0::2] = 3; wide[1::2, 0::2] = 4; wide[0::2, 1::2] = 5; wide[1::2, 1::2] = 7
    out = data.fit_for_resample(wide, 224)
    assert out.shape == (2, data.MAX_SOURCE_WIDTH, 3) and out.dtype == np.uint8 and (out == 5).all()      # (3+4+5+7+2) >> 2
    tall = np.full((data.MAX_ROW_REDUCTION * 8 + 9, 6, 3), 200, np.uint8)
    assert data.fit_for_resample(tall, 8).shape[0] <= data.MAX_ROW_REDUCTION * 8

    class Host(list):
        def set_epoch(self, e):
            self.epoch = e
    # set_epoch(e) -> the draws of Generator(seed + e), the same ones every time epoch e is iterated
    ld = data.ResampleLoader(Host(), "cpu", seed=7)
    ld.set_epoch(4)
    assert ld.batches.epoch == 4 and ld._epoch == 4


def test_resample_entry_point_validates_on_the_host():
    """bad descriptors are refused before anything is launched (no GPU needed to see that)"""
    from vision_transformers_torch_xla_amd import _lib, ops
    lib = _lib.load()
    d = np.zeros(1, dtype=ops.RESAMPLE_DESC)
    d[0] = (0, 10, 10, 0, 0, 10, 10, 8, 8, 0, 0, 0, 0)
    dummy = ctypes.create_string_buffer(4096)
    addr = ctypes.addressof(dummy) // 16 * 16 + 16

    def call(desc, src_bytes=300, size=8, kernel=0):
        return lib.vitk_resample_images(addr, src_bytes, desc.ctypes.data, addr, addr, 1, size, kernel, 0.0, 1.0, None, None, None)
    bad = d.copy(); bad[0]["box_w"] = 11
    assert call(bad) != 0 and b"outside" in lib.vitk_last_error()
    assert call(d, src_bytes=299) != 0 and b"does not fit" in lib.vitk_last_error()
    assert call(d, size=6) != 0 and call(d, kernel=2) != 0
    far = d.copy(); far[0]["src_h"] = far[0]["box_h"] = 6000; far[0]["out_h"] = 8
    assert call(far, src_bytes=6000 * 10 * 3) != 0 and b"filter taps" in lib.vitk_last_error()


# ------------------------------------------------------------------------------------------------- device (GPU)
def _oracle_batch(images, desc, size, kernel, normalize=True):
    out = []
    for b, im in enumerate(images):
        d = desc[b]
        box = RR.to_float01(im)[d["box_y"]:d["box_y"] + d["box_h"], d["box_x"]:d["box_x"] + d["box_w"]]
        r = RR.resize_antialias(box, int(d["out_h"]), int(d["out_w"]), kernel)
        win = np.zeros((size, size, 3), np.float32)
        y0, x0 = int(d["win_y"]), int(d["win_x"])
        ys, xs = max(y0, 0), max(x0, 0)
        ye, xe = min(y0 + size, r.shape[0]), min(x0 + size, r.shape[1])
        win[ys - y0:ye - y0, xs - x0:xe - x0] = r[ys:ye, xs:xe]
        out.append(RP.preprocess_image(win, bool(d["flip"]), 0.0, 1.0, "imagenet" if normalize else "none"))
    return torch.stack(out)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["triangle", "keyscubic"])
def test_resample_kernel_is_bit_exact_with_the_oracle(device, kernel):
    from vision_transformers_torch_xla_amd import data, ops
    from vision_transformers_torch_xla_amd.data import IMAGENET_DEFAULT_MEAN as M, IMAGENET_DEFAULT_STD as SD
    images = [_img(h, w, 10 + i) for i, (h, w) in enumerate([(120, 160), (64, 48), (333, 500), (40, 40), (90, 17), (256, 300)])]
    flat, offs = data.pack_images(images)
    size = 56
    boxes = [(10, 20, 100, 120), (0, 0, 64, 48), (33, 100, 280, 390), (5, 5, 30, 30), (0, 0, 90, 17), (100, 50, 57, 57)]   # down, same, up, 1:1
    train = data.train_descriptors(images, offs, size, None, boxes=boxes, flips=[0, 1, 1, 0, 1, 0])
    evald = data.eval_descriptors(images, offs, 64, size)
    evald[4]["flip"] = 1                                                   # a padded, flipped image
    src = torch.from_numpy(flat).to(device)
    for desc in (train, evald):
        got = ops.resample_images(src, desc, size, kernel=kernel, mean=M, std=SD)
        want = _oracle_batch(images, desc, size, kernel)
        assert got.shape == (len(images), 3, size, size)
        diff = (got.cpu() - want).abs().max().item()
        assert torch.equal(got.cpu(), want), diff
    raw = ops.resample_images(src, train, size, kernel=kernel)           # no normalisation: values stay in [0, 1] (+ cubic overshoot)
    assert torch.equal(raw.cpu(), _oracle_batch(images, train, size, kernel, normalize=False))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["triangle", "keyscubic"])
def test_resample_to_the_patch_matrix_is_bit_exact(device, kernel):
    """patch=p: crop + resize + flip + normalise written as the model's bf16 patch matrix == im2col of the fp32 result (oracle
    and two-kernel route), padded rows and flipped images included; the loader's patch option hands the same matrix on"""
    from vision_transformers_torch_xla_amd import _lib, data, ops
    from vision_transformers_torch_xla_amd.data import IMAGENET_DEFAULT_MEAN as M, IMAGENET_DEFAULT_STD as SD
    images = [_img(h, w, 30 + i) for i, (h, w) in enumerate([(120, 160), (64, 48), (333, 500), (40, 40), (90, 17)])]
    flat, offs = data.pack_images(images)
    size, p = 64, 16
    boxes = [(10, 20, 100, 120), (0, 0, 64, 48), (33, 100, 280, 390), (5, 5, 30, 30), (0, 0, 90, 17)]
    train = data.train_descriptors(images, offs, size, None, boxes=boxes, flips=[0, 1, 1, 0, 1])
    evald = data.eval_descriptors(images, offs, 72, size)
    evald[4]["flip"] = 1
    src = torch.from_numpy(flat).to(device)

    def matrix(nchw):
        B, C, H, W = nchw.shape
        return nchw.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B * (H // p) * (W // p), C * p * p).to(torch.bfloat16)
    for desc in (train, evald):
        got = ops.resample_images(src, desc, size, kernel=kernel, mean=M, std=SD, patch=p)
        assert got.dtype == torch.bfloat16 and got.shape == (len(images) * 16, 3 * p * p)
        assert torch.equal(got.cpu(), matrix(_oracle_batch(images, desc, size, kernel)))
        assert torch.equal(got, ops.im2col_patches(ops.resample_images(src, desc, size, kernel=kernel, mean=M, std=SD), p))
    with pytest.raises(_lib.VitkError, match="multiple of 8"):
        ops.resample_images(src, train, size, kernel=kernel, patch=12)
    with pytest.raises(_lib.VitkError, match="divides"):
        ops.resample_images(src, train, 56, kernel=kernel, patch=16)
    host = [(images, np.arange(len(images)))]
    (x, y), = list(data.ResampleLoader(host, device, size=size, train=False, resize=72, patch=p))
    evald[4]["flip"] = 0
    assert torch.equal(x.cpu(), matrix(_oracle_batch(images, evald, size, "keyscubic"))) and y.tolist() == [0, 1, 2, 3, 4]


@pytest.mark.gpu
def test_resample_loader_over_an_image_folder(device, tmp_path):
    """decode (PIL) -> pack -> GPU crop / resize / flip / normalise, for the training and the evaluation chain, against the oracle
    composed from the same decoded pixels and the loader's own draws"""
    from PIL import Image
    from vision_transformers_torch_xla_amd import data
    sizes = [(80, 120), (64, 64), (150, 100), (50, 210), (96, 96), (33, 47), (120, 90)]
    for i, (h, w) in enumerate(sizes):
        cls = tmp_path / f"class_{i % 3}"
        cls.mkdir(exist_ok=True)
        Image.fromarray(_img(h, w, 70 + i)).save(cls / f"im_{i}.png")
    folder = data.DecodedImageFolder(str(tmp_path), batch_size=3, shuffle=True, seed=5, workers=2)
    assert folder.classes == ["class_0", "class_1", "class_2"] and len(folder) == 2
    for train in (True, False):
        loader = data.ResampleLoader(folder, device, size=32, train=train, resize=40, seed=11)
        loader.set_epoch(2)
        seen = 0
        ref_folder = data.DecodedImageFolder(str(tmp_path), batch_size=3, shuffle=True, seed=5, workers=1)
        ref_folder.set_epoch(2)
        rng = np.random.default_rng(11 + 2)                             # epoch 2's draws: Generator(seed + epoch)
        for (x, y), (imgs, labels) in zip(loader, ref_folder):
            flat, offs = data.pack_images(imgs)
            desc = data.train_descriptors(imgs, offs, 32, rng) if train else data.eval_descriptors(imgs, offs, 40, 32)
            want = _oracle_batch(imgs, desc, 32, "triangle" if train else "keyscubic")
            assert x.shape == (3, 3, 32, 32) and x.dtype == torch.float32 and x.device.type == "cuda"
            assert torch.equal(x.cpu(), want)
            assert torch.equal(y.cpu(), torch.from_numpy(labels))
            seen += 1
        assert seen == 2


@pytest.mark.gpu
def test_training_and_evaluation_through_the_loaders(device, tmp_path):
    """an image folder -> ResampleLoader -> train_one_epoch / evaluate: the loaders hand the loops what they consume"""
    from PIL import Image
    from vision_transformers_torch_xla_amd import data, engine, optim_factory
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model
    for i in range(12):
        cls = tmp_path / f"c{i % 4}"
        cls.mkdir(exist_ok=True)
        Image.fromarray(_img(70 + 5 * i, 90 + 3 * i, 200 + i)).save(cls / f"{i}.jpg", quality=95)
    torch.manual_seed(0)
    model = create_model("vit_tiny_patch16_224", num_classes=8, global_pool="avg", depth=1, img_size=64).to(device)

    class A:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
    opt = optim_factory.create_optimizer(A, model)
    train = data.ResampleLoader(data.DecodedImageFolder(str(tmp_path), 4, shuffle=True, seed=1, workers=2), device, size=64, train=True, seed=1)
    stats = engine.train_one_epoch(model, LabelSmoothingCrossEntropy(0.1), train, opt, device, 0, None, 0, None, None, log_freq=1)
    assert set(stats) == {"loss", "lr", "class_acc"} and np.isfinite(stats["loss"]) and 1.0 < stats["loss"] < 4.0
    val = data.ResampleLoader(data.DecodedImageFolder(str(tmp_path), 6, drop_last=False, workers=2), device, size=64, train=False, resize=72)
    ev = engine.evaluate(val, model, device)
    assert set(ev) == {"loss", "acc1", "acc5"} and 0.0 <= ev["acc1"] <= ev["acc5"] <= 100.0
