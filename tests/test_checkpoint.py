"""Checkpoint save / resume / fine-tune loading (host logic; CPU only): timm key names, optimizer state round trip,
auto-resume picks the latest epoch, old checkpoints are pruned, a head of another size is dropped, and the position
table is resampled for another resolution."""
import argparse

import torch

from vision_transformers_torch_xla_amd import optim_factory, utils
from vision_transformers_torch_xla_amd.vit import create_model


def _args(tmp, **kw):
    a = argparse.Namespace(output_dir=str(tmp), save_ckpt_num=2, save_ckpt_freq=1, auto_resume=True, resume="", model_ema=False,
                           eval=False, start_epoch=0, opt="adamw", lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=None)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_save_resume_roundtrip(tmp_path):
    torch.manual_seed(0)
    m = create_model("vit_tiny_patch16_224", num_classes=16, global_pool="avg", depth=2)
    args = _args(tmp_path)
    opt = optim_factory.create_optimizer(args, m)
    opt._build_plan()
    opt._m.uniform_(-1, 1); opt._v.uniform_(0, 1); opt._steps = 7
    for st in opt.state.values():
        st["step"] = torch.tensor(7.0)
    for ep in range(4):
        utils.save_model(args, ep, m, m, opt, None)
    names = sorted(p.name for p in tmp_path.iterdir())
    assert names == ["checkpoint-2.pth", "checkpoint-3.pth"]            # 0 and 1 were pruned (keep 2)
    ck = torch.load(tmp_path / "checkpoint-3.pth", weights_only=True)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args"} and ck["epoch"] == 3 and ck["args"]["lr"] == 1e-3
    assert {"cls_token", "pos_embed", "patch_embed.proj.weight", "blocks.1.attn.qkv.bias", "blocks.0.mlp.fc2.weight", "fc_norm.weight",
            "head.bias"} <= set(ck["model"])

    torch.manual_seed(1)
    m2 = create_model("vit_tiny_patch16_224", num_classes=16, global_pool="avg", depth=2)
    args2 = _args(tmp_path)
    opt2 = optim_factory.create_optimizer(args2, m2)
    utils.auto_load_model(args2, m2, m2, opt2, None)
    assert args2.resume.endswith("checkpoint-3.pth") and args2.start_epoch == 4
    for (n1, p1), (n2, p2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2)
    for p1, p2 in zip(m.parameters(), m2.parameters()):                 # moments of every parameter (padding is not state)
        assert torch.equal(opt.state[p1]["exp_avg"], opt2.state[p2]["exp_avg"])
        assert torch.equal(opt.state[p1]["exp_avg_sq"], opt2.state[p2]["exp_avg_sq"])
        assert opt2.state[p2]["exp_avg"].data_ptr() >= opt2._m.data_ptr()      # still views of the flat moment buffer
    assert opt2._steps == 7
    assert m2._views_ok()                                               # parameters still alias the flat buffer


def test_finetune_drops_head_and_resamples_pos_embed(tmp_path):
    torch.manual_seed(0)
    src = create_model("vit_tiny_patch16_224", num_classes=16, global_pool="avg", depth=1)
    torch.save({"model": src.state_dict()}, tmp_path / "pre.pth")
    dst = create_model("vit_tiny_patch16_224", num_classes=8, global_pool="avg", depth=1, img_size=384)
    assert dst.pos_embed.shape == (1, 577, 192)
    head_before = dst.head.weight.detach().clone()
    missing, unexpected = utils.load_finetune_checkpoint(dst, tmp_path / "pre.pth")
    assert sorted(missing) == ["head.bias", "head.weight"] and unexpected == []
    assert torch.equal(dst.head.weight, head_before)                    # the mismatched head keeps its init
    assert torch.equal(dst.blocks[0].attn.qkv.weight, src.blocks[0].attn.qkv.weight)
    pe = dst.pos_embed.detach()
    assert torch.equal(pe[:, 0], src.pos_embed.detach()[:, 0])          # class token position carried over
    # resampling is smooth interpolation of the 14x14 grid: same mean level, corners close to the source corners
    g_src = src.pos_embed.detach()[0, 1:].reshape(14, 14, -1)
    g_dst = pe[0, 1:].reshape(24, 24, -1)
    assert abs(float(g_dst.mean()) - float(g_src.mean())) < 2e-3
    assert float((g_dst[0, 0] - g_src[0, 0]).abs().max()) < 0.05 and float(g_dst.std()) < float(g_src.std()) * 1.05
    # identity when the grid does not change
    assert utils.resample_abs_pos_embed(src.pos_embed.detach(), (14, 14)) is not None
    assert torch.equal(utils.resample_abs_pos_embed(src.pos_embed.detach(), (14, 14)), src.pos_embed.detach())


def test_finetune_from_1000_classes_onto_37(tmp_path):
    """main.py:653-674: --finetune loads a checkpoint trained on 1000 classes into a model with another head; the mismatched head is
    dropped, everything else is taken.  37 is not a multiple of 8 (the head then runs on zero-padded copies on the GPU)."""
    torch.manual_seed(0)
    src = create_model("vit_tiny_patch16_224", num_classes=1000, global_pool="avg", depth=1)
    torch.save({"model": src.state_dict()}, tmp_path / "in1k.pth")
    dst = create_model("vit_tiny_patch16_224", num_classes=37, global_pool="avg", depth=1)
    assert dst.head.weight.shape == (37, 192) and dst._pad_head and dst._cls_pad == 40
    missing, unexpected = utils.load_finetune_checkpoint(dst, tmp_path / "in1k.pth")
    assert sorted(missing) == ["head.bias", "head.weight"] and unexpected == []
    assert torch.equal(dst.blocks[0].mlp.fc1.weight, src.blocks[0].mlp.fc1.weight)
    assert set(dst.state_dict()) == set(src.state_dict()) and dst.state_dict()["head.weight"].shape == (37, 192)


def test_create_model_checkpoint_path_formats(tmp_path):
    """create_model(checkpoint_path=...) as models/_factory.py:151-152 -> models/_helpers.py:44-140: a '.safetensors' file or a
    torch file; EMA weights ('state_dict_ema', 'model_ema') before 'state_dict' before 'model' before the bare dict; 'module.' and
    '_orig_mod.' prefixes stripped; strict; a missing file raises FileNotFoundError.  --finetune reads safetensors too."""
    import pytest
    torch.manual_seed(0)
    src = create_model("vit_tiny_patch16_224", num_classes=16, global_pool="avg", depth=1)
    sd = {k: v.detach().clone() for k, v in src.state_dict().items()}
    other = {k: v + 1.0 for k, v in sd.items()}

    def same(model, want):
        return all(torch.equal(v, want[k]) for k, v in model.state_dict().items())
    kw = dict(num_classes=16, global_pool="avg", depth=1)
    utils.save_safetensors(src, tmp_path / "m.safetensors")
    assert same(create_model("vit_tiny_patch16_224", checkpoint_path=str(tmp_path / "m.safetensors"), **kw), sd)
    torch.save(sd, tmp_path / "bare.pth")
    assert same(create_model("vit_tiny_patch16_224", checkpoint_path=str(tmp_path / "bare.pth"), **kw), sd)
    torch.save({"model": sd, "model_ema": other, "epoch": 3}, tmp_path / "ema.pth")
    assert same(create_model("vit_tiny_patch16_224", checkpoint_path=str(tmp_path / "ema.pth"), **kw), other)       # EMA weights first
    assert all(torch.equal(v, sd[k]) for k, v in utils.load_checkpoint_state_dict(str(tmp_path / "ema.pth"), use_ema=False).items())
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}, "model": other}, tmp_path / "ddp.pth")
    assert same(create_model("vit_tiny_patch16_224", checkpoint_path=str(tmp_path / "ddp.pth"), **kw), sd)          # 'state_dict' before 'model'
    torch.save({"_orig_mod." + k: v for k, v in sd.items()}, tmp_path / "compiled.pth")
    assert same(create_model("vit_tiny_patch16_224", checkpoint_path=str(tmp_path / "compiled.pth"), **kw), sd)
    with pytest.raises(FileNotFoundError):
        create_model("vit_tiny_patch16_224", checkpoint_path=str(tmp_path / "nothing.pth"), **kw)
    with pytest.raises(RuntimeError):                                    # strict: another head size is an error on this path
        create_model("vit_tiny_patch16_224", checkpoint_path=str(tmp_path / "bare.pth"), num_classes=24, global_pool="avg", depth=1)
    # the fine-tune path reads the same file formats and drops the head instead
    m = create_model("vit_tiny_patch16_224", num_classes=24, global_pool="avg", depth=1)
    missing, unexpected = utils.load_finetune_checkpoint(m, str(tmp_path / "m.safetensors"))
    assert sorted(missing) == ["head.bias", "head.weight"] and not unexpected
    assert torch.equal(m.state_dict()["blocks.0.attn.qkv.weight"], sd["blocks.0.attn.qkv.weight"])
