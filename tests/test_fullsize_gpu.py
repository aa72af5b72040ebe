"""The kernels at the full sizes of the benchmark workloads - ViT-B/16 224 px at 512 images per GPU (M = 100 864 token rows,
D = 768, 12 heads, F = 3072), ViT-L/16 224 px at 256 images (M = 50 432, D = 1024, 16 heads, F = 4096: 3.08 tile rounds of the
persistent GEMM, 4 096 heads of attention) and ViT-B/16 384 px at 128 images (577 tokens per image, M = 73 856) - checked through
properties that need no full-size CPU run:

  * sampled rows / entries against an fp64 product of exactly those rows (the CPU oracle's arithmetic on a sample),
  * linearity of the weight-gradient GEMM over a split of the token rows, its bias gradient against column sums,
  * softmax attention with constant keys (every probability 1/N: out = mean of V, dQ = 0, dK rows sum to 0),
  * LayerNorm rows (zero mean, unit variance with gamma = 1, beta = 0; dx rows sum to zero),
  * two training steps from the same state are bit-identical and leave finite parameters.

Every call goes through the C ABI (vision_transformers_torch_xla_amd.ops)."""
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu

bf16, f32 = torch.bfloat16, torch.float32
# name -> (images, tokens per image, width, heads, MLP width, model, image size)
CONFIGS = {"vitb224": (512, 197, 768, 12, 3072, "vit_base_patch16_224", 224),
           "vitl224": (256, 197, 1024, 16, 4096, "vit_large_patch16_224", 224),
           "vitb384": (128, 577, 768, 12, 3072, "vit_base_patch16_384", 384)}


@pytest.fixture(params=list(CONFIGS))
def cfg(request):
    return CONFIGS[request.param]


def _ops():
    from vision_transformers_torch_xla_amd import ops
    return ops


def _rand(shape, gen, device, scale=1.0):
    return (torch.randn(shape, generator=gen, device=device) * scale).to(bf16)


@pytest.mark.parametrize("shape,epi", [("qkv", "bias"), ("fc2", "res"), ("fc1", "gelu"), ("dqkv", "plain")])
def test_gemm_nt_full_size_sampled_rows(device, cfg, shape, epi):
    from oracle import ref_ops as R
    B, NTOK, D, H, F = cfg[:5]
    M = B * NTOK
    N, K = {"qkv": (3 * D, D), "fc2": (D, F), "fc1": (F, D), "dqkv": (D, 3 * D)}[shape]
    from vision_transformers_torch_xla_amd import _lib
    ops = _ops()
    g = torch.Generator(device=device).manual_seed(N + K)
    a, w = _rand((M, K), g, device), _rand((N, K), g, device, 0.05)
    bias = torch.randn(N, generator=g, device=device)
    res = _rand((M, N), g, device)
    rows = torch.cat([torch.tensor([0, 1, 255, 256, M - 257, M - 2, M - 1]), torch.randint(0, M, (57,))]).to(device)
    acc = a[rows].double() @ w.double().t()
    if epi == "plain":
        out = ops.gemm_nt(a, w)
        want = acc
    elif epi == "bias":
        out = ops.gemm_nt(a, w, epilogue=_lib.EPI_BIAS, bias=bias)
        want = acc + bias.double()
    elif epi == "res":
        out = ops.gemm_nt(a, w, epilogue=_lib.EPI_BIAS_RES, bias=bias, res=res)
        want = acc + bias.double() + res[rows].double()
    else:
        out2 = torch.empty((M, N), dtype=bf16, device=device)
        out = ops.gemm_nt(a, w, epilogue=_lib.EPI_BIAS_GELU, bias=bias, out2=out2)
        want = acc + bias.double()
        h = out[rows].float().cpu()                      # GELU is applied to the stored (rounded) pre-activation
        assert rel_l2(out2[rows].float().cpu(), R.gelu(h).to(bf16).float()) <= 1e-3
        # the variant the model runs: the derivative is kept instead of h; the backward GEMM multiplies by it
        out2b = torch.empty((M, N), dtype=bf16, device=device)
        dg = ops.gemm_nt(a, w, epilogue=_lib.EPI_BIAS_GELU_DG, bias=bias, out2=out2b)
        hf = want.float().cpu()                          # here the pre-activation stays in fp32: nothing stores it
        assert rel_l2(out2b[rows].float().cpu(), R.gelu(hf).to(bf16).float()) <= 1e-3
        assert rel_l2(dg[rows].float().cpu(), R.dgelu(hf).to(bf16).float()) <= 1e-3
        wt = _rand((K, N), g, device, 0.05)              # dh[M, K] = (dout[M, N] @ wt[K, N]^T) * aux[M, K]
        aux = _rand((M, K), g, device)
        dh = ops.gemm_nt(out2, wt, epilogue=_lib.EPI_MUL, aux=aux)
        want_dh = (out2[rows].double() @ wt.double().t()) * aux[rows].double()
        assert rel_l2(dh[rows].float().cpu(), want_dh.float().to(bf16).float().cpu()) <= 1e-3
    assert rel_l2(out[rows].float().cpu(), want.float().to(bf16).float().cpu()) <= 1e-3
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("shape", ["qkv", "proj", "fc1", "fc2"])
def test_gemm_tn_full_size_linearity_and_bias(device, cfg, shape):
    ops = _ops()
    B, NTOK, D, H, F = cfg[:5]
    M = B * NTOK
    N, K = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (F, D), "fc2": (D, F)}[shape]
    g = torch.Generator(device=device).manual_seed(3 * N + K)
    a, b = _rand((M, N), g, device), _rand((M, K), g, device)
    whole = torch.empty((N, K), device=device)
    db = torch.empty(N, device=device)
    ops.gemm_tn(a, b, whole, dbias=db)
    # contraction rows split at an odd place: the two partial products must add up to the whole one
    cut = NTOK * (B // 2 - 45) + 5
    parts = torch.empty((N, K), device=device)
    dbp = torch.empty(N, device=device)
    ops.gemm_tn(a[:cut], b[:cut], parts, dbias=dbp)
    ops.gemm_tn(a[cut:], b[cut:], parts, dbias=dbp, accumulate=True)
    print(f"gemm_tn N={N} K={K}: split vs whole {rel_l2(parts, whole):.2e}, bias gradient vs fp64 sums {rel_l2(db.cpu(), a.double().sum(0).float().cpu()):.2e}")
    assert rel_l2(parts, whole) <= 2e-6
    assert rel_l2(dbp, db) <= 2e-6
    assert rel_l2(db.cpu(), a.double().sum(0).float().cpu()) <= 1e-5
    # sampled entries against fp64 dot products over all token rows
    ni, ki = torch.randint(0, N, (48,)), torch.randint(0, K, (48,))
    want = (a[:, ni.to(device)].double() * b[:, ki.to(device)].double()).sum(0)
    assert rel_l2(whole[ni.to(device), ki.to(device)].cpu(), want.float().cpu()) <= 1e-5
    again = torch.empty((N, K), device=device)
    ops.gemm_tn(a, b, again)
    assert torch.equal(again, whole)                      # fixed summation order, with or without the bias gradient


def test_attention_full_size_constant_keys(device, cfg):
    ops = _ops()
    B, NTOK, D, H, F = cfg[:5]
    M = B * NTOK
    g = torch.Generator(device=device).manual_seed(7)
    qkv = _rand((M, 3 * D), g, device)
    kconst = _rand((1, D), g, device)
    qkv[:, D:2 * D] = kconst                              # every key of a head is the same vector: uniform probabilities
    out, lse = ops.attention_fwd(qkv, B, NTOK, H, 0.125)
    v = qkv[:, 2 * D:].float().view(B, NTOK, D)
    want = v.mean(1, keepdim=True).expand(B, NTOK, D).reshape(M, D)
    assert rel_l2(out.float(), want.to(bf16).float()) <= 2e-3
    s = (qkv[:, :D].float().view(M, H, 64) * kconst.float().view(1, H, 64)).sum(-1) * 0.125        # the one score of a row
    want_lse = (s + torch.log(torch.tensor(float(NTOK)))).view(B, NTOK, H).permute(0, 2, 1)
    assert (lse - want_lse).abs().max().item() <= 2e-3 * max(1.0, want_lse.abs().max().item())
    dout = _rand((M, D), g, device)
    dqkv = ops.attention_bwd(qkv, out, dout, lse, B, NTOK, H, 0.125)
    dq, dk, dv = dqkv[:, :D].float(), dqkv[:, D:2 * D].float(), dqkv[:, 2 * D:].float()
    # dS rows sum to zero and all keys are equal: dQ = 0 up to the rounding of P and dS to bf16
    print(f"constant keys: out rel {rel_l2(out.float(), want.to(bf16).float()):.2e}, max |dQ| {dq.abs().max().item():.2e} "
          f"(max |dO| {dout.float().abs().max().item():.2f}), max |sum_keys dK| {dk.view(B, NTOK, D).sum(1).abs().max().item():.2e} (max |dK| {dk.abs().max().item():.2e})")
    assert dq.abs().max().item() <= 4e-3 * dout.float().abs().max().item()
    # dV[key] = sum_q P[q, key] dO[q] = mean over the queries of dO, the same for every key of a (batch, head)
    want_dv = dout.float().view(B, NTOK, D).mean(1, keepdim=True).expand(B, NTOK, D).reshape(M, D)
    assert rel_l2(dv, want_dv.to(bf16).float()) <= 5e-3
    # sum over keys of dK = sum_q (sum_key dS[q, key]) q * scale = 0
    assert dk.view(B, NTOK, D).sum(1).abs().max().item() <= 4e-3 * max(1.0, dk.abs().max().item()) * NTOK ** 0.5
    assert torch.isfinite(dqkv.float()).all()


def test_layernorm_full_size_row_properties(device, cfg):
    ops = _ops()
    B, NTOK, D, H, F = cfg[:5]
    M = B * NTOK
    g = torch.Generator(device=device).manual_seed(9)
    x = _rand((M, D), g, device, 3.0) + 1.5
    ones, zeros = torch.ones(D, device=device), torch.zeros(D, device=device)
    y, mean, rstd = ops.layernorm_fwd(x, ones, zeros)
    yf = y.float()
    assert yf.mean(1).abs().max().item() <= 4e-3          # bf16 rounding of 768 values of size <= 4
    assert (yf.var(1, unbiased=False) - 1).abs().max().item() <= 1e-2
    xf = x.float()
    assert (mean - xf.mean(1)).abs().max().item() <= 1e-4
    assert rel_l2(rstd, (xf.var(1, unbiased=False) + 1e-6).rsqrt()) <= 1e-5
    dy = _rand((M, D), g, device)
    dgamma, dbeta = torch.empty(D, device=device), torch.empty(D, device=device)
    dx = ops.layernorm_bwd(dy, x, mean, rstd, ones, dgamma, dbeta)
    # with gamma = 1 every row of dx sums to zero and is orthogonal to the normalised row
    scale = (dy.float().abs().max() * rstd.max()).item()
    print(f"layernorm rows: max |mean y| {yf.mean(1).abs().max().item():.2e}, max |var y - 1| {(yf.var(1, unbiased=False) - 1).abs().max().item():.2e}, "
          f"max |sum dx| {dx.float().sum(1).abs().max().item():.2e} (scale {scale:.2f})")
    assert dx.float().sum(1).abs().max().item() <= 4e-3 * scale * D ** 0.5
    assert rel_l2(dbeta.cpu(), dy.double().sum(0).float().cpu()) <= 1e-5
    xhat = (xf - mean[:, None]) * rstd[:, None]
    assert rel_l2(dgamma.cpu(), (dy.double() * xhat.double()).sum(0).float().cpu()) <= 1e-4


def test_training_steps_full_size_are_reproducible(device, cfg):
    """Two runs of two steps of a benchmark workload from the same state: same loss bits, same parameter bits (every
    reduction in the step has a fixed order; stochastic-depth draws follow torch.manual_seed)."""
    B, name, size = cfg[0], cfg[5], cfg[6]
    from vision_transformers_torch_xla_amd import optim_factory
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model

    g = torch.Generator(device=device).manual_seed(0)
    x = torch.randn(B, 3, size, size, generator=g, device=device)
    t = torch.randint(0, 1000, (B,), generator=g, device=device)

    def run():
        torch.manual_seed(0)
        m = create_model(name, pretrained=False, num_classes=1000, global_pool="avg", drop_path_rate=0.1).to(device)
        m.train()

        class A:
            opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
        opt = optim_factory.create_optimizer(A, m)
        crit = LabelSmoothingCrossEntropy(0.1)
        losses = []
        for _ in range(2):
            loss = crit(m(x), t)
            loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append(loss.detach().clone())
        flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
        del m, opt
        return torch.stack(losses), flat

    l1, p1 = run()
    l2, p2 = run()
    assert torch.isfinite(l1).all() and torch.isfinite(p1).all()
    assert 6.0 < l1[0].item() < 8.0                       # ln(1000) = 6.9 at initialisation
    assert torch.equal(l1, l2)
    assert torch.equal(p1, p2)


def test_kd_workload_full_size_is_reproducible(device):
    """Knowledge distillation at its benchmark size - DeiT-B/16 distilled student, ViT-B/16 teacher, 512 images, through
    StudentWithDistillation + DistillationLoss(LabelSmoothingCE(0.1), alpha 0.7, T 4) and train_one_epoch for two steps:
    the same bits twice, a loss of the size the formula gives at initialisation, a teacher that does not change."""
    from vision_transformers_torch_xla_amd import engine, optim_factory
    from vision_transformers_torch_xla_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy, StudentWithDistillation
    from vision_transformers_torch_xla_amd.vit import create_model

    Bk = 512
    g = torch.Generator(device=device).manual_seed(0)
    x = torch.randn(Bk, 3, 224, 224, generator=g, device=device)
    t = torch.randint(0, 1000, (Bk,), generator=g, device=device)

    def run():
        torch.manual_seed(0)
        student = create_model("deit_base_distilled_patch16_224", pretrained=False, num_classes=1000, drop_path_rate=0.1).to(device)
        teacher = create_model("vit_base_patch16_224", pretrained=False, num_classes=1000, global_pool="avg", drop_path_rate=0.0).to(device)
        teacher.eval()
        for p in teacher.parameters():
            p.requires_grad = False
        t0 = teacher._flat["p"].clone()

        class A:
            opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
        opt = optim_factory.create_optimizer(A, student)
        model = StudentWithDistillation(student, teacher)
        stats = engine.train_one_epoch(model, DistillationLoss(LabelSmoothingCrossEntropy(0.1), 0.7, 4.0), [(x, t), (x, t)], opt, device, 0,
                                       None, 0, None, None, num_training_steps_per_epoch=2, update_freq=1, log_freq=1)
        assert torch.equal(teacher._flat["p"], t0)
        flat = student._flat["p"].clone()
        del model, student, teacher, opt
        return stats, flat

    s1, p1 = run()
    s2, p2 = run()
    assert set(s1) == {"loss", "lr", "class_acc"} and torch.isfinite(p1).all()
    # at initialisation both models are near-uniform over 1000 classes: CE ~ ln(1000) = 6.9, KL ~ 0 -> loss ~ 0.3 * 6.9
    assert 1.8 < s1["loss"] < 2.6, s1
    assert s1 == s2 and torch.equal(p1, p2)
