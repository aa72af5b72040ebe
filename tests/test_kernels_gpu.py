"""Per-kernel parity: every libvitk entry point (through the C ABI) against the CPU oracle's op on the same seeded
inputs.  Floating point: relative L2 <= 1e-3 against the oracle evaluated with the same bf16 storage points
(and never more than a couple of bf16 ulps per element); index work (patch gather order, argmax): bit-exact."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import REL_TOL, bf16_excess, fp32_anchor, rb, rel_l2

pytestmark = pytest.mark.gpu

bf16, f32 = torch.bfloat16, torch.float32


def _ops():
    from vision_transformers_torch_xla_amd import ops
    return ops


def _randb(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(bf16)


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,D", [(788, 768), (37, 192), (100, 1024), (5, 384), (1, 768), (4099, 768)])
def test_layernorm_fwd_bwd(device, M, D):
    from oracle import ref_ops as R
    ops = _ops()
    g = torch.Generator().manual_seed(M * 131 + D)
    x = _randb((M, D), g, 2.0) + 0.5
    x = x.to(bf16)
    gamma = torch.randn(D, generator=g) * 0.5 + 1.0
    beta = torch.randn(D, generator=g) * 0.1
    dy = _randb((M, D), g)
    dres = _randb((M, D), g)

    xr = x.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    yr = R.layernorm(xr, gr, br, eps=1e-6)
    yr.backward(dy.float())

    y, mean, rstd = ops.layernorm_fwd(x.to(device), gamma.to(device), beta.to(device), 1e-6)
    assert rel_l2(y.float(), rb(yr)) <= REL_TOL
    assert bf16_excess(y, yr) <= 1.0
    fp32_anchor(f"kernel.layernorm.fwd[{M}x{D}]", rel_l2(y.float(), yr))
    assert rel_l2(mean, x.float().mean(-1)) <= 1e-5 or M == 1
    dgamma = torch.zeros(D, device=device)
    dbeta = torch.zeros(D, device=device)
    dx = ops.layernorm_bwd(dy.to(device), x.to(device), mean, rstd, gamma.to(device), dgamma, dbeta, dres=dres.to(device))
    ref_dx = xr.grad + dres.float()
    assert rel_l2(dx.float(), rb(ref_dx)) <= REL_TOL
    fp32_anchor(f"kernel.layernorm.bwd.dx[{M}x{D}]", rel_l2(dx.float(), ref_dx))
    assert rel_l2(dgamma, gr.grad) <= REL_TOL
    assert rel_l2(dbeta, br.grad) <= REL_TOL
    # accumulate=True adds to the existing gradient, no dres
    dx2 = ops.layernorm_bwd(dy.to(device), x.to(device), mean, rstd, gamma.to(device), dgamma, dbeta, accumulate=True)
    assert rel_l2(dx2.float(), rb(xr.grad)) <= REL_TOL
    assert rel_l2(dgamma, 2 * gr.grad) <= REL_TOL
    assert rel_l2(dbeta, 2 * br.grad) <= REL_TOL
    # fused stochastic-depth copy: bit-identical to scaling the stored bf16 dx in a separate pass
    rpg = 3 if M % 3 == 0 else 1
    sc = (torch.rand(M // rpg, generator=g) > 0.3).float() / 0.7
    dg2, db2 = torch.zeros(D, device=device), torch.zeros(D, device=device)
    dx3, dxs = ops.layernorm_bwd(dy.to(device), x.to(device), mean, rstd, gamma.to(device), dg2, db2, dres=dres.to(device),
                                 rowscale=sc.to(device), rows_per_group=rpg)
    assert torch.equal(dx3, dx)
    assert torch.equal(dxs, ops.rowscale(dx, sc.to(device), rpg))
    # two-call form: first stage only (partial sums in a buffer of the call's own), the reduction later - bit-identical, also
    # when it accumulates
    pending = []
    dg3, db3 = torch.full((D,), 7.0, device=device), torch.full((D,), -3.0, device=device)
    dx4 = ops.layernorm_bwd(dy.to(device), x.to(device), mean, rstd, gamma.to(device), dg3, db3, dres=dres.to(device), partial_out=pending)
    assert torch.equal(dx4, dx) and len(pending) == 1
    assert torch.all(dg3 == 7.0) and torch.all(db3 == -3.0)            # untouched until the second call
    ops.layernorm_bwd_reduce(pending[0])
    assert torch.equal(dg3, dg2) and torch.equal(db3, db2)
    pending = []
    ops.layernorm_bwd(dy.to(device), x.to(device), mean, rstd, gamma.to(device), dg3, db3, accumulate=True, partial_out=pending)
    ops.layernorm_bwd_reduce(pending[0])
    dg4, db4 = dg2.clone(), db2.clone()
    ops.layernorm_bwd(dy.to(device), x.to(device), mean, rstd, gamma.to(device), dg4, db4, accumulate=True)
    assert torch.equal(dg3, dg4) and torch.equal(db3, db4)
    assert torch.equal(dxs.cpu(), (dx.float().cpu() * sc.repeat_interleave(rpg)[:, None]).to(bf16))


# ----------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (591, 192, 192), (130, 1000, 768), (1000, 2304, 768), (77, 64, 1000), (128, 768, 3072),
                                   (3000, 200, 3072), (70000, 136, 128)])
def test_gemm_nt_plain_bias(device, M, N, K):
    from vision_transformers_torch_xla_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = _randb((M, K), g)
    b = _randb((N, K), g, 0.05)
    bias = torch.randn(N, generator=g)
    ref = a.double() @ b.double().t()
    c = ops.gemm_nt(a.to(device), b.to(device))
    assert rel_l2(c.float(), rb(ref.float())) <= REL_TOL
    assert bf16_excess(c, ref, atol=1e-4) <= 1.0
    c = ops.gemm_nt(a.to(device), b.to(device), epilogue=_lib.EPI_BIAS, bias=bias.to(device))
    assert rel_l2(c.float(), rb((ref + bias.double()).float())) <= REL_TOL
    c = ops.gemm_nt(a.to(device), b.to(device), epilogue=_lib.EPI_BIAS_F32, bias=bias.to(device))
    assert c.dtype == f32
    assert rel_l2(c, (ref + bias.double()).float()) <= 1e-5


def test_gemm_nt_asymmetric_identity(device):
    """A = I against an asymmetric B catches a transposed accumulator map."""
    ops = _ops()
    n = 256
    a = torch.eye(n).to(bf16)
    b = (torch.arange(n * n).reshape(n, n) % 251).float().to(bf16)       # b[i][j] != b[j][i]
    c = ops.gemm_nt(a.to(device), b.to(device))                          # C = I @ b^T
    assert torch.equal(c.cpu(), b.t().contiguous())


# the last three take the persistent large-tile kernels (M >= 512, K a multiple of 64): full tiles, ragged M and N
@pytest.mark.parametrize("M,N,K", [(394, 768, 768), (200, 192, 192), (1024, 768, 768), (1182, 392, 256), (2 * 591, 3072, 128)])
def test_gemm_nt_epilogues(device, M, N, K):
    from vision_transformers_torch_xla_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(7 * M + N)
    a = _randb((M, K), g)
    b = _randb((N, K), g, 0.05)
    bias = torch.randn(N, generator=g) * 0.1
    acc = (a.double() @ b.double().t() + bias.double()).float()
    # GELU: out = h (bf16), out2 = gelu(h_bf16)
    out2 = torch.empty((M, N), dtype=bf16, device=device)
    h = ops.gemm_nt(a.to(device), b.to(device), epilogue=_lib.EPI_BIAS_GELU, bias=bias.to(device), out2=out2)
    assert rel_l2(h.float(), rb(acc)) <= REL_TOL
    assert rel_l2(out2.float(), rb(F.gelu(h.float().cpu()))) <= REL_TOL
    # residual + per-sample scale: rows_per_group rows share one scale
    rpg = M // 2
    res = _randb((M, N), g)
    sc = torch.tensor([0.0, 1.0 / 0.9])
    assert rpg * 2 == M
    ref = res.float() + sc.repeat_interleave(rpg).unsqueeze(1) * acc
    c = ops.gemm_nt(a.to(device), b.to(device), epilogue=_lib.EPI_BIAS_RES, bias=bias.to(device), res=res.to(device),
                    rowscale=sc.to(device), rows_per_group=rpg)
    assert rel_l2(c.float(), rb(ref)) <= REL_TOL
    c = ops.gemm_nt(a.to(device), b.to(device), epilogue=_lib.EPI_BIAS_RES, bias=bias.to(device), res=res.to(device))
    assert rel_l2(c.float(), rb(res.float() + acc)) <= REL_TOL
    tag = f"[{M}x{N}x{K}]"
    fp32_anchor("kernel.gemm_nt.bias_res" + tag, rel_l2(c.float(), res.float() + acc))
    # dgelu: out = (a b^T) * gelu'(aux)
    aux = _randb((M, N), g)
    auxr = aux.float().requires_grad_(True)
    F.gelu(auxr).sum().backward()
    ref = (a.double() @ b.double().t()).float() * auxr.grad
    c = ops.gemm_nt(a.to(device), b.to(device), epilogue=_lib.EPI_DGELU, aux=aux.to(device))
    assert rel_l2(c.float(), rb(ref)) <= REL_TOL
    # GELU keeping the derivative: out = bf16(gelu'(h)), out2 = bf16(gelu(h)) with h = acc + bias in fp32 (h is never stored, so it
    # is not rounded); then out = (a b^T) * aux
    from oracle import ref_ops as R
    out2b = torch.empty((M, N), dtype=bf16, device=device)
    dg = ops.gemm_nt(a.to(device), b.to(device), epilogue=_lib.EPI_BIAS_GELU_DG, bias=bias.to(device), out2=out2b)
    assert rel_l2(out2b.float(), rb(F.gelu(acc))) <= REL_TOL
    assert rel_l2(dg.float(), rb(R.dgelu(acc))) <= REL_TOL
    fp32_anchor("kernel.gemm_nt.gelu_dg.act" + tag, rel_l2(out2b.float(), F.gelu(acc)))
    fp32_anchor("kernel.gemm_nt.gelu_dg.derivative" + tag, rel_l2(dg.float(), R.dgelu(acc)))
    c = ops.gemm_nt(a.to(device), b.to(device), epilogue=_lib.EPI_MUL, aux=dg)
    assert rel_l2(c.float(), rb((a.double() @ b.double().t()).float() * dg.float().cpu())) <= REL_TOL
    # the chain fc1 epilogue -> multiply epilogue against exact math: (a b^T) * gelu'(a b^T + bias), nothing rounded
    fp32_anchor("kernel.gemm_nt.mul_by_kept_derivative" + tag, rel_l2(c.float(), (a.double() @ b.double().t()).float() * R.dgelu(acc)))
    # in-place residual (out aliases res), as the model uses it
    resd = res.to(device).clone()
    ops.gemm_nt(a.to(device), b.to(device), epilogue=_lib.EPI_BIAS_RES, bias=bias.to(device), res=resd, out=resd)
    assert rel_l2(resd.float(), rb(res.float() + acc)) <= REL_TOL


@pytest.mark.parametrize("per,samples,kept", [(197, 12, 9), (128, 10, 7), (130, 9, 9), (300, 6, 4), (577, 4, 3), (50, 24, 17), (17, 64, 40)])
def test_gemm_nt_res_scatter(device, per, samples, kept):
    """bias + residual over the kept samples of a stochastic-depth branch: GEMM row c * per + j goes to row smap[c] * per + j of the
    residual stream, scaled by rowscale[c]; the other samples' rows stay as they are.  Samples of 128 rows and more take the
    per-tile lookup of the persistent kernel (sample boundaries inside a wave's 128 rows, the last sample, ragged last tile), the
    smaller ones the per-row path"""
    from vision_transformers_torch_xla_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(per * 31 + kept)
    N, K = 320, 128
    M = kept * per
    a = _randb((M, K), g)
    b = _randb((N, K), g, 0.05)
    bias = torch.randn(N, generator=g) * 0.1
    x = _randb((samples * per, N), g)
    smap = torch.randperm(samples, generator=g)[:kept].sort().values.to(torch.int32)
    rs = torch.rand(kept, generator=g) + 0.5
    acc = (a.double() @ b.double().t() + bias.double()).float()
    want = x.float().clone()
    for c in range(kept):
        rows = slice(int(smap[c]) * per, (int(smap[c]) + 1) * per)
        want[rows] += rs[c] * acc[c * per:(c + 1) * per]
    xd = x.to(device).clone()
    ops.gemm_nt(a.to(device), b.to(device), epilogue=_lib.EPI_BIAS_RES, bias=bias.to(device), res=xd, out=xd, rowscale=rs.to(device),
                rows_per_group=per, smap=smap.to(device), rows_per_sample=per)
    got = xd.float().cpu()
    dropped = torch.ones(samples, dtype=torch.bool)
    dropped[smap.long()] = False
    keep_rows = (~dropped).repeat_interleave(per)
    assert torch.equal(got[~keep_rows], x.float()[~keep_rows])            # dropped samples: bit for bit untouched
    assert rel_l2(got[keep_rows], rb(want)[keep_rows]) <= REL_TOL


def test_gemm_nt_concurrent_streams(device):
    """The persistent kernel deals its tiles through per-stream counters that it leaves zeroed: launches that overlap on two streams,
    and repeated launches on one, give bit for bit what a lone launch gives"""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    a1, b1 = _randb((4096, 768), g).to(device), _randb((768, 768), g, 0.05).to(device)
    a2, b2 = _randb((3000, 256), g).to(device), _randb((1024, 256), g, 0.05).to(device)
    want1, want2 = ops.gemm_nt(a1, b1), ops.gemm_nt(a2, b2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(6):
        with torch.cuda.stream(s1):
            o1 = ops.gemm_nt(a1, b1)
        with torch.cuda.stream(s2):
            o2 = ops.gemm_nt(a2, b2)
        outs.append((o1, o2))
    torch.cuda.synchronize()
    for o1, o2 in outs:
        assert torch.equal(o1, want1) and torch.equal(o2, want2)


def test_gemm_nt_dirty_tile_counters_are_reported(device):
    """The persistent kernel expects its stream's tile counters zeroed.  A counter word dirtied by hand makes the next launch skip
    tiles - and report it: the call after that refuses to launch and names the cause until the counters are reset."""
    from vision_transformers_torch_xla_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    a, b = _randb((256 * 300, 128), g).to(device), _randb((256, 128), g, 0.05).to(device)     # 300 tiles: every residue claims
    want = ops.gemm_nt(a, b)
    words, reporting, bad = ops.gemm_nt_sched_state()
    assert words == [0] * 16 and reporting in (1, 2) and not bad, (words, reporting, bad)     # a finished launch leaves them zeroed
    ops.gemm_nt_sched_poke(3, 5)                 # five claims that nobody made on the counter of blockIdx % 8 == 3
    assert ops.gemm_nt_sched_state()[0][3] == 5
    ops.gemm_nt(a, b)                            # launches (the host cannot know), skips tiles, reports
    words, reporting, bad = ops.gemm_nt_sched_state()
    assert bad, (words, reporting)
    with pytest.raises(_lib.VitkError, match="tile counters"):
        ops.gemm_nt(a, b)
    with pytest.raises(_lib.VitkError, match="tile counters"):          # sticky
        ops.gemm_nt(a, b)
    ops.gemm_nt_sched_reset()
    assert torch.equal(ops.gemm_nt(a, b), want)
    # an arrival ticket left behind (a launch that died after some workgroups had entered) is caught by the last one to enter
    ops.gemm_nt_sched_poke(0, 0x01000001)
    ops.gemm_nt(a, b)
    torch.cuda.synchronize()
    with pytest.raises(_lib.VitkError, match="tile counters"):
        ops.gemm_nt(a, b)
    ops.gemm_nt_sched_reset()
    assert torch.equal(ops.gemm_nt(a, b), want)
    # many streams: more than the library has counter sets; the least recently used set is recycled
    streams = [torch.cuda.Stream() for _ in range(8)]
    for s_ in streams:
        with torch.cuda.stream(s_):
            assert torch.equal(ops.gemm_nt(a, b), want)
    torch.cuda.synchronize()


def test_gemm_nt_capped_grid(tmp_path):
    """VITK_GEMM_CUS below 8 (or not a multiple of 8) would leave tiles unclaimed - claim c of residue x is tile G + 8 c + x - so
    the cap is rounded to a multiple of 8, at least 8.  Read once per process: run in a child."""
    import subprocess
    import sys
    code = (
        "import torch\n"
        "from vision_transformers_torch_xla_amd import ops\n"
        "g = torch.Generator().manual_seed(3)\n"
        "a = torch.randn(256 * 21 + 9, 192, generator=g).bfloat16().cuda(); b = (torch.randn(520, 192, generator=g) * 0.05).bfloat16().cuda()\n"
        "got = ops.gemm_nt(a, b).float().cpu()\n"
        "want = (a.float().cpu().double() @ b.float().cpu().double().t()).float().bfloat16().float()\n"
        "err = ((got - want).norm() / want.norm()).item()\n"
        "assert err < 1e-3, err\n"
        "print('ok', err)\n")
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for cap in ("4", "12", "20"):
        import os
        env = dict(os.environ, VITK_GEMM_CUS=cap, PYTHONPATH=str(root))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, (cap, r.stdout, r.stderr)


@pytest.mark.parametrize("M,N,K", [(256 * 40, 768, 768), (256 * 700 + 77, 768, 768), (50432, 1024, 1024), (197 * 300, 2304, 768),
                                   (197 * 290 + 5, 520, 3072), (256 * 9, 256, 2304)])
def test_gemm_nt_four_wave_kernel_is_bit_identical(device, monkeypatch, M, N, K):
    """gemm_ntq_kernel (four waves, output stores and side-operand loads deferred under the neighbouring tiles' MFMAs) against
    gemm_ntp_kernel (eight waves, the default; the four-wave kernel runs with VITK_GEMM_NTQ=1 only): the same fp32 accumulation order and the same epilogue arithmetic on fp32 values,
    so every epilogue it takes must agree bit for bit - full and ragged tiles, one tile per workgroup and many (the deferred path),
    stochastic-depth row maps with sample boundaries inside a wave's rows, in-place residual, two launches in a row."""
    from vision_transformers_torch_xla_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a, b = _randb((M, K), g).to(device), _randb((N, K), g, 0.05).to(device)
    bias = (torch.randn(N, generator=g) * 0.1).to(device)
    side = _randb((M, N), g).to(device)
    per = 197 if M % 197 == 0 else (256 if M % 256 == 0 else M)
    kept = M // per
    samples = kept + 3
    smap = torch.randperm(samples, generator=g)[:kept].sort().values.to(torch.int32).to(device)
    rs = (torch.rand(kept, generator=g) + 0.5).to(device)
    stream = _randb((samples * per, N), g).to(device)

    def run():
        outs = [ops.gemm_nt(a, b), ops.gemm_nt(a, b, epilogue=_lib.EPI_BIAS, bias=bias),
                ops.gemm_nt(a, b, epilogue=_lib.EPI_BIAS_RES, bias=bias, res=side),
                ops.gemm_nt(a, b, epilogue=_lib.EPI_MUL, aux=side)]
        if per >= 128:
            outs.append(ops.gemm_nt(a, b, epilogue=_lib.EPI_BIAS_RES, bias=bias, res=side, rowscale=rs, rows_per_group=per))
            x = stream.clone()
            ops.gemm_nt(a, b, epilogue=_lib.EPI_BIAS_RES, bias=bias, res=x, out=x, rowscale=rs, rows_per_group=per, smap=smap, rows_per_sample=per)
            outs.append(x)
        inplace = side.clone()
        ops.gemm_nt(a, b, epilogue=_lib.EPI_BIAS_RES, bias=bias, res=inplace, out=inplace)
        outs.append(inplace)
        outs.append(ops.gemm_nt(a, b))           # a second launch on counters the first one left
        torch.cuda.synchronize()
        return outs

    monkeypatch.setenv("VITK_GEMM_NTQ", "0")
    want = run()
    monkeypatch.setenv("VITK_GEMM_NTQ", "1")
    got = run()
    for i, (w, g_) in enumerate(zip(want, got)):
        assert torch.equal(w, g_), (i, (w.float() - g_.float()).abs().max().item())
    ref = rb((a.float().cpu().double() @ b.float().cpu().double().t()).float())
    assert rel_l2(got[0].float(), ref) <= REL_TOL


def test_gemm_nt_shape_sweep(device):
    """Persistent-kernel shapes chosen at random (seeded): ragged M and N, tile counts below, around and far above the
    compute-unit count (so some workgroups claim no second tile, some many), K from two steps up; plain and residual epilogues"""
    from vision_transformers_torch_xla_amd import _lib
    ops = _ops()
    rng = torch.Generator().manual_seed(2024)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))
    shapes = [(512, 128, 128), (513, 136, 192), (256 * 17 + 3, 264, 128), (256 * 40, 256 * 7, 128), (256 * 33 + 100, 1000, 320)]
    shapes += [(ri(512, 9000), 8 * ri(16, 160), 64 * ri(2, 9)) for _ in range(14)]
    for (M, N, K) in shapes:
        a, b = _randb((M, K), rng).to(device), _randb((N, K), rng, 0.05).to(device)
        want = rb(a.float().cpu().double() @ b.float().cpu().double().t())
        got = ops.gemm_nt(a, b)
        assert rel_l2(got.float(), want) <= REL_TOL, (M, N, K)
        assert bf16_excess(got.float().cpu(), (a.float().cpu().double() @ b.float().cpu().double().t()).float(), atol=1e-4) <= 1, (M, N, K)
        res = _randb((M, N), rng).to(device)
        got = ops.gemm_nt(a, b, epilogue=_lib.EPI_BIAS_RES, bias=torch.zeros(N, device=device), res=res)
        assert rel_l2(got.float(), rb(res.float().cpu().double() + a.float().cpu().double() @ b.float().cpu().double().t())) <= REL_TOL, (M, N, K)


def test_gemm_nt_patch_epilogue(device):
    from vision_transformers_torch_xla_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, P, pre, D, K = 3, 196, 2, 192, 768
    a = _randb((B * P, K), g)
    w = _randb((D, K), g, 0.05)
    bias = torch.randn(D, generator=g) * 0.1
    pos = torch.randn(P + pre, D, generator=g) * 0.02
    x = torch.zeros((B * (P + pre), D), dtype=bf16, device=device)
    ops.gemm_nt(a.to(device), w.to(device), epilogue=_lib.EPI_PATCH, bias=bias.to(device), out=x, pos=pos.to(device),
                patches=P, prefix=pre)
    ref = (a.double() @ w.double().t()).float().reshape(B, P, D) + bias + pos[pre:]
    got = x.reshape(B, P + pre, D)
    assert rel_l2(got[:, pre:].float(), rb(ref)) <= REL_TOL
    assert torch.count_nonzero(got[:, :pre]) == 0    # prefix rows untouched


# the last four take the large-tile path (M >= 8192, N, K >= 256): full tiles, ragged M (not a multiple of 64), ragged
# N / K (not multiples of 256), and a single row range per tile
@pytest.mark.parametrize("M,N,K,bias", [(1000, 192, 768, True), (4096, 768, 768, True), (130, 1000, 768, False),
                                        (197 * 8, 2304, 768, True), (64, 8, 8, True), (513, 3072, 768, False),
                                        (8192, 768, 768, True), (197 * 48 + 5, 1000, 328, True), (197 * 64, 2304, 768, False),
                                        (8200, 4096, 3072, True)])
def test_gemm_tn(device, M, N, K, bias):
    ops = _ops()
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = _randb((M, N), g)
    b = _randb((M, K), g)
    ref = (a.double().t() @ b.double()).float()
    refb = a.double().sum(0).float()
    out = torch.zeros((N, K), device=device)
    db = torch.zeros(N, device=device) if bias else None
    ops.gemm_tn(a.to(device), b.to(device), out, dbias=db)
    assert rel_l2(out, ref) <= 1e-5
    if bias:
        assert rel_l2(db, refb) <= 1e-5
    ops.gemm_tn(a.to(device), b.to(device), out, dbias=db, accumulate=True)
    assert rel_l2(out, 2 * ref) <= 1e-5
    if bias:
        assert rel_l2(db, 2 * refb) <= 1e-5


def test_gemm_tn_deterministic(device):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    a = _randb((8192, 768), g).to(device)
    b = _randb((8192, 768), g).to(device)
    o1 = torch.empty((768, 768), device=device)
    o2 = torch.empty((768, 768), device=device)
    ops.gemm_tn(a, b, o1)
    ops.gemm_tn(a, b, o2)
    assert torch.equal(o1, o2)


def test_gemm_rejects_bad_shapes(device):
    from vision_transformers_torch_xla_amd import _lib
    ops = _ops()
    a = torch.zeros((16, 12), dtype=bf16, device=device)     # K % 8 != 0
    b = torch.zeros((16, 12), dtype=bf16, device=device)
    with pytest.raises(_lib.VitkError):
        ops.gemm_nt(a, b)
    with pytest.raises(_lib.VitkError):
        ops.gemm_nt(a.cpu(), b.cpu())


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,N,H", [(2, 197, 3), (1, 198, 12), (1, 577, 2), (3, 17, 1), (2, 50, 2), (1, 224, 1), (1, 225, 1), (2, 1, 1), (1, 608, 1), (1, 300, 2), (2, 400, 1), (1, 512, 1), (1, 257, 1),
                                   (1, 193, 2), (2, 208, 1), (1, 209, 1), (1, 192, 1)])   # either side of the 13-tile kernels (193 ... 208 tokens)
def test_attention_fwd_bwd(device, B, N, H):
    from oracle import ref_ops as R
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + N * 7 + H)
    D = H * 64
    qkv = _randb((B, N, 3 * D), g, 1.5)
    dout = _randb((B, N, D), g)
    scale = 64 ** -0.5
    res = {}
    for emu in (True, False):
        q = qkv.float().requires_grad_(True)
        o = R.attention(q, H, scale, emulate=emu)
        o.backward(dout.float())
        res[emu] = (o.detach(), q.grad)
    out, lse = ops.attention_fwd(qkv.reshape(B * N, 3 * D).to(device), B, N, H, scale)
    assert rel_l2(out.float().reshape(B, N, D), res[True][0]) <= REL_TOL
    assert rel_l2(out.float().reshape(B, N, D), res[False][0]) <= 6e-3      # vs exact math: bf16 noise floor
    fp32_anchor(f"kernel.attention.fwd[{B}x{N}x{H}]", rel_l2(out.float().reshape(B, N, D), res[False][0]))
    # lse against the definition
    qh, kh, _ = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref_lse = torch.logsumexp(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    assert rel_l2(lse, ref_lse) <= 1e-5
    dqkv = ops.attention_bwd(qkv.reshape(B * N, 3 * D).to(device), out, dout.reshape(B * N, D).to(device), lse, B, N, H, scale)
    got = dqkv.float().reshape(B, N, 3 * D)
    # identical inputs: the oracle's backward is fed the O and log-sum-exp the kernel itself consumed
    want = R.attention_backward(qkv.float(), out.float().cpu().reshape(B, N, D), lse.cpu(), dout.float(), H, scale, emulate=True)
    errs = {}
    for which, name in enumerate("qkv"):
        sl = slice(which * D, (which + 1) * D)
        if N == 1 and name in "qk":        # one key: P = 1 and dS = dP - delta = 0 exactly; both sides hold rounding residue only
            assert got[..., sl].abs().max().item() <= 1e-5 and want[..., sl].abs().max().item() <= 1e-5, name
            errs[name] = 0.0
            continue
        errs[name] = rel_l2(got[..., sl], rb(want[..., sl]))
        assert rel_l2(got[..., sl], res[False][1][..., sl]) <= 1e-2, name
        fp32_anchor(f"kernel.attention.bwd.d{name}[{B}x{N}x{H}]", rel_l2(got[..., sl], res[False][1][..., sl]))
    print(f"attention backward B={B} N={N} H={H}: " + ", ".join(f"d{k} {v:.1e}" for k, v in errs.items()))
    assert max(errs.values()) <= REL_TOL, errs


def test_label_out_of_range_gives_nan_not_a_wild_read(device):
    """a label outside [0, C) - an ignore_index of another code base, a head of another size - must not become an address: that
    row's loss and gradient are NaN (the step fails loudly), every other row is untouched"""
    from vision_transformers_torch_xla_amd.losses import CrossEntropyLoss, LabelSmoothingCrossEntropy
    g = torch.Generator().manual_seed(3)
    z = torch.randn(6, 40, generator=g)
    good = torch.randint(0, 40, (6,), generator=g)
    for bad_label in (-100, 40, 2 ** 40):
        t = good.clone()
        t[2] = bad_label
        zg = z.clone().to(device).requires_grad_(True)
        loss = LabelSmoothingCrossEntropy(0.1)(zg, t.to(device))
        loss.backward()
        assert torch.isnan(loss).item()
        assert torch.isnan(zg.grad[2]).all() and torch.isfinite(zg.grad[[0, 1, 3, 4, 5]]).all()
        zr = z.clone().requires_grad_(True)
        keep = [0, 1, 3, 4, 5]
        (F.cross_entropy(zr[keep], good[keep], label_smoothing=0.1, reduction="sum") / 6).backward()
        assert rel_l2(zg.grad[keep], zr.grad[keep]) <= 1e-5
    assert torch.isfinite(CrossEntropyLoss()(z.to(device), good.to(device))).item()


@pytest.mark.parametrize("B,N,H,hd", [(2, 197, 3, 48), (2, 197, 4, 72), (1, 256, 1, 96), (3, 50, 2, 8), (1, 1, 1, 40), (1, 198, 2, 80),
                                      (2, 197, 4, 96), (3, 224, 2, 96), (2, 33, 3, 96), (1, 1, 1, 96), (2, 130, 5, 96)])   # 96 <= 224 tokens: MFMA
def test_attention_other_head_dims(device, B, N, H, hd):
    """head dims other than 64 (48 and 72 are my_vit_mini / my_vit_xs): the vector-arithmetic kernels against the oracle with the
    same bf16 storage points, forward and backward (backward fed the kernel's own O and log-sum-exp), and against exact math"""
    from oracle import ref_ops as R
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + N * 7 + H + hd)
    D = H * hd
    qkv = _randb((B, N, 3 * D), g, 1.5)
    dout = _randb((B, N, D), g)
    scale = hd ** -0.5
    res = {}
    for emu in (True, False):
        q = qkv.float().requires_grad_(True)
        o = R.attention(q, H, scale, emulate=emu)
        o.backward(dout.float())
        res[emu] = (o.detach(), q.grad)
    out, lse = ops.attention_fwd(qkv.reshape(B * N, 3 * D).to(device), B, N, H, scale)
    assert out.shape == (B * N, D)
    assert rel_l2(out.float().reshape(B, N, D), res[True][0]) <= REL_TOL
    assert rel_l2(out.float().reshape(B, N, D), res[False][0]) <= 6e-3
    qh, kh, _ = qkv.float().reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    assert rel_l2(lse, torch.logsumexp(qh @ kh.transpose(-1, -2) * scale, dim=-1)) <= 1e-5
    dqkv = ops.attention_bwd(qkv.reshape(B * N, 3 * D).to(device), out, dout.reshape(B * N, D).to(device), lse, B, N, H, scale)
    got = dqkv.float().reshape(B, N, 3 * D)
    want = R.attention_backward(qkv.float(), out.float().cpu().reshape(B, N, D), lse.cpu(), dout.float(), H, scale, emulate=True)
    for which, name in enumerate("qkv"):
        sl = slice(which * D, (which + 1) * D)
        assert rel_l2(got[..., sl], rb(want[..., sl])) <= REL_TOL, name
        assert rel_l2(got[..., sl], res[False][1][..., sl]) <= 1e-2, name


def test_attention_matches_sdpa(device):
    """independent check of the kernel against torch's own scaled_dot_product_attention on CPU"""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    B, N, H = 2, 197, 2
    D = H * 64
    qkv = _randb((B, N, 3 * D), g)
    q, k, v = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, D)
    out, _ = ops.attention_fwd(qkv.reshape(B * N, 3 * D).to(device), B, N, H, 0.125)
    assert rel_l2(out.float().reshape(B, N, D), ref) <= 6e-3


# ------------------------------------------------------------------------------------ embedding / pooling
def test_im2col_bit_exact(device):
    from oracle import ref_ops as R
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    img = torch.randn((3, 3, 64, 48), generator=g)
    got = ops.im2col_patches(img.to(device), 16)
    ref = R.im2col_patches(img, 16).reshape(-1, 768).to(bf16)
    assert torch.equal(got.cpu().view(torch.int16), ref.view(torch.int16))
    # index identity: an image holding its own flat index must land at k = c*256 + ky*16 + kx of patch gy*gw+gx
    idx = torch.arange(1 * 3 * 32 * 32, dtype=torch.float32).reshape(1, 3, 32, 32) % 256
    got = ops.im2col_patches(idx.to(device), 16).float().cpu().reshape(2, 2, 3, 16, 16)
    for gy in range(2):
        for gx in range(2):
            assert torch.equal(got[gy, gx], idx[0, :, gy * 16:(gy + 1) * 16, gx * 16:(gx + 1) * 16])


def test_prefix_pool_gather_scatter(device):
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    B, N, D, pre = 4, 50, 192, 2
    tokens = torch.randn(pre, D, generator=g)
    pos = torch.randn(N, D, generator=g)
    x = _randb((B * N, D), g).to(device)
    x0 = x.clone()
    ops.prefix_tokens_fwd(tokens.to(device), pos.to(device), x, B, N, D, pre)
    xr = x.reshape(B, N, D)
    assert torch.equal(xr[:, :pre].cpu(), (tokens + pos[:pre]).to(bf16).unsqueeze(0).expand(B, -1, -1))
    assert torch.equal(xr[:, pre:], x0.reshape(B, N, D)[:, pre:])
    # pos / token gradient
    dx = _randb((B, N, D), g)
    dpos = torch.zeros((N, D), device=device)
    dtok = torch.zeros((pre, D), device=device)
    ops.pos_tokens_bwd(dx.reshape(B * N, D).to(device), dpos, dtok, B, N, D, pre)
    assert rel_l2(dpos, dx.float().sum(0)) <= 1e-5
    assert rel_l2(dtok, dx.float().sum(0)[:pre]) <= 1e-5
    # average pool over tokens >= first
    y = ops.pool_avg_fwd(dx.reshape(B * N, D).to(device), B, N, D, 1)
    assert rel_l2(y.float(), rb(dx.float()[:, 1:].mean(1))) <= REL_TOL
    dy = _randb((B, D), g)
    dxx = ops.pool_avg_bwd(dy.to(device), B, N, D, 1).reshape(B, N, D)
    assert torch.count_nonzero(dxx[:, 0]) == 0
    assert rel_l2(dxx[:, 1:].float(), rb(dy.float() / (N - 1)).unsqueeze(1).expand(-1, N - 1, -1)) <= REL_TOL
    # gather / scatter of one token row
    gth = ops.gather_rows(dx.reshape(B * N, D).to(device), B, N, D, 1)
    assert torch.equal(gth.cpu(), dx[:, 1])
    buf = torch.full((B * N, D), 7.0, dtype=bf16, device=device)
    ops.scatter_rows(dy.to(device), buf, B, N, D, 0, True)
    bb = buf.reshape(B, N, D)
    assert torch.equal(bb[:, 0].cpu(), dy) and torch.count_nonzero(bb[:, 1:]) == 0
    ops.scatter_rows(dy.to(device), buf, B, N, D, 1, False)
    assert torch.equal(bb[:, 1].cpu(), dy) and torch.equal(bb[:, 0].cpu(), dy)


def test_elementwise(device):
    ops = _ops()
    g = torch.Generator().manual_seed(22)
    x = _randb((6 * 10, 192), g)
    sc = torch.tensor([0.0, 1.25, 1.0 / 0.9, 0.0, 1.0, 1.0 / 0.7])
    y = ops.rowscale(x.to(device), sc.to(device), 10)
    assert torch.equal(y.cpu(), (x.float() * sc.repeat_interleave(10).unsqueeze(1)).to(bf16))
    f = torch.randn(1003, generator=g)
    assert torch.equal(ops.cast_f32_to_bf16(f.to(device)).cpu(), f.to(bf16))
    assert torch.equal(ops.cast_bf16_to_f32(x.to(device)).cpu(), x.float())
    t = _randb((1000, 768), g)
    assert torch.equal(ops.transpose_bf16(t.to(device)).cpu(), t.t().contiguous())
    t = _randb((70, 130), g)
    assert torch.equal(ops.transpose_bf16(t.to(device)).cpu(), t.t().contiguous())
    # all matrices of a list in one launch (ragged shapes, bit-exact)
    mats = [_randb(sh, g).to(device) for sh in ((768, 2304), (70, 130), (1000, 768), (64, 64), (1, 8))]
    outs = [torch.empty((m.shape[1], m.shape[0]), dtype=bf16, device=device) for m in mats]
    ops.transpose_bf16_batched(ops.transpose_plan(list(zip(mats, outs))))
    for m, o in zip(mats, outs):
        assert torch.equal(o, m.t().contiguous())
    a, b = _randb((1003,), g), _randb((1003,), g)
    assert torch.equal(ops.add_bf16(a.to(device), b.to(device)).cpu(), (a.float() + b.float()).to(bf16))


# ------------------------------------------------------------------------------------------------- losses
@pytest.mark.parametrize("B,C", [(64, 1000), (7, 10), (3, 1000)])
def test_losses(device, B, C):
    from oracle import ref_ops as R
    ops = _ops()
    g = torch.Generator().manual_seed(B + C)
    logits = torch.randn((B, C), generator=g) * 3
    target = torch.randint(0, C, (B,), generator=g)
    for eps in (0.0, 0.1):
        lr_ = logits.clone().requires_grad_(True)
        ref = R.label_smoothing_ce(lr_, target, eps)
        ref.backward()
        rows, dl = ops.xent_label(logits.to(device), target.to(device), eps, 1.0 / B)
        assert abs(rows.mean().item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
        assert rel_l2(dl, lr_.grad) <= 1e-4
    if C > 10:
        ce = F.cross_entropy(logits, target)
        rows, _ = ops.xent_label(logits.to(device), target.to(device), 0.0, 1.0, want_grad=False)
        assert abs(rows.mean().item() - ce.item()) <= 1e-5 * abs(ce.item())
    soft = torch.softmax(torch.randn((B, C), generator=g), -1)
    lr_ = logits.clone().requires_grad_(True)
    ref = R.soft_target_ce(lr_, soft)
    ref.backward()
    rows, dl = ops.xent_soft(logits.to(device), soft.to(device), 1.0 / B)
    assert abs(rows.mean().item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert rel_l2(dl, lr_.grad) <= 1e-4
    teacher = torch.randn((B, C), generator=g) * 2
    lr_ = logits.clone().requires_grad_(True)
    T = 4.0
    ref = F.kl_div(torch.log_softmax(lr_ / T, 1), torch.softmax(teacher / T, 1), reduction="batchmean") * T * T
    ref.backward()
    rows, ds = ops.kd_kl(logits.to(device), teacher.to(device), T, 1.0 / B)
    assert abs(rows.sum().item() / B - ref.item()) <= 1e-4 * max(abs(ref.item()), 1e-3)
    assert rel_l2(ds, lr_.grad) <= 1e-4


def test_argmax_bit_exact(device):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    logits = torch.randn((300, 1000), generator=g)
    logits[5, 17] = logits[5, 900] = 50.0          # tie: first index wins
    logits[6, :] = 1.0
    got = ops.argmax_rows(logits.to(device)).cpu()
    assert torch.equal(got, logits.max(-1)[-1]) or torch.equal(got, logits.argmax(-1))
    assert got[5] == 17 and got[6] == 0
    # strided view (padded rows)
    wide = torch.randn((9, 1024), generator=g)
    got = ops.argmax_rows(wide.to(device)[:, :1000]).cpu()
    assert torch.equal(got, wide[:, :1000].argmax(-1))


# ----------------------------------------------------------------------------------------------- optimizer
def test_adamw_matches_torch(device):
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    n = 100003
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    p = p0.clone().to(device)
    m = torch.zeros(n, device=device)
    v = torch.zeros(n, device=device)
    pb = torch.empty(n, dtype=bf16, device=device)
    ema = p0.clone().to(device)
    ema_ref = p0.clone()
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (0.1 * step)
        lr = 1e-3 * step
        opt.param_groups[0]["lr"] = lr
        ref_p.grad = grad.clone()
        opt.step()
        ops.adamw_step(p, (grad * 4).to(device), m, v, lr=lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.05, step=step,
                       grad_scale=0.25, p_bf16=pb, ema=ema, ema_decay=0.99)
        ema_ref = ema_ref * 0.99 + 0.01 * ref_p.detach()
    assert rel_l2(p, ref_p.detach()) <= 1e-6
    assert rel_l2(m, opt.state[ref_p]["exp_avg"]) <= 1e-5
    assert rel_l2(v, opt.state[ref_p]["exp_avg_sq"]) <= 1e-5
    assert torch.equal(pb.cpu(), p.cpu().to(bf16))
    assert rel_l2(ema, ema_ref) <= 1e-6


def test_sumsq(device):
    ops = _ops()
    g = torch.Generator().manual_seed(10)
    x = torch.randn(1234567, generator=g)
    out = torch.zeros(1, device=device)
    ops.sumsq(x.to(device), out)
    assert abs(out.item() - x.double().pow(2).sum().item()) <= 1e-5 * x.double().pow(2).sum().item()


def test_copy_from_pinned_host_buffer(device):
    """the copy kernel reads page-locked host memory directly (the per-step stochastic-depth tables travel this way)"""
    ops = _ops()
    for n, dt in ((4, torch.int32), (2 * 24 * 512, torch.int32), (2 * 24 * 512, torch.float32), (1 << 20, torch.float32)):
        src = torch.empty(n, dtype=dt, pin_memory=True)
        src.copy_(torch.arange(n).to(dt))
        got = ops.copy_from_pinned(src, device)
        assert got.dtype == dt and torch.equal(got.cpu(), src)
    with pytest.raises(Exception):
        ops.copy_from_pinned(torch.zeros(8, dtype=torch.int32), device)          # pageable memory is refused
    with pytest.raises(Exception):
        ops.copy_from_pinned(torch.zeros(3, dtype=torch.int32).pin_memory(), device)   # not a multiple of 16 bytes
