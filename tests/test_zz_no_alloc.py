"""vitk_layernorm_bwd_noalloc (the entry point that allocates nothing: its dgamma / dbeta partial sums live in a scratch array
inside the library) against vitk_layernorm_bwd with a caller workspace, through the raw C ABI.  Same kernels, same grid, same
summation order: every output must be bit-identical.  Two calls in a row share the scratch, the second with a smaller input."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(M, D, seed, device, with_res):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, D, generator=g).mul(1.7).add(0.3).to(torch.bfloat16).to(device)
    dy = torch.randn(M, D, generator=g).to(torch.bfloat16).to(device)
    dres = torch.randn(M, D, generator=g).to(torch.bfloat16).to(device) if with_res else None
    gamma = (1.0 + 0.1 * torch.randn(D, generator=g)).to(device)
    xf = x.float()
    mean = xf.mean(1).contiguous()
    rstd = (xf.var(1, unbiased=False) + 1e-6).rsqrt().contiguous()
    return x, dy, dres, gamma, mean, rstd


def _run(lib, which, x, dy, dres, gamma, mean, rstd, accumulate, seed_out):
    from vision_transformers_torch_xla_amd._lib import check
    M, D = x.shape
    dev = x.device
    dx = torch.full((M, D), 7.0, dtype=torch.bfloat16, device=dev)
    g = torch.Generator().manual_seed(seed_out)
    dgamma = torch.randn(D, generator=g).to(dev)        # accumulate=1 adds to these, accumulate=0 must overwrite them
    dbeta = torch.randn(D, generator=g).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    ptr = lambda t: None if t is None else t.data_ptr()
    if which == "workspace":
        need = lib.vitk_layernorm_bwd_workspace(M, D)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        check(lib.vitk_layernorm_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx), ptr(dgamma), ptr(dbeta),
                                     int(accumulate), M, D, ptr(ws), ws.numel(), s), "layernorm_bwd")
    else:
        check(lib.vitk_layernorm_bwd_noalloc(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx), ptr(dgamma),
                                             ptr(dbeta), int(accumulate), M, D, s), "layernorm_bwd_noalloc")
    torch.cuda.synchronize()
    return dx, dgamma, dbeta


def test_noalloc_matches_workspace_entry_point_twice_in_a_row(device):
    from vision_transformers_torch_xla_amd import _lib
    lib = _lib.load()
    # first call: more rows than the grid has blocks (every scratch row in use, D = 1024 is the widest row);
    # second call: a smaller input that leaves most of the scratch holding the first call's partial sums
    cases = [(5000, 1024, True, 0), (333, 768, False, 1), (64, 192, True, 0)]
    want = [_run(lib, "workspace", *_inputs(M, D, 10 + i, device, res), acc, 50 + i) for i, (M, D, res, acc) in enumerate(cases)]
    ins = [_inputs(M, D, 10 + i, device, res) for i, (M, D, res, acc) in enumerate(cases)]
    _run(lib, "noalloc", *ins[2], cases[2][3], 52)                      # the first call on the device looks the scratch up
    got = [_run(lib, "noalloc", *ins[i], cases[i][3], 50 + i) for i in range(len(cases))]
    for (M, D, res, acc), w, g_ in zip(cases, want, got):
        for name, a, b in zip(("dx", "dgamma", "dbeta"), w, g_):
            assert torch.equal(a, b), f"{name} differs at M={M} D={D} res={res} accumulate={acc}"
    # and against the definition, so that two identically wrong paths cannot pass: fp64 LayerNorm backward of the first case
    x, dy, dres, gamma, mean, rstd = ins[0]
    xd = x.double().cpu().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xd, (1024,), gamma.double().cpu(), torch.zeros(1024, dtype=torch.float64), 1e-6)
    y.backward(dy.double().cpu())
    ref_dx = xd.grad + dres.double().cpu()
    err = ((got[0][0].double().cpu() - ref_dx).norm() / ref_dx.norm()).item()
    assert err <= 5e-3, err          # bf16 output of an fp32 kernel whose statistics came from torch: rounding-level agreement


def test_noalloc_rejects_rows_wider_than_its_scratch(device):
    from vision_transformers_torch_xla_amd import _lib
    lib = _lib.load()
    t = torch.zeros(8, 1032, dtype=torch.bfloat16, device=device)
    f = torch.zeros(1032, device=device)
    rc = lib.vitk_layernorm_bwd_noalloc(t.data_ptr(), t.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), None, t.data_ptr(),
                                        f.data_ptr(), f.data_ptr(), 0, 8, 1032, torch.cuda.current_stream().cuda_stream)
    assert rc == -22 and b"1024" in lib.vitk_last_error()
