"""Data-parallel path on the GPU with the real kernels: two ranks share cuda:0 (gloo, because RCCL refuses two ranks on
one device), each runs forward/backward of its half of a batch through libvitk with the GradReducer launching buckets
from inside backward on its side stream.

Checked: the reduced gradient on every rank is bit-identical to 0.5 * (g_half0 + g_half1) computed without any
communication (the kernels are deterministic and scaling by 0.5 is exact), buckets cover the flat buffer once in
increasing order, both ranks end with identical parameters after a fused AdamW step, and the returned meters are the
cross-rank averages."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vision_transformers_torch_xla_amd import engine, optim_factory
        from vision_transformers_torch_xla_amd.dp import DistributedDataParallel
        from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
        from vision_transformers_torch_xla_amd.vit import create_model

        torch.manual_seed(100 + rank)                       # different init per rank: the wrap-time broadcast must fix it
        model = create_model("vit_small_patch16_224", num_classes=40, global_pool="avg", depth=3, drop_path_rate=0.0).to(dev)
        ddp = DistributedDataParallel(model, device_ids=[0], bucket_bytes=1 << 20)
        w = model._flat["p"].detach().cpu()
        w0 = w.clone()
        dist.broadcast(w0, src=0)
        assert torch.equal(w, w0)

        g = torch.Generator().manual_seed(7)
        X = torch.randn(8, 3, 224, 224, generator=g).to(dev)
        Y = torch.randint(0, 40, (8,), generator=g).to(dev)
        crit = LabelSmoothingCrossEntropy(0.1)

        # the two halves without communication
        halves = []
        with ddp.no_sync():
            for r in range(2):
                model.zero_grad()
                crit(ddp(X[4 * r:4 * r + 4]), Y[4 * r:4 * r + 4]).backward()
                torch.cuda.synchronize()
                halves.append(model._flat["g"].clone())
        want = 0.5 * halves[0] + 0.5 * halves[1]

        # this rank's half with the reducer on
        model.zero_grad()
        crit(ddp(X[4 * rank:4 * rank + 4]), Y[4 * rank:4 * rank + 4]).backward()
        launched = list(ddp.reducer.launched)
        # no reducer.wait(): backward itself orders this stream after every bucket, so a plain reader of the gradients
        # (this clone, enqueued on the launch stream before any host synchronisation) sees the averaged values
        got = model._flat["g"].clone()
        torch.cuda.synchronize()
        assert launched[0][0] == 0 and launched[-1][1] == model._total and len(launched) >= 3, launched
        assert all(a[1] == b[0] for a, b in zip(launched, launched[1:]))
        assert torch.equal(got, want), float((got - want).abs().max())

        # two engine steps: identical parameters on both ranks afterwards, meters averaged over ranks
        class OptArgs:
            opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None
        opt = optim_factory.create_optimizer(OptArgs, model)
        loader = [(X[4 * rank:4 * rank + 4], Y[4 * rank:4 * rank + 4])] * 2
        stats = engine.train_one_epoch(ddp, crit, loader, opt, dev, 0, None, 0, None, None, update_freq=1, log_freq=1)
        torch.cuda.synchronize()
        p = model._flat["p"].detach().cpu()
        p0 = p.clone()
        dist.broadcast(p0, src=0)
        assert torch.equal(p, p0)
        assert not torch.equal(p, w)                        # the step did move the weights
        s = torch.tensor([stats["loss"]], dtype=torch.float64)
        s0 = s.clone()
        dist.broadcast(s0, src=0)
        assert torch.equal(s, s0)                           # the returned averages are global
        (Path(tmp) / f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_one_gpu(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
