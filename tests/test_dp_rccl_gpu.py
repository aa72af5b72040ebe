"""The RCCL branch of the data-parallel path on one GPU: a one-rank `nccl` process group (RCCL on ROCm) whose collectives are
really issued (`single_rank_collectives=True`) - process-group start with a bound device, the wrap-time parameter broadcast of
a GPU tensor, `ReduceOp.AVG` all-reduces of every gradient bucket on the reducer's side stream from inside backward, the
optimizer step behind the bucket events, and a clean teardown.  This is the branch that runs on 8 GPUs
(main.py:855-858 DistributedDataParallel / engine.py:185 xm.optimizer_step in the reference); with one rank the average is
the identity, so two steps of engine.train_one_epoch must leave gradients and parameters bit-identical to the same two
steps without data parallelism.

Runs in a child process: a process group belongs to a process, and the test suite's own process stays without one."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, port, tmp):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from vision_transformers_torch_xla_amd import engine, optim_factory
    from vision_transformers_torch_xla_amd.dp import DistributedDataParallel
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model

    class OptArgs:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1e-3, 0.05, 1e-8, None

    g = torch.Generator().manual_seed(7)
    data = [(torch.randn(8, 3, 224, 224, generator=g), torch.randint(0, 40, (8,), generator=g)) for _ in range(2)]
    crit = LabelSmoothingCrossEntropy(0.1)

    def run(wrap):
        torch.manual_seed(5)                                # weights, then the host-side stochastic-depth draws
        model = create_model("vit_small_patch16_224", num_classes=40, global_pool="avg", depth=4, drop_path_rate=0.1).to(dev)
        net = DistributedDataParallel(model, device_ids=[0], bucket_bytes=4 << 20, single_rank_collectives=True) if wrap else model
        opt = optim_factory.create_optimizer(OptArgs, model)
        torch.manual_seed(17)
        stats = engine.train_one_epoch(net, crit, data, opt, dev, 0, None, 0, None, None, update_freq=1, log_freq=1)
        torch.cuda.synchronize()
        return model, net, stats

    plain, _, stats_plain = run(False)

    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        model, ddp, stats = run(True)
        red = ddp.reducer
        assert not red._skip                                              # the collectives were not short-circuited
        launched = list(red.launched)
        assert len(launched) >= 3, launched                               # several buckets, not one all-reduce at the end
        assert launched[0][0] == 0 and launched[-1][1] == model._total    # they cover the flat gradient buffer ...
        assert all(a[1] == b[0] for a, b in zip(launched, launched[1:]))  # ... once, in increasing address order
        assert red._stream is not None and red._stream != torch.cuda.current_stream(dev)     # on the reducer's own stream
        assert not red._events and not red._works                         # backward joined every bucket
        assert torch.equal(model._flat["g"], plain._flat["g"])            # AVG over one rank is the identity: same bits
        assert torch.equal(model._flat["p"], plain._flat["p"])            # and so are the weights after two AdamW steps
        assert torch.equal(model._flat["b"], plain._flat["b"])
        assert stats == stats_plain, (stats, stats_plain)
        # a reduction by hand after backward is a no-op
        red.wait()
        dist.barrier()
    finally:
        dist.destroy_process_group()
    assert not dist.is_initialized()
    (Path(tmp) / "ok").write_text("ok")


def test_rccl_one_rank_group_runs_every_bucket(tmp_path):
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok").exists()
