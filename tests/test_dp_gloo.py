"""World-size-2 rehearsal of the data-parallel path on the CPU (gloo): the same GradReducer / DistributedDataParallel /
MetricLogger code that runs over RCCL on the GPUs, fed gradients produced by the oracle.

Checked: rank 0's parameters are broadcast at wrap time; buckets are launched in increasing address order as backward
reports progress and cover the flat gradient buffer exactly once; the reduced gradient equals the mean over ranks and
equals the single-process gradient of the concatenated batch (the DP-equivalence property); no_sync()/accumulation
skips communication; meters are summed across ranks like utils/__init__.py:122-142."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_grads_into(model, ref, x, y):
    """run the oracle on (x, y) and write its gradients into the HIP model's flat gradient buffer, reporting progress to
    the model's gradient-ready hook in the order the real backward does (head, blocks L-1..0, rest)"""
    from oracle import ref_ops as R
    ref.zero_grad()
    R.label_smoothing_ce(ref(x), y, 0.1).backward()
    refp = dict(ref.named_parameters())
    with torch.no_grad():
        for n in model._pnames:
            model._gview[n].copy_(refp[n].grad)
    hook = model._grad_ready_hook
    hook(model._head_end, False)
    for i in reversed(range(model.depth)):
        hook(model._layer_end[i], False)
    hook(model._total, True)


def _worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.ref_vit import create_ref_model
        from vision_transformers_torch_xla_amd import utils
        from vision_transformers_torch_xla_amd.dp import DistributedDataParallel
        from vision_transformers_torch_xla_amd.vit import create_model
        from tests.util import rel_l2

        torch.manual_seed(100 + rank)                       # different init per rank on purpose (main.py:405-406)
        model = create_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", depth=2)
        ddp = DistributedDataParallel(model, device_ids=None, bucket_bytes=1 << 20)
        # rank 0's weights everywhere
        w = model._flat["p"].clone()
        w0 = w.clone()
        dist.broadcast(w0, src=0)
        assert torch.equal(w, w0)

        ref = create_ref_model("vit_tiny_patch16_224", num_classes=40, global_pool="avg", depth=2)
        ref.load_state_dict(model.state_dict())
        g = torch.Generator().manual_seed(7)
        X = torch.randn(4, 3, 224, 224, generator=g)
        Y = torch.randint(0, 40, (4,), generator=g)
        xs, ys = X[2 * rank:2 * rank + 2], Y[2 * rank:2 * rank + 2]      # the global batch split across the 2 ranks

        red = ddp.reducer
        _oracle_grads_into(model, ref, xs, ys)
        launched = list(red.launched)
        red.wait()
        # buckets: increasing, contiguous, covering the whole flat buffer exactly once
        assert launched[0][0] == 0 and launched[-1][1] == model._total
        assert all(a[1] == b[0] for a, b in zip(launched, launched[1:])) and len(launched) >= 3
        mine = model._flat["g"].clone()

        # DP equivalence: mean of the per-rank gradients == gradient of the whole batch in one process
        from oracle import ref_ops as R
        ref.zero_grad()
        R.label_smoothing_ce(ref(X), Y, 0.1).backward()
        refp = dict(ref.named_parameters())
        whole = torch.zeros_like(mine)
        for n in model._pnames:
            o = model._offs[n]
            whole[o:o + refp[n].numel()] = refp[n].grad.reshape(-1)
        assert rel_l2(mine, whole) < 1e-5, rel_l2(mine, whole)
        other = mine.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(mine, other)                     # every rank holds the same reduced gradient

        # accumulation: no communication while the reducer is disabled
        with ddp.no_sync():
            _oracle_grads_into(model, ref, xs, ys)
            assert red.launched == []
        local = model._flat["g"].clone()
        assert rel_l2(local, whole) > 1e-3                  # still the local (un-averaged) gradient

        # meters: SUM of [count, total] over ranks
        ml = utils.MetricLogger()
        ml.meters["loss"].update(1.0 + rank, n=1)
        ml.meters["class_acc"].update(0.25 * (rank + 1), n=2 + rank)
        ml.synchronize_between_processes()
        assert ml.meters["loss"].count == 2 and abs(ml.meters["loss"].global_avg - 1.5) < 1e-12
        assert abs(ml.meters["class_acc"].global_avg - (0.25 * 2 + 0.5 * 3) / 5) < 1e-12
        sv = utils.SmoothedValue()
        sv.update(float(rank), n=1)
        sv.synchronize_between_processes()
        assert sv.count == 2 and sv.total == 1.0
        (Path(tmp) / f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def test_data_parallel_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
