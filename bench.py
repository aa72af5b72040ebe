#!/usr/bin/env python3
"""Headline benchmark: images/sec of the ViT-B/16 224px bf16 training step on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A step = one pass of the hot path over one synthetic batch: forward, label-smoothing CE, backward, gradient
all-reduce (N > 1), fused AdamW - driven by this package's engine.train_one_epoch, i.e. the same loop a user of the
reference's engine.train_one_epoch would run.  Per-GPU batch is fixed at 512 (global 4096 at 8 GPUs, the reference
recipe, run_train.sh:14-15), so scaling is weak.  Inputs are generated on the GPU before the timed region.
Prints ONE JSON line on stdout (rank 0); everything else goes to stderr.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# Eight hardware queues per process (HIP reads this when it starts; its default is 4): the launch stream, the weight-gradient side
# stream, the gradient reducer's stream and RCCL's own stream then each have a queue of their own.
try:
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
except ValueError:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

import torch
import torch.distributed as dist

PMC_HBM_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "pmc_hbm.json")
PEAK_BF16_TFLOPS = 2500.0                                    # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0                                        # HBM3E spec (6.3 TB/s measured achievable, same guide)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class SyntheticLoader:
    """K identical-shape batches that already live in HBM (generated once per rank with seed + rank).

    prof_every > 0: the launches of every prof_every-th step are bracketed with HIP events (the per-kernel roofline figures
    are averages over those steps of the timed region); bracketing every launch of every step costs 2-3 % of the step
    time, because an event between two kernels keeps the second from starting under the first one's tail."""

    def __init__(self, samples, targets, n, prof_every=0, prof_switch=None):
        self.samples, self.targets, self.n = samples, targets, n
        self.prof_every, self.prof_switch = prof_every, prof_switch

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            if self.prof_every:
                self.prof_switch(i % self.prof_every == 0)
            yield self.samples, self.targets
        if self.prof_every:
            self.prof_switch(False)


def train_macs_per_image(model):
    """forward multiply-accumulates per image of the GEMM / attention contractions: (patch embedding + head, attention branch of
    one block, MLP branch of one block).  Training work = 6 x (fwd 2, data gradient 2, weight gradient 2 flop per MAC)."""
    N, D, L = model.num_tokens, model.embed_dim, model.depth
    F = model.blocks[0].mlp.fc1.out_features
    heads = 2 if model.distilled else 1
    fixed = model.num_patches * (model.in_chans * model.patch_size ** 2) * D + heads * D * model.num_classes
    attn = N * D * 3 * D + N * D * D + 2 * N * N * D
    mlp = 2 * N * D * F
    return fixed, attn, mlp, L


def _cpu_steps(name, batch, steps, warm):
    from oracle import ref_engine, ref_ops
    from oracle.ref_vit import create_ref_model
    torch.manual_seed(0)
    model = create_ref_model(name, num_classes=1000, global_pool="avg", drop_path_rate=0.1)
    opt = ref_engine.make_optimizer(model, 4e-3, 0.05)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(batch, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (batch,), generator=g)
    crit = lambda o, t: ref_ops.label_smoothing_ce(o, t, 0.1)
    ref_engine.ref_train_one_epoch(model, crit, [(x, y)] * warm, opt)               # warm-up
    t0 = time.time()
    ref_engine.ref_train_one_epoch(model, crit, [(x, y)] * steps, opt)
    return batch * steps / (time.time() - t0)


def host_cpu():
    """(model string, physical cores of the machine, logical CPUs of the machine, CPUs this process may use)"""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid)); pid = cid = None
        if pid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    return model, len(phys) or None, os.cpu_count() or 1, usable


def cpu_baseline():
    """The oracle's reference-semantics fp32 train step (engine.py:257-274 restated) on this box's host cores: the headline model at
    the reference's batch of 64 on a bounded sample, and the reference's own CPU-runnable case (ViT-Tiny/16, 64 images per step)
    beside it.  Threads: every CPU this job may use - the scheduler's share of the machine (OMP_NUM_THREADS where the box sets
    it, else the affinity mask), which on a shared GPU host is fewer than the machine's physical cores; both counts and the CPU
    model are in the record."""
    model, phys, logical, usable = host_cpu()
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    threads = max(1, min(usable, share) if share > 0 else min(usable, phys or usable))
    torch.set_num_threads(threads)
    # ViT-B at 64 images is about 6 s per step on 16 threads: 3 warm-up + 10 timed steps would be 75 s of a run that has to finish
    # in minutes, so the sample is cut to 1 warm-up + 3 timed steps (192 images); ViT-Tiny runs the full 3 + 10
    out = {"value": _cpu_steps("vit_base_patch16_224", 64, 3, 1), "unit": "images/s", "cores": threads, "kind": "port",
           "cpu_model": model, "machine_physical_cores": phys, "machine_logical_cpus": logical, "cpus_usable_by_this_job": usable,
           "sample": "ViT-B/16 224px fp32 CPU oracle train step (fwd+bwd+AdamW), batch 64, 3 timed steps after 1 warm-up (cut from 10 after 3: 6 s per step)"}
    out["vit_tiny_batch64"] = {"value": _cpu_steps("vit_tiny_patch16_224", 64, 10, 3), "unit": "images/s",
                               "sample": "ViT-Tiny/16 224px fp32, batch 64, 10 timed steps after 3 warm-up (the reference's CPU-runnable configuration)"}
    return out


def pmc_traffic(args):
    if args.model == "vit_base_patch16_224" and args.batch == 512 and os.path.exists(PMC_HBM_FILE):
        try:
            rec = json.load(open(PMC_HBM_FILE))["gemm_nt"]
            return {"traffic": rec["bytes_per_launch"], "traffic_unit": f"bytes/launch (algorithmic: {rec['algorithmic_bytes_per_launch']:.3g}); "
                                                                        f"{rec['source']}"}
        except (KeyError, ValueError) as e:
            log(f"[bench] {PMC_HBM_FILE}: {e!r}")
    return {"traffic": None, "traffic_unit": "bytes/launch"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a child process (this
    process never touches the GPU), pass its stdout - rank 0's JSON line - through, and leave with its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log(f"[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd)}")
    raise SystemExit(subprocess.run(cmd).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch")
    ap.add_argument("--model", default="vit_base_patch16_224")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--prof-every", type=int, default=10, help="bracket the kernel launches of every n-th timed step with HIP events")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)

    # stdout carries the one JSON line and nothing else: everything a library prints there (RCCL writes a version banner to
    # stdout under NCCL_DEBUG=VERSION) goes to stderr instead, at the file-descriptor level
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if os.environ.get("VITK_BENCH_WATCHDOG"):        # debugging aid: dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VITK_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"[bench] WORLD_SIZE={world} does not match --gpus {args.gpus}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...), or run "
                         f"`python bench.py --gpus {args.gpus}` without a launcher and it starts the ranks itself")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP kernels are the only implementation of the hot path")
    # Rehearsal switches for a box with fewer GPUs than ranks (never set by the driver): VITK_SHARE_GPU=1 puts every rank
    # on cuda:0, VITK_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
    if os.environ.get("VITK_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("VITK_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    force_dist = os.environ.get("VITK_BENCH_FORCE_DIST") == "1"     # rehearsal: RCCL collectives in a one-rank group
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
    if world > 1 or force_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank, device_id=device)
        else:
            dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)
        if dist.get_world_size() != args.gpus and not force_dist:
            raise SystemExit(f"[bench] process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    n_ranks = dist.get_world_size() if dist.is_initialized() else 1

    from vision_transformers_torch_xla_amd import engine, ops, optim_factory, utils
    from vision_transformers_torch_xla_amd.dp import DistributedDataParallel
    from vision_transformers_torch_xla_amd.losses import LabelSmoothingCrossEntropy
    from vision_transformers_torch_xla_amd.vit import create_model

    torch.manual_seed(0)                                       # identical init on every rank (plus rank-0 broadcast)
    core = create_model(args.model, pretrained=False, num_classes=1000, global_pool="avg", drop_path_rate=0.1).to(device)
    torch.manual_seed(0 + rank)                                # main.py:405: every rank draws its own stochastic-depth masks
    bucket = int(os.environ.get("VITK_DP_BUCKET_MB", "32")) << 20
    model = (DistributedDataParallel(core, device_ids=[local], bucket_bytes=bucket, single_rank_collectives=force_dist)
             if (world > 1 or force_dist) else core)

    class OptArgs:
        opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 4e-3, 0.05, 1e-8, None
    optimizer = optim_factory.create_optimizer(OptArgs, core)
    criterion = LabelSmoothingCrossEntropy(0.1)
    total_steps = args.warmup + args.steps
    lr_sched = utils.cosine_scheduler(4e-3, 1e-6, 1, total_steps, warmup_steps=max(1, args.warmup), warmup_epochs=1)
    wd_sched = utils.cosine_scheduler(0.05, 0.05, 1, total_steps)

    g = torch.Generator(device=device).manual_seed(0 + rank)    # main.py:405: seed + rank
    samples = torch.randn(args.batch, 3, 224, 224, generator=g, device=device)
    targets = torch.randint(0, 1000, (args.batch,), generator=g, device=device)

    def run(nsteps, start, prof_every=0):
        with contextlib.redirect_stdout(sys.stderr):
            return engine.train_one_epoch(model, criterion, SyntheticLoader(samples, targets, nsteps, prof_every, ops.prof_enable), optimizer, device, 0, None, 0,
                                          None, None, start_steps=start, lr_schedule_values=lr_sched, wd_schedule_values=wd_sched,
                                          num_training_steps_per_epoch=nsteps, update_freq=1, log_freq=10)

    run(args.warmup, 0)
    core._branch_samples[:] = [0, 0, 0, 0]
    reducer = getattr(model, "reducer", None)
    if reducer is not None:
        reducer.timing = True                                   # timed events around every collective of the timed region
    want_roof = not args.no_roofline and args.prof_every > 0
    if want_roof:
        ops.prof_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = run(args.steps, args.warmup, args.prof_every if want_roof else 0)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if want_roof:
        ops.prof_enable(False)
    # every rank's own view of the timed region: its step time (its own clock, taken before the closing barrier) and the time its
    # gradient all-reduces spent between "inputs ready" and "done" on the reducer's stream (transfer + waiting for the slowest
    # peer); gathered on rank 0 so that one record shows whether a bad scaling number is one slow rank or the collectives
    per_rank = None
    if reducer is not None:
        comm_ms, comm_n, comm_elems = reducer.comm_ms()
        mine = torch.tensor([rank, own_elapsed / args.steps * 1e3, comm_ms / args.steps, comm_n / args.steps, comm_elems * 4.0 / args.steps],
                            dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        rows = [torch.zeros_like(mine) for _ in range(n_ranks)]
        dist.all_gather(rows, mine)
        per_rank = [{"rank": int(r[0]), "ms_per_step": round(float(r[1]), 3), "allreduce_ms_per_step": round(float(r[2]), 3),
                     "allreduce_share_of_step": round(float(r[2]) / max(float(r[1]), 1e-9), 4), "collectives_per_step": round(float(r[3]), 2),
                     "allreduce_bytes_per_step": int(r[4])} for r in rows]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * args.batch * args.steps / elapsed
        fixed, attn, mlp, L = train_macs_per_image(core)
        flop_img = 6.0 * (fixed + L * (attn + mlp))                     # the reference's work: every branch for every sample
        st = core._branch_samples                                      # stochastic depth: branches run for their kept samples only
        fa, fm = (st[0] / st[1], st[2] / st[3]) if st[1] and st[3] else (1.0, 1.0)
        flop_img_exec = 6.0 * (fixed + L * (attn * fa + mlp * fm))
        result = {
            "metric": "images/sec (whole node) ViT-B/16 224px bf16", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.model} train step (fwd + label-smoothing CE + bwd + grad all-reduce + AdamW), "
                                   f"224x224 synthetic randn batches resident in HBM, drop_path 0.1",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "loss_last": stats.get("loss")},
            # ranks of the process group the gradient all-reduce ran in (RCCL unless a rehearsal chose gloo); 0 = no group
            "ranks": n_ranks, "rccl_ranks": n_ranks if (dist.is_initialized() and backend == "nccl") else 0,
            # algorithmic = 6 x forward MACs for every sample and branch (what the reference computes); executed = without the
            # residual branches that stochastic depth dropped and this implementation therefore never ran
            "step_tflops_per_gpu": round(flop_img * args.batch / (ms * 1e-3) / 1e12, 1),
            "step_tflops_per_gpu_executed": round(flop_img_exec * args.batch / (ms * 1e-3) / 1e12, 1),
            "gflop_per_image": {"algorithmic": round(flop_img / 1e9, 2), "executed": round(flop_img_exec / 1e9, 2)},
        }
        if per_rank is not None:
            # the all-reduces run on a side stream under the backward kernels: their share is not a share of the critical path
            result["per_rank"] = per_rank
        if want_roof:
            nt_ms, nt_fl, nt_n = ops.prof_read(ops.PROF_GEMM_NT)
            tn_ms, tn_fl, tn_n = ops.prof_read(ops.PROF_GEMM_TN)
            ach = nt_fl / (nt_ms * 1e-3) / 1e12 if nt_ms > 0 else 0.0
            result["roofline"] = {
                "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                # HBM-side bytes per launch (average over this kernel family's launches of a step), from two separate
                # rocprofv3 --pmc passes over this same command (FETCH_SIZE doubled as gfx950 needs, + WRITE_SIZE), written
                # to tools/pmc_hbm.json by tools/pmc_hbm.sh + tools/summarize_pmc_hbm.py.  Counters cannot be read from
                # inside the process, so this is the committed measurement for the default workload, null for any other.
                **pmc_traffic(args),
                "kernel": "gemm_ntp_kernel / gemm_nt_kernel (bf16 MFMA 16x16x32; all epilogue instantiations: forward linears + data gradients)",
                "launches": nt_n, "avg_launch_ms": round(nt_ms / max(nt_n, 1), 4),
                "flop_per_launch_avg": round(nt_fl / max(nt_n, 1) / 1e9, 2),
                "also": {"kernel": "gemm_tnb_kernel / gemm_tn_kernel (weight gradients)", "achieved": round(tn_fl / (tn_ms * 1e-3) / 1e12, 1) if tn_ms > 0 else 0.0,
                         "launches": tn_n, "avg_launch_ms": round(tn_ms / max(tn_n, 1), 4)},
                "sampled_steps": len(range(0, args.steps, args.prof_every)),
                # a bracketed launch is timed with the chip to itself: in the sampled steps the weight-gradient GEMMs stay on the
                # launch stream; in all other steps they overlap the data-gradient chain from a side stream
                "sampling": "sampled steps run every kernel on one stream",
                "gemm_share_of_step": round((nt_ms + tn_ms) / (elapsed * 1e3 * len(range(0, args.steps, args.prof_every)) / args.steps), 3),
            }
            # the HBM-bound kernels of the step, same event brackets: algorithmic bytes / launch time against the 8 TB/s peak
            hbm = []
            for name, kind in (("ln_fwd_kernel (LayerNorm forward)", ops.PROF_LN_FWD), ("ln_bwd_kernel (LayerNorm backward + residual-gradient add + scaled copy)", ops.PROF_LN_BWD),
                               ("attn_fwd_kernel (fused softmax attention forward)", ops.PROF_ATTN_FWD),
                               ("attn_bwd_res_kernel (fused attention backward)", ops.PROF_ATTN_BWD), ("adamw_kernel (fused AdamW + bf16 copy-out)", ops.PROF_ADAMW)):
                k_ms, k_bytes, k_n = ops.prof_read(kind)
                if k_n:
                    gbs = k_bytes / (k_ms * 1e-3) / 1e9
                    hbm.append({"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": round(gbs / PEAK_HBM_GBS, 4), "launches": k_n, "avg_launch_us": round(k_ms / k_n * 1e3, 1),
                                "bytes_per_launch_avg": round(k_bytes / k_n)})
            result["hbm_rooflines"] = hbm
            ops.prof_reset()
    if world > 1:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline()
            except Exception as e:   # the baseline is a reported extra; never lose the GPU line over it
                log(f"[bench] cpu baseline failed: {e!r}")
                result["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
