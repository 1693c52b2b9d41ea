"""Time one training iteration (PPNTrainer.train_step: SURVEY 8d config 4, per-GPU shard of 32 frames) on cuda:0.

    python tools/bench_train.py [--batch 32] [--size 384] [--dtype bf16] [--steps 5] [--warmup 2] [--phases]

Prints ms/step, images/s and the achieved fraction of the bf16 MFMA roofline on the 3 x 95.3 GFLOP/img
(fwd + dgrad + wgrad) figure of SURVEY 8d; --phases times forward / loss / backward / GradNorm probes / optimiser
separately with HIP events.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_pose_proposal_network_amd import arch as A, lib as L, prng, synth, targets  # noqa: E402
from pytorch_pose_proposal_network_amd.trainer import PPNTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--phases", action="store_true")
    ap.add_argument("--first-order", action="store_true",
                    help="skip d Lgrad/d theta (main.py:759): model gradient = d loss/d theta only")
    args = ap.parse_args()
    # one process per GPU under torchrun (RANK / LOCAL_RANK / WORLD_SIZE): gradients and task weights are
    # exchanged over RCCL inside train_step; each rank works on its own 32-frame shard (weak scaling)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    sd = synth.make_state_dict("drn_d_22", 0)
    dt = L.PPN_BF16 if args.dtype == "bf16" else L.PPN_F32
    tr = PPNTrainer("drn_d_22", sd, compute_dtype=dt, insize=(args.size, args.size), second_order=not args.first_order)
    frames = prng.u8_frames(99 + rank, args.batch, (args.size, args.size))
    x = torch.from_numpy(synth.normalized_frames(frames)).to(dev)
    # 1..4 synthetic people per frame, targets built by the on-device encoder (csrc/encode.hip)
    tg = targets.synthetic_targets(99 + 1000 * rank, args.batch, (args.size, args.size), device=dev)
    for _ in range(args.warmup):
        tr.train_step(x, tg)
    torch.cuda.synchronize()
    if args.phases:
        # the phases below use the public building blocks one by one (not train_step's fused plumbing): run them once
        # untimed first, some of their kernels are otherwise loaded on first use inside the timed region
        for timed in (False, True):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            ev[0].record()
            head = tr.forward(x)
            ev[1].record()
            w = tr.task.w.tolist()
            losses, gh = tr.criterion.forward_backward(head, tg, coeff=[v / 5 for v in w])
            ev[2].record()
            tr.backward(gh)
            ev[3].record()
            tr.probe_norms(head, tg, [v / 5 for v in w], gh)
            ev[4].record()
            tr.opt.step(tr.grad)
            ev[5].record()
            torch.cuda.synchronize()
        names = ["forward", "loss fwd+bwd", "backward", "GradNorm probes", "adam"]
        for i, n in enumerate(names):
            print(f"{n:20s} {ev[i].elapsed_time(ev[i + 1]):9.3f} ms")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train_step(x, tg)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    if world > 1:
        t = torch.tensor([ms], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        if rank != 0:
            dist.destroy_process_group()
            return
    flops = 3 * A.conv_flops(A.build_program("drn_d_22"), args.size, args.size) * args.batch
    peak = 2.5e15 if args.dtype == "bf16" else 157.3e12
    print(json.dumps({"metric": "training images/sec (fwd+bwd+GradNorm+Adam)", "n_gpus": world,
                      "value": round(world * args.batch / ms * 1e3, 2),
                      "ms_per_step": round(ms, 3), "batch_per_gpu": args.batch, "dtype": args.dtype,
                      "roofline_frac_3x_fwd_flops": round(flops / (ms * 1e-3) / peak, 4)}))


    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
