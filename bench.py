#!/usr/bin/env python
"""Benchmark of the PPN hot path on MI355X: images/sec, DRN-D-22 @ 384x384, bf16 MFMA, batch 32 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path over one batch that is already resident in HBM:
  u8 frames [32,384,384,3] -> fused normalisation + 35 fused conv launches (head f32 [32,7605,24,24])
  -> limb arg-max + root NMS + limb parse (compact people list left on the device).
Frames are independent units, so N GPUs = N ranks each with its own 32 frames (weak scaling), no collective
on the data path; torch.distributed (RCCL) is used only for the timing barrier and the MAX over ranks.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel (by time): algorithmic FLOPs of its launches / their HIP-event duration, every launch
                timed once in plan order (agrees with rocprofv3 `--lanes 1`); `back_to_back` = four repeats per launch
  cpu_baseline  the CPU oracle (torch-CPU fp32 forward + NumPy decode) on a bounded sample, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pytorch_pose_proposal_network_amd import config as cfg, prng, synth  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3


def load_bn_stats(arch, seed=0):
    path = os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", f"bn_calib_{arch}_seed{seed}.npz")
    g = np.load(path)
    return {k: g[k] for k in g.files}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (tools/pmc_bench.sh ->
    profiles/*_pmc_traffic.json: FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, separate passes).
    bench.py cannot run the profiler itself; returns None when no measurement of this kernel is committed."""
    import glob
    import re
    m = re.match(r"(\w+)<(.*)>", kernel_name)
    if not m:
        return None
    parts = [a.strip() for a in m.group(2).split(",")]
    frag = m.group(1) + "I" + "".join(("DF16b" if a == "__bf16" else "f" if a == "float" else
                                       ("Lb1E" if a == "true" else "Lb0E" if a == "false" else f"Li{a}E"))
                                      for a in parts)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        table = json.load(open(path))
        for k, v in table.items():
            if frag in k:
                return {"bytes_per_launch": v["hbm_bytes_per_launch"], "source": os.path.basename(path),
                        "date": table.get("_date") or time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(path)))}
    return None


def mfma_busy(kernel_name, achieved_tflops, peak_tflops):
    """MFMA utilisation BY COUNTER (north_star: "rocprof reports ... MFMA utilisation for the conv stack") and the roofline
    fraction against the peak AT THE CLOCK THE CHIP HOLDS under this kernel.  Committed constants with provenance, like
    `traffic`: bench.py cannot run the profiler on itself.  SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) per
    dispatch, one SQ + GRBM pass of tools/pmc_mfma.sh -> tools/pmc_mfma_summary.py -> profiles/rNN_mfma_busy.json; the held
    clock is the in-kernel s_memtime / s_memrealtime ratio of the same launches inside the forward pass
    (tools/clock_conv_seq.py -> profiles/r05/conv_in_sequence_vs_back_to_back.txt)."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*_mfma_busy.json")))
    if not files:
        return {"mfma_busy": None, "frac_of_held_clock_peak": None}
    d = json.load(open(files[-1]))

    def norm(n):
        return "".join(ch for ch in n if ch.isalnum()).lower()
    want = norm(kernel_name)
    busy = None
    for k, v in d.get("kernels", {}).items():
        # rocprofv3 reports the mangled name: conv_igemm_big_kernelIDF16bLi192ELi256ELi8ELb0ELb0E... <-> <__bf16, 192, 256, 8, false>
        if "conv_igemm_big_kernel" in kernel_name and "conv_igemm_big_kernel" in k:
            tile = kernel_name.split("<")[1].split(">")[0].replace(" ", "").split(",")
            tag = {"__bf16": "DF16b", "_Float16": "DF16_", "float": "f"}.get(tile[0], "?")
            if f"I{tag}Li{tile[1]}ELi{tile[2]}ELi{tile[3]}ELb{1 if tile[4] == 'true' else 0}ELb0E" in k:
                busy = v
        elif norm(k).startswith(want[:20]):
            busy = v
    held = d.get("held_clock_ghz_in_sequence")
    out = {"mfma_busy": (busy or {}).get("mfma_busy"),
           "mfma_busy_conv_stack": d.get("conv_stack_mfma_busy"),
           "mfma_busy_source": os.path.basename(files[-1]),
           "mfma_busy_provenance": ("a COMMITTED constant, not measured by this run: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x "
                                    "1024 SIMDs), one rocprofv3 --kernel-trace --pmc pass of `python3 bench.py --lanes 1 --shared-plan "
                                    "--no-extras --no-verify --no-cpu-baseline` (tools/pmc_mfma.sh, file date " + str(d.get("_date")) +
                                    "); `mfma_busy` = this kernel, `mfma_busy_conv_stack` = stem + all convolutions + head, time-weighted"),
           "held_clock_ghz": held,
           "frac_of_held_clock_peak": (round(achieved_tflops / (peak_tflops * held / 2.4), 4) if held else None),
           "held_clock_note": "peak x held clock / 2.4 GHz: what the matrix pipes could deliver at the clock the power management "
                              "holds under this kernel inside the forward pass (in-kernel s_memtime / s_memrealtime)"}
    return out


def cpu_baseline(arch, sample, size):
    """The oracle timed on this host's cores on `sample` frames of the same workload (reported, not a target)."""
    from oracle import decode_ref as D, forward_ref as Fr
    # a 1-GPU box owns a 16-core share of its host; more threads than that only oversubscribe it
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = synth.make_state_dict(arch, 0, bn_stats=load_bn_stats(arch))
    u8 = prng.u8_frames(1234, sample, (size, size))
    x = Fr.normalize_u8(u8)
    Fr.forward_ref(sd, x[:1], arch)                               # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        head = Fr.forward_ref(sd, Fr.normalize_u8(u8), arch).numpy()
        for i in range(sample):
            D.decode_ref(head[i], insize=(size, size))
        reps += 1
        if time.perf_counter() - t0 > 10.0 or reps >= 200:
            break
    dt = time.perf_counter() - t0
    return {"value": round(sample * reps / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{reps} x {sample} frames {size}x{size}: torch-CPU fp32 oracle forward + NumPy decode, {dt:.1f} s"}


def cpu_baseline_train(arch, size):
    """One training iteration of the oracle (torch-CPU autograd restatement of main.py:664-777, f32) at batch 2 on
    this host's cores, repeated for ~10 s; reported per image."""
    from oracle import forward_ref as Fr, targets_ref as Tg, train_ref
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = synth.make_state_dict(arch, 0)
    b = 2
    x = Fr.normalize_u8(prng.u8_frames(99, b, (size, size)))
    tg = Tg.synthetic_batch(99, b, insize=(size, size), outsize=(size // 16, size // 16))
    t0 = time.perf_counter()
    reps = 0
    while True:
        train_ref.train_iteration_ref(sd, x, tg, [1.0] * 5, [1.0] * 5, arch, (size, size), dtype=torch.float32)
        reps += 1
        if time.perf_counter() - t0 > 10.0:
            break
    dt = time.perf_counter() - t0
    return {"value": round(b * reps / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} x {b} frames {size}x{size}: torch-CPU fp32 oracle train step "
                      f"(train-mode forward, loss, backward, 5 GradNorm probe gradients), {dt:.1f} s"}


def _per_rank_stats(dist, backend, dev, world, rate, aux=0.0):
    """N > 1: every rank's own images/s (its K steps until ITS GPU was done, before the closing barrier) and one auxiliary
    per-rank number, gathered on all ranks: a straggler, or a rank that fell back to something slow, shows as min << median."""
    if dist is None:
        return None
    t = torch.tensor([rate, aux], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    got = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(got, t)
    rates = sorted(float(g[0].item()) for g in got)
    auxs = sorted(float(g[1].item()) for g in got)

    def mmm(v, nd):
        return {"min": round(v[0], nd), "median": round(v[len(v) // 2], nd), "max": round(v[-1], nd)}
    return {"rates": dict(mmm(rates, 2), unit="images/sec per rank (own completion time, before the closing barrier)",
                          ranks=[round(float(g[0].item()), 2) for g in got]),
            "aux": dict(mmm(auxs, 4), ranks=[round(float(g[1].item()), 4) for g in got])}


def _fail_hook(rank):
    """Test hook (tests/test_bench_child_gpu.py): PPN_BENCH_FAIL_RANK=r makes rank r raise after the process group is
    up, to prove that a failing rank fails the whole run (non-zero exit, no JSON line) instead of hanging it."""
    if os.environ.get("PPN_BENCH_FAIL_RANK") == str(rank):
        raise RuntimeError(f"PPN_BENCH_FAIL_RANK: rank {rank} fails on request")


def main_train(args):
    """--workload train: BASELINE configs[3] per-GPU shard -- one PPNTrainer.train_step per step (train-mode forward,
    PPNLoss fwd+bwd, backward, GradNorm probes + task-weight step, gradient all-reduce over RCCL, Adam)."""
    from pytorch_pose_proposal_network_amd import arch as A, lib as L, targets
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.dtype in ("f16", "f16x3"):
        raise SystemExit("--workload train: the float16 / float16x3 modes are inference only (bf16 / f32)")
    # PPN_BENCH_BACKEND=gloo rehearses the N>1 control flow on a box with fewer GPUs than ranks (ranks share devices)
    backend = os.environ.get("PPN_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local if backend == "nccl" else local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    dist = None
    # PPN_BENCH_FORCE_DIST=1: join a process group even as the only rank (under torch.distributed.run with one process):
    # the RCCL calls of the N > 1 path -- init with device_id, barrier, all-reduce -- then execute on a 1-GPU box
    if world > 1 or os.environ.get("PPN_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # the gradient buckets are exchanged on RCCL's own streams while the backward still runs: ask for
            # high-priority ones, so that a bucket's ring kernel is dispatched ahead of queued conv workgroups
            kw = {}
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kw["pg_options"] = opts
            except Exception:                                      # noqa: BLE001 -- an older binding: default streams
                pass
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    _fail_hook(rank)
    B, S = args.batch, args.size
    tr = PPNTrainer(args.arch, synth.make_state_dict(args.arch, 0),
                    compute_dtype=L.PPN_BF16 if args.dtype == "bf16" else L.PPN_F32, insize=(S, S), device=dev,
                    second_order=not args.first_order)
    x = torch.from_numpy(synth.normalized_frames(prng.u8_frames(99 + rank, B, (S, S)))).to(dev)
    tg = targets.synthetic_targets(99 + 1000 * rank, B, (S, S), device=dev)      # encoded on the device
    for _ in range(args.warmup):
        tr.train_step(x, tg)

    def fence():
        torch.cuda.synchronize(dev)
        t_local = time.perf_counter()                 # this rank's own GPU is done (before it waits for the others)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        return t_local

    from pytorch_pose_proposal_network_amd import train as T_
    T_.BucketedAllReduce.measure = dist is not None
    fence()
    T_.BucketedAllReduce.exposed_ms()                 # drop the warm-up's events
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses, w = tr.train_step(x, tg)
    host_submit = time.perf_counter() - t0
    t_local = fence()
    dt = time.perf_counter() - t0
    exposed = T_.BucketedAllReduce.exposed_ms()
    per_rank = _per_rank_stats(dist, backend, dev, world, B * args.steps / (t_local - t0),
                               (sum(exposed) / len(exposed)) if exposed else 0.0)
    if dist is not None:
        t = torch.tensor([dt, host_submit], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, host_submit = float(t[0].item()), float(t[1].item())
    if rank == 0:
        flops = 3 * A.conv_flops(A.build_program(args.arch), S, S) * B          # fwd + dgrad + wgrad, SURVEY 8d
        peak = BF16_DENSE_PEAK_TFLOPS if args.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
        ach = flops / (dt / args.steps) / 1e12
        result = {
            "metric": "training images/sec (384x384, DRN-D-22, fwd/bwd + GradNorm + Adam)",
            "value": round(world * B * args.steps / dt, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "rccl_ranks": world if (dist is not None and backend == "nccl") else (1 if world == 1 else 0),
            "host": {"submit_ms_per_step_max_over_ranks": round(host_submit / args.steps * 1e3, 4),
                     "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "cpu_count": os.cpu_count()},
            "config": {"workload": f"{args.arch} PPN training step {args.dtype}, batch {B}/GPU synthetic {S}x{S} frames, "
                                   "targets of 1-4 synthetic people per frame encoded on the device "
                                   "(BASELINE configs[3] per-GPU shard; GradNorm "
                                   + ("without" if args.first_order else "with")
                                   + " the second-order term of main.py:759)",
                       "frames_per_gpu": B, "parallelism": f"minibatch sharded over {world} GPU(s), one all-reduce of "
                                                           "the flat 128.5 MB gradient buffer in 32 MB buckets under the "
                                                           "backward; the 5 task weights ride on the last bucket"},
            "roofline": {"bound": "mfma", "kernel": "whole step (3 x forward conv FLOPs / step time)",
                         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "traffic": None},
            "losses": [round(float(v), 4) for v in losses.tolist()], "task_weights": [round(float(v), 4) for v in w.tolist()],
        }
        if per_rank is not None:
            result["per_rank"] = per_rank["rates"]
            result["allreduce_exposed_ms_per_step"] = per_rank["aux"]
            result["allreduce_exposed_ms_per_step"]["what"] = (
                "time the main stream sits idle in BucketedAllReduce.finish() until the last gradient bucket has arrived (two "
                "timing events around the waits), mean over the timed steps, per rank: what the exchange adds to the step "
                "beyond what ran under the backward")
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_train(args.arch, S)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _people_equal(a, b):
    if a["n"] != b["n"]:
        return False
    return all(np.array_equal(a[k], b[k]) for k in ("kp_cell", "limb_arg", "bbox", "score"))


def verify_against_slices(net, frames, out):
    """Outside the timed region: the last step's people lists (batch B, the tile instantiations the timed steps ran)
    must equal, bit for bit, what B/2 independent batch-2 passes return (frames are independent units; the batch-2
    path is the one the golden-vector tests pin to the reference).  Returns the `batch_consistency` object of the JSON
    line: SELF-consistency of the benchmarked dtype, not parity with the reference -- that is `reference_agreement`
    (printed inside the same object) and the f32 parity mode."""
    from pytorch_pose_proposal_network_amd import rt
    if getattr(out, "ready", None) is not None:
        out.ready.synchronize()
    full = out.to_host()
    bad = []
    B = frames.shape[0]
    for i in range(0, B - 1, 2):
        part = rt.inference_batch(frames[i:i + 2].contiguous(), net).to_host()
        for j in range(2):
            if not _people_equal(full[i + j], part[j]):
                bad.append(i + j)
    return {"check": f"last timed step's decode result == {B // 2} independent batch-2 passes of the SAME dtype (every "
                     "index, box, score): self-consistency of the timed path, not parity with the reference",
            "frames": B, "people": int(sum(r["n"] for r in full)), "mismatching_frames": bad, "ok": not bad}


def _time_steps(fn, dev, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps


def extra_sections(args, dev, net, frames, dec):
    """Driver-visible numbers for the other BASELINE configs and modes (rank 0, N=1, outside the headline's timed
    loop; each a few steps, < 60 s in total).  A section that fails reports its error instead of failing the run."""
    from pytorch_pose_proposal_network_amd import arch as A, decode, drn, lib as L, model, rt, targets
    B, S = args.batch, args.size
    out = {}

    def section(name, fn):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:                                    # noqa: BLE001 -- reported, never hidden
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        torch.cuda.empty_cache()

    flops = A.conv_flops(A.build_program(args.arch), S, S) * B

    def materialized():
        # the model.forward() + get_humans_by_feature path: the f32 head [B,7605,24,24] is written and re-read
        def step():
            dec(net.forward_u8(frames))
        dt = _time_steps(step, dev, 10)
        return {"what": "forward writing the f32 NCHW head (17.5 MB/image) + stand-alone arg-max/NMS/parse decode",
                "images_per_sec": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3)}

    def f32_mode():
        n32 = model.PoseProposalNet(getattr(drn, args.arch)(), insize=(S, S), outsize=(S // 16, S // 16),
                                    compute_dtype="float32").cuda(dev)
        n32.load_state_dict(net.state_dict())
        d32 = decode.Decoder(B, (S // 16, S // 16), (S, S), device=dev)

        def step():
            u, k = n32.forward_u8(frames, fused_decode=True)
            d32.decode_fused(u, k)
        dt = _time_steps(step, dev, 3, warmup=3)
        return {"what": "the 1e-4 parity mode: exact-f32 MFMA (v_mfma_f32_16x16x4_f32), same fused path, batch %d" % B,
                "images_per_sec": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                "tflops": round(flops / dt / 1e12, 2), "frac_of_f32_mfma_peak": round(flops / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}

    def decode_stress():
        # BASELINE configs[4]: 32 planted 16-person crowds (+4 decoys), stand-alone decode of the dense f32 heads
        heads = np.stack([synth.planted_crowd_head(7 + i) for i in range(B)])
        hs = [torch.from_numpy(heads).to(dev) for _ in range(4)]   # 4 copies: 2.2 GB, far beyond the 256 MB MALL
        d = decode.Decoder(B, device=dev)
        res = {}
        for name, fn in (("limb_argmax", d.limb_argmax), ("decode", d)):
            for h in hs:
                fn(h)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for i, (a, b) in enumerate(ev):
                a.record(); fn(hs[i % 4]); b.record()
            torch.cuda.synchronize(dev)
            ms = sorted(a.elapsed_time(b) for a, b in ev)
            med = ms[len(ms) // 2]
            res[name] = {"us": round(med * 1e3, 1), "gbps": round(heads.nbytes / med / 1e6, 1),
                         "frac_of_hbm_peak": round(heads.nbytes / med / 1e6 / 8000.0, 4)}
        people = int(d(hs[0]).count.sum().item())
        # the same decode with TWO batches in flight (two Decoders on two streams, batches alternate): the 15 us parse
        # kernel of one batch -- one workgroup per image, latency-bound, it cannot start before its image's last arg-max
        # workgroup -- runs under the arg-max launch of the next.  Throughput over 40 batches, not a per-batch latency.
        sts = [torch.cuda.Stream(device=dev) for _ in range(2)]
        ds = [d, decode.Decoder(B, device=dev)]
        torch.cuda.synchronize(dev)
        def run(n):
            for i in range(n):
                with torch.cuda.stream(sts[i & 1]):
                    ds[i & 1](hs[i % 4])
        for s_ in sts:
            s_.wait_stream(torch.cuda.current_stream(dev))
        run(8)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s_ in sts:
            s_.wait_stream(torch.cuda.current_stream(dev))
        run(40)
        for s_ in sts:
            torch.cuda.current_stream(dev).wait_stream(s_)
        e1.record()
        torch.cuda.synchronize(dev)
        per = e0.elapsed_time(e1) / 40
        res["decode_two_in_flight"] = {"us_per_batch": round(per * 1e3, 1), "gbps": round(heads.nbytes / per / 1e6, 1),
                                       "frac_of_hbm_peak": round(heads.nbytes / per / 1e6 / 8000.0, 4),
                                       "what": "throughput with two batches in flight on two streams (40 batches)"}
        del hs
        return {"what": f"{B} planted-crowd heads f32 [7605,24,24] (seeds 7..{6 + B}): limb arg-max + NMS + limb parse, "
                        "median of 20 over 4 rotating copies", "algorithmic_bytes": int(heads.nbytes),
                "people": people, **res}

    def d54():
        # BASELINE configs[4]'s backbone end to end in every mode, with what each 16-bit mode costs in PEOPLE against the
        # f32 pipeline of the same network (no reference-generated people fixture exists for D-54 at 384x384; the f32 head
        # is pinned to the reference by tests/golden/forward_d54_384.npz)
        g = load_bn_stats("drn_d_54")
        sd54 = synth.make_state_dict("drn_d_54", 0, bn_stats=g)
        fl = A.conv_flops(A.build_program("drn_d_54"), S, S) * B
        res = {"what": f"DRN-D-54 (Bottleneck trunk) end to end, batch {B}, fused decode, one lane; people_vs_f32 = this "
                       "mode's people on the benchmark frames against the f32 pipeline's (same root / reproduced exactly)"}
        ref_people = None
        # round 5: D-54's 16-bit POLICY is float16 behind an exact prefix up to layer4 (6.2 % of its FLOPs as f32 / float16x3
        # launches): the CPU study tests/precision_study_d54.py (profiles/r05/precision_d54_*.txt) puts the 16-bit error of the
        # Bottleneck trunk in the same place as D-22's -- the first layers -- only ~20x more amplified: same root 40 -> 71 of 77
        # f32 people with that prefix (f16 tail), while no prefix below 24 % of the FLOPs rescues a bf16 tail (14 -> 46 -> 71)
        for mode, key, steps, kw in (("float32", "f32", 2, {}), ("float16x3", "f16x3", 3, {}),
                                     ("float16", "f16_exact_prefix4", 4, {"exact_prefix": 4}),
                                     ("float16", "f16", 5, {}), ("bfloat16", "bf16", 5, {})):
            n54 = model.PoseProposalNet(drn.drn_d_54(), insize=(S, S), outsize=(S // 16, S // 16), compute_dtype=mode, **kw).cuda(dev)
            n54.load_state_dict(sd54)
            d54_ = decode.Decoder(B, (S // 16, S // 16), (S, S), device=dev)

            def step():
                u, k = n54.forward_u8(frames, fused_decode=True)
                return d54_.decode_fused(u, k)
            dt = _time_steps(step, dev, steps, warmup=2)
            people = step().to_host()
            people = [{k: (v.copy() if hasattr(v, "copy") else v) for k, v in r.items()} for r in people]
            if ref_people is None:
                ref_people = people
            tot = np.zeros(5, np.int64)
            for a_, b_ in zip(ref_people, people):
                tot += np.array(decode.people_agreement(a_, b_))
            peak = F32_MFMA_PEAK_TFLOPS if mode == "float32" else BF16_DENSE_PEAK_TFLOPS
            res[key] = {"images_per_sec": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                        "tflops": round(fl / dt / 1e12, 1),
                        "frac_of_mfma_peak": round(fl / dt / 1e12 / peak, 4) if mode != "float16x3" else None,
                        "people_vs_f32": {"f32_people": int(tot[0]), "reproduced_exactly": int(tot[1]),
                                          "same_root": int(tot[2]), "same_root_frac": round(float(tot[2]) / max(int(tot[0]), 1), 4)},
                        "task_equivalent": bool(mode in ("float32", "float16x3") or tot[1] >= 0.85 * tot[0])}
            if kw:
                res[key]["policy"] = ("D-54's 16-bit policy: PoseProposalNet(drn_d_54(), compute_dtype='float16', exact_prefix=4) -- "
                                      "stem + layer3 + layer4 exact; shipped because it lifts the same-root share above 0.90")
            del n54, d54_
            torch.cuda.empty_cache()
        return res

    def f16x3_mode():
        # the tolerance-meeting mode above the exact-f32 MFMA rate: half-pair storage, three f16 MFMA
        # products per operand pair, f32 accumulation (csrc/conv_big.hip X3); stem and the cin < 64 convs stay exact f32
        nx = model.PoseProposalNet(getattr(drn, args.arch)(), insize=(S, S), outsize=(S // 16, S // 16),
                                   compute_dtype="float16x3").cuda(dev)
        nx.load_state_dict(net.state_dict())
        dx = decode.Decoder(B, (S // 16, S // 16), (S, S), device=dev)

        def step():
            u, k = nx.forward_u8(frames, fused_decode=True)
            dx.decode_fused(u, k)
        dt1 = _time_steps(step, dev, 5, warmup=3)
        lanes = max(1, args.lanes)
        pipex = rt.MultiLaneInference(nx, B, (S, S), device=dev, lanes=lanes)
        for _ in range(3 * lanes):
            pipex.submit(frames)
        dtl = _time_steps(lambda: pipex.submit(frames), dev, 10, warmup=3)
        pipex.close()
        res = {"what": f"float16x3: every value a half pair (hi, lo), a_hi w_hi + a_hi w_lo + a_lo w_hi on the f16 MFMA, f32 "
                       f"accumulation; meets north_star's 1e-4 / people tolerance (tests/test_x3_gpu.py), batch {B}",
               "images_per_sec": round(B / dtl, 1), "ms_per_step": round(dtl * 1e3, 3), "lanes": lanes,
               "images_per_sec_one_lane": round(B / dt1, 1)}
        if args.arch == "drn_d_22" and S == 384:
            res["agreement"] = _agreement(nx, "e2e_d22_384")
            res["agreement_tuned_checkpoint"] = _agreement(nx, "e2e_tuned_d22_384")
            g = np.load(os.path.join(ROOT, "tests", "golden", "forward_d22_384.npz"))
            u8 = torch.from_numpy(prng.u8_frames(int(g["seed_in"]), int(g["batch"]), (384, 384))).to(dev)
            sdg = synth.make_state_dict("drn_d_22", int(g["seed_w"]), bn_stats={k[3:]: g[k] for k in g.files if k.startswith("bn/")})
            nx.load_state_dict(sdg)
            hv = nx.forward_u8(u8).cpu().numpy().reshape(-1)[g["head_idx"]]
            res["head_max_abs_err_vs_reference"] = float(np.abs(hv - g["head_val"]).max())
        return res

    def train_shard():
        from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
        tr = PPNTrainer(args.arch, synth.make_state_dict(args.arch, 0), compute_dtype=L.PPN_BF16, insize=(S, S),
                        device=dev)
        x = torch.from_numpy(synth.normalized_frames(prng.u8_frames(99, B, (S, S)))).to(dev)
        tg = targets.synthetic_targets(99, B, (S, S), device=dev)
        dt = _time_steps(lambda: tr.train_step(x, tg), dev, 4, warmup=2)
        return {"what": f"BASELINE configs[3] per-GPU shard: one PPNTrainer.train_step (train-mode fwd, PPNLoss, bwd, "
                        f"GradNorm incl. second-order term, Adam), bf16, batch {B}",
                "images_per_sec": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                "frac_of_mfma_peak_3x_fwd_flops": round(3 * flops / dt / 1e12 / BF16_DENSE_PEAK_TFLOPS, 4)}

    def _fixture(name):
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        over = {k[len("override/"):]: g[k] for k in g.files if k.startswith("override/")}
        return g, int(g["batch"]), int(g["size"]), over

    def _agreement(model_, name):
        """`model_`'s people vs the reference pipeline's on fixture `name` (weights = seed 0 + the fixture's overrides)."""
        g, nb, sz, over = _fixture(name)
        if over:
            sd_ = dict(net.state_dict()); sd_.update({k: torch.from_numpy(v) for k, v in over.items()})
            model_.load_state_dict(sd_)
        fr = torch.from_numpy(prng.u8_frames(int(g["seed_in"]), nb, (sz, sz))).to(dev)
        got = rt.inference_batch(fr, model_).to_host()
        tot = np.zeros(5, np.int64)
        for i in range(nb):
            exp = {k: g[f"{i}/{k}"] for k in ("n", "kp_cell", "limb_arg")}
            tot += np.array(decode.people_agreement(exp, got[i]))
        if over:
            model_.load_state_dict(net.state_dict())
        n, exact, same, kp_eq, kp_all = (int(v) for v in tot)
        return {"reference_people": n, "reproduced_exactly": exact, "same_root": same, "keypoint_cells_equal": kp_eq,
                "keypoint_cells_compared": kp_all, "same_root_frac": round(same / max(n, 1), 4),
                "keypoint_cell_agreement": round(kp_eq / max(kp_all, 1), 4)}

    def bf16_agreement():
        # the benchmarked dtype against the REFERENCE pipeline's people lists (tests/golden/e2e_d22_384.npz: 8 frames,
        # forward + get_humans_by_feature of the reference itself); gated in tests/test_e2e_gpu.py
        g, nb, sz, _ = _fixture("e2e_d22_384")
        if args.arch != str(g["arch"]) or S != sz or args.dtype != "bf16":
            return {"skipped": "fixture is drn_d_22 384x384, bf16 mode"}
        nb_ = model.PoseProposalNet(getattr(drn, args.arch)(), insize=(S, S), outsize=(S // 16, S // 16),
                                    compute_dtype="bfloat16").cuda(dev)
        nb_.load_state_dict(net.state_dict())
        res = {"what": "bf16 fused path (the benchmarked configuration: stem + layer3-4 = 6.9 % of the FLOPs in IEEE half, "
                       "bf16 from layer5 on) vs the reference pipeline's people on 8 calibrated frames (dense synthetic "
                       "heads: ~490 root candidates per frame with near-equal scores)", **_agreement(nb_, "e2e_d22_384"),
               "tuned_checkpoint": {"what": "the same frames on the reference-fine-tuned checkpoint (bn2 / conv3.bias trained "
                                            "by the reference's PPNLoss + Adam until < 40 root candidates per frame: 76 people)",
                                    **_agreement(nb_, "e2e_tuned_d22_384")}}
        # the same with every launch in bf16 (round 3's configuration), and its speed beside the default's: the half prefix
        # costs nothing (same kernels, same rate), so `value` does not depend on it
        def one_lane(m_):
            d_ = decode.Decoder(B, (S // 16, S // 16), (S, S), device=dev)

            def step():
                u, k = m_.forward_u8(frames, fused_decode=True)
                d_.decode_fused(u, k)
            return round(B / _time_steps(step, dev, 10, warmup=3), 1)
        npure = model.PoseProposalNet(getattr(drn, args.arch)(), insize=(S, S), outsize=(S // 16, S // 16),
                                      compute_dtype="bfloat16", stem_dtype="bfloat16", half_prefix=-1).cuda(dev)
        npure.load_state_dict(net.state_dict())
        res["pure_bf16"] = {"what": "every launch in bf16, stem included (round 3's configuration)",
                            **_agreement(npure, "e2e_d22_384"),
                            "tuned_checkpoint_reproduced_exactly": _agreement(npure, "e2e_tuned_d22_384")["reproduced_exactly"],
                            "images_per_sec_one_lane": one_lane(npure)}
        res["images_per_sec_one_lane"] = one_lane(nb_)
        return res

    def ap_vs_reference():
        # what the reduced-precision modes cost in the TASK metric: the reference pipeline's people (fixture) taken as
        # ground truth, the HIP pipeline's people scored with the reference's own matcher/metric (evaluate.evaluation ==
        # datatest.evaluation).  `self` = the reference people scored against themselves: the metric's ceiling on these
        # dense synthetic crowds (overlapping people tie in the matcher), NOT 100.
        from pytorch_pose_proposal_network_amd import evaluate
        g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_d22_384.npz"))
        nb, sz = int(g["batch"]), int(g["size"])
        if args.arch != str(g["arch"]) or S != sz:
            return {"skipped": "fixture is drn_d_22 384x384"}
        fr = torch.from_numpy(prng.u8_frames(int(g["seed_in"]), nb, (sz, sz))).to(dev)
        exp = [{k: g[f"{i}/{k}"] for k in ("n", "kp_cell", "limb_arg", "bbox", "score")} for i in range(nb)]
        names = ["head", "shoulder", "elbow", "wrist", "hip", "knee", "ankle", "total"]
        res = {"what": "8 AP values (datatest.evaluation's metric, PCKh@0.5 matching) of each mode's people on the 8 "
                       "calibrated frames, the reference pipeline's 260 people as ground truth (keypoint = box centre, "
                       "head box = instance box)", "order": names,
               "self": [round(v, 2) for v in evaluate.ap_against_people(exp, exp)]}
        for mode in ("float32", "bfloat16") + (("float16",) if hasattr(L, "PPN_F16") else ()):
            n_ = model.PoseProposalNet(getattr(drn, args.arch)(), insize=(S, S), outsize=(S // 16, S // 16),
                                       compute_dtype=mode).cuda(dev)
            n_.load_state_dict(net.state_dict())
            got = rt.inference_batch(fr, n_).to_host()
            res[{"float32": "f32", "bfloat16": "bf16", "float16": "f16"}[mode]] = \
                [round(v, 2) for v in evaluate.ap_against_people(exp, got)]
            del n_
        return res

    def f16_mode():
        # the same fused path with IEEE-half operands (v_mfma_f32_16x16x32_f16: the bf16 MFMA rate, 11 significant bits
        # instead of 8): throughput (one lane and `lanes` lanes) and agreement with the reference pipeline's people
        n16 = model.PoseProposalNet(getattr(drn, args.arch)(), insize=(S, S), outsize=(S // 16, S // 16),
                                    compute_dtype="float16").cuda(dev)
        n16.load_state_dict(net.state_dict())
        d16 = decode.Decoder(B, (S // 16, S // 16), (S, S), device=dev)

        def step():
            u, k = n16.forward_u8(frames, fused_decode=True)
            d16.decode_fused(u, k)
        dt1 = _time_steps(step, dev, 10, warmup=3)
        lanes = max(1, args.lanes)
        pipe16 = rt.MultiLaneInference(n16, B, (S, S), device=dev, lanes=lanes)
        for _ in range(3 * lanes):
            pipe16.submit(frames)
        dtl = _time_steps(lambda: pipe16.submit(frames), dev, 20, warmup=3)
        pipe16.close()
        res = {"what": f"the fused path with f16 storage / MFMA operands (not BASELINE's dtype: reported beside it), batch {B}",
               "images_per_sec": round(B / dtl, 1), "ms_per_step": round(dtl * 1e3, 3), "lanes": lanes,
               "images_per_sec_one_lane": round(B / dt1, 1)}
        if args.arch == "drn_d_22" and S == 384:
            res["f16_agreement"] = _agreement(n16, "e2e_d22_384")
            res["f16_agreement_tuned_checkpoint"] = _agreement(n16, "e2e_tuned_d22_384")
        # the same trunk behind an EXACT prefix (stem + layer3 as f32 / float16x3 launches, 4.3 % of the FLOPs): the noise of
        # the first layers is what every later layer amplifies
        del pipe16
        nxp = model.PoseProposalNet(getattr(drn, args.arch)(), insize=(S, S), outsize=(S // 16, S // 16),
                                    compute_dtype="float16", exact_prefix=3).cuda(dev)
        nxp.load_state_dict(net.state_dict())
        pipex = rt.MultiLaneInference(nxp, B, (S, S), device=dev, lanes=lanes)
        for _ in range(3 * lanes):
            pipex.submit(frames)
        dtx = _time_steps(lambda: pipex.submit(frames), dev, 20, warmup=3)
        pipex.close()
        xp = {"what": "float16 trunk, stem + layer3 exact (PoseProposalNet(compute_dtype='float16', exact_prefix=3))",
              "images_per_sec": round(B / dtx, 1), "ms_per_step": round(dtx * 1e3, 3), "lanes": lanes}
        if args.arch == "drn_d_22" and S == 384:
            xp["agreement"] = _agreement(nxp, "e2e_d22_384")
            xp["agreement_tuned_checkpoint"] = _agreement(nxp, "e2e_tuned_d22_384")
        res["exact_prefix_3"] = xp
        return res

    def tile_policy_1():
        # the same bf16 path with the conv tiles chosen by efficiency alone (ppn_set_conv_tile_policy(1): 256x256 for every
        # >= 256-wide layer).  Not the headline configuration: the 256x256 tile makes 576 workgroups = 2.25 rounds on the
        # 48x48 layers, a last round only the OTHER lanes fill -- measured one launch in flight (what `roofline` reports) it
        # is slower than 192x256 at three whole rounds.  Reported beside it as what a throughput-only deployment gets.
        lanes = max(1, args.lanes)
        if lanes < 2:
            return {"skipped": "needs several lanes"}
        n1 = model.PoseProposalNet(getattr(drn, args.arch)(), insize=(S, S), outsize=(S // 16, S // 16),
                                   compute_dtype="bfloat16").cuda(dev)
        n1.load_state_dict(net.state_dict())
        pipe1 = rt.MultiLaneInference(n1, B, (S, S), device=dev, lanes=lanes, tile_policy=1)
        try:
            for _ in range(3 * lanes):
                pipe1.submit(frames)
            torch.cuda.synchronize(dev)
            dts = sorted(_time_steps(lambda: pipe1.submit(frames), dev, 20, warmup=3) for _ in range(3))
            last = pipe1.submit(frames)
            pipe1.flush()
            people = int(last.count.sum().item())
        finally:
            pipe1.close()
        return {"what": "bf16, conv tiles by efficiency alone (--tile-policy 1), same lanes; median of three 20-step windows",
                "images_per_sec": round(B / dts[1], 1), "ms_per_step": round(dts[1] * 1e3, 3), "lanes": lanes,
                "people_last_step": people}

    section("tile_policy_1", tile_policy_1)
    section("bf16_agreement", bf16_agreement)
    section("ap_vs_reference", ap_vs_reference)
    section("f16_mode", f16_mode)
    section("f16x3_mode", f16x3_mode)
    section("materialized_head", materialized)
    section("decode_stress", decode_stress)
    section("f32_parity_mode", f32_mode)
    section("d54_end_to_end", d54)
    section("train_shard", train_shard)
    return out


def self_launch(argv, gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1) as a CHILD process, relay its output, exit with its code.
    Runs before anything touches the GPU in this process; never exec."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in proc.stdout:                     # rank 0's JSON line (other ranks print nothing on stdout)
        lines.append(line)
    rc = proc.wait()
    json_lines = [l for l in lines if l.lstrip().startswith("{")]
    for l in lines:
        if l not in json_lines:
            sys.stderr.write(l)
    if rc != 0 or not json_lines:
        sys.stderr.write(f"bench.py: the {gpus}-rank run failed (exit code {rc})\n")
        raise SystemExit(rc or 1)
    sys.stdout.write(json_lines[-1])
    sys.stdout.flush()
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="inference", choices=["inference", "train"],
                    help="inference = BASELINE configs[1] (the headline metric, default); train = configs[3] shard")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=10,
                    help="timing windows of --steps steps each (the first is `value`; all go into `value_windows`)")
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU")
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--arch", default="drn_d_22")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16", "f16x3"],
                    help="bf16 = BASELINE configs[1] (the headline); f32 = the 1e-4 parity mode; f16 = IEEE half at the bf16 MFMA "
                         "rate (inference only); f16x3 = split-f16 operands, three products per pair: meets the 1e-4 tolerance "
                         "at ~2.7x the f32 rate (inference only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the batch-consistency check of the last timed step (profiling runs: its batch-2 passes would mix "
                         "other kernel instantiations into the per-kernel statistics)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra sections (materialized_head, decode_stress, f32_parity_mode, d54_end_to_end, "
                         "train_shard) reported beside the headline at N=1")
    ap.add_argument("--layers", action="store_true", help="print the per-layer table to stderr")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("PPN_LANES", "2")),
                    help="stream lanes successive steps alternate between (rt.MultiLaneInference); 1 = one lane with "
                         "the same kernels (what the rocprofv3 per-kernel durations are compared with).  Default 2 since round 5: "
                         "with the weight-prefetch hint and the persistent one-launch BasicBlocks two lanes run 11.4 k images/s "
                         "where three run 11.15-11.2 k (three interleaved pairs on one box, profiles/r05/lanes_x_policy.txt); "
                         "rounds 2-4 ran three")
    ap.add_argument("--shared-plan", action="store_true",
                    help="with --lanes 1: run the MULTI-lane plan (generic 64 -> 64 kernel, no lone-launch tiles) with one launch in "
                         "flight -- profiler runs, so that per-kernel durations describe the kernels the multi-lane headline runs")
    ap.add_argument("--first-order", action="store_true",
                    help="--workload train: model gradient = d loss/d theta only (skip d Lgrad/d theta of main.py:759)")
    ap.add_argument("--tile-policy", type=int, default=0, choices=[0, 1, 2],
                    help="1 = conv tiles by efficiency alone (ppn_set_conv_tile_policy; +1.5 %% with three lanes, but the "
                         "per-launch roofline then describes the 256x256 tile whose partial last round only the other lanes fill)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="plain serial path on the caller's stream: forward, then decode (single-stream tile policy)")
    ap.add_argument("--materialize-head", action="store_true",
                    help="write the f32 head tensor [B,7605,24,24] and decode it with the stand-alone arg-max kernel "
                         "(model.forward + get_humans_by_feature path) instead of the fused inference path")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_launch(sys.argv[1:], args.gpus)
    if args.workload == "train":
        return main_train(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one process per GPU.  PPN_BENCH_BACKEND=gloo rehearses the N>1 control flow on a box with fewer GPUs than
    # ranks (ranks then share devices; timing is meaningless there, the driver's runs use RCCL = "nccl").
    backend = os.environ.get("PPN_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or os.environ.get("PPN_BENCH_FORCE_DIST") == "1":     # (see main_train)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from pytorch_pose_proposal_network_amd import decode, drn, model

    _fail_hook(rank)
    B, S = args.batch, args.size
    net = model.PoseProposalNet(getattr(drn, args.arch)(), insize=(S, S), outsize=(S // 16, S // 16),
                                compute_dtype={"bf16": "bfloat16", "f32": "float32", "f16": "float16",
                                               "f16x3": "float16x3"}[args.dtype]).cuda(dev)
    net.load_state_dict(synth.make_state_dict(args.arch, 0, bn_stats=load_bn_stats(args.arch)))
    net.eval()
    # resident in HBM: NROT distinct batches, step i reads batch i % NROT (no step re-reads its predecessor's frames)
    NROT = 3
    frame_sets = [torch.from_numpy(prng.u8_frames(1234 + rank + 7919 * j, B, (S, S))).to(dev) for j in range(NROT)]
    frames = frame_sets[0]
    dec = decode.Decoder(B, (S // 16, S // 16), (S, S), device=dev)

    fused = not args.materialize_head
    pipe = None
    if fused and not args.no_pipeline:
        from pytorch_pose_proposal_network_amd import rt
        pipe = rt.MultiLaneInference(net, B, (S, S), device=dev, lanes=max(1, args.lanes),
                                     tile_policy=args.tile_policy, shared_plan=True if args.shared_plan else None)

    step_no = [0]

    def step():
        fr = frame_sets[step_no[0] % NROT]
        step_no[0] += 1
        if pipe is not None:   # conv stack of step i+1 overlaps the NMS/limb-parse kernel of step i (side stream)
            return pipe.submit(fr)
        if fused:       # rt_test.inference path: the head conv's epilogue runs the limb arg-max, no head tensor
            unary, keys = net.forward_u8(fr, fused_decode=True)
            return dec.decode_fused(unary, keys)
        head = net.forward_u8(fr)
        return dec(head)

    if pipe is not None:            # set-up, not measurement: every lane's plan reaches its captured-graph state
        for _ in range(3 * max(1, args.lanes)):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize(dev)
        t_local = time.perf_counter()                 # this rank's own GPU is done (before it waits for the others)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        return t_local

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    host_submit = time.perf_counter() - t0                      # this rank's host time enqueueing the K steps
    t_local = fence()
    dt = time.perf_counter() - t0
    per_rank = _per_rank_stats(dist, backend, dev, world, B * args.steps / (t_local - t0))
    if dist is not None:
        t = torch.tensor([dt, host_submit], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, host_submit = float(t[0].item()), float(t[1].item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    people = int(out.count.sum().item())
    last_frames = frame_sets[(step_no[0] - 1) % NROT]            # what the last timed step read
    if fused and rank == 0 and not args.no_verify:
        # `out` lives in its lane's / decoder's buffers: read it before the extra timing windows below reuse them
        if getattr(out, "ready", None) is not None:
            out.ready.synchronize()
        out_host_early = out.to_host()
    else:
        out_host_early = None
    # ---- the same K-step window nine more times (every rank takes part: the fences are collective) --------------
    # `value` stays the FIRST window (the contract's exactly-K timed steps); `value_windows` says how representative
    # that one ~60 ms window is.
    win = [dt]
    for _ in range(0 if args.windows < 2 else args.windows - 1):
        fence()
        t0w = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dtw = time.perf_counter() - t0w
        if dist is not None:
            t = torch.tensor([dtw], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtw = float(t.item())
        win.append(dtw)
    win_vals = sorted(world * B * args.steps / w for w in win)

    result = None
    if rank == 0:
        # the people lists of the last timed step, checked before anything else reuses the lanes' buffers
        if out_host_early is not None:
            class _Held:                               # the early host copy of the last timed step's result
                ready = None

                @staticmethod
                def to_host():
                    return out_host_early
            out = _Held
        verified = verify_against_slices(net, last_frames, out) if (fused and B % 2 == 0 and not args.no_verify) else None
        # ---- per-kernel durations: HIP events on the launch stream around every launch ----------------
        # per-launch durations with one launch in flight (what rocprofv3 shows for `--lanes 1`)
        # Primary figures: every launch ONCE in plan order (in sequence: each launch finds the caches as the previous
        # layer left them -- what rocprofv3 `--lanes 1` averages).  Secondary (`back_to_back`): four repeats of the same
        # launch between its two events, i.e. warm caches and no launch gap, the figure round 1 quoted (~5 % higher).
        agg, table, agg4 = {}, [], {}
        reps = 5
        # the same plan flags as the timed path (a multi-lane plan leaves out the filter-bank kernel and the lone-launch
        # tiles): the per-launch figures describe the kernels `value` ran
        pflags = getattr(pipe, "_conv_flags", 0) if pipe is not None else 0
        net.profile_layers(frames, src_is_u8=True, repeats=1, fused_decode=fused, conv_flags=pflags)   # untimed: first direct launches
        for r in range(reps):
            for name, kern, ms, fl in net.profile_layers(frames, src_is_u8=True, repeats=1, fused_decode=fused, conv_flags=pflags):
                a = agg.setdefault(kern, [0.0, 0.0, 0])
                a[0] += ms; a[1] += fl; a[2] += 1
                if r == 0:
                    table.append((name, kern, ms, fl))
            for name, kern, ms, fl in net.profile_layers(frames, src_is_u8=True, repeats=4, fused_decode=fused, conv_flags=pflags):
                a = agg4.setdefault(kern, [0.0, 0.0, 0])
                a[0] += ms; a[1] += fl; a[2] += 1
        # An event pair around ONE launch also spans the marker and launch latency (6-7 us around an empty kernel on
        # this stack; rocprofv3's begin-to-end durations do not contain it): calibrated here on a 64-byte fill and
        # subtracted per launch, so that `avg_launch_us` is comparable with the rocprofv3 summary under profiles/.
        probe = torch.zeros(16, device=dev)
        pe = [torch.cuda.Event(enable_timing=True) for _ in range(100)]
        for i in range(50):
            pe[2 * i].record(); probe.zero_(); pe[2 * i + 1].record()
        torch.cuda.synchronize(dev)
        ev_over_ms = min(pe[2 * i].elapsed_time(pe[2 * i + 1]) for i in range(10, 50))
        raw_dom = max(agg.items(), key=lambda kv: kv[1][0])
        raw_us = raw_dom[1][0] / raw_dom[1][2] * 1e3
        for a in agg.values():
            a[0] = max(a[0] - a[2] * ev_over_ms, 1e-6)
        fwd_ms = sum(a[0] for a in agg.values()) / reps
        fwd_flops = sum(a[1] for a in agg.values()) / reps
        dom = max(agg.items(), key=lambda kv: kv[1][0])
        dk, (dms, dfl, dn) = dom
        # f16 MFMA = the bf16 rate; f16x3 spends three f16 MFMAs per algorithmic multiply-add: a third of it
        peak = (F32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else
                BF16_DENSE_PEAK_TFLOPS / 3.0 if args.dtype == "f16x3" else BF16_DENSE_PEAK_TFLOPS)
        achieved = dfl / (dms * 1e-3) / 1e12
        if args.layers:
            for name, kern, ms, fl in table:
                print(f"{name:28s} {ms*1e3:9.1f} us {fl/ms/1e9 if ms > 0 else 0:9.1f} TFLOP/s  {kern}", file=sys.stderr)
        # decode kernels: dense bytes of the head read once
        # median of seven calls behind two untimed ones (the first call of a Decoder sets kernel attributes)
        if fused:
            unary, keys = net.forward_u8(frames, fused_decode=True)
            dfn = lambda: dec.decode_fused(unary, keys)                  # noqa: E731
        else:
            head = net.forward_u8(frames)
            dfn = lambda: dec(head)                                      # noqa: E731
        dfn(); dfn()
        torch.cuda.synchronize(dev)
        dts = []
        for _ in range(7):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record(); dfn(); ev[1].record()
            torch.cuda.synchronize(dev)
            dts.append(ev[0].elapsed_time(ev[1]))
        dec_ms = sorted(dts)[3]
        head_bytes = B * cfg.lastsize() * (S // 16) * (S // 16) * 4
        result = {
            "metric": "images/sec (384x384, DRN-D-22) at 1/2/4/8 MI355X",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "lanes": (max(1, args.lanes) if pipe is not None else 1),
            "host": {"submit_ms_per_step_max_over_ranks": round(host_submit / args.steps * 1e3, 4),
                     "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "cpu_count": os.cpu_count(),
                     "what": "host time one rank spends enqueueing a step (max over ranks): the N ranks of a node share its "
                             "cores; bench.py's self-launch gives each rank cpu_count // N OpenMP threads"},
            "value_windows": {"what": f"{len(win)} consecutive windows of {args.steps} steps each, same fences; `value` "
                                      "is the first", "n": len(win), "min": round(win_vals[0], 2),
                              "median": round(win_vals[len(win_vals) // 2], 2), "max": round(win_vals[-1], 2)},
            "config": {"workload": f"{args.arch} PPN inference {args.dtype}"
                                   + ((f" (stem + layer3-{net.half_prefix} = 6.9 % of the FLOPs in IEEE half at the same MFMA rate, "
                                       "the model's default: `bf16_agreement.pure_bf16` has the all-bf16 numbers)")
                                      if (args.dtype == "bf16" and getattr(net, "half_prefix", -1) >= 3) else "")
                                   + f", batch {B}/GPU synthetic {S}x{S} u8 frames: "
                                   "fused normalise + conv stack + head + decode/NMS/limb-parse (BASELINE configs[1])"
                                   + ("" if fused else ", head tensor materialised")
                                   + (f", batches go round-robin over {max(1, args.lanes)} stream lanes (rt.MultiLaneInference)"
                                      if pipe is not None else ""),
                       "frames_per_gpu": B,
                       "input": f"{S}x{S}x3 u8 resident in HBM, {NROT} distinct batches read round-robin by the steps",
                       "tolerances": ("bf16 (BASELINE configs[1]'s dtype) does NOT meet north_star's 1e-4 head / bit-exact "
                                      "index tolerance (head within 0.15 max / 0.02 mean of the reference head; "
                                      "`reference_agreement` and `ap_vs_reference` say what that costs); the mode that "
                                      "does is `f32_parity_mode`.  D-54 f32 deviates from 1e-4 by rule: its head must be "
                                      "within 1e-4 of the reference OR within 1.5x the reference's own f32-vs-f64 distance "
                                      "(2.8e-4 @96, 4e-4 @384) of the fp64 head"),
                       "head": (f"{cfg.lastsize()}x{S//16}x{S//16} f32 per image, NOT materialised: the head conv's "
                                "epilogue keeps the 108 unary channels and one arg-max key per (edge, cell)"
                                if fused else f"{cfg.lastsize()}x{S//16}x{S//16} f32 written to HBM"),
                       "excluded_from_value": "H2D of the u8 frames and D2H of the compact people lists "
                                              "(`pcie_inclusive` is the rate with both)",
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
            "rccl_ranks": world if (dist is not None and backend == "nccl") else (1 if world == 1 else 0),
            **({"per_rank": per_rank["rates"]} if per_rank is not None else {}),
            "roofline": {"bound": "mfma", "kernel": dk, "launches_per_step": dn // reps,
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "timing": "HIP events on the launch stream around every launch of one pass over the plan, one "
                                   "launch in flight, in sequence (as rocprofv3 `--lanes 1`; with several lanes dispatches "
                                   "share the GPU and each one's begin-to-end time grows); the event pair's own "
                                   "marker + launch latency, calibrated on an empty kernel, is subtracted per launch",
                         "event_overhead_us": round(ev_over_ms * 1e3, 2), "avg_launch_us_raw": round(raw_us, 2),
                         "back_to_back": {"what": "the same launches, four repeats between the two events (warm caches)",
                                          "avg_launch_us": round(agg4[dk][0] / agg4[dk][2] * 1e3, 2),
                                          "frac": round(agg4[dk][1] / (agg4[dk][0] * 1e-3) / 1e12 / peak, 4)},
                         "traffic": (pmc_traffic(dk) or {}).get("bytes_per_launch"),
                         "traffic_source": (pmc_traffic(dk) or {}).get("source"),
                         "traffic_provenance": ("a COMMITTED constant, not measured by this run: HBM bytes per launch of this "
                                                "kernel from two separate rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE, "
                                                "--lanes 1) of `python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "
                                                "--no-extras --no-verify --lanes 1`, tools/pmc_bench.sh -> "
                                                "tools/pmc_traffic_summary.py -> profiles/" +
                                                str((pmc_traffic(dk) or {}).get("source")) + " (file date " +
                                                str((pmc_traffic(dk) or {}).get("date")) + "); bench.py cannot run the profiler on itself"),
                         "avg_launch_us": round(dms / dn * 1e3, 2),
                         "flops_per_launch_avg": round(dfl / dn),
                         **mfma_busy(dk, achieved, peak)},
            "step_tflops": round(fwd_flops / ms_per_step / 1e9, 2),      # conv FLOPs / whole-step time (lanes overlap)
            "conv_stack": {"ms": round(fwd_ms, 4), "tflops": round(fwd_flops / fwd_ms / 1e9, 2),
                           "frac_of_mfma_peak": round(fwd_flops / fwd_ms / 1e9 / peak, 4),
                           "gflop_per_image": round(fwd_flops / B / 1e9, 3)},
            "decode": ({"mode": "fused into the head conv epilogue (no head tensor); NMS + limb parse kernel only",
                        "ms": round(dec_ms, 4), "people": people} if fused else
                       {"mode": "stand-alone: dense limb arg-max over the materialised head + NMS + limb parse",
                        "ms": round(dec_ms, 4), "gbps": round(head_bytes / dec_ms / 1e6, 1),
                        "frac_of_hbm_peak": round(head_bytes / dec_ms / 1e6 / 8000.0, 4), "people": people}),
        }
        if world == 1:
            # PCIe-inclusive rate (NOT `value`): every step first copies the u8 frames from pinned host memory and
            # ends with the compact decode result (counts, cells, boxes, scores) back on the host.
            host = torch.from_numpy(prng.u8_frames(1234, B, (S, S))).pin_memory()
            n = max(6, args.steps)
            if pipe is not None:                 # untimed: every lane allocates its pinned read-back buffers once
                for r in [pipe.submit(host, to_host=True) for _ in range(max(1, args.lanes))]:
                    r.ready.synchronize()
            torch.cuda.synchronize(dev)
            # the two copies alone (no compute queued): says whether a low PCIe-inclusive rate is the link / the NUMA
            # placement of the pinned pages on this box, or the pipeline
            h2d_dst = torch.empty_like(host, device=dev)
            h2d_dst.copy_(host, non_blocking=True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(10):
                h2d_dst.copy_(host, non_blocking=True)
            torch.cuda.synchronize(dev)
            h2d_ms = (time.perf_counter() - t0) / 10 * 1e3
            del h2d_dst
            t1 = time.perf_counter()
            if pipe is not None:
                # pinned frames -> the lane's own input buffer (H2D on the lane's stream) -> step -> the compact result
                # into the lane's pinned buffers (D2H behind the decode); the host unpacks batch i - lanes + 1 while the
                # later batches run: no synchronous copy anywhere
                pending = []
                host_s = [0.0, 0.0, 0.0]             # host seconds in submit / waiting on the ready event / unpack

                def collect(r):
                    ta = time.perf_counter()
                    r.ready.synchronize()
                    tb = time.perf_counter()
                    out = r.hosted.unpack()
                    host_s[1] += tb - ta
                    host_s[2] += time.perf_counter() - tb
                    return out

                for it in range(n):
                    ta = time.perf_counter()
                    pending.append(pipe.submit(host, to_host=True))
                    host_s[0] += time.perf_counter() - ta
                    if len(pending) == max(1, args.lanes):
                        hosted = collect(pending.pop(0))
                for r in pending:
                    hosted = collect(r)
                pcie_people = sum(h["n"] for h in hosted)
            else:
                for it in range(n):
                    frames.copy_(host, non_blocking=True)
                    hosted = step().to_host()
                pcie_people = sum(h["n"] for h in hosted)
            torch.cuda.synchronize(dev)
            dt1 = time.perf_counter() - t1
            result["pcie_inclusive"] = {"value": round(B * n / dt1, 2), "unit": "images/sec",
                                        "ms_per_step": round(dt1 / n * 1e3, 4), "h2d_bytes_per_step": host.numel(),
                                        "host_ms_per_step": ({k: round(v / n * 1e3, 4) for k, v in
                                                              zip(("submit", "wait", "unpack"), host_s)}
                                                             if pipe is not None else None),
                                        "h2d_alone_ms": round(h2d_ms, 4),
                                        "h2d_alone_gbps": round(host.numel() / h2d_ms / 1e6, 2),
                                        "people": int(pcie_people),
                                        "slow_mode": bool(B * n / dt1 < 0.75 * value),
                                        "slow_mode_note": "true = this process ran the PCIe loop in the slow mode seen on some boxes / processes "
                                                          "(< 0.75 of `value` although h2d_alone_gbps is normal: host submit or GPU-side "
                                                          "wait 2-3x the usual, profiles/r03/pcie_modes_sdma.txt); `value` is unaffected "
                                                          "(frames resident in HBM)",
                                        "note": "per step: H2D of the pinned u8 frames into the lane's input buffer, the "
                                                "step, D2H of the compact result (first 64 people slots per image) into pinned "
                                                "buffers, all queued on the lane's stream; the host unpacks a batch while the "
                                                "next ones run"}
        if world == 1 and not args.no_extras:
            if pipe is not None:
                pipe.close()
            result.update(extra_sections(args, dev, net, frames, dec))
        # what `value` is worth in the TASK metric, right beside it: the timed mode's people against the reference pipeline's
        # (8 calibrated frames, the reference's own matcher; `ap_vs_reference` / `bf16_agreement` hold the details)
        apv, agr = result.get("ap_vs_reference", {}), result.get("bf16_agreement", {})
        if args.dtype == "bf16" and "bf16" in apv and "reproduced_exactly" in agr:
            result["value_fidelity"] = {"total_ap": apv["bf16"][-1], "ap_ceiling": apv["self"][-1],
                                        "reference_people_reproduced_exactly": f"{agr['reproduced_exactly']}/{agr['reference_people']}",
                                        "same_root": f"{agr['same_root']}/{agr['reference_people']}",
                                        "task_equivalent": bool(agr["reproduced_exactly"] >= 0.97 * agr["reference_people"]),
                                        "modes_that_are": "float16x3 (`f16x3_mode`), float32 (`f32_parity_mode`): all people, AP = ceiling"}
        if verified is not None:
            if "bf16_agreement" in result or "reference_agreement" in result:
                ra = result.get("reference_agreement", result.get("bf16_agreement"))
                verified["reference_agreement"] = {k: ra[k] for k in ("reference_people", "reproduced_exactly", "same_root",
                                                                      "same_root_frac", "keypoint_cell_agreement")
                                                   if k in ra}
            result["batch_consistency"] = verified
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.arch, 4, S)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
        if result.get("batch_consistency") is not None and not result["batch_consistency"]["ok"]:
            raise SystemExit("bench.py: the timed path's output differs from the batch-2 reference path")


if __name__ == "__main__":
    main()
