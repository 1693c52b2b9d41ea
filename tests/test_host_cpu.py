"""CPU-only checks of the host side: the C-ABI library exports every symbol include/ppn.h declares, the ctypes
structs match the C structs, rejects bad arguments without a GPU, and frame sharding over 2 gloo ranks
reproduces the single-process result."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from pytorch_pose_proposal_network_amd import build, lib
    if not os.path.exists(lib.LIB_PATH):
        build.build(verbose=False)
    return lib


def test_header_symbols_exported():
    lib = _lib()
    hdr = open(os.path.join(ROOT, "include", "ppn.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ppn_[a-z0-9_]+)\s*\(", hdr))
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (ppn_[a-z0-9_]+)", out))
    assert declared, "no declarations parsed"
    assert declared <= exported, f"declared but not exported: {sorted(declared - exported)}"
    assert set(lib.EXPORTS) == declared, (sorted(set(lib.EXPORTS) ^ declared))


def test_library_loads_and_reports_errors_without_gpu():
    lib = _lib()
    l = lib.load()
    assert l.ppn_version() >= 1
    # argument validation happens before any device work
    assert l.ppn_conv2d_fused(None, None) != 0
    assert b"NULL" in l.ppn_last_error()
    cfg = lib.DecodeCfg()
    assert l.ppn_decode(C.byref(cfg), None, 1, None, None, None, None, None, None, None) != 0
    k, c, o = C.c_int32(), C.c_int32(), C.c_int32()
    assert l.ppn_conv_tiling(lib.PPN_BF16, 512, 512, 3, C.byref(k), C.byref(c), C.byref(o)) == 0
    assert (k.value, c.value, o.value) == (64, 256, 1)
    assert l.ppn_conv_tiling(lib.PPN_F32, 512, 7605, 1, C.byref(k), C.byref(c), C.byref(o)) == 0
    assert (k.value, c.value) == (32, 256)      # pad granularity = largest channel tile for this Cout
    assert l.ppn_conv_tiling(lib.PPN_BF16, 16, 32, 3, C.byref(k), C.byref(c), C.byref(o)) == 0
    assert o.value == 2 and k.value == 144
    assert l.ppn_conv_tiling(7, 16, 32, 3, None, None, None) != 0


def test_ctypes_structs_match_header():
    """sizeof of the ctypes mirrors == sizeof of the C structs (compiled with gcc from include/ppn.h)."""
    lib = _lib()
    src = '#include <stdio.h>\n#include "ppn.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu ", sizeof(ppn_decode_cfg), sizeof(ppn_conv_desc), sizeof(ppn_bn_desc), sizeof(ppn_bn_bwd_desc), sizeof(ppn_loss_cfg), sizeof(ppn_wgrad_desc));printf("%zu\\n", sizeof(ppn_block_desc));return 0;}\n'
    exe = os.path.join("/tmp", "ppn_sizeof")
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    a, b, c, d, e, f, g = map(int, subprocess.check_output([exe]).split())
    assert f == C.sizeof(lib.WgradDesc) and g == C.sizeof(lib.BlockDesc)
    assert a == C.sizeof(lib.DecodeCfg) and b == C.sizeof(lib.ConvDesc)
    assert c == C.sizeof(lib.BnDesc) and d == C.sizeof(lib.BnBwdDesc) and e == C.sizeof(lib.LossCfg)


def test_decode_cfg_tables():
    from pytorch_pose_proposal_network_amd import config as cfg, decode
    c = decode.make_cfg()
    assert (c.K, c.E, c.sH, c.sW, c.H, c.W, c.max_humans) == (18, 17, 21, 21, 24, 24, 576)
    order = [c.edge_order[i] for i in range(c.E)]
    assert sorted(order) == list(range(17))
    seen = {0}
    for e in order:                       # parent-before-child
        assert c.edge_src[e] in seen
        seen.add(c.edge_dst[e])


def test_frame_shard_roundtrip():
    from pytorch_pose_proposal_network_amd import shard
    for n in (0, 1, 7, 32):
        for world in (1, 2, 3, 8):
            parts = [[f"f{i}" for i in shard.frame_shard(n, r, world)] for r in range(world)]
            assert shard.merge_shards(parts, n) == [f"f{i}" for i in range(n)]
    with pytest.raises(ValueError):
        shard.frame_shard(4, 2, 2)


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch.distributed as dist
from oracle import decode_ref as D
from pytorch_pose_proposal_network_amd import shard, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 6
mine = [int(D.decode_ref(synth.planted_crowd_head(7 + i))["n"]) for i in shard.frame_shard(n, rank, world)]
allr = shard.gather_results(mine, n)
if rank == 0:
    print("RESULT", allr)
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding_matches_single_process(tmp_path):
    """world_size-2 rehearsal of the N>1 inference path on CPU (gloo): shard, decode locally, gather."""
    from oracle import decode_ref as D
    from pytorch_pose_proposal_network_amd import synth
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT")][0]
    expected = [int(D.decode_ref(synth.planted_crowd_head(7 + i))["n"]) for i in range(6)]
    assert eval(line[len("RESULT "):]) == expected


_TRAIN_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from pytorch_pose_proposal_network_amd import train as T
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = torch.Generator().manual_seed(100 + rank)
flat = torch.randn(1000, generator=g)            # this rank's flat gradient buffer
w = torch.tensor([1.0, 2.0, 0.5, 1.5, 0.25]) * (rank + 1)
scale = T.allreduce_mean_(flat)                  # SUM over ranks, 1/world handed to the optimiser kernel
dist.all_reduce(w)                               # main.py:769-771
if rank == 0:
    print("RESULT", scale, float((flat * scale).sum()), (w / world).tolist())
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_gloo_gradient_exchange(tmp_path):
    """world_size-2 rehearsal of the training exchange step on CPU (gloo): one SUM all-reduce of the flat gradient
    buffer, the 1/world factor folded into the Adam launch, task weights averaged (main.py:769-771, 1233-1238)."""
    import torch
    script = tmp_path / "train_worker.py"
    script.write_text(_TRAIN_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29619", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT")][0]
    parts = line.split(" ", 3)
    scale, total, w = float(parts[1]), float(parts[2]), eval(parts[3])
    grads = [torch.randn(1000, generator=torch.Generator().manual_seed(100 + r)) for r in range(2)]
    assert scale == 0.5
    assert abs(total - float(((grads[0] + grads[1]) / 2).sum())) < 1e-4
    assert w == [1.5, 3.0, 0.75, 2.25, 0.375]


_BUCKET_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from pytorch_pose_proposal_network_amd import train as T
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 1000
flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
ex = T.BucketedAllReduce(flat, bucket_elems=300)           # buckets [700,1000) [400,700) [100,400) [0,100)
calls = []
ex.ready(950)                                               # nothing complete yet
a = len(ex.handles)
ex.ready(650, before_issue=lambda: calls.append(1))         # first bucket [700,1000) is final
b = len(ex.handles)
ex.ready(120)                                               # one more: [400,700); [100,400) still has open entries
c = len(ex.handles)
scale = ex.finish()
if rank == 0:
    print("RESULT", a, b, c, len(calls), scale, float(flat.sum()))
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_gloo_bucketed_exchange(tmp_path):
    """BucketedAllReduce: tail-to-head buckets are issued as `ready(offset)` passes their lower bound, the whole
    buffer ends up summed over the ranks, the 1/world factor is returned for the optimiser launch."""
    script = tmp_path / "bucket_worker.py"
    script.write_text(_BUCKET_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29621", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT")][0].split()
    a, b, c, ncalls, scale, total = int(line[1]), int(line[2]), int(line[3]), int(line[4]), float(line[5]), float(line[6])
    assert (a, b, c, ncalls) == (0, 1, 2, 1)
    assert scale == 0.5 and total == 3 * sum(range(1000))


_PIGGY_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from pytorch_pose_proposal_network_amd import train as T
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 1000
store = torch.zeros(16 + n)
flat = store[16:]
flat.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
task = T.GradNormWeights("cpu", lr=0.01)
task.bind(torch.empty(0).set_(store.untyped_storage(), 0, (5,), (1,)))          # as trainer.PPNTrainer does
calls = []
def before_last():
    calls.append(len(ex.handles))                                               # buckets already out when it runs
    task.w.copy_(torch.tensor([1.0, 2.0, 0.5, 1.5, 0.25]) * (rank + 1))          # stands in for the local Adam step
ex = T.BucketedAllReduce(flat, bucket_elems=300, store=store, before_last=before_last)
ex.ready(650)
ex.ready(120)
scale = ex.finish()
if rank == 0:
    print("RESULT", calls, scale, float(flat.sum()), store[:16].tolist())
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 8])
def test_task_weights_ride_on_the_last_gradient_bucket(tmp_path, world):
    """The exchange step of the training path at world 2 and at world 8 (BASELINE configs[3]: 8 ranks) on CPU (gloo): the
    five GradNorm task weights live in the 16-float prefix of the gradient store, their local step runs right before the
    LAST bucket is issued (three buckets are already out), and that bucket carries them: SUM over the ranks with no
    collective of their own (/root/reference/main.py:769-771 is a separate all-reduce)."""
    script = tmp_path / "piggy_worker.py"
    script.write_text(_PIGGY_WORKER)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT")][0]
    calls, rest = eval(line[len("RESULT "):line.index("]") + 1]), line[line.index("]") + 2:].split(" ", 2)
    scale, total, prefix = float(rest[0]), float(rest[1]), eval(rest[2])
    tri = world * (world + 1) // 2
    assert calls == [3]                                        # once, with the three earlier buckets already issued
    assert scale == 1.0 / world and total == tri * sum(range(1000))
    assert prefix[:5] == [1.0 * tri, 2.0 * tri, 0.5 * tri, 1.5 * tri, 0.25 * tri] and prefix[5:] == [0.0] * 11


def test_ap_against_people_on_the_reference_fixture():
    """evaluate.ap_against_people (the task-metric account of the 16-bit modes, bench.py::ap_vs_reference): a people list
    scored against itself gives the metric's ceiling on that list (the same number whatever the order of the people);
    removing people lowers every AP; on a fixture without overlapping people the ceiling is 100."""
    import os
    import numpy as np
    from pytorch_pose_proposal_network_amd import evaluate as E
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_tuned_d22_384.npz"))
    nb = int(g["batch"])
    exp = [{k: g[f"{i}/{k}"] for k in ("n", "kp_cell", "limb_arg", "bbox", "score")} for i in range(nb)]
    ceil_ = np.array(E.ap_against_people(exp, exp))
    assert np.all(ceil_ > 85.0) and np.all(ceil_ <= 100.0)
    rev = [dict(n=e["n"], kp_cell=e["kp_cell"][::-1], limb_arg=e["limb_arg"][::-1], bbox=e["bbox"][::-1],
                score=e["score"][::-1]) for e in exp]
    assert np.allclose(E.ap_against_people(exp, rev), ceil_, atol=3.0)          # (ties in the matcher depend on order)
    thin = []
    for e in exp:
        keep = [i for i in range(int(e["n"])) if i % 2 == 0]
        thin.append(dict(n=len(keep), kp_cell=e["kp_cell"][keep], limb_arg=e["limb_arg"][keep], bbox=e["bbox"][keep],
                         score=e["score"][keep]))
    less = np.array(E.ap_against_people(exp, thin))
    assert np.all(less < ceil_ - 15.0)
    # two well separated single-keypoint-complete people: identical lists score 100
    one = dict(n=2, kp_cell=np.zeros((2, 18), np.int32), limb_arg=np.zeros((2, 17), np.int32),
               bbox=np.zeros((2, 18, 4), np.float32), score=np.full((2, 18), 0.9, np.float32))
    for p, (cy, cx) in enumerate(((60.0, 60.0), (300.0, 300.0))):
        for k in range(18):
            one["bbox"][p, k] = [cy + k - 10, cx + 2 * k - 10, cy + k + 10, cx + 2 * k + 10]
        one["bbox"][p, 0] = [cy - 40, cx - 40, cy + 40, cx + 40]
    assert np.allclose(E.ap_against_people([one], [one]), 100.0)


def test_per_launch_dtype_policies_of_the_inference_modes():
    """Host logic of the mixed-precision inference modes (no GPU: constructing a PoseProposalNet only lowers the program):
    which launches run in which type.  bf16 mode: stem + layer3-4 in IEEE half (half_prefix=4), everything else bf16;
    float16x3: f32 where cin < 64, split-f16 elsewhere; float16 with exact_prefix=3: stem + layer3 as in float16x3, half
    from layer4 on, no fused shortcut inside the prefix."""
    from pytorch_pose_proposal_network_amd import lib as L, model
    m = model.PoseProposalNet("drn_d_22", compute_dtype="bfloat16")
    kinds = {o.name: m._op_dtype(o) for o in m._ops}
    assert m.half_prefix == 4 and m.stem_dtype == L.PPN_F16
    assert all(v == L.PPN_F16 for k, v in kinds.items() if k.startswith(("backbone.0", "backbone.3.", "backbone.4.")))
    assert all(v == L.PPN_BF16 for k, v in kinds.items() if not k.startswith(("backbone.0", "backbone.3.", "backbone.4.")))
    pure = model.PoseProposalNet("drn_d_22", compute_dtype="bfloat16", stem_dtype="bfloat16", half_prefix=-1)
    assert {pure._op_dtype(o) for o in pure._ops} == {L.PPN_BF16}
    with pytest.raises(ValueError):
        model.PoseProposalNet("drn_d_22", compute_dtype="bfloat16", stem_dtype="bfloat16", half_prefix=4)
    x3 = model.PoseProposalNet("drn_d_22", compute_dtype="float16x3")
    for o in x3._ops:
        want = L.PPN_F16X3 if (o.k != 7 and o.cin % 64 == 0) else L.PPN_F32
        assert x3._op_dtype(o) == want, o.name
    assert not any(o.ds_src for o in x3._ops)                         # no fused shortcut in split launches
    xp = model.PoseProposalNet("drn_d_22", compute_dtype="float16", exact_prefix=3)
    kinds = {o.name: xp._op_dtype(o) for o in xp._ops}
    assert kinds["backbone.0.0"] == L.PPN_F32 and kinds["backbone.3.0.conv1"] == L.PPN_F32
    assert kinds["backbone.3.0.conv2"] == kinds["backbone.3.1.conv2"] == L.PPN_F16X3
    assert all(v == L.PPN_F16 for k, v in kinds.items() if not k.startswith(("backbone.0", "backbone.1", "backbone.2", "backbone.3.")))
    assert any(o.ds_src for o in xp._ops if o.name.startswith("backbone.4."))      # the trunk keeps its fused shortcuts
    with pytest.raises(ValueError):
        model.PoseProposalNet("drn_d_22", compute_dtype="bfloat16", exact_prefix=3)
    # Bottleneck trunks lower the same way
    d54 = model.PoseProposalNet("drn_d_54", compute_dtype="bfloat16")
    assert d54._op_dtype(d54._ops[-1]) == L.PPN_BF16 and d54._op_dtype(d54._ops[0]) == L.PPN_F16


_PER_RANK_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
r = bench._per_rank_stats(dist, "gloo", torch.device("cpu"), world, 1000.0 * (rank + 1), 0.25 * (rank + 1))
if rank == 0:
    print("RESULT", r)
dist.barrier(); dist.destroy_process_group()
'''


def test_per_rank_statistics_of_the_bench_line(tmp_path):
    """bench.py N > 1 (round 5): every rank's own rate and one auxiliary per-rank number (training: the exposed all-reduce time)
    are gathered on all ranks, so a straggler or a rank that fell back is visible -- rehearsed with 3 gloo ranks on CPU
    (/root/reference/main.py:240-245,769-771 is the DDP set-up whose exchange this instruments)."""
    script = tmp_path / "worker.py"
    script.write_text(_PER_RANK_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="3")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(3)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    r = eval([l for l in outs[0].splitlines() if l.startswith("RESULT")][0][len("RESULT "):])
    assert r["rates"]["ranks"] == [1000.0, 2000.0, 3000.0] and (r["rates"]["min"], r["rates"]["median"], r["rates"]["max"]) == (1000.0, 2000.0, 3000.0)
    assert r["aux"]["ranks"] == [0.25, 0.5, 0.75] and r["aux"]["max"] == 0.75


def test_mfma_busy_constants_resolve_for_the_dominant_kernel():
    """roofline.mfma_busy / frac_of_held_clock_peak (round 5) are committed constants with provenance: the newest
    profiles/r*_mfma_busy.json must hold the benchmarked kernel and a held clock."""
    import bench
    r = bench.mfma_busy("conv_igemm_big_kernel<__bf16, 192, 256, 8, false>", 1250.0, 2500.0)
    assert 0.3 < r["mfma_busy"] < 0.9 and 0.2 < r["mfma_busy_conv_stack"] < 0.9
    assert 1.5 < r["held_clock_ghz"] < 2.5 and abs(r["frac_of_held_clock_peak"] - 0.5 * 2.4 / r["held_clock_ghz"]) < 1e-3
    assert "tools/pmc_mfma.sh" in r["mfma_busy_provenance"]
