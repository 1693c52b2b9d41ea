"""Diagnostic: why do the plain 512 -> 512 3x3 launches of the benchmarked step take 282-302 us IN
SEQUENCE (inside the forward pass) but 256 us BACK TO BACK?  Runs the real batch-32 bf16 plan with a -DPPN_CLOCK=2 build of
conv_big.hip (python tools/build_variant.py clock2 conv_big.hip -DPPN_CLOCK=2): every workgroup of the chosen launches
records when it entered and left (chip-wide 100 MHz clock) and the cycles of its prologue / K loop / epilogue.  The same
launches are then repeated back to back (ppn_plan_run_timed with repeats) and the two pictures are printed side by side:
launch span, workgroups per round, per-round phase cycles, shader clock, ramp and tail.

    PPN_LIB=tools/bin/libppn_clock2.so python tools/clock_conv_seq.py [--lanes-flags] [--targets a,b]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PPN_LIB", os.path.join(ROOT, "tools", "bin", "libppn_clock2.so"))
import numpy as np
import torch
from pytorch_pose_proposal_network_amd import drn, lib as L, model as M, prng, synth

targets = ["backbone.6.1.conv1", "backbone.7.0"]
for i, a in enumerate(sys.argv):
    if a == "--targets":
        targets = sys.argv[i + 1].split(",")
flags = (L.PPN_CONV_NO_FILTER_BANK | L.PPN_CONV_SHARED_GPU) if "--lanes-flags" in sys.argv else 0
B, S = 32, 384
dev = torch.device("cuda")
net = M.PoseProposalNet(drn.drn_d_22(), insize=(S, S), outsize=(S // 16, S // 16), compute_dtype="bfloat16").cuda()
net.load_state_dict(synth.make_state_dict("drn_d_22", 0))
dbg = {}
for t in targets:                                  # plain launches: scale2 NULL, shift2 = the diagnostic channel
    assert (t + ".s2") not in net._dev, t + " has a second output"
    dbg[t] = net._dev[t + ".b2"] = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
frames = torch.from_numpy(prng.u8_frames(1, B, (S, S))).to(dev)
os.environ["PPN_PLAN_GRAPH"] = os.environ.get("PPN_PLAN_GRAPH", "1")


def table(tag, d, wall_us=None):
    t = d.cpu().numpy().reshape(-1, 8, 8)          # [workgroup][wave][slot]
    t = t[t[:, 0, 1] > 0]
    w0 = t[:, 0, :]                                 # wave 0 of every workgroup
    ent, ext = w0[:, 5].astype(np.float64) / 100.0, t[:, :, 6].max(1).astype(np.float64) / 100.0   # us
    t0 = ent.min()
    ent -= t0; ext -= t0
    clk = np.median(w0[:, 0] / w0[:, 1]) * 100e6
    order = np.argsort(ent)
    n = len(order)
    rounds = [order[i:i + 256] for i in range(0, n, 256)]
    print(f"--- {tag}: {n} workgroups, span {ext.max():.1f} us" + (f" (HIP events: {wall_us:.1f} us)" if wall_us else "") +
          f", shader clock {clk / 1e9:.3f} GHz")
    for r, idx in enumerate(rounds):
        print(f"  round {r}: {len(idx):3d} wgs  enter {ent[idx].min():7.1f} .. {ent[idx].max():7.1f} us (median {np.median(ent[idx]):7.1f})  "
              f"leave {ext[idx].min():7.1f} .. {ext[idx].max():7.1f} (median {np.median(ext[idx]):7.1f})  "
              f"life {np.median(ext[idx] - ent[idx]):6.1f} us | prologue {np.median(w0[idx, 2]):6.0f}  K loop {np.median(w0[idx, 0]):7.0f}  "
              f"epilogue {np.median(w0[idx, 4]):6.0f} cycles | clock {np.median(w0[idx, 0] / w0[idx, 1]) * 0.1:.3f} GHz")
    return ext.max()


def timed(repeats):
    return {n: ms * 1e3 for n, _, ms, _ in net.profile_layers(frames, True, repeats=repeats, fused_decode=True, conv_flags=flags)}


# warm: sustained load, as the benchmark runs it
st = torch.cuda.current_stream()
import time
t_end = time.time() + 2.5
while time.time() < t_end:
    for _ in range(20):
        net.forward_u8(frames, fused_decode=True, conv_flags=flags)
    torch.cuda.synchronize()
seq = timed(1)                                      # events around every launch of ONE pass, in plan order
for _ in range(10):
    net.forward_u8(frames, fused_decode=True, conv_flags=flags)
torch.cuda.synchronize()
snap = {t: dbg[t].clone() for t in targets}
for t in targets:
    table(f"{t} IN SEQUENCE", snap[t], seq.get(t))
b2b = timed(20)                                     # 20 launches of each op back to back between its events
for t in targets:
    table(f"{t} BACK TO BACK (last of 20)", dbg[t], b2b.get(t))
print("per-launch HIP-event durations (us), in sequence / back to back:")
for n in seq:
    print(f"  {n:36s} {seq[n]:8.1f} {b2b[n]:8.1f}")
