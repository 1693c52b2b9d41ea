"""Host-side timeline of PPNTrainer.train_step: how long the Python thread spends ENQUEUEING each phase (not how long
the GPU takes).  Where a phase's host time exceeds its GPU time the GPU waits for launches; a long host time in a phase
with few launches is a host sync (tolist / event.synchronize)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import lib as L, synth, prng, targets
from pytorch_pose_proposal_network_amd.trainer import PPNTrainer

dev = torch.device("cuda")
tr = PPNTrainer("drn_d_22", synth.make_state_dict("drn_d_22", 0), compute_dtype=L.PPN_BF16, insize=(384, 384))
x = torch.from_numpy(synth.normalized_frames(prng.u8_frames(99, 32, (384, 384)))).to(dev)
tg = targets.synthetic_targets(99, 32, (384, 384), device=dev)
log = []


def wrap(obj, name, label=None):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        log.append((label or name, t0, time.perf_counter()))
        return r
    setattr(obj, name, g)


wrap(tr, "forward"); wrap(tr.criterion, "forward_backward", "loss"); wrap(tr.criterion, "forward_backward_dz", "loss+dz"); wrap(tr, "_second_order_tail", "so_tail")
wrap(tr, "_limb_probe", "limb_probe(sync)"); wrap(tr, "backward"); wrap(tr.task, "step", "task.step"); wrap(tr.opt, "step", "adam")
wrap(tr.task, "host_weights", "host_weights(sync)"); wrap(tr, "probe_grad")
# GPU-side stamps (no profiler): end of the loss on the main stream, first / last probe launch on the probe stream,
# start of the second-order tail on the main stream
gpu = {}
_fb = tr.criterion.forward_backward
def fb(*a, **k):
    r = _fb(*a, **k)
    if k.get("coeff_dev") is not None:
        e = torch.cuda.Event(enable_timing=True); e.record(); gpu.setdefault("loss_end", []).append(e)
    return r
tr.criterion.forward_backward = fb
_fbz = tr.criterion.forward_backward_dz
def fbz(*a, **k):
    r = _fbz(*a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); gpu.setdefault("loss_end", []).append(e)
    return r
tr.criterion.forward_backward_dz = fbz
_ub = tr.criterion.unary_backward
def ub(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); gpu.setdefault("probe", []).append(e)
    return _ub(*a, **k)
tr.criterion.unary_backward = ub
_st = tr._second_order_tail
def st(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream()); gpu.setdefault("tail", []).append(e)
    r = _st(*a, **k)
    e2 = torch.cuda.Event(enable_timing=True); e2.record(); gpu.setdefault("tail_end", []).append(e2)
    return r
tr._second_order_tail = st
from pytorch_pose_proposal_network_amd import train as _T
_ps = _T.probe_stats
def ps(*a, **k):
    # the main stream has just waited for the side (weight-gradient) and probe streams: how long did it sit there?
    e = torch.cuda.Event(enable_timing=True); e.record(); gpu.setdefault("joined", []).append(e)
    return _ps(*a, **k)
_T.probe_stats = ps
for _ in range(3):
    tr.train_step(x, tg)
torch.cuda.synchronize()
log.clear(); gpu.clear()
T0 = time.perf_counter()
marks = []
for it in range(3):
    marks.append(time.perf_counter())
    tr.train_step(x, tg)
tend = time.perf_counter()
torch.cuda.synchronize()
tsync = time.perf_counter()
print(f"3 steps: host returned after {(tend - T0) * 1e3:.2f} ms, GPU done after {(tsync - T0) * 1e3:.2f} ms")
for name, a, b in log:
    if name == "probe_grad":
        continue
    print(f"  {name:22s} start {(a - T0) * 1e3:8.2f} ms   host time {(b - a) * 1e3:7.2f} ms")
pg = [(b - a) for n, a, b in log if n == "probe_grad"]
print(f"  probe_grad x{len(pg)}: {sum(pg) * 1e3 / max(1, len(pg)):.2f} ms of host time each")

for i in range(3):
    le = gpu["loss_end"][i]
    pr = gpu["probe"][4 * i:4 * i + 4]
    print(f"iteration {i}: GPU time after the end of the loss: first probe starts +{le.elapsed_time(pr[0]):.3f} ms, fourth probe starts "
          f"+{le.elapsed_time(pr[3]):.3f}, main stream reaches the second-order tail +{le.elapsed_time(gpu['tail'][i]):.3f}, "
          f"side + probe streams joined +{le.elapsed_time(gpu['joined'][i]):.3f}, tail done +{le.elapsed_time(gpu['tail_end'][i]):.3f}")
