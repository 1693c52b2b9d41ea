"""GPU: the world > 1 branch of PPNTrainer.train_step (/root/reference/main.py:240-245, 289,
769-771, 1233-1238) executed for real: two ranks share the one GPU of the box and exchange over gloo (the driver's
multi-GPU runs use the same code with backend "nccl" = RCCL).  Expected values are assembled in THIS process from two
single-rank local passes: summed gradient buckets, Adam with the 1/world factor, task weights local-step -> SUM ->
/world -> clamp -> renormalise."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.parametrize("second_order,dtype,size", [(False, "f32", 96), (True, "f32", 96), (True, "bf16", 384)],
                         ids=["first_order", "second_order", "second_order_bf16_384"])
def test_two_ranks_equal_mean_of_single_rank_runs(tmp_path, second_order, dtype, size):
    sys.path.insert(0, HERE)
    import dp_worker as W
    port = _free_port()
    outs = [str(tmp_path / f"rank{r}.pt") for r in range(2)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), "2", port, outs[r],
                               "1" if second_order else "0", dtype, str(size)], env=env) for r in range(2)]
    try:
        rcs = [p.wait(timeout=600) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0, 0], rcs
    got = [torch.load(o) for o in outs]
    assert got[0]["world"] == 2
    # both ranks hold the same summed gradient, parameters and task weights after the step
    for k in ("grad", "flat", "w"):
        assert torch.equal(got[0][k], got[1][k]), k

    # ---- expected, from single-rank pieces in this process ------------------------------------------------------
    trs, local = [], []
    for r in range(2):
        tr = W.make_trainer(second_order, dtype, size)
        x, tg = W.shard_inputs(r, size)
        losses, gn, scale = tr.local_pass(torch.as_tensor(x).cuda(),
                                          {k: torch.from_numpy(v).cuda() for k, v in tg.items()})
        assert scale == 1.0
        tr.task.local_step(losses, gn, tr.base)
        trs.append(tr)
        local.append((losses.cpu(), tr.grad.clone(), tr.task.w.clone()))
    gsum = local[0][1] + local[1][1]
    gmax = float(gsum.abs().max())
    err = float((got[0]["grad"].cuda() - gsum).abs().max())
    print(f"summed gradient: max|dp - (g0+g1)| = {err:.3e} (max|g| {gmax:.3e})")
    assert err <= 1e-5 * gmax
    assert not torch.equal(local[0][1], local[1][1])                      # the shards really differ
    for r in range(2):
        assert torch.allclose(got[r]["losses"], local[r][0], rtol=1e-6)
    tr = trs[0]
    tr.task.w.copy_(local[0][2] + local[1][2])
    tr.task.renorm(2)
    assert torch.allclose(got[0]["w"].cuda(), tr.task.w, atol=1e-6), (got[0]["w"], tr.task.w)
    tr.grad.copy_(gsum)
    tr.opt.step(tr.grad, grad_scale=0.5)
    torch.cuda.synchronize()
    # Adam's first step is lr * g / (|g| + 1e-8): elements with a vanishing gradient amplify rounding differences of
    # the sum, so they are compared only where |g| is resolvable
    diff = (got[0]["flat"].cuda() - tr.flat).abs()
    mask = gsum.abs() > 1e-5
    print(f"parameters after Adam: bitwise equal {bool(torch.equal(got[0]['flat'].cuda(), tr.flat))}, "
          f"max|dp - expected| = {float(diff.max()):.3e} ({float(diff[mask].max()):.3e} where |g| > 1e-5)")
    assert float(mask.float().mean()) > 0.5 and float(diff[mask].max()) <= 5e-6
    assert float(diff.max()) <= 2 * 7e-4 * 1.001
