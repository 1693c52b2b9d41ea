"""GPU: `python bench.py --gpus 2` as the driver starts it WITHOUT a launcher -- bench.py starts its own ranks as a child
`torch.distributed.run` (before this process touches the GPU, never exec) -- rehearsed on the one GPU of the box with
PPN_BENCH_BACKEND=gloo (both ranks share cuda:0; the driver's multi-GPU runs use "nccl" = RCCL, same code otherwise;
/root/reference/main.py:240-245,289,769-771,1233-1238 is the DDP set-up this replaces).  Checks the contract of the N>1
line and that a failing rank fails the run instead of hanging it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=420, gpus=2):
    env = dict(os.environ, PPN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("OMP_NUM_THREADS", None)                            # bench.py's self-launch divides the cores itself
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--windows", "2",
           "--no-extras", "--no-cpu-baseline"] + extra
    # own session: on a timeout the WHOLE group (bench.py, torch.distributed.run, the ranks) is killed, so that no rank is left
    # holding the GPU for the tests that follow; the timeout stays well below pytest's per-test limit (tools/gpu_round.sh: 900 s)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)
        proc.communicate()
        raise
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)


def test_self_launched_two_rank_inference_line():
    p = _run([])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 0 and r["scaling"] == "weak" and r["steps"] == 3
    assert r["batch_consistency"]["ok"] and r["batch_consistency"]["frames"] == 32
    assert r["value"] > 0 and r["value_windows"]["n"] == 2
    assert r["value_windows"]["min"] <= r["value_windows"]["median"] <= r["value_windows"]["max"]
    assert "cpu_baseline" not in r                              # rank 0 at N=1 only


def test_self_launched_two_rank_training_line():
    p = _run(["--workload", "train", "--batch", "4"])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 0 and r["value"] > 0
    assert len(r["losses"]) == 5 and all(v == v for v in r["losses"])         # finite, not NaN
    assert abs(sum(r["task_weights"]) - 5.0) < 1e-2                              # renormalised to sum 5 (main.py:773-777)


@pytest.mark.parametrize("workload", ["inference", "train"])
def test_self_launched_four_rank_rehearsal(workload):
    """A wide N > 1 rehearsal on the one GPU of the box: `python bench.py --gpus 4` (the pool's process guard admits at most
    SIX processes on a card and this pytest process holds the GPU too; four ranks leave one slot of margin -- round 5: with five
    the guard counted a seventh process and killed the run -- so the driver's N = 8 cannot be started here; world 8 itself is
    rehearsed on CPU tensors by tests/test_host_cpu.py::test_task_weights_ride_on_the_last_gradient_bucket).
    Four ranks over gloo share cuda:0, four frames each: one JSON line, the self-launch gave each rank cpu_count // 4 OpenMP threads,
    the per-rank host submit time is in the line; training: finite losses, task weights renormalised to sum 5 after they
    rode on the last gradient bucket (/root/reference/main.py:240-245,289,769-771,1233-1238)."""
    p = _run(["--workload", workload, "--batch", "4"], gpus=4, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 4 and r["rccl_ranks"] == 0 and r["scaling"] == "weak" and r["steps"] == 3 and r["value"] > 0
    assert r["host"]["omp_num_threads"] == str(max(1, (os.cpu_count() or 8) // 4))
    assert 0 < r["host"]["submit_ms_per_step_max_over_ranks"] < 1e4
    # round 5: every rank's own rate (min / median / max) is in the N > 1 line, so a straggler or a rank that fell back shows
    assert len(r["per_rank"]["ranks"]) == 4 and 0 < r["per_rank"]["min"] <= r["per_rank"]["median"] <= r["per_rank"]["max"]
    if workload == "train":
        assert len(r["losses"]) == 5 and all(v == v for v in r["losses"])
        assert abs(sum(r["task_weights"]) - 5.0) < 1e-2
        assert len(r["allreduce_exposed_ms_per_step"]["ranks"]) == 4
    else:
        assert r["config"]["frames_per_gpu"] == 4 and r["batch_consistency"]["ok"]


@pytest.mark.parametrize("workload", ["inference", "train"])
def test_a_failing_rank_fails_the_run_without_hanging(workload):
    p = _run(["--workload", workload, "--batch", "4"], {"PPN_BENCH_FAIL_RANK": "1"}, timeout=300)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.lstrip().startswith("{")], p.stdout
    assert "PPN_BENCH_FAIL_RANK" in p.stderr


@pytest.mark.parametrize("workload", ["inference", "train"])
def test_rccl_branch_runs_with_one_rank(workload):
    """The `nccl` (= RCCL) calls of the N > 1 path -- init_process_group(device_id=...), barrier, the MAX all-reduce of the
    timing, and in training the bucketed asynchronous gradient all-reduce against side-stream weight gradients plus the
    task-weight all-reduce -- executed for real on the one GPU of the box: bench.py under torch.distributed.run with ONE
    process, PPN_BENCH_FORCE_DIST=1 (join the group although world = 1) and PPN_FORCE_DP=1 (run the exchange although a
    SUM over one rank is the identity).  Two ranks cannot share a GPU under RCCL, so this is as far as one GPU goes."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PPN_BENCH_FORCE_DIST="1", PPN_FORCE_DP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PPN_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--windows", "2", "--no-extras", "--no-cpu-baseline", "--workload", workload]
    if workload == "train":
        cmd += ["--batch", "8"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["rccl_ranks"] == 1 and r["value"] > 0
    if workload == "train":
        assert len(r["losses"]) == 5 and all(v == v for v in r["losses"])
        assert abs(sum(r["task_weights"]) - 5.0) < 1e-2
    else:
        assert r["batch_consistency"]["ok"]
