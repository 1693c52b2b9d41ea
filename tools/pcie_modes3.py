"""PCIe-inclusive slow mode vs the number of hardware queues HIP may use (GPU_MAX_HW_QUEUES, default 4)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(tag, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--windows", "10"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            r = json.loads(line); pc = r["pcie_inclusive"]
            print(f"{tag:28s} value {r['value']:9.1f} (median {r['value_windows']['median']:9.1f})  pcie {pc['value']:9.1f}  host {pc['host_ms_per_step']}", flush=True)
            return
    print(tag, "no JSON", p.returncode, flush=True)
for i in range(3):
    run(f"default #{i}")
    run(f"GPU_MAX_HW_QUEUES=8 #{i}", {"GPU_MAX_HW_QUEUES": "8"})
