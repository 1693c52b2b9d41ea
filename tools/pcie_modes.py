"""Why is `pcie_inclusive` bimodal across processes?  Runs bench.py several times as CHILD processes (fresh process =
fresh HSA queues / SDMA engine assignment) under different copy-engine settings and prints pcie_inclusive of each.

    python tools/pcie_modes.py [N_DEFAULT] [N_NOSDMA]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(env_extra, tag):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline", "--no-verify",
                        "--windows", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            r = json.loads(line)
            pc = r["pcie_inclusive"]
            print(f"{tag:28s} value {r['value']:9.1f}  pcie {pc['value']:9.1f} img/s  {pc['ms_per_step']:.3f} ms/step  "
                  f"host {pc['host_ms_per_step']}  h2d alone {pc['h2d_alone_gbps']} GB/s", flush=True)
            return
    print(f"{tag}: no JSON line (rc {p.returncode})", flush=True)


if __name__ == "__main__":
    n_def = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n_no = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for i in range(n_def):
        run({}, f"default #{i}")
    for i in range(n_no):
        run({"HSA_ENABLE_SDMA": "0"}, f"HSA_ENABLE_SDMA=0 #{i}")
