"""Which part of a full bench.py run puts the PCIe-inclusive loop into its slow mode?  Child processes, one variable each."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(flags, tag, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + flags, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            r = json.loads(line); pc = r["pcie_inclusive"]
            print(f"{tag:44s} value {r['value']:9.1f}  pcie {pc['value']:9.1f}  host {pc['host_ms_per_step']}", flush=True)
            return
    print(tag, "no JSON", p.returncode, flush=True)
run(["--no-extras", "--no-verify", "--windows", "1"], "no-verify, 1 window")
run(["--no-extras", "--no-verify", "--windows", "10"], "no-verify, 10 windows")
run(["--no-extras", "--windows", "1"], "verify, 1 window")
run(["--no-extras", "--windows", "10"], "verify, 10 windows")
run(["--no-extras", "--windows", "10"], "verify, 10 windows, PPN_PLAN_GRAPH=0", {"PPN_PLAN_GRAPH": "0"})
