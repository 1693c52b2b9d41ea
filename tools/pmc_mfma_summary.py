"""Per-kernel MFMA-busy share from tools/pmc_mfma.sh (one SQ + GRBM pass over bench.py --lanes 1 --shared-plan).

MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMD cycles of the dispatch), SIMD cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8
XCDs) x 1024 SIMDs (256 CUs x 4).  SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles (MI355X_MICROARCH.md, cycle constants); a
16x16x32 bf16 MFMA holds its pipe 16 cycles, so at 100 % busy the chip would run 2 x 16*16*32 x 1024 / 16 FLOP per cycle =
2.5 PFLOP/s at 2.4 GHz.  The held clock is GRBM_GUI_ACTIVE / 8 / (dispatch wall time) (reads high on dispatches < 0.3 ms).

    python tools/pmc_mfma_summary.py <dir with */*counter_collection.csv> out.txt out.json
"""
import collections, csv, glob, json, os, sys, time

src, out_txt, out_json = sys.argv[1], sys.argv[2], sys.argv[3]
f = max(glob.glob(f"{src}/*/*counter_collection.csv"), key=os.path.getmtime)
disp = collections.OrderedDict()                    # dispatch id -> {name, counters, t0, t1}
for r in csv.DictReader(open(f)):
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
SIMDS = 1024
per = collections.OrderedDict()
for d in disp.values():
    if "GRBM_GUI_ACTIVE" not in d or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
        continue
    p = per.setdefault(d["name"], collections.defaultdict(float))
    p["n"] += 1
    p["ns"] += d["t1"] - d["t0"]
    for k, v in d.items():
        if k not in ("name", "t0", "t1"):
            p[k] += v


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n[:78]


rows = []
conv = collections.defaultdict(float)
for name, p in per.items():
    cyc = p["GRBM_GUI_ACTIVE"] / 8.0
    busy = p["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS) if cyc else 0.0
    clk = cyc / p["ns"] if p["ns"] else 0.0          # GHz
    wave = p.get("SQ_WAVE_CYCLES", 0.0)
    rows.append({"kernel": name, "launches": int(p["n"]), "avg_us": p["ns"] / p["n"] / 1e3, "mfma_busy": busy, "clock_ghz": clk,
                 "wait_any": p.get("SQ_WAIT_ANY", 0.0) / wave if wave else None,
                 "wait_inst": p.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else None,
                 "issuing": p.get("SQ_ACTIVE_INST_ANY", 0.0) / wave if wave else None, "total_us": p["ns"] / 1e3})
    is_conv = any(s in name for s in ("conv_igemm", "stem012", "head_limb_argmax", "conv64", "stem7x7", "stem3x3"))
    if is_conv:
        conv["mfma"] += p["SQ_VALU_MFMA_BUSY_CYCLES"]; conv["cyc"] += cyc; conv["ns"] += p["ns"]
rows.sort(key=lambda r: -r["total_us"])
lines = [f"# MFMA busy by counter, from {os.path.relpath(f, src)} ({time.strftime('%Y-%m-%d')}); bench.py --lanes 1 --shared-plan (the multi-lane plan's kernels, one launch in flight)",
         "# busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / dispatch time (reads high on short dispatches)",
         f"{'kernel':78s} {'n':>5s} {'avg us':>8s} {'MFMA busy':>9s} {'clock':>6s} {'issuing':>7s} {'wait_inst':>9s} {'wait_any':>8s}"]
for r in rows:
    if r["total_us"] < 1.0:
        continue
    f3 = lambda v: "   -  " if v is None else f"{100 * v:5.1f}%"
    lines.append(f"{short(r['kernel']):78s} {r['launches']:5d} {r['avg_us']:8.1f} {100 * r['mfma_busy']:8.1f}% {r['clock_ghz']:6.2f} {f3(r['issuing']):>7s} {f3(r['wait_inst']):>9s} {f3(r['wait_any']):>8s}")
stack = conv["mfma"] / (conv["cyc"] * SIMDS) if conv["cyc"] else 0.0
dom = next((r for r in rows if "conv_igemm_big_kernel" in r["kernel"] and "Li192ELi256E" in r["kernel"]), rows[0] if rows else None)
lines.append(f"# conv stack (stem + every convolution + head), time-weighted: MFMA busy {100 * stack:.1f} % of SIMD cycles; mean clock {conv['cyc'] / conv['ns'] if conv['ns'] else 0:.3f} GHz")
open(out_txt, "w").write("\n".join(lines) + "\n")
json.dump({"_date": time.strftime("%Y-%m-%d"), "conv_stack_mfma_busy": round(stack, 4),
           "conv_stack_clock_ghz": round(conv["cyc"] / conv["ns"], 4) if conv["ns"] else None,
           "dominant_kernel": dom["kernel"] if dom else None, "dominant_mfma_busy": round(dom["mfma_busy"], 4) if dom else None,
           "dominant_clock_ghz": round(dom["clock_ghz"], 4) if dom else None,
           "kernels": {r["kernel"]: {"launches": r["launches"], "avg_us": round(r["avg_us"], 2), "mfma_busy": round(r["mfma_busy"], 4),
                                     "clock_ghz": round(r["clock_ghz"], 3)} for r in rows if r["total_us"] >= 1.0}},
          open(out_json, "w"), indent=1)
print("\n".join(lines))
