#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one round into the tracked evidence files under profiles/:
  --trace  kernel_trace.csv of a bench run   -> per-kernel, per-grid launch-duration summary (median / mean / n)
  --pmc    counter_collection.csv files      -> per-kernel counter means (FETCH_SIZE x2 correction for gfx950 applied here)"""
import argparse, csv, collections, json, statistics, sys

def short(name):
    n = name
    for a, b in (("_ZN12_GLOBAL__N_1", ""), ("(anonymous namespace)::", "")):
        n = n.replace(a, b)
    return n[:110]

def trace(path, out):
    rows = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
            rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    items = sorted(rows.items(), key=lambda kv: -sum(kv[1]))
    with open(out, "w") as f:
        f.write("kernel,grid_x,grid_y,wg_x,launches,median_us,mean_us,min_us,total_ms\n")
        for (k, gx, gy, wx), v in items[:80]:
            f.write(f"\"{k}\",{gx},{gy},{wx},{len(v)},{statistics.median(v):.1f},{statistics.mean(v):.1f},{min(v):.1f},{sum(v)/1e3:.2f}\n")
    print("wrote", out)

def pmc(paths, out):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in acc.items():
        if not any(t in k for t in ("gconv_kernel", "wgrad_kernel")):
            continue
        res[k] = {c: {"mean": statistics.mean(v), "n": len(v)} for c, v in cs.items()}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace"); ap.add_argument("--pmc", nargs="*"); ap.add_argument("--out", required=True)
    a = ap.parse_args()
    if a.trace: trace(a.trace, a.out)
    else: pmc(a.pmc, a.out)
