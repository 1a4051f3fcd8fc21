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
        if not any(t in k for t in ("gconv_kernel", "wgrad_kernel", "march_")):
            continue
        res[k] = {c: {"mean": statistics.mean(v), "n": len(v)} for c, v in cs.items()}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)

LABELS = (("gconv_kernelIDF16bLi256ELi192ELi2ELi3ELi2ELi1", "gconv_kernel<bf16,256,192,2,3,2,HALO> trunk Conv3x3 768->768 @32x16 B=32 forward"),
          ("gconv_kernelIDF16bLi256ELi192ELi2ELi3ELi2ELi0", "gconv_kernel<bf16,256,192,2,3,2> (generic loop) trunk Conv3x3 768->768 @32x16 B=32 forward"),
          ("gconv_kernelIDF16bLi256ELi256ELi4ELi2ELi2ELi0", "gconv_kernel<bf16,256,256,4,2,2> discriminator Conv4x4 256->512 @65x33 B=32 forward"),
          ("wgrad_kernelIDF16bLi256", "wgrad_kernel<bf16,256> trunk weight gradient 768x6912, 16384 pixels"),
          ("march_s_kernel<48, 96, 64, false, false>", "march_s_kernel<48,96,64> Conv3x3 s2 48->96 @512x256 B=32 forward (marching, round 4)"),
          ("march_u_kernel<96, 48, 64, false, false>", "march_u_kernel<96,48,64> ConvTranspose3x3 s2 96->48 @256x128 B=32 forward (marching, round 4)"),
          ("dfirst_fwd_kernel", "dfirst_fwd_kernel<4> discriminator first layer Conv4x4 s2 4->64 @512x256 2B=64 forward + LeakyReLU (round 5; algorithmic 134 + 272 MB)"),
          ("dlast_fwd_partial_kernel", "dlast_fwd_partial_kernel<16> discriminator head Conv4x4 512->1 @66x34 2B=64 forward, pass 1 (round 5; algorithmic 147 MB read)"),
          ("dlast_dgrad_kernel", "dlast_dgrad_kernel discriminator head input gradient @66x34 2B=64 with addend and fused InstanceNorm-backward sums (round 5; algorithmic 147 MB written + 2 x 147 MB read)"))


def derive(paths, out, trunk_out):
    """The tracked JSONs behind bench.py's `roofline.traffic` and DESIGN.md's MFMA-busy figures, from the counter CSVs of
    tools/collect_profiles.sh (targets: tools/pmc_targets.py, ten launches per kernel).  Counter rows of one dispatch
    are summed (rocprofv3 reports one row per dispatch and counter here, already accumulated over XCDs / SEs)."""
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                for key, label in LABELS:
                    if key in r["Kernel_Name"]:
                        per[label][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    res = {"recorded": __import__("datetime").date.today().isoformat() + f" ({ROUND})",
           "how": "tools/collect_profiles.sh: rocprofv3 --kernel-trace --pmc <one block per pass> -- python3 tools/pmc_targets.py "
                  "(FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_* issue counters), "
                  "summarised by tools/summarize_profile.py --derive",
           "corrections": "gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> read bytes = 2 * FETCH_SIZE "
                          "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; requests served by the Infinity Cache are included (upper bound "
                          "on HBM traffic). mfma_pipe_busy_fraction = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs).",
           "kernels": {}}
    for _, label in LABELS:
        cs = per.get(label)
        if not cs:
            continue
        m = {c: statistics.mean(v.values()) for c, v in cs.items()}
        k = {"launches_averaged": max(len(v) for v in cs.values())}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            k["FETCH_SIZE_KB"] = m["FETCH_SIZE"]; k["WRITE_SIZE_KB"] = m["WRITE_SIZE"]
            k["memory_side_bytes_per_launch"] = int(2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024)
        if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            k["GRBM_GUI_ACTIVE_sum_over_8_XCDs"] = m["GRBM_GUI_ACTIVE"]
            k["kernel_cycles_per_XCD"] = m["GRBM_GUI_ACTIVE"] / 8
            k["SQ_VALU_MFMA_BUSY_CYCLES_sum_over_1024_SIMDs"] = m["SQ_VALU_MFMA_BUSY_CYCLES"]
            k["mfma_busy_cycles_per_SIMD"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
            k["mfma_pipe_busy_fraction"] = (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (m["GRBM_GUI_ACTIVE"] / 8)
        for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in m: k[c + "_quad"] = m[c]
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU_MFMA_MOPS_BF16"):
            if c in m: k[c] = m[c]
        res["kernels"][label] = k
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)
    t = res["kernels"].get(LABELS[0][1])
    if trunk_out and t and "memory_side_bytes_per_launch" in t:
        json.dump({"kernel": LABELS[0][1] + " (256 workgroups of 256x192)", "recorded": res["recorded"], "how": res["how"],
                   "FETCH_SIZE_KB_mean": t["FETCH_SIZE_KB"], "WRITE_SIZE_KB_mean": t["WRITE_SIZE_KB"],
                   "correction": "read bytes = 2 * FETCH_SIZE * 1024 on gfx950; WRITE_SIZE exact",
                   "hbm_bytes_per_launch": t["memory_side_bytes_per_launch"], "algorithmic_bytes_per_launch": 61000000,
                   "note": "memory-side requests, Infinity-Cache hits included: 25 MB of activations + the 10.6 MB packed weight matrix "
                           "once per XCD L2 (85 MB) + 9-tap halo re-reads that miss L2; far below the HBM roof: the kernel is MFMA / LDS "
                           f"bound (mfma pipe busy fraction in profiles/{ROUND_TAG}_mfma_pmc.json)"}, open(trunk_out, "w"), indent=1)
        print("wrote", trunk_out)


def traffic(fetch_csv, write_csv, steps, out):
    """Per-kernel memory-side bytes of one step from two --pmc passes (FETCH_SIZE, WRITE_SIZE) over the same eager run of
    `steps` step-equivalents -> markdown table (profiles/r02_step_pmc_traffic.md)."""
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])            # launches, fetch KB, write KB, ns
    for path, col in ((fetch_csv, 1), (write_csv, 2)):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                acc[k][col] += float(r["Counter_Value"])
                if col == 1:
                    acc[k][0] += 1
                    acc[k][3] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows = sorted(acc.items(), key=lambda kv: -(2 * kv[1][1] + kv[1][2]))
    tot_r = sum(2 * v[1] for v in acc.values()) * 1024 / steps / 1e9
    tot_w = sum(v[2] for v in acc.values()) * 1024 / steps / 1e9
    with open(out, "w") as f:
        f.write(f"# Memory-side traffic of one training step (configs[1], B=32, bf16), {ROUND}\n\n"
                "`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, in a second pass, `--pmc WRITE_SIZE` on `python3 bench.py --steps 1 "
                f"--warmup 1 --no-graph --no-probes --no-cpu-baseline --no-mdct` ({steps} step-equivalents per pass); FETCH_SIZE x2 (gfx950 counts 64 B per "
                "128-B request), WRITE_SIZE as is; Infinity-Cache hits included (upper bounds on HBM traffic); times are those of the "
                "instrumented FETCH pass.\n\n"
                f"**Whole step: {tot_r:.1f} GB read + {tot_w:.1f} GB written** (round 1: 86 + 22; round 2: 76.5 + 19.8; round 3: 73.5 + 19.7; round 4: 49.1 + 17.6).\n\n"
                "| kernel | launches/step | read GB/step | written GB/step | time ms/step |\n|---|---|---|---|---|\n")
        for k, v in rows[:40]:
            f.write(f"| `{k[-70:]}` | {v[0] / steps:.1f} | {2 * v[1] * 1024 / steps / 1e9:.2f} | {v[2] * 1024 / steps / 1e9:.2f} | {v[3] / steps / 1e6:.2f} |\n")
        fill = sum(v[2] for k, v in acc.items() if "FillFunctor<float>" in k) * 1024 / 1e9
        f.write(f"\nNot step work: the `FillFunctor<float>` row is the optimisers' zero-initialisation of their flat parameter / gradient / "
                f"moment buffers when the model is built ({fill:.2f} GB once per process, shown here divided by the {steps} step-equivalents); "
                "the step itself zeroes no gradient buffer (lazy `zero_grad`: the first weight gradient overwrites, optim.py).\n")
    print("wrote", out, f"read {tot_r:.1f} GB written {tot_w:.1f} GB per step")


def one_step(path, out):
    """Per-kernel time of ONE graph-replayed step of a `bench.py` kernel trace: the dispatches between the discriminator's
    Adam launch of one replay and the next (the last but one such interval: a timed step)."""
    import re
    with open(path) as f:
        rows = list(csv.DictReader(f))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_dev" in r["Kernel_Name"]]
    pairs = list(zip(adam[1::2], adam[3::2]))
    sizes = collections.Counter(b - a for a, b in pairs)
    n_common = sizes.most_common(1)[0][0]                          # the replayed steps all have the same launch count
    a, b = [p for p in pairs if p[1] - p[0] == n_common][-2]

    def fam(n):
        n = short(n).replace("void ", "")
        n = re.sub(r"^\d+", "", n)
        m = re.match(r"gconv_kernelI(DF16b|f|NS_5fp8_tE)Li(\d+)ELi(\d+)(?:ELi\d+ELi\d+ELi\d+ELi(\d+)E)?", n)
        if m: return f"gconv_kernel<{m.group(2)}x{m.group(3)}{' halo' if m.group(4) == '1' else ''}>"
        m = re.match(r"wgrad_kernelI(DF16b|f)Li(\d+)", n)
        if m: return f"wgrad_kernel<{m.group(2)}>"
        return re.sub(r"\(.*", "", re.sub(r"<.*", "", re.sub(r"I(DF16b|f).*", "", n)))[:44]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[a + 1:b + 1]:
        k = fam(r["Kernel_Name"])
        acc[k][0] += 1
        acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in acc.values())
    wall = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
    with open(out, "w") as f:
        f.write(f"# one graph-replayed step of bench.py under rocprofv3 --kernel-trace ({ROUND}): {b - a} kernel launches, "
                f"{tot / 1e3:.2f} ms of kernel time in {wall / 1e3:.2f} ms of wall time\n")
        f.write("kernel,launches,ms,percent\n")
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k},{v[0]},{v[1] / 1e3:.3f},{100 * v[1] / tot:.1f}\n")
    print("wrote", out, f"{b - a} launches {tot / 1e3:.2f} ms")


import os
ROUND_TAG = os.environ.get("ROUND_TAG", "r05")
ROUND = "round " + str(int(ROUND_TAG[1:]))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--one-step")
    ap.add_argument("--trace"); ap.add_argument("--pmc", nargs="*"); ap.add_argument("--derive", nargs="*")
    ap.add_argument("--trunk-out"); ap.add_argument("--traffic", nargs=2); ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    if a.one_step: one_step(a.one_step, a.out)
    elif a.trace: trace(a.trace, a.out)
    elif a.traffic: traffic(a.traffic[0], a.traffic[1], a.steps, a.out)
    elif a.derive: derive(a.derive, a.out, a.trunk_out)
    else: pmc(a.pmc, a.out)
