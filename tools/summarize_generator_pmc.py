#!/usr/bin/env python3
"""MFMA pipe utilisation of the whole GlobalGenerator forward + backward from one rocprofv3 --pmc pass over tools/pmc_generator.py:
the dispatches behind the LAST marker launch (mdct4_*), summed:

    busy = sum_k SQ_VALU_MFMA_BUSY_CYCLES_k / 1024 SIMDs   /   sum_k GRBM_GUI_ACTIVE_k / 8 XCDs

(the same ratio tools/summarize_profile.py --derive reports per kernel; every kernel of the pass is in the denominator, the
InstanceNorm / pack / unpack launches with zero MFMA cycles included; kernels run one at a time under --pmc).

    python tools/summarize_generator_pmc.py gpurun_out/<tag>g/runc/*_counter_collection.csv profiles/<tag>_generator_mfma_busy.json
"""
import collections
import csv
import json
import sys


def short(n):
    for a in ("_ZN12_GLOBAL__N_1", "(anonymous namespace)::"):
        n = n.replace(a, "")
    return n.split("(")[0][:70]


def main(path, out):
    disp = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": collections.defaultdict(float)})
            d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(disp)
    marks = [i for i in ids if "mdct4" in disp[i]["name"]]
    assert marks, "no marker launch in the counter file"
    seg = [i for i in ids if i > marks[-1]]
    busy = sum(disp[i]["c"]["SQ_VALU_MFMA_BUSY_CYCLES"] for i in seg) / 1024.0
    act = sum(disp[i]["c"]["GRBM_GUI_ACTIVE"] for i in seg) / 8.0
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for i in seg:
        k = fam[short(disp[i]["name"])]
        k[0] += 1; k[1] += disp[i]["c"]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0; k[2] += disp[i]["c"]["GRBM_GUI_ACTIVE"] / 8.0
    top = sorted(fam.items(), key=lambda kv: -kv[1][2])[:14]
    res = {"what": "MFMA pipe utilisation of GlobalGenerator forward + backward (configs[1], B = 32, bf16), hardware counters",
           "how": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/pmc_generator.py; "
                  "dispatches behind the last marker launch; tools/summarize_generator_pmc.py",
           "launches": len(seg),
           "mfma_busy_cycles_per_SIMD": busy, "kernel_cycles_per_XCD": act,
           "mfma_pipe_busy_fraction": busy / act,
           "note": "time-weighted over EVERY kernel of the pass (InstanceNorm, pack, unpack and the other non-MFMA launches count with zero busy "
                   "cycles); under --pmc kernels run one at a time, so GRBM_GUI_ACTIVE includes each launch's ramp; this is the pipe-busy "
                   "reading of 'MFMA utilisation' -- the FLOP reading (6 M_G FLOP / time / 2.5 PF) is bench.py's config.G_fwd_bwd_frac_of_bf16_peak",
           "by_kernel_family": [{"kernel": k, "launches": v[0], "share_of_cycles": v[2] / act, "mfma_pipe_busy_fraction": (v[1] / v[2]) if v[2] else 0.0} for k, v in top]}
    json.dump(res, open(out, "w"), indent=1)
    print(f"{len(seg)} launches, MFMA pipe busy {busy / act:.3f} of {act / 1e6:.2f} M cycles per XCD -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
