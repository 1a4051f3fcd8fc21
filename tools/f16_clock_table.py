#!/usr/bin/env python3
"""Table for tools/collect_f16_clock.sh: per kernel, for the bf16 and the fp16 build, the average launch duration (kernel trace)
and the counters per launch; cycles / duration = the clock the kernel ran at."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
for dt in ("bf16", "f16"):
    dur, cnt = defaultdict(list), defaultdict(lambda: defaultdict(list))
    for p in glob.glob(f"{root}/{dt}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for p in glob.glob(f"{root}/{dt}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {dt}")
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        if not any(t in k for t in ("gconv_kernel", "wgrad_kernel", "march_")) or "thin" in k:
            continue
        d = dur[k][2:] or dur[k]
        us = sum(d) / len(d) / 1e3
        c = {n: sum(v[2:] or v) / len(v[2:] or v) for n, v in cnt[k].items()}
        gui = c.get("GRBM_GUI_ACTIVE", 0) / 8
        print(f"{k[:70]:70s} {len(dur[k]):3d} launches {us:8.1f} us | GUI_ACTIVE/XCD {gui:10.0f} cyc -> {gui / us / 1e3 if us else 0:5.2f} GHz | "
              f"MFMA busy/SIMD {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024:10.0f} | SQ busy {c.get('SQ_BUSY_CYCLES', 0):12.0f}")
