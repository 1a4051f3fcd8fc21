"""Which torch operators still launch kernels inside one EAGER training step of configs[1] (the launches the HIP library
does not own: elementwise glue, fills, copies), with the source line that issued them.

    python tools/list_step_launches.py [batch]        (GPU)

Prints one line per (operator, input shapes, python frame of this package) with its launch count and device time.
"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.cuda.set_device(0)
    from pix2pixhdaudiosr_amd.models.models import create_model
    torch.manual_seed(1234)
    opt = bench.make_opt(B)
    model = create_model(opt)
    T = (bench.FRAMES - 1) * opt.hop_length
    hr = 0.1 * torch.randn(B, T, device="cuda")
    lr = 0.1 * torch.randn(B, T, device="cuda")
    for _ in range(2):
        model.train_step(lr, hr)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        model.train_step(lr, hr)
        torch.cuda.synchronize()
    rows = defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith("aten::") or not ev.kernels:
            continue
        # innermost aten op that owns kernels directly: skip parents whose children own the same kernels
        if any(c.name.startswith("aten::") and c.kernels for c in ev.cpu_children):
            continue
        frame = "?"
        for fr in ev.stack or []:
            if ("pix2pixhdaudiosr_amd" in fr or "bench.py" in fr) and "torch/" not in fr:
                frame = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
                break
        shapes = str(ev.input_shapes)[:60]
        key = (ev.name, shapes, frame[:110])
        rows[key][0] += len(ev.kernels)
        rows[key][1] += sum(k.duration for k in ev.kernels)
    tot_n = sum(v[0] for v in rows.values())
    tot_us = sum(v[1] for v in rows.values())
    print(f"# torch-operator kernel launches in one eager step (B={B}): {tot_n} launches, {tot_us:.0f} us")
    for (name, shapes, frame), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:4d} {us:9.1f} us  {name:28s} {shapes:60s} {frame}")


if __name__ == "__main__":
    main()
