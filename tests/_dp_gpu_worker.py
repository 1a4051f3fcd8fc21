"""Worker of tests/test_gpu_dp.py: one data-parallel rank of the PRODUCT step (HIP kernels, captured graphs, flat
gradient all-reduce) on cuda:0; the ranks talk over gloo so that two of them can share the one GPU of a test box."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]                       # mode: graphed | eager | graphed_wire_bf16
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from test_gpu_model import _model
    from pix2pixhdaudiosr_amd import parallel_state as PS
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_step.npz"))
    model = _model(g, mask=False)                                   # same initial weights on every rank, no mask noise
    PS.enable_data_parallel(model, world, wire_dtype=torch.bfloat16 if mode.endswith("wire_bf16") else torch.float32)
    gen = torch.Generator().manual_seed(77)
    B = 2 * world
    # every shard holds the same two clips (in a different order): the spectrogram normalisation uses the min / max of
    # the tensor a rank sees (like the reference's DataParallel replicas), so only then is "single process on the whole
    # batch" the exact counterpart of the data-parallel run
    hr0 = 0.1 * torch.randn(2, g["hr"].shape[1], generator=gen)
    lr0 = 0.1 * torch.randn(2, g["hr"].shape[1], generator=gen)
    hr_all = torch.cat([hr0 if r % 2 == 0 else hr0.flip(0) for r in range(world)])
    lr_all = torch.cat([lr0 if r % 2 == 0 else lr0.flip(0) for r in range(world)])
    a, b = PS.shard_batch(B, rank, world)
    hr, lr = hr_all[a:b].cuda(), lr_all[a:b].cuda()
    step = model.train_step_graphed if mode.startswith("graphed") else model.train_step
    losses, first = [], None
    for i in range(5):
        ld = step(lr, hr)
        losses.append({k: float(v) for k, v in ld.items()})
        if i == 0:
            first = {"G": model.optimizer_G.flat_p.cpu().clone(), "D": model.optimizer_D.flat_p.cpu().clone()}
    torch.cuda.synchronize()
    spread = PS.replica_checksum_spread(model)
    final_G = model.optimizer_G.flat_p.cpu().clone()
    if rank == 1:                                                   # sensitivity of the checksum: the lowest bit of one weight on one rank
        v = model.optimizer_G.flat_p.view(torch.int32)
        v[v.numel() // 3] ^= 1
    poked = PS.replica_checksum_spread(model)
    torch.save({"G": final_G, "D": model.optimizer_D.flat_p.cpu(), "losses": losses, "first": first, "spread": spread,
                "spread_after_a_poke": poked,
                "buckets_G": list(model.optimizer_G.bucket_log), "buckets_D": list(model.optimizer_D.bucket_log),
                "total_G": model.optimizer_G._total, "total_D": model.optimizer_D._total,
                "steps": model.optimizer_G.steps_taken(), "graphed": getattr(model, "_graph_state", None) is not None
                and model._graph_state["graphs"] is not None},
               os.path.join(out_dir, f"{mode}_rank{rank}.pt"))
    if rank == 0 and mode == "eager":
        # single process on the whole batch: InstanceNorm is per sample and every loss is a mean over equal shards, so the
        # data-parallel average of the shard gradients IS the full-batch gradient
        ref = _model(g, mask=False)
        ref_ld = ref.train_step(lr_all.cuda(), hr_all.cuda())          # ONE step: later steps drift chaotically (Adam)
        torch.cuda.synchronize()
        torch.save({"G": ref.optimizer_G.flat_p.cpu(), "D": ref.optimizer_D.flat_p.cpu(),
                    "losses": {k: float(v) for k, v in ref_ld.items()}}, os.path.join(out_dir, "single.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
