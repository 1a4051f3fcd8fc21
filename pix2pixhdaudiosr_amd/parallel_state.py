"""Data parallelism for the hot path: one process per GPU, identical replicated weights, the minibatch sharded
along B, and ONE summing all-reduce per network per step over the flat gradient buffer (RCCL over xGMI through
torch.distributed, backend 'nccl').  Replaces the reference's single-process nn.DataParallel
(models/models.py:17-18); as there, every replica normalises its own shard (per-shard min/max in to_spectro)
and InstanceNorm needs no cross-GPU statistics.

The G-gradient all-reduce is launched right after the G backward and runs beside the D backward
(Pix2PixHDModel.train_step); Adam then consumes sum/world on every rank, which keeps the replicas identical.
"""
import torch


def enable_data_parallel(model, world_size, process_group=None, broadcast=True):
    import torch.distributed as dist
    if broadcast:
        # start from rank 0's weights whatever the local seeds were
        for opt in (model.optimizer_G, model.optimizer_D):
            dist.broadcast(opt.flat_p, src=0, group=process_group)
        from . import _ops
        _ops.bump_weight_epoch()
    model.optimizer_G.enable_data_parallel(world_size, process_group)
    model.optimizer_D.enable_data_parallel(world_size, process_group)
    return model
