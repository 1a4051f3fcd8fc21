"""Data parallelism for the hot path: one process per GPU, identical replicated weights, the minibatch sharded
along B, and ONE summing all-reduce per network per step over the flat gradient buffer (RCCL over xGMI through
torch.distributed, backend 'nccl').  Replaces the reference's single-process nn.DataParallel
(models/models.py:17-18); as there, every replica normalises its own shard (per-shard min/max in to_spectro)
and InstanceNorm needs no cross-GPU statistics.

The G-gradient all-reduce is launched right after the G backward and runs beside the D backward
(Pix2PixHDModel.train_step); Adam then consumes sum/world on every rank, which keeps the replicas identical.
"""
import torch


def all_reduce_flat_async(flat_grad, process_group=None):
    """Start the summing all-reduce of one flat gradient buffer; returns the work handle.  The buffer is reduced in
    place; the 1/world_size scaling is folded into the Adam kernel (grad_scale).  Backend-agnostic: RCCL ('nccl') on
    the GPUs, 'gloo' in the CPU tests."""
    import torch.distributed as dist
    return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=process_group, async_op=True)


def shard_batch(global_batch, rank, world_size):
    """[start, stop) of this rank's slice of the minibatch (independent samples, no data-path collective)."""
    per = global_batch // world_size
    assert per * world_size == global_batch, "global batch must divide evenly over the ranks"
    return rank * per, (rank + 1) * per


def enable_data_parallel(model, world_size, process_group=None, broadcast=True, force_collectives=False):
    import torch.distributed as dist
    if broadcast:
        # start from rank 0's weights whatever the local seeds were
        for opt in (model.optimizer_G, model.optimizer_D):
            dist.broadcast(opt.flat_p, src=0, group=process_group)
        from . import _ops
        _ops.bump_weight_epoch()
    model.optimizer_G.enable_data_parallel(world_size, process_group, force_collectives)
    model.optimizer_D.enable_data_parallel(world_size, process_group, force_collectives)
    return model
