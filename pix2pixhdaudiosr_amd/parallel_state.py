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


def enable_data_parallel(model, world_size, process_group=None, broadcast=True, force_collectives=False, wire_dtype=torch.float32):
    import torch.distributed as dist
    if broadcast:
        # start from rank 0's weights whatever the local seeds were
        for opt in (model.optimizer_G, model.optimizer_D):
            dist.broadcast(opt.flat_p, src=0, group=process_group)
        from . import _ops
        _ops.bump_weight_epoch()
    # (`wire_dtype` = torch.bfloat16: the generator's gradient buckets travel as bf16 -- 95 % of the exchanged bytes; the
    # discriminator's 22 MB stay fp32)
    model.optimizer_G.enable_data_parallel(world_size, process_group, force_collectives, wire_dtype)
    model.optimizer_D.enable_data_parallel(world_size, process_group, force_collectives)
    return model


def flat_checksum(flat):
    """64-bit position-weighted checksum of a flat fp32 buffer's BITS (int64 arithmetic, wraps): equal on two replicas iff --
    up to a 2^-64 collision -- every element has the same bits in the same place."""
    bits = flat.detach().contiguous().view(torch.int32).to(torch.int64)
    w = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64) % 2147483629 + 1
    return int((bits * w).sum().item())


def replica_checksum_spread(model, process_group=None):
    """{"G": s, "D": s}: max - min over the ranks of the checksum of each network's master weights.  0 = the replicas are
    bit-identical, which is what data parallelism with a summing all-reduce and identical Adam steps must preserve; anything
    else means an exchange was missed, applied twice or read before it had finished (round-4 review, weak 13: a one-rank
    rehearsal cannot see a missing stream dependency, so the first real N-rank run must say so itself -- bench.py prints
    this in its `dist` block)."""
    import torch.distributed as dist
    out = {}
    for name, opt in (("G", model.optimizer_G), ("D", model.optimizer_D)):
        c = flat_checksum(opt.flat_p)
        dev = opt.flat_p.device if dist.get_backend(process_group) == "nccl" else torch.device("cpu")
        hi = torch.tensor([c], dtype=torch.int64, device=dev)
        lo = hi.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=process_group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=process_group)
        out[name] = int(hi.item()) - int(lo.item())
    return out


def masked_compute_stream(device, free_cus):
    """A HIP stream whose kernels may run on every CU but `free_cus` of them (hipExtStreamCreateWithCUMask), wrapped for
    torch.  Every MFMA kernel of the step takes a CU's whole LDS (one workgroup per CU), so on N > 1 GPUs the RCCL
    kernels of the gradient exchange only get onto the chip when a workgroup retires; running the step on this stream
    keeps `free_cus` CUs out of the step's reach so that the collectives co-run (opt.comm_cus / bench.py --comm-cus).
    The price is a tile round on every layer whose grid filled the chip exactly, which is why it is a knob."""
    import ctypes
    dev = torch.device(device)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    free_cus = int(free_cus)
    if not 0 < free_cus < n_cu:
        raise ValueError(f"comm_cus must be in (0, {n_cu}), got {free_cus}")
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(n_cu - free_cus):
        mask[i // 32] |= 1 << (i % 32)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    handle = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), words, mask)
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed (code {rc})")
    st = torch.cuda.ExternalStream(handle.value, device=dev)
    st._p2phd_free_cus = free_cus
    return st
