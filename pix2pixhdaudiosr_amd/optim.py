"""Flat-buffer Adam on the HIP kernel p2phd_adam_step, with the torch.optim.Optimizer surface train.py uses
(param_groups[...]['lr'], zero_grad(), step(), state_dict()).

MI355X-first layout: all parameters of one network live in ONE contiguous fp32 buffer and all their gradients in
another.  That makes the optimiser a single HBM-bound launch and the data-parallel exchange a single (or a few
large) RCCL all-reduce over xGMI instead of one collective per layer.  ``nn.Parameter`` objects keep their
identity: their storage is re-pointed at slices of the flat buffer, so ``state_dict``/checkpoints are unchanged.
Update rule = torch.optim.Adam(lr, betas, eps=1e-8), which is what the reference builds
(models/pix2pixHD_model.py:131,140).
"""
import torch

from . import _lib, _ops


class DeviceGradScaler:
    """torch.cuda.amp.GradScaler (train.py:62-67,165-181: ONE scaler for both losses) with its state on the device, so that the
    captured training step replays unchanged: `scale(loss)` multiplies by the current scale (read from device memory by the
    kernel), FlatAdam.step_local scans its gradient buffer for inf / nan, unscales inside the Adam kernel and skips the update
    on a non-finite gradient, `update()` applies backoff / growth.  Used with fp16 storage; bf16 and fp32 need none."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.zeros(8, dtype=torch.float32, device=device)
        self.state[0] = float(init_scale)
        self.state[1] = 1.0 / float(init_scale)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)

    def scale(self, loss):
        return loss * self.state[0]

    def update(self):
        _lib.check(_lib.lib().p2phd_scaler_update(_lib.ptr(self.state), self.growth_factor, self.backoff_factor, self.growth_interval,
                                                  _lib.stream_ptr()), "scaler_update")

    def get_scale(self):
        return float(self.state[0].item())

    def state_dict(self):
        """Scale, its reciprocal and the growth tracker (the found-inf flags are per-step scratch)."""
        return {"state": self.state.detach().cpu().clone(), "growth_factor": self.growth_factor,
                "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval}

    def load_state_dict(self, sd):
        st = sd["state"].to(dtype=torch.float32)
        if st.numel() != self.state.numel():
            raise ValueError("DeviceGradScaler: state of another layout")
        self.state.copy_(st.to(self.state.device))
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])

    def load_file(self, path):
        import os
        if os.path.isfile(path):
            self.load_state_dict(torch.load(path, map_location="cpu"))
            return True
        return False


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8, process_group=None):
        params = [p for p in params]
        if not params:
            raise ValueError("FlatAdam got an empty parameter list")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        dev = params[0].device
        if dev.type != "cuda":
            raise _lib.P2PHDError("FlatAdam runs on the GPU only (no CPU fallback)")
        sizes = [p.numel() for p in params]
        # 16-byte aligned slices so every parameter / gradient view can be read with float4
        # (8 elements: a bf16 copy at the same element offsets would be 16-byte aligned per parameter as well)
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 7) & ~7
        self._params, self._offs, self._total = params, offs, total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                v = self._view(self.flat_p, p, o)
                v.copy_(p.detach().float())
                p.data = v
        self._install_grad_views()
        self.step_count = 0
        # learning rate and step counter live on the device (p2phd_adam_step_dev): a captured step replays unchanged
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._lr_pushed = None
        self.process_group = process_group
        self.world_size = 1
        self._collectives = False
        self._pending = None
        self.bucket_log = []
        # exchange-exposure instrumentation (bench.py, tests): when `timing` is on, every wait_gradients() that has
        # something to wait for is bracketed by events on the compute stream -- the time between them is the time the
        # compute stream stood still for the collectives, i.e. the EXPOSED part of the exchange -- and by a host clock
        # (backends whose wait blocks the host: gloo).
        self.timing = False
        self._wait_events = []
        self._wait_host_s = 0.0
        self._waits = 0
        _ops.bump_weight_epoch()

    @staticmethod
    def _view(flat, p, o):
        """The slice of a flat buffer that holds parameter p, shaped like p.  Conv weights marked K-major
        (networks._ConvStep: the big stride-1 layers) are STORED [K][R][S][C] -- the layout the kernels' packed rows have --
        and shown to torch as the permuted [K, C, R, S] view: state_dict, initialisers and checkpoints see nothing new."""
        n = p.numel()
        if getattr(p, "_p2phd_kmajor", False) and p.dim() == 4:
            K, Cc, R, S = p.shape
            return flat[o:o + n].view(K, R, S, Cc).permute(0, 3, 1, 2)
        return flat[o:o + n].view(p.shape)

    def _install_grad_views(self):
        for p, o in zip(self._params, self._offs):
            v = self._view(self.flat_g, p, o)
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v
            p._p2phd_direct_grad = True                          # _ops.ConvBlockFn adds its weight gradients in place

    def zero_grad(self, set_to_none=False, lazy=False):
        """Gradients stay views of the flat buffer.  Default: one memset instead of one per tensor.  `lazy` (the training
        step): no memset at all -- every parameter is marked fresh, the conv blocks' first weight gradient of the step
        overwrites it (_ops.ConvBlockFn.backward), and whatever is still fresh when the gradients are used (exchange,
        update) is zeroed then (`_flush_fresh`)."""
        self._install_grad_views()
        if lazy:
            # weights (>= 2-D): overwritten by their first gradient; the small 1-D parameters (biases: several kernels ADD their
            # column sums) are zeroed here, all in one launch
            small = [(o, p.numel()) for p, o in zip(self._params, self._offs) if p.dim() < 2]
            if small:
                if getattr(self, "_small_seg", None) is None:
                    self._small_seg = torch.tensor([v for pair in small for v in pair], dtype=torch.int64, device=self.flat_g.device)
                _lib.check(_lib.lib().p2phd_zero_segments(_lib.ptr(self.flat_g), _lib.ptr(self._small_seg), len(small), _lib.stream_ptr()),
                           "zero_segments")
            for p in self._params:
                p._p2phd_fresh = p.dim() >= 2
            self._lazy = True
            return
        self.flat_g.zero_()
        for p in self._params:
            p._p2phd_fresh = False
        self._lazy = False

    def _flush_fresh(self, start=0, stop=None):
        """Zero the gradient of every parameter in [start, stop) that no backward pass has written since a lazy zero_grad."""
        if not getattr(self, "_lazy", False):
            return
        stop = self._total if stop is None else stop
        for p, o in zip(self._params, self._offs):
            if getattr(p, "_p2phd_fresh", False) and start <= o < stop:
                self.flat_g[o:o + p.numel()].zero_()
                p._p2phd_fresh = False

    # ---- data parallel: one summing all-reduce of the whole gradient buffer over RCCL ----
    def enable_data_parallel(self, world_size, process_group=None, force_collectives=False, wire_dtype=torch.float32):
        """`force_collectives`: issue the all-reduces even with one rank (rehearsal of the RCCL call sequence on a
        one-GPU box: same launches, streams and waits as N ranks; the sum over one rank is the identity).
        `wire_dtype` = torch.bfloat16: the gradient buckets travel as bf16 -- half the bytes over xGMI (205 MB instead of
        410 MB for configs[1]'s generator) -- and are widened back into the fp32 gradient buffer before the update, so Adam
        and its moments stay fp32; every rank receives the same reduced values, so the replicas stay bit-identical.  The sum
        itself is rounded to 8 mantissa bits per hop, which is why fp32 stays the default."""
        self.world_size = int(world_size)
        self.process_group = process_group
        self._collectives = self.world_size > 1 or bool(force_collectives)
        if wire_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("wire_dtype must be torch.float32 or torch.bfloat16")
        self.wire_dtype = wire_dtype
        self._wire = torch.empty(self._total, dtype=torch.bfloat16, device=self.flat_g.device) if wire_dtype == torch.bfloat16 else None

    def param_offset(self, index):
        """Element offset inside the flat buffers of parameter `index` (construction order)."""
        return self._offs[index]

    def reduce_range_async(self, start, stop):
        """Start the summing all-reduce of flat_g[start:stop] (one bucket of a staged backward).  Buckets must not
        overlap; the launches are recorded in `bucket_log` (tests assert order and coverage)."""
        self._flush_fresh(start, stop)
        if self._collectives and stop > start:
            from .parallel_state import all_reduce_flat_async
            if self._pending is None:
                self._pending = []
            if getattr(self, "_wire", None) is not None:
                w = self._wire[start:stop]
                w.copy_(self.flat_g[start:stop])                   # fp32 -> bf16 for the wire
                self._pending.append((all_reduce_flat_async(w, self.process_group), int(start), int(stop)))
            else:
                self._pending.append((all_reduce_flat_async(self.flat_g[start:stop], self.process_group), None, None))
        self.bucket_log.append((int(start), int(stop)))

    def reduce_gradients_async(self):
        """Start the gradient all-reduce of the whole buffer (call right after backward); step() waits for it."""
        if self._collectives and self._pending is None:
            self._install_grad_views()
            self.bucket_log = []
            self.reduce_range_async(0, self._total)

    def wait_gradients(self):
        """Make the current stream wait for every gradient all-reduce started since the last wait."""
        if self._pending is not None:
            if self.timing:
                import time
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                t0 = time.perf_counter()
            for h, a, b in self._pending:
                h.wait()
                if a is not None:                                  # widen the reduced bf16 bucket back into the fp32 buffer Adam reads
                    self.flat_g[a:b].copy_(self._wire[a:b])
            if self.timing:
                self._wait_host_s += time.perf_counter() - t0
                e1.record()
                self._wait_events.append((e0, e1))
                self._waits += 1
            self._pending = None

    def reset_exchange_timing(self, on=True):
        self.timing = bool(on)
        self._wait_events, self._wait_host_s, self._waits = [], 0.0, 0

    def exchange_timing(self):
        """{'waits', 'exposed_stream_ms', 'host_wait_ms'} summed since reset_exchange_timing(); synchronises."""
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._wait_events)
        return {"waits": self._waits, "exposed_stream_ms": float(ms), "host_wait_ms": self._wait_host_s * 1e3}

    @torch.no_grad()
    def step(self, closure=None, use_scaler=False):
        if closure is not None:
            raise NotImplementedError("closures are not supported")
        self._install_grad_views()
        if self._collectives:
            self.reduce_gradients_async()
            self.wait_gradients()
        self.step_local(use_scaler)

    @torch.no_grad()
    def step_local(self, use_scaler=False):
        """The update itself, without the data-parallel exchange (graph-capturable: every argument is constant).
        `use_scaler`: the gradients carry the attached DeviceGradScaler's scale (Pix2PixHDModel.train_step with fp16 storage);
        a caller that runs torch's own GradScaler (train.py's loop) hands over unscaled gradients and leaves it False."""
        g = self.param_groups[0]
        self._flush_fresh()
        self.sync_hyper()
        self.step_count += 1
        b1, b2 = g["betas"]
        sc = getattr(self, "scaler", None) if use_scaler else None
        if sc is not None:                                         # fp16 storage: scaled gradients, skip on inf / nan (DeviceGradScaler)
            _lib.check(_lib.lib().p2phd_adam_step_scaled(_lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.exp_avg),
                                                         _lib.ptr(self.exp_avg_sq), self._total, _lib.ptr(self.lr_dev),
                                                         _lib.ptr(self.step_dev), float(b1), float(b2), float(g["eps"]),
                                                         1.0 / self.world_size, _lib.ptr(sc.state), int(self.scaler_index),
                                                         _lib.stream_ptr()), "adam_step_scaled")
        else:
            _lib.check(_lib.lib().p2phd_adam_step_dev(_lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.exp_avg),
                                                      _lib.ptr(self.exp_avg_sq), self._total, _lib.ptr(self.lr_dev),
                                                      _lib.ptr(self.step_dev), float(b1), float(b2), float(g["eps"]),
                                                      1.0 / self.world_size, _lib.stream_ptr()), "adam_step")
        _ops.bump_weight_epoch()

    def sync_hyper(self):
        """Push param_groups[0]['lr'] to the device copy the kernel reads (only when it changed).  Called by step();
        a caller that REPLAYS a captured step calls it itself before the replay."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_pushed:
            self.lr_dev.fill_(lr)
            self._lr_pushed = lr

    def steps_taken(self):
        """Number of updates applied so far, including replays of a captured step (device counter; synchronises)."""
        return int(self.step_dev.item())
