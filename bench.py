#!/usr/bin/env python3
"""bench.py -- MDCT spectrogram frames/s through one full GAN training step on MI355X.

Step (= train.py:148-184 of the reference): MDCT4 encode of hr + lr audio -> dB/sign encoding -> GlobalGenerator
forward -> MultiscaleDiscriminator x3 -> LSGAN + feature-matching losses -> G backward -> D backward -> Adam x2,
on synthetic audio already resident in HBM.  Workload = BASELINE.json configs[1]: ngf=48, n_local_enhancers=0
(netG 'global', 4 down-samplings, 9 residual blocks), 512x256 spectrograms (n_fft 1024, hop 512), bf16 MFMA
compute, per-GPU batch 32.  N > 1: one process per GPU (torch.distributed / RCCL), the minibatch is sharded, the
flat G / D gradient buffers are all-reduced over xGMI; weak scaling.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel: the implicit-GEMM conv of
the residual trunk, timed live with HIP events) and `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES = 256                      # spectrogram frames per sample at 512x256
BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
FP8_DENSE_PEAK_TFLOPS = 5000.0    # ~5 PF dense fp8 (block-scaled f8f6f4 MFMA)
M_G = 61.03e9                     # conv MACs / sample, GlobalGenerator ngf48 nd4 nb9 @512x256 (SURVEY 8a probe)
M_D = 8.98e9                      # conv MACs / sample, MultiscaleDiscriminator num_D 2 @512x256


HALF = torch.bfloat16              # --fp16-storage: torch.float16 (the fp16 build of the library serves every launch)

# --config: the headline workload is BASELINE configs[1] ("cfg2" in SURVEY 8d's numbering); the two others are NON-headline
# variants (round-4 review item 5) whose JSON line says so (`headline: false`), names its configuration in config.workload and
# prices ITS dominant kernel in `roofline`.  step_gflop: SURVEY 8(d), per sample and step, minimal result-identical schedule.
# trunk = (channels, plane rows, plane columns) of the residual trunk's Conv3x3 behind ReflectionPad2d(1): the layer with the most
# MACs of each generator.
WORKLOADS = {
    "cfg2": dict(overrides={}, frames=256, batch=32, step_gflop=None, trunk=(768, 32, 16), headline=True,
                 workload="configs[1]: ngf=48 n_local_enhancers=0 (GlobalGenerator nd4 nb9) + MultiscaleDiscriminator "
                          "num_D=2, 512x256 MDCT4 (n_fft 1024, hop 512), LSGAN + feature matching, Adam, bf16 MFMA"),
    "cfg3": dict(overrides=dict(netG="local", n_blocks_global=3, n_local_enhancers=1, n_blocks_local=2), frames=256, batch=32,
                 step_gflop=398.0, trunk=(1536, 16, 8), headline=False,
                 workload="configs[2]/[3] per-rank workload: GEN_VCTK_G3L2_48ngf as the reference's opt.txt reads it (LocalEnhancer ngf 48, "
                          "4 global down-samplings, 3 global blocks, 1 local enhancer, 2 local blocks: 156 050 690 parameters) + "
                          "MultiscaleDiscriminator num_D=2, 512x256 MDCT4, LSGAN + feature matching, Adam, bf16 MFMA"),
    "cfg5": dict(overrides=dict(netG="local", ngf=64, n_local_enhancers=1, n_blocks_local=3, n_fft=2048, hop_length=1024, win_length=2048,
                                num_D=3), frames=512, batch=8, step_gflop=3994.0, trunk=(2048, 32, 16), headline=False,
                 workload="configs[4] per-rank workload: n_fft 2048 (1024x512 MDCT4 spectrograms), LocalEnhancer ngf 64 (730 713 346 "
                          "parameters), 3-scale MultiscaleDiscriminator, LSGAN + feature matching, Adam, bf16 MFMA (--fp8: + e4m3 forward of the wide convs)"),
}


def make_opt(batch, dtype_bf16=True, fp8=False, fp16_storage=False):
    return SimpleNamespace(
        gpu_ids=[0], isTrain=True, checkpoints_dir="/tmp/p2phd_bench", name="bench", model="pix2pixHD",
        input_nc=2, output_nc=2, label_nc=0, hr_sampling_rate=48000, lr_sampling_rate=8000,
        n_fft=1024, hop_length=512, win_length=1024, center=True, no_instance=True,
        ngf=48, netG="global", n_downsample_global=4, n_blocks_global=9, n_local_enhancers=0, n_blocks_local=3,
        norm="instance", no_lsgan=False, ndf=64, n_layers_D=3, num_D=2, no_ganFeat_loss=False,
        use_hifigan_D=False, use_time_D=False, verbose=False, continue_train=False, load_pretrain="",
        which_epoch="latest", pool_size=0, lr=0.0002, beta1=0.5, no_vgg_loss=True, use_match_loss=False,
        niter_fix_global=0, explicit_encoding=True, alpha=0.6, min_value=1e-7, mask=True, mask_mode="mode2",
        lambda_feat=10.0, fp16=dtype_bf16, fp16_storage=fp16_storage, fp8=fp8, niter_decay=100, instance_feat=False, label_feat=False, batchSize=batch)


def time_trunk_conv(batch, iters=20, trunk=(768, 32, 16)):
    """Average launch duration (HIP events on the launch stream) of the dominant kernel: Conv3x3 768->768 on
    [B,32,16,768] bf16 behind ReflectionPad2d(1) -- 18 of the 28 generator convs, 80 % of its MACs (`trunk` = (channels, rows,
    columns) of the same layer in another --config)."""
    from pix2pixhdaudiosr_amd import _ops
    import ctypes as C
    ch, th, tw = trunk
    spec = _ops.ConvSpec(ch, ch, 3, 1, 1, 1, False, 0, True, _ops.ACT_RELU)
    x = torch.randn(batch, th, tw, ch, device="cuda").to(HALF)
    w = (torch.randn(ch, ch, 3, 3, device="cuda") * 0.02)
    d = spec.desc(batch, th, tw, HALF)
    wp = spec.packed(w, 0, d)
    y = torch.empty_like(x)
    stats = torch.zeros(batch, ch, 2, device="cuda")
    L = _ops.lib_for(HALF)
    ws = torch.empty(max(L.p2phd_conv_fwd_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device="cuda")
    call = lambda: _ops.check(L.p2phd_conv_fwd(C.byref(d), _ops.ptr(x), _ops.ptr(wp), None, 0, _ops.ptr(y), _ops.ptr(stats),
                                               _ops.ptr(ws), _ops.stream_ptr()))
    sec = time_graphed(call, iters)
    flops = 2.0 * batch * th * tw * ch * ch * 9
    return sec, flops


def time_graphed(call, iters=20):
    """Average duration of one `call()` (a chain of launches on the current stream): `iters` repetitions captured into
    one HIP graph, HIP events around a replay.  Consecutive graph nodes start within ~1-2 us of each other, so
    elapsed / iters is kernel time (agrees with rocprofv3's per-kernel average)."""
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph.capture_begin(capture_error_mode="thread_local")   # see Pix2PixHDModel.train_step_graphed
        for _ in range(iters):
            call()
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) / 1e3 / iters
        best = sec if best is None else min(best, sec)
    return best


def time_generator(model, batch):
    """GlobalGenerator forward + backward alone (north_star target: >= 40 % MFMA on G fwd+bwd at 512x256 bf16): the
    generator of the benchmarked model on a [B,2,512,256] input, every weight gradient into the flat buffer, captured into a
    HIP graph and timed with events.  Algorithmic work: 6 M_G FLOP per sample (forward + input gradients + weight
    gradients; the first layer has no input gradient -- stated in the JSON)."""
    from pix2pixhdaudiosr_amd import _ops
    netG, optG = model.netG, model.optimizer_G
    x = torch.rand(batch, 2, 512, 256, device="cuda")
    gy = None

    def call():
        nonlocal gy
        _ops.begin_step(model.device)
        y = netG.forward_physical(netG.input_physical(x))
        _ops.end_arena(model.device)
        if gy is None:
            gy = (torch.randn(y.shape, device="cuda") * 1e-3).to(y.dtype)
            gy[..., 2:] = 0
        y.backward(gy, inputs=list(optG._params))

    sec = time_graphed(call, iters=3)
    optG.zero_grad()
    return sec


HBM_PEAK_TBS = 8.0                # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s float4-copy achievable)


def time_mdct(batch):
    """MDCT4 and IMDCT4 alone: frames/s and GB/s on the algorithmic bytes (read hop + write n_fft/2 floats
    per frame = 4 KiB at n_fft 1024, 8 KiB at 2048), at the configs[1] geometry [B, 130560] and at n_fft 2048."""
    from pix2pixhdaudiosr_amd.models import mdct as MM
    from pix2pixhdaudiosr_amd.util.util import kbdwin
    out = {}
    for n_fft, frames, rows in ((1024, 256, batch), (1024, 256, 2 * batch), (2048, 512, batch)):
        hop = n_fft // 2
        T = (frames - 1) * hop
        w = kbdwin(n_fft).cuda()
        tables = MM._Tables.get(n_fft, w.device)
        x = 0.1 * torch.randn(rows, T, device="cuda")
        sp, _, nf = MM.frame_layout(batch, T, hop, n_fft, True)
        S = MM._run_mdct(x, n_fft, hop, n_fft, w, tables, sp, nf, 1.0)
        t_f = time_graphed(lambda: MM._run_mdct(x, n_fft, hop, n_fft, w, tables, sp, nf, 1.0))
        t_i = time_graphed(lambda: MM._run_imdct(S, n_fft, hop, n_fft, w, tables, n_fft // 2, T, 4.0 / n_fft))
        nframes = rows * nf
        bytes_alg = nframes * 4 * n_fft          # (hop + n_fft/2) floats = 4 n_fft bytes per frame (4 KiB at n_fft 1024)
        tag = "" if rows == batch else f"_rows{rows}"        # rows = 2 x batch: hr and lr of one step in ONE launch (encode_input)
        for name, t in (("mdct4", t_f), ("imdct4", t_i)):
            out[f"{name}_n{n_fft}{tag}"] = {"shape": [rows, T], "frames": nframes, "us": t * 1e6, "frames_per_s": nframes / t,
                                       "GB_per_s": bytes_alg / t / 1e9, "frac_of_hbm_peak": bytes_alg / t / 1e12 / HBM_PEAK_TBS,
                                       "bytes_per_frame": 4 * n_fft}
    return out


def host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))          # a 1-GPU box's CPU share is 16 cores


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def _cpu_leg(ngf, sample_batch, steps):
    from oracle import model as OM
    from oracle import mdct4 as M4
    opt = OM.default_opt(ngf=ngf, netG="global", n_downsample_global=4, n_blocks_global=9)
    pG = OM.N.init_params(OM.netG_spec(opt), seed=1)
    pD = OM.N.init_params(OM.netD_spec(opt), seed=2)
    hr, lr, noise = OM.synthetic_batch(sample_batch, opt)
    w = M4.kbdwin(opt.win_length)
    sG, sD = {}, {}
    _, pG, pD = OM.full_step(hr, lr, noise, pG, pD, opt, w, sG, sD)           # one warm-up step
    t0 = time.time()
    for _ in range(steps):
        _, pG, pD = OM.full_step(hr, lr, noise, pG, pD, opt, w, sG, sD)
    dt = (time.time() - t0) / steps
    return sample_batch * FRAMES / dt, dt


def cpu_baseline(sample_batch=2, steps=3):
    """The CPU oracle (oracle/model.py full_step: a port of the reference step, validated against the reference in
    tests/) on the host cores: the benchmarked workload (configs[1]'s networks) and the reference's own CPU-runnable case
    (configs[0]: ngf 32), each at batch 2, 1 warm-up + 3 timed steps."""
    torch.set_num_threads(host_cores())
    v1, dt1 = _cpu_leg(48, sample_batch, steps)
    v0, dt0 = _cpu_leg(32, sample_batch, steps)
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    return {"value": v1, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port", "cpu": cpu_model_name(),
            "cores_visible": visible, "cores_note": "threads used = min(visible cores, 16): a one-GPU box's CPU share is 16 cores",
            "sample": f"oracle full_step (CPU port of train.py:148-184), configs[1]'s network/config at batch {sample_batch}, "
                      f"1 warm-up + {steps} timed step(s), fp32, {dt1:.2f} s/step",
            "configs0": {"value": v0, "unit": "frames/s",
                         "sample": f"same step with configs[0]'s generator (ngf 32) at batch {sample_batch}, {dt0:.2f} s/step"}}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: THIS process has not touched the GPU yet (importing
    torch does not), so it starts one rank per GPU through torch.distributed.run as child processes, relays rank 0's
    JSON line and exits with the launcher's code.  Never an exec of a process that initialised HIP."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env)
    try:
        rc = proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        rc = proc.wait()
    sys.exit(rc)


_JSON_OUT = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 32; 8 for --config cfg5)")
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="cfg2",
                    help="cfg2 (default) = BASELINE configs[1], the headline workload.  cfg3 / cfg5 = the per-rank workloads of configs[2]/[3] and "
                         "configs[4]: NON-headline variants, marked `headline: false` in the JSON line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager step on one GPU too (default: captured HIP graph)")
    ap.add_argument("--allow-eager-fallback", action="store_true",
                    help="keep measuring with the eager step if graph capture / replay fails (default: exit non-zero)")
    ap.add_argument("--no-mdct", action="store_true", help="skip the stand-alone MDCT4 / IMDCT4 measurement")
    ap.add_argument("--no-probes", action="store_true",
                    help="skip the extra eager steps of the dominant-kernel probe and the generator-only timing (profiling runs that "
                         "count launches or bytes per step: tools/collect_step_traffic.sh)")
    ap.add_argument("--comm-cus", type=int, default=0,
                    help="data parallel: run the step on a stream whose CU mask leaves this many CUs to the RCCL kernels (0 = off)")
    ap.add_argument("--wire-bf16", action="store_true",
                    help="data parallel: the generator's gradient buckets travel as bf16 (half the xGMI bytes; fp32 accumulation in Adam)")
    ap.add_argument("--fp16-storage", action="store_true",
                    help="variant: IEEE fp16 activations (the reference's autocast type) through libp2phd_hip_f16.so with the device "
                         "loss scaler; same kernels, same byte counts (NOT the headline dtype: BASELINE configs[1] says bf16)")
    ap.add_argument("--fp8", action="store_true",
                    help="variant of BASELINE configs[4]: e4m3 forward of the wide stride-1 convs on top of bf16 (NOT the headline dtype)")
    a = ap.parse_args()
    wl = WORKLOADS[a.config]
    if a.batch is None:
        a.batch = wl["batch"]

    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    # ONE JSON line on stdout, nothing else: libraries print banners there (RCCL 2.26 writes its version block to stdout
    # when the first communicator is created), so fd 1 is pointed at stderr for the whole run and the line goes to a
    # duplicate of the original stdout
    global _JSON_OUT
    if _JSON_OUT is None and ("WORLD_SIZE" in os.environ or a.gpus == 1):
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:
            launch_ranks(a.gpus, sys.argv[1:])                     # does not return
        world, rank, local = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if world != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # rehearsal hooks (never set by the driver): run several ranks on one GPU over gloo to exercise the DP code path
    backend = os.environ.get("P2PHD_DIST_BACKEND", "nccl")
    if "P2PHD_FORCE_DEVICE" in os.environ:
        local = int(os.environ["P2PHD_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    # P2PHD_REHEARSE_RCCL=1 (one rank, never set by the driver): a one-rank RCCL group with the collectives forced on, so
    # that the exact call sequence of the N-rank step -- bucketed all-reduces between graph replays, waits, Adam --
    # runs against real RCCL on a one-GPU box
    rehearse = world == 1 and os.environ.get("P2PHD_REHEARSE_RCCL") == "1"
    dist_on = world > 1 or rehearse
    if rehearse:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=torch.device("cuda", local))
    elif world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from pix2pixhdaudiosr_amd.models.models import create_model
    from pix2pixhdaudiosr_amd import parallel_state

    torch.manual_seed(1234)                      # same initial weights on every rank (reference default seed)
    if a.fp16_storage:
        global HALF
        HALF = torch.float16
        if a.fp8:
            raise SystemExit("--fp8 rides on bf16; not with --fp16-storage")
    opt = make_opt(a.batch, fp8=a.fp8, fp16_storage=a.fp16_storage)
    for k_, v_ in wl["overrides"].items():
        setattr(opt, k_, v_)
    frames = wl["frames"]
    opt.gpu_ids = [local]
    opt.comm_cus = a.comm_cus
    model = create_model(opt)
    if rehearse:
        opt.grad_buckets = 4
    if dist_on:
        parallel_state.enable_data_parallel(model, world, force_collectives=rehearse,
                                            wire_dtype=torch.bfloat16 if a.wire_bf16 else torch.float32)
    T = (frames - 1) * opt.hop_length
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    hr = 0.1 * torch.randn(a.batch, T, device="cuda", generator=g)
    lr = 0.1 * torch.randn(a.batch, T, device="cuda", generator=g)

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    log(f"model built, batch {a.batch}, world {world}")
    # The step is captured once into HIP graphs and replayed (same kernels, same order, no host work per step); with
    # data parallelism the two RCCL all-reduces run between the replays, the G one overlapping the D backward.
    graphed = not a.no_graph
    graph_error = None
    step = model.train_step_graphed if graphed else model.train_step

    def graph_failed(what, e):
        # a failed capture / replay leaves the side stream and the graph pool in an undefined state: the measurement is
        # only continued (eagerly) on request, and the JSON line then says so
        nonlocal graphed, step, graph_error
        graph_error = f"{what}: {type(e).__name__}: {e}"
        log(graph_error)
        if not a.allow_eager_fallback:
            raise SystemExit(f"bench.py: {graph_error} (pass --allow-eager-fallback or --no-graph to measure the eager step)")
        graphed, step = False, model.train_step
        model._graph_state = None
        torch.cuda.synchronize()

    if graphed:
        try:
            for _ in range(3):                                     # two eager steps + capture, before the warm-up steps
                step(lr, hr)
            torch.cuda.synchronize()
            log("step captured into HIP graphs (forward + G backward | D backward | Adam), collectives between replays")
        except Exception as e:
            graph_failed("graph capture failed", e)
    for i in range(a.warmup):
        try:
            step(lr, hr)
            torch.cuda.synchronize()
        except Exception as e:
            if not graphed:
                raise
            graph_failed("graph replay failed in warm-up", e)
            step(lr, hr)
            torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    barrier()
    for o_ in (model.optimizer_G, model.optimizer_D):
        o_.reset_exchange_timing(dist_on)                          # events around every wait for a collective (N > 1 / rehearsal)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(lr, hr)
    barrier()
    dt = time.perf_counter() - t0
    exch = {"G": model.optimizer_G.exchange_timing(), "D": model.optimizer_D.exchange_timing()} if dist_on else None
    for o_ in (model.optimizer_G, model.optimizer_D):
        o_.reset_exchange_timing(False)
    rank_ms = [dt / a.steps * 1e3]
    dist_info = None
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(v.item()) / a.steps * 1e3 for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # exposed exchange: time the COMPUTE stream stood still in wait_gradients() (events around the waits), per step,
        # max over ranks; host_wait_ms is the host-side blocking of the same waits (gloo)
        ex = torch.tensor([exch["G"]["exposed_stream_ms"] / a.steps, exch["D"]["exposed_stream_ms"] / a.steps,
                           (exch["G"]["host_wait_ms"] + exch["D"]["host_wait_ms"]) / a.steps], device="cuda", dtype=torch.float64)
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        gb = [int(b - a) for a, b in model.optimizer_G.bucket_log]
        torch.cuda.synchronize()
        spread = parallel_state.replica_checksum_spread(model)      # collective: every rank calls it
        dist_info = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "rehearsal_one_rank": rehearse,
                     "g_gradient_buckets": gb,
                     "bucket_bytes": {"G": [(2 if a.wire_bf16 else 4) * n for n in gb], "D": [4 * int(model.optimizer_D._total)]},
                     "gradient_dtype_on_the_wire": {"G": "bf16" if a.wire_bf16 else "fp32", "D": "fp32"},
                     "exposed_exchange_ms": float(ex[0] + ex[1]), "exposed_exchange_ms_G": float(ex[0]),
                     "exposed_exchange_ms_D": float(ex[1]), "host_wait_ms": float(ex[2]),
                     "waits_per_step": (exch["G"]["waits"] + exch["D"]["waits"]) / a.steps,
                     "exposed_how": "HIP events on the compute stream around every wait for a collective (optim.FlatAdam.wait_gradients): "
                                    "the G buckets are waited for after the D backward, the D bucket after the generator's Adam",
                     "comm_cus": a.comm_cus,
                     # max - min over the ranks of a 64-bit checksum of the master weights after the timed steps: 0 = the
                     # replicas are still bit-identical (a missed / early-read exchange would show here, not in the speed)
                     "replica_checksum_spread": spread,
                     "devices": sorted({local}) if "P2PHD_FORCE_DEVICE" in os.environ else list(range(world))}

    if rank == 0:
        ms = dt / a.steps * 1e3
        frames_per_s = world * a.batch * frames * a.steps / dt
        # minimal result-identical schedule = (3 M_G + 8 M_D) MACs = (6 M_G + 16 M_D) FLOPs per sample
        step_flops = ((6 * M_G + 16 * M_D) if wl["step_gflop"] is None else wl["step_gflop"] * 1e9) * a.batch
        out = {
            "metric": "MDCT spectrogram frames/sec (G+D fwd+bwd) at 512x256",
            "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16+fp8(e4m3 forward of the wide convs)" if a.fp8 else ("fp16" if a.fp16_storage else "bf16"), "data": "synthetic",
            "config": {"workload": wl["workload"],
                       "per_gpu_batch": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                       "launch": "hip-graph replay" if graphed else "eager",
                       "step_tflops_per_gpu": step_flops / (dt / a.steps) / 1e12,
                       "step_frac_of_bf16_peak": step_flops / (dt / a.steps) / 1e12 / BF16_DENSE_PEAK_TFLOPS,
                       "step_flops_model": "algorithmic unit of SURVEY 8(d): (6 M_G + 16 M_D) FLOP per sample; the shared "
                                           "D(fake) forward executes ~3.5 % fewer"},
            "per_rank_ms_per_step": rank_ms,
        }
        if not wl["headline"]:
            out["headline"] = False
            out["config"]["variant_of"] = ("NOT the BASELINE headline workload (configs[1], the default run): `python bench.py --config %s`; "
                                           "metric, unit and timing protocol are the headline's" % a.config)
        if a.fp16_storage:
            out["fp16_storage"] = {"library": "libp2phd_hip_f16.so (the same sources with the 16-bit type = _Float16)",
                                   "loss_scale_after_the_run": model.scaler.get_scale(),
                                   "updates_applied": {"G": model.optimizer_G.steps_taken(), "D": model.optimizer_D.steps_taken()},
                                   "what": "the reference's AMP storage type (train.py:62-67) with optim.DeviceGradScaler inside the captured step; "
                                           "fp32 master weights and fp32 MFMA accumulation as in bf16"}
        if a.fp8:
            from pix2pixhdaudiosr_amd import _ops as _o
            out["fp8"] = {"layers_switched": model.fp8_layers, "fp8_conv_launches_recorded_by_the_host": _o._FP8_CALLS[0],
                          "what": "OCP e4m3 operands on the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (unit e8m0 scales, 2x the bf16 rate) for the forward of the stride-1 convs with >= 256 "
                                  "input channels; fp32 master weights, bf16 activations and the bf16 backward unchanged"}
        if dist_info is not None:
            out["dist"] = dist_info
        if graph_error is not None:
            out["graph_error"] = graph_error
        log(f"timed region done: {ms:.1f} ms/step")
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure comes
        # from the separate rocprofv3 --pmc passes recorded in profiles/ (same kernel, same shapes, per launch)
        traffic, traffic_source = None, None
        for name in ("r05_trunk_pmc.json", "r04_trunk_pmc.json", "r03_trunk_pmc.json", "r02_trunk_pmc.json", "r01_trunk_pmc.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    rec = json.load(f)
                if a.batch == 32 and wl["headline"]:
                    traffic = rec["hbm_bytes_per_launch"]
                    traffic_source = {"file": "profiles/" + name, "recorded": rec.get("recorded"), "how": rec.get("how")}
                break
            except Exception:
                continue
        # Dominant kernel, timed live INSIDE a step: HIP events on the launch stream around each of the 18 trunk-conv
        # launches of two extra eager steps (the graph-replayed steps above run exactly these kernels).  The
        # stand-alone back-to-back figure is logged too: 20 MFMA-bound launches in a row run at a lower sustained clock.
        from pix2pixhdaudiosr_amd import _ops
        if a.no_probes:                                            # byte / launch counting runs: nothing beyond the steps themselves
            print(json.dumps(out), file=_JSON_OUT or sys.stdout, flush=True)
            if dist_on:
                torch.distributed.barrier()
                torch.distributed.destroy_process_group()
            return
        tch, trows, tcols = wl["trunk"]
        sec_iso, flops = time_trunk_conv(a.batch, trunk=wl["trunk"])
        sec = sec_iso
        sec_dgrad = None
        if world == 1 and not a.no_probes:                         # extra steps on one rank only would desynchronise the collectives
            import ctypes as C
            L = _ops.lib_for(HALF)

            def probe(pad_mode, esize):
                # events around the gconv launches of the trunk layer only (cin pitch 768, GEMM-K 9 * 768, 32 x 16 grid), on
                # their launch stream, inside two extra eager steps.  The forward (reflect gather, pad_mode 1) and the
                # input gradient (reflect adjoint, pad_mode 2) run on the same grid and are told apart by the gather's
                # padding rule; `esize` tells the e4m3 forward of --fp8 from bf16 launches
                _ops.check(L.p2phd_probe_gconv_ex(1, tch, 9 * tch, trows, tcols, pad_mode, esize))
                for _ in range(2):
                    model.train_step(lr, hr)
                torch.cuda.synchronize()
                buf = (C.c_float * 4096)()
                n_ev = L.p2phd_probe_read(buf, 4096)
                _ops.check(L.p2phd_probe_gconv_ex(0, 0, 0, 0, 0, -1, 0))
                return (sum(buf[i] for i in range(n_ev)) / n_ev / 1e3, n_ev) if n_ev else (None, 0)

            got, n_ev = probe(1, 1 if a.fp8 else 2)
            if got is not None:
                sec = got
                log(f"trunk conv FORWARD inside the step: {sec * 1e6:.1f} us/launch over {n_ev} launches "
                    f"(stand-alone conv_fwd incl. statistics merge, back-to-back: {sec_iso * 1e6:.1f} us)")
            sec_dgrad, n_dg = probe(3, 2)                          # (round 4: the adjoint reads the reflection extras, gather rule 3)
            if sec_dgrad is None:
                sec_dgrad, n_dg = probe(2, 2)
            if sec_dgrad is not None:
                log(f"trunk conv INPUT GRADIENT inside the step: {sec_dgrad * 1e6:.1f} us/launch over {n_dg} launches")
        log(f"trunk conv {sec * 1e6:.1f} us/launch")
        out["roofline"] = {"bound": "mfma", "achieved": flops / sec / 1e12, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": flops / sec / 1e12 / BF16_DENSE_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                           "kernel": "gconv_kernel<bf16,256,192,HALO> implicit-GEMM Conv3x3 768->768 @32x16, forward (residual trunk, 18 of 28 generator convs)"
                                     if wl["headline"] else f"gconv_kernel<bf16> implicit-GEMM Conv3x3 {tch}->{tch} @{trows}x{tcols}, forward (residual trunk of the global generator)",
                           "launch_us": sec * 1e6, "flops_per_launch": flops,
                           "launches_timed": "forward launches only (gather pad_mode 1); the same-shaped input-gradient launches: dgrad_launch_us",
                           "dgrad_launch_us": None if sec_dgrad is None else sec_dgrad * 1e6}
        if a.fp8 and world == 1:
            # the probed launches are the e4m3 forward of the trunk on the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (unit
            # e8m0 scales; twice the bf16 rate, MI355X_MICROARCH.md matrix-core table): priced against the 5 PF dense fp8 peak;
            # the recorded HBM traffic (taken on the bf16 kernel) does not apply
            out["roofline"].update({"peak": FP8_DENSE_PEAK_TFLOPS, "frac": flops / sec / 1e12 / FP8_DENSE_PEAK_TFLOPS,
                                    "traffic": None, "traffic_source": None,
                                    "peak_note": "block-scaled e4m3 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4): priced against the 5 PF dense fp8 peak",
                                    "kernel": "gconv_kernel<fp8 e4m3> implicit-GEMM Conv3x3 768->768 @32x16 (forward of the residual trunk)"})
        if world == 1 and not a.no_probes and wl["headline"]:
            try:
                tg = time_generator(model, a.batch)
                gf = 6 * M_G * a.batch
                out["config"].update({"G_fwd_bwd_ms": tg * 1e3, "G_fwd_bwd_tflops": gf / tg / 1e12,
                                      "G_fwd_bwd_frac_of_bf16_peak": gf / tg / 1e12 / BF16_DENSE_PEAK_TFLOPS,
                                      "G_fwd_bwd_how": "GlobalGenerator forward + backward alone (all weight gradients), graph replay, HIP "
                                                       "events; 6 M_G FLOP per sample, convs only (InstanceNorm passes are in the time)"})
                log(f"generator fwd+bwd alone: {tg * 1e3:.2f} ms = {gf / tg / 1e12:.0f} TFLOP/s")
            except Exception as e:                                 # a measurement extra: never lose the headline line to it
                log(f"generator-only timing failed: {type(e).__name__}: {e}")
        if world == 1 and not a.no_mdct and wl["headline"]:
            out["mdct"] = time_mdct(a.batch)
            log("mdct alone: " + ", ".join(f"{k} {v['us']:.1f} us {v['GB_per_s']:.0f} GB/s" for k, v in out["mdct"].items()))
        if world == 1 and not a.no_cpu_baseline and wl["headline"]:
            log("cpu baseline (oracle, batch 2: configs[1] and configs[0] networks, 1 + 3 steps each) ...")
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), file=_JSON_OUT or sys.stdout, flush=True)
    if dist_on:
        torch.distributed.barrier()                               # rank 0 is still measuring its dominant kernel
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
