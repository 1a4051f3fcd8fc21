"""world_size-2 data-parallel path on CPU (gloo): the product's gradient exchange (parallel_state.all_reduce_flat_async,
batch sharding, sum / world scaling) driven with the CPU oracle as the compute, checked against a single-process run.

What must hold: every rank ends a step with identical weights; those weights equal what one process gets
from the two shards' gradients averaged; shards normalise their own spectrograms (per-shard min/max), exactly as the
reference's nn.DataParallel replicas do."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_opt():
    from oracle import model as OM
    return OM.default_opt(n_fft=64, hop_length=32, win_length=64, ngf=4, netG="global", n_downsample_global=2,
                          n_blocks_global=1, ndf=4, n_layers_D=2, num_D=2)


def _flatten(d):
    return torch.cat([v.reshape(-1) for v in d.values()])


def _unflatten(flat, like):
    out, o = {}, 0
    for k, v in like.items():
        out[k] = flat[o:o + v.numel()].view(v.shape).clone()
        o += v.numel()
    return out


def _shard_grads(opt, pG, pD, hr, lr, noise, w):
    from oracle import model as OM
    with torch.no_grad():
        hr_s, _, _ = OM.to_spectro(hr, opt, w, mask=False)
        lr_s, _, _ = OM.to_spectro(lr, opt, w, mask=True, noise=noise)
    _, gG, gD = OM.step_grads(pG, pD, lr_s, hr_s, opt)
    return gG, gD


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import model as OM, mdct4 as M4
    from pix2pixhdaudiosr_amd import parallel_state as PS
    opt = _tiny_opt()
    w = M4.kbdwin(opt.win_length)
    pG = OM.N.init_params(OM.netG_spec(opt), seed=1)
    pD = OM.N.init_params(OM.netD_spec(opt), seed=2)
    hr, lr, noise = OM.synthetic_batch(4, opt, seed=5)
    a, b = PS.shard_batch(4, rank, world)
    sG, sD = {}, {}
    for _ in range(2):                                              # two optimisation steps
        gG, gD = _shard_grads(opt, pG, pD, hr[a:b], lr[a:b], noise[a:b], w)
        flatG, flatD = _flatten(gG), _flatten(gD)
        hG = PS.all_reduce_flat_async(flatG)                         # G exchange runs beside ...
        hD = PS.all_reduce_flat_async(flatD)                         # ... the D one, as in train_step
        hG.wait(); hD.wait()
        pG = OM.adam_step(dict(pG), _unflatten(flatG / world, gG), sG, opt.lr, opt.beta1)
        pD = OM.adam_step(dict(pD), _unflatten(flatD / world, gD), sD, opt.lr, opt.beta1)
    # the replica-consistency check bench.py prints (round 5): the product function over gloo, on stand-ins for the two
    # optimisers' flat master weights; then its sensitivity -- one flipped low bit on one rank
    from types import SimpleNamespace as NS
    fake = NS(optimizer_G=NS(flat_p=_flatten(pG)), optimizer_D=NS(flat_p=_flatten(pD)))
    spread = PS.replica_checksum_spread(fake)
    if rank == 1:
        fake.optimizer_D.flat_p.view(torch.int32)[5] ^= 1
    poked = PS.replica_checksum_spread(fake)
    torch.save({"G": _flatten(pG), "D": _flatten(pD), "spread": spread, "poked": poked}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import model as OM, mdct4 as M4
    torch.set_num_threads(1)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["G"], r1["G"]) and torch.equal(r0["D"], r1["D"])          # replicas stay identical
    assert r0["spread"] == r1["spread"] == {"G": 0, "D": 0}
    assert r0["poked"] == r1["poked"] and r0["poked"]["G"] == 0 and r0["poked"]["D"] != 0
    # single-process emulation: average of the two shards' gradients (each shard normalised on its own)
    opt = _tiny_opt()
    w = M4.kbdwin(opt.win_length)
    pG = OM.N.init_params(OM.netG_spec(opt), seed=1)
    pD = OM.N.init_params(OM.netD_spec(opt), seed=2)
    hr, lr, noise = OM.synthetic_batch(4, opt, seed=5)
    sG, sD = {}, {}
    for _ in range(2):
        acc = None
        for (a, b) in ((0, 2), (2, 4)):
            gG, gD = _shard_grads(opt, pG, pD, hr[a:b], lr[a:b], noise[a:b], w)
            cur = (_flatten(gG), _flatten(gD))
            acc = cur if acc is None else (acc[0] + cur[0], acc[1] + cur[1])
        pG = OM.adam_step(dict(pG), _unflatten(acc[0] / 2, gG), sG, opt.lr, opt.beta1)
        pD = OM.adam_step(dict(pD), _unflatten(acc[1] / 2, gD), sD, opt.lr, opt.beta1)
    # Conv biases in front of an InstanceNorm have an exactly-zero true gradient: theirs is rounding noise, Adam's
    # sign-like first steps turn it into +-lr moves, and the worker processes use a different thread count than this
    # one.  Everything that carries signal (the weights) must agree.
    torch.set_num_threads(1)
    for tag, ref in (("G", pG), ("D", pD)):
        got = _unflatten(r0[tag], ref)
        for k in ref:
            if k.endswith(".weight"):
                assert float(((got[k] - ref[k]).abs() > 1e-6).float().mean()) < 2e-3, (tag, k)


def test_shard_batch():
    from pix2pixhdaudiosr_amd import parallel_state as PS
    assert [PS.shard_batch(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    with pytest.raises(AssertionError):
        PS.shard_batch(7, 0, 2)
