"""Two data-parallel ranks of the product step on ONE GPU (gloo between them): the flat-buffer gradient exchange inside
`train_step` and between the replays of `train_step_graphed`.  Replicas must end bit-identical; the
data-parallel result must equal a single process on the concatenated batch up to fp32 / bf16 rounding noise."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, out_dir):
    port = 29600 + (os.getpid() + {"eager": 0, "graphed": 7, "graphed_wire_bf16": 13}[mode]) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_dp_gpu_worker.py"), str(out_dir), mode]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("mode", ["eager", "graphed", "graphed_wire_bf16"])
def test_two_ranks_one_gpu(tmp_path, mode):
    """(`graphed_wire_bf16`, round 4: the generator's buckets travel as bf16 and are widened back into the fp32 gradient
    buffer before Adam -- the replicas must still end bit-identical.)"""
    _run(mode, tmp_path)
    r0 = torch.load(tmp_path / f"{mode}_rank0.pt")
    r1 = torch.load(tmp_path / f"{mode}_rank1.pt")
    assert r0["steps"] == r1["steps"] == 5
    assert r0["graphed"] == mode.startswith("graphed")
    assert torch.equal(r0["G"], r1["G"]) and torch.equal(r0["D"], r1["D"])      # replicas stay bit-identical
    assert r0["spread"] == r1["spread"] == {"G": 0, "D": 0}                     # ... and the checksum bench.py prints says so
    assert r0["spread_after_a_poke"]["G"] != 0 and r0["spread_after_a_poke"]["D"] == 0   # sensitivity: one flipped bit on one rank
    # bucketed exchange: the generator gradients left in 4 all-reduces, last layers first (the order the backward
    # produces them), contiguous, non-overlapping, covering the whole flat buffer; D in one
    for r in (r0, r1):
        bk = r["buckets_G"]
        assert len(bk) == 4 and bk[0][1] == r["total_G"] and bk[-1][0] == 0, bk
        assert all(a < b for a, b in bk) and all(bk[i][0] == bk[i + 1][1] for i in range(3)), bk
        assert r["buckets_D"] == [(0, r["total_D"])]
    assert torch.isfinite(r0["G"]).all() and torch.isfinite(r0["D"]).all()
    for step_losses in r0["losses"]:
        assert all(abs(v) < 1e3 for v in step_losses.values())
    if mode == "eager":
        s = torch.load(tmp_path / "single.pt")
        # losses of the first step: mean over the full batch == mean of the two shard means
        r1l = r1["losses"][0]
        for k, v in s["losses"].items():
            dp = 0.5 * (r0["losses"][0][k] + r1l[k])
            assert abs(dp - v) <= 1e-4 * max(1.0, abs(v)), (k, dp, v)
        # weights after ONE update: the first Adam step moves every element by lr * sign(grad), so DP and single process
        # differ (by 2 lr) only where rounding flipped the sign of a near-zero gradient
        for k in ("G", "D"):
            d = (r0["first"][k] - s[k]).abs()
            assert float(d.max()) <= 4.5e-4, (k, float(d.max()))
            assert float((d > 1e-5).float().mean()) < 0.05, (k, float((d > 1e-5).float().mean()))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (before touching the GPU) and
    print ONE JSON line with n_gpus == 2.  Rehearsal environment: both ranks on the box's one GPU, gloo between them."""
    import json
    env = dict(os.environ, P2PHD_DIST_BACKEND="gloo", P2PHD_FORCE_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dist"]["ranks"] == 2 and len(out["per_rank_ms_per_step"]) == 2
    # exchange-exposure instrumentation (round 3): per-bucket bytes, waits per step, exposed time on the compute stream
    d = out["dist"]
    assert len(d["bucket_bytes"]["G"]) == 4 and sum(d["bucket_bytes"]["G"]) == 4 * sum(d["g_gradient_buckets"])
    assert len(d["bucket_bytes"]["D"]) == 1 and d["bucket_bytes"]["D"][0] > 0
    assert d["waits_per_step"] == 2 and d["exposed_exchange_ms"] >= 0 and d["host_wait_ms"] >= 0
    assert abs(d["exposed_exchange_ms"] - d["exposed_exchange_ms_G"] - d["exposed_exchange_ms_D"]) < 1e-6
    assert d["replica_checksum_spread"] == {"G": 0, "D": 0}, d["replica_checksum_spread"]     # the run says itself that the replicas agree
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["launch"] == "hip-graph replay" and "graph_error" not in out
    assert out["value"] > 0 and out["scaling"] == "weak"



@pytest.mark.parametrize("comm_cus", [0, 8])
def test_bench_step_against_real_rccl_with_one_rank(comm_cus):
    """The N-rank call sequence (4 gradient buckets all-reduced between graph replays, D exchange, waits, Adam) against
    the real RCCL backend: a one-rank 'nccl' group with the collectives forced on (P2PHD_REHEARSE_RCCL) -- what a
    one-GPU box can exercise of the path the driver's multi-GPU bench takes.  Second case: the same with the step on a
    CU-masked stream (--comm-cus 8: 8 CUs left to the RCCL kernels)."""
    import json
    env = dict(os.environ, P2PHD_REHEARSE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "P2PHD_DIST_BACKEND", "P2PHD_FORCE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "2", "--no-cpu-baseline",
           "--no-mdct", "--comm-cus", str(comm_cus)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["dist"]["backend"] == "nccl" and out["dist"]["rehearsal_one_rank"] is True
    # comm_cus > 0: the step ran (and was captured / replayed) on a stream whose CU mask leaves that many CUs to RCCL
    assert out["dist"]["comm_cus"] == comm_cus and out["dist"]["ranks"] == 1
    assert out["dist"]["waits_per_step"] == 2 and out["dist"]["exposed_exchange_ms"] >= 0
    assert out["dist"]["replica_checksum_spread"] == {"G": 0, "D": 0}
    assert len(out["dist"]["bucket_bytes"]["G"]) == 4
    assert len(out["dist"]["g_gradient_buckets"]) == 4 and min(out["dist"]["g_gradient_buckets"]) > 0
    assert out["config"]["launch"] == "hip-graph replay" and "graph_error" not in out and out["value"] > 0
