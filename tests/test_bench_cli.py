"""bench.py command-line contract that needs no GPU: the rank launcher and its argument checks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**kw):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


def test_bench_refuses_mismatched_world():
    """Under a launcher (WORLD_SIZE set) --gpus must agree with it: a silent n_gpus != N line is what round 1 printed."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1"], cwd=ROOT, capture_output=True, text=True,
                       env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_bench_gpus_n_starts_n_ranks_and_propagates_failure():
    """`bench.py --gpus 2` without a launcher starts torch.distributed.run with two ranks as CHILD processes (the parent
    never touches the GPU); without a GPU both ranks fail at set_device and the parent must exit non-zero, not print a
    1-GPU line."""
    import torch
    if torch.cuda.is_available():
        return                                                      # covered by tests/test_gpu_dp.py on the GPU box
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, env=_clean_env(), timeout=600)
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
    err = r.stdout + r.stderr
    assert "local_rank: 1" in err or "rank: 1" in err or "rank      : 1" in err, err[-3000:]
