"""CPU: `python bench.py --gpus N` must produce the N-rank line BY ITSELF -- the driver invokes it that way.  With
WORLD_SIZE unset and N > 1 bench.py starts N fresh child ranks under torch.distributed.run (before anything touches a GPU)
and relays rank 0's JSON; launched by torch.distributed.run directly it checks --gpus against WORLD_SIZE.  Rehearsed here
over gloo with ASR_BENCH_SPAWN_TEST=1 (rendezvous + one all-reduce, no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["ASR_BENCH_SPAWN_TEST"] = "1"
    return env


def test_bench_gpus_n_spawns_n_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["sum_of_ones"] == 2.0


def test_bench_refuses_world_size_mismatch():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29911")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
