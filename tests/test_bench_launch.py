"""`python bench.py --gpus N` must start its own ranks (the driver's scaling command has no torchrun in front).  CPU check of
that plumbing: --dry-run = launcher + gloo rendezvous + the step's data-parallel exchange (ONE flat all-reduce of all
1 219 274 gradients), no GPU work; rank 0 prints the one JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks_dry_run():
    line = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-run"], {"NEAT_BENCH_RAYS": "64"})
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["dry_run"] is True
    assert line["allreduce_ok"] is True and line["allreduce_elements"] == 1219274
    assert line["scaling"] == "weak" and line["config"]["rays_per_gpu"] == 64


def test_bench_single_rank_dry_run():
    line = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--dry-run"])
    assert line["n_gpus"] == 1 and line["allreduce_ok"] is True


def test_bench_c4_workload_splits_the_global_batch_dry_run():
    """--workload c4 (BASELINE configs[3]): 4096 rays per step in TOTAL, 2048 per rank at world 2 (512 at 8), strong scaling; DTU model
    (1024 junction latents: 1 465 034... the bucket follows the model's parameter list) and the packed exchange of a replayed step."""
    line = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run", "--workload", "c4"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["allreduce_ok"] is True
    assert line["config"]["rays_per_gpu"] == 2048 and line["config"]["global_rays"] == 4096
    assert line["allreduce_elements"] == 1219274 + (1024 - 64) * 256
