"""The host thread pool follows the container's CPU quota (neat_amd/__init__.py): with torch's default (one thread per machine
core) a 16-CPU container on a 256-core host throttles itself on every CPU random draw of the sampler."""
import os
import subprocess
import sys


def test_cpu_quota_is_bounded_by_affinity_and_cgroup():
    import neat_amd
    q = neat_amd.cpu_quota()
    assert 1 <= q <= len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            assert q <= max(1, int(quota) // int(period))
    except OSError:
        pass


def test_thread_pool_is_capped_unless_the_user_decides():
    code = "import torch, neat_amd; print(torch.get_num_threads(), neat_amd.cpu_quota())"
    env = {k: v for k, v in os.environ.items() if k != "OMP_NUM_THREADS"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    threads, quota = map(int, subprocess.check_output([sys.executable, "-c", code], env=env).split())
    assert threads <= max(1, min(8, quota // 4)) or threads == 1
    env["OMP_NUM_THREADS"] = "3"
    threads, _ = map(int, subprocess.check_output([sys.executable, "-c", code], env=env).split())
    assert threads == 3
