"""The host thread pool follows the container's CPU quota (neat_amd.cap_host_threads, called by the GPU step drivers): with torch's
default (one thread per machine core) a 16-CPU container on a 256-core host throttles itself on every CPU random draw of the
sampler.  Importing the package alone changes nothing."""
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


def test_thread_pool_is_capped_on_request_unless_the_user_decides():
    code = ("import torch; t0 = torch.get_num_threads(); import neat_amd; t1 = torch.get_num_threads(); neat_amd.cap_host_threads(); "
            "print(t0, t1, torch.get_num_threads(), neat_amd.cpu_quota())")
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "NEAT_NO_THREAD_CAP")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    run = lambda e: list(map(int, subprocess.check_output([sys.executable, "-c", code], env=e, stderr=subprocess.DEVNULL).split()))
    t0, t1, t2, quota = run(env)
    assert t1 == t0, "importing the package must not touch torch's thread pool"
    assert t2 <= max(1, min(8, quota // 4)) or t2 == 1 or t2 == t0 <= 8
    t0, t1, t2, _ = run(dict(env, OMP_NUM_THREADS="3"))
    assert (t0, t1, t2) == (3, 3, 3)
    t0, t1, t2, _ = run(dict(env, NEAT_NO_THREAD_CAP="1"))
    assert t2 == t0
