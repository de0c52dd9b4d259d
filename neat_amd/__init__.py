"""neat_amd: the NEAT hot path (rays -> depth samples -> SDF/heads -> compositing -> junction block -> loss -> Adam) on MI355X."""
import os


def cpu_quota():
    """CPUs this process may actually use: the cgroup-v2 quota (cpu.max) if there is one, else the affinity mask."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            avail = min(avail, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return avail


_capped = False


def cap_host_threads():
    """torch sizes its intra-op pool by the machine's core count (128 on the MI355X hosts), not by the container's CPU quota (16).
    The host side of the hot path only draws a few hundred KB of CPU randoms per step; with a 128-thread pool every such draw leaves
    OpenMP workers spinning, the cgroup runs out of quota and the kernel parks the whole process -- including the thread that feeds
    the GPU -- for the rest of the 100 ms period (measured: every other train step with the sampler on took 90 ms instead of 6).
    Called by the GPU step drivers (train.Trainer on a CUDA device, neat_amd.runner), NOT at import: a host application that only
    imports the package keeps its thread pool.  Unless OMP_NUM_THREADS is set or NEAT_NO_THREAD_CAP=1, the pool is lowered (never
    raised) to a quarter of this rank's share of the quota, at most 8 threads; said once on stderr."""
    global _capped
    if _capped or "OMP_NUM_THREADS" in os.environ or os.environ.get("NEAT_NO_THREAD_CAP") == "1":
        return
    _capped = True
    import torch
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))      # torchrun: the ranks of this node share the quota
    want = max(1, min(8, cpu_quota() // (4 * ranks)))
    have = torch.get_num_threads()
    if have > want:
        torch.set_num_threads(want)
        import sys
        print(f"[neat_amd] torch intra-op threads {have} -> {want} (CPU quota {cpu_quota()}; OMP_NUM_THREADS or NEAT_NO_THREAD_CAP=1 to keep)",
              file=sys.stderr, flush=True)
