"""Host-side sizing: how many cores this process may really use, and this rank's share of them."""
import math
import os


def usable_cpus():
    """min(visible CPUs, scheduler affinity, cgroup CPU quota).  The GPU boxes of this pool show 256 CPUs under a cgroup quota
    of 16 (cpu.max "1600000 100000"): a thread pool as wide as the machine only gets throttled."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def host_threads_per_rank(cap=None):
    """This rank's share of the usable cores: one process per GPU, LOCAL_WORLD_SIZE of them on the node (torch.distributed.run
    exports it; 1 when not launched that way).  At least 1; at most `cap` when given."""
    try:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local_world = 1
    n = max(1, usable_cpus() // local_world)
    return n if cap is None else max(1, min(int(cap), n))
