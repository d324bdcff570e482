"""How many host cores this process may really use, and keeping torch's CPU thread pool inside that.

An MI355X host shows every core to a container (256 on the measured boxes) while the container's cgroup grants a CPU-time quota
(16 cores' worth per GPU there).  Two things follow for the code either side of the hot path:

* thread and worker counts must come from the QUOTA, not from the visible cores;
* torch's intra-op pool defaults to one OpenMP thread per visible physical core (128), and after every parallel CPU op (a `clone`, a
  `copy_` of >32 K elements, `pin_memory`) all of them spin for a few milliseconds before sleeping (libgomp's wait policy).  One 11 MB
  `clone` costs ~0.7-0.9 CPU-seconds that way (tools/jpeg_host_probe.py --gpu, profiles/r04_jpeg_host_probe.txt), which is most of a
  second of a 16-core quota: the cgroup then throttles EVERY thread of the container -- the loader workers, the entropy-decode threads
  and the thread that launches GPU work -- for the rest of the 100 ms accounting period.  `limit_torch_threads()` caps the pool at the
  quota; the pipelines call it once before they start.
"""
from __future__ import annotations

import math
import os


def _cgroup_quota() -> float | None:
    """CPU quota of this container in cores (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`), or None when unlimited."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return int(quota) / int(period)
        return None
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def cpu_budget() -> int:
    """Cores this process can keep busy: the smaller of its affinity set and its cgroup quota (at least 1)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cgroup_quota()
    if quota is not None:
        n = min(n, max(1, math.floor(quota)))
    return max(1, n)


def limit_torch_threads() -> int:
    """Cap torch's intra-op CPU pool at `cpu_budget()` (never raises it).  Returns the thread count in force afterwards."""
    import torch
    budget = cpu_budget()
    if torch.get_num_threads() > budget:
        torch.set_num_threads(budget)
    return torch.get_num_threads()
