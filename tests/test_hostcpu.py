"""Host CPU budget (concepthash_amd/hostcpu.py): the cgroup quota, not the visible cores, sizes thread pools; torch's pool is capped."""
import builtins
import io
import os

import torch

from concepthash_amd import hostcpu


def _fake_open(files):
    real = builtins.open

    def opener(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise FileNotFoundError(path)
            return io.StringIO(files[path])
        if str(path).startswith("/sys/fs/cgroup"):
            raise FileNotFoundError(path)
        return real(path, *a, **k)
    return opener


def test_quota_v2_v1_and_unlimited(monkeypatch):
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    assert hostcpu._cgroup_quota() == 16.0
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert hostcpu._cgroup_quota() is None
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "250000\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert hostcpu._cgroup_quota() == 2.5
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert hostcpu._cgroup_quota() is None


def test_budget_is_min_of_affinity_and_quota(monkeypatch):
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    monkeypatch.setattr(hostcpu, "_cgroup_quota", lambda: 16.0)
    assert hostcpu.cpu_budget() == 16
    monkeypatch.setattr(hostcpu, "_cgroup_quota", lambda: 0.5)
    assert hostcpu.cpu_budget() == 1
    monkeypatch.setattr(hostcpu, "_cgroup_quota", lambda: None)
    assert hostcpu.cpu_budget() == 256
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: {0, 1, 2})
    monkeypatch.setattr(hostcpu, "_cgroup_quota", lambda: 16.0)
    assert hostcpu.cpu_budget() == 3


def test_limit_torch_threads_only_lowers(monkeypatch):
    before = torch.get_num_threads()
    try:
        monkeypatch.setattr(hostcpu, "cpu_budget", lambda: 10 ** 6)
        assert hostcpu.limit_torch_threads() == before            # never raised
        monkeypatch.setattr(hostcpu, "cpu_budget", lambda: 1)
        assert hostcpu.limit_torch_threads() == 1
    finally:
        torch.set_num_threads(before)
