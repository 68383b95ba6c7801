"""CPU: the farm sizes its host threads by what the process MAY use (sperr_amd/csrc/host_cpus.hpp), over made-up
cgroup trees.

The reference's chunk loop is sized by the caller, 0 meaning omp_get_max_threads()
(/root/reference/src/SPERR3D_OMP_C.cpp:12-20).  The farm starts worker and helper threads of its own; inside a
container the machine's CPU count is the wrong yardstick (the pool's GPU box: 256 shown, cpu.max "1600000 100000").
No device is touched here."""
import ctypes as C
import os

import pytest


@pytest.fixture(scope="module")
def lib():
    from sperr_amd import api
    lib = api.load_library()
    lib.sperrhip_host_cpus.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_double), C.POINTER(C.c_size_t)]
    lib.sperrhip_host_throttle.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.sperrhip_farm_threads.argtypes = [C.c_size_t, C.c_size_t] + [C.POINTER(C.c_size_t)] * 5
    return lib


def cpus(lib, root, proc):
    vis, aff, use = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    quota = C.c_double(-1)
    assert lib.sperrhip_host_cpus(str(root).encode(), str(proc).encode(), C.byref(vis), C.byref(aff), C.byref(quota),
                                  C.byref(use)) == 0
    return vis.value, aff.value, quota.value, use.value


def plan(lib, nthreads, ndev):
    v = [C.c_size_t(0) for _ in range(5)]
    assert lib.sperrhip_farm_threads(nthreads, ndev, *[C.byref(x) for x in v]) == 0
    return dict(zip(("workers", "dec_workers", "helpers", "threads", "usable"), (x.value for x in v)))


def v2_tree(tmp_path, cpu_max, rel="", stat=None, nested_max=None):
    root = tmp_path / "cg"
    (root / rel.lstrip("/")).mkdir(parents=True, exist_ok=True)
    (root / "cpu.max").write_text(cpu_max)
    if nested_max is not None:
        (root / rel.lstrip("/") / "cpu.max").write_text(nested_max)
    if stat is not None:
        (root / rel.lstrip("/") / "cpu.stat").write_text(stat)
    proc = tmp_path / "proc_cgroup"
    proc.write_text(f"0::{rel or '/'}\n")
    return root, proc


def test_quota_of_the_pools_box(lib, tmp_path):
    root, proc = v2_tree(tmp_path, "1600000 100000\n")
    vis, aff, quota, use = cpus(lib, root, proc)
    assert quota == 16.0
    assert aff == len(os.sched_getaffinity(0)) and vis >= 1
    assert use == min(16, aff, vis)


def test_no_limit_and_malformed(lib, tmp_path):
    root, proc = v2_tree(tmp_path, "max 100000\n")
    vis, aff, quota, use = cpus(lib, root, proc)
    assert quota == 0.0 and use == min(vis, aff)
    (root / "cpu.max").write_text("junk\n")
    assert cpus(lib, root, proc)[2] == 0.0
    # no tree at all: the affinity mask alone
    assert cpus(lib, tmp_path / "nowhere", tmp_path / "nofile")[3] == min(vis, aff)


def test_fractional_quota_rounds_up_and_nested_group_takes_the_tightest(lib, tmp_path):
    root, proc = v2_tree(tmp_path, "800000 100000\n", rel="/kubepods/pod1/ctr", nested_max="250000 100000\n")
    _, aff, quota, use = cpus(lib, root, proc)
    assert quota == 2.5 and use == min(3, aff)
    # the ancestor is the tighter one
    (root / "kubepods" / "pod1" / "ctr" / "cpu.max").write_text("max 100000\n")
    (root / "kubepods" / "cpu.max").write_text("150000 100000\n")
    assert cpus(lib, root, proc)[2] == 1.5


def test_cgroup_v1(lib, tmp_path):
    root = tmp_path / "cg"
    d = root / "cpu" / "docker" / "abc"
    d.mkdir(parents=True)
    (d / "cpu.cfs_quota_us").write_text("400000\n")
    (d / "cpu.cfs_period_us").write_text("100000\n")
    (root / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    (root / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    (d / "cpu.stat").write_text("nr_periods 10\nnr_throttled 3\nthrottled_time 7000000\n")
    proc = tmp_path / "proc_cgroup"
    proc.write_text("12:cpuset:/docker/abc\n5:cpu,cpuacct:/docker/abc\n1:name=systemd:/x\n")
    assert cpus(lib, root, proc)[2] == 4.0
    nr, us = C.c_ulonglong(0), C.c_ulonglong(0)
    assert lib.sperrhip_host_throttle(str(root).encode(), str(proc).encode(), C.byref(nr), C.byref(us)) == 0
    assert (nr.value, us.value) == (3, 7000)


def test_cgroup_v1_controller_list_that_starts_with_cpuset(lib, tmp_path):
    """A v1 line whose controller list names cpu BEHIND cpuset ("cpuset,cpu,cpuacct") is the CPU bandwidth controller's
    all the same; "cpuset" alone and a name that merely contains the letters ("cpuxyz") are not."""
    root = tmp_path / "cg"
    d = root / "cpu" / "grp"
    d.mkdir(parents=True)
    (d / "cpu.cfs_quota_us").write_text("300000\n")
    (d / "cpu.cfs_period_us").write_text("100000\n")
    proc = tmp_path / "proc_cgroup"
    proc.write_text("7:cpuset,cpu,cpuacct:/grp\n1:name=systemd:/x\n")
    assert cpus(lib, root, proc)[2] == 3.0
    proc.write_text("7:cpuset:/grp\n6:cpuxyz:/grp\n")
    assert cpus(lib, root, proc)[2] <= 0    # (no quota found)


def test_throttle_counters_v2(lib, tmp_path):
    root, proc = v2_tree(tmp_path, "1600000 100000\n", rel="/pod",
                         stat="usage_usec 5\nuser_usec 1\nnr_periods 100\nnr_throttled 42\nthrottled_usec 123456\n")
    nr, us = C.c_ulonglong(0), C.c_ulonglong(0)
    assert lib.sperrhip_host_throttle(str(root).encode(), str(proc).encode(), C.byref(nr), C.byref(us)) == 0
    assert (nr.value, us.value) == (42, 123456)
    assert lib.sperrhip_host_throttle(str(tmp_path / "nowhere").encode(), str(proc).encode(), C.byref(nr), C.byref(us)) == 1


@pytest.mark.parametrize("quota,ndev", [(16, 1), (16, 8), (4, 8), (256, 8), (2, 1)])
def test_farm_threads_stay_within_the_quota(lib, tmp_path, monkeypatch, quota, ndev):
    """Eight devices x two workers x (1 + helpers) threads on a 16-CPU quota was CFS throttling by construction
    (round 4: helpers sized by hardware_concurrency() = 256)."""
    root, proc = v2_tree(tmp_path, f"{quota * 100000} 100000\n")
    monkeypatch.setenv("SPERR_HIP_CGROUP_ROOT", str(root))
    monkeypatch.setenv("SPERR_HIP_PROC_CGROUP", str(proc))
    for k in ("SPERR_HIP_FARM_WORKERS", "SPERR_HIP_FARM_HELPERS", "SPERR_HIP_FARM_DEC_WORKERS"):
        monkeypatch.delenv(k, raising=False)
    pl = plan(lib, 0, ndev)
    usable = min(quota, len(os.sched_getaffinity(0)))
    assert pl["usable"] == usable
    workers = ndev * pl["workers"]
    assert 1 <= pl["helpers"] <= 12
    # the helpers (the threads that burn CPU: they copy rows) never exceed half the budget, unless one each already does
    assert workers * pl["helpers"] <= max(workers, usable // 2)
    if usable >= 8 * workers:
        assert pl["helpers"] >= 4
    # a caller's own team size is honoured as given, like the reference's set_num_threads
    assert plan(lib, 6 * workers, ndev)["helpers"] == 6
