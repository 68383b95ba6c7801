"""CPU: NUMA placement of the chunk farm (sperr_amd/csrc/numa_place.hpp) over a made-up sysfs tree.

A farm worker binds itself -- and with it its helper threads and the staging memory it pins -- to the
CPUs of the NUMA node its device hangs off.  The reference's counterpart is the OpenMP team of its
chunk loop (/root/reference/src/SPERR3D_OMP_C.cpp:94-130, SPERR3D_OMP_D.cpp:101-127), placed by the
OpenMP runtime.  No device is touched here: the probe and the binding take the PCI address as text."""
import ctypes as C
import os
import threading

import pytest


@pytest.fixture(scope="module")
def lib():
    from sperr_amd import api
    lib = api.load_library()
    lib.sperrhip_numa_probe.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_size_t,
                                        C.POINTER(C.c_size_t)]
    lib.sperrhip_numa_bind_self.argtypes = [C.c_char_p, C.c_char_p]
    return lib


def make_tree(root, devices, nodes):
    """devices: {bdf: numa_node text}, nodes: {node: cpulist text}"""
    for bdf, node in devices.items():
        d = root / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(node)
    for node, cpus in nodes.items():
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus)


def probe(lib, root, bdf):
    node, n = C.c_int(-7), C.c_size_t(0)
    cpus = (C.c_int * 4096)()
    assert lib.sperrhip_numa_probe(str(root).encode(), bdf.encode(), C.byref(node), cpus, 4096, C.byref(n)) == 0
    return node.value, list(cpus[:n.value])


def test_probe_reads_node_and_cpulist(lib, tmp_path):
    # an eight-GPU, two-socket node as such machines report it: four devices per socket, SMT siblings
    # of a socket in a second range
    devs = {f"0000:{b:02x}:00.0": ("0" if i < 4 else "1") + "\n"
            for i, b in enumerate((0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5))}
    make_tree(tmp_path, devs, {0: "0-63,128-191\n", 1: "64-127,192-255\n"})
    node, cpus = probe(lib, tmp_path, "0000:05:00.0")
    assert node == 0 and cpus == list(range(0, 64)) + list(range(128, 192))
    # (hipDeviceGetPCIBusId prints hex digits in upper case, sysfs in lower case)
    node, cpus = probe(lib, tmp_path, "0000:F5:00.0")
    assert node == 1 and cpus == list(range(64, 128)) + list(range(192, 256))


def test_probe_unknown_node_or_device(lib, tmp_path):
    make_tree(tmp_path, {"0000:01:00.0": "-1\n", "0000:02:00.0": "3\n", "0000:03:00.0": "junk\n"},
              {0: "0-3\n", 2: "5,7,9-10\n"})
    assert probe(lib, tmp_path, "0000:01:00.0") == (-1, [])       # the platform does not say (a VM)
    assert probe(lib, tmp_path, "0000:02:00.0") == (3, [])        # a node without a cpulist
    assert probe(lib, tmp_path, "0000:03:00.0") == (-1, [])
    assert probe(lib, tmp_path, "0000:99:00.0") == (-1, [])       # no such device
    # single CPUs and ranges mixed
    make_tree(tmp_path / "b", {"0000:01:00.0": "2"}, {2: "5,7,9-10"})
    assert probe(lib, tmp_path / "b", "0000:01:00.0") == (2, [5, 7, 9, 10])


def in_thread(fn):
    out = []
    t = threading.Thread(target=lambda: out.append(fn()))
    t.start()
    t.join()
    return out[0]


def test_bind_narrows_the_calling_thread_only(lib, tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("one CPU: nothing to narrow")
    half = allowed[: len(allowed) // 2]
    # the node also lists CPUs this process may not use (a cpuset): they are left out, not an error
    cpulist = ",".join(str(c) for c in half) + ",60000"
    make_tree(tmp_path, {"0000:0a:00.0": "1", "0000:0b:00.0": "-1", "0000:0c:00.0": "0"},
              {1: cpulist, 0: "60001-60002"})

    def bound(bdf):
        def run():
            n = lib.sperrhip_numa_bind_self(str(tmp_path).encode(), bdf.encode())
            return n, sorted(os.sched_getaffinity(0))   # (pid 0: the calling thread)
        return in_thread(run)

    assert bound("0000:0a:00.0") == (len(half), half)
    assert bound("0000:0b:00.0") == (0, allowed)          # unknown node: left alone
    assert bound("0000:0c:00.0") == (0, allowed)          # no CPU in common with the cpuset: left alone
    assert sorted(os.sched_getaffinity(0)) == allowed      # the test's own thread never moved

    # threads started by a bound worker inherit its mask (the farm's helper threads rely on that)
    def worker():
        lib.sperrhip_numa_bind_self(str(tmp_path).encode(), b"0000:0a:00.0")
        return in_thread(lambda: sorted(os.sched_getaffinity(0)))
    assert in_thread(worker) == half


def test_switch_off(lib, tmp_path, monkeypatch):
    allowed = sorted(os.sched_getaffinity(0))
    make_tree(tmp_path, {"0000:0a:00.0": "0"}, {0: str(allowed[0])})
    monkeypatch.setenv("SPERR_HIP_FARM_NUMA", "0")
    assert in_thread(lambda: (lib.sperrhip_numa_bind_self(str(tmp_path).encode(), b"0000:0a:00.0"),
                              sorted(os.sched_getaffinity(0)))) == (0, allowed)
