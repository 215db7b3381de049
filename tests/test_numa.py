"""Host placement next to a device (zune-jpeg_amd/csrc/zj_numa.cpp, round 6; SURVEY.md 8e "one host thread per GPU ... per-GPU
PCIe links").  CPU half: the sysfs parsing and the binding rules against a small fake tree (ZJ_SYSFS_ROOT), in child processes so
that no affinity leaks into the test run.  GPU half: the device's node from its PCI bus id, zj_multi's slot threads and
zj_pool's slot threads bound to it, ZJ_NUMA=off leaving everything alone."""
import importlib
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(code, env=None, prefix=()):
    e = dict(os.environ)
    e.pop("ZJ_NUMA", None)
    e.update(env or {})
    r = subprocess.run(list(prefix) + [sys.executable, "-c", "import sys, importlib\nsys.path.insert(0, %r)\nzj = importlib.import_module('zune-jpeg_amd')\nL = zj.lib()\n" % ROOT + textwrap.dedent(code)],
                       capture_output=True, text=True, env=e, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()


def _fake_sysfs(tmp_path, lists):
    for node, cpus in lists.items():
        d = tmp_path / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    return str(tmp_path)


def _cpus():
    c = sorted(os.sched_getaffinity(0))
    if len(c) < 4:
        pytest.skip("needs four CPUs")
    return c


def test_bind_to_node_uses_the_nodes_cpus_within_the_initial_affinity(tmp_path):
    c = _cpus()
    half = len(c) // 2
    root = _fake_sysfs(tmp_path, {0: ",".join(map(str, c[:half])), 1: f"{c[half]}-{c[-1]}" if c[-1] - c[half] == len(c) - half - 1 else ",".join(map(str, c[half:]))})
    out = _child("""
        import os
        print(L.zj_bind_thread_to_numa_node(1), sorted(os.sched_getaffinity(0)), L.zj_thread_numa_node())
        print(L.zj_bind_thread_to_numa_node(0), sorted(os.sched_getaffinity(0)), L.zj_thread_numa_node())
        print(L.zj_bind_thread_to_numa_node(7))
        """, {"ZJ_SYSFS_ROOT": root})
    assert out[0] == f"{len(c) - half} {c[half:]} 1"
    assert out[1] == f"{half} {c[:half]} 0"     # (rebinding goes by the affinity the process STARTED with, not the current one)
    assert out[2] == "-1"                        # no such node: not bound


def test_numa_off_and_a_confined_process_are_left_alone(tmp_path):
    c = _cpus()
    root = _fake_sysfs(tmp_path, {0: str(c[0]), 1: ",".join(map(str, c[1:]))})
    out = _child("import os\nprint(L.zj_bind_thread_to_numa_node(1), len(os.sched_getaffinity(0)))", {"ZJ_SYSFS_ROOT": root, "ZJ_NUMA": "off"})
    assert out[0] == f"-1 {len(c)}"
    # started under `taskset` on node 0's only CPU: node 1 has nothing the process may use -> not bound, affinity as it was
    out = _child("import os\nprint(L.zj_bind_thread_to_numa_node(1), sorted(os.sched_getaffinity(0)))\nprint(L.zj_bind_thread_to_numa_node(0))",
                 {"ZJ_SYSFS_ROOT": root}, prefix=("taskset", "-c", str(c[0])))
    assert out[0] == f"-1 {[c[0]]}" and out[1] == "1"


def test_without_a_device_nothing_is_known():
    if importlib.import_module("zune-jpeg_amd").device_count() > 0:
        pytest.skip("a device is present")
    out = _child("print(L.zj_device_numa_node(0), L.zj_bind_thread_near_device(0))")
    assert out[0] == "-1 -1"


# ---- GPU box --------------------------------------------------------------------------------------------------------------
def _device_node_from_sysfs(zj):
    import ctypes as C
    buf = C.create_string_buffer(64)
    assert zj.lib().zj_device_pci_bus_id(0, buf, 64) == 0
    pci = buf.value.decode().lower()
    try:
        return int(open(f"/sys/bus/pci/devices/{pci}/numa_node").read()), pci
    except OSError:
        return -1, pci


@pytest.mark.gpu
def test_device_node_and_slot_threads_on_the_gpu_box():
    zj = importlib.import_module("zune-jpeg_amd")
    node, pci = _device_node_from_sysfs(zj)
    assert len(pci.split(":")) == 3
    assert zj.device_numa_node(0) == (node if node >= 0 else -1)
    m = zj.Multi([0, 0])
    try:
        for dev_node, thread_node, bound in m.slot_numa():
            assert dev_node == zj.device_numa_node(0)
            if dev_node >= 0:
                assert bound and thread_node == dev_node   # the slot's host thread runs on its GPU's socket
            else:
                assert not bound
    finally:
        m.close()
    with zj.Pool(threads=3, devices=[0, 0]) as pool:
        for dev_node, n_bound, n_threads in pool.slot_numa():
            assert n_threads == 3 + 3 and dev_node == zj.device_numa_node(0)   # three entropy workers + three submitters per slot
            assert n_bound == (n_threads if dev_node >= 0 else 0)
    # a process-per-GPU rank: the calling thread, and what it starts afterwards
    out = _child("""
        import threading
        print(zj.bind_thread_near_device(0), zj.thread_numa_node())
        r = []
        t = threading.Thread(target=lambda: r.append(zj.thread_numa_node())); t.start(); t.join()
        print(r[0])
        """)
    if node >= 0:
        assert out[0] == f"{node} {node}" and out[1] == str(node)
    else:
        assert out[0].startswith("-1")


@pytest.mark.gpu
def test_numa_off_binds_nothing_on_the_gpu_box():
    out = _child("""
        import os
        before = len(os.sched_getaffinity(0))
        m = zj.Multi([0])
        print(m.slot_numa()[0][2], zj.bind_thread_near_device(0), len(os.sched_getaffinity(0)) == before)
        m.close()
        with zj.Pool(threads=2) as pool:
            print(pool.slot_numa()[0][1])
        """, {"ZJ_NUMA": "off"})
    assert out[0] == "False -1 True" and out[1] == "0"


@pytest.mark.gpu
def test_freed_pinned_blocks_are_handed_out_again():
    """zj_free_pinned keeps blocks for later requests of at least half their size (zj_api.cpp: a decoder made per file with
    pinned planes would otherwise pin 0.17 ms per MB and unpin 0.09 per file); ZJ_PINNED_CACHE_MB=0 turns that off.  Each
    setting in a fresh process: the limit is read once."""
    code = r"""
import ctypes as C, importlib, sys
sys.path.insert(0, %r)
zj = importlib.import_module("zune-jpeg_amd")
L = zj.lib()
L.zj_alloc_pinned.restype = C.c_void_p; L.zj_alloc_pinned.argtypes = [C.c_size_t]; L.zj_free_pinned.argtypes = [C.c_void_p]
ctx = zj.Context()
a = L.zj_alloc_pinned(48 << 20); C.memset(a, 7, 48 << 20); L.zj_free_pinned(a)
b = L.zj_alloc_pinned(40 << 20)            # at least half of 48 MB: the same block
c = L.zj_alloc_pinned(40 << 20)            # nothing cached any more: a new one
L.zj_free_pinned(b)
d = L.zj_alloc_pinned(8 << 20)             # less than half of 48 MB: not that block
print(int(a == b), int(c != b and c is not None), int(d != b))
for p in (c, d): L.zj_free_pinned(p)
o = zj.ZuneJpegOptions(); o.pinned_planes = True
data = open(%r, "rb").read()
outs = []
for _ in range(3):                          # decoders made per file share the planes' pinned blocks; the pixels do not change
    dec = zj.Decoder(o, ctx); outs.append(dec.decode_buffer(data).tobytes()); dec.close()
print(int(outs[0] == outs[1] == outs[2]))
""" % (ROOT, os.path.join(ROOT, "tests", "golden", "test-baseline.jpg"))
    for env, want in (({}, "1 1 1"), ({"ZJ_PINNED_CACHE_MB": "0"}, None)):
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.strip().splitlines()
        assert lines[-1] == "1", lines
        if want:
            assert lines[-2] == want, lines
