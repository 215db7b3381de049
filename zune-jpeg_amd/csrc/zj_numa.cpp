// zj_numa.cpp -- where a device slot's host side should live on a multi-socket node ("one host thread (or
// process) per GPU ... per-GPU PCIe links").  The 8-GPU MI355X node is two sockets with four GPUs behind each: a slot's
// feeder threads, the pinned planes they fill and the GPU that DMAs them belong on one socket.  This file answers "which
// NUMA node is device N on" (PCI bus id -> sysfs) and binds threads there; zj_pool / zj_multi use it for their slot threads,
// bench.py's ranks and C callers through the functions below.  The ROCm runtime already places hipHostMalloc memory on the
// node of the CURRENT device (profiles/r06_feeder_ab.txt: a thread on socket 1 gets pages on socket 0 for a GPU on socket 0),
// so the planes follow zj_set_thread_device; what was missing is the threads.
//
//   ZJ_NUMA=off (or 0)   no binding anywhere: the behaviour before round 6
//   ZJ_SYSFS_ROOT        another root than /sys (tests build a small tree)
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <ctype.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/zjhip.h"

namespace {

cpu_set_t g_initial;       // the affinity the process was started with: a `taskset` around the application is respected
bool g_have_initial = false;
__attribute__((constructor)) void capture_initial_affinity()
{
    CPU_ZERO(&g_initial);
    g_have_initial = sched_getaffinity(0, sizeof g_initial, &g_initial) == 0;
}

std::string sysfs_root()
{
    const char* e = getenv("ZJ_SYSFS_ROOT");
    return e && *e ? std::string(e) : std::string("/sys");
}

bool numa_enabled()
{
    const char* e = getenv("ZJ_NUMA");
    return !(e && (!strcmp(e, "off") || !strcmp(e, "0")));
}

bool read_small(const std::string& path, char* buf, size_t cap)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    const size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return n > 0;
}

// "0-63,128-191" -> set; false if nothing parsed
bool parse_cpulist(const char* s, cpu_set_t* set)
{
    CPU_ZERO(set);
    bool any = false;
    while (*s) {
        while (*s && !isdigit((unsigned char)*s)) s++;
        if (!*s) break;
        char* end;
        long a = strtol(s, &end, 10), b = a;
        s = end;
        if (*s == '-') { b = strtol(s + 1, &end, 10); s = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) { if (c >= 0) { CPU_SET((int)c, set); any = true; } }
    }
    return any;
}

bool node_cpus(int node, cpu_set_t* set)
{
    char buf[4096];
    if (node < 0 || !read_small(sysfs_root() + "/devices/system/node/node" + std::to_string(node) + "/cpulist", buf, sizeof buf)) return false;
    return parse_cpulist(buf, set);
}

} // namespace

extern "C" {

int zj_device_numa_node(int device)
{
    char id[64];
    if (zj_device_pci_bus_id(device, id, sizeof id) != ZJ_OK) return -1;
    for (char* p = id; *p; p++) *p = (char)tolower((unsigned char)*p);
    char buf[64];
    if (!read_small(sysfs_root() + "/bus/pci/devices/" + id + "/numa_node", buf, sizeof buf)) return -1;
    const int node = atoi(buf);
    return node >= 0 ? node : -1; // (-1: a single-node machine, or firmware that does not say)
}

int zj_bind_thread_to_numa_node(int node)
{
    if (!numa_enabled()) return -1;
    cpu_set_t want;
    if (!node_cpus(node, &want)) return -1;
    if (g_have_initial) {
        cpu_set_t both;
        CPU_AND(&both, &want, &g_initial);
        if (CPU_COUNT(&both) == 0) return -1; // the process was confined elsewhere on purpose
        want = both;
    }
    // (a cpuset cgroup narrows this further by itself; EINVAL if nothing is left)
    if (sched_setaffinity(0, sizeof want, &want) != 0) return -1;
    return CPU_COUNT(&want);
}

int zj_bind_thread_near_device(int device)
{
    const int node = zj_device_numa_node(device);
    if (node < 0) return -1;
    return zj_bind_thread_to_numa_node(node) > 0 ? node : -1;
}

int zj_thread_numa_node(void)
{
    const int cpu = sched_getcpu();
    if (cpu < 0) return -1;
    for (int node = 0; node < 64; node++) {
        cpu_set_t set;
        if (!node_cpus(node, &set)) { if (node > 8) break; else continue; }
        if (cpu < CPU_SETSIZE && CPU_ISSET(cpu, &set)) return node;
    }
    return -1;
}

} // extern "C"
