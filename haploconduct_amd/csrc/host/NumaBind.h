// NumaBind.h — the stage's host threads run on the NUMA node its device hangs on: they copy the overlaps file's text into
// page-locked buffers the HIP runtime placed there, and the FASTQ reader fills the arrays the read store is uploaded from.
#ifndef HC_NUMA_BIND_H_
#define HC_NUMA_BIND_H_
#include <ctype.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../../include/hcedge.h"

namespace hc {

// The CPUs of the NUMA node a device is attached to: /sys/bus/pci/devices/<bus id>/numa_node -> node<k>/cpulist.
// Empty when the machine has one node, the node is unknown, or HC_NUMA=0.
inline std::vector<int> cpus_near_device(int device) {
    std::vector<int> cpus;
    if (const char* e = getenv("HC_NUMA"))
        if (atoi(e) == 0) return cpus;
    char bus[64] = {0};
    if (hc_device_bus_id(device, bus, (uint32_t)sizeof bus) != HC_OK) return cpus;
    for (char* c = bus; *c; c++) *c = (char)tolower((unsigned char)*c);
    auto slurp = [](const std::string& path) {
        std::string text;
        if (FILE* f = fopen(path.c_str(), "r")) {
            char buf[4096];
            const size_t n = fread(buf, 1, sizeof buf - 1, f);
            fclose(f);
            text.assign(buf, n);
        }
        return text;
    };
    const std::string node = slurp(std::string("/sys/bus/pci/devices/") + bus + "/numa_node");
    if (node.empty() || atoi(node.c_str()) < 0) return cpus;
    if (slurp("/sys/devices/system/node/online").find_first_of(",-") == std::string::npos) return cpus;  // a single node
    const std::string list = slurp("/sys/devices/system/node/node" + std::to_string(atoi(node.c_str())) + "/cpulist");
    for (size_t i = 0; i < list.size();) {  // "64-127,192-255"
        if (!isdigit((unsigned char)list[i])) {
            i++;
            continue;
        }
        char* end = nullptr;
        const long a = strtol(list.c_str() + i, &end, 10);
        long b = a;
        if (*end == '-') b = strtol(end + 1, &end, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) cpus.push_back((int)c);
        i = (size_t)(end - list.c_str());
    }
    return cpus;
}

// The calling thread onto those of `cpus` it is ALLOWED on: the mask the process was started with (taskset, numactl, a
// scheduler that pins by affinity) is never widened — when none of the node's CPUs is in it, nothing changes.
inline void bind_thread_to(const std::vector<int>& cpus) {
    if (cpus.empty()) return;
    cpu_set_t allowed, set;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    CPU_ZERO(&set);
    int n = 0;
    for (int c : cpus)
        if (CPU_ISSET(c, &allowed)) {
            CPU_SET(c, &set);
            n++;
        }
    if (n == 0) return;
    (void)sched_setaffinity(0, sizeof set, &set);  // best effort: a cgroup may not grant these CPUs
}

// the calling thread (and the threads it starts meanwhile) next to the device for a scope, then where it was allowed before
struct BoundForNow {
    cpu_set_t before;
    bool restore = false;
    explicit BoundForNow(const std::vector<int>& cpus) {
        if (cpus.empty()) return;
        CPU_ZERO(&before);
        restore = sched_getaffinity(0, sizeof before, &before) == 0;
        bind_thread_to(cpus);
    }
    ~BoundForNow() {
        if (restore) (void)sched_setaffinity(0, sizeof before, &before);
    }
    BoundForNow(const BoundForNow&) = delete;
    BoundForNow& operator=(const BoundForNow&) = delete;
};

}  // namespace hc
#endif
