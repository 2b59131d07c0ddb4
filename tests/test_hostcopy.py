"""hc_hostcopy.h: the touch-ahead copy into untouched pageable memory must never write a touch byte behind the copy
(one summed counter let fast threads release a stretch a slow thread had not touched yet).  The ordering is
free of HIP (`copy_touched_ahead`), so it is driven here with memcpy as the copy: fresh anonymous mappings, more threads than
cores (taskset to two), every byte compared; and once under ThreadSanitizer."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build", "hostcopy")

SRC = r'''
#define HC_HOSTCOPY_NO_HIP
#include "hc_hostcopy.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, char** argv) {
    const uint64_t bytes = strtoull(argv[1], 0, 10), chunk = strtoull(argv[2], 0, 10);
    const unsigned T = atoi(argv[3]), reps = atoi(argv[4]);
    std::vector<char> src(bytes);
    for (uint64_t i = 0; i < bytes; i++) src[i] = (char)(1 + i % 251);  // no zero byte anywhere: a late touch shows
    uint64_t bad = 0;
    for (unsigned r = 0; r < reps; r++) {
        char* dst = (char*)mmap(0, bytes + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);  // untouched pages
        if (dst == MAP_FAILED) return 2;
        char* to = dst + (r % 2 ? 24 : 0);  // an unaligned destination every other time
        const char* s = src.data();
        int rc = hc::copy_touched_ahead(to, bytes, chunk, T, [s](char* at, uint64_t off, uint64_t len) { memcpy(at, s + off, len); return 0; });
        if (rc) return 3;
        for (uint64_t i = 0; i < bytes; i++) bad += to[i] != s[i];
        munmap(dst, bytes + 4096);
    }
    int rc = hc::copy_touched_ahead((void*)0, 0, chunk, T, [](char*, uint64_t, uint64_t) { return 7; });  // nothing to copy: no call
    int er = 0; { std::vector<char> d(3 * chunk); er = hc::copy_touched_ahead(d.data(), d.size(), chunk, T, [](char*, uint64_t off, uint64_t) { return off ? 5 : 0; }); }
    printf("%llu %d %d\n", (unsigned long long)bad, rc, er);
    return bad != 0;
}
'''


def _build(name, flags):
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(BUILD, "hostcopy_test.cpp")
    with open(src, "w") as f:
        f.write(SRC)
    exe = os.path.join(BUILD, name)
    subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", *flags, "-I", os.path.join(ROOT, "haploconduct_amd", "csrc"), "-o", exe, src],
                   check=True, capture_output=True, text=True)
    return exe


def _prefix():
    return ["taskset", "-c", "0,1"] if shutil.which("taskset") and len(os.sched_getaffinity(0)) >= 2 else []


def test_touches_never_land_behind_the_copy():
    exe = _build("hostcopy_test", [])
    # 192 MiB in 2 MiB stretches, eight touching threads on two cores, six fresh mappings
    r = subprocess.run(_prefix() + [exe, str(192 << 20), str(2 << 20), "8", "6"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    bad, rc_empty, rc_err = r.stdout.split()
    assert bad == "0" and rc_empty == "0", r.stdout
    assert rc_err == "5", "an error of the copy must come back, and stop the stretches after it"


def test_ragged_sizes_and_one_thread():
    exe = _build("hostcopy_test", [])
    for nbytes, chunk, T in ((1, 4096, 3), (4097, 4096, 8), ((8 << 20) + 123, 1 << 20, 1), ((8 << 20) - 1, 3 << 20, 5)):
        r = subprocess.run([exe, str(nbytes), str(chunk), str(T), "2"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.split()[0] == "0", (nbytes, chunk, T, r.stdout, r.stderr)


def test_under_thread_sanitizer():
    try:
        exe = _build("hostcopy_tsan", ["-fsanitize=thread", "-g"])
    except subprocess.CalledProcessError as e:  # pragma: no cover
        pytest.skip("no ThreadSanitizer runtime: " + e.stderr[-200:])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, str(24 << 20), str(1 << 20), "6", "2"], capture_output=True, text=True, timeout=600, env=env)
    if "FATAL: ThreadSanitizer" in r.stderr and "unexpected memory mapping" in r.stderr:  # pragma: no cover
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-2000:]
