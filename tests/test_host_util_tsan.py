"""
CPU: the host library's thread pool (prosstt_amd/csrc/host/host_util.cpp) under ThreadSanitizer -- two callers at once,
each widening, widening from the byte wire and scattering through the pool several times with changing pool sizes:
no data race reported, every result right.  (The pool hands out pieces by one atomic counter and is serialised per call;
this is the test that says so.)
"""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>
extern "C" int prosstt_amd_host_widen_i32_i64(const int32_t*, int64_t*, uint64_t, int32_t);
extern "C" int prosstt_amd_host_widen_u8_i32(const uint8_t*, int32_t*, uint64_t, int32_t);
extern "C" int prosstt_amd_host_widen_u16_i64(const uint16_t*, int64_t*, uint64_t, int32_t);
extern "C" int prosstt_amd_host_scatter_i32(void*, int32_t, const int64_t*, const int32_t*, uint64_t, int32_t);
int main()
{
    const uint64_t n = 2000017;
    int bad = 0;
    auto work = [&](int seed) {
        std::vector<int32_t> src(n), d32(n);
        std::vector<int64_t> dst(n), d64(n);
        std::vector<uint8_t> s8(n);
        std::vector<uint16_t> s16(n);
        for (uint64_t i = 0; i < n; ++i) { src[i] = (int32_t)(i * 2654435761u + seed); s8[i] = (uint8_t)(i + seed); s16[i] = (uint16_t)(i * 7 + seed); }
        for (int rep = 0; rep < 3; ++rep) {
            prosstt_amd_host_widen_i32_i64(src.data(), dst.data(), n, 5 + rep);
            prosstt_amd_host_widen_u8_i32(s8.data(), d32.data(), n, 3 + rep);
            prosstt_amd_host_widen_u16_i64(s16.data(), d64.data(), n, 2 + 3 * rep);
            for (uint64_t i = 0; i < n; i += 997) if (dst[i] != (int64_t)src[i] || d32[i] != (int32_t)s8[i] || d64[i] != (int64_t)s16[i]) ++bad;
            std::vector<int64_t> pos; std::vector<int32_t> val;
            for (uint64_t i = 0; i < n; i += 7) { pos.push_back((int64_t)i); val.push_back((int32_t)i - 5); }
            prosstt_amd_host_scatter_i32(dst.data(), 8, pos.data(), val.data(), pos.size(), 6);
            for (uint64_t i = 0; i < n; i += 7 * 131) if (dst[i] != (int64_t)((int32_t)i - 5)) ++bad;
        }
    };
    std::thread a(work, 1), b(work, 2);
    a.join(); b.join();
    printf("bad %d\n", bad);
    return bad != 0;
}
"""


def test_the_pool_is_free_of_data_races_under_tsan():
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    tmp = tempfile.mkdtemp(prefix="host_tsan_")
    try:
        open(os.path.join(tmp, "driver.cpp"), "w").write(DRIVER)
        exe = os.path.join(tmp, "driver")
        build = subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-o", exe, os.path.join(tmp, "driver.cpp"),
                                os.path.join(ROOT, "prosstt_amd", "csrc", "host", "host_util.cpp")], capture_output=True, text=True)
        if build.returncode != 0 and ("sanitize" in build.stderr or "tsan" in build.stderr):
            pytest.skip("this g++ has no ThreadSanitizer runtime")
        assert build.returncode == 0, build.stderr[-2000:]
        run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        if run.returncode != 0 and "unexpected memory mapping" in run.stderr:
            pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
        assert run.returncode == 0 and "bad 0" in run.stdout and "ThreadSanitizer" not in run.stderr, (run.stdout + run.stderr)[-3000:]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
