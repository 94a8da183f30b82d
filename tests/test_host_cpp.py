"""The C++ host mirror (racing-slam_amd/host/) over the C-ABI: compile on the CPU (always),
run its self-test against the oracle on the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "host_cpp", "test_host.bin")


def build_host_test(rs, oracle):
    rs.load()
    oracle.lib()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(ROOT, "tests", "host_cpp", "test_host.cpp"), os.path.join(ROOT, "racing-slam_amd", "host", "slam_host.cpp")]
    deps = srcs + [os.path.join(ROOT, "racing-slam_amd", "host", "slam_host.h"), os.path.join(ROOT, "include", "rsgpu.h"),
                   os.path.join(ROOT, "racing-slam_amd", "librsgpu.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if os.path.exists(BIN) and all(os.path.getmtime(d) <= os.path.getmtime(BIN) for d in deps):
        return BIN
    cmd = [hipcc, "-O2", "-std=c++17", "-Wall", "-o", BIN] + srcs + [
        "-L" + os.path.join(ROOT, "racing-slam_amd"), "-lrsgpu", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
        "-Wl,-rpath," + os.path.join(ROOT, "racing-slam_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"]
    subprocess.check_call(cmd)
    return BIN


BENCH_BIN = os.path.join(ROOT, "tests", "host_cpp", "bench_boundary.bin")


def build_boundary_bench(rs):
    """tests/host_cpp/bench_boundary.cpp: the caller's view of the drop-in (host objects in, host results out), timed.
    No oracle in it: it is a measurement of the product path (bench.py --boundary runs it)."""
    rs.load()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(ROOT, "tests", "host_cpp", "bench_boundary.cpp"), os.path.join(ROOT, "racing-slam_amd", "host", "slam_host.cpp")]
    deps = srcs + [os.path.join(ROOT, "racing-slam_amd", "host", "slam_host.h"), os.path.join(ROOT, "include", "rsgpu.h"),
                   os.path.join(ROOT, "racing-slam_amd", "librsgpu.so")]
    if os.path.exists(BENCH_BIN) and all(os.path.getmtime(d) <= os.path.getmtime(BENCH_BIN) for d in deps):
        return BENCH_BIN
    cmd = [hipcc, "-O2", "-std=c++17", "-Wall", "-o", BENCH_BIN] + srcs + [
        "-L" + os.path.join(ROOT, "racing-slam_amd"), "-lrsgpu", "-Wl,-rpath," + os.path.join(ROOT, "racing-slam_amd"), "-lm"]
    subprocess.check_call(cmd)
    return BENCH_BIN


def test_host_mirror_compiles_and_links(rs, oracle):
    assert os.path.exists(build_host_test(rs, oracle))
    assert os.path.exists(build_boundary_bench(rs))


@pytest.mark.gpu
def test_host_mirror_selftest(rs, oracle):
    exe = build_host_test(rs, oracle)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert "all checks passed" in r.stdout


def test_host_only_entry_points_under_address_sanitizer():
    """csrc/host.cpp + csrc/pose_graph.cpp (no GPU code in them) built with g++ -fsanitize=address and driven with
    exactly-sized heap buffers: out-of-bounds reads that only crash under another heap layout abort here."""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = os.path.join(ROOT, "tests", "host_cpp", "asan_host.bin")
    srcs = [os.path.join(ROOT, "tests", "host_cpp", "asan_host.cpp"), os.path.join(ROOT, "racing-slam_amd", "csrc", "host.cpp"),
            os.path.join(ROOT, "racing-slam_amd", "csrc", "pose_graph.cpp")]
    subprocess.check_call([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off",
                           "-o", exe] + srcs + ["-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "asan host checks passed" in r.stdout
