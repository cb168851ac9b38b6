"""The caching device allocator (stract_amd/csrc/hb_pool.h) sits under every hipMalloc / hipFree of the library since round 4.
Its bookkeeping is host code: tests/pool_harness.cpp runs it against tests/fake_hip (a stand-in for the six runtime calls it
makes) - reuse, split, best fit, coalescing, trim, the limit, the out-of-memory retry, bad frees.  The GPU suite covers it on the
real runtime (tests/test_gpu.py: pool pressure, every load)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pool_bookkeeping_on_a_fake_device(tmp_path):
    exe = str(tmp_path / "pool_harness")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "fake_hip"),
                           "-I", os.path.join(ROOT, "stract_amd", "csrc"), os.path.join(ROOT, "tests", "pool_harness.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
