"""Driver entry points: build() (CPU: hipcc cross-compiles without a GPU), bench.py's behaviour without a
device, smoke() on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_entry_point():
    import __graft_entry__ as g

    g.build()  # incremental: a no-op when the libraries are up to date
    for rel in ("stract_amd/lib/libhyperball.so", "stract_amd/lib/libhb_synth.so", "oracle/libhb_oracle.so"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel


def test_bench_refuses_to_run_without_a_gpu():
    from stract_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stdout + r.stderr)


def test_bench_help_lists_the_contract_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], cwd=ROOT, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout


@pytest.mark.gpu
def test_smoke_entry_point():
    import __graft_entry__ as g

    g.smoke()
