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


def test_bench_line_at_one_gpu_is_the_scaling_runs_first_point(monkeypatch):
    """The driver takes BENCH from `bench.py --steps K --warmup W` and the first point of the scaling curve from `bench.py --gpus 1
    --steps K --warmup W`; efficiency at N > 1 is computed against that point.  Both must be the SAME measurement of the SAME
    workload - BASELINE.json configs[3] (C4: the graph the north-star target is quoted on, 100M hosts / 2B edges) - and N > 1 must
    run that workload too (strong scaling), with the north-star decomposition (edge partition + all-reduce) as `value`."""
    import json

    import bench
    from stract_amd import synth

    monkeypatch.delenv("HB_BENCH_CONFIG", raising=False)
    monkeypatch.delenv("HB_BENCH_INPUT", raising=False)

    def args(*argv):
        monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
        return vars(bench.parse())

    plain, one, eight = args("--steps", "20", "--warmup", "5"), args("--gpus", "1", "--steps", "20", "--warmup", "5"), args("--gpus", "8", "--steps", "20", "--warmup", "5")
    assert plain == one
    assert {k: v for k, v in eight.items() if k != "gpus"} == {k: v for k, v in one.items() if k != "gpus"}
    assert one["config"] == "C4" and one["partition"] == "both" and one["parity"] == "auto" and one["input"] == "records"
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "100M-host / 2B-edge" in base["configs"][3] and synth.CONFIGS["C4"]["label"] == "100M-host / 2B-edge"
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'main_part = "single" if world == 1 else ("dest" if a.partition == "dest" else "edge")' in src  # `value` at N > 1 = edge partition


@pytest.mark.gpu
def test_smoke_entry_point():
    import __graft_entry__ as g

    g.smoke()


@pytest.mark.gpu
def test_offline_bridge_tool(tmp_path):
    """tools/centrality_from_records.py on a dump of raw records = the operator mirror on the same records."""
    import numpy as np

    from oracle import hbo
    from stract_amd import synth
    from stract_amd.harmonic import EdgeListGraph, HarmonicCentrality

    g = synth.RmatGraph(11, 12_000)
    e = g.edges(salt=1, salt_seed=2)
    rec = tmp_path / "records.bin"
    e.tofile(str(rec))
    out = tmp_path / "out.csv"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "centrality_from_records.py"), str(rec), str(out),
                        "--chunk-records", "5000"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [line.strip().split(",") for line in open(out)][1:]
    hc = HarmonicCentrality.calculate(EdgeListGraph(e))
    ids, vals = hc.arrays()
    assert [int(x[0], 16) for x in rows] == [(int(h) << 64) | int(l) for l, h in zip(ids["lo"], ids["hi"])]
    assert [float(x[1]) for x in rows] == vals.tolist()
    assert [int(x[2]) for x in rows] == hbo.rank_results(vals).tolist()
