"""world_size-2 (and two world_size-3) CPU tests of the two multi-GPU decompositions (edge partition + all-reduce(max);
destination partition + all-gather of the owned rows).  Edge-partitioned algorithm with the
`gloo` backend: each rank pulls over its own edge subset (oracle arithmetic), the pending
counters are all-reduced with MAX, every rank finishes the pass; the result must be
bit-identical to the single-process run.  This is the decomposition the GPU path uses
(local merge -> ncclAllReduce(max, u8) -> estimator/Kahan); on the GPU box the collective
is RCCL, issued by the C library."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


import pytest


@pytest.mark.parametrize("mode,world", [("edge", 2), ("edge_ranges", 2), ("edge_changed", 2), ("dest", 2), ("dest_changed", 2),
                                        ("edge_changed", 3), ("dest_changed", 3)])  # 3 ranks: uneven slices, a rank with a short last range
def test_partitioned_pass_is_exact(tmp_path, mode, world):
    from oracle import hbo
    from stract_amd import synth

    scale, m = 11, 15_000
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(scale), str(m), str(tmp_path), mode]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g = synth.RmatGraph(scale, m)
    o = hbo.Dense(g.id_low64(), g.row_ptr, g.src)
    T = o.run()
    vals, keep, k = o.finish()
    total_edges = 0
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert int(z["passes"]) == T
        assert np.array_equal(z["keep"], keep)
        assert np.array_equal(z["vals"], vals.view(np.uint64))
        total_edges += int(z["local_edges"])
    assert total_edges == g.m
