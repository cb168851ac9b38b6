"""Worker for tests/test_dist_cpu.py - launched by torch.distributed.run, one process per
logical rank, `gloo` backend, CPU only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as td

    from oracle import hbo
    from stract_amd import dist, synth

    scale, m, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    td.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    g = synth.RmatGraph(scale, m, threads=1)
    rp, src = dist.partition_dense(g.row_ptr, g.src, rank, world)
    o = hbo.Dense(g.id_low64(), rp, src, threads=1)
    has, passes = True, 0
    while has:
        o.step_local(hbo.FRONTIER)
        pend = torch.from_numpy(o.pending())
        td.all_reduce(pend, op=td.ReduceOp.MAX)  # in place on the oracle's pending counters
        has, _ = o.step_finish(hbo.FRONTIER)
        passes += 1
    vals, keep, k = o.finish()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), passes=passes, vals=vals.view(np.uint64), keep=keep,
             local_edges=len(src))
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
