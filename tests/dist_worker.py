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
    mode = sys.argv[4] if len(sys.argv) > 4 else "edge"
    td.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    g = synth.RmatGraph(scale, m, threads=1)
    split = dist.partition_dense_by_dest if mode.startswith("dest") else dist.partition_dense
    rp, src = split(g.row_ptr, g.src, rank, world)
    o = hbo.Dense(g.id_low64(), rp, src, threads=1)
    has, passes = True, 0
    while has:
        o.step_local(hbo.FRONTIER)
        pend = torch.from_numpy(o.pending())
        if mode == "dest_changed":
            # HB_FLAG_CHANGED_ONLY protocol: all-gather the changed bits of the owned rows, then every rank broadcasts
            # only its changed counters (packed, ascending row order); unchanged foreign rows keep the old value
            old = torch.from_numpy(o.registers())
            per = (g.n + world - 1) // world
            mine_changed = torch.zeros(per, dtype=torch.uint8)
            own_rows = torch.arange(rank, g.n, world)
            mine_changed[:len(own_rows)] = (pend[own_rows] != old[own_rows]).any(dim=1).to(torch.uint8)
            bits = [torch.zeros_like(mine_changed) for _ in range(world)]
            td.all_gather(bits, mine_changed)
            for r in range(world):
                rows_r = torch.arange(r, g.n, world)
                ch = bits[r][:len(rows_r)].bool()
                packed = pend[rows_r[ch]].clone() if r == rank else torch.zeros((int(ch.sum()), 64), dtype=torch.uint8)
                if packed.numel():
                    td.broadcast(packed, src=r)
                if r != rank:
                    pend[rows_r] = old[rows_r]
                    pend[rows_r[ch]] = packed
        elif mode == "edge_changed":
            # HB_FLAG_CHANGED_ONLY with the edge partition: every rank marks the rows its LOCAL merge changed, the marks are
            # all-gathered and OR-ed, and the all-reduce(max) runs over the rows of that union only, packed in ascending
            # row order (the same on every rank); all other rows keep the old value
            old = torch.from_numpy(o.registers())
            mine = (pend != old).any(dim=1).to(torch.uint8)
            marks = [torch.zeros_like(mine) for _ in range(world)]
            td.all_gather(marks, mine)
            union = torch.stack(marks).max(dim=0).values.bool()
            packed = pend[union].clone()
            if packed.numel():
                td.all_reduce(packed, op=td.ReduceOp.MAX)
            pend[:] = old
            pend[union] = packed
        elif mode == "dest":

            # all-gather of the owned rows (rank r owns the rows r, r + world, ...), padded to equal length
            per = (g.n + world - 1) // world
            mine = torch.zeros((per, 64), dtype=torch.uint8)
            own = pend[rank::world]
            mine[:len(own)] = own
            parts = [torch.zeros_like(mine) for _ in range(world)]
            td.all_gather(parts, mine)
            for r in range(world):
                cnt = len(range(r, g.n, world))
                pend[r::world] = parts[r][:cnt]
        elif mode == "edge_ranges":
            # the pipelined form of the edge partition (hb_api.hip edge_overlap): the node rows are all-reduced in four row
            # ranges, one collective per range, in order - the library issues range k while it still merges range k + 1
            tiles = (g.n + 63) // 64
            bounds = [min(g.n, tiles * k // 4 * 64) for k in range(5)]
            for lo, hi in zip(bounds, bounds[1:]):
                if hi > lo:
                    part = pend[lo:hi].clone()
                    td.all_reduce(part, op=td.ReduceOp.MAX)
                    pend[lo:hi] = part
        else:
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
