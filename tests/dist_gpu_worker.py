"""Worker of tests/test_gpu.py::test_multi_process_library_exchanges_on_one_device - launched by torch.distributed.run, one
process per rank, ALL ranks on cuda:0, `gloo` backend.  Every rank drives the LIBRARY (hb_run on a world_size-N context); the
per-pass exchanges go through stract_amd.dist.HostStagedCollectives (hb_set_collectives): the pass driver's multi-rank
control flow, the merge / all-reduce / epilogue pipeline's event ordering and the changed-only packing run as the real
code, not as a Python mirror of the protocol."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    import torch.distributed as td

    from stract_amd import _lib, dist, synth

    scale, m, out_dir, modes = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4].split(",")
    td.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    g = synth.RmatGraph(scale, m, threads=1)
    out = {}
    for mode in modes:
        flags = _lib.HB_FLAG_NO_RCCL
        if mode.startswith("dest"):
            flags |= _lib.HB_FLAG_DEST_PARTITION
        if mode.endswith("changed"):
            flags |= _lib.HB_FLAG_CHANGED_ONLY
        tune = (0, 0x1000) if mode == "edge_no_pipeline" else ()   # tune[1] bit 12: one all-reduce over all rows instead of 4 ranges
        split = dist.partition_dense_by_dest if mode.startswith("dest") else dist.partition_dense
        rp, src = split(g.row_ptr, g.src, rank, world)
        with _lib.Context(device=0, flags=flags, rank=rank, world_size=world, tune=tune) as ctx:
            coll = dist.HostStagedCollectives(ctx)
            ctx.load_dense(g.ids, rp, src)
            st = ctx.run()
            ids, vals = ctx.results()
            hr, hk = ctx.state_hash()
            ps = ctx.pass_stats()
        # every rank must hold the same counters after the last pass
        mine = torch.tensor([hr & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(every, mine)
        out[mode] = {"passes": int(st["passes"]), "results": int(len(vals)), "ids_lo_sum": int(ids["lo"].sum() & 0xFFFFFFFFFFFFFFFF) if len(ids) else 0,
                     "vals_bits": vals.view(np.uint64).tolist() if rank == 0 else None,
                     "same_counters_on_all_ranks": bool(all(int(e.item()) == int(mine.item()) for e in every)),
                     "local_edges": int(len(src)), "calls": coll.calls, "callback_error": coll.error,
                     "modes_of_passes": [int(p["mode"]) for p in ps], "wire_bytes": int(st["wire_bytes"])}
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(out, f)
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
