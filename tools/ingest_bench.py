#!/usr/bin/env python3
"""hb_load_edges end to end on raw 40-byte SmallEdge records (what the Rust shim hands over):
GPU ingest (hb_ingest.hip) vs host ingest (hb_host.cpp).  usage: tools/ingest_bench.py <config>"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from stract_amd import _lib, synth  # noqa: E402


def main():
    cfg = synth.CONFIGS[sys.argv[1]]
    g = synth.RmatGraph(cfg["scale"], cfg["m"])
    e = g.edges(salt=0)
    out = {"config": sys.argv[1], "records": int(len(e)), "record_GB": round(e.nbytes / 1e9, 2)}
    ref = None
    for name, flags in (("gpu", 0), ("host", _lib.HB_FLAG_HOST_INGEST)):
        with _lib.Context(flags=flags) as ctx:
            t0 = time.perf_counter()
            ctx.load_edges(e)
            dt = time.perf_counter() - t0
            st = ctx.stats()
            out[name] = {"s_load_edges": round(dt, 3), "ms_ingest": round(st["ms_ingest"], 1), "ms_plan": round(st["ms_plan"], 1),
                         "ms_h2d": round(st["ms_h2d"], 1), "n": st["n"], "m_eff": st["m_eff"]}
            run = ctx.run()
            ids, vals = ctx.results()
            sig = (len(vals), int(vals.view(np.uint64).sum() & 0xFFFFFFFFFFFF))
            ref = ref or sig
            out[name]["same_result"] = sig == ref
            out[name]["ms_loop"] = round(run["ms_loop"], 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
