#!/usr/bin/env python3
"""Planner / kernel knob sweep on one resident graph (GPU box).
usage: tools/sweep.py <config> "<chunk>:<flags>:<tune,comma>" ...   prints one line per variant."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from stract_amd import _lib, synth  # noqa: E402


def main():
    g, _, _ = synth.make_config(sys.argv[1])
    ref = None
    for spec in sys.argv[2:]:
        chunk, flags, tune = spec.split(":")
        tune = tuple(int(x) for x in tune.split(",")) if tune else ()
        with _lib.Context(chunk=int(chunk), flags=int(flags), tune=tune) as ctx:
            t0 = time.perf_counter()
            ctx.load_dense(g.ids, g.row_ptr, g.src)
            t_load = time.perf_counter() - t0
            ctx.run()
            best = None
            for _ in range(3):
                st = ctx.run()
                if best is None or st["ms_loop"] < best["ms_loop"]:
                    best = st
                    ps = ctx.pass_stats()
            ids, vals = ctx.results()
            sig = (len(vals), int(vals.view(np.uint64).sum() & 0xFFFFFFFFFFFF))
            if ref is None:
                ref = sig
            dense = [p for p in ps if p["mode"] == 0]
            front = [p for p in ps if p["mode"] == 1]
            sparse = [p for p in ps if p["mode"] == 2]
            print(json.dumps({
                "spec": spec, "ms_loop": round(best["ms_loop"], 3), "gteps": round(g.m * best["passes"] / best["ms_loop"] / 1e6, 2),
                "passes": best["passes"], "dense_ms_gpu": round(float(np.mean([p["ms_gpu"] for p in dense])), 3) if dense else None,
                "dense_ms_main": round(float(np.mean([p["ms_main"] for p in dense])), 3) if dense else None,
                "front_ms_gpu": round(float(np.mean([p["ms_gpu"] for p in front])), 3) if front else None,
                "front_ms_sum": round(float(np.sum([p["ms_gpu"] for p in front])), 3) if front else None,
                "sparse_ms_sum": round(float(np.sum([p["ms_gpu"] for p in sparse])), 3) if sparse else None,
                "changed": [p["changed"] for p in ps], "active_pct": [round(100.0 * p["active_edges"] / g.m, 2) for p in ps],
                "modes": [p["mode"] for p in ps], "ms": [round(p["ms_gpu"], 3) for p in ps],
                "ms_level1_or_expand": [round(p["ms_level1"], 3) for p in ps], "ms_node_rows": [round(p["ms_main"], 3) for p in ps],
                "virtual_rows": best["virtual_rows"], "s_load": round(t_load, 2), "ms_plan": round(best["ms_plan"]),
                "same_result": sig == ref}), flush=True)


if __name__ == "__main__":
    main()
