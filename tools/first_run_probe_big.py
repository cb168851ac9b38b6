#!/usr/bin/env python3
"""The first hb_run of a fresh context at C4 size, traced (HB_TRACE_RESULTS=1): where do the 0.4 s go that the end-to-end leg's s_run
shows over the steady 0.21 s?  usage: HB_SYNTH_CACHE=... tools/first_run_probe_big.py [config] [tune1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stract_amd import _lib, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
tune = (0, int(sys.argv[2], 0)) if len(sys.argv) > 2 else ()
g, _, label = synth.make_config(cfg)
for k in range(2):
    with _lib.Context(tune=tune) as ctx:
        ctx.load_dense(g.ids, g.row_ptr, g.src)
        for r in range(2):
            os.environ["HB_TRACE_RESULTS"] = "1" if r == 0 else ""
            if r:
                os.environ.pop("HB_TRACE_RESULTS", None)
            t0 = time.perf_counter()
            st = ctx.run()
            print("context %d run %d: %.1f ms (stages %d, list %d, ms_d2h %.2f)" % (k, r, (time.perf_counter() - t0) * 1e3, st["result_stages"], st["result_list"], st["ms_d2h"]), flush=True)
