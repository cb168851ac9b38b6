#!/usr/bin/env python3
"""Wall time of the FIRST hb_run of a fresh context (what a drop-in user sees: entrypoint/centrality.rs:46-53 runs once) next to the
second and third, with the staged result download on (default), off (tune[1] bit 14) and forced after every pass (bit 15).
usage: tools/first_run_probe.py [config, default C3]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stract_amd import _lib, synth  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    g, _, label = synth.make_config(cfg)
    out = {"config": label}
    for name, tune in (("staged_default", ()), ("staged_off", (0, 0x4000)), ("staged_every_pass", (0, 0x8000)), ("staged_default_again", ())):
        with _lib.Context(tune=tune) as ctx:
            ctx.load_dense(g.ids, g.row_ptr, g.src)
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                st = ctx.run()
                times.append(round((time.perf_counter() - t0) * 1e3, 3))
            ids, vals = ctx.results()
            out[name] = {"ms_run_1_2_3": times, "ms_finish": round(st["ms_d2h"], 3), "stages": int(st["result_stages"]), "list": int(st["result_list"]),
                         "results": int(len(vals)), "sum_bits": int(vals.view("u8").sum() & 0xFFFFFFFFFFFFFFFF)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
