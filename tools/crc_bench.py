#!/usr/bin/env python3
"""CRC-32 of hb_load_webgraph's store check (hb_webgraph.cpp): carry-less-multiplication form against the slicing-by-8 table form
(HB_CRC32_TABLES=1), one thread and the parallel pieces-and-combine form, on THIS host.  usage: tools/crc_bench.py [GiB, default 2]"""
import ctypes
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(gib):
    import numpy as np
    from stract_amd import _lib
    lib = _lib.load()
    lib.hbw_debug_crc32.restype = ctypes.c_uint32
    lib.hbw_debug_crc32.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    small = np.ones(100 << 20, dtype=np.uint8)  # below 128 MiB: one thread
    big = np.ones(int(gib * (1 << 30)), dtype=np.uint8)
    out = {}
    for name, b in (("one_thread_GBs", small), ("parallel_GBs", big)):
        best = 0.0
        for _ in range(3):
            t = time.perf_counter()
            lib.hbw_debug_crc32(b.ctypes.data, b.nbytes)
            best = max(best, b.nbytes / (time.perf_counter() - t) / 1e9)
        out[name] = round(best, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(float(sys.argv[2]))
    else:
        gib = sys.argv[1] if len(sys.argv) > 1 else "2"
        res = {}
        for form, env in (("clmul", {}), ("tables", {"HB_CRC32_TABLES": "1"})):
            e = dict(os.environ, **env)
            e.pop("HB_CRC32_TABLES", None) if not env else None
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", gib], capture_output=True, text=True, env=e)
            res[form] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-300:]}
        res["cpus"] = os.cpu_count()
        print(json.dumps(res))
