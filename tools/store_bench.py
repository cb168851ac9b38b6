#!/usr/bin/env python3
"""hb_store_harmonic throughput: N random 128-bit NodeIDs -> both speedy_kv databases (include/hb_store.h), host only.
usage: tools/store_bench.py [N] [--dir DIR] [--check K]   (--check: read K random keys back with tests/speedy_kv_reader.py)"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from stract_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=5_000_000)
    ap.add_argument("--dir", default="")
    ap.add_argument("--check", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    ids = np.zeros(a.n, dtype=_lib.U128)
    ids["lo"] = rng.integers(0, 1 << 63, a.n, dtype=np.uint64) * 2 + rng.integers(0, 2, a.n, dtype=np.uint64)
    ids["hi"] = rng.integers(1 << 62, 1 << 63, a.n, dtype=np.uint64)
    vals = rng.random(a.n)
    ranks = rng.permutation(a.n).astype(np.uint64)
    base = a.dir or tempfile.mkdtemp(prefix="hb_store_bench_")
    out = os.path.join(base, "out")
    shutil.rmtree(out, ignore_errors=True)
    t0 = time.perf_counter()
    _lib.store_harmonic(out, ids, vals, ranks)
    dt = time.perf_counter() - t0
    size = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(out) for f in fs)
    res = {"entries_per_store": a.n, "stores": 2, "seconds": round(dt, 3), "entries_per_s_per_store": round(a.n / dt), "entries_per_s_both": round(2 * a.n / dt),
           "bytes_written": size, "write_GBs": round(size / dt / 1e9, 3), "threads": os.cpu_count(), "dir": base}
    if a.check:
        from tests import speedy_kv_reader as kv
        db = kv.Db(os.path.join(out, "harmonic_rank"), "u64", tempfile.gettempdir())
        ints = kv.ids_to_ints(ids)
        pick = rng.integers(0, a.n, a.check)
        res["checked"] = int(a.check)
        res["check_ok"] = bool(all(db.get(ints[j]) == int(ranks[j]) for j in pick.tolist()))
    print(json.dumps(res))
    if not a.dir:
        shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
