#!/usr/bin/env python3
"""Do two builds of the library write the SAME speedy_kv store files?  Four key sets (hashed NodeIDs; small integers; every bincode length
class around its boundaries; long shared prefixes) x {parallel, sequential} fst build x {other build, this build}: SHA-256 of every segment
file.  Used in round 6 to show that the rewritten fst writer (implicit key tails, staged writes, cheap CRC combine, fst beside the blob files)
produces byte-identical files to the build before it.
usage: HB_OLD_LIB=/path/to/other/libhyperball.so tools/store_compare_builds.py"""
import os, sys, glob, hashlib, subprocess, json, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OLD = os.environ.get("HB_OLD_LIB") or sys.exit("set HB_OLD_LIB to the library to compare with")
code = r'''
import sys, os, numpy as np
sys.path.insert(0, os.environ["HB_REPO_ROOT"])
from stract_amd import _lib
case, out = sys.argv[1], sys.argv[2]
rng = np.random.default_rng(7)
if case == "hashed":
    n = 1_500_000
    ids = np.zeros(n, dtype=_lib.U128); ids["lo"] = rng.integers(0, 1 << 63, n, dtype=np.uint64) * 2 + rng.integers(0, 2, n, dtype=np.uint64); ids["hi"] = rng.integers(1 << 62, 1 << 63, n, dtype=np.uint64)
elif case == "small":
    n = 70_000
    ids = np.zeros(n, dtype=_lib.U128); ids["lo"] = np.arange(n, dtype=np.uint64)
elif case == "classes":
    vals = set()
    for b in (0, 250, 251, 255, 256, 65535, 65536, 2**32 - 1, 2**32, 2**63, 2**64 - 1):
        for d in range(-3, 4):
            if 0 <= b + d < 2**64: vals.add(b + d)
    lo = sorted(vals)
    ids = np.zeros(len(lo) * 3, dtype=_lib.U128)
    ids["lo"][:len(lo)] = np.array(lo, dtype=np.uint64)
    ids["lo"][len(lo):2*len(lo)] = np.array(lo, dtype=np.uint64); ids["hi"][len(lo):2*len(lo)] = 1
    ids["lo"][2*len(lo):] = np.array(lo, dtype=np.uint64); ids["hi"][2*len(lo):] = 2**63
    n = len(ids)
elif case == "dense_prefix":
    # many keys sharing long prefixes: consecutive 64-bit integers above 2^32 (9-byte keys differing in the low bytes)
    n = 300_000
    ids = np.zeros(n, dtype=_lib.U128); ids["lo"] = (np.uint64(1) << np.uint64(40)) + np.arange(n, dtype=np.uint64) * np.uint64(3)
vals = rng.random(n); ranks = rng.permutation(n).astype(np.uint64)
_lib.store_harmonic(out, ids, vals, ranks)
'''
res = {}
for case in ("hashed", "small", "classes", "dense_prefix"):
    h = {}
    for tag, lib in (("old", OLD), ("new", "")):
        for mode in ("parallel", "sequential"):
            out = "/dev/shm/cmp_%s_%s_%s" % (case, tag, mode)
            shutil.rmtree(out, ignore_errors=True)
            env = dict(os.environ, HB_LIB_PATH=lib, HB_REPO_ROOT=ROOT)
            if mode == "sequential": env["HB_STORE_FST"] = "sequential"
            subprocess.check_call([sys.executable, "-c", code, case, out], env=env)
            d = {}
            for db in ("harmonic", "harmonic_rank"):
                for f in glob.glob(out + "/" + db + "/*"):
                    ext = os.path.splitext(f)[1] or os.path.basename(f)
                    if ext == ".json": continue
                    d[db + ext] = hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
            h[tag + "_" + mode] = d
            shutil.rmtree(out, ignore_errors=True)
    same = all(h["old_parallel"] == v for v in h.values())
    print(case, "identical across old/new x parallel/sequential:", same, h["new_parallel"].get("harmonic.ids"))
    if not same:
        for k, v in h.items(): print("  ", k, v)
