#!/usr/bin/env python3
"""Throughput of the AMPC counter shard (include/hb_ampc.h) with its DEVICE key index (round 5; VERDICT r4 #8: "report upserts/s
at 10 M keys"): batch_set of K distinct keys (every pair inserts), batch_upsert of random pairs over those keys (every pair finds
its slot, the pairs of one key applied in batch order), batch_get; from pageable numpy buffers (what the ctypes mirror hands
over) and from page-locked ones (hb_pinned_alloc: what a worker that owns its receive buffers would use).  Semantics:
crates/core/src/ampc/dht/upsert.rs:66-89, dht/store.rs:159-190; driver: entrypoint/ampc/harmonic_centrality/mapper.rs:52-118.
usage: tools/ampc_bench.py [keys, default 10000000] [pairs per batch, default 1000000]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stract_amd import _lib, ampc  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    rng = np.random.default_rng(1)
    keys = np.zeros(K, dtype=_lib.U128)
    keys["lo"] = rng.permutation(K).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    keys["hi"] = rng.integers(0, 1 << 63, K, dtype=np.uint64)
    out = {"keys": K, "pairs_per_batch": B, "bytes_per_pair": 80}
    for label, pinned in (("pageable", False), ("pinned", True)):
        kb = _lib.PinnedRecords(B, dtype=_lib.U128) if pinned else None
        vb = _lib.PinnedRecords(B * 64, dtype=np.uint8) if pinned else None
        kbuf = kb.array if pinned else np.zeros(B, dtype=_lib.U128)
        vbuf = (vb.array if pinned else np.zeros(B * 64, dtype=np.uint8)).reshape(B, 64)
        vbuf[:] = rng.integers(0, 40, (B, 64), dtype=np.uint8)
        res = {}
        with ampc.CounterTable(capacity_hint=K) as tab:
            lib, h = tab.lib, tab.h
            acts = np.zeros(B, dtype=np.uint8)
            found = np.zeros(B, dtype=np.uint8)
            getbuf = _lib.PinnedRecords(B * 64, dtype=np.uint8) if pinned else None
            gout = getbuf.array if pinned else np.zeros(B * 64, dtype=np.uint8)
            t_set = t_up = t_get = 0.0
            for b in range(0, K, B):                      # setup_counters: every pair inserts a new key
                n = min(B, K - b)
                kbuf[:n] = keys[b:b + n]
                t0 = time.perf_counter()
                tab._check(lib.hbu_batch_set(h, _lib._ptr(kbuf), _lib._ptr(vbuf), n))
                t_set += time.perf_counter() - t0
            assert len(tab) == K
            rounds = max(K // B, 1)
            for r in range(rounds):                       # update_counters: random destinations, ~1/8 of the batch on 1000 hub keys
                idx = rng.integers(0, K, B)
                hub = rng.random(B) < 0.125
                idx[hub] = rng.integers(0, 1000, int(hub.sum()))
                kbuf[:] = keys[idx]
                t0 = time.perf_counter()
                tab._check(lib.hbu_batch_upsert(h, _lib._ptr(kbuf), _lib._ptr(vbuf), B, _lib._ptr(acts)))
                t_up += time.perf_counter() - t0
            for r in range(rounds):                       # get_old_counters
                kbuf[:] = keys[rng.integers(0, K, B)]
                t0 = time.perf_counter()
                tab._check(lib.hbu_batch_get(h, _lib._ptr(kbuf), B, _lib._ptr(gout), _lib._ptr(found)))
                t_get += time.perf_counter() - t0
            assert found.all() and len(tab) == K
            res = {"batch_set_inserting_Mpairs_per_s": round(K / t_set / 1e6, 2), "batch_upsert_Mpairs_per_s": round(rounds * B / t_up / 1e6, 2),
                   "batch_get_Mkeys_per_s": round(rounds * B / t_get / 1e6, 2),
                   "upsert_link_GBs": round(rounds * B * 81 / t_up / 1e9, 2), "actions_merged_share_last_batch": round(float((acts == ampc.MERGED).mean()), 3)}
            if getbuf:
                getbuf.close()
        for x in (kb, vb):
            if x:
                x.close()
        out[label] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
