"""ctypes binding of include/hb_ampc.h: a GPU-resident shard of the AMPC harmonic-centrality counter table
(`DefaultDhtTable<NodeID, HyperLogLog<64>>`, crates/core/src/entrypoint/ampc/harmonic_centrality/mod.rs:47-53) with the
three batch operations its mappers use (mapper.rs:52-118): batch_set, batch_get, batch_upsert(HyperLogLog64Upsert)."""
import ctypes

import numpy as np

from . import _lib

NO_CHANGE, MERGED, INSERTED = 0, 1, 2  # UpsertAction, dht/upsert.rs:24-28


class CounterTable:
    def __init__(self, device=-1, capacity_hint=0):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self.lib.hbu_create(device, capacity_hint, ctypes.byref(h))
        if rc != _lib.HB_OK:
            raise _lib.HyperballError(rc, (self.lib.hbu_last_error(None) or b"").decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.hbu_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _lib.HB_OK:
            raise _lib.HyperballError(rc, (self.lib.hbu_last_error(self.h) or b"").decode())

    def __len__(self):
        n = ctypes.c_uint64(0)
        self._check(self.lib.hbu_len(self.h, ctypes.byref(n)))
        return n.value

    @staticmethod
    def _args(keys, counters):
        keys = np.ascontiguousarray(keys, dtype=_lib.U128)
        counters = np.ascontiguousarray(counters, dtype=np.uint8).reshape(len(keys), 64)
        return keys, counters

    def batch_set(self, keys, counters):
        keys, counters = self._args(keys, counters)
        self._check(self.lib.hbu_batch_set(self.h, _lib._ptr(keys), _lib._ptr(counters), len(keys)))

    def batch_get(self, keys):
        keys = np.ascontiguousarray(keys, dtype=_lib.U128)
        out = np.zeros((len(keys), 64), dtype=np.uint8)
        found = np.zeros(len(keys), dtype=np.uint8)
        self._check(self.lib.hbu_batch_get(self.h, _lib._ptr(keys), len(keys), _lib._ptr(out), _lib._ptr(found)))
        return out, found.astype(bool)

    def batch_upsert(self, keys, counters):
        keys, counters = self._args(keys, counters)
        actions = np.zeros(len(keys), dtype=np.uint8)
        self._check(self.lib.hbu_batch_upsert(self.h, _lib._ptr(keys), _lib._ptr(counters), len(keys), _lib._ptr(actions)))
        return actions
