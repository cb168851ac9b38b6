"""ctypes binding of include/hb_webgraph.h: the native reader of Stract's on-disk webgraph edge store
(`<webgraph>/edges`, crates/core/src/webgraph/store.rs:60-74) and hb_load_webgraph."""
import ctypes

import numpy as np

from . import _lib

HBW_VERIFY_CRC = 0x1
HBW_PAGE_IDS = 0x2


class EdgeStoreReader:
    """Yields the SmallEdge records `Webgraph::host_edges()` would stream (before de-duplication), in the
    reference's order: segments as listed in meta.json, documents ascending (store.rs:297-314,360-417)."""

    def __init__(self, edges_dir, verify_crc=False, page_ids=False):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        flags = (HBW_VERIFY_CRC if verify_crc else 0) | (HBW_PAGE_IDS if page_ids else 0)
        rc = self.lib.hbw_open(edges_dir.encode(), flags, ctypes.byref(h))
        if rc != _lib.HB_OK:
            raise _lib.HyperballError(rc, (self.lib.hbw_last_error(None) or b"").decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.hbw_close(self.h)
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

    def num_segments(self):
        n = ctypes.c_uint64(0)
        self.lib.hbw_num_segments(self.h, ctypes.byref(n))
        return n.value

    def segment_info(self, i):
        buf = ctypes.create_string_buffer(33)
        rows = ctypes.c_uint64(0)
        rc = self.lib.hbw_segment_info(self.h, i, buf, ctypes.byref(rows))
        if rc != _lib.HB_OK:
            raise _lib.HyperballError(rc, "segment index out of range")
        return buf.value.decode(), rows.value

    def total_rows(self):
        n = ctypes.c_uint64(0)
        self.lib.hbw_total_rows(self.h, ctypes.byref(n))
        return n.value

    def read(self, first=0, count=None, page_level=False):
        """page_level: {from_id, to_id, rel_flags} of the same documents (reader opened with page_ids=True)."""
        if count is None:
            count = self.total_rows() - first
        out = np.zeros(count, dtype=_lib.EDGE)
        fn = self.lib.hbw_read_page_edges if page_level else self.lib.hbw_read_host_edges
        rc = fn(self.h, first, count, out.ctypes.data if count else None)
        if rc != _lib.HB_OK:
            raise _lib.HyperballError(rc, "range outside the store (or page-level ids not opened)")
        return out


def load_webgraph(ctx, edges_dir, verify_crc=False, page_ids=False):
    """hb_load_webgraph: stream the store into a Context (then ctx.run()).  page_ids: also the page-level records the
    reference's tail mode follows (ctx created with HB_FLAG_REFERENCE_TAIL)."""
    flags = (HBW_VERIFY_CRC if verify_crc else 0) | (HBW_PAGE_IDS if page_ids else 0)
    rc = ctx.lib.hb_load_webgraph(ctx.h, edges_dir.encode(), flags)
    if rc != _lib.HB_OK:
        msg = (ctx.lib.hb_last_error(ctx.h) or b"").decode() or (ctx.lib.hbw_last_error(None) or b"").decode()
        raise _lib.HyperballError(rc, msg)
