#!/usr/bin/env python3
"""Static upper bounds, from the device plan alone (CPU only), for two locality ideas the dense pass was asked to try
(VERDICT r1 #4): (a) 128-byte pairing - how many gathers have their pair partner (hotness rank ^ 1, the other half of
the 128-byte line) in the SAME row, so that one request could serve both; (b) clustering small node rows by their
dominant warm slice - how many direct gathers of a row fall into its most frequent slice of 64 Ki counters beyond the hot
slice 0 (those could become L2 hits if rows were processed slice by slice).
usage: tools/locality_potential.py [config, default 22:40000000]"""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stract_amd import _lib, synth  # noqa: E402


def main():
    g, _, label = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "22:40000000")
    p = _lib.host_plan(g.row_ptr, g.src)
    rp, src, lb, n_pad = p["row_ptr"], p["src"], p["level_begin"], p["n_pad"]
    print("%s: n = %d, m = %d, level-1 rows = %d" % (label, g.n, g.m, int(lb[1] - lb[0]) if len(lb) > 1 else 0))

    def rows(lo, hi):
        b, e = int(rp[lo]), int(rp[hi])
        s = src[b:e].astype(np.int64)
        row = np.repeat(np.arange(lo, hi), np.diff(rp[lo:hi + 1]).astype(np.int64))
        ok = s < n_pad  # real sources only
        return s[ok], row[ok]

    for name, (lo, hi) in (("level-1 chunk rows", (int(lb[0]), int(lb[1])) if len(lb) > 1 else (n_pad, n_pad)), ("node rows (direct)", (0, n_pad))):
        s, row = rows(lo, hi)
        if not len(s):
            continue
        partner = (row[1:] == row[:-1]) & ((s[1:] ^ 1) == s[:-1]) & ((s[:-1] & 1) == 0)
        hot = partner & (s[:-1] < 65536)
        print("  %-20s gathers %d; in a same-row 128-B pair: %.2f %% (%.2f %% outside the L2-resident hot slice)"
              % (name, len(s), 200.0 * partner.sum() / len(s), 200.0 * (partner.sum() - hot.sum()) / len(s)))
    s, row = rows(0, n_pad)
    sl = s >> 16
    warm = sl >= 1
    key = row[warm] * 4096 + np.minimum(sl[warm], 4095)
    uniq, cnt = np.unique(key, return_counts=True)
    best = np.zeros(n_pad, dtype=np.int64)
    np.maximum.at(best, uniq // 4096, cnt)
    print("  node rows: %.1f %% of the direct gathers hit slice 0; at most %.1f %% more would hit if every row's dominant warm "
          "slice were L2-resident while the row runs (%.2f such gathers per row with direct sources)"
          % (100.0 * (~warm).sum() / len(s), 100.0 * best.sum() / len(s), best.sum() / max((best > 0).sum(), 1)))


if __name__ == "__main__":
    main()
