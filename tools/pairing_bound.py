#!/usr/bin/env python3
"""How many L2-miss REQUESTS could a 128-byte-pair-oriented layout of the cold counters save? (CPU only.)

Facts it builds on (profiles/r05a_gather_request_size.txt, tools/gather_bench.hip): the miss path of the dense pass is bound by
requests (~45-52 G/s), whatever their size - 32-byte gathers run at exactly the 64-byte rate, 128-byte ones (both halves of one
line) at the same REQUEST rate, i.e. twice the counters per second; and the two halves of a line are merged into one request when
they are gathered by the same quad OR by neighbouring quads (rows 2k, 2k + 1) of a wave in the same instruction.  So a request is
saved exactly when the two counters of one 128-byte line ("line mates") are gathered together by one row or by a row and its
neighbour in the same slot.  A counter has ONE line mate.  The layout is free (any order of the cold counters, any order of
the hub chunks, any slot order - max is order-free, harmonic.rs:116-157), so the question is how many such co-gathers the GRAPH
allows:

  designed   the planner can make every line serve ONE co-gather by construction (pick a row, make two of its cold sources line
             mates): at most one saved request per line = n_cold / 2 per pass, however the order is chosen;
  by chance  every further saving needs the SAME two counters to meet again in another row (or row pair): counted here on the
             real work-row graph under the greedy designed order, same-row and - optimistically, slots ignored - neighbouring rows.

usage: tools/pairing_bound.py [config, default 22:40000000] [resident counters, default 65 * 65536 = slices 0..64]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stract_amd import _lib, synth  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "22:40000000"
    resident = int(sys.argv[2]) if len(sys.argv) > 2 else 65 * 65536
    t0 = time.time()
    g, _, label = synth.make_config(cfg)
    p = _lib.host_plan(g.row_ptr, g.src)
    rp, src, lb, n_pad = p["row_ptr"], p["src"], p["level_begin"], int(p["n_pad"])
    rows_total = len(rp) - 1
    print("%s: n = %d, m = %d, work rows = %d (plan in %.0f s)" % (label, g.n, g.m, rows_total, time.time() - t0))
    # every gather of a REAL source by a level-1 chunk row or a node row, in work-row order (= the order rows run in)
    row = np.repeat(np.arange(rows_total, dtype=np.int64), np.diff(rp).astype(np.int64))
    s = src.astype(np.int64)
    real = s < n_pad
    row, s = row[real], s[real]
    G = len(s)
    cold = s >= resident
    rc, sc = row[cold], s[cold]
    Gc = len(sc)
    n_cold = int(len(np.unique(sc)))
    print("  gathers of real sources: %d; of COLD sources (hotness rank >= %d): %d = %.1f %%; distinct cold sources: %d (%.2f gathers each)"
          % (G, resident, Gc, 100.0 * Gc / G, n_cold, Gc / max(n_cold, 1)))
    print("  designed bound: one co-gather per 128-byte line = n_cold / 2 = %d saved requests per pass = %.2f %% of the cold gathers, %.2f %% of all gathers"
          % (n_cold // 2, 50.0 * n_cold / max(Gc, 1), 50.0 * n_cold / G))
    # greedy designed order: a cold source is placed by the FIRST row that gathers it; that row's newly placed sources become line
    # mates two by two (one saved request per line); a row's odd one out is paired with the next row's (no co-gather by design)
    uniq, first = np.unique(sc, return_index=True)
    first_row = rc[first]
    order = np.lexsort((uniq, first_row))  # placement order: by placing row
    placed_src, placed_row = uniq[order], first_row[order]
    # position of every placed source among its row's placed sources
    start = np.r_[0, np.flatnonzero(placed_row[1:] != placed_row[:-1]) + 1]
    cnt = np.diff(np.r_[start, len(placed_row)])
    pos = np.arange(len(placed_row)) - np.repeat(start, cnt)
    in_pair = pos < np.repeat(cnt - (cnt & 1), cnt)  # all but a row's odd one out
    designed = int(in_pair.sum() // 2)
    line = np.empty(len(placed_src), dtype=np.int64)
    line[in_pair] = np.arange(int(in_pair.sum())) // 2
    rest = np.flatnonzero(~in_pair)
    line[rest] = designed + np.arange(len(rest)) // 2
    half = np.empty(len(placed_src), dtype=np.int64)
    half[in_pair] = np.arange(int(in_pair.sum())) & 1
    half[rest] = np.arange(len(rest)) & 1
    line_of = np.zeros(int(uniq.max()) + 1, dtype=np.int64)
    half_of = np.zeros(int(uniq.max()) + 1, dtype=np.int8)
    line_of[placed_src], half_of[placed_src] = line, half
    gl, gh = line_of[sc], half_of[sc]
    # same-row co-gathers: both halves of a line present in one row
    key = rc * (int(line.max()) + 1) + gl
    o = np.argsort(key, kind="stable")
    k2, h2 = key[o], gh[o]
    same = int(((k2[1:] == k2[:-1]) & (h2[1:] != h2[:-1])).sum())
    # neighbouring rows (2k, 2k + 1), slots ignored (optimistic): the two halves in the two rows of a pair
    key = (rc >> 1) * (int(line.max()) + 1) + gl
    o = np.argsort(key, kind="stable")
    k2, h2 = key[o], gh[o]
    near = int(((k2[1:] == k2[:-1]) & (h2[1:] != h2[:-1])).sum())
    print("  greedy designed order: %d lines with a designed co-gather (%.1f %% of the bound: rows with an odd number of new sources lose one)"
          % (designed, 200.0 * designed / max(n_cold, 1)))
    print("  co-gathers found on the work-row graph under that order: same row %d (designed %d + by chance %d); row or its neighbour, slots "
          "ignored: %d" % (same, designed, same - designed, near))
    print("  => requests saved per pass: %.2f %% of the cold gathers, %.2f %% of all gathers (by chance alone: %.3f %% of the cold gathers)"
          % (100.0 * near / max(Gc, 1), 100.0 * near / G, 100.0 * (near - designed) / max(Gc, 1)))


if __name__ == "__main__":
    main()
