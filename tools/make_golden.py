#!/usr/bin/env python3
"""Generates tests/golden/hyperball_golden.json from the CPU oracle.

The reference (Rust) cannot run in this image, so these are REGRESSION vectors of the
oracle, not outputs of the reference: they freeze the oracle's behaviour (which is pinned
against the reference's own known answers in tests/test_oracle.py) so that an accidental
change of the oracle or of the product is caught, also on the GPU box where only the
fixtures travel."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hbo  # noqa: E402
from tests import graphs  # noqa: E402


def main():
    rng = np.random.default_rng(20240917)
    regs = graphs.random_registers(rng, 400)
    gold = {"size_cases": {"registers": regs.tolist(), "sizes": hbo.hll_sizes(regs).tolist()}, "graphs": []}
    cases = {
        "reference_fixture": [(f, t) for f, t in graphs.FIXTURE],
        "lcg_200_1200": graphs.lcg_graph(),
        "lcg_50_120": graphs.lcg_graph(50, 120, 99),
        "chain_40": [(i, i + 1) for i in range(1, 40)],
        "star_in_300": [(i, 1) for i in range(2, 302)],
        "wide_ids": [((i * 0x9E3779B97F4A7C15F39CC0605CEDC835) % (1 << 128), ((i + 1) * 0x9E3779B97F4A7C15F39CC0605CEDC835) % (1 << 128))
                     for i in range(1, 60)] + [(0, 1 << 64), (1 << 64, 0)],
    }
    for name, edges in cases.items():
        ids, row_ptr, src = graphs.dense_from_tuples(edges)
        o = hbo.Dense(np.ascontiguousarray(ids["lo"]), row_ptr, src)
        T = o.run()
        vals, keep, k = o.finish()
        gold["graphs"].append({
            "name": name, "edges": [list(e) for e in edges], "passes": int(T),
            "centrality_hex": {str((int(h) << 64) | int(l)): float(v).hex()
                               for l, h, v in zip(ids["lo"][keep], ids["hi"][keep], vals[keep])}})
    L = hbo.load()
    diffs = sum(1 for e in np.linspace(20.0, 330.0, 200_001) if L.hbo_hll_estimate_bias(e, 0) != L.hbo_hll_estimate_bias(e, 1))
    gold["bsearch_variant_disagreements_linspace_20_330_200001"] = diffs
    out = os.path.join(ROOT, "tests", "golden", "hyperball_golden.json")
    with open(out, "w") as f:
        json.dump(gold, f)
    print("wrote", out, "variant disagreements:", diffs)


if __name__ == "__main__":
    main()
