#!/usr/bin/env python3
"""Turn a tools/profile.sh output directory (rocprofv3 rocpd .db files) into the small text
summaries that are committed under profiles/.
usage: export_profile.py <gpurun_out/prof_dir> <profiles/prefix>
writes <prefix>_kernel_stats.csv (rocprofv3 --kernel-trace --stats: per launch class calls/avg/total)
       <prefix>_pmc.json         (per launch class: FETCH_SIZE / WRITE_SIZE / TCC hit+miss per dispatch)

Launch classes: the hub-chunk kernel `pass_kernel<false, ...>` is launched once per tree LEVEL in every pass with
the same template arguments; the level-1 launch (the dominant kernel: all real hub edges) and the small upper-level
launches must not be averaged together.  Dispatches are walked in dispatch order and a hub launch gets the suffix
`#L<k>` = it is the k-th consecutive hub launch since the last kernel of another kind (the node-row / sweep /
init launch that ends a pass)."""
import csv
import glob
import json
import os
import sqlite3
import sys


def classify(rows):
    """rows: (dispatch_id, name, ...) in dispatch order -> list of class names (hub launches get #L<level>)."""
    out, lvl = [], 0
    for r in rows:
        name = r[1].split("(")[0].strip()
        if name.startswith("void "):
            name = name[5:]
        if name.startswith("hbk::pass_kernel<false,"):
            lvl += 1
            out.append("%s#L%d" % (name, lvl))
        else:
            lvl = 0
            out.append(name)
    return out


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    dbs = glob.glob(os.path.join(src, "trace", "**", "*.db"), recursive=True)
    if dbs:
        c = sqlite3.connect(dbs[0])
        rows = c.execute("select dispatch_id, name, duration from kernels order by dispatch_id").fetchall()
        agg = {}
        for cls, r in zip(classify(rows), rows):
            a = agg.setdefault(cls, [0, 0, None, None, []])
            a[0] += 1
            a[1] += r[2]
            a[2] = r[2] if a[2] is None else min(a[2], r[2])
            a[3] = r[2] if a[3] is None else max(a[3], r[2])
            a[4].append(r[2])
        tot = sum(a[1] for a in agg.values()) or 1
        with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            # the first seven columns are rocprofv3 --stats'; [r6] the last two leave out the zero-row warm-up launches the library makes at
            # load (a dispatch below 1e-3 of its class' longest one): the average to compare with bench.py's event-timed launch time
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "CallsWithoutWarmup", "AverageNsWithoutWarmup"])
            for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                real = [d for d in a[4] if d >= 1e-3 * a[3]]
                w.writerow([name, a[0], int(a[1]), round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 3), int(a[2]), int(a[3]), len(real), round(sum(real) / max(len(real), 1), 1)])
    pmc = {}
    for db in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*.db"), recursive=True)):
        c = sqlite3.connect(db)
        rows = c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id").fetchall()
        # several counters per dispatch: classify on the distinct dispatches
        disp = []
        for r in rows:
            if not disp or disp[-1][0] != r[0]:
                disp.append((r[0], r[1]))
        cls_of = dict(zip([d[0] for d in disp], classify(disp)))
        for did, name, counter, value in rows:
            if "rocclr" in name:
                continue
            e = pmc.setdefault(cls_of[did], {}).setdefault(counter, {"dispatches": 0, "sum": 0.0, "max": 0.0, "values": []})
            e["dispatches"] += 1
            e["values"].append(value)
            e["sum"] += value
            e["max"] = max(e["max"], value)  # the largest single dispatch (sweep kernels: the first sweep pass of a run, cf. MaxNs of the trace)
    for k, cs in pmc.items():
        for e in cs.values():
            # [r6] the library warms every kernel with a zero-row launch at load: such a dispatch (counters ~ 0) must not dilute the class
            # average - launches below 1e-4 of the class' largest one are not counted
            small = sum(1 for v in e.pop("values") if v < 1e-4 * e["max"])
            e["warmup_dispatches_ignored"] = small
            e["per_dispatch"] = e["sum"] / max(e["dispatches"] - small, 1)
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            # FETCH_SIZE / WRITE_SIZE are KiB.  Calibration on this access pattern
            # (tools/gather_bench.hip, exactly-once 64-byte quad gathers over 8 GiB): FETCH_SIZE
            # reported 9.12-9.56 GB for 9.13 GB fetched -> factor 1.0 for the gather kernels
            # (the x2 correction of the guide applies to wide coalesced streams only).
            cs["hbm_bytes_per_dispatch"] = (cs["FETCH_SIZE"]["per_dispatch"] + cs["WRITE_SIZE"]["per_dispatch"]) * 1024.0
        if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
            h, m = cs["TCC_HIT_sum"]["sum"], cs["TCC_MISS_sum"]["sum"]
            cs["l2_hit_rate"] = h / (h + m) if h + m else None
            cs["l2_misses_per_dispatch"] = cs["TCC_MISS_sum"]["per_dispatch"]
    with open(prefix + "_pmc.json", "w") as f:
        json.dump(pmc, f, indent=1, sort_keys=True)
    print("wrote", prefix + "_kernel_stats.csv", prefix + "_pmc.json")


if __name__ == "__main__":
    main()
