#!/usr/bin/env python3
"""bench.py - HyperBall (webgraph harmonic centrality) hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by
torch.distributed.run, one rank per GPU).  Prints ONE JSON line on rank 0.

  step      = one complete HyperBall run over the resident graph: initialize + every pass of
              the loop harmonic.rs:237-280 until a pass changes nothing (T passes, the last
              one being the reference's no-change pass) + normalize + result download.
  metric    = traversed edges per second: m_eff * T * K / t   (SURVEY.md §8(d)), the graph
              (CSR by destination) already resident in HBM when the timed region starts.
  workload  = BASELINE.json configs[2] (10M-host / 200M-edge R-MAT, the roofline config) by
              default; --config C2 selects configs[1] (1M/20M: fits the 256 MiB Infinity Cache, so
              it says little about HBM), C4 = configs[3], C5 = configs[4], LT = the long-tail graph.
  N > 1     = the same graph partitioned over the ranks (strong scaling), one RCCL collective of
              the counters per pass: --partition dest (default) = rows by owner,
              ncclAllGather of the owned slices; --partition edge = the north-star edge partition
              with ncclAllReduce(max, u8).
  roofline  = SURVEY.md §8(d): HBM-bound.  Headline `achieved`/`frac` = the WHOLE generic dense pass
              (dense passes t >= 1: B_t with their own A_t) over its measured GPU time (HIP events on
              the library's stream) vs 8 TB/s.  Pass 0 (68*m_eff + 192.25*n algorithmic bytes) is
              reported apart in `pass0`: the library streams the sources' single initial registers
              (2 B per edge) there instead of gathering counters, so it must not lift the headline.
              `dominant_kernel` = the level-1 hub-chunk launch alone: 68 B x the REAL edges it
              gathers (no partial-row traffic booked), over its own event-timed duration.
              `whole_loop` = sum_t B_t / t_loop with B_t = 68*A_t + 4*(m-A_t) + 184*V_t + 8n + n/4.
  parity    = every line carries `parity_bit_exact`: final (NodeID, f64) list vs the oracle when the
              CPU run converges inside its budget (always with --verify), else an order-independent
              checksum of all registers + Kahan state after the last pass the CPU finished.
  cpu_baseline = the CPU oracle's dense OpenMP port of the reference arithmetic, timed on
              this box's host cores on the same graph for a bounded number of passes; the GPU time
              of the SAME passes is reported next to it (like for like); `cpu_faithful` = the
              single-thread structure-faithful form on C1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("HB_BENCH_CONFIG", "C3"), help="C1|C2|C3|C4|C5|LT or scale:m")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline time bound (0 = skip)")
    ap.add_argument("--partition", default="dest", choices=["dest", "edge"],
                    help="N > 1: destination partition + all-gather per pass (default) or the north-star "
                         "edge partition + all-reduce(max) per pass")
    ap.add_argument("--changed-only", action="store_true",
                    help="N > 1, --partition dest: exchange only the counters that changed (HB_FLAG_CHANGED_ONLY)")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--tune", default="", help="comma separated hb_options.tune values")
    ap.add_argument("--verify", action="store_true", help="run the CPU oracle to convergence and compare the final result")
    ap.add_argument("--pass-log", default="", help="write per-pass stats JSON here")
    return ap.parse_args()


def pass_bytes(ps, n, m_eff, rows_with_in):
    """Algorithmic HBM bytes of one pass, SURVEY.md §8(d):
    B_t = 68*A_t + 4*(m_eff - A_t) + 184*V_t + 8*n + n/4; pass 0 = 68*m_eff + 192.25*n."""
    a = min(int(ps["active_edges"]), m_eff)
    if ps["pass"] == 0:
        a, v = m_eff, n
    elif ps["mode"] == 0:
        v = rows_with_in  # dense pass: every row with an in-edge has (almost surely) an active one
    else:
        v = int(ps["touched"])
    return 68.0 * a + 4.0 * (m_eff - a) + 184.0 * v + 8.0 * n + n / 4.0


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(a.gpus, 1):
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % a.gpus)
    import torch

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HyperBall library has no CPU fallback")
    torch.cuda.set_device(local_rank)
    td = None
    if world > 1:
        import torch.distributed as td

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from stract_amd import _lib, dist, synth

    # ---- synthetic input (identical on every rank, deterministic seed)
    t0 = time.perf_counter()
    g, scale, label = synth.make_config(a.config)
    t_gen = time.perf_counter() - t0
    n, m_eff = int(g.n), int(g.m)

    rccl_id = None
    flags = a.flags
    if world > 1:
        rccl_id = dist.torch_unique_id(rank, world)
        if a.partition == "dest":
            flags |= _lib.HB_FLAG_DEST_PARTITION
            if a.changed_only:
                flags |= _lib.HB_FLAG_CHANGED_ONLY
    tune = tuple(int(x) for x in a.tune.split(",")) if a.tune else ()
    ctx = _lib.Context(device=local_rank, flags=flags, chunk=a.chunk, rank=rank, world_size=world, rccl_id=rccl_id,
                       tune=tune)
    if world > 1:
        split = dist.partition_dense_by_dest if a.partition == "dest" else dist.partition_dense
        rp, src = split(g.row_ptr, g.src, rank, world)
    else:
        rp, src = g.row_ptr, g.src
    t0 = time.perf_counter()
    ctx.load_dense(g.ids, rp, src)
    t_load = time.perf_counter() - t0

    def barrier():
        torch.cuda.synchronize()
        if td is not None:
            td.barrier()
        torch.cuda.synchronize()

    def one_step():
        ctx.run()  # hb_begin + loop + hb_finish (normalise + result download), blocking
        return ctx.stats()

    for _ in range(a.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    loop_ms, gpu_ms, coll_ms, d2h_ms = 0.0, 0.0, 0.0, 0.0
    passes = 0
    all_pass_stats = []
    for _ in range(a.steps):
        st = one_step()
        passes = int(st["passes"])
        loop_ms += st["ms_loop"]
        gpu_ms += st["ms_loop_gpu"]
        coll_ms += st["ms_collective"]
        d2h_ms += st["ms_d2h"]
        all_pass_stats.append(ctx.pass_stats())
    barrier()
    dt = time.perf_counter() - t0
    if td is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())
    ids, vals = ctx.results()
    stats = ctx.stats()
    last_pass_stats = all_pass_stats[-1] if all_pass_stats else []

    # ---- parity + CPU baseline (rank 0 computes; a checksum rerun, if needed, is collective)
    cpu, parity = None, None
    if a.cpu_seconds > 0 or a.verify:
        cpu, parity = cpu_and_parity(a, g, ctx, td, rank, world, passes, ids, vals, last_pass_stats)

    if rank == 0:
        steps = max(a.steps, 1)
        teps = m_eff * passes * steps / dt
        rows_in = int(stats["rows_with_in_edges"])
        # per-pass averages over the timed steps
        T = len(last_pass_stats)
        avg = []
        for t in range(T):
            rows = [s[t] for s in all_pass_stats if len(s) == T]
            d = dict(rows[-1])
            for k in ("ms_gpu", "ms_main", "ms_level1", "ms_collective"):
                d[k] = float(np.mean([r[k] for r in rows]))
            d["alg_bytes"] = pass_bytes(d, n, m_eff, rows_in)
            avg.append(d)
        # the generic dense pass = dense passes t >= 1.  Pass 0 is reported on its own: since r02x it streams the sources'
        # single initial register with the edge list (2 B per edge) instead of gathering 64-byte counters, so its
        # 8(d) bytes are not what it moves and it must not lift the headline (HB_FLAG_NO_INIT_PASS restores the gathers)
        init_streamed = not (flags & _lib.HB_FLAG_NO_INIT_PASS)
        dense = [d for d in avg if d["mode"] == 0 and (d["pass"] > 0 or not init_streamed)]
        if not dense:
            dense = [d for d in avg if d["mode"] == 0]
        roof = None
        if dense:
            b_dense = sum(d["alg_bytes"] for d in dense)
            ms_dense = sum(d["ms_gpu"] - d["ms_collective"] for d in dense)
            achieved = b_dense / (ms_dense * 1e-3) / 1e9
            b_loop = sum(d["alg_bytes"] for d in avg)
            ms_loop_gpu = sum(d["ms_gpu"] for d in avg)
            l1_edges = int(stats["level1_edges"])
            l1_ms = float(np.mean([d["ms_level1"] for d in dense]))
            dom = None
            if l1_edges and l1_ms > 0:
                dom_b = 68.0 * l1_edges
                traffic, traffic_src = _pmc_traffic(a.config)
                dom = {"kernel": "hbk::pass_kernel<false,false,false,false,4> level-1 launch (dense pull over hub chunks)",
                       "alg_bytes_per_launch": dom_b, "alg_bytes_def": "68 B x real edges gathered by the launch (%d)" % l1_edges,
                       "avg_launch_ms": round(l1_ms, 4), "launches": len(dense) * steps,
                       "achieved": round(dom_b / (l1_ms * 1e-3) / 1e9, 1), "frac": round(dom_b / (l1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "traffic": traffic, "traffic_source": traffic_src}
            p0 = avg[0]
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": dom["traffic"] if dom else None,
                    "what": "whole dense pass (all launches of the pass), algorithmic bytes B_t of SURVEY.md 8(d) over "
                            "event-timed GPU time, mean over the %d dense passes t >= 1%s" % (
                                len(dense), " (pass 0 apart: it streams 2 B per edge instead of gathering, see pass0)" if init_streamed else ""),
                    "frac_of_measured_copy_6.29TBs": round(achieved / HBM_COPY_GBS, 4),
                    "pass0": {"alg_bytes": p0["alg_bytes"], "ms": round(p0["ms_gpu"] - p0["ms_collective"], 4),
                              "achieved": round(p0["alg_bytes"] / ((p0["ms_gpu"] - p0["ms_collective"]) * 1e-3) / 1e9, 1),
                              "streamed_initial_registers": bool(init_streamed),
                              "moved_bytes_model": (2.0 * m_eff + 192.25 * n) if init_streamed else p0["alg_bytes"]},
                    "dominant_kernel": dom,
                    "whole_loop": {"alg_bytes": b_loop, "ms_gpu": round(ms_loop_gpu, 4),
                                   "achieved": round(b_loop / (ms_loop_gpu * 1e-3) / 1e9, 1),
                                   "frac": round(b_loop / (ms_loop_gpu * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                    "per_pass": [{"t": int(d["pass"]), "mode": int(d["mode"]), "A_t": int(d["active_edges"]),
                                  "V_t": (n if d["pass"] == 0 else rows_in if d["mode"] == 0 else int(d["touched"])),
                                  "ms": round(d["ms_gpu"], 4),
                                  "frac": round(d["alg_bytes"] / (max(d["ms_gpu"], 1e-6) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                 for d in avg]}
        gathered = sum(min(int(d["active_edges"]), m_eff) if d["pass"] else m_eff for d in avg)
        n_pad = (n + 63) // 64 * 64
        wire = None
        if world > 1:
            wire = {"ran": a.partition + ("+changed-only" if (a.partition == "dest" and a.changed_only) else ""),
                    "received_bytes_per_gpu_per_run": int(stats["wire_bytes"]),
                    "edge_allreduce_bytes_per_gpu_per_pass": 2.0 * (world - 1) / world * n_pad * 64,
                    "dest_allgather_bytes_per_gpu_per_pass": (world - 1) / world * (n_pad * 64 + n_pad / 8),
                    "ms_collective_per_pass": round(coll_ms / steps / max(passes, 1), 4)}
        out = {
            "metric": "HyperBall traversed edges/sec (GTEPS)",
            "value": round(teps / 1e9, 4),
            "unit": "GTEPS",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt * 1e3 / steps, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s %s" % (a.config, label),
                       "n_hosts": n, "m_eff": m_eff, "passes_T": passes,
                       "parallelism": ("1 GPU" if world == 1 else
                                       "destination-partition x%d + allgather(u8)/pass" % world if a.partition == "dest" else
                                       "edge-partition x%d + allreduce(max,u8)/pass" % world)},
            "parity_bit_exact": None if parity is None else parity["bit_exact"],
            "parity": parity,
            "roofline": roof,
            "cpu_baseline": cpu,
            "detail": {"ms_loop_per_step": round(loop_ms / steps, 3), "ms_gpu_passes_per_step": round(gpu_ms / steps, 3),
                       "ms_collective_per_step": round(coll_ms / steps, 3), "ms_finish_per_step": round(d2h_ms / steps, 3),
                       "loop_gteps": round(m_eff * passes / (loop_ms / steps * 1e-3) / 1e9, 4) if loop_ms else None,
                       "gathered_edges_per_run": gathered,
                       "gathered_gteps": round(gathered / (loop_ms / steps * 1e-3) / 1e9, 4) if loop_ms else None,
                       "collective": wire,
                       "results": int(len(vals)), "s_generate": round(t_gen, 2), "s_load": round(t_load, 2),
                       "ms_plan": round(stats["ms_plan"], 1), "ms_h2d": round(stats["ms_h2d"], 1),
                       "device_bytes": int(stats["device_bytes"]), "virtual_rows": int(stats["virtual_rows"]),
                       "level1_edges": int(stats["level1_edges"]), "direct_edges": int(stats["direct_edges"])},
        }
        if a.pass_log:
            with open(a.pass_log, "w") as f:
                json.dump({"config": out["config"], "passes": avg}, f, indent=1)
        print(json.dumps(out), flush=True)
    ctx.close()
    if td is not None:
        td.barrier()
        td.destroy_process_group()


def _pmc_traffic(config):
    """HBM bytes per launch of the dominant kernel: NOT measured by this run - read from the newest
    committed rocprofv3 PMC summary (profiles/current_<config>_pmc.json, tools/export_profile.py);
    returns (bytes or None, source file or None)."""
    rel = os.path.join("profiles", "current_%s_pmc.json" % config)
    try:
        with open(os.path.join(ROOT, rel)) as f:
            d = json.load(f)
        for k, v in d.items():
            if "pass_kernel<false, false, false, false, 4>" in k:
                return v.get("hbm_bytes_per_dispatch"), rel + " (committed rocprofv3 --pmc run of the same command)"
    except Exception:
        pass
    return None, None


def _cpu_quota():
    """CPUs the cgroup lets this process use (cgroup v2 cpu.max or v1 cfs quota), or None."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(round(float(q) / float(p))))
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return max(1, int(round(q / p)))
    except Exception:
        pass
    return None


def _pick_threads(hbo, synth, ncpu, quota):
    """OpenMP thread count that is fastest on THIS box (SMT, NUMA, container CPU quotas): two dense passes
    of a small calibration graph per candidate."""
    cal = synth.RmatGraph(19, 4_000_000)
    best_t, cores = None, ncpu
    cands = {ncpu, max(ncpu // 2, 1), max(ncpu // 4, 1), min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}
    if quota:
        cands |= {min(ncpu, quota), min(ncpu, 2 * quota), min(ncpu, 4 * quota)}
    for th in sorted(cands, reverse=True):
        oc = hbo.Dense(cal.id_low64(), cal.row_ptr, cal.src, threads=th)
        oc.step(0)
        t0 = time.perf_counter()
        oc.step(0)
        oc.step(0)
        dt = time.perf_counter() - t0
        oc.close()
        if best_t is None or dt < best_t:
            best_t, cores = dt, th
    return cores


def cpu_and_parity(a, g, ctx, td, rank, world, gpu_passes, gpu_ids, gpu_vals, gpu_pass_stats):
    """Rank 0: oracle (dense OpenMP port of the reference arithmetic) on this box's host cores, on the
    same graph, for as many passes as fit in --cpu-seconds (all of them with --verify).  Parity: the final
    (NodeID, f64) list when the oracle converged, else a checksum of registers (+ Kahan state on one GPU)
    after the last pass the oracle finished - the GPU is re-run for that many passes (collectively for
    N > 1).  Returns (cpu_baseline dict or None, parity dict)."""
    import torch

    cpu, parity, done, o = None, None, 0, None
    if rank == 0:
        from oracle import hbo
        from stract_amd import synth

        ncpu = os.cpu_count() or 1
        quota = _cpu_quota()  # containers: the cgroup CPU quota can be far below the visible hardware threads
        cores = _pick_threads(hbo, synth, ncpu, quota)
        o = hbo.Dense(g.id_low64(), g.row_ptr, g.src, threads=cores)
        t0 = time.perf_counter()
        has = True
        cpu_pass_s = []
        while has and (a.verify or time.perf_counter() - t0 < a.cpu_seconds):
            t1 = time.perf_counter()
            has, _ = o.step(hbo.FRONTIER)
            cpu_pass_s.append(time.perf_counter() - t1)
            done += 1
        dt = time.perf_counter() - t0
        gpu_same_ms = sum(ps["ms_gpu"] for ps in gpu_pass_stats[:done])
        cpu = {"value": round(g.m * done / dt / 1e9, 5), "unit": "GTEPS", "cores": cores, "kind": "port",
               "sample": "first %d of %d passes of the same graph, oracle dense OpenMP port (oracle/hb_oracle.c), %.1f s, "
                         "%d OpenMP threads (fastest of a calibration sweep; %d hardware threads visible, cgroup CPU quota %s)"
                         % (done, gpu_passes, dt, cores, ncpu, quota if quota else "none"),
               "converged": not has, "seconds": round(dt, 3),
               "gpu_same_passes": {"passes": done, "ms": round(gpu_same_ms, 3),
                                   "gteps": round(g.m * done / (gpu_same_ms * 1e-3) / 1e9, 3) if gpu_same_ms else None,
                                   "speedup": round(dt * 1e3 / gpu_same_ms, 1) if gpu_same_ms else None}}
        if quota:
            cpu["cpu_quota"] = quota
        if not has:
            cpu["seconds_to_convergence"] = round(dt, 3)
            ovals, keep, k = o.finish()
            same = (done == gpu_passes and k == len(gpu_vals) and np.array_equal(gpu_ids, g.ids[keep]) and
                    np.array_equal(gpu_vals.view(np.uint64), ovals[keep].view(np.uint64)))
            parity = {"bit_exact": bool(same), "scope": "final (NodeID, f64) list after all %d passes, %d results" % (done, k),
                      "oracle": "oracle/hb_oracle.c dense form (parity unpinned against the Rust reference, see DESIGN.md)"}
        # the structure-faithful single-thread form (what `stract centrality harmonic` does), C1 only
        try:
            c1 = synth.RmatGraph(synth.CONFIGS["C1"]["scale"], synth.CONFIGS["C1"]["m"])
            _, _, fst = hbo.faithful_run(c1.edges())
            cpu["cpu_faithful"] = {"value": round(fst["m_eff"] * fst["passes"] / fst["seconds_loop"] / 1e9, 6), "unit": "GTEPS",
                                   "cores": 1, "sample": "C1 (%d hosts / %d edges), all %d passes, %.2f s, single thread, "
                                   "ordered map + per-pass clone + re-dedup + bloom (hbo.faithful_run)"
                                   % (fst["n"], fst["m_eff"], fst["passes"], fst["seconds_loop"])}
        except Exception as e:  # pragma: no cover
            cpu["cpu_faithful"] = {"error": str(e)}
    # did the oracle stop early?  then compare state checksums after `done` passes
    need = torch.tensor([done if (rank == 0 and parity is None) else 0], dtype=torch.int64, device="cuda")
    if td is not None:
        td.broadcast(need, src=0)
    k = int(need.item())
    if k > 0:
        ctx.begin()
        for _ in range(k):
            ctx.step()
        hr, hk = ctx.state_hash()
        if rank == 0:
            ohr, ohk = o.state_hash()
            same = (hr == ohr) and (world > 1 or hk == ohk)
            parity = {"bit_exact": bool(same),
                      "scope": "checksum of all %d x 64 registers%s after pass %d of %d (the CPU budget ended there)"
                               % (g.n, "" if world > 1 else " and of every Kahan (sum, err)", k, gpu_passes),
                      "oracle": "oracle/hb_oracle.c dense form (parity unpinned against the Rust reference, see DESIGN.md)"}
        ctx.run()  # leave the context finished (results valid)
    if rank == 0 and o is not None:
        o.close()
    if world > 1:
        cpu = None  # the CPU baseline is reported at N = 1 only; parity is reported at every N
    return cpu, parity


if __name__ == "__main__":
    main()
