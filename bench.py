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
              it says little about HBM), C4 = configs[3].
  N > 1     = the same graph partitioned over the ranks (strong scaling), one RCCL collective of
              the counters per pass: --partition dest (default) = rows by owner,
              ncclAllGather of the owned slices; --partition edge = the north-star edge partition
              with ncclAllReduce(max, u8).
  roofline  = dominant kernel (dense pull over the hub chunks): algorithmic bytes per launch / its
              mean duration (HIP events on the library's stream) vs 8 TB/s; the whole dense pass
              (68*m_eff + 192.25*n bytes) is reported next to it.
  cpu_baseline = the CPU oracle's dense OpenMP port of the reference arithmetic, timed on
              this box's host cores on the same graph for a bounded number of passes.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("HB_BENCH_CONFIG", "C3"), help="C1|C2|C3|C4 or scale:m")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline time bound (0 = skip)")
    ap.add_argument("--partition", default="dest", choices=["dest", "edge"],
                    help="N > 1: destination partition + all-gather per pass (default) or the north-star "
                         "edge partition + all-reduce(max) per pass")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--tune", default="", help="comma separated hb_options.tune values")
    ap.add_argument("--verify", action="store_true", help="compare the final result with the oracle (full CPU run)")
    ap.add_argument("--pass-log", default="", help="write per-pass stats JSON here")
    return ap.parse_args()


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
    if a.config in synth.CONFIGS:
        cfg = synth.CONFIGS[a.config]
        scale, m_target, label = cfg["scale"], cfg["m"], cfg["label"]
    else:
        scale, m_target = (int(x) for x in a.config.split(":"))
        label = "R-MAT scale %d / %d edges" % (scale, m_target)
    t0 = time.perf_counter()
    g = synth.RmatGraph(scale, m_target)
    t_gen = time.perf_counter() - t0
    n, m_eff = int(g.n), int(g.m)

    rccl_id = None
    flags = a.flags
    if world > 1:
        rccl_id = dist.torch_unique_id(rank, world)
        if a.partition == "dest":
            flags |= _lib.HB_FLAG_DEST_PARTITION
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
    hub_ms, main_ms, dense_passes, loop_ms, gpu_ms, coll_ms, d2h_ms = 0.0, 0.0, 0, 0.0, 0.0, 0.0, 0.0
    passes = 0
    last_pass_stats = []
    for _ in range(a.steps):
        st = one_step()
        passes = int(st["passes"])
        loop_ms += st["ms_loop"]
        gpu_ms += st["ms_loop_gpu"]
        coll_ms += st["ms_collective"]
        d2h_ms += st["ms_d2h"]
        last_pass_stats = ctx.pass_stats()
        for ps in last_pass_stats:
            if ps["mode"] == 0:  # dense pass: hub-level launches (ev0..ev1) + the real-row launch (ev1..ev2)
                hub_ms += ps["ms_gpu"] - ps["ms_main"] - ps["ms_collective"]
                main_ms += ps["ms_main"]
                dense_passes += 1
    barrier()
    dt = time.perf_counter() - t0
    if td is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())
    ids, vals = ctx.results()
    stats = ctx.stats()

    if rank == 0:
        steps = max(a.steps, 1)
        teps = m_eff * passes * steps / dt
        # Dominant kernel (rocprofv3 --stats, profiles/): the dense pull over the hub chunks,
        # pass_kernel<REAL=false, FRONTIER=false, ...>, launched once per virtual level per dense pass.
        # Algorithmic bytes of one dense pass over the virtual rows: per gathered source 64 B counter + 4 B
        # index; per virtual row 64 B partial read + 64 B partial write + 8 B row pointer.  Per launch =
        # that / levels (the same average rocprofv3 reports for the kernel symbol).
        levels = max(int(stats["levels"]), 1)
        v_edges, v_rows = int(stats["virtual_edges"]), int(stats["virtual_rows"])
        roof = None
        if dense_passes and v_rows:
            alg_bytes = (68.0 * v_edges + 136.0 * v_rows) / levels
            avg_ms = hub_ms / dense_passes / levels
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            traffic = _pmc_traffic(a.config)
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": "hbk::pass_kernel<false,false,false,false,4> (dense pull over hub chunks)",
                    "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 4),
                    "launches": dense_passes * levels,
                    "whole_dense_pass": {"alg_bytes": 68.0 * m_eff + 192.25 * n,
                                         "avg_ms": round((hub_ms + main_ms) / dense_passes, 4),
                                         "achieved_GBs": round((68.0 * m_eff + 192.25 * n) /
                                                               ((hub_ms + main_ms) / dense_passes * 1e-3) / 1e9, 1)}}
        cpu = None
        if world == 1 and a.cpu_seconds > 0:
            cpu = cpu_baseline(g, a.cpu_seconds, passes, ids, vals, a.verify)
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
            "config": {"workload": "%s %s (R-MAT scale %d, a,b,c,d=.57,.19,.19,.05, seed 0x5712AC7)" % (a.config, label, scale),
                       "n_hosts": n, "m_eff": m_eff, "passes_T": passes,
                       "parallelism": ("1 GPU" if world == 1 else
                                       "destination-partition x%d + allgather(u8)/pass" % world if a.partition == "dest" else
                                       "edge-partition x%d + allreduce(max,u8)/pass" % world)},
            "roofline": roof,
            "cpu_baseline": cpu,
            "detail": {"ms_loop_per_step": round(loop_ms / steps, 3), "ms_gpu_passes_per_step": round(gpu_ms / steps, 3),
                       "ms_collective_per_step": round(coll_ms / steps, 3), "ms_finish_per_step": round(d2h_ms / steps, 3),
                       "loop_gteps": round(m_eff * passes / (loop_ms / steps * 1e-3) / 1e9, 4) if loop_ms else None,
                       "results": int(len(vals)), "s_generate": round(t_gen, 2), "s_load": round(t_load, 2),
                       "ms_plan": round(stats["ms_plan"], 1), "ms_h2d": round(stats["ms_h2d"], 1),
                       "device_bytes": int(stats["device_bytes"]), "virtual_rows": int(stats["virtual_rows"])},
        }
        if a.pass_log:
            with open(a.pass_log, "w") as f:
                json.dump({"config": out["config"], "passes": last_pass_stats}, f, indent=1)
        print(json.dumps(out), flush=True)
    ctx.close()
    if td is not None:
        td.barrier()
        td.destroy_process_group()


def _pmc_traffic(config):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/pmc_<config>.json, produced by tools/pmc_summary.py), or None."""
    p = os.path.join(ROOT, "profiles", "current_%s_pmc.json" % config)
    try:
        with open(p) as f:
            d = json.load(f)
        for k, v in d.items():
            if "pass_kernel<false, false, false, false, 4>" in k:
                return v.get("hbm_bytes_per_dispatch")
    except Exception:
        pass
    return None


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


def cpu_baseline(g, seconds, gpu_passes, gpu_ids, gpu_vals, verify):
    """Oracle (dense OpenMP port of the reference arithmetic) on this box's host cores, on the
    same graph, for as many passes as fit in `seconds` (all of them with --verify)."""
    from oracle import hbo

    ncpu = os.cpu_count() or 1
    quota = _cpu_quota()  # containers: the cgroup CPU quota can be far below the visible hardware threads
    # pick the OpenMP thread count that is fastest on THIS box (all hardware threads is not always best:
    # SMT, NUMA, container CPU quotas): two dense passes of a small calibration graph per candidate
    from stract_amd import synth
    cal = synth.RmatGraph(19, 4_000_000)
    best_t, cores = None, ncpu
    cands = {ncpu, max(ncpu // 2, 1), max(ncpu // 4, 1), min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}
    if quota:
        cands |= {min(ncpu, quota), min(ncpu, 2 * quota)}
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
    o = hbo.Dense(g.id_low64(), g.row_ptr, g.src, threads=cores)
    t0 = time.perf_counter()
    done, has = 0, True
    while has and (verify or time.perf_counter() - t0 < seconds):
        has, _ = o.step(hbo.FRONTIER)
        done += 1
    dt = time.perf_counter() - t0
    res = {"value": round(g.m * done / dt / 1e9, 5), "unit": "GTEPS", "cores": cores, "kind": "port",
           "sample": "first %d of %d passes of the same graph, oracle dense OpenMP port (oracle/hb_oracle.c), %.1f s, "
                     "%d OpenMP threads (fastest of a calibration sweep; %d hardware threads visible, cgroup CPU quota %s)"
                     % (done, gpu_passes, dt, cores, ncpu, quota if quota else "none")}
    if quota:
        res["cpu_quota"] = quota
    if not has:  # converged inside the budget: a free end-to-end parity check
        vals, keep, k = o.finish()
        same = (done == gpu_passes and k == len(gpu_vals) and np.array_equal(gpu_ids, g.ids[keep]) and
                np.array_equal(gpu_vals.view(np.uint64), vals[keep].view(np.uint64)))
        res["parity_bit_exact"] = bool(same)
    return res


if __name__ == "__main__":
    main()
