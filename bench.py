#!/usr/bin/env python3
"""bench.py -- trial-source misfit evaluations per second on MI355X (BASELINE.json metric).

Workload (config.workload = "cfg3-bilat"): BASELINE.json configs[2], the configuration the
north_star target is quoted on -- `bilateral` extended rupture discretised into 100 centroids,
50 receivers x 3 components (n,e,d), 4096-sample Green's functions (ng = 10, bilinear
interpolation = 4 neighbour traces), time-domain L2 misfit over a 4096-sample tapered window.
One "step" = one pass of the hot path (geometry -> accumulate -> misfit) over a batch of
--batch trial sources per GPU (a strike sweep, kiwibench.py:136), everything already resident in
HBM.  N > 1: every rank evaluates its own contiguous shard of the trial list (weak scaling) and
the per-source global misfits are all-gathered over RCCL.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)

# 100 centroids: nx=10, ny=2, nt=5 at effective dt 0.5 (source_bilat.f90:274-315)
BENCH_BILAT = [0., 0., 0., 10000., 1e20, 91., 87., 164., 0., 4800., 2000., 2000., 3000., 2.]


def setup_product(device, nrec, L, batch, trial0):
    from kiwi_amd import Engine, synthetic
    from kiwi_amd.engine import discretize
    gf = synthetic.make_gfdb(L=L)
    lat, lon, depth, comps, dist = synthetic.make_receivers(nrec)
    p = Engine(device)
    p.set_database(gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"], gf["data"], gf["first"], gf["nsamp"])
    p.set_receivers(lat, lon, depth, comps)
    p.set_source_location(40.0, 30.0, 0.0)
    p.set_effective_dt(0.5)
    p.set_local_interpolation("bilinear")
    p.set_misfit_method("l2norm")
    dt = gf["dt"]
    # reference traces = synthetics of the "true" source over a 4096-sample window per receiver
    firsts = [int(round(d / 6000.0 / dt)) for d in dist]
    tapers = {}
    for ir in range(nrec):
        for k in range(3):
            p.set_ref_seismogram(ir + 1, k + 1, firsts[ir], np.zeros(L, np.float32))
        tapers[ir + 1] = synthetic.full_taper(firsts[ir], L, dt)
        p.set_misfit_taper(ir + 1, *tapers[ir + 1])
    # one full-size launch with the "true" source in slot 0 and the synthetics kept on the device
    # (every accumulate launch of this process then has the same size, so rocprof's per-kernel average
    # is the figure quoted in `roofline`)
    trials = synthetic.bilat_strike_sweep(batch, step=0.1, base=BENCH_BILAT)
    trials[:, 5] += 0.1 * trial0
    first = trials.copy()
    first[0] = np.array(BENCH_BILAT, np.float32)
    p.set_keep_synthetics(1)
    p.set_source_params("bilateral", first)
    p.eval()
    refs = {}
    for ir in range(nrec):
        for k in range(3):
            refs[(ir + 1, k + 1)] = p.get_synthetics(0, ir + 1, k + 1, 1)
    p.set_keep_synthetics(0)
    for (ir, k), (lo, d) in refs.items():
        p.set_ref_seismogram(ir, k, lo, d)
    p.set_source_params("bilateral", trials)
    for _ in range(3):          # bring clocks and caches to steady state before anything is timed
        p.eval()
    p.sync()
    ncent = len(discretize("bilateral", trials[0], 0.5)[0])
    return p, gf, (lat, lon, depth, comps), refs, tapers, trials, ncent


def cpu_baseline(gf, recv, refs, tapers, trials, gpu_global, budget_s=20.0):
    """The oracle (C restatement, OpenMP over receivers like minimizer_engine.f90:893-903) timed on
    this box's host cores for a bounded number of the SAME trial sources."""
    from oracle import ko
    cores = os.cpu_count() or 1
    nx, nz, ng, L = gf["data"].shape
    db = ko.Gfdb(nx, nz, ng, gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"])
    for ix in range(nx):
        for iz in range(nz):
            for ig in range(ng):
                db.set_trace(ix + 1, iz + 1, ig + 1, int(gf["first"][ix, iz, ig]), gf["data"][ix, iz, ig])
    e = ko.Engine(db)
    lat, lon, depth, comps = recv
    e.set_receivers(lat, lon, depth, comps)
    e.set_source_location(40.0, 30.0, 0.0)
    e.set_effective_dt(0.5)
    e.set_interpolation(True)
    e.set_nthreads(cores)
    for (ir, k), (lo, d) in refs.items():
        e.set_reference(ir, k, lo, d)
    for ir, (x, y) in tapers.items():
        e.set_taper(ir, x, y)
    e.set_source_params(1, trials[0])
    e.get_misfits()                                   # warm-up (allocations)
    t0 = time.perf_counter()
    e.set_source_params(1, trials[1 % len(trials)])
    e.get_misfits()
    t1 = time.perf_counter() - t0
    n = int(max(2, min(len(trials), budget_s / max(t1, 1e-3))))
    gl = []
    t0 = time.perf_counter()
    for i in range(n):
        e.set_source_params(1, trials[i])
        gl.append(e.get_misfits()[2])
    dtm = time.perf_counter() - t0
    gl = np.array(gl)
    err = float(np.max(np.abs(gpu_global[:n] - gl) / np.abs(gl)))
    e.close()
    db.close()
    return {"value": n / dtm, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "%d of the timed trial sources, oracle/libko.so, OpenMP over receivers" % n,
            "max_rel_misfit_diff_vs_gpu": err}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="trial sources per GPU per step")
    ap.add_argument("--receivers", type=int, default=50)
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    ngpus = world

    from kiwi_amd.shard import shard_range, gather_misfits
    lo, hi = shard_range(args.batch * ngpus, ngpus, rank)
    p, gf, recv, refs, tapers, trials, ncent = setup_product(local_rank, args.receivers, args.samples, hi - lo, lo)
    nmis = p.nmisfits()

    def step():
        p.eval()
        p.sync()
        _, _, g = p.get_misfits()
        return gather_misfits(g, dist, local_rank)

    for _ in range(args.warmup):
        allg = step()
    p.kernel_ms()                                   # reset the HIP-event accumulators
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allg = step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms, launches = p.kernel_ms()

    if rank == 0:
        total_evals = args.batch * ngpus * args.steps
        value = total_evals / elapsed
        n_ip, ng, L, W = 4, gf["data"].shape[2], args.samples, args.samples
        b_eval = ncent * args.receivers * ng * n_ip * L * 4 + args.receivers * 3 * W * 4 * 2
        acc_s = float(ms[1]) * 1e-3
        bytes_launched = b_eval * args.batch * args.steps           # rank 0's launches
        achieved = bytes_launched / acc_s / 1e9 if acc_s > 0 else 0.0
        # fabric (L2 <-> Infinity Cache / HBM) bytes per launch from the committed PMC passes of this same
        # command (profiles/*_summary.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE); PMC
        # counters cannot be collected from inside this process, so this is null when no profile matches
        traffic = None
        try:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
                prof = json.load(open(f))
                tb = prof.get("traffic_bytes_per_launch")
                if tb and tb.get("batch") and os.environ.get("KIWI_HIP_ACCUM") != "direct":
                    traffic = tb["total"] * args.batch / tb["batch"]
        except Exception:
            traffic = None
        out = {
            "metric": "trial-source misfit evals/s", "value": value, "unit": "evals/s",
            "n_gpus": ngpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg3-bilat: bilateral rupture, %d centroids x %d receivers x 3 comp x %d samples, "
                                   "ng=10, bilinear GF interpolation, time-domain l2norm, tapered %d-sample window"
                                   % (ncent, args.receivers, L, W),
                       "trial_sources_per_gpu_per_step": args.batch, "misfits_per_source": nmis,
                       "parallelism": "trial-source shard x%d, all-gather of global misfits" % ngpus},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "accumulate_grouped_kernel<10>" if os.environ.get("KIWI_HIP_ACCUM") != "direct" else "accumulate_kernel<10>", "launches": int(launches[1]),
                         "avg_launch_ms": float(ms[1]) / max(int(launches[1]), 1),
                         "algorithmic_bytes_per_eval": b_eval,
                         "other_kernels_ms_per_step": {"geometry": float(ms[0]) / args.steps,
                                                       "misfit": float(ms[2]) / args.steps}},
        }
        if ngpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(gf, recv, refs, tapers, trials, np.asarray(allg))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
