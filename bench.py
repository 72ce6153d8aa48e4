#!/usr/bin/env python3
"""bench.py -- trial-source misfit evaluations per second on MI355X (BASELINE.json metric).

Default workload (config.workload = "cfg3-bilat"): BASELINE.json configs[2], the configuration the
north_star target is quoted on -- `bilateral` extended rupture discretised into 100 centroids,
50 receivers x 3 components (n,e,d), 4096-sample Green's functions (ng = 10, bilinear
interpolation = 4 neighbour traces), time-domain L2 misfit over a 4096-sample tapered window.
`--workload cfg2|cfg4|cfg5` run the other BASELINE.json configurations (moment-tensor grid;
mt_eikonal 468 centroids x 200 receivers; spectral comparator with frequency filter) the same way;
`cfg3-100pt` is the same source type with 100 sub-fault POINTS (200 centroids), `cfg3-scatter` the
cfg3 source over a shuffled location grid (no Green's function rows shared between neighbouring trials), `cfg5-td` the cfg5
trial set under a time-domain norm on frequency-filtered traces.
One "step" = one pass of the hot path (geometry -> accumulate -> misfit) over a batch of
--batch trial sources per GPU, everything already resident in HBM.  N > 1: every rank evaluates
its own contiguous shard of the trial list (weak scaling) and the per-source global misfits are
all-gathered over RCCL; `python bench.py --gpus N` starts the N ranks itself.

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
# Peak FP32 (vector) 157.3 TFLOP/s counts fused multiply-adds.  Under the library's default arithmetic contract ("exact": the
# reference rounds every multiply and every add on its own, parity is bit-level) a lane does one flop per packed slot, half
# that rate; under the "fused" contract (KIWI_HIP_ARITH=fused / kiwi_hip_set_arithmetic: multiply + consuming add as one FMA,
# tolerance class) the kernels are priced against the full figure.
FP32_PEAK_TFLOPS = 157.3


def valu_peak(arith):
    return FP32_PEAK_TFLOPS if arith == "fused" else FP32_PEAK_TFLOPS / 2.0


VALU_PEAK_TFLOPS = FP32_PEAK_TFLOPS / 2.0


def valu_roofline(ach_tflops, arith, **more):
    """roofline block of a vector-ALU-bound kernel: `peak` / `frac` against the rate the kernel's arithmetic contract allows (no
    fused multiply-add under `exact`: one flop per lane and packed slot), and ALWAYS next to them the guide's figure,
    MI355X_MICROARCH.md "Peak FP32 (vector)" 157.3 TFLOP/s, which counts a fused multiply-add as two flops (`peak_fma`,
    `frac_of_fma_peak`)."""
    out = {"bound": "valu_issue", "achieved": ach_tflops, "peak": valu_peak(arith), "unit": "TFLOP/s", "frac": ach_tflops / valu_peak(arith),
           "peak_is": "fp32 vector rate WITHOUT fused multiply-add (the exact contract rounds every multiply and add on its own)"
                      if arith != "fused" else "fp32 vector rate with fused multiply-add",
           "peak_fma": FP32_PEAK_TFLOPS, "frac_of_fma_peak": ach_tflops / FP32_PEAK_TFLOPS, "arithmetic": arith}
    out.update(more)
    return out
NORM_ID = {"l2norm": 1, "l1norm": 2, "ampspec_l2norm": 3, "ampspec_l1norm": 4}


def setup_product(device, wl, L):
    from kiwi_amd import Engine, synthetic
    from kiwi_amd.engine import discretize, discretize_eikonal
    nrec = wl["nrec"]
    gf = synthetic.make_gfdb(nx=wl["nx"], nz=wl.get("nz", 6), L=L, ng=wl.get("ng", 10), variant=wl.get("variant", "probe"))
    lat, lon, depth, comps, dist = synthetic.make_receivers(nrec)
    p = Engine(device)
    # the engine gets the traces as a database reader delivers them: gap-compressed spans (trace_pack), which is also
    # how the CPU oracle stores the same array -- the comparator's transform lengths follow these spans
    pk = synthetic.pack_gfdb(gf)
    p.set_database(gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"], pk["data"], pk["first"], pk["nsamp"])
    del pk
    p.set_receivers(lat, lon, depth, comps)
    p.set_source_location(40.0, 30.0, 0.0)
    p.set_effective_dt(0.5)
    p.set_local_interpolation("bilinear")
    p.set_misfit_method(wl["method"])
    if wl["crust"] is not None:
        p.set_source_crust(wl["crust"], wl["crust"])
        p.set_source_constraints(*wl["constraints"])
    dt = gf["dt"]
    # reference traces = synthetics of the "true" source over an L-sample window per receiver
    firsts = [int(round(d / 6000.0 / dt)) for d in dist]
    tapers = {}
    for ir in range(nrec):
        for k in range(3):
            p.set_ref_seismogram(ir + 1, k + 1, firsts[ir], np.zeros(L, np.float32))
        # misfit window: the whole trace, or (cfg3-w256 / cfg3-w600) a short taper inside it -- the window is the taper's span
        # (comparator.f90:782-792)
        if wl.get("window"):
            tapers[ir + 1] = synthetic.full_taper(firsts[ir] + wl.get("window_offset", 0), wl["window"], dt, ramp=min(10.0, 0.2 * wl["window"] * dt))
        else:
            tapers[ir + 1] = synthetic.full_taper(firsts[ir], L, dt)
        p.set_misfit_taper(ir + 1, *tapers[ir + 1])
        if wl["filter"] is not None:
            p.set_misfit_filter(ir + 1, *wl["filter"])
    # one full-size launch with the "true" source in slot 0 and the synthetics kept on the device
    # (every accumulate launch of this process then has the same size, so rocprof's per-kernel average
    # is the figure quoted in `roofline`)
    trials = wl["trials"]
    first = trials.copy()
    first[0] = wl["true"]
    p.set_keep_synthetics(1)
    p.set_source_params(wl["sourcetype"], first)
    p.eval()
    refs = {}
    for ir in range(nrec):
        for k in range(3):
            refs[(ir + 1, k + 1)] = p.get_synthetics(0, ir + 1, k + 1, 1)
    p.set_keep_synthetics(0)
    for (ir, k), (lo, d) in refs.items():
        p.set_ref_seismogram(ir, k, lo, d)
    p.set_source_params(wl["sourcetype"], trials)
    for _ in range(3):          # bring clocks and caches to steady state before anything is timed
        p.eval()
    p.sync()
    sample = trials[:: max(1, len(trials) // 16)]
    if wl["crust"] is not None:
        st = 4 if wl["sourcetype"] == "eikonal" else 5
        tabs = [discretize_eikonal(st, t, 0.5, wl["crust"], *wl["constraints"])[0] for t in sample]
    else:
        tabs = [discretize(wl["sourcetype"], t, 0.5)[0] for t in sample]
    nc = [len(t) for t in tabs]
    # sub-fault POINTS (runs of centroids at one position: the time steps of a sub-fault share their blended traces)
    wl["npoints"] = float(np.mean([1 + np.count_nonzero(np.any(t[1:, :3] != t[:-1, :3], axis=1)) for t in tabs]))
    return p, gf, (lat, lon, depth, comps), refs, tapers, float(np.mean(nc))


def oracle_engine(wl, gf, recv, refs, tapers, cores, fresh=False):
    """The CPU oracle set up exactly like the product in setup_product; returns (engine, db, evaluate) where
    evaluate(params) = set_source_params + get_misfits (seismosizer.py:703-718) -> (misfits, norms, global).
    fresh=True: every evaluation runs on a NEW engine over the same database -- probes and strips of the reference
    never shrink, so its spans (and with them the spectral comparator's transform lengths) remember every source
    evaluated before; the product gives each trial source what a fresh engine gives it."""
    from oracle import ko
    from kiwi_amd.engine import SOURCE_TYPES
    nx, nz, ng, L = gf["data"].shape
    db = ko.Gfdb(nx, nz, ng, gf["dt"], gf["dx"], gf["dz"], gf["firstx"], gf["firstz"])
    for ix in range(nx):
        for iz in range(nz):
            for ig in range(ng):
                db.set_trace(ix + 1, iz + 1, ig + 1, int(gf["first"][ix, iz, ig]), gf["data"][ix, iz, ig])
    lat, lon, depth, comps = recv

    def make():
        e = ko.Engine(db)
        e.set_receivers(lat, lon, depth, comps)
        e.set_source_location(40.0, 30.0, 0.0)
        e.set_effective_dt(0.5)
        e.set_interpolation(True)
        e.set_nthreads(cores)
        e.set_misfit_method(NORM_ID[wl["method"]])
        for (ir, k), (lo, d) in refs.items():
            e.set_reference(ir, k, lo, d)
        for ir, (x, y) in tapers.items():
            e.set_taper(ir, x, y)
            if wl["filter"] is not None:
                e.set_filter(ir, *wl["filter"])
        return e

    e = make()
    st = SOURCE_TYPES[wl["sourcetype"]]
    if wl["crust"] is not None:
        c = wl["crust"]
        prof = ko.crust_profile(c[0:8], c[8:16], c[16:24], c[24:31])

    def evaluate(t, inspect=None):
        """inspect: called with the engine right after the evaluation (tests read the probes' scales there); its result is
        appended to (misfits, norms, global)"""
        eng = make() if fresh else e
        try:
            if wl["crust"] is not None:
                cent, mo, ri, _ = ko.discretize_eikonal(st, t, 0.5, prof, *wl["constraints"])
                eng.set_centroids(cent, mo, ri)
            else:
                eng.set_source_params(st, t)
            res = eng.get_misfits()
            return res if inspect is None else tuple(res) + (inspect(eng),)
        finally:
            if fresh:
                eng.close()

    return e, db, evaluate


def cpu_baseline(wl, gf, recv, refs, tapers, gpu_global, gpu_misfits, gpu_norms, budget_s=20.0, one_core=True):
    """The oracle (C restatement, OpenMP over receivers like minimizer_engine.f90:893-903) timed on
    this box's host cores for a bounded number of the SAME trial sources."""
    # the reference parallelises make_seismogram over receivers (minimizer_engine.f90:893-903): no more threads than
    # receivers can do work
    from kiwi_amd import lib as _lib
    avail = int(_lib.load().kiwi_hip_effective_cpus())      # hardware threads cut to the container's CPU quota
    cores = min(avail, wl["nrec"])
    trials = wl["trials"]
    e, db, evaluate = oracle_engine(wl, gf, recv, refs, tapers, cores)

    def one(t):
        return evaluate(t)

    one(trials[0])                                    # warm-up (allocations)
    t0 = time.perf_counter()
    one(trials[1 % len(trials)])
    t1 = time.perf_counter() - t0
    n = int(max(2, min(len(trials), budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    res = [one(trials[i]) for i in range(n)]
    dtm = time.perf_counter() - t0
    gl = np.array([r[2] for r in res])
    # (a grid that holds the true source has a trial whose misfit is exactly zero on both sides: absolute there)
    err = float(np.max(np.abs(gpu_global[:n] - gl) / np.where(gl != 0.0, np.abs(gl), 1.0)))
    # every per-receiver-component misfit of the sample, not only the global one: difference relative to the slot's
    # norm factor (a misfit can be arbitrarily close to zero), and the norm factors themselves
    om = np.array([r[0] for r in res], np.float64)
    on = np.array([r[1] for r in res], np.float64)
    slot_err = float(np.max(np.abs(gpu_misfits[:n] - om) / on))
    norm_err = float(np.max(np.abs(gpu_norms[:n] - on) / on))
    # the same on ONE core (SURVEY 8d asks for both), two sources
    v1 = None
    if one_core:
        e.set_nthreads(1)
        t0 = time.perf_counter()
        n1 = 2
        for i in range(n1):
            one(trials[i])
        v1 = n1 / (time.perf_counter() - t0)
    e.close()
    db.close()
    return {"value": n / dtm, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "%d of the timed trial sources, oracle/libko.so, OpenMP over the %d receivers (host: %d hardware threads, "
                      "CPU quota of this container %d)" % (n, wl["nrec"], os.cpu_count() or 1, avail),
            "value_1core": v1,
            "reference_in_dev_container": "the reference itself (amdflang -O2, 135 centroids, 8 vCPU Xeon 2.1 GHz): 0.76 evals/s "
                                          "on 8 threads, 0.11 on one (BASELINE.md section 2); this port there: 6.2 on 8 threads",
            "max_rel_misfit_diff_vs_gpu": err,
            "max_slot_misfit_diff_vs_gpu_rel_to_norm": slot_err, "max_rel_norm_factor_diff_vs_gpu": norm_err}


def _finite(x):
    """JSON has no NaN/Infinity: a figure that is not finite is reported as null."""
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, float) and not np.isfinite(x):
        return None
    return x


def measured_copy_bandwidth(torch, device):
    """What a plain device-to-device copy reaches on this box (read + written bytes per second): the practical HBM
    ceiling next to the nominal 8 TB/s."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    ev0.record()
    for _ in range(reps):
        b.copy_(a)
    ev1.record()
    torch.cuda.synchronize()
    return 2.0 * n * reps / (ev0.elapsed_time(ev1) * 1e-3) / 1e9


def visible_gpus():
    """GPUs of this node without initialising HIP / HSA in the calling process: KFD topology nodes with compute units
    (CPU nodes have simd_count 0), cut by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES where set."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n):
    """Parent of a multi-GPU run: one child process per GPU through torch.distributed.run (RCCL rendezvous on 127.0.0.1).
    Nothing is retried or restarted in place: a failing rank ends the run with a non-zero status."""
    import socket
    import subprocess
    have = visible_gpus()                            # from sysfs: nothing in this process touches the HIP runtime
    if have < n:
        print("bench.py: --gpus %d requested but only %d GPU(s) are visible" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as so:                      # a free rendezvous port
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(os.cpu_count() or n, 64) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "--max-restarts", "0",
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


KERNEL_SOURCES = ("kiwi_common.hpp", "kiwi_geometry.hpp", "kiwi_accum.inc", "kiwi_accum.hip", "kiwi_apply_asm.inc", "kiwi_accum_api.hpp",
                  "kiwi_misfit.hpp", "kiwi_hip.hip", "kiwi_libm32.hpp", "Makefile")


def _code_only(text, makefile=False):
    """A source text without its comments and blank space: what the compiler is given.  (A comment edit must not orphan the
    committed counters, and nobody should have to re-stamp a summary by hand after one.)  String literals are left alone."""
    import re
    if makefile:
        return "\n".join(l.rstrip() for l in text.split("\n") if l.strip() and not l.lstrip().startswith("#"))
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == "'" and i + 2 < n and (text[i + 2] == "'" or (text[i + 1] == "\\" and i + 3 < n and text[i + 3] == "'")):
            j = i + (3 if text[i + 2] == "'" else 4)    # character literal ('"' must not open a string)
            out.append(text[i:j])
            i = j
        elif c == '"':                                 # string literal: copy through its closing quote
            j = i + 1
            while j < n and text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return "\n".join(" ".join(l.split()) for l in "".join(out).split("\n") if l.strip())


def kernel_sources_sha256():
    """Hash of the CODE the device side is built from (comments and blank space stripped): committed counters are attached to a
    bench line only when they were collected on exactly this code.  profiles/summarize.py computes the same figure at collection
    time, next to the hashes of the raw rocprofv3 files the summary was condensed from."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "kiwi_amd", "csrc", f), "r", errors="replace") as fh:
            h.update(_code_only(fh.read(), makefile=(f == "Makefile")).encode())
            h.update(b"\0")
    # the extra flags the loaded library was really built with (make EXTRA=-D...): same sources, other device code
    try:
        import ctypes
        from kiwi_amd import lib as _lib
        buf = ctypes.create_string_buffer(1024)
        if _lib.load().kiwi_hip_build_flags(buf, 1024) == 0 and buf.value:
            h.update(b"EXTRA=" + buf.value)
    except Exception:
        pass
    return h.hexdigest()


def timed_evals(p, steps):
    """`steps` passes of the hot path on an engine that is set up and warm -> (wall s, accumulate-kernel ms per pass)"""
    p.sync()
    p.kernel_ms()
    t0 = time.perf_counter()
    for _ in range(steps):
        p.eval()
        p.sync()
        p.get_misfits()
    dt = time.perf_counter() - t0
    ms, launches = p.kernel_ms()
    return dt, float(ms[1]) / steps          # (accumulate ms per pass: a pass over a large batch is several launches)


def required_flops(ncent, npts, nrec, ng, W):
    """What the accumulate kernel has to execute per trial source: fp32 multiplies and adds in the
    reference's order -- apply: 4 per GF component, centroid and output sample + 8 for the per-centroid rotation
    (seismogram.f90:171-250, sparse_trace.f90:684-703); blend: 7 per component and sample, once per sub-fault POINT and
    receiver (gfdb.f90:944-949; the time steps of a point share the blended trace).  The same count under both arithmetic
    contracts (a fused multiply-add is two of them)."""
    return ncent * nrec * W * (4 * ng + 8) + npts * nrec * W * ng * 7


def other_contract(p, batch, flops_eval, steps=6):
    """The same resident batch under the OTHER arithmetic contract of the accumulate kernels (exact <-> fused), timed the same
    way; the engine is left in the contract it came in.  `fused_vs_exact`: the worst difference of the two contracts' results over
    the whole timed batch -- per-slot misfits relative to the misfit itself and relative to the slot's norm factor (the fused
    contract promises 1e-6 of the larger of the two), global misfits relative to sqrt(g^2 + 1)."""
    was = p.arithmetic()
    oth = "exact" if was == "fused" else "fused"
    p.eval()
    p.sync()
    m0, n0, g0 = (np.asarray(x, np.float64).copy() for x in p.get_misfits())
    p.set_arithmetic(oth)
    for _ in range(2):
        p.eval()
    dt, acc_ms = timed_evals(p, steps)
    m1, n1, g1 = (np.asarray(x, np.float64) for x in p.get_misfits())
    p.set_arithmetic(was)
    (me, ge), (mf, gf_) = ((m0, g0), (m1, g1)) if was == "exact" else ((m1, g1), (m0, g0))
    dm = np.abs(mf - me)
    delta = {"max_abs_diff_over_misfit": float(np.max(dm / np.maximum(np.abs(me), 1e-300))),
             "max_abs_diff_over_norm_factor": float(np.max(dm / np.maximum(n0, 1e-300))),
             "max_abs_diff_over_max_of_both": float(np.max(dm / np.maximum(np.maximum(np.abs(me), n0), 1e-300))),
             "max_global_misfit_diff_over_sqrt_g2_plus_1": float(np.max(np.abs(gf_ - ge) / np.sqrt(ge * ge + 1.0))),
             "norm_factors_identical": bool(np.array_equal(n0, n1)), "sources": int(len(ge)), "slots": int(me.shape[1])}
    ach = flops_eval * batch / (acc_ms * 1e-3) / 1e12 if acc_ms > 0 else 0.0
    return {"arithmetic": oth, "value": batch * steps / dt, "unit": "evals/s", "steps": steps, "accumulate_ms_per_step": acc_ms,
            "roofline": valu_roofline(ach, oth),
            "fused_vs_exact": delta}


def host_inclusive(p, wl, value_resident, reps=3, piece=0, longer=None):
    """The WHOLE path of north_star for the same trial sources: parameter list -> host discretiser (A2-A5) -> upload of the centroid
    tables -> geometry / accumulate / misfit kernels -> download of every misfit, through the one call the Python and Fortran hosts
    use for a trial list (kiwi_hip_misfits_for_params: the discretiser of one piece runs under the device's evaluation of another);
    the loop it replaces is seismosizer.py:682-722.  Nothing is resident when the clock starts except the database, receivers and
    references.  Leaves the engine holding the head piece of the list."""
    trials = wl["trials"]
    p.eval()
    p.sync()
    gm, gn, gg = (x.copy() for x in p.get_misfits())
    nhead = len(trials)
    if longer is not None:
        # eikonal workloads: a step is ONE piece of the one-call evaluation (128 fast-marching solves) -- nothing to overlap.  The
        # figure is taken on a list of several pieces whose head is the step's batch (the results of the head are compared)
        trials = longer
    p.misfits_for_params(wl["sourcetype"], trials, piece)            # (buffers of the piece size)
    t0 = time.perf_counter()
    for _ in range(reps):
        m, n, g, st = p.misfits_for_params(wl["sourcetype"], trials, piece)
    dt = (time.perf_counter() - t0) / reps
    return {"value": len(trials) / dt, "unit": "evals/s", "ms_per_step": dt * 1e3, "steps": reps, "trial_sources_per_step": int(len(trials)),
            "frac_of_resident": len(trials) / dt / value_resident if value_resident > 0 else None,
            "identical_to_resident": bool(np.array_equal(m[:nhead], gm) and np.array_equal(n[:nhead], gn) and np.array_equal(g[:nhead], gg)),
            "failed_sources": int(np.count_nonzero(st)),
            "timed": "parameter list -> host discretiser -> H2D -> kernels -> D2H of all misfits (kiwi_hip_misfits_for_params, "
                     "pieces of %d sources, discretiser overlapped with the device)" % (piece or (128 if wl["crust"] is not None else 2048))}


def sweep_block(device, L, n=20000, first=40000, workload="cfg5"):
    """A slice of BASELINE.json configs[4] in the default line: `n` consecutive points of the 10^5-point (strike, dip, slip-rake, depth,
    time) grid that hold the source the references were made from, host-inclusive (one kiwi_hip_misfits_for_params call), spectral
    comparator with frequency filter; the argmin of the global misfits has to be the planted source.  (`--sweep 100000` runs the
    whole grid, sharded over the ranks.)"""
    from kiwi_amd import synthetic
    wl = synthetic.workload(workload, 512, first)
    p, gf, recv, refs, tapers, ncent = setup_product(device, wl, L)
    trials = synthetic.workload(workload, n, first)["trials"]
    p.misfits_for_params(wl["sourcetype"], trials[:2048])              # warm-up: transform plans, buffers
    p.sync()
    t0 = time.perf_counter()
    m, nn, g, st = p.misfits_for_params(wl["sourcetype"], trials)
    dt = time.perf_counter() - t0
    best = int(np.nanargmin(np.where(np.isfinite(g), g, np.inf)))
    same = np.where(np.all(trials == wl["true"][None, :], axis=1))[0]
    p.close()
    return {"workload": "%s: points %d .. %d of the 10^5-point source-parameter grid, %s%s, %d receivers x 3 comp x %d samples"
                        % (wl["name"], first, first + n - 1, wl["method"], " + frequency filter" if wl["filter"] is not None else "", wl["nrec"], L),
            "value": n / dt, "unit": "evals/s", "wall_s": dt, "trial_sources": n, "argmin": best, "argmin_misfit": float(g[best]),
            "true_source_index": int(same[0]) if len(same) else None, "argmin_is_true_source": bool(len(same) and best == int(same[0])),
            "failed_sources": int(np.count_nonzero(st)),
            "timed": "parameter list -> host discretiser -> H2D -> kernels -> D2H (inputs NOT resident), one call"}


def also_workload(device, L, name, batch, steps=3, cpu_budget_s=3.0, host_pieces=0):
    """Another workload in the driver's line: `steps` passes of the hot path over `batch` resident trial sources,
    timed like the main figure (wall clock around eval + sync + download of the misfits), with its own roofline block (required
    flops over the accumulate kernels' HIP-event time) and a SHORT CPU baseline (the oracle on the host cores for `cpu_budget_s`
    seconds of the same trial sources, results compared with the device's)."""
    from kiwi_amd import synthetic
    wl = synthetic.workload(name, batch, 0)
    p, gf, recv, refs, tapers, ncent = setup_product(device, wl, L)
    for _ in range(2):
        p.eval()
    dt, acc_ms = timed_evals(p, steps)
    npts, nrec, ng, W = wl["npoints"], wl["nrec"], gf["data"].shape[2], L
    flops_eval = required_flops(ncent, npts, nrec, ng, W)
    ar = p.arithmetic()
    ach = flops_eval * batch / (acc_ms * 1e-3) / 1e12 if acc_ms > 0 else 0.0
    gm, gn, gg = p.get_misfits()
    out = {"workload": "%s: %s source, %.0f centroids (%.0f sub-fault points) x %d receivers x 3 comp x %d samples, %s"
                       % (wl["name"], wl["sourcetype"], ncent, npts, nrec, W, wl["method"]),
           "arithmetic": ar, "value": batch * steps / dt, "unit": "evals/s", "trial_sources_per_step": batch, "steps": steps,
           "ms_per_step": dt / steps * 1e3,
           "roofline": valu_roofline(ach, ar, accumulate_ms_per_step=acc_ms, flops_per_eval=flops_eval)}
    try:
        out["cpu_baseline"] = cpu_baseline(wl, gf, recv, refs, tapers, np.asarray(gg), gm, gn, budget_s=cpu_budget_s, one_core=False) if cpu_budget_s > 0 else None
    except Exception as ex:                      # (the secondary blocks must not take the line down)
        out["cpu_baseline"] = {"error": str(ex)[:200]}
    if host_pieces > 0:
        # the whole path for a list of `host_pieces` pieces whose head is this batch: parameter list -> discretiser (a fast-marching solve
        # per trial for the rupture-shape sweep) -> upload -> kernels -> download, the discretiser running ahead of the device
        longer = synthetic.workload(name, host_pieces, 0)["trials"] if host_pieces > 64 else synthetic.workload(name, host_pieces * batch, 0)["trials"]
        out["host_inclusive"] = host_inclusive(p, wl, out["value"], longer=longer, reps=2 if host_pieces > 64 else 3)
    p.close()
    return out


def also_bigdb4(device, L, batch=1024, steps=6):
    """The HBM regime in the driver's line: the cfg3 source over a 4.2 GB Green's function database (sixteen times the Infinity
    Cache), one trial location per distance node in shuffled order -- the one workload whose accumulate kernel runs against the
    memory, i.e. where north_star's "fraction of the HBM roofline" is the meaningful figure.

    One definition per key.  `achieved_counter_gbs`: memory-side bytes of the committed rocprofv3 counters of THESE
    kernel sources (FETCH_SIZE x 2 + WRITE_SIZE, scaled to the batch; Infinity-Cache hits included; null without a matching
    profile) over the accumulate kernels' HIP-event time of this run.  `hbm_side_estimate_gbs`: that minus the share the Infinity
    Cache can have served (at most 1/16 of a pass over a database 16 x its size).  `no_reuse_model_gbs`: the SURVEY 8d byte model
    (every (point, receiver) pair fetches its own 40 rows over the window; an UPPER bound of the traffic -- it counts again the
    rows co-resident workgroups of neighbouring receivers share in L2, which is how it came out above the chip's peak in round 5).
    `pure_read_ceiling_gbs`: measured in THIS run by the library's own read kernel (kiwi_hip_measure_read_bandwidth).
    `achieved` (and `frac` = achieved / 8 TB/s) is the HBM-side estimate when there are counters, otherwise the model -- and no
    figure of the block is reported above a ceiling of the block: one that comes out higher is capped and says so in `notes`."""
    from kiwi_amd import synthetic
    import ctypes
    wl = synthetic.workload("cfg3-bigdb4", batch, 0)
    p, gf, recv, refs, tapers, ncent = setup_product(device, wl, L)
    for _ in range(2):
        p.eval()
    dt, acc_ms = timed_evals(p, steps)
    ar = p.arithmetic()
    npts, nrec, ng, W = wl["npoints"], wl["nrec"], gf["data"].shape[2], L
    t = acc_ms * 1e-3
    model_bytes = npts * nrec * ng * 4 * W * 4.0 * batch
    flops_eval = required_flops(ncent, npts, nrec, ng, W)
    prof, note = committed_counters("cfg3-bigdb4", ar)
    traffic = prof["hbm_bytes_per_launch"] * (batch / prof["batch"]) if prof.get("hbm_bytes_per_launch") else None
    gbs_now = ctypes.c_double(0.0)
    try:
        p._ck(p.L.kiwi_hip_measure_read_bandwidth(p.h, 4 << 30, 10, ctypes.byref(gbs_now)), "measure_read_bandwidth")
        ceiling = float(gbs_now.value)
    except Exception:
        ceiling = None
    notes = []
    top = min(x for x in (ceiling, HBM_PEAK_GBS) if x)

    def capped(v, what):
        if v is not None and v > top:
            notes.append("%s came out at %.0f GB/s, above the lower of the block's ceilings (%.0f GB/s): reported capped" % (what, v, top))
            return top
        return v

    model = model_bytes / t / 1e9 if t > 0 else 0.0
    cnt = capped(traffic / t / 1e9 if traffic and t > 0 else None, "achieved_counter_gbs")
    hbm_side = cnt * 15.0 / 16.0 if cnt else None
    achieved = hbm_side if hbm_side else capped(model, "no_reuse_model_gbs (used as `achieved`: no counters of these kernel sources)")
    out = {"workload": "cfg3-bigdb4: cfg3 source, %.0f centroids (%.0f points) x %d receivers, database of %.1f GB (%d x %d nodes), one trial "
                       "location per distance node, shuffled" % (ncent, npts, nrec, gf["data"].nbytes / 1e9, gf["data"].shape[0], gf["data"].shape[1]),
           "arithmetic": ar, "value": batch * steps / dt, "unit": "evals/s", "trial_sources_per_step": batch, "steps": steps,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "achieved_is": "hbm_side_estimate_gbs (committed counters of these kernel sources / kernel time of this run, less the "
                                       "Infinity Cache's possible share)" if hbm_side else
                                       "no_reuse_model_gbs capped at the block's ceilings (no committed counters of these kernel sources)",
                        "achieved_counter_gbs": cnt, "traffic": traffic, "hbm_side_estimate_gbs": hbm_side,
                        "no_reuse_model_gbs": model, "no_reuse_model_bytes_per_step": model_bytes,
                        "pure_read_ceiling_gbs": ceiling, "pure_read_ceiling_is": "kiwi_hip_measure_read_bandwidth, 4 GiB, 10 passes, this run",
                        "frac_of_pure_read_ceiling": achieved / ceiling if ceiling else None,
                        "committed_pure_read_ceiling": read_ceiling(),
                        "accumulate_ms_per_step": acc_ms,
                        "valu": valu_roofline(flops_eval * batch / t / 1e12 if t > 0 else 0.0, ar),
                        "notes": notes, "profile_note": note}}
    p.close()
    return out


def read_ceiling():
    """what a pure 16 B / lane read reaches on an MI355X of this pool (profiles/microbench/hbm_read.hip, collected with the profiles)"""
    try:
        import glob
        fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_read.json")))
        return json.load(open(fs[-1])) if fs else None
    except Exception:
        return None


def committed_counters(workload, arith):
    """counters of `workload` from the committed rocprofv3 passes (profiles/r*_summary.json), attached only when they were collected
    on exactly these kernel sources -> (entry or {}, note)"""
    prof, note = {}, "no committed profile of this workload"
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json"))):
            js = json.load(open(f))
            w = js.get("workloads", {}).get(workload + ("@fused" if arith == "fused" else ""))
            if w and w.get("batch") and os.environ.get("KIWI_HIP_ACCUM") != "direct":
                if js.get("kernel_sources_sha256") == kernel_sources_sha256():
                    prof = dict(w, file=os.path.basename(f), profile_head=js.get("head"))
                    note = "collected on commit %s, same kernel sources as this build" % js.get("head")
                elif not prof:
                    note = "%s was collected on other kernel sources (commit %s): counters not attached" % (os.path.basename(f), js.get("head"))
    except Exception:
        prof = {}
    return prof, note


def also_cfg3_100pt(p_main, device, L, batch=512, steps=6):
    """The north star's "100 sub-faults" read literally (100 sub-fault points x 2 time steps = 200 centroids), timed the
    same way right after the main workload: evals/s with inputs resident, its own roofline block, both arithmetic contracts."""
    from kiwi_amd import synthetic
    p_main.close()
    wl = synthetic.workload("cfg3-100pt", batch, 0)
    p, gf, recv, refs, tapers, ncent = setup_product(device, wl, L)
    for _ in range(2):
        p.eval()
    dt, acc_ms = timed_evals(p, steps)
    npts, nrec, ng, W = wl["npoints"], wl["nrec"], gf["data"].shape[2], L
    flops_eval = required_flops(ncent, npts, nrec, ng, W)
    ar = p.arithmetic()
    ach = flops_eval * batch / (acc_ms * 1e-3) / 1e12 if acc_ms > 0 else 0.0
    out = {"workload": "cfg3-100pt: %.0f centroids (%.0f sub-fault points) x %d receivers x 3 comp x %d samples" % (ncent, npts, nrec, W),
           "arithmetic": ar, "value": batch * steps / dt, "unit": "evals/s", "trial_sources_per_step": batch, "steps": steps,
           "roofline": valu_roofline(ach, ar, accumulate_ms_per_step=acc_ms, flops_per_eval=flops_eval,
                                     kernel="accumulate_multi_kernel<10,FUSE,4|2> (+ accumulate_grouped_kernel for the pairs it leaves)"),
           "roofline_frac": ach / valu_peak(ar),
           "other_contract": other_contract(p, batch, flops_eval, steps)}
    p.close()
    return out


def sweep(args, torch, dist, rank, local_rank, ngpus, force_dist):
    """Strong scaling: a fixed trial list (cfg5: the 10^5-point grid of BASELINE.json configs[4]) split over the ranks in
    list order; every rank evaluates its shard through ONE call (kiwi_hip_misfits_for_params: host discretiser of a piece
    under the device's evaluation of another), one all-gather of the global misfits at the end.  Timed: everything from the
    parameter list to the gathered misfits."""
    from kiwi_amd.shard import shard_range, gather_misfits
    from kiwi_amd import synthetic
    N = args.sweep
    lo, hi = shard_range(N, ngpus, rank)
    setup_n = min(512, hi - lo)
    wl = synthetic.workload(args.workload, setup_n, lo)
    p, gf, recv, refs, tapers, ncent = setup_product(local_rank, wl, args.samples)
    trials = synthetic.workload(args.workload, hi - lo, lo)["trials"]
    counts = [shard_range(N, ngpus, r)[1] - shard_range(N, ngpus, r)[0] for r in range(ngpus)]
    p.misfits_for_params(wl["sourcetype"], trials[:setup_n])            # warm-up: plans, buffers, clocks
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m, n, g, st = p.misfits_for_params(wl["sourcetype"], trials)
    allg = gather_misfits(g, dist, local_rank, counts, force=force_dist)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        allg = np.asarray(allg)
        best = int(np.nanargmin(np.where(np.isfinite(allg), allg, np.inf)))
        # where the list holds the source the references were made from, that is where the minimum has to be
        full = synthetic.workload(args.workload, N, 0)
        same = np.where(np.all(full["trials"] == full["true"][None, :], axis=1))[0]
        out = {"metric": "trial-source misfit evals/s", "value": N / elapsed, "unit": "evals/s", "n_gpus": ngpus,
               "rccl_world_size": dist.get_world_size() if dist is not None else None, "steps": 1, "warmup": 1,
               "ms_per_step": elapsed * 1e3, "wall_s": elapsed, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s: sweep of %d trial sources (%s, %s%s), %d receivers x 3 comp x %d samples" %
                                      (wl["name"], N, wl["sourcetype"], wl["method"], " + frequency filter" if wl["filter"] is not None else "",
                                       wl["nrec"], args.samples),
                          "parallelism": "trial list cut into %d contiguous shards, one all-gather of global misfits" % ngpus,
                          "timed": "parameter list -> host discretiser -> upload -> kernels -> download -> gather (inputs NOT resident)"},
               "argmin": best, "argmin_misfit": float(allg[best]), "true_source_index": int(same[0]) if len(same) else None,
               "argmin_is_true_source": bool(len(same) and best == int(same[0])),
               "failed_sources": int(np.count_nonzero(st)), "roofline": None, "cpu_baseline": None}
        print(json.dumps(_finite(out)))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: WORLD_SIZE when started by a launcher, else 1)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "cfg4-nukl", "cfg5", "cfg5-td", "cfg3-100pt", "cfg3-scatter", "cfg3-bigdb", "cfg3-bigdb4", "cfg3-bigdb4-ordered", "cfg3-w256", "cfg3-w600", "cfg3-ng8", "cfg3-static"],
                    help="BASELINE.json configs[1..4]; cfg3 (default) is the one the metric is quoted on")
    ap.add_argument("--batch", type=int, default=0,
                    help="trial sources per GPU per step (default: 12960 cfg2, 4096 cfg3, 1024 cfg3-scatter / cfg3-bigdb, 512 cfg3-100pt / cfg5 / cfg5-td, 128 cfg4)")
    ap.add_argument("--sweep", type=int, default=0,
                    help="strong scaling: ONE pass over a fixed trial list of this many sources (cfg5: its 10^5-point grid), split over "
                         "the ranks, host discretiser and transfers included; reports wall seconds and evals/s")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary figure (cfg3-100pt) of the default run")
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--piece", type=int, default=0, help="host_inclusive: sources per piece of the one-call evaluation (0: the library's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It starts N ranks (one per
        # GPU) under torch.distributed.run BEFORE anything here has touched the GPU (the devices are counted from sysfs), relays
        # their output -- rank 0 prints the JSON line -- and exits with their status.
        sys.exit(launch_ranks(args.gpus))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # KIWI_BENCH_FORCE_DIST=1: go through RCCL even with one rank (self-test of the collective path on a 1-GPU box)
    force_dist = bool(os.environ.get("KIWI_BENCH_FORCE_DIST"))
    # KIWI_BENCH_BACKEND=gloo + KIWI_BENCH_DEVICE=d: the same N-rank run with every rank's engine on device d and the collective
    # through gloo -- the dress rehearsal of `--gpus N` on a box with one GPU (tests/test_gpu_fullsize.py); RCCL wants a device per rank
    backend = os.environ.get("KIWI_BENCH_BACKEND", "nccl")
    if os.environ.get("KIWI_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["KIWI_BENCH_DEVICE"])
    if world > 1 or force_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # First contact with a multi-GPU node is unattended: no collective may wait for a rank that is gone or stuck longer than
        # this many seconds (default 120; the whole default run is shorter).  RCCL: the process group's watchdog aborts the process
        # when a collective exceeds it; gloo: the collective raises.  Either way the rank exits non-zero, torch.distributed.run
        # (--max-restarts 0) ends the others, and the run's status is non-zero -- nothing is retried or restarted in place.
        ctimeout = datetime.timedelta(seconds=float(os.environ.get("KIWI_BENCH_COLLECTIVE_TIMEOUT", "120")))
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=ctimeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=ctimeout)
    ngpus = world
    if ngpus > 1 and not os.environ.get("KIWI_HIP_DISC_THREADS"):
        # N ranks share the box's host cores: each rank's discretiser team gets its share instead of all of them
        from kiwi_amd import lib as _lib
        os.environ["KIWI_HIP_DISC_THREADS"] = str(max(1, int(_lib.load().kiwi_hip_effective_cpus()) // ngpus))

    from kiwi_amd.shard import shard_range, gather_misfits, DeviceGather
    from kiwi_amd import synthetic
    if args.batch <= 0:
        # (cfg3: 4096 sources per step -- 0.14 s -- so that the driver's 20 steps time 2.8 s of device work)
        args.batch = {"cfg2": 12960, "cfg3": 4096, "cfg3-scatter": 1024, "cfg3-bigdb": 1024, "cfg3-100pt": 512, "cfg4": 128, "cfg4-nukl": 128, "cfg5": 512,
                      "cfg5-td": 512, "cfg3-w256": 16384, "cfg3-w600": 8192, "cfg3-bigdb4": 1024, "cfg3-bigdb4-ordered": 1024, "cfg3-ng8": 4096, "cfg3-static": 4096}[args.workload]
    if args.sweep > 0:
        return sweep(args, torch, dist, rank, local_rank, ngpus, force_dist)
    lo, hi = shard_range(args.batch * ngpus, ngpus, rank)
    wl = synthetic.workload(args.workload, hi - lo, lo)
    p, gf, recv, refs, tapers, ncent = setup_product(local_rank, wl, args.samples)
    nrec = wl["nrec"]
    nmis = p.nmisfits()

    counts = [shard_range(args.batch * ngpus, ngpus, r)[1] - shard_range(args.batch * ngpus, ngpus, r)[0] for r in range(ngpus)]

    # the one collective of the sharded search: over RCCL straight from the engine's device buffer (DeviceGather), through the
    # host only with a CPU backend (gloo rehearsal) or a single process
    dgather = DeviceGather(dist, local_rank, counts) if dist is not None and backend == "nccl" else None

    def step():
        p.eval()
        if dgather is not None:
            return dgather.gather(p)                 # (synchronises the engine's stream; the result stays on the device)
        p.sync()
        _, _, g = p.get_misfits()
        return gather_misfits(g, dist, local_rank, counts, force=force_dist)

    if dist is not None:                             # RCCL sets itself up lazily at the first collective: not in the timed region
        dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        allg = step()
    # test hooks (tests/test_gpu_fullsize.py, tests/test_shard_gloo.py): a rank that DIES mid-run and a rank that HANGS mid-run -- the
    # launcher has to come back with a non-zero status within the collective timeout, not sit in a collective until someone kills it
    if dist is not None and os.environ.get("KIWI_BENCH_FAIL_RANK") == str(rank):
        print("bench.py: rank %d leaves the run (KIWI_BENCH_FAIL_RANK)" % rank, file=sys.stderr, flush=True)
        os._exit(17)
    if dist is not None and os.environ.get("KIWI_BENCH_HANG_RANK") == str(rank):
        print("bench.py: rank %d stops answering (KIWI_BENCH_HANG_RANK)" % rank, file=sys.stderr, flush=True)
        time.sleep(3600)
    p.kernel_ms()                                   # reset the HIP-event accumulators
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_s = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        allg = step()
        step_s.append(time.perf_counter() - ts)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if os.environ.get("KIWI_BENCH_VERBOSE"):
        print("rank %d step ms: %s" % (rank, " ".join("%.2f" % (1e3 * v) for v in step_s)), file=sys.stderr)
    per_rank_ms = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                   # every rank's own time: a slow device or a starved discretiser team shows up by rank
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if dgather is not None:
        allg = dgather.host()
    ms, launches = p.kernel_ms()

    if rank == 0:
        total_evals = args.batch * ngpus * args.steps
        value = total_evals / elapsed
        n_ip, ng, L, W = 4, gf["data"].shape[2], args.samples, (wl.get("window") or args.samples)
        assert ng == wl.get("ng", 10)
        npts = wl["npoints"]
        acc_s = float(ms[1]) * 1e-3
        launches_acc = max(int(launches[1]), 1)
        flops_eval = required_flops(ncent, npts, nrec, ng, W)
        arith = p.arithmetic()
        peak_tf = valu_peak(arith)
        achieved_tflops = flops_eval * args.batch * args.steps / acc_s / 1e12 if acc_s > 0 else 0.0
        # ---- the SURVEY 8d byte model (no reuse at all: every centroid re-reads its n_g x n_ip rows over the window)
        b_eval = int(ncent * nrec * ng * n_ip * W * 4 + nrec * 3 * W * 4 * 2)
        no_reuse_gbs = b_eval * args.batch * args.steps / acc_s / 1e9 if acc_s > 0 else 0.0
        # ---- measured counters of this same command from the committed rocprofv3 passes (profiles/r*_summary.json):
        # PMC counters cannot be collected from inside this process, so these are null when no profile matches
        prof, prof_note = committed_counters(args.workload, arith)
        scale = args.batch / prof["batch"] if prof else 0.0
        traffic = prof["hbm_bytes_per_launch"] * scale if prof.get("hbm_bytes_per_launch") else None
        avg_ms = float(ms[1]) / launches_acc
        try:
            copy_gbs = measured_copy_bandwidth(torch, torch.device("cuda", local_rank))
        except Exception:
            copy_gbs = None
        kernel = "accumulate_kernel (KIWI_HIP_ACCUM=direct)" if os.environ.get("KIWI_HIP_ACCUM") == "direct" else \
            ("accumulate_cellw_kernel<NG,FUSE,COMPACT> (cell runs, a tile per wave; + accumulate_grouped_kernel for the pairs it leaves)" if npts > 0.5 * ncent
             else ("accumulate_grouped_kernel<10,256> (runs of sources sharing their tiles)" if wl["sourcetype"] == "moment_tensor"
                   else "accumulate_multi_kernel<10,FUSE,4|2> (four / two trial sources per workgroup; + accumulate_grouped_kernel for the pairs it leaves)"))
        out = {
            "metric": "trial-source misfit evals/s", "value": value, "unit": "evals/s",
            "n_gpus": ngpus, "rccl_world_size": dist.get_world_size() if dist is not None else None,
            "collective": None if dist is None else ("all_gather_into_tensor over %s, device to device from the engine's buffer" % backend
                                                      if dgather is not None else "all-gather over %s through the host" % backend),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            # N > 1: what an operator needs when the figure is off -- every rank's own step time (value uses the slowest), the host
            # threads each rank's discretiser team may use (the CPU quota divided by the ranks) and the collective's timeout
            "ms_per_step_by_rank": per_rank_ms,
            "host_threads_per_rank": int(os.environ.get("KIWI_HIP_DISC_THREADS", "0")) or int(p.L.kiwi_hip_effective_cpus()),
            "host_cpus_effective": int(p.L.kiwi_hip_effective_cpus()),
            "collective_timeout_s": float(os.environ.get("KIWI_BENCH_COLLECTIVE_TIMEOUT", "120")) if dist is not None else None,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # arithmetic contract of the accumulate kernels this line was measured under (include/kiwi_hip.h KIWI_ARITH_*;
            # `other_contract` below carries the same batch under the other one)
            "arithmetic": arith,
            "config": {"workload": "%s: %s source, %.0f centroids (%.0f sub-fault points x %.0f time steps%s) x %d receivers x 3 comp x %d samples, "
                                   "ng=%d%s, bilinear GF interpolation, %s%s, tapered %d-sample window%s"
                                   % (wl["name"], wl["sourcetype"], ncent, npts, ncent / max(npts, 1.0),
                                      ", every trial at ONE location (strike sweep, 0.1 degree steps)" if args.workload == "cfg3" else "",
                                      nrec, L, ng, " (static end values, interior gaps)" if wl.get("variant") == "static" else "", wl["method"],
                                      " + frequency filter" if wl["filter"] is not None else "", W,
                                      "; `also` = the literal 100-sub-fault-point reading (200 centroids), `also_scatter` = the same source "
                                      "over a shuffled location grid" if args.workload == "cfg3" else ""),
                       "trial_sources_per_gpu_per_step": args.batch, "misfits_per_source": nmis,
                       "parallelism": "trial-source shard x%d, all-gather of global misfits" % ngpus},
            # The dominant kernel runs against the vector ALU's issue rate, not against HBM: the Green's function tensor
            # (130-160 MB) is resident in L2 / Infinity Cache and HBM is nearly idle, by design (see `hbm` below).  `achieved`
            # = required flops / measured kernel time, `peak` = the unfused fp32 vector rate, so frac <= 1 by construction.
            "roofline": {"bound": "valu_issue", "achieved": achieved_tflops, "peak": peak_tf, "unit": "TFLOP/s",
                         "frac": achieved_tflops / peak_tf, "arithmetic": arith,
                         # `peak` is the builder's ceiling under the exact contract (bit parity forbids the fused multiply-add: half
                         # the chip's rate); the guide's figure and the fraction of it are carried next to it
                         "peak_is": "fp32 vector rate WITHOUT fused multiply-add (exact contract)" if arith != "fused" else "fp32 vector rate with fused multiply-add",
                         "peak_fma": FP32_PEAK_TFLOPS, "frac_of_fma_peak": achieved_tflops / FP32_PEAK_TFLOPS,
                         "traffic": traffic,
                         "kernel": kernel, "launches": int(launches[1]), "avg_launch_ms": avg_ms,
                         "flops_per_eval": flops_eval,
                         "issue_slots": {"valu_busy_frac": prof.get("valu_issue_frac"), "lds_busy_frac": prof.get("lds_busy_frac"),
                                         "valu_insts_per_launch": prof["valu_insts_per_launch"] * scale if prof.get("valu_insts_per_launch") else None,
                                         "source": prof.get("file")},
                         "profile_head": prof.get("profile_head"), "profile_note": prof_note,
                         "hbm": {"actual_gbs": traffic / (avg_ms * 1e-3) / 1e9 if traffic else None, "peak_gbs": HBM_PEAK_GBS,
                                 "frac": traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
                                 "peak_measured_copy_gbs": copy_gbs,
                                 "l2_request_gbs": prof["l2_request_bytes_per_launch"] * scale / (avg_ms * 1e-3) / 1e9
                                 if prof.get("l2_request_bytes_per_launch") else None,
                                 "algorithmic_no_reuse_gbs": no_reuse_gbs, "algorithmic_bytes_per_eval": b_eval,
                                 "reuse_factor": b_eval * args.batch / traffic if traffic else None,
                                 "note": "algorithmic_no_reuse_gbs is the SURVEY 8d byte model (every centroid re-reads its 40 rows) "
                                         "over the kernel time; it exceeds the HBM peak by reuse_factor because the rows are served "
                                         "from cache and every blended tile is shared by the time steps of a sub-fault"},
                         # second resource the kernel saturates in its build phase (measured r03: build alone = 31 TB/s of L2 -> CU reads)
                         "l2": {"request_gbs": prof["l2_request_bytes_per_launch"] * scale / (avg_ms * 1e-3) / 1e9
                                if prof.get("l2_request_bytes_per_launch") else None, "peak_gbs": 34500.0},
                         "other_kernels_ms_per_step": {"geometry": float(ms[0]) / args.steps,
                                                       "misfit": float(ms[2]) / args.steps}},
        }
        if os.environ.get("KIWI_BENCH_DUMP_MISFITS"):        # (tests: the gathered global misfits of the last step, in trial order)
            out["gathered_global_misfits"] = [float(x) for x in np.asarray(allg).ravel()]
        def secondary(fn, *a, **k):
            """a secondary block must never take the driver's line down: what goes wrong in one is reported in its place"""
            try:
                return fn(*a, **k)
            except Exception as ex:              # noqa: BLE001
                return {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}

        if ngpus == 1 and not args.no_cpu_baseline:
            gm, gn, gg = p.get_misfits()
            out["cpu_baseline"] = secondary(cpu_baseline, wl, gf, recv, refs, tapers, np.asarray(allg), gm, gn)
        else:
            out["cpu_baseline"] = None
        if ngpus == 1 and not args.no_also:
            out["other_contract"] = secondary(other_contract, p, args.batch, flops_eval)
            # the whole path (discretiser + transfers included) for the same trial sources, against the resident-input figure
            longer = synthetic.workload(args.workload, 4 * args.batch, 0)["trials"] if wl["crust"] is not None else None
            out["host_inclusive"] = secondary(host_inclusive, p, wl, value, piece=args.piece, longer=longer)
        if ngpus == 1 and args.workload == "cfg3" and not args.no_also:
            out["also"] = secondary(also_cfg3_100pt, p, local_rank, args.samples)
            # the other BASELINE.json configurations and the unfriendly reading of cfg3, driver-timed: three steps each, own roofline
            # block, a short CPU baseline on the same trial sources
            out["also_cfg2"] = secondary(also_workload, local_rank, args.samples, "cfg2", 12960)
            out["also_cfg4"] = secondary(also_workload, local_rank, args.samples, "cfg4", 128, cpu_budget_s=4.0)
            out["also_scatter"] = secondary(also_workload, local_rank, args.samples, "cfg3-scatter", 1024)
            out["also_ng8"] = secondary(also_workload, local_rank, args.samples, "cfg3-ng8", 4096)          # far-field database (8 components)
            # BASELINE config 4's source type swept over what it is inverted for -- nucleation point, rupture velocity: a fast-marching
            # solve per trial on the host --: resident rate and the host-inclusive rate of a four-piece list
            # (host_pieces = 1350: the WHOLE 25 x 9 x 6 grid of the sweep as one list)
            out["also_nukl"] = secondary(also_workload, local_rank, args.samples, "cfg4-nukl", 128, cpu_budget_s=0.0, host_pieces=1350)
            out["sweep"] = secondary(sweep_block, local_rank, args.samples)
            out["also_hbm"] = secondary(also_bigdb4, local_rank, args.samples)
        print(json.dumps(_finite(out)))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
