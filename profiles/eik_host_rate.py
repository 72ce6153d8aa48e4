"""Host rate of the eikonal discretiser (no GPU needed): ms per discretisation of a cfg4-nukl trial (another fast-marching
solve per trial; the solve cache is switched off) on one thread and on `--threads` threads, plain against optimised march.
VERDICT r05 item 1(c): solves per second and core, and the cores one GPU's device rate needs.

    python profiles/eik_host_rate.py [--threads 16] [--n 64] [--device-rate 557]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(plain, threads, n):
    os.environ["KIWI_HIP_EIK_CACHE"] = "0"
    os.environ["KIWI_HIP_EIK_PLAIN"] = "1" if plain else "0"
    import numpy as np
    from kiwi_amd import lib as klib, synthetic as syn
    w = syn.workload("cfg4-nukl", n)
    cp, cn = w["constraints"]
    trials = w["trials"]

    L = klib.load()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    prof = np.ascontiguousarray(w["crust"], np.float32)
    pts, nrm = np.ascontiguousarray(cp, np.float32), np.ascontiguousarray(cn, np.float32)

    def one(p):                                              # ONE library call per trial (engine.discretize_eikonal makes two)
        p = np.ascontiguousarray(p, np.float32)
        cent = np.empty((4096, 10), np.float32)
        n, mo, ri = C.c_int(), C.c_float(), C.c_float()
        rc = L.kiwi_hip_discretize_eikonal(5, fp(p), len(p), 0.5, fp(prof), len(pts), fp(pts), fp(nrm), fp(cent), 4096,
                                           C.byref(n), C.byref(mo), C.byref(ri))
        assert rc == 0, rc
        return n.value

    one(trials[0])
    t0 = time.perf_counter()
    nc = [one(p) for p in trials[:max(4, n // 8)]]
    t1 = (time.perf_counter() - t0) / len(nc)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(one, trials[:threads]))                 # warm the per-thread work arrays
        t0 = time.perf_counter()
        list(ex.map(one, trials))
        tn = time.perf_counter() - t0
    print(json.dumps(dict(plain=plain, ms_per_discretisation_1thread=1e3 * t1, threads=threads, n=n,
                          solves_per_s=n / tn, centroids=int(np.mean(nc)))))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--device-rate", type=float, default=557.0, help="evals/s of one GPU with the sources resident (cfg4-nukl)")
    ap.add_argument("--child", type=int, default=-1)
    a = ap.parse_args()
    if a.child >= 0:
        child(a.child, a.threads, a.n)
        sys.exit(0)
    from kiwi_amd import lib as klib
    threads = a.threads or klib.load().kiwi_hip_effective_cpus()
    out = {}
    for plain in (1, 0):
        r = subprocess.run([sys.executable, __file__, "--child", str(plain), "--threads", str(threads), "--n", str(a.n)],
                           capture_output=True, text=True, check=True)
        out["plain" if plain else "optimised"] = json.loads(r.stdout.strip().splitlines()[-1])
    o = out["optimised"]
    per_core = 1e3 / o["ms_per_discretisation_1thread"]
    out["summary"] = dict(solves_per_s_per_core=per_core, cores_for_device_rate=a.device_rate / per_core,
                          device_rate=a.device_rate, speedup_1thread=out["plain"]["ms_per_discretisation_1thread"] / o["ms_per_discretisation_1thread"],
                          speedup_threads=o["solves_per_s"] / out["plain"]["solves_per_s"])
    print(json.dumps(out, indent=1))
