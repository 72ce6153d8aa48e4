#!/usr/bin/env python3
"""Condenses the rocprofv3 output of profiles/collect.sh into profiles/<tag>_summary.json (+ the kernel-stats CSVs).
Usage (in the repo, after the gpurun call has merged its output): python profiles/summarize.py r04 gpurun_out/r04
Workload keys: `<workload>` (exact arithmetic contract) and `<workload>@fused`.
Records the commit and a hash of the kernel sources the passes were taken on: bench.py attaches these counters to its line
only when its own build comes from the same sources (`roofline.profile_head`).

Per workload the summary holds what bench.py's `roofline` block quotes (keys under "workloads"):
  ("per launch" = per bench step: a step over a large batch is several dispatches, launches_per_step)
  hbm_bytes_per_launch        FETCH_SIZE x 2 + WRITE_SIZE, summed over the accumulate kernels of one step.  The x 2: gfx950 tallies
                              coalesced reads at half their bytes (MI355X_MICROARCH.md "HBM") -- checked on this pool with a pure
                              read of KNOWN size (profiles/microbench/hbm_read.hip, 4 294 963 200 bytes per pass: FETCH_SIZE
                              2 097 000 KiB for 16-byte AND for 4-byte loads per lane, TCC_MISS_sum x 128 B and TCC_EA0_RDREQ_sum x
                              128 B = the bytes read; profiles/README.md round 5).
                              These are the L2's memory-side requests: Infinity-Cache hits are included, so this is an
                              upper bound of what reached HBM.
  l2_request_bytes_per_launch (TCC_HIT_sum + TCC_MISS_sum) x 128 B
  valu_insts_per_launch       SQ_INSTS_VALU (wave instructions)
  valu_issue_frac             SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8
  lds_busy_frac               SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles)
  accumulate_ms               median duration per step of the accumulate kernels (kernel trace)
"""
import csv
import glob
import json
import os
import shutil
import sys

tag, src = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import subprocess  # noqa: E402
import bench  # noqa: E402
head = subprocess.run(["git", "-C", os.path.dirname(here), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
dirty = bool(subprocess.run(["git", "-C", os.path.dirname(here), "status", "--porcelain", "kiwi_amd/csrc"], capture_output=True, text=True).stdout.strip())
import hashlib  # noqa: E402


def raw_hashes(workload_key):
    """SHA-256 of every raw rocprofv3 file (and bench line) a workload's entry was condensed from: the summary cannot be edited
    without these ceasing to match gpurun_out/<tag>/ -- and nothing in it needs editing by hand: the source hash below is computed
    here, at collection time, from the code alone (bench.kernel_sources_sha256 strips comments)."""
    hs = {}
    for d in ("kt", "fetch", "write", "l2", "sq", "sq2"):
        for f in sorted(glob.glob(os.path.join(src, d + "_" + workload_key, "**", "*.csv"), recursive=True)):
            if f.endswith("_kernel_trace.csv"):
                continue
            hs[os.path.relpath(f, src)] = hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
    b = os.path.join(src, "bench_%s.json" % workload_key)
    if os.path.exists(b):
        hs[os.path.relpath(b, src)] = hashlib.sha256(open(b, "rb").read()).hexdigest()[:16]
    return hs


out = {"tag": tag, "command": "rocprofv3 <pass> -- python3 bench.py --workload <w> --steps 5 --warmup 2 --no-cpu-baseline --no-also",
       "head": (head + ("+uncommitted kernel changes" if dirty else "")) if head else None, "kernel_sources_sha256": bench.kernel_sources_sha256(),
       "kernel_sources_sha256_is": "SHA-256 over the comment-stripped text of " + ", ".join(bench.KERNEL_SOURCES) + " (bench.kernel_sources_sha256), "
                                   "computed by this script when it condensed the raw files",
       "note": "per step = sum over the accumulate kernels of one bench step of the per-dispatch medians; FETCH_SIZE / WRITE_SIZE "
               "are in KiB as rocprofv3 reports them", "workloads": {}}
# a re-collection of some workloads keeps the entries of the others
_prev = os.path.join(here, tag + "_summary.json")
if os.path.exists(_prev):
    try:
        _p = json.load(open(_prev))
        if _p.get("kernel_sources_sha256") == out["kernel_sources_sha256"]:
            out["workloads"].update(_p.get("workloads", {}))
    except ValueError:
        pass


def counters(d):
    """{kernel name: {counter: median over dispatches}}"""
    fs = glob.glob(os.path.join(src, d, "**", "*_counter_collection.csv"), recursive=True)
    vals = {}
    if not fs:
        return {}
    for r in csv.DictReader(open(fs[0])):
        if "accumulate" in r["Kernel_Name"]:
            vals.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    res = {}
    for k, cs in vals.items():
        res[k] = {"_dispatches": max(len(v) for v in cs.values())}
        for c, v in cs.items():
            v.sort()
            res[k][c] = v[len(v) // 2]
    return res


def total(cnt, name):
    """sum over the accumulate kernels of a TIMED step: the instantiations dispatched every step (the set-up evaluation that
    makes the reference traces runs the unfused instantiation once and is left out; the complement launch of the cell mode
    is dispatched every step and nearly empty)"""
    if not cnt:
        return 0.0
    most = max(c["_dispatches"] for c in cnt.values())
    return sum(c.get(name, 0.0) for c in cnt.values() if c["_dispatches"] >= most - 1 and c["_dispatches"] > 2)


for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    w = os.path.basename(f)[len("bench_"):-len(".json")]
    try:
        line = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception:
        continue
    # a step over a large batch is several launches (the engine bounds its workspace): counters and durations are per-dispatch
    # medians, a step is `lps` of them
    lps = max(1, int(round(line["roofline"]["launches"] / float(line["steps"])))) if line.get("roofline") else 1
    e = {"batch": line["config"]["trial_sources_per_gpu_per_step"], "launches_per_step": lps, "arithmetic": line.get("arithmetic", "exact"),
         "bench_line_under_rocprof": line}
    c_f, c_w, c_l2, c_sq, c_sq2 = (counters(p + "_" + w) for p in ("fetch", "write", "l2", "sq", "sq2"))
    if c_f and c_w:
        rd = total(c_f, "FETCH_SIZE") * 1024 * 2 * lps
        wr = total(c_w, "WRITE_SIZE") * 1024 * lps
        e["hbm_bytes_per_launch"] = rd + wr
        e["fabric_read_corrected"] = rd
        e["fabric_write"] = wr
    if c_l2:
        hit, miss = total(c_l2, "TCC_HIT_sum"), total(c_l2, "TCC_MISS_sum")
        e["l2_request_bytes_per_launch"] = (hit + miss) * 128 * lps
        e["l2_hit_rate"] = hit / (hit + miss) if hit + miss else None
    if c_sq2:
        cyc = total(c_sq2, "GRBM_GUI_ACTIVE") / 8.0 * lps
        e["kernel_cycles"] = cyc
        e["valu_insts_per_launch"] = total(c_sq2, "SQ_INSTS_VALU") * lps
        if cyc:
            e["valu_issue_frac"] = e["valu_insts_per_launch"] * 4.0 / (1024.0 * cyc)
            e["lds_busy_frac"] = total(c_sq2, "SQ_LDS_IDX_ACTIVE") * lps / (256.0 * cyc)
            e["scalar_issue_frac"] = total(c_sq2, "SQ_ACTIVE_INST_SCA") * lps / (1024.0 * cyc)
        e["lds_bank_conflict_cycles"] = total(c_sq2, "SQ_LDS_BANK_CONFLICT") * lps
    e["counters"] = {"sq": c_sq, "sq2": c_sq2, "l2": c_l2, "fetch": c_f, "write": c_w}
    e["raw_sha256_16"] = raw_hashes(w)
    # kernel stats of the trace pass
    fs = glob.glob(os.path.join(src, "kt_" + w, "**", "*_kernel_stats.csv"), recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(here, "%s_kernel_stats_%s.csv" % (tag, w)))
        rows = list(csv.DictReader(open(fs[0])))
        e["kernel_stats"] = [{"name": r["Name"][:70], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                              "pct": float(r["Percentage"])} for r in rows[:8]]
        acc = [r for r in rows if "accumulate" in r["Name"]]
        if acc:                              # the timed steps' kernels (the set-up launch uses the unfused instantiation once)
            most = max(int(r["Calls"]) for r in acc)
            e["accumulate_ms"] = lps * sum(float(r["AverageNs"]) / 1e6 for r in acc if int(r["Calls"]) >= most - 1 and int(r["Calls"]) > 2)
    out["workloads"][w] = e
for f in sorted(glob.glob(os.path.join(src, "hbm_read_*.json"))):          # pure-read microbench collected with the profiles (collect.sh)
    try:
        js = json.loads(open(f).read().strip().splitlines()[-1])
        out.setdefault("hbm_read_microbench", {})[os.path.basename(f)] = js
        if f.endswith("hbm_read_4g.json"):
            json.dump(js, open(os.path.join(here, tag + "_hbm_read.json"), "w"))
    except (ValueError, IndexError):
        pass
# ... and what the counters say for its KNOWN number of bytes (4 GiB buffer, every pass reads all of it once)
mb = {}
for d in sorted(glob.glob(os.path.join(src, "hbmread_pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        vals = {}
        for r in csv.DictReader(open(f)):
            vals.setdefault((r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in vals.items():
            v.sort()
            mb.setdefault(k, {})[c] = v[len(v) // 2]
if mb:
    out.setdefault("hbm_read_microbench", {})["counters_per_pass_of_4294963200_bytes"] = mb
json.dump(out, open(os.path.join(here, tag + "_summary.json"), "w"), indent=1)
for w, e in out["workloads"].items():
    print(w, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if k not in ("counters", "bench_line_under_rocprof", "kernel_stats")})
