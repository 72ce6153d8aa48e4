"""Rewrites the per-workload tables of DESIGN.md (section 3), README.md and profiles/README.md from profiles/r03_summary.json
(after `python profiles/summarize_r03.py r03 gpurun_out/r03`):  python profiles/refresh_tables_r03.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = json.load(open(os.path.join(ROOT, "profiles", "r03_summary.json")))
W = S["workloads"]
ORDER = ["cfg3", "cfg3-100pt", "cfg3-scatter", "cfg3-bigdb", "cfg4", "cfg2", "cfg5", "cfg5-td"]


def g(k):
    v = W[k]
    b = v["bench_line_under_rocprof"]
    return v, b, b["roofline"]


def evs(ev):
    if ev < 1000:
        return "%d" % round(ev)
    if ev < 1e5:
        return "%.1f·10³" % (ev / 1e3)
    return "%.1f·10⁵" % (ev / 1e5)


def between(s, start, end_marker):
    i = s.index(start)
    j = s.index(end_marker, i)
    return i, j


# ---- DESIGN.md
LABEL = {"cfg3": "cfg3 (100 centroids = 20 points × 5, 50 rec)", "cfg3-100pt": "cfg3-100pt (200 centroids = 100 points × 2)",
         "cfg3-scatter": "cfg3-scatter (cfg3 source, shuffled location grid: no groups of sources)",
         "cfg3-bigdb": "cfg3-bigdb (1.04 GB tensor, every source in a cell of its own: the HBM regime)",
         "cfg4": "cfg4 (`mt_eikonal`, 468 centroids, 200 rec)", "cfg2": "cfg2 (moment-tensor grid, 12 960 sources)",
         "cfg5": "cfg5 (spectral comparator + filter)", "cfg5-td": "cfg5-td (cfg5 trials, `l2norm` on filtered traces)"}
rows = []
for k in ORDER:
    v, b, r = g(k)
    ms = r["avg_launch_ms"]
    per = " (%.1f per 1024)" % (ms / 4) if k == "cfg3" else ""
    frac = ("**%.2f**" if k in ("cfg3", "cfg3-100pt", "cfg4", "cfg2") else "%.2f") % r["frac"]
    tb = v["hbm_bytes_per_launch"] / v["accumulate_ms"] / 1e9
    rows.append("| %s | %s | %.1f%s | %s | %s | %.0f %% | %.0f %% | %.1f %% | %.1f | %.1f (%s) |" % (
        LABEL[k], "12 960" if v["batch"] == 12960 else v["batch"], ms, per, evs(b["value"]), frac, 100 * v["valu_issue_frac"],
        100 * v["lds_busy_frac"], 100 * v["l2_hit_rate"], v["l2_request_bytes_per_launch"] / v["accumulate_ms"] / 1e9,
        v["hbm_bytes_per_launch"] / 1e9, ("**%.1f**" % tb) if k == "cfg3-bigdb" else "%.2f" % tb))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
i, j = between(s, "| workload (`bench.py --workload`, **r03**", "\n\n")
hdr = ("| workload (`bench.py --workload`, **r03**, `profiles/r03_summary.json`, commit `%s`) | trial sources / launch | accumulate ms / launch | evals/s | "
       "`frac` (of 78.65 TF) | VALU issue | LDS busy | L2 hit | L2 → CU TB/s | memory-side GB / launch (TB/s) |\n|---|---|---|---|---|---|---|---|---|---|\n" % S.get("head"))
open(p, "w").write(s[:i] + hdr + "\n".join(rows) + s[j:])

# ---- README.md
WHAT = {"cfg3": "`cfg3` (default; the configuration the north-star target is quoted on) | `bilateral`, 100 centroids (20 sub-fault points x 5 time steps) x 50 receivers x 3 components x 4096 samples, bilinear GF interpolation, tapered L2",
        "cfg3-100pt": "`cfg3-100pt` | the same with 100 sub-fault POINTS (200 centroids)",
        "cfg3-scatter": "`cfg3-scatter` | cfg3 source over a shuffled location grid (no rows shared between neighbouring trials)",
        "cfg3-bigdb": "`cfg3-bigdb` | cfg3 source over a 1 GB database, every trial in a cell of its own: the rows come from HBM (7.6 TB/s, 95 % of the peak)",
        "cfg4": "`cfg4` | `mt_eikonal`, 468 centroids x 200 receivers, rise-time fold",
        "cfg2": "`cfg2` | moment-tensor grid, 12 960 point sources x 50 receivers",
        "cfg5": "`cfg5` | spectral comparator (in-LDS transform) with frequency filter",
        "cfg5-td": "`cfg5-td` | the cfg5 trials under a time-domain L2 on frequency-filtered traces (transform forward and back in LDS)"}
R2 = {"cfg3": "23 800 / 0.40", "cfg3-100pt": "7 400 / 0.33", "cfg3-scatter": "22 900 / 0.38", "cfg3-bigdb": "–", "cfg4": "465 / 0.27",
      "cfg2": "5.5·10^5 / 0.33", "cfg5": "16 100 / 0.40", "cfg5-td": "16 000 / 0.40"}
rows = []
for k in ORDER:
    v, b, r = g(k)
    ev = b["value"]
    if ev >= 1e5:
        t = "%.1f·10^5" % (ev / 1e5)
    else:
        t = format(int(round(ev, -2 if ev > 3000 else 0)), ",").replace(",", " ")
    if k == "cfg3":
        t = "**%s**" % t
    rows.append("| %s | %s | %s | %.1f | %.2f | %.0f %% | %s |" % (WHAT[k], "12 960" if v["batch"] == 12960 else v["batch"], t, r["avg_launch_ms"],
                                                                   r["frac"], 100 * v["valu_issue_frac"], R2[k]))
p = os.path.join(ROOT, "README.md")
s = open(p).read()
i, j = between(s, "| `--workload` | what |", "\n\n")
hdr = ("| `--workload` | what | trial sources / launch | evals/s | accumulate kernel, ms / launch | `roofline.frac` (required flops / 78.65 TFLOP/s "
       "unfused fp32 vector peak) | vector issue slots busy | round 2: evals/s / frac |\n|---|---|---|---|---|---|---|---|\n")
open(p, "w").write(s[:i] + hdr + "\n".join(rows) + s[j:])

# ---- profiles/README.md
rows = []
for k in ORDER:
    v, b, r = g(k)
    rows.append("| %s | %s | %.1f | %.1f %% | %.1f %% | %.1f %% | %.1f | %.0f | %.3g | %.3g |" % (
        k, v["batch"], v["accumulate_ms"], 100 * v["valu_issue_frac"], 100 * v["lds_busy_frac"], 100 * v["l2_hit_rate"],
        v["hbm_bytes_per_launch"] / 1e9, v["l2_request_bytes_per_launch"] / 1e9, v["valu_insts_per_launch"], v["lds_bank_conflict_cycles"]))
p = os.path.join(ROOT, "profiles", "README.md")
s = open(p).read()
i, j = between(s, "| workload | sources / launch | accumulate ms | VALU issue | LDS busy | L2 hit | memory-side GB / launch | L2 requests GB / launch | VALU instructions / launch | LDS bank-conflict", "\n\n")
hdr = ("| workload | sources / launch | accumulate ms | VALU issue | LDS busy | L2 hit | memory-side GB / launch | L2 requests GB / launch | "
       "VALU instructions / launch | LDS bank-conflict cycles / launch |\n|---|---|---|---|---|---|---|---|---|---|\n")
s = s[:i] + hdr + "\n".join(rows) + s[j:]
s = re.sub(r"Collected on commit `[0-9a-f]+`\.", "Collected on commit `%s`." % S.get("head"), s)
open(p, "w").write(s)
print("tables refreshed from", S.get("head"), S.get("kernel_sources_sha256", "")[:16])
