#!/usr/bin/env python3
"""Prints the markdown table of profiles/<tag>_summary.json (what profiles/README.md and DESIGN.md quote).   python profiles/table.py r05"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
here = os.path.dirname(os.path.abspath(__file__))
d = json.load(open(os.path.join(here, tag + "_summary.json")))
order = ["cfg3", "cfg3-100pt", "cfg3-scatter", "cfg3-ng8", "cfg3-static", "cfg3-w256", "cfg3-w600", "cfg3-bigdb", "cfg3-bigdb4", "cfg3-bigdb4-ordered", "cfg2", "cfg4", "cfg4-nukl", "cfg5", "cfg5-td"]
print("| workload | contract | sources / step | evals/s | accumulate ms / step | `frac` | VALU issue | LDS busy | L2 hit | memory-side GB / step (TB/s) | L2 requests GB / step | VALU instructions / step |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for w in order:
    for suf in ("", "@fused"):
        e = d["workloads"].get(w + suf)
        if not e:
            continue
        line = e["bench_line_under_rocprof"]
        r = line["roofline"]
        acc = e.get("accumulate_ms") or r["avg_launch_ms"] * r["launches"] / line["steps"]
        mem = e.get("hbm_bytes_per_launch")
        print("| %s | %s | %s | %s | %.1f | %.2f | %.1f %% | %.1f %% | %.1f %% | %s | %.0f | %.2e |" % (
            w, e["arithmetic"], "{:,}".format(e["batch"]).replace(",", " "), "{:,.0f}".format(line["value"]).replace(",", " "), acc, r["frac"],
            100 * e.get("valu_issue_frac", 0), 100 * e.get("lds_busy_frac", 0), 100 * (e.get("l2_hit_rate") or 0),
            "%.1f (%.2f)" % (mem / 1e9, mem / 1e9 / acc) if mem else "-", (e.get("l2_request_bytes_per_launch") or 0) / 1e9, e.get("valu_insts_per_launch", 0)))
