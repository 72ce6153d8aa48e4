"""Markdown rows of the per-workload table (README.md, profiles/README.md) from a summary written by
summarize_r03.py:  python profiles/table_r03.py [profiles/r03_summary.json]"""
import json
import sys

s = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r03_summary.json"))
print("head", s.get("head"), "kernel sources", s.get("kernel_sources_sha256", "")[:16])
print("| workload | trial sources / launch | accumulate ms / launch | evals/s | frac | VALU issue | LDS busy | L2 hit | L2 -> CU TB/s | memory-side GB / launch (TB/s) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k, v in s["workloads"].items():
    b = v["bench_line_under_rocprof"]
    r = b["roofline"]
    ms = r["avg_launch_ms"]
    print("| %s | %d | %.1f (rocprofv3: %.1f) | %.3g | %.3f | %.0f %% | %.0f %% | %.1f %% | %.1f | %.1f (%.2f) |" % (
        k, v["batch"], ms, v["accumulate_ms"], b["value"], r["frac"], 100 * v["valu_issue_frac"], 100 * v["lds_busy_frac"],
        100 * v["l2_hit_rate"], v["l2_request_bytes_per_launch"] / v["accumulate_ms"] / 1e9,
        v["hbm_bytes_per_launch"] / 1e9, v["hbm_bytes_per_launch"] / v["accumulate_ms"] / 1e9))
