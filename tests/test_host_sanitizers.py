"""The round-6 host code under the sanitizers (CPU build only; the GPU pool has none): the optimised fast march and grid passes
against the statement-by-statement ones under AddressSanitizer + UndefinedBehaviorSanitizer (600 solves on random grids, 240
discretisations incl. 25 m grids), and the fast-marching solve cache -- hits, misses, eviction with storage reuse, the miss-streak
bypass -- under ThreadSanitizer with eight threads.  Sources: tests/host_sanitizers/*.cpp (they include the product's headers)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_sanitizers")
INC = os.path.join(ROOT, "kiwi_amd", "csrc")


def build_and_run(tmp_path, name, flags, env=None):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / name)
    cmd = [cxx, "-O1", "-g", "-std=c++17", "-fno-fast-math", "-ffp-contract=off", "-fno-omit-frame-pointer", "-pthread", "-I", INC] + flags + \
        ["-o", exe, os.path.join(SRC, name + ".cpp")]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and ("cannot find -l" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("this compiler has no %s runtime" % flags[0])
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    # a sanitizer runtime that cannot set up its shadow memory in this container says so before main() runs: not a finding
    if r.returncode != 0 and any(m in r.stderr for m in ("FATAL: ThreadSanitizer", "ReserveShadowMemoryRange failed", "Shadow memory range interleaves",
                                                         "unexpected memory mapping")):
        pytest.skip("the sanitizer runtime does not start here: " + r.stderr.strip().splitlines()[0][:120])
    return r


def test_march_and_grid_passes_under_asan_and_ubsan(tmp_path):
    r = build_and_run(tmp_path, "asan_host", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                      env={"ASAN_OPTIONS": "detect_leaks=0"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "asan host run: 0 bad" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, (r.stdout + r.stderr)[-3000:]


def test_solve_cache_under_tsan(tmp_path):
    r = build_and_run(tmp_path, "tsan_cache", ["-fsanitize=thread"])
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "tsan cache run: 0 bad" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout + r.stderr)[-3000:]
