"""The four GPU configurations of BASELINE.json at FULL size (the shapes `bench.py --workload cfgN` runs), small
batches: a few trial sources against the CPU oracle, plus properties that do not need the oracle --
the true source reproduces its references exactly, doubling the moment doubles every sample exactly
(power-of-two scaling commutes with every fp32 rounding on the path), negating a moment tensor negates them,
evaluating a batch in pieces gives the same bits, the misfit grows monotonically along a strike sweep."""
import os

import numpy as np
import pytest

import bench
from kiwi_amd import lib as _lib
from kiwi_amd import synthetic
from tests import common

pytestmark = pytest.mark.gpu
CORES = int(_lib.load().kiwi_hip_effective_cpus())      # hardware threads cut to the container's CPU quota


def rel(a, b, n=None):
    """largest difference of device misfits a from oracle misfits b, relative to the misfit -- under the fused arithmetic
    contract to max(misfit, norm factor n): a misfit is the norm of a difference of traces, its round-off scales with the traces"""
    scale = np.maximum(np.abs(b), 1e-300)
    if common.arith() == "fused" and n is not None:
        scale = np.maximum(scale, np.abs(n))
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / scale))


def reproduces(m, n, g):
    """the source the references were made from: exact arithmetic gives identical traces and a misfit of exactly zero; under the
    fused contract the kernel that kept the reference synthetics and the one with the comparator in its epilogue contract
    on their own: zero within 1e-6 of the norm factor"""
    if common.arith() == "exact":
        return bool(np.all(m == 0.0) and g == 0.0)
    return bool(np.all(m <= 1e-6 * n) and g <= 1e-6)


def setup(name, batch):
    wl = synthetic.workload(name, batch, 0)
    p, gf, recv, refs, tapers, ncent = bench.setup_product(0, wl, 4096)
    return wl, p, gf, recv, refs, tapers, ncent


def all_synthetics(p, isrc, nrec, which=1):
    return [p.get_synthetics(isrc, ir + 1, k + 1, which)[1] for ir in range(nrec) for k in range(3)]


def test_cfg3_bilateral_100_centroids_50_receivers():
    wl, p, gf, recv, refs, tapers, ncent = setup("cfg3", 24)
    assert ncent == 100 and wl["nrec"] == 50
    tr = wl["trials"].copy()
    tr[0] = wl["true"]                                   # slot 0: the source the references were made from
    tr[1] = wl["true"]; tr[1, 4] *= 2.0                  # slot 1: twice the moment
    p.set_keep_synthetics(1)
    p.set_source_params("bilateral", tr)
    p.eval()
    m, n, g = p.get_misfits()
    assert m.shape == (24, 150)
    assert reproduces(m[0], n[0], g[0])                  # identical traces -> exactly zero (exact arithmetic)
    s0, s1 = all_synthetics(p, 0, 50), all_synthetics(p, 1, 50)
    assert all(np.array_equal(2.0 * a, b) for a, b in zip(s0, s1))
    assert np.all(np.diff(g[2:12]) > 0)                  # strike sweep away from the true strike
    p.set_keep_synthetics(0)
    # oracle on three of the trial sources
    e, db, evaluate = bench.oracle_engine(wl, gf, recv, refs, tapers, CORES)
    for i in (2, 11, 23):
        om, on, og = evaluate(tr[i])
        assert rel(m[i], om, on) <= 1e-6 and common.misfit_close(g[i], og, glob=True) and np.array_equal(n[i], on)
    e.close(); db.close()
    # pieces == whole
    p.eval(0, 7); p.eval(7, 17)
    m2, _, g2 = p.get_misfits()
    assert np.array_equal(m, m2) and np.array_equal(g, g2)


@pytest.mark.parametrize("name,ncent_want", [("cfg3-100pt", 200), ("cfg3-scatter", 100), ("cfg3-ng8", 100), ("cfg3-static", 100)])
def test_cfg3_variants_at_full_size(name, ncent_want):
    """The literal "100 sub-faults" source (100 sub-fault points x 2 time steps), the cfg3 source over a shuffled
    location grid (no Green's function rows shared between neighbouring trials), over a far-field database of eight
    components and over a database with static end values and interior gaps -- full size, oracle spot checks."""
    wl, p, gf, recv, refs, tapers, ncent = setup(name, 20)
    assert ncent == ncent_want and wl["nrec"] == 50
    if name == "cfg3-100pt":
        assert wl["npoints"] == 100
    tr = wl["trials"].copy()
    tr[0] = wl["true"]
    p.set_source_params("bilateral", tr)
    p.eval()
    m, n, g = p.get_misfits()
    assert m.shape == (20, 150) and reproduces(m[0], n[0], g[0])
    if name == "cfg3-scatter":                              # the trials really are scattered: 4 km steps north / east
        assert len(np.unique(tr[1:, 1])) > 5 and len(np.unique(tr[1:, 2])) > 5 and np.all(g[1:] > 0.1)
    e, db, evaluate = bench.oracle_engine(wl, gf, recv, refs, tapers, CORES)
    for i in (1, 7, 19):
        om, on, og = evaluate(tr[i])
        assert rel(m[i], om, on) <= 1e-6 and common.misfit_close(g[i], og, glob=True) and np.array_equal(n[i], on)
    e.close(); db.close()


def test_cfg2_moment_tensor_grid():
    wl, p, gf, recv, refs, tapers, ncent = setup("cfg2", 512)
    tr = wl["trials"].copy()
    tr[0] = wl["true"]
    tr[1] = wl["true"]; tr[1, 4:10] *= 2.0
    tr[2] = wl["true"]; tr[2, 4:10] *= -1.0
    p.set_keep_synthetics(1)
    p.set_source_params("moment_tensor", tr)
    p.eval()
    m, n, g = p.get_misfits()
    assert reproduces(m[0], n[0], 0.0)
    s0, s1, s2 = (all_synthetics(p, i, 50) for i in range(3))
    assert all(np.array_equal(2.0 * a, b) for a, b in zip(s0, s1))
    assert all(np.array_equal(-a, b) for a, b in zip(s0, s2))
    p.set_keep_synthetics(0)
    e, db, evaluate = bench.oracle_engine(wl, gf, recv, refs, tapers, CORES)
    for i in (3, 100, 257, 511):
        om, on, og = evaluate(tr[i])
        assert rel(m[i], om, on) <= 1e-6 and common.misfit_close(g[i], og, glob=True)
    e.close(); db.close()


def test_cfg4_mt_eikonal_468_centroids_200_receivers():
    wl, p, gf, recv, refs, tapers, ncent = setup("cfg4", 4)
    assert 400 < ncent < 600 and wl["nrec"] == 200
    tr = wl["trials"].copy()
    tr[0] = wl["true"]
    tr[1] = wl["true"]; tr[1, 4] = 2.0                   # moment factor
    p.set_keep_synthetics(1)
    p.set_source_params("mt_eikonal", tr)
    p.eval()
    m, n, g = p.get_misfits()
    assert m.shape == (4, 600) and reproduces(m[0], n[0], 0.0)
    s0, s1 = all_synthetics(p, 0, 200), all_synthetics(p, 1, 200)
    assert all(np.array_equal(2.0 * a, b) for a, b in zip(s0, s1))
    p.set_keep_synthetics(0)
    e, db, evaluate = bench.oracle_engine(wl, gf, recv, refs, tapers, CORES)
    for i in (2, 3):
        om, on, og = evaluate(tr[i])
        assert rel(m[i], om, on) <= 1e-6 and common.misfit_close(g[i], og, glob=True) and np.array_equal(n[i], on)
    e.close(); db.close()


@pytest.mark.parametrize("name", ["cfg5", "cfg5-td"])
def test_cfg5_spectral_comparator_with_filter(name):
    """cfg5: amplitude-spectrum L2 with frequency filter; cfg5-td: the same trials under the time-domain L2 on filtered traces
    (transform forward, filter, transform back: comparator.f90:810-813,1224-1263) -- same tolerances."""
    wl, p, gf, recv, refs, tapers, ncent = setup(name, 16)
    tr = wl["trials"].copy()
    tr[0] = wl["true"]
    p.set_source_params("bilateral", tr)
    p.eval()
    m, n, g = p.get_misfits()
    # the true source against its own references: transform round-off only (the references went through the same kernels)
    assert np.all(m[0] <= 1e-5 * n[0])
    # Spectral tolerance: no fixed figure.  Per slot |m - m_oracle| <= 1e-6 max(n, m) + what fp32 transforms of this length explain
    # (tests/common.py fft_roundoff_bound: c eps log2N of the tapered traces' norms pushed through the norm; the oracle transforms
    # in fp64 -- FFTW's own rounding is "parity unpinned"), the same for the norm factors; the global misfit within
    # SURVEY 8c's 1e-5.  The transform length of every (source, slot) pair is what a fresh reference engine gives that source
    # (comparator.f90:222-271,464-486), hence a fresh oracle engine per source.
    e, db, evaluate = bench.oracle_engine(wl, gf, recv, refs, tapers, CORES, fresh=True)
    comps = recv[3]
    for i in (1, 9, 15):
        om, on, og, scales = evaluate(tr[i], inspect=lambda eng: common.slot_scales(eng, comps, gf["dt"]))
        ok, ratio = common.spectral_close(wl["method"], gf["dt"], m[i], om, on, scales, pn=n[i])
        r = np.abs(m[i] - om) / on
        assert ok and np.median(r) <= 5e-6, (i, ratio, np.median(r), r.max())
        assert abs(g[i] - og) <= 1e-5 * og
    e.close(); db.close()
    p.eval(0, 5); p.eval(5, 11)
    m2, _, g2 = p.get_misfits()
    assert np.array_equal(m, m2) and np.array_equal(g, g2)


def test_bench_line_through_rccl_with_one_rank():
    """bench.py under torch.distributed.run with the collective path forced on (KIWI_BENCH_FORCE_DIST): RCCL init bound to
    the device, barrier, all-gather of the global misfits, all-reduce of the elapsed time -- what the N > 1 runs do,
    exercised with the one GPU a test box has.  Checks the contract fields of the JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KIWI_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3",
                          "--warmup", "1", "--batch", "16", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                         env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert 0.0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["bound"] == "valu_issue"


@pytest.mark.parametrize("nranks", [4, 8])
def test_bench_dress_rehearsal_ranks_on_one_gpu(nranks):
    """What the driver runs unattended the first time a multi-GPU node exists -- `bench.py --gpus N` under torch.distributed.run --
    rehearsed with FOUR and with EIGHT ranks (the node size the scaling bench is run at) on the one GPU of a test box: every rank's engine on device 0 (KIWI_BENCH_DEVICE), the collective
    through gloo (RCCL wants a device per rank).  Exercises the launcher's environment, shard_range with N = 4, the split of the
    host cores between the ranks' discretiser teams (KIWI_HIP_DISC_THREADS), the barrier / max-over-ranks timing, the all-gather
    of the global misfits in trial order and the JSON contract; the gathered misfits equal a single rank's evaluation of the
    same 4 x batch trial sources."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KIWI_BENCH_BACKEND="gloo", KIWI_BENCH_DEVICE="0", KIWI_BENCH_DUMP_MISFITS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("KIWI_HIP_DISC_THREADS", None)
    batch = 24 if nranks == 4 else 12
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr",
                          "127.0.0.1", "--master-port", str(29546 + nranks), "--max-restarts", "0", os.path.join(root, "bench.py"), "--gpus", str(nranks),
                          "--steps", "2", "--warmup", "1", "--batch", str(batch), "--workload", "cfg3"], capture_output=True, text=True,
                         timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == nranks and d["rccl_world_size"] == nranks and d["steps"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["trial_sources_per_gpu_per_step"] == batch and d["value"] > 0
    assert "gloo" in d["collective"]
    got = np.array(d["gathered_global_misfits"], np.float32)
    assert got.shape == (96,)
    # one rank, the same 96 trial sources
    wl, p, gf, recv, refs, tapers, ncent = setup("cfg3", 96)
    p.eval()
    want = p.get_misfits()[2]
    assert common.same_bits(got, want)
    assert np.all(np.diff(got[1:]) != 0)                 # a strike sweep: every trial its own misfit, in trial order


@pytest.mark.parametrize("how", ["dies", "hangs"])
def test_bench_rehearsal_with_a_rank_that_fails(how):
    """The failure path of `bench.py --gpus N`: four rehearsal ranks on the one GPU, rank 2 leaves the run
    after the warm-up (exit 17) or stops answering (sleeps).  The launcher must come back NON-ZERO and soon -- torch.distributed.run
    ends the other ranks when one has failed; a rank that hangs is found by the collective's timeout (KIWI_BENCH_COLLECTIVE_TIMEOUT,
    here 15 s; default 120) in the ranks that wait for it -- and no JSON line may be printed.  Nothing is restarted in place."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KIWI_BENCH_BACKEND="gloo", KIWI_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", KIWI_BENCH_COLLECTIVE_TIMEOUT="15")
    env["KIWI_BENCH_FAIL_RANK" if how == "dies" else "KIWI_BENCH_HANG_RANK"] = "2"
    t0 = time.time()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                          "127.0.0.1", "--master-port", str(29560 + (how == "hangs")), "--max-restarts", "0", os.path.join(root, "bench.py"),
                          "--gpus", "4", "--steps", "2", "--warmup", "1", "--batch", "24", "--workload", "cfg3"], capture_output=True, text=True,
                         timeout=600, env=env, cwd=root)
    took = time.time() - t0
    assert out.returncode != 0, out.stdout[-1000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], out.stdout[-500:]
    assert ("leaves the run" if how == "dies" else "stops answering") in out.stderr
    assert took < 240, took


def test_database_beyond_two_to_the_31_samples(monkeypatch):
    """A Green's function tensor of 9.3 GB (2.3e9 floats: ordinary for real Kiwi databases on a 288 GB device): the
    LDS-staged kernels address a group's rows with 32-bit offsets relative to a 64-bit per-group base, so sources whose
    nodes sit beyond 2^31 floats must give the same synthetics as the direct kernel (which indexes with size_t), bit for
    bit.  Only the traces around the nodes in use are stored; the host array is untouched zero pages elsewhere."""
    from kiwi_amd import Engine
    nx, nz, ng, L = 1100, 50, 10, 4096
    dt, dx, dz, firstx, firstz = 0.5, 1000.0, 500.0, 100e3, 2e3
    data = np.zeros((nx, nz, ng, L), np.float32)
    first = np.zeros((nx, nz, ng), np.int32)
    nsamp = np.zeros((nx, nz, ng), np.int32)
    rng = np.random.default_rng(31)
    ix0, iz0 = 1070, 8                                    # rows from ((1070 * 50 + 8) * 10) * 4224 = 2.26e9 floats on
    sl = (slice(ix0, ix0 + 24), slice(iz0, iz0 + 16))
    i = np.arange(L)
    for a in range(ix0, ix0 + 24):
        for b in range(iz0, iz0 + 16):
            for g in range(ng):
                data[a, b, g] = (1e-20 * np.sin(0.02 * i * (1 + 0.05 * g) + 0.3 * g + 0.1 * b) *
                                 np.exp(-((i - 700 - 30 * g) / 350.0) ** 2) * (1 + 0.01 * (a - ix0))).astype(np.float32)
    first[sl] = 300
    nsamp[sl] = L
    x = firstx + (ix0 + 10.3) * dx
    lat, lon, depth, comps, dist = synthetic.make_receivers(6, dmin=x, dspan=3000.0)
    trials = synthetic.bilat_strike_sweep(3, step=5.0)
    trials[:, 3] = firstz + (iz0 + 7.4) * dz
    res = {}
    for mode in ("grouped", "cell", "direct"):
        monkeypatch.delenv("KIWI_HIP_ACCUM", raising=False)
        monkeypatch.delenv("KIWI_HIP_CELL", raising=False)
        if mode == "direct":
            monkeypatch.setenv("KIWI_HIP_ACCUM", "direct")
        if mode == "cell":
            monkeypatch.setenv("KIWI_HIP_CELL", "1")
        p = Engine(0)
        p.set_database(dt, dx, dz, firstx, firstz, data, first, nsamp)
        assert p.device_bytes() > 9.0e9
        p.set_receivers(lat, lon, depth, comps)
        p.set_source_location(40.0, 30.0, 0.0)
        p.set_effective_dt(0.5)
        p.set_local_interpolation("bilinear")
        p.set_source_params("bilateral", trials)
        p.set_keep_synthetics(1)
        p.eval()
        rows = p.get_geometry(0, 1)["row"]
        assert rows.min() >= 0 and int(rows.min()) * 4224 > 2 ** 31
        res[mode] = [p.get_synthetics(s, ir, k, 1)[1] for s in range(3) for ir in range(1, 7) for k in (1, 2, 3)]
        p.close()
    assert any(np.any(a != 0) for a in res["direct"])
    for mode in ("grouped", "cell"):
        for a, b in zip(res[mode], res["direct"]):
            assert a.tobytes() == b.tobytes(), mode


_TWO_RANK_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["KIWI_ROOT"])
import torch
import torch.distributed as dist
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
backend = os.environ["KIWI_TEST_BACKEND"]
if backend == "nccl":                                   # one GPU per rank, RCCL
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
else:                                                   # one-GPU box: both ranks' engines on device 0, gloo carries the gather
    local = 0
    dist.init_process_group("gloo", rank=rank, world_size=world)
from kiwi_amd import synthetic
from tests import common
from kiwi_amd.shard import sharded_misfits_for_sources
from tests.common import Scenario
from tests.test_gpu_parity import build
out = {}
for method, with_filter in (("l2norm", False), ("ampspec_l2norm", False), ("l2norm", True)):
    sc = Scenario()
    e = sc.oracle(); sc.make_references(e); sc.apply_setup(e, True)
    p = sc.product(local); sc.apply_setup(p, False)
    p.set_misfit_method(method)
    if with_filter:
        for ir in range(1, sc.nrec + 1):
            p.set_misfit_filter(ir, [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.])
    trials = np.array([[0.3 * i, 0., 0., 9500. + 300 * i] + synthetic.mt_from_sdr(40. * i, 50. + 5 * i, -60. + 30 * i) + [1.0]
                       for i in range(7)], np.float32)
    trials[1, 10] = 150.0; trials[5, 10] = 170.0          # two transform lengths in the list, one in each shard
    m, n, fails = sharded_misfits_for_sources(p, "moment_tensor", trials, dist, local)
    out[method + ("+filter" if with_filter else "")] = (m, n)
    if rank == 0:
        m1, n1, f1 = p.make_misfits_for_sources("moment_tensor", trials)       # unsharded, this rank alone
        assert fails == f1 == []
        assert m.shape == m1.shape and m.tobytes() == m1.tobytes() and n.tobytes() == n1.tobytes(), (method, with_filter)
        assert len({x.tobytes() for x in n1}) > (1 if method != "l2norm" or with_filter else 0)
dist.barrier()
if rank == 0:
    print("TWO_RANK_OK world=%d" % dist.get_world_size())
dist.destroy_process_group()
"""


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_two_ranks_sharded_equals_unsharded(tmp_path, backend):
    """Trial sources sharded over TWO ranks (one process per rank, all-gather of the per-receiver misfits,
    kiwi_amd/shard.py) == the same list evaluated by one rank, bit for bit -- time-domain L2, amplitude-spectrum L2 and
    filtered L2 (whose transform lengths must not depend on which sources share a rank).  "nccl": one GPU per rank over
    RCCL, needs two devices; "gloo": the same two processes with their engines on one GPU, runs on any GPU box."""
    import os
    import subprocess
    import sys
    import torch
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (this box has %d)" % torch.cuda.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two_rank_worker.py"
    script.write_text(_TWO_RANK_WORKER)
    env = dict(os.environ, KIWI_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0", KIWI_TEST_BACKEND=backend)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29543" if backend == "nccl" else "29544", "--max-restarts", "0", str(script)],
                         capture_output=True, text=True,
                         timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "TWO_RANK_OK world=2" in out.stdout


def test_bench_gpus_flag_is_honoured():
    """`python bench.py --gpus N` starts N ranks itself (the driver's command shape); with fewer devices than asked for it
    fails loudly instead of silently measuring one GPU; a launcher/flag mismatch is an error too."""
    import json
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ndev = torch.cuda.device_count()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ndev + 1), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode != 0 and "only %d GPU" % ndev in out.stderr and not out.stdout.strip()
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
    if ndev >= 2:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64"],
                             capture_output=True, text=True, timeout=900, cwd=root)
        assert out.returncode == 0, out.stderr[-3000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2 and d["cpu_baseline"] is None


def test_multi_device_context_equals_one_device(tmp_path, monkeypatch):
    """kiwi_hip_init_multi: ONE context over several devices of this process -- setters repeated on every device, the trial
    list of kiwi_hip_misfits_for_params cut into contiguous shards, each evaluated on its device by a thread of its own into its
    slice of the caller's arrays.  Byte-identical to one device for time-domain L2, amplitude-spectrum L2 and filtered L2
    (two transform lengths in the list), with a failing trial in the list; through the Python engine and through the Fortran
    protocol host (KIWI_HIP_NDEV=2, eval_sources).  Real devices where the box has them, else both contexts on device 0
    (KIWI_HIP_MULTI_OVERSUBSCRIBE=1)."""
    import torch
    from kiwi_amd import Engine, protocol
    from tests.common import Scenario
    if torch.cuda.device_count() < 2:
        monkeypatch.setenv("KIWI_HIP_MULTI_OVERSUBSCRIBE", "1")
    for method, with_filter in (("l2norm", False), ("ampspec_l2norm", False), ("l2norm", True)):
        sc = Scenario()
        e = sc.oracle(); sc.make_references(e); sc.apply_setup(e, True)
        res = []
        for ndev in (None, 2, 3):
            g = sc.gf
            first, nsamp, data = sc.odb.dense_tables()
            p = Engine(0) if ndev is None else Engine(ndev=ndev)
            assert p.ndevices() == (ndev or 1)
            p.set_database(g["dt"], g["dx"], g["dz"], g["firstx"], g["firstz"], data, first, nsamp)
            p.set_receivers(sc.lat, sc.lon, sc.depth, sc.comps)
            p.set_source_location(40.0, 30.0, 0.0)
            p.set_effective_dt(sc.effective_dt)
            p.set_local_interpolation("bilinear")
            sc.apply_setup(p, False)
            p.set_misfit_method(method)
            if with_filter:
                for ir in range(1, sc.nrec + 1):
                    p.set_misfit_filter(ir, [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.])
            trials = np.array([[0.3 * i, 0., 0., 9500. + 300 * i] + synthetic.mt_from_sdr(40. * i, 50. + 5 * i, -60. + 30 * i) + [1.0]
                               for i in range(7)], np.float32)
            trials[1, 10] = 150.0; trials[5, 10] = 170.0          # two transform lengths in the list
            m, n, gl, st = p.misfits_for_params("moment_tensor", trials, 2)
            # a setter forwarded to the other devices, then getters on the owning context: the owner's device must be the
            # current one again (the forward left the LAST device current, prepare() then allocated the owner's tables
            # there -- invisible with both contexts stacked on one device, a fault on two)
            if with_filter:
                for ir in range(1, sc.nrec + 1):
                    p.set_misfit_filter(ir, [0.01, 0.03, 0.25, 0.4], [0., 1., 1., 0.])
                lo_f, syn_f = p.get_synthetics(0, 1, 1, 3)
                p.eval(0, 1)
                gm2 = p.get_misfits(0, 1)[0]
                assert len(syn_f) > 10 and np.all(np.isfinite(syn_f)) and np.array_equal(gm2[0], m[0])
            res.append((m, n, gl, st))
            p.close()
        for other in res[1:]:
            for a, b in zip(res[0], other):
                assert np.asarray(a).tobytes() == np.asarray(b).tobytes(), (method, with_filter)
        assert np.all(res[0][2] > 0)
    # ---- the Fortran protocol host over two devices
    if not common.HAVE_FLANG:
        return
    sc = Scenario(nrec=4)
    e = sc.oracle(); sc.make_references(e); sc.apply_setup(e, True)
    gf = dict(sc.gf)
    first, nsamp, data = sc.odb.dense_tables()
    gf.update(first=first, nsamp=nsamp, data=data)
    base = str(tmp_path / "db")
    protocol.write_flat_gfdb(base, gf)
    protocol.write_receivers(str(tmp_path / "receivers.table"), sc.lat, sc.lon, sc.comps)
    dt = gf["dt"]
    for (ir, k), (lo, d) in sc.refs.items():
        protocol.write_table(str(tmp_path / ("ref-%d-%s.table" % (ir, sc.comps[ir - 1][k - 1]))), (lo - 1) * dt, dt, d)
    trials = synthetic.bilat_strike_sweep(9, step=1.5)
    pf = tmp_path / "params.txt"
    with open(pf, "w") as f:
        for t in trials:
            f.write(" ".join("%.9g" % v for v in t) + "\n")
    outs = []
    for ndev in (None, "2"):
        env = dict(os.environ)
        if ndev:
            env["KIWI_HIP_NDEV"] = ndev
        p = protocol.MinimizerProcess(env=env)
        try:
            p.do("set_database", base)
            p.do("set_effective_dt", sc.effective_dt)
            p.do("set_local_interpolation", "bilinear")
            p.do("set_receivers", str(tmp_path / "receivers.table"))
            p.do("set_source_location", 40.0, 30.0, 0.0)
            p.do("set_ref_seismograms", str(tmp_path / "ref"), "table")
            p.do("set_misfit_method", "l2norm")
            for ir, (x, y) in sc.tapers.items():
                p.do("set_misfit_taper", ir, *[v for xy in zip(x, y) for v in xy])
            out = str(tmp_path / ("out%s.txt" % (ndev or "1")))
            assert p.do("eval_sources", "bilateral", str(pf), out) == "9"
            outs.append(open(out, "rb").read())
        finally:
            p.close()
    assert outs[0] == outs[1] and len(outs[0]) > 500


def test_bench_blocks_host_inclusive_and_sweep():
    """The blocks round 5 added to bench.py's default line, at test size: `host_inclusive` (the timed trial sources through the one-call
    evaluation: discretiser + transfers + kernels; bit-identical to the resident evaluation) and `sweep` (a slice of configuration 5's
    10^5-point grid around the planted source, host-inclusive: the argmin is the planted source)."""
    wl, p, gf, recv, refs, tapers, ncent = setup("cfg3", 96)
    p.eval()
    p.sync()
    h = bench.host_inclusive(p, wl, 1000.0, reps=1, piece=32)
    assert h["identical_to_resident"] and h["failed_sources"] == 0 and h["trial_sources_per_step"] == 96 and h["value"] > 0
    assert h["frac_of_resident"] == pytest.approx(h["value"] / 1000.0)
    p.close()
    s = bench.sweep_block(0, 4096, n=700, first=50200)          # the planted source is point 50 555 of the grid
    assert s["true_source_index"] == 355 and s["argmin_is_true_source"] and s["argmin_misfit"] <= 1e-6 and s["failed_sources"] == 0
    assert s["trial_sources"] == 700 and s["value"] > 0
