"""Eikonal sources against golden vectors generated from the reference's own modules
(tests/golden/make_golden_eikonal.py): the oracle restatement and the product's host discretiser
(kiwi_amd/csrc/kiwi_host_eikonal.hpp through kiwi_hip_discretize_eikonal) must both reproduce the
reference's centroid tables bit for bit.  Runs without /root/reference and without a GPU."""
import os

import numpy as np
import pytest

from oracle import ko
from kiwi_amd import engine as ke
from kiwi_amd.lib import KiwiHipError

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eikonal_vectors.npz"))
N = int(G["n"])


def oracle_profile(v):
    return ko.crust_profile(v[0:8], v[8:16], v[16:24], v[24:31])


def test_default_constraints():
    thick = ko.crust_thickness(oracle_profile(G["origin_profile"]))
    for k in range(N):
        cp, cn = G["e%d_con" % k]
        limit = float(G["e%d_limit" % k])
        assert np.array_equal(cp[0], [0, 0, 1500]) and np.array_equal(cn[0], [0, 0, -1]) and np.array_equal(cn[1], [0, 0, 1])
        assert cp[1][2] == np.float32(min(limit, thick) if limit > 0 else thick)


@pytest.mark.parametrize("k", range(N))
def test_oracle_matches_reference(k):
    cp, cn = G["e%d_con" % k]
    a, mo, ri, _ = ko.discretize_eikonal(int(G["e%d_type" % k]), G["e%d_params" % k], float(G["e%d_edt" % k]),
                                         oracle_profile(G["rupture_profile"]), cp, cn)
    assert np.array_equal(a.view(np.uint32), G["e%d_cent" % k].view(np.uint32))
    assert np.array_equal(np.array([mo, ri], np.float32), G["e%d_mr" % k])


@pytest.mark.parametrize("k", range(N))
def test_product_matches_reference(k):
    cp, cn = G["e%d_con" % k]
    a, mo, ri = ke.discretize_eikonal(int(G["e%d_type" % k]), G["e%d_params" % k], float(G["e%d_edt" % k]),
                                      G["rupture_profile"], cp, cn)
    assert np.array_equal(a.view(np.uint32), G["e%d_cent" % k].view(np.uint32))
    assert np.array_equal(np.array([mo, ri], np.float32), G["e%d_mr" % k])


def test_failures_reported_like_the_reference():
    p = G["fail_empty_params"]
    cp, cn = G["e0_con"]
    with pytest.raises(ValueError, match="Empty rupture area"):
        ko.discretize_eikonal(5, p, 1.0, oracle_profile(G["rupture_profile"]), cp, cn)
    with pytest.raises(KiwiHipError, match="Empty rupture area"):
        ke.discretize_eikonal(5, p, 1.0, G["rupture_profile"], cp, cn)
    q = G["e%d_params" % (N - 1)].copy()            # an mt_eikonal case: nucleation point moved out of the circle
    q[10] = 3 * q[9]
    with pytest.raises(ValueError, match="nucleation"):
        ko.discretize_eikonal(5, q, 1.0, oracle_profile(G["rupture_profile"]), cp, cn)
    with pytest.raises(KiwiHipError, match="nucleation"):
        ke.discretize_eikonal(5, q, 1.0, G["rupture_profile"], cp, cn)
    with pytest.raises(KiwiHipError):
        ke.discretize_eikonal(5, q[:5], 1.0, G["rupture_profile"], cp, cn)


def test_product_matches_oracle_on_random_cases():
    rng = np.random.default_rng(77)
    prof = G["rupture_profile"]
    cp, cn = G["e0_con"]
    done = 0
    for i in range(60):
        st = 4 + i % 2
        common = [rng.uniform(-1, 1), rng.uniform(-3e3, 3e3), rng.uniform(-3e3, 3e3), rng.uniform(2e3, 3e4)]
        strike, dip = rng.uniform(-180, 180), rng.uniform(0, 90)
        bord = [rng.uniform(-2e3, 2e3), rng.uniform(-2e3, 2e3), rng.uniform(5e2, 9e3)]
        nukl = [rng.uniform(-1, 1) * 0.7 * bord[2], rng.uniform(-1, 1) * 0.7 * bord[2]]
        relv, rise = rng.uniform(0.5, 1.0), rng.uniform(0, 3)
        if st == 5:
            p = common + [1.0, strike, dip] + bord + nukl + [relv] + list(rng.standard_normal(6)) + [rise]
        else:
            p = common + [1e18, strike, dip, rng.uniform(-180, 180)] + bord + nukl + [relv, rise]
        edt = float(rng.choice([1.0, 2.0, 4.0]))
        try:
            a, mo, ri, _ = ko.discretize_eikonal(st, p, edt, oracle_profile(prof), cp, cn)
        except ValueError as e:
            with pytest.raises(KiwiHipError, match=str(e)[:12]):
                ke.discretize_eikonal(st, p, edt, prof, cp, cn)
            continue
        b, bmo, bri = ke.discretize_eikonal(st, p, edt, prof, cp, cn)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and (mo, ri) == (bmo, bri)
        done += 1
    assert done >= 20


def _cache_stats(reset=False):
    import ctypes as C
    from kiwi_amd import lib as klib
    h, m = C.c_longlong(0), C.c_longlong(0)
    klib.load().kiwi_hip_eikonal_cache_stats(C.byref(h), C.byref(m), 1 if reset else 0)
    return h.value, m.value


def test_solve_cache_is_exact_and_hits_on_shifted_sources():
    """The fast-marching solves are kept by their complete inputs (speed grid, spacing, start cell): a rupture shifted north /
    east / in time, or given another moment tensor, takes the stored arrival times -- and its centroid table is bit for bit
    what the oracle (which solves every time) and an un-cached product build (KIWI_HIP_EIK_CACHE=0, child process) give.
    A depth change crosses layer boundaries: other speed grid, another solve."""
    import subprocess
    import sys
    prof = G["rupture_profile"]
    cp, cn = G["e0_con"]
    base = np.array(G["e%d_params" % (N - 1)], np.float32)             # an mt_eikonal case of the golden set
    assert int(G["e%d_type" % (N - 1)]) == 5
    edt = float(G["e%d_edt" % (N - 1)])
    trials = []
    for dn in (0.0, 400.0, -800.0):
        for de in (0.0, 250.0):
            for dtm in (0.0, 0.7):
                p = base.copy()
                p[1] += dn; p[2] += de; p[0] += dtm
                p[13:19] *= 1.0 + 0.1 * len(trials)                     # another moment tensor
                trials.append(p)
    deeper = base.copy()
    deeper[3] += 2500.0
    trials.append(deeper)
    _cache_stats(reset=True)
    tables = []
    for p in trials:
        a, mo, ri, _ = ko.discretize_eikonal(5, p, edt, oracle_profile(prof), cp, cn)
        b, bmo, bri = ke.discretize_eikonal(5, p, edt, prof, cp, cn)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and (mo, ri) == (bmo, bri)
        tables.append(b)
    hits, misses = _cache_stats()
    assert misses <= 3 and hits >= len(trials) - 3, (hits, misses)       # one solve for the twelve shifted ones, one for the deeper
    # the same list through a build with the cache switched off: identical bytes
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from kiwi_amd import engine as ke; "
            "G = np.load(%r); T = np.load(sys.argv[1]); cp, cn = G['e0_con']; "
            "out = [ke.discretize_eikonal(5, p, %r, G['rupture_profile'], cp, cn)[0] for p in T]; "
            "np.save(sys.argv[2], np.concatenate([o.ravel() for o in out]))") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                os.path.join(os.path.dirname(__file__), "golden", "eikonal_vectors.npz"), edt)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "t.npy"), np.array(trials, np.float32))
        env = dict(os.environ, KIWI_HIP_EIK_CACHE="0")
        subprocess.check_call([sys.executable, "-c", code, os.path.join(td, "t.npy"), os.path.join(td, "o.npy")], env=env)
        ref = np.load(os.path.join(td, "o.npy"))
    got = np.concatenate([t.ravel() for t in tables])
    assert got.view(np.uint32).tobytes() == ref.view(np.uint32).tobytes()


def _random_ruptures(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        st = 4 + i % 2
        common = [rng.uniform(-1, 1), rng.uniform(-3e3, 3e3), rng.uniform(-3e3, 3e3), rng.uniform(2e3, 3e4)]
        strike, dip = rng.uniform(-180, 180), rng.uniform(0, 90)
        bord = [rng.uniform(-2e3, 2e3), rng.uniform(-2e3, 2e3), rng.uniform(5e2, 9e3)]
        nukl = [rng.uniform(-1, 1) * 0.7 * bord[2], rng.uniform(-1, 1) * 0.7 * bord[2]]
        relv, rise = rng.uniform(0.5, 1.0), rng.uniform(0, 3)
        if st == 5:
            p = common + [1.0, strike, dip] + bord + nukl + [relv] + list(rng.standard_normal(6)) + [rise]
        else:
            p = common + [1e18, strike, dip, rng.uniform(-180, 180)] + bord + nukl + [relv, rise] + [0.0] * 5
        out.append([st, float(rng.choice([0.5, 1.0, 2.0, 4.0]))] + p)
    return np.array(out, np.float64)


def test_optimised_discretiser_equals_the_plain_statements_on_200_random_ruptures():
    """Round 6 rewrote the host discretiser's three passes over the fine grid -- the fast-marching solve on its own layout
    (kiwi_host_fmm.hpp), the speed grid and the binning of the arrival times four points at a time (kiwi_host_eikonal.hpp) --
    and kept the statement-by-statement versions behind KIWI_HIP_EIK_PLAIN=1.  200 random `eikonal` / `mt_eikonal` ruptures
    (any strike and dip, circles clipped by the default constraints, nucleation points on and off the cell centres, four
    effective dt), each discretised by both in its own process: identical centroid tables, moments and rise times, identical
    rejections."""
    import subprocess
    import sys
    import tempfile
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from kiwi_amd import engine as ke; from kiwi_amd.lib import KiwiHipError; "
            "G = np.load(%r); T = np.load(sys.argv[1]); cp, cn = G['e0_con']; out = []\n"
            "for row in T:\n"
            "    st = int(row[0]); p = row[2:2 + (20 if st == 5 else 15)].astype(np.float32)\n"
            "    try:\n"
            "        a, mo, ri = ke.discretize_eikonal(st, p, float(row[1]), G['rupture_profile'], cp, cn)\n"
            "        out.append(np.concatenate([[len(a), mo, ri], a.ravel()]).astype(np.float32))\n"
            "    except KiwiHipError as e:\n"
            "        out.append(np.array([-1.0, float(len(str(e)))], np.float32))\n"
            "np.save(sys.argv[2], np.concatenate(out))") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                os.path.join(os.path.dirname(__file__), "golden", "eikonal_vectors.npz"))
    trials = _random_ruptures(200, 20261004)
    res = {}
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "t.npy"), trials)
        for mode in ("0", "1"):
            env = dict(os.environ, KIWI_HIP_EIK_PLAIN=mode, KIWI_HIP_EIK_CACHE="0")
            subprocess.check_call([sys.executable, "-c", code, os.path.join(td, "t.npy"), os.path.join(td, "o%s.npy" % mode)], env=env)
            res[mode] = np.load(os.path.join(td, "o%s.npy" % mode))
    assert res["0"].size > 200 * 10
    assert res["0"].view(np.uint32).tobytes() == res["1"].view(np.uint32).tobytes()


GB = np.load(os.path.join(os.path.dirname(__file__), "golden", "eikonal_vectors_big.npz"))


@pytest.mark.parametrize("k", range(int(GB["n"])))
def test_reference_tables_at_the_resolution_of_config_4(k):
    """Ruptures of 8 .. 15 km radius at effective dt 0.5 s -- the 25 m fine grid of BASELINE config 4, 10^5 .. 1.4 10^6 points per
    fast-marching solve, centroid tables of up to 5000 rows -- discretised by the reference's own modules
    (tests/golden/make_golden_eikonal_big.py).  The oracle restatement and the product's host discretiser (round 6: the march on
    its own layout, the grid passes four points at a time; and their statement-by-statement versions, in a child process) reproduce
    those tables bit for bit."""
    import subprocess
    import sys
    cp, cn = GB["e%d_con" % k]
    st, p, edt = int(GB["e%d_type" % k]), GB["e%d_params" % k], float(GB["e%d_edt" % k])
    want = GB["e%d_cent" % k]
    a, mo, ri, _ = ko.discretize_eikonal(st, p, edt, oracle_profile(GB["rupture_profile"]), cp, cn)
    assert np.array_equal(a.view(np.uint32), want.view(np.uint32))
    b, bmo, bri = ke.discretize_eikonal(st, p, edt, GB["rupture_profile"], cp, cn)
    assert np.array_equal(b.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(np.array([bmo, bri], np.float32), GB["e%d_mr" % k])
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from kiwi_amd import engine as ke; G = np.load(%r); k = int(sys.argv[1]); "
            "cp, cn = G['e%%d_con' %% k]; a, mo, ri = ke.discretize_eikonal(int(G['e%%d_type' %% k]), G['e%%d_params' %% k], float(G['e%%d_edt' %% k]), "
            "G['rupture_profile'], cp, cn); sys.exit(0 if np.array_equal(a.view(np.uint32), G['e%%d_cent' %% k].view(np.uint32)) else 3)") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                os.path.join(os.path.dirname(__file__), "golden", "eikonal_vectors_big.npz"))
    env = dict(os.environ, KIWI_HIP_EIK_PLAIN="1")
    assert subprocess.call([sys.executable, "-c", code, str(k)], env=env) == 0
