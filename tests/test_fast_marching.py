"""The host's fast-marching solve (kiwi_amd/csrc/kiwi_host_fmm.hpp through kiwi_hip_fast_marching; needs no GPU).

Two routines live there: the reference's statements one by one (`plain`) and the layout-optimised march the eikonal
discretiser uses.  Both must give the arrival times of the oracle's restatement (oracle/ko_eikonal.c, pinned bit for bit on
the reference's eikonal_solver_fmm by tests/test_oracle_eikonal.py and on its outputs by tests/golden/eikonal_vectors.npz)
-- including the order in which heap.f90 accepts nodes of EQUAL arrival time, which uniform speed fields on square cells
produce in quantity (fourfold and eightfold symmetric fronts)."""
import ctypes as C

import numpy as np
import pytest

from oracle import ko
from kiwi_amd import lib as klib

NAN = float("nan")


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def product(speed, origin, delta, start, discard=NAN, plain=0):
    ny, nx = speed.shape
    t = np.zeros((ny, nx), np.float32)
    fb = C.c_longlong(0)
    rc = klib.load().kiwi_hip_fast_marching(fp(speed), nx, ny, fp(origin), fp(delta), fp(start), discard, plain, fp(t), C.byref(fb))
    assert rc == 0
    return t, fb.value


def oracle(speed, origin, delta, start):
    ny, nx = speed.shape
    t = np.zeros((ny, nx), np.float32)
    ko.lib().ko_eikonal_solver_fmm(fp(speed), C.c_int(nx), C.c_int(ny), fp(origin), fp(delta), fp(start), fp(t))
    return t


def fields(rng, kind, nx, ny):
    if kind == "layered":           # psm_make_*_grid: layers down dip, an outside region at half the slowest speed
        layers = np.sort(rng.choice([2100., 2600., 3200., 3500., 3900.], 3)) * np.float32(rng.uniform(0.6, 1.0))
        s = np.zeros((ny, nx), np.float32)
        for iy in range(ny):
            s[iy, :] = layers[min(2, (3 * iy) // max(ny, 1))]
        yy, xx = np.mgrid[0:ny, 0:nx]
        out = (xx - nx / 2.) ** 2 / max(nx / 2., 1) ** 2 + (yy - ny / 2.) ** 2 / max(ny / 2., 1) ** 2 > 1.0
        if out.all():
            out[:] = False
        s[out] = np.float32(s[~out].min() * np.float32(0.5))
        return s
    if kind == "uniform":           # equal keys by the hundred
        return np.full((ny, nx), np.float32(3000.0 * rng.uniform(0.5, 1.0)), np.float32)
    if kind == "blocks":            # sharp edges: the fall-back branch of update_neighbor (eikonal.f90:170-177)
        s = np.zeros((ny, nx), np.float32)
        for by in range(0, ny, 5):
            for bx in range(0, nx, 7):
                s[by:by + 5, bx:bx + 7] = np.float32(rng.choice([800., 1500., 3000., 6000.]))
        return s
    raise ValueError(kind)


CASES = [(kind, seed) for kind in ("layered", "uniform", "blocks") for seed in range(14)]


@pytest.mark.parametrize("kind,seed", CASES)
def test_both_marches_give_the_oracles_times(kind, seed):
    rng = np.random.default_rng(1000 * len(kind) + seed)
    nx = 1 if seed == 0 else int(rng.integers(2, 90))
    ny = 1 if seed == 1 else int(rng.integers(2, 70))
    if seed == 2:
        nx, ny = 1, 1
    speed = fields(rng, kind, nx, ny)
    origin = rng.uniform(-5000, 0, 2).astype(np.float32)
    d = np.float32(rng.uniform(100, 900))
    delta = np.array([d, d], np.float32) if kind == "uniform" else rng.uniform(100, 900, 2).astype(np.float32)
    # start point: centre of a cell for the uniform fields (symmetric fronts), anywhere otherwise -- outside of the grid too
    if kind == "uniform":
        start = (origin + (np.array([rng.integers(0, nx), rng.integers(0, ny)]) + 0.5) * delta).astype(np.float32)
    else:
        start = (origin + rng.uniform(-0.2, 1.2, 2) * delta * [nx, ny]).astype(np.float32)
    want = oracle(speed, origin, delta, start)
    plain, _ = product(speed, origin, delta, start, plain=1)
    fast, _ = product(speed, origin, delta, start, plain=0)
    assert np.array_equal(plain.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(fast.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("seed", range(6))
def test_early_termination_keeps_every_time_that_is_kept(seed):
    """`discard` = the speed of the points outside of the rupture: the march may stop once every other node is accepted.  The
    times of all nodes of another speed are those of the full solve, in both routines (the partial solve against
    the full one on all non-discard nodes)."""
    rng = np.random.default_rng(500 + seed)
    nx, ny = int(rng.integers(20, 120)), int(rng.integers(20, 80))
    speed = fields(rng, "layered", nx, ny)
    origin = np.array([-1000., -700.], np.float32)
    delta = rng.uniform(20, 60, 2).astype(np.float32)
    start = (origin + rng.uniform(0.3, 0.7, 2) * delta * [nx, ny]).astype(np.float32)
    dis = float(speed.min())
    keep = speed != np.float32(dis)
    want = oracle(speed, origin, delta, start)
    for plain in (0, 1):
        got, _ = product(speed, origin, delta, start, discard=dis, plain=plain)
        assert np.array_equal(got[keep].view(np.uint32), want[keep].view(np.uint32))
    full, _ = product(speed, origin, delta, start, plain=0)
    assert np.array_equal(full.view(np.uint32), want.view(np.uint32))


def test_inputs_outside_of_the_state_encoding_go_to_the_plain_routine():
    """The optimised march keeps `accepted` in the sign bit of a time and `far away` as the value `infinity`
    (0.1 * huge, eikonal.f90:56): a time that is negative, NaN or >= `infinity` cannot be held.  Such solves (a speed of zero,
    a negative speed) are handed to the plain routine -- counted -- and come out as the plain routine alone gives them."""
    speed = np.full((6, 9), 2000.0, np.float32)
    speed[2, 3] = 0.0                                  # dx / 0 = +Inf next to it
    speed[4, 6] = -1500.0                              # negative times
    origin = np.zeros(2, np.float32)
    delta = np.array([100., 120.], np.float32)
    start = np.array([450., 350.], np.float32)
    plain, fb0 = product(speed, origin, delta, start, plain=1)
    fast, fb1 = product(speed, origin, delta, start, plain=0)
    assert fb1 == fb0 + 1
    assert np.array_equal(plain.view(np.uint32), fast.view(np.uint32))
    ok = np.full((6, 9), 2000.0, np.float32)
    _, fb2 = product(ok, origin, delta, start, plain=0)
    assert fb2 == fb1


def test_cfg4_sized_grid_against_the_plain_routine():
    """1200 x 360 nodes of 25 m (BASELINE config 4's rupture at effective_dt 0.5 s): two layers, the bounding circle's outside
    at half speed, early termination -- the optimised march against the plain one on every node that is kept."""
    nx, ny = 1200, 360
    yy, xx = np.mgrid[0:ny, 0:nx]
    depth = 6500.0 + (yy + 0.5) * 25.0
    speed = np.where(depth <= 12000.0, 3500.0, 3700.0).astype(np.float32) * np.float32(0.9)
    out = ((xx + 0.5 - nx / 2) * 25.0) ** 2 + (depth - 11000.0) ** 2 > 15000.0 ** 2
    dis = np.float32(speed.min() * np.float32(0.5))
    speed[out] = dis
    origin = np.array([-15000., -4500.], np.float32)
    delta = np.array([25., 25.], np.float32)
    start = np.array([2300., -1000.], np.float32)
    plain, _ = product(speed, origin, delta, start, discard=float(dis), plain=1)
    fast, _ = product(speed, origin, delta, start, discard=float(dis), plain=0)
    keep = ~out
    assert np.array_equal(plain[keep].view(np.uint32), fast[keep].view(np.uint32))
    assert np.array_equal(plain.view(np.uint32), fast.view(np.uint32))      # (the unfinished ones too: same sequence of steps)
